// press_packed.h - packed 16-bit helpers of the sample-value kernels (press_chunked.hip, press_huffman.hip):
// two samples per register, zig-zag, running sums.  Device code only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ph {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// inclusive wave scan with DPP (row_shr 1,2,4,8 inside rows of 16, then row_bcast 15 / 31)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v)
{
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, true); // row_shr:1
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, true); // row_shr:2
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, true); // row_shr:4
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, true); // row_shr:8
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1,3
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2,3
	return v;
}

// inverse zig-zag of two packed 16-bit values (trans.c:80)
__device__ __forceinline__ uint32_t unzz_pair(uint32_t z)
{
	const u16x2 v = __builtin_bit_cast(u16x2, z);
	const u16x2 one = { 1, 1 };
	const u16x2 zero = { 0, 0 };
	const u16x2 r = (v >> one) ^ (zero - (v & one));
	return __builtin_bit_cast(uint32_t, r);
}

__device__ __forceinline__ uint32_t pk_add16(uint32_t a, uint32_t b)
{
	return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b));
}

// running sums of the 8 packed deltas of a lane: d[q] = [s(2q), s(2q+1)]; returns the total
__device__ __forceinline__ uint32_t lane_prefix8(uint32_t d[4])
{
	uint32_t run = 0; // previous total in both halves
#pragma unroll
	for (int q = 0; q < 4; q++) {
		uint32_t t = d[q] + (d[q] << 16);             // [a, a+b]
		t = pk_add16(t, run);
		d[q] = t;
		run = __builtin_amdgcn_perm(t, t, 0x03020302); // broadcast the high half
	}
	return run >> 16;
}

// 8 one-byte values -> 4 packed pairs
__device__ __forceinline__ void expand8(uint2 dd, uint32_t v[4])
{
	v[0] = __builtin_amdgcn_perm(0, dd.x, 0x0c010c00);
	v[1] = __builtin_amdgcn_perm(0, dd.x, 0x0c030c02);
	v[2] = __builtin_amdgcn_perm(0, dd.y, 0x0c010c00);
	v[3] = __builtin_amdgcn_perm(0, dd.y, 0x0c030c02);
}

} // namespace ph
