// micro-benchmark: do unaligned 8-byte global stores/loads cost bandwidth on gfx950?
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_unaligned.hip -o gpurun_out/ubench_unaligned
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint2 __attribute__((aligned(1))) uint2_u;
typedef uint4 __attribute__((aligned(1))) uint4_u;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// copy16: aligned 16B load + aligned 16B store
__global__ void copy16(const uint4 *in, uint4 *out, size_t n16)
{
	size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	size_t stride = (size_t) gridDim.x * blockDim.x;
	for (; i < n16; i += stride)
		out[i] = in[i];
}

// narrow: 16B aligned load -> 8B store at byte offset `shift` (encode-like 2:1)
__global__ void narrow(const uint4 *in, uint8_t *out, size_t n16, int shift)
{
	size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	size_t stride = (size_t) gridDim.x * blockDim.x;
	for (; i < n16; i += stride) {
		uint4 v = in[i];
		uint2 r;
		r.x = __builtin_amdgcn_perm(v.y, v.x, 0x06040200);
		r.y = __builtin_amdgcn_perm(v.w, v.z, 0x06040200);
		*reinterpret_cast<uint2_u *>(out + shift + i * 8) = r;
	}
}

// widen: 8B load at byte offset shift -> 16B aligned store (decode-like 1:2)
__global__ void widen(const uint8_t *in, uint4 *out, size_t n16, int shift)
{
	size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	size_t stride = (size_t) gridDim.x * blockDim.x;
	for (; i < n16; i += stride) {
		uint2 r = *reinterpret_cast<const uint2_u *>(in + shift + i * 8);
		uint4 v;
		v.x = __builtin_amdgcn_perm(0, r.x, 0x0c010c00);
		v.y = __builtin_amdgcn_perm(0, r.x, 0x0c030c02);
		v.z = __builtin_amdgcn_perm(0, r.y, 0x0c010c00);
		v.w = __builtin_amdgcn_perm(0, r.y, 0x0c030c02);
		out[i] = v;
	}
}

int main()
{
	const size_t bytes = (size_t) 2 << 30; // 2 GiB in
	const size_t n16 = bytes / 16;
	uint4 *a, *b;
	uint8_t *c;
	CK(hipMalloc(&a, bytes));
	CK(hipMalloc(&b, bytes));
	CK(hipMalloc(&c, bytes / 2 + 4096));
	CK(hipMemset(a, 1, bytes));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	const int grids[] = { 2048, 8192, 65536 };
	for (int gi = 0; gi < 3; gi++) {
		const int grid = grids[gi];
		for (int test = 0; test < 9; test++) {
			float best = 1e9;
			const int shift = test == 0 ? 0 : (test - 1) % 4 == 0 ? 0 : (test - 1) % 4 == 1 ? 1 : (test - 1) % 4 == 2 ? 4 : 7;
			for (int rep = 0; rep < 6; rep++) {
				CK(hipEventRecord(e0));
				if (test == 0)
					hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, a, b, n16);
				else if (test <= 4)
					hipLaunchKernelGGL(narrow, dim3(grid), dim3(256), 0, 0, a, c, n16, shift);
				else
					hipLaunchKernelGGL(widen, dim3(grid), dim3(256), 0, 0, c, b, n16, shift);
				CK(hipEventRecord(e1));
				CK(hipEventSynchronize(e1));
				float ms;
				CK(hipEventElapsedTime(&ms, e0, e1));
				if (ms < best)
					best = ms;
			}
			const double moved = test == 0 ? 2.0 * bytes : 1.5 * bytes;
			printf("grid %6d %-7s shift %d : %.3f ms  %.1f GB/s\n", grid,
			       test == 0 ? "copy16" : test <= 4 ? "narrow" : "widen", shift, best, moved / best / 1e6);
		}
	}
	return 0;
}
