// press_kernels.hip - HIP kernels for gfx950 (MI355X, CDNA4): the per-read
// zig-zag-delta -> pack -> entropy hot path of the reference's press/ library.
//
// Execution model (v1): one 256-thread workgroup (4 wave64) per read; the read is
// walked in tiles of 2048 samples = 8 samples (one 16-byte load) per thread, so
//   * a thread's 8 key bits are exactly one svb16 key byte (two svb32 key bytes),
//   * every global load/store of samples is a coalesced 16-byte access,
//   * variable-length output goes through an LDS byte (or bit) FIFO whose byte 0 is
//     congruent to the global destination mod 16: whole 16-byte chunks are flushed
//     with aligned dwordx4 stores, the < 16 leftover bytes are carried to the next
//     tile; only the first and last chunk of a stream use byte stores,
//   * running offsets (bytes, bits, int16 prefix sums) are carried across tiles in
//     registers - no inter-workgroup communication at all.
// All arithmetic is integer (u8/u16/u32); no MFMA: the path is HBM/LDS bound.
//
// Formats and the reference lines they restate are cited per kernel.

#include <stdlib.h>

#include "press_internal.h"

namespace ph {

constexpr int WG = 256;            // threads per workgroup
constexpr int SPT = 8;             // samples per thread per tile (one dwordx4)
constexpr int TILE = WG * SPT;     // 2048 samples
constexpr uint32_t FAIL32 = 0xFFFFFFFFu;
constexpr uint64_t FAIL64 = ~0ull;

// ------------------------------------------------------------------ scans

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t t = __shfl_up(v, d, 64);
		if (lane >= d)
			v += t;
	}
	return v;
}

// exclusive prefix over the workgroup; `slot` is a 4-entry LDS array that the caller
// alternates between consecutive scans (so one barrier per scan suffices)
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *slot, uint32_t &tot)
{
	const uint32_t inc = wave_incl_scan(v);
	const int w = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 63)
		slot[w] = inc;
	__syncthreads();
	const uint32_t t0 = slot[0], t1 = slot[1], t2 = slot[2], t3 = slot[3];
	const uint32_t base = (w > 0 ? t0 : 0u) + (w > 1 ? t1 : 0u) + (w > 2 ? t2 : 0u);
	tot = t0 + t1 + t2 + t3;
	return base + inc - v;
}

__device__ __forceinline__ uint32_t block_or(uint32_t v, uint32_t *slot)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1)
		v |= __shfl_xor(v, d, 64);
	if ((threadIdx.x & 63) == 0)
		slot[threadIdx.x >> 6] = v;
	__syncthreads();
	return slot[0] | slot[1] | slot[2] | slot[3];
}

// ------------------------------------------------------------------ zig-zag delta of one thread's 8 samples

// trans.c:75 on a 16-bit difference
__device__ __forceinline__ uint32_t zz16(int32_t d)
{
	const int32_t d16 = (int32_t) (int16_t) d;
	return (uint32_t) ((d16 << 1) ^ (d16 >> 15)) & 0xFFFFu;
}

// trans.c:80
__device__ __forceinline__ int32_t unzz16(uint32_t z)
{
	return (int32_t) (z >> 1) ^ -(int32_t) (z & 1);
}

// Load samples [i0, i0+8) of a read (16-byte aligned, i0 % 8 == 0, i0 < n) and turn
// them into zig-zag-delta values (trans.c:215: prev = 0 before the first sample).
// Samples at or beyond n come out as z = 0 and are excluded through `nvalid`.
// q: ex-zd shift applied to every sample first (ex_zd.c:383).  ZD = false: the raw
// samples as u16 (svb12 without zd, press.c:1573).
template <bool ZD>
__device__ __forceinline__ void load_z8(const int16_t *in, uint32_t n, uint32_t i0, int q,
					uint32_t z[SPT], uint32_t &nvalid, uint32_t &ored)
{
	const uint4 v = *reinterpret_cast<const uint4 *>(in + i0);
	int32_t s[SPT];
	s[0] = (int16_t) (v.x & 0xFFFF); s[1] = (int16_t) (v.x >> 16);
	s[2] = (int16_t) (v.y & 0xFFFF); s[3] = (int16_t) (v.y >> 16);
	s[4] = (int16_t) (v.z & 0xFFFF); s[5] = (int16_t) (v.z >> 16);
	s[6] = (int16_t) (v.w & 0xFFFF); s[7] = (int16_t) (v.w >> 16);
	nvalid = min((uint32_t) SPT, n - i0);
	int32_t prev = 0;
	if (ZD && i0 > 0)
		prev = (int32_t) in[i0 - 1] >> q;
	ored = 0;
#pragma unroll
	for (int j = 0; j < SPT; j++) {
		const bool ok = (uint32_t) j < nvalid;
		if (ok)
			ored |= (uint32_t) s[j] & 0xFFFFu;
		const int32_t cur = s[j] >> q;
		uint32_t zj = ZD ? zz16(cur - prev) : ((uint32_t) cur & 0xFFFFu);
		z[j] = ok ? zj : 0u;
		prev = cur;
	}
}

// ------------------------------------------------------------------ LDS byte FIFO -> aligned global stores

constexpr int FIFO_BYTES = 16 + TILE * 2 + 16; // carried (<16) + worst tile + slack for the tail copy

// Flush the first `total` bytes of `buf` (buf[0] <-> global address g, g 16-byte aligned).
// Whole 16-byte chunks go out as dwordx4 stores; the remainder moves to the front of buf.
// `skip` leading bytes of the very first chunk are not part of the stream (they precede
// its start address) and are never stored.  Must be called by all threads after a barrier
// that made the tile's LDS writes visible.  Returns with g/fill/skip advanced; the caller's
// next LDS writes must be separated from this call by a barrier.
__device__ __forceinline__ void fifo_flush(uint8_t *buf, uint32_t total, uint8_t *&g, uint32_t &fill,
					   uint32_t &skip)
{
	const uint32_t nch = total >> 4;
	for (uint32_t c = threadIdx.x; c < nch; c += WG) {
		const uint4 v = reinterpret_cast<const uint4 *>(buf)[c];
		if (c == 0 && skip) {
			const uint32_t w[4] = { v.x, v.y, v.z, v.w };
			for (uint32_t b = skip; b < 16; b++)
				g[b] = (uint8_t) (w[b >> 2] >> (8 * (b & 3)));
		} else {
			reinterpret_cast<uint4 *>(g)[c] = v;
		}
	}
	if (nch) {
		if (threadIdx.x == 0) {
			// same thread read chunk 0 above, so overwriting it here is ordered
			const uint4 t = reinterpret_cast<const uint4 *>(buf)[nch];
			reinterpret_cast<uint4 *>(buf)[0] = t;
		}
		skip = 0;
	}
	g += (size_t) nch * 16;
	fill = total & 15;
}

// Store what is left in the FIFO with byte stores (end of a stream).  Needs a barrier
// between the last fifo_flush and this call.
__device__ __forceinline__ void fifo_finish(const uint8_t *buf, uint8_t *g, uint32_t fill, uint32_t skip)
{
	for (uint32_t b = skip + threadIdx.x; b < fill; b += WG)
		g[b] = buf[b];
}

// ------------------------------------------------------------------ svb16 / svb32 encode (a4, a6)
//
// svb16 (KEY2 = false): svb16/encode.hpp:11, encode_scalar.hpp:14 - ceil(n/8) key bytes,
//   bit i%8 of byte i/8 set iff value i takes 2 bytes; then the data bytes.
// svb32 (KEY2 = true): streamvbyte_encode.c:70 on the zd values widened to u32
//   (press.c:1590) - ceil(n/4) key bytes, 2 bits per value (0 or 1 here: values < 65536);
//   the data bytes are the same 1-or-2-byte sequence as svb16's.
template <bool KEY2, bool ZD>
__global__ __launch_bounds__(WG) void k_svb_encode(BatchArgs a)
{
	__shared__ __attribute__((aligned(16))) uint8_t buf[FIFO_BYTES];
	__shared__ uint32_t slots[8];

	const uint32_t r = blockIdx.x;
	const uint64_t o0 = a.off[r];
	const uint32_t n = a.nsamp[r];
	const int16_t *in = a.sig + o0;
	uint8_t *out = a.out + a.out_off[r];
	const uint64_t cap = a.out_off[r + 1] - a.out_off[r];
	const uint32_t klen = KEY2 ? (n + 3) / 4 : (n >> 3) + (((n & 7) + 7) >> 3);

	if ((uint64_t) klen + 2ull * n > cap) { // worst case of the format
		if (threadIdx.x == 0)
			a.out_len[r] = FAIL64;
		return;
	}

	uint8_t *data = out + klen;
	uint32_t skip = (uint32_t) ((uintptr_t) data & 15);
	uint8_t *g = data - skip;
	uint32_t fill = skip;
	uint64_t produced = 0;

	const uint32_t ntiles = (n + TILE - 1) / TILE;
	for (uint32_t t = 0; t < ntiles; t++) {
		const uint32_t i0 = t * TILE + threadIdx.x * SPT;
		uint32_t z[SPT], nvalid = 0, ored;
		uint32_t key = 0;
		if (i0 < n) {
			load_z8<ZD>(in, n, i0, 0, z, nvalid, ored);
#pragma unroll
			for (int j = 0; j < SPT; j++)
				key |= (z[j] > 255u ? 1u : 0u) << j;
			if (!KEY2) {
				out[i0 >> 3] = (uint8_t) key;
			} else {
				// 2-bit codes, value j of a quad in bits 2j: code 1 = two bytes
				uint32_t k0 = 0, k1 = 0;
#pragma unroll
				for (int j = 0; j < 4; j++) {
					k0 |= ((key >> j) & 1u) << (2 * j);
					k1 |= ((key >> (j + 4)) & 1u) << (2 * j);
				}
				out[i0 >> 2] = (uint8_t) k0;
				if (i0 + 4 < n)
					out[(i0 >> 2) + 1] = (uint8_t) k1;
			}
		}
		const uint32_t cnt = nvalid + __popc(key);
		uint32_t tot;
		uint32_t pos = fill + block_excl_scan(cnt, slots + 4 * (t & 1), tot);
#pragma unroll
		for (int j = 0; j < SPT; j++) {
			if ((uint32_t) j < nvalid) {
				buf[pos++] = (uint8_t) z[j];
				if (z[j] > 255u)
					buf[pos++] = (uint8_t) (z[j] >> 8);
			}
		}
		__syncthreads();
		fifo_flush(buf, fill + tot, g, fill, skip);
		produced += tot;
	}
	__syncthreads();
	fifo_finish(buf, g, fill, skip);
	if (threadIdx.x == 0)
		a.out_len[r] = (uint64_t) klen + produced;
}

// ------------------------------------------------------------------ LDS staging of an unaligned global byte range

constexpr int STAGE_BYTES = 16 + TILE * 2 + 16;

// Copy global bytes [src, src+len) into `buf` with aligned 16-byte loads so that
// buf[phase + k] = src[k]; returns phase = src & 15.  Chunks that start at or beyond
// `end` (one past the last readable byte of this stream) are zero filled - an aligned
// 16-byte load that starts inside a stream never leaves its page.
__device__ __forceinline__ uint32_t stage_in(uint8_t *buf, const uint8_t *src, uint32_t len, const uint8_t *end)
{
	const uint32_t phase = (uint32_t) ((uintptr_t) src & 15);
	const uint8_t *ga = src - phase;
	const uint32_t nch = (phase + len + 15) >> 4;
	for (uint32_t c = threadIdx.x; c < nch; c += WG) {
		uint4 v = make_uint4(0, 0, 0, 0);
		if (ga + (size_t) c * 16 < end)
			v = reinterpret_cast<const uint4 *>(ga)[c];
		reinterpret_cast<uint4 *>(buf)[c] = v;
	}
	return phase;
}

// Prefix-sum the thread-local running sums of one tile over the workgroup, add the
// carry of the previous tiles and store the 8 int16 results (trans.c:260 in parallel).
__device__ __forceinline__ void store_prefix8(int16_t *out, uint32_t n, uint32_t i0, uint32_t nvalid,
					      const uint32_t s[SPT], uint32_t base)
{
	if (nvalid == SPT) {
		uint4 v;
		v.x = ((base + s[0]) & 0xFFFFu) | ((base + s[1]) << 16);
		v.y = ((base + s[2]) & 0xFFFFu) | ((base + s[3]) << 16);
		v.z = ((base + s[4]) & 0xFFFFu) | ((base + s[5]) << 16);
		v.w = ((base + s[6]) & 0xFFFFu) | ((base + s[7]) << 16);
		*reinterpret_cast<uint4 *>(out + i0) = v;
	} else {
		for (uint32_t j = 0; j < nvalid; j++)
			out[i0 + j] = (int16_t) (base + s[j]);
	}
	(void) n;
}

// ------------------------------------------------------------------ svb16 / svb32 decode (a5, a6)
//
// svb16/decode.hpp:23 + decode_scalar.hpp:31; streamvbyte_decode.c:62 + trans.c:272.
// The stream does not carry the sample count: n comes from the caller (press.c:1689).
template <bool KEY2, bool ZD>
__global__ __launch_bounds__(WG) void k_svb_decode(DecodeArgs a)
{
	__shared__ __attribute__((aligned(16))) uint8_t buf[STAGE_BYTES];
	__shared__ uint32_t slots[16];

	const uint32_t r = blockIdx.x;
	const uint64_t o0 = a.off[r];
	const uint32_t n = a.nsamp[r];
	int16_t *out = a.sig + o0;
	const uint8_t *in = a.in + a.in_off[r];
	const uint64_t in_len = a.in_len[r];
	const uint8_t *end = in + in_len;
	const uint32_t klen = KEY2 ? (n + 3) / 4 : (n >> 3) + (((n & 7) + 7) >> 3);

	if (klen > in_len) { // not even the keys fit
		if (threadIdx.x == 0)
			a.out_n[r] = FAIL32;
		return;
	}
	const uint8_t *data = in + klen;
	uint64_t consumed = 0;
	uint32_t carry = 0; // running sample value (mod 2^16)
	bool bad = false;

	const uint32_t ntiles = (n + TILE - 1) / TILE;
	for (uint32_t t = 0; t < ntiles; t++) {
		const uint32_t i0 = t * TILE + threadIdx.x * SPT;
		uint32_t nvalid = 0, key = 0;
		if (i0 < n) {
			nvalid = min((uint32_t) SPT, n - i0);
			if (!KEY2) {
				key = in[i0 >> 3];
			} else {
				const uint32_t k0 = in[i0 >> 2];
				const uint32_t k1 = (i0 + 4 < n) ? in[(i0 >> 2) + 1] : 0u;
				// codes 2 and 3 cannot come from a 16-bit signal; they still move the
				// data pointer by 3 / 4 bytes as streamvbyte_decode.c:18-40 does
#pragma unroll
				for (int j = 0; j < 4; j++) {
					key |= ((k0 >> (2 * j)) & 3u) << (2 * j);
					key |= ((k1 >> (2 * j)) & 3u) << (2 * (j + 4));
				}
			}
			if (nvalid < SPT)
				key &= KEY2 ? ((1u << (2 * nvalid)) - 1u) : ((1u << nvalid) - 1u);
		}
		uint32_t cnt = nvalid;
		if (!KEY2) {
			cnt += __popc(key);
		} else {
#pragma unroll
			for (int j = 0; j < SPT; j++)
				cnt += (key >> (2 * j)) & 3u;
		}
		uint32_t tot;
		const uint32_t ex = block_excl_scan(cnt, slots + 4 * (t & 1), tot);
		if (KEY2 && tot > 2u * TILE) {
			// 3- and 4-byte codes: not a stream of 16-bit values, and more than the
			// staging buffer holds
			bad = true;
			break;
		}
		const uint32_t phase = stage_in(buf, data + consumed, tot, end);
		__syncthreads();

		uint32_t s[SPT];
		uint32_t p = phase + ex;
		uint32_t acc = 0;
#pragma unroll
		for (int j = 0; j < SPT; j++) {
			uint32_t v = 0;
			if ((uint32_t) j < nvalid) {
				const uint32_t code = KEY2 ? ((key >> (2 * j)) & 3u) : ((key >> j) & 1u);
				v = buf[p];
				if (code >= 1)
					v |= (uint32_t) buf[p + 1] << 8;
				p += 1 + code;
			}
			if (ZD) {
				acc += (uint32_t) unzz16(v & 0xFFFFu);
				s[j] = acc;
			} else {
				s[j] = v;
			}
		}
		uint32_t base = 0;
		if (ZD) {
			uint32_t tsum;
			base = carry + block_excl_scan(acc, slots + 8 + 4 * (t & 1), tsum);
			carry += tsum;
		} else {
			__syncthreads(); // buf is re-staged by the next tile
		}
		if (i0 < n)
			store_prefix8(out, n, i0, nvalid, s, base);
		consumed += tot;
	}
	if (threadIdx.x == 0)
		a.out_n[r] = (!bad && klen + consumed <= in_len) ? n : FAIL32;
}

// ------------------------------------------------------------------ exception split: pass A (scan)
//
// vbe21_press and siblings, first loop (press.c:2693-2705); ex_press (ex_zd.c:44-72):
// find the values of zd[1..n) that exceed 255.  Writes, per read, zd[0], the exception
// count, the exception list (position in zd[1..], raw value) and the OR of all samples.
// REDO = true is the second launch for ex-zd: only reads whose samples are all
// divisible by 2^q, q > 0, are scanned again on the shifted samples.
template <bool REDO>
__global__ __launch_bounds__(WG) void k_ex_scan(BatchArgs a)
{
	__shared__ uint32_t slots[12];

	const uint32_t r = blockIdx.x;
	const uint64_t o0 = a.off[r];
	const uint32_t n = a.nsamp[r];
	const int16_t *in = a.sig + o0;
	uint32_t *lpos = a.ex_pos + o0;
	uint32_t *lval = a.ex_val + o0;
	ReadMeta *m = a.meta + r;

	int q = 0;
	if (REDO) {
		const uint32_t ored = m->ored;
		// ex_zd.c:358-381: largest q <= 5 with every sample divisible by 2^q
		while (q < 5 && !((ored >> q) & 1u))
			q++;
		if (q == 0 || n == 0)
			return;
	}

	uint32_t rank0 = 0, oracc = 0, zd0 = 0;
	const uint32_t ntiles = (n + TILE - 1) / TILE;
	for (uint32_t t = 0; t < ntiles; t++) {
		const uint32_t i0 = t * TILE + threadIdx.x * SPT;
		uint32_t z[SPT], nvalid = 0, ored = 0, mask = 0;
		if (i0 < n) {
			load_z8<true>(in, n, i0, q, z, nvalid, ored);
#pragma unroll
			for (int j = 0; j < SPT; j++)
				mask |= (z[j] > 255u ? 1u : 0u) << j;
			if (i0 == 0) {
				zd0 = z[0];
				mask &= ~1u; // zd[0] is stored raw, never an exception
			}
		}
		oracc |= ored;
		uint32_t tot;
		uint32_t rank = rank0 + block_excl_scan(__popc(mask), slots + 4 * (t & 1), tot);
		if (mask) {
#pragma unroll
			for (int j = 0; j < SPT; j++) {
				if ((mask >> j) & 1u) {
					lpos[rank] = i0 + j - 1;
					lval[rank] = z[j];
					rank++;
				}
			}
		}
		rank0 += tot;
	}
	const uint32_t ored_all = REDO ? 0u : block_or(oracc, slots + 8);
	if (threadIdx.x == 0) {
		m->nex = rank0;
		m->zd0 = zd0;
		m->q = (uint32_t) q;
		if (!REDO)
			m->ored = ored_all;
	}
}

// ------------------------------------------------------------------ exception split: section builder
//
// Writes the header and "u32 nex || section" of one read and decides whether the read
// fits its slot.  One wave per read; lane 0 does the (tiny: ~5 exceptions per read on
// NA12878, thesis/plots/ex-tab.tex:14) serial work.
//   vbe21   press.c:2707-2716        vbbe21  press.c:2826-2872 (bit-pack press.c:285-397)
//   vbsbe21 press.c:3028-3082        vbsse21 press.c:3232-3276   ex-zd ex_zd.c:83-154

__device__ __forceinline__ void put8(uint8_t *p, uint32_t v) { p[0] = (uint8_t) v; }
__device__ __forceinline__ void put16(uint8_t *p, uint32_t v) { p[0] = (uint8_t) v; p[1] = (uint8_t) (v >> 8); }
__device__ __forceinline__ void put32(uint8_t *p, uint32_t v)
{
	p[0] = (uint8_t) v; p[1] = (uint8_t) (v >> 8); p[2] = (uint8_t) (v >> 16); p[3] = (uint8_t) (v >> 24);
}
__device__ __forceinline__ uint32_t get16(const uint8_t *p) { return (uint32_t) p[0] | ((uint32_t) p[1] << 8); }
__device__ __forceinline__ uint32_t get32(const uint8_t *p)
{
	return (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16) | ((uint32_t) p[3] << 24);
}

__device__ __forceinline__ uint32_t svb32_nbytes(uint32_t v)
{
	return v < (1u << 8) ? 1u : v < (1u << 16) ? 2u : v < (1u << 24) ? 3u : 4u;
}

__device__ __forceinline__ uint32_t minbits32(uint32_t max)
{
	return max ? 32u - (uint32_t) __clz((int) max) : 0u; // press.c:463
}

// delta-coded position k (trans.c:129): p0, p[k]-p[k-1]-1
__device__ __forceinline__ uint32_t dpos(const uint32_t *pos, uint32_t k)
{
	return k ? pos[k] - pos[k - 1] - 1u : pos[0];
}

// sizes of the two coded blocks of a section with nex > 1
__device__ void exsec_sizes(int fmt, const uint32_t *pos, const uint32_t *val, uint32_t nex,
			    uint32_t &len_pos, uint32_t &len_val, uint32_t &bits_pos, uint32_t &bits_val)
{
	uint32_t maxd = 0, maxv = 0, dbytes = 0, vbytes = 0;
	for (uint32_t k = 0; k < nex; k++) {
		const uint32_t d = dpos(pos, k), v = val[k] - 256u;
		maxd = max(maxd, d);
		maxv = max(maxv, v);
		dbytes += svb32_nbytes(d);
		vbytes += (fmt == EXF_EXZD) ? svb32_nbytes(v) : (v > 255u ? 2u : 1u);
	}
	bits_pos = minbits32(maxd);
	bits_val = minbits32(maxv);
	len_pos = (fmt == EXF_VBBE21) ? 1u + (uint32_t) (((uint64_t) nex * bits_pos + 7) / 8)
				      : (nex + 3) / 4 + dbytes;
	if (fmt == EXF_VBBE21 || fmt == EXF_VBSBE21)
		len_val = 1u + (uint32_t) (((uint64_t) nex * bits_val + 7) / 8);
	else if (fmt == EXF_VBSSE21)
		len_val = (nex + 7) / 8 + vbytes;
	else
		len_val = (nex + 3) / 4 + vbytes;
}

// [bits][values, `bits` bits each, most significant bit first] (press.c:486-505)
__device__ void bitpack_write(uint8_t *o, uint32_t nex, uint32_t bits, const uint32_t *pos, const uint32_t *val)
{
	o[0] = (uint8_t) bits;
	if (!bits)
		return;
	uint32_t acc = 0, nacc = 0;
	uint8_t *p = o + 1;
	for (uint32_t k = 0; k < nex; k++) {
		const uint32_t v = pos ? dpos(pos, k) : val[k] - 256u;
		for (int b = (int) bits - 1; b >= 0; b--) {
			acc = (acc << 1) | ((v >> b) & 1u);
			if (++nacc == 8) {
				*p++ = (uint8_t) acc;
				acc = 0;
				nacc = 0;
			}
		}
	}
	if (nacc)
		*p = (uint8_t) (acc << (8 - nacc));
}

// streamvbyte_encode.c:36 over position deltas (pos != NULL) or value-256
__device__ void svb32_write(uint8_t *o, uint32_t nex, const uint32_t *pos, const uint32_t *val)
{
	uint8_t *d = o + (nex + 3) / 4;
	uint32_t key = 0;
	for (uint32_t k = 0; k < nex; k++) {
		const uint32_t v = pos ? dpos(pos, k) : val[k] - 256u;
		const uint32_t nb = svb32_nbytes(v);
		for (uint32_t b = 0; b < nb; b++)
			*d++ = (uint8_t) (v >> (8 * b));
		key |= (nb - 1) << (2 * (k & 3));
		if ((k & 3) == 3 || k + 1 == nex) {
			o[k >> 2] = (uint8_t) key;
			key = 0;
		}
	}
}

// svb16/encode_scalar.hpp:14 without delta/zigzag over value-256 (press.c:3262: svb12_press)
__device__ void svb16_write(uint8_t *o, uint32_t nex, const uint32_t *val)
{
	uint8_t *d = o + (nex + 7) / 8;
	uint32_t key = 0;
	for (uint32_t k = 0; k < nex; k++) {
		const uint32_t v = (val[k] - 256u) & 0xFFFFu;
		*d++ = (uint8_t) v;
		if (v > 255u) {
			*d++ = (uint8_t) (v >> 8);
			key |= 1u << (k & 7);
		}
		if ((k & 7) == 7 || k + 1 == nex) {
			o[k >> 3] = (uint8_t) key;
			key = 0;
		}
	}
}

__global__ __launch_bounds__(64) void k_ex_section(BatchArgs a, int fmt, int huff)
{
	const uint32_t r = blockIdx.x;
	if (threadIdx.x != 0)
		return;
	const uint64_t o0 = a.off[r];
	const uint32_t n = a.nsamp[r];
	const uint32_t *pos = a.ex_pos + o0;
	const uint32_t *val = a.ex_val + o0;
	uint8_t *out = a.out + a.out_off[r];
	const uint64_t cap = a.out_off[r + 1] - a.out_off[r];
	ReadMeta *m = a.meta + r;
	const uint32_t nex = m->nex;
	const uint32_t hdr = (fmt == EXF_EXZD) ? 12u : 2u;

	m->hdr = hdr;
	m->status = 1;
	m->seclen = 0;
	a.out_len[r] = FAIL64; // until proven to fit
	if (n == 0)
		return; // the reference reads zd[0] of an empty array: outside its domain

	uint32_t len_pos = 0, len_val = 0, bits_pos = 0, bits_val = 0;
	uint64_t seclen = 4;
	if (fmt == EXF_VBE21) {
		seclen += 6ull * nex;
	} else if (nex == 1) {
		seclen += (fmt == EXF_EXZD) ? 8 : 6;
	} else if (nex > 1) {
		exsec_sizes(fmt, pos, val, nex, len_pos, len_val, bits_pos, bits_val);
		seclen += 8ull + len_pos + len_val;
	}
	const uint64_t nlow = (uint64_t) (n - 1) - nex;
	// one-byte stream: raw, or at least the 4-byte count of the Huffman stream
	// (huff: 0 plain, 1 / 2 static Huffman (v1 / chunked pass B), 3 range coder: its stream is sized by k_rcs_encode)
	const uint64_t need = hdr + seclen + (huff == 3 ? 0 : huff ? 4 : nlow);
	if (need > cap)
		return;
	// press.c:4520,4636,4752: the b/sb/ss Huffman variants keep the section length in a uint16_t
	if (huff && huff != 3 && fmt != EXF_VBE21 && seclen > 65535)
		return;
	// ex_zd.c:411: the reference works in a 2n+1024-byte buffer
	if (fmt == EXF_EXZD && need > 2ull * n + 1024)
		return;

	uint8_t *p = out;
	if (fmt == EXF_EXZD) {
		p[0] = 0; // version
		put32(p + 1, n);
		put32(p + 5, 0);
		p[9] = (uint8_t) m->q;
		put16(p + 10, m->zd0);
	} else {
		put16(p, m->zd0);
	}
	p += hdr;
	put32(p, nex);
	p += 4;
	if (fmt == EXF_VBE21) {
		// nex x u32 positions, nex x u16 values: copied by k_ex_fill_vbe21, a whole wave per read
	} else if (nex == 1) {
		put32(p, pos[0]);
		if (fmt == EXF_EXZD)
			put32(p + 4, val[0] - 256u);
		else
			put16(p + 4, val[0] - 256u);
	} else if (nex > 1) {
		put32(p, len_pos);
		p += 4;
		if (fmt == EXF_VBBE21)
			bitpack_write(p, nex, bits_pos, pos, nullptr);
		else
			svb32_write(p, nex, pos, nullptr);
		p += len_pos;
		put32(p, len_val);
		p += 4;
		if (fmt == EXF_VBBE21 || fmt == EXF_VBSBE21)
			bitpack_write(p, nex, bits_val, nullptr, val);
		else if (fmt == EXF_VBSSE21)
			svb16_write(p, nex, val);
		else
			svb32_write(p, nex, nullptr, val);
	}
	m->seclen = (uint32_t) seclen;
	m->nlow = (uint32_t) nlow;
	if (huff == 2) { // chunked Huffman pass B: the payload follows the symbol count (huffman.c:1203: htonl)
		uint8_t *h = out + hdr + seclen;
		h[0] = (uint8_t) (nlow >> 24);
		h[1] = (uint8_t) (nlow >> 16);
		h[2] = (uint8_t) (nlow >> 8);
		h[3] = (uint8_t) nlow;
	}
	m->status = 0;
	if (!huff)
		a.out_len[r] = (uint64_t) hdr + seclen + nlow; // the Huffman pass B knows its own length
}

// vbe21's exception section (press.c:2703-2715: nex x u32 position, nex x u16 raw value) is a
// plain copy of the lists: one wave per read instead of k_ex_section's single lane, so that a
// read with thousands of exceptions costs microseconds, not milliseconds.
__global__ __launch_bounds__(64) void k_ex_fill_vbe21(BatchArgs a)
{
	const uint32_t r = blockIdx.x;
	const ReadMeta *m = a.meta + r;
	if (m->status)
		return;
	const uint32_t nex = m->nex;
	const uint64_t o0 = a.off[r];
	const uint32_t *pos = a.ex_pos + o0;
	const uint32_t *val = a.ex_val + o0;
	uint8_t *p = a.out + a.out_off[r] + m->hdr + 4;
	for (uint32_t k = threadIdx.x; k < nex; k += 64) {
		put32(p + 4ull * k, pos[k]);
		put16(p + 4ull * nex + 2ull * k, val[k]);
	}
}

// ------------------------------------------------------------------ exception split: pass B (one-byte stream)
//
// Plain (HUFF = false): the non-exception values of zd[1..n) as bytes, in order
// (press.c:2717-2725).  HUFF = true: their static-Huffman stream (huffman.c:1184 +
// do_memory_encode :848): u32 big-endian symbol count, then the codes packed from bit 0
// of each byte upwards - i.e. a little-endian bit stream, which is exactly an LDS dword
// array filled with OR-ed (code << bitpos) words.

constexpr int HBITS_MAX = 24;                                  // longest code the encoder accepts
constexpr int HBUF_DW = (128 + TILE * HBITS_MAX + 31) / 32 + 4 + 4; // carried bits + worst tile (+ slack)

template <bool HUFF>
__global__ __launch_bounds__(WG) void k_low_encode(BatchArgs a)
{
	__shared__ __attribute__((aligned(16))) uint32_t wbuf[HUFF ? HBUF_DW : (FIFO_BYTES / 4)];
	__shared__ uint32_t slots[8];
	__shared__ uint32_t enc[HUFF ? 256 : 1];

	const uint32_t r = blockIdx.x;
	const ReadMeta m = a.meta[r];
	if (m.status) {
		if (threadIdx.x == 0)
			a.out_len[r] = FAIL64;
		return;
	}
	const uint64_t o0 = a.off[r];
	const uint32_t n = a.nsamp[r];
	const int16_t *in = a.sig + o0;
	uint8_t *out = a.out + a.out_off[r];
	const uint64_t cap = a.out_off[r + 1] - a.out_off[r];
	const int q = (int) m.q;
	uint8_t *buf = reinterpret_cast<uint8_t *>(wbuf);

	uint8_t *start = out + m.hdr + m.seclen;
	if (HUFF) {
		if (threadIdx.x < 256)
			enc[threadIdx.x] = a.huff->enc[threadIdx.x];
		for (uint32_t i = threadIdx.x; i < (uint32_t) HBUF_DW; i += WG)
			wbuf[i] = 0;
		if (threadIdx.x == 0) { // huffman.c:1203: htonl(symbol count)
			start[0] = (uint8_t) (m.nlow >> 24);
			start[1] = (uint8_t) (m.nlow >> 16);
			start[2] = (uint8_t) (m.nlow >> 8);
			start[3] = (uint8_t) m.nlow;
		}
		start += 4;
		__syncthreads();
	}
	uint32_t skip = (uint32_t) ((uintptr_t) start & 15);
	uint8_t *g = start - skip;
	uint32_t fill = HUFF ? skip * 8 : skip; // bits / bytes waiting at the front of the FIFO
	uint64_t produced = 0;                  // bits / bytes of the stream so far
	bool failed = false;

	const uint32_t ntiles = (n + TILE - 1) / TILE;
	for (uint32_t t = 0; t < ntiles; t++) {
		const uint32_t i0 = t * TILE + threadIdx.x * SPT;
		uint32_t z[SPT], nvalid = 0, ored, mask = 0;
		uint32_t cnt = 0;
		if (i0 < n) {
			load_z8<true>(in, n, i0, q, z, nvalid, ored);
#pragma unroll
			for (int j = 0; j < SPT; j++)
				mask |= ((z[j] > 255u || (uint32_t) j >= nvalid) ? 1u : 0u) << j;
			if (i0 == 0)
				mask |= 1u; // zd[0] lives in the header
#pragma unroll
			for (int j = 0; j < SPT; j++) {
				if (!((mask >> j) & 1u))
					cnt += HUFF ? (enc[z[j]] >> 24) : 1u;
			}
		}
		uint32_t tot;
		uint32_t pos = fill + block_excl_scan(cnt, slots + 4 * (t & 1), tot);
		if (cnt) {
#pragma unroll
			for (int j = 0; j < SPT; j++) {
				if ((mask >> j) & 1u)
					continue;
				if (!HUFF) {
					buf[pos++] = (uint8_t) z[j];
				} else {
					const uint32_t e = enc[z[j]];
					const uint32_t len = e >> 24, sh = pos & 31;
					const uint64_t w = (uint64_t) (e & 0xFFFFFFu) << sh;
					if ((uint32_t) w)
						atomicOr(&wbuf[pos >> 5], (uint32_t) w);
					if ((uint32_t) (w >> 32))
						atomicOr(&wbuf[(pos >> 5) + 1], (uint32_t) (w >> 32));
					pos += len;
				}
			}
		}
		__syncthreads();
		const uint32_t total = fill + tot;
		produced += tot;
		if (HUFF) {
			// capacity: the stream is never allowed past the slot
			const uint64_t end_off = (uint64_t) (start - out) + (produced + 7) / 8;
			if (end_off > cap) {
				failed = true;
				break;
			}
			// flush whole 128-bit chunks; every thread clears the chunk it stored
			const uint32_t nch = total >> 7;
			for (uint32_t c = threadIdx.x; c < nch; c += WG) {
				const uint4 v = reinterpret_cast<const uint4 *>(wbuf)[c];
				if (c == 0 && skip) {
					const uint32_t w[4] = { v.x, v.y, v.z, v.w };
					for (uint32_t b = skip; b < 16; b++)
						g[b] = (uint8_t) (w[b >> 2] >> (8 * (b & 3)));
				} else {
					reinterpret_cast<uint4 *>(g)[c] = v;
				}
				reinterpret_cast<uint4 *>(wbuf)[c] = make_uint4(0, 0, 0, 0);
			}
			if (nch) {
				if (threadIdx.x == 0) {
					const uint4 tl = reinterpret_cast<const uint4 *>(wbuf)[nch];
					reinterpret_cast<uint4 *>(wbuf)[nch] = make_uint4(0, 0, 0, 0);
					reinterpret_cast<uint4 *>(wbuf)[0] = tl;
				}
				skip = 0;
			}
			g += (size_t) nch * 16;
			fill = total & 127;
		} else {
			fifo_flush(buf, total, g, fill, skip);
		}
	}
	__syncthreads();
	if (failed) {
		if (threadIdx.x == 0)
			a.out_len[r] = FAIL64;
		return;
	}
	if (HUFF) {
		fifo_finish(buf, g, (fill + 7) / 8, skip);
		if (threadIdx.x == 0)
			a.out_len[r] = (uint64_t) m.hdr + m.seclen + 4 + (produced + 7) / 8;
	} else {
		fifo_finish(buf, g, fill, skip);
		if (threadIdx.x == 0)
			a.out_len[r] = (uint64_t) m.hdr + m.seclen + produced;
	}
}

// ------------------------------------------------------------------ decode: parse header + section
//
// vbe21_depress and siblings (press.c:2731, 2890, 3098, 3291), ex_depress (ex_zd.c:174)
// up to the point where the exception list is known.  One wave per read, lane 0 works.

__device__ uint32_t svb32_read(const uint8_t *in, uint32_t len, uint32_t nex, uint32_t *dst)
{
	const uint32_t klen = (nex + 3) / 4;
	if (klen > len)
		return 1;
	uint32_t d = klen;
	for (uint32_t k = 0; k < nex; k++) {
		const uint32_t nb = ((in[k >> 2] >> (2 * (k & 3))) & 3u) + 1;
		if (d + nb > len)
			return 1;
		uint32_t v = 0;
		for (uint32_t b = 0; b < nb; b++)
			v |= (uint32_t) in[d + b] << (8 * b);
		dst[k] = v;
		d += nb;
	}
	return 0;
}

__device__ uint32_t svb16_read(const uint8_t *in, uint32_t len, uint32_t nex, uint32_t *dst)
{
	const uint32_t klen = (nex + 7) / 8;
	if (klen > len)
		return 1;
	uint32_t d = klen;
	for (uint32_t k = 0; k < nex; k++) {
		const uint32_t nb = ((in[k >> 3] >> (k & 7)) & 1u) + 1;
		if (d + nb > len)
			return 1;
		dst[k] = nb == 2 ? get16(in + d) : in[d];
		d += nb;
	}
	return 0;
}

__device__ uint32_t bitpack_read(const uint8_t *in, uint32_t len, uint32_t nex, uint32_t *dst)
{
	if (len < 1)
		return 1;
	const uint32_t bits = in[0];
	if (bits > 32 || 1ull + ((uint64_t) nex * bits + 7) / 8 > len)
		return 1;
	uint64_t bp = 0;
	for (uint32_t k = 0; k < nex; k++) {
		uint32_t v = 0;
		for (uint32_t b = 0; b < bits; b++, bp++)
			v = (v << 1) | ((in[1 + (bp >> 3)] >> (7 - (bp & 7))) & 1u);
		dst[k] = v;
	}
	return 0;
}

__global__ __launch_bounds__(64) void k_ex_parse(DecodeArgs a, int fmt, int huff)
{
	const uint32_t r = blockIdx.x;
	if (threadIdx.x != 0)
		return;
	const uint64_t o0 = a.off[r];
	const uint32_t cap = a.nsamp[r]; // samples the caller has room for
	uint32_t *pos = a.ex_pos + o0;
	uint32_t *val = a.ex_val + o0;
	const uint8_t *in = a.in + a.in_off[r];
	const uint64_t len = a.in_len[r];
	ReadMeta *m = a.meta + r;
	const uint32_t hdr = (fmt == EXF_EXZD) ? 12u : 2u;

	m->status = 1;
	m->hdr = hdr;
	m->nex = 0;
	m->q = 0;
	if (len < (uint64_t) hdr + 4 || cap == 0)
		return;
	if (fmt == EXF_EXZD) {
		// ex_zd.c:495-519: version 0, u64 n, q <= 5
		if (in[0] != 0 || get32(in + 5) != 0 || in[9] > 5)
			return;
		const uint32_t n = get32(in + 1);
		if (n == 0 || n > cap)
			return;
		m->q = in[9];
		m->zd0 = get16(in + 10);
	} else {
		m->zd0 = get16(in);
	}
	const uint8_t *p = in + hdr;
	uint64_t left = len - hdr - 4;
	const uint32_t nex = get32(p);
	p += 4;
	if ((uint64_t) nex >= cap)
		return; // more exceptions than zd[1..] can hold
	uint64_t seclen = 4;
	uint32_t bad = 0;
	if (nex == 0) {
	} else if (fmt == EXF_VBE21) {
		if (left < 6ull * nex)
			return;
		// the lists themselves: k_ex_parse_fill_vbe21, a whole wave per read
		seclen += 6ull * nex;
	} else if (nex == 1) {
		const uint32_t need = (fmt == EXF_EXZD) ? 8u : 6u;
		if (left < need)
			return;
		pos[0] = get32(p);
		val[0] = (((fmt == EXF_EXZD) ? get32(p + 4) : get16(p + 4)) + 256u) & 0xFFFFu;
		seclen += need;
	} else {
		if (left < 4)
			return;
		const uint32_t lp = get32(p);
		if (left < 8ull + lp)
			return;
		bad |= (fmt == EXF_VBBE21) ? bitpack_read(p + 4, lp, nex, pos) : svb32_read(p + 4, lp, nex, pos);
		const uint32_t lv = get32(p + 4 + lp);
		if (left < 8ull + lp + lv)
			return;
		const uint8_t *pv = p + 8 + lp;
		if (fmt == EXF_VBBE21 || fmt == EXF_VBSBE21)
			bad |= bitpack_read(pv, lv, nex, val);
		else if (fmt == EXF_VBSSE21)
			bad |= svb16_read(pv, lv, nex, val);
		else
			bad |= svb32_read(pv, lv, nex, val);
		if (bad)
			return;
		// trans.c:186 + "value - 256" (press.c:3345: out = ex + UINT8_MAX + 1, 16-bit for the non-ex-zd forms)
		uint32_t prev = 0;
		for (uint32_t k = 0; k < nex; k++) {
			const uint32_t pk = k ? prev + pos[k] + 1u : pos[0];
			pos[k] = pk;
			prev = pk;
			val[k] = (val[k] + 256u) & 0xFFFFu;
		}
		seclen += 8ull + lp + lv;
	}
	// positions must be strictly increasing and inside zd[1..cap)
	if (fmt != EXF_VBE21) {
		for (uint32_t k = 0; k < nex; k++) {
			if (pos[k] >= cap - 1 || (k && pos[k] <= pos[k - 1]))
				return;
		}
	}
	m->nex = nex;
	m->seclen = (uint32_t) seclen;
	uint64_t nlow;
	if (huff == 3) {
		// press.c:5465: the caller passes the exact sample count, the rest are one-byte values
		nlow = (uint64_t) cap - 1 - nex;
	} else if (huff) {
		// huffman.c:1236 + :704: at least one payload byte behind the 4-byte count
		const uint64_t hl = len - hdr - seclen;
		if (hl <= 4)
			return;
		const uint8_t *h = in + hdr + seclen;
		nlow = ((uint32_t) h[0] << 24) | ((uint32_t) h[1] << 16) | ((uint32_t) h[2] << 8) | h[3];
	} else {
		nlow = len - hdr - seclen;
	}
	if (1ull + nlow + nex > cap)
		return;
	m->nlow = (uint32_t) nlow;
	m->status = 0;
}

// vbe21: exception lists out of the section and their validation (strictly increasing positions
// inside zd[1..cap)), one wave per read; a violation fails the read like k_ex_parse would.
__global__ __launch_bounds__(64) void k_ex_parse_fill_vbe21(DecodeArgs a)
{
	const uint32_t r = blockIdx.x;
	ReadMeta *m = a.meta + r;
	if (m->status)
		return;
	const uint32_t nex = m->nex;
	const uint32_t cap = a.nsamp[r];
	const uint64_t o0 = a.off[r];
	uint32_t *pos = a.ex_pos + o0;
	uint32_t *val = a.ex_val + o0;
	const uint8_t *p = a.in + a.in_off[r] + m->hdr + 4;
	bool bad = false;
	for (uint32_t k = threadIdx.x; k < nex; k += 64) {
		const uint32_t pk = get32(p + 4ull * k);
		pos[k] = pk;
		val[k] = get16(p + 4ull * nex + 2ull * k);
		if (pk >= cap - 1 || (k && pk <= get32(p + 4ull * (k - 1))))
			bad = true;
	}
	if (__ballot(bad) && threadIdx.x == 0)
		m->status = 1;
}

// ------------------------------------------------------------------ decode: merge + undo zig-zag delta
//
// Second half of vbe21_depress (press.c:2757-2771) fused with unzigdelta_u16_16
// (trans.c:260): sample i >= 1 is exception k if pos[k] == i-1, otherwise the
// (i-1-#exceptions before it)-th one-byte value.  Exception ranks come from a binary
// search of the (sorted) position list, so tiles need no scan for their offsets.

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *p, uint32_t n, uint32_t key)
{
	uint32_t lo = 0, hi = n;
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (p[mid] < key)
			lo = mid + 1;
		else
			hi = mid;
	}
	return lo;
}

template <bool HUFF>
__global__ __launch_bounds__(WG) void k_low_decode(DecodeArgs a)
{
	__shared__ __attribute__((aligned(16))) uint8_t buf[STAGE_BYTES];
	__shared__ uint32_t slots[8];

	const uint32_t r = blockIdx.x;
	const ReadMeta m = a.meta[r];
	if (m.status) {
		if (threadIdx.x == 0)
			a.out_n[r] = FAIL32;
		return;
	}
	const uint64_t o0 = a.off[r];
	int16_t *out = a.sig + o0;
	const uint32_t *pos = a.ex_pos + o0;
	const uint32_t *val = a.ex_val + o0;
	const uint32_t nex = m.nex;
	const uint32_t n = 1u + m.nlow + nex; // press.c:2774: *nout = i (+1 for zd[0])
	const uint8_t *low = HUFF ? a.low + o0 : a.in + a.in_off[r] + m.hdr + m.seclen;
	const uint8_t *end = low + m.nlow;
	const int q = (int) m.q;
	uint32_t carry = 0;

	const uint32_t ntiles = (n + TILE - 1) / TILE;
	for (uint32_t t = 0; t < ntiles; t++) {
		const uint32_t t0 = t * TILE;
		const uint32_t tn = min((uint32_t) TILE, n - t0);
		// one-byte values of this tile: stream indices [u0 - e0, u1 - e1)
		const uint32_t u0 = t0 ? t0 - 1 : 0, u1 = t0 + tn - 1;
		const uint32_t e0 = lower_bound_u32(pos, nex, u0);
		const uint32_t e1 = lower_bound_u32(pos, nex, u1);
		const uint32_t l0 = u0 - e0, l1 = u1 - e1;
		const uint32_t phase = stage_in(buf, low + l0, l1 - l0, end);
		__syncthreads();

		const uint32_t i0 = t0 + threadIdx.x * SPT;
		uint32_t s[SPT], nvalid = 0, acc = 0;
		if (i0 < n) {
			nvalid = min((uint32_t) SPT, n - i0);
			const uint32_t uf = i0 ? i0 - 1 : 0;
			uint32_t e = lower_bound_u32(pos, nex, uf);
			uint32_t p = phase + (uf - e) - l0;
			uint32_t nextpos = e < nex ? pos[e] : FAIL32;
#pragma unroll
			for (int j = 0; j < SPT; j++) {
				uint32_t z = 0;
				if ((uint32_t) j < nvalid) {
					const uint32_t i = i0 + j;
					if (i == 0) {
						z = m.zd0;
					} else if (i - 1 == nextpos) {
						z = val[e];
						e++;
						nextpos = e < nex ? pos[e] : FAIL32;
					} else {
						z = buf[p++];
					}
				}
				acc += (uint32_t) unzz16(z & 0xFFFFu);
				s[j] = acc;
			}
		} else {
#pragma unroll
			for (int j = 0; j < SPT; j++)
				s[j] = 0;
		}
		uint32_t tsum;
		const uint32_t base = carry + block_excl_scan(acc, slots + 4 * (t & 1), tsum);
		carry += tsum;
		if (i0 < n) {
			if (q) { // ex_zd.c:396 do_rev_qts_inplace
#pragma unroll
				for (int j = 0; j < SPT; j++)
					s[j] = ((base + s[j]) << q) - base;
			}
			store_prefix8(out, n, i0, nvalid, s, base);
		}
	}
	if (threadIdx.x == 0)
		a.out_n[r] = n;
}

// ------------------------------------------------------------------ launchers

void launch_svb_encode(const BatchArgs &a, bool key2bit, bool zd, hipStream_t s)
{
	if (!a.nreads)
		return;
	const dim3 grid(a.nreads), block(WG);
	ktime_begin(0, s);
	if (key2bit)
		hipLaunchKernelGGL((k_svb_encode<true, true>), grid, block, 0, s, a);
	else if (zd)
		hipLaunchKernelGGL((k_svb_encode<false, true>), grid, block, 0, s, a);
	else
		hipLaunchKernelGGL((k_svb_encode<false, false>), grid, block, 0, s, a);
	ktime_end(0, s);
}

void launch_svb_decode(const DecodeArgs &a, bool key2bit, bool zd, hipStream_t s)
{
	if (!a.nreads)
		return;
	const dim3 grid(a.nreads), block(WG);
	ktime_begin(1, s);
	if (key2bit)
		hipLaunchKernelGGL((k_svb_decode<true, true>), grid, block, 0, s, a);
	else if (zd)
		hipLaunchKernelGGL((k_svb_decode<false, true>), grid, block, 0, s, a);
	else
		hipLaunchKernelGGL((k_svb_decode<false, false>), grid, block, 0, s, a);
	ktime_end(1, s);
}

void launch_ex_section(const BatchArgs &a, int fmt, int ent, hipStream_t s)
{
	hipLaunchKernelGGL(k_ex_section, dim3(a.nreads), dim3(64), 0, s, a, fmt, ent == 1 ? 2 : ent == 2 ? 3 : 0);
	if (fmt == EXF_VBE21)
		hipLaunchKernelGGL(k_ex_fill_vbe21, dim3(a.nreads), dim3(64), 0, s, a);
}

void launch_ex_encode(const BatchArgs &a0, int fmt, bool huff, hipStream_t s)
{
	if (!a0.nreads)
		return;
	const BatchArgs &a = a0;
	const dim3 grid(a.nreads);
	hipLaunchKernelGGL((k_ex_scan<false>), grid, dim3(WG), 0, s, a);
	if (fmt == EXF_EXZD)
		hipLaunchKernelGGL((k_ex_scan<true>), grid, dim3(WG), 0, s, a);
	hipLaunchKernelGGL(k_ex_section, grid, dim3(64), 0, s, a, fmt, huff ? 1 : 0);
	if (fmt == EXF_VBE21)
		hipLaunchKernelGGL(k_ex_fill_vbe21, grid, dim3(64), 0, s, a);
	ktime_begin(0, s);
	if (huff)
		hipLaunchKernelGGL((k_low_encode<true>), grid, dim3(WG), 0, s, a);
	else
		hipLaunchKernelGGL((k_low_encode<false>), grid, dim3(WG), 0, s, a);
	ktime_end(0, s);
}

// k_ex_parse and, for the Huffman variants, the stream decode into a.low (timed as the
// dominant kernel of those methods)
void launch_ex_parse_huff(const DecodeArgs &a, int fmt, int ent, hipStream_t s)
{
	hipLaunchKernelGGL(k_ex_parse, dim3(a.nreads), dim3(64), 0, s, a, fmt, ent == 2 ? 3 : ent);
	if (fmt == EXF_VBE21)
		hipLaunchKernelGGL(k_ex_parse_fill_vbe21, dim3(a.nreads), dim3(64), 0, s, a);
	if (ent) {
		ktime_begin(1, s);
		if (ent == 2)
			launch_rcs_decode(a, s);
		else
			launch_huff_decode(a, s);
		ktime_end(1, s);
	}
}

void launch_ex_decode(const DecodeArgs &a, int fmt, bool huff, hipStream_t s)
{
	if (!a.nreads)
		return;
	const dim3 grid(a.nreads);
	launch_ex_parse_huff(a, fmt, huff, s);
	if (huff) {
		hipLaunchKernelGGL((k_low_decode<true>), grid, dim3(WG), 0, s, a);
	} else {
		ktime_begin(1, s);
		hipLaunchKernelGGL((k_low_decode<false>), grid, dim3(WG), 0, s, a);
		ktime_end(1, s);
	}
}

} // namespace ph
