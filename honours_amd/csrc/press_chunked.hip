// press_chunked.hip - v2 svb16 / svb32 kernels for gfx950: chunked, register resident,
// wave autonomous.
//
// A read is cut into chunks of CHUNK = 32768 samples; one 256-thread workgroup per chunk,
// handed out in ticket order.  Each of the 4 waves owns a contiguous quarter of the chunk
// (8192 samples = 16 sub-tiles of 512 samples; a lane holds 8 samples = one dwordx4 per
// sub-tile), so
//   * all 16 loads of a lane are issued back to back (64 KiB in flight per workgroup) and
//     the zig-zag-delta values stay in registers as packed 16-bit pairs
//     (v_alignbit + v_pk_sub_i16 + v_pk_lshlrev/ashrrev + v_xor per pair);
//   * a sub-tile without exceptions (98 % of them on NA12878-like data) is written with
//     two v_perm_b32 and ONE 8-byte global store per lane at an unaligned address
//     (unaligned dwordx2 stores run at 0.9x the aligned rate on MI355X,
//     tools/ubench_unaligned.hip); a sub-tile with exceptions takes a wave scan and the
//     few lanes that hold an exception store byte-wise.  No LDS staging, no barrier in
//     the data path;
//   * what chains the chunks of a read is only the NUMBER of exceptions before the chunk
//     (the byte offset of sample i is i + #exceptions before i): a decoupled look-back
//     over 8-byte {status,count} granules, each written by one relaxed agent-scope atomic
//     store (cdna_hip_programming.md Guideline 16, form R2: the data is the flag).
//     Ticket order makes the look-back deadlock free: every predecessor of a chunk was
//     taken by a workgroup that is already running.
//
// Formats: svb16/encode.hpp:11, encode_scalar.hpp:14, decode.hpp:23, decode_scalar.hpp:31;
// streamvbyte_encode.c:70, streamvbyte_decode.c:62; press.c:1573-1611, 1678-1694.

#include "press_internal.h"
#include "press_packed.h"

namespace ph {

constexpr int CWG = 256;                     // threads per workgroup
constexpr int CK = 16;                       // sub-tiles per wave
constexpr uint32_t SUB = 512;                // samples per sub-tile (64 lanes x 8)
constexpr uint32_t WAVE_SAMPLES = CK * SUB;  // 8192
static_assert(WAVE_SAMPLES * 4 == CHUNK, "chunk = 4 waves");

#ifndef PERSISTENT_GRID_N
#define PERSISTENT_GRID_N 1024 // = what is resident (4 workgroups of ~110 VGPRs per CU); 768-1280 measured equal, 2048 2.5 % slower
#endif
constexpr uint32_t PERSISTENT_GRID = PERSISTENT_GRID_N;   // workgroups of the ticket-loop kernel (k_svb_encode_chunked)
constexpr uint64_t G_A = 1ull << 62;         // granule: aggregate of this chunk only
constexpr uint64_t G_P = 2ull << 62;         // granule: inclusive prefix up to this chunk
constexpr uint64_t G_MASK = (1ull << 62) - 1;
constexpr uint64_t CFAIL64 = ~0ull;
constexpr uint32_t CFAIL32 = 0xFFFFFFFFu;

typedef short s16x2 __attribute__((ext_vector_type(2)));

// wave-uniform values: tell the compiler (scalar registers, scalar branches)
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v)
{
	return ((uint64_t) uni((uint32_t) (v >> 32)) << 32) | uni((uint32_t) v);
}

// 16 bytes of a stream that is read once: non-temporal policy (tools/ubench_stream.hip: a read-only sweep reaches
// 7.1 TB/s with it against 6.3 TB/s with the default policy, a copy 6.0 against 5.2-5.6)
#ifndef PRESS_NO_NT
__device__ __forceinline__ uint4 ld16_stream(const void *p)
{
	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
	return make_uint4(v.x, v.y, v.z, v.w);
}
#else
__device__ __forceinline__ uint4 ld16_stream(const void *p) { return *reinterpret_cast<const uint4 *>(p); }
#endif
// 8 bytes at any byte address of a stream that is read once
#ifndef PRESS_NO_NT
__device__ __forceinline__ unsigned long long ld8_stream(const void *p)
{
	typedef unsigned long long __attribute__((aligned(1))) u64_u;
	return __builtin_nontemporal_load(reinterpret_cast<const u64_u *>(p));
}
#else
__device__ __forceinline__ unsigned long long ld8_stream(const void *p)
{
	unsigned long long v;
	__builtin_memcpy(&v, p, 8);
	return v;
}
#endif
// ... and 16 bytes of a stream that is written once and not read again by this call
#if !defined(PRESS_NO_NT) && !defined(PRESS_NO_NT_ST)
__device__ __forceinline__ void st16_stream(void *p, uint4 v)
{
	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	__builtin_nontemporal_store((u32x4){ v.x, v.y, v.z, v.w }, reinterpret_cast<u32x4 *>(p));
}
#else
__device__ __forceinline__ void st16_stream(void *p, uint4 v) { *reinterpret_cast<uint4 *>(p) = v; }
#endif

__device__ __forceinline__ uint32_t wave_incl_scan32(uint32_t v)
{
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t t = __shfl_up(v, d, 64);
		if (lane >= d)
			v += t;
	}
	return v;
}

// value of the previous lane (lane 0 receives `lane0`)
__device__ __forceinline__ uint32_t prev_lane(uint32_t v, uint32_t lane0)
{
	// DPP wave_shr:1 - lane l reads lane l-1; lane 0 keeps `old`
	return (uint32_t) __builtin_amdgcn_update_dpp((int) lane0, (int) v, 0x138, 0xf, 0xf, false);
}

// zig-zag delta of the two samples packed in `cur`, given the dword holding the two
// samples before them (trans.c:75,215 on packed 16-bit lanes)
__device__ __forceinline__ uint32_t zd_pair(uint32_t cur, uint32_t prevdw)
{
	const uint32_t sh = __builtin_amdgcn_alignbit(cur, prevdw, 16); // [prev.hi, cur.lo]
	const s16x2 d = __builtin_bit_cast(s16x2, cur) - __builtin_bit_cast(s16x2, sh);
	const s16x2 z = (d << 1) ^ (d >> 15);
	return __builtin_bit_cast(uint32_t, z);
}

// the same, and which of the two differences do not fit 16 bits (bit 15 / bit 31 of `ov`): slow5lib
// takes the delta in 32 bits (streamvbyte_zigzag.c:15), so such a value is 17 bits wide.
// If d16 is the wrapped difference and z16 its zig-zag, the true zig-zag is (~z16 & 0xFFFF) | 0x10000.
__device__ __forceinline__ uint32_t zd_pair_ovf(uint32_t cur, uint32_t prevdw, uint32_t &ov)
{
	const uint32_t sh = __builtin_amdgcn_alignbit(cur, prevdw, 16); // [prev.hi, cur.lo]
	const s16x2 d = __builtin_bit_cast(s16x2, cur) - __builtin_bit_cast(s16x2, sh);
	const uint32_t du = __builtin_bit_cast(uint32_t, d);
	ov |= (cur ^ sh) & (cur ^ du) & 0x80008000u; // a - b overflows iff sign(a) != sign(b) and sign(a - b) != sign(a)
	const s16x2 z = (d << 1) ^ (d >> 15);
	return __builtin_bit_cast(uint32_t, z);
}

// relaxed agent-scope granule access (sc1: L2-coherent, bypasses this CU's L1)
__device__ __forceinline__ void gran_store(uint64_t *g, uint64_t v)
{
	__hip_atomic_store(g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t gran_load(uint64_t *g)
{
	return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Decoupled look-back, executed by ONE WAVE (all 64 lanes call it): publish this chunk's
// aggregate, sum the aggregates of the chunks before it in the same read - 64 granules per
// round trip, nearest first - and publish the inclusive prefix.  `t` = this chunk's ticket,
// j = its index in the read (chunk t-j is the read's first and always publishes a prefix).
// Returns the exclusive prefix (same value in every lane).  `last`: no chunk of this read
// follows, nothing needs publishing.
__device__ __forceinline__ uint64_t lookback(uint64_t *gran, uint32_t t, uint32_t j, uint64_t mine, bool last)
{
	const uint32_t lane = threadIdx.x & 63;
	if (j == 0) {
		if (!last && lane == 0)
			gran_store(gran + t, G_P | mine);
		return 0;
	}
	if (!last && lane == 0)
		gran_store(gran + t, G_A | mine);
	uint64_t sum = 0;
	uint32_t done = 0; // predecessors already summed
	for (;;) {
		// lane l looks at predecessor number done + l (0 = the chunk right before this one)
		const uint32_t idx = done + lane;
		const bool want = idx < j;
		uint64_t g = 0;
		if (want)
			g = gran_load(gran + (t - 1 - idx));
		const unsigned long long pmask = __ballot(want && (g >> 62) == 2);
		const unsigned long long zmask = __ballot(want && (g >> 62) == 0);
		// usable lanes: everything nearer than the first unpublished granule, up to and
		// including the first prefix
		uint32_t stop = 64;
		if (zmask)
			stop = (uint32_t) __builtin_ctzll(zmask);
		uint32_t firstp = 64;
		if (pmask)
			firstp = (uint32_t) __builtin_ctzll(pmask);
		const bool have_p = firstp < stop;
		const uint32_t take = have_p ? firstp + 1 : stop; // lanes [0, take) are added
		uint64_t v = (lane < take) ? (g & G_MASK) : 0;
#pragma unroll
		for (int dd = 32; dd >= 1; dd >>= 1) {
			const uint32_t lo = (uint32_t) __shfl_xor((int) (uint32_t) v, dd, 64);
			const uint32_t hi = (uint32_t) __shfl_xor((int) (uint32_t) (v >> 32), dd, 64);
			v += ((uint64_t) hi << 32) | lo;
		}
		sum += v;
		done += take;
		if (have_p || done >= j)
			break;
		if (take == 0)
			__builtin_amdgcn_s_sleep(2);
	}
	if (!last && lane == 0)
		gran_store(gran + t, G_P | (sum + mine));
	return sum;
}

// the fields of a chunk descriptor as wave-uniform (scalar) values
struct ChunkU {
	uint64_t sig_off, out_base;
	uint32_t n, j, read, cap_ok;
};
__device__ __forceinline__ ChunkU load_chunk(const ChunkDesc *dp)
{
	ChunkU c;
	c.sig_off = uni64(dp->sig_off);
	c.out_base = uni64(dp->out_base);
	c.n = uni(dp->n);
	c.j = uni(dp->j);
	c.read = uni(dp->read);
	c.cap_ok = uni(dp->cap_ok);
	return c;
}

// ------------------------------------------------------------------ chunk table

// One thread per read: a wave adds up its reads' chunk counts, takes a contiguous range of
// chunk ids with ONE atomic (ids need only be contiguous and ascending within a read),
// writes the descriptors and clears the look-back granules of its chunks.
template <bool DEC, bool KEY2>
__global__ __launch_bounds__(256) void k_chunk_prep(const uint64_t *off, const uint32_t *nsamp,
						    const uint64_t *slot_off, const uint64_t *in_len,
						    uint32_t nreads, ChunkDesc *chunks, uint64_t *gran,
						    ChunkCtl *ctl, uint32_t max_chunks, uint64_t *out_len,
						    uint32_t *out_n, uint32_t *first_chunk, ReadMeta *meta = nullptr,
						    const uint8_t *in = nullptr, uint32_t hdr = 0)
{
	const uint32_t r = blockIdx.x * 256 + threadIdx.x;
	uint32_t n = 0, nch = 0;
	if (r < nreads) {
		n = nsamp[r];
		nch = (n + CHUNK - 1) / CHUNK;
	}
	const uint32_t inc = wave_incl_scan32(nch);
	uint32_t base = 0;
	if ((threadIdx.x & 63) == 63 && inc)
		base = atomicAdd(&ctl->nchunks, inc);
	base = (uint32_t) __builtin_amdgcn_readlane((int) base, 63);
	const uint32_t first = base + inc - nch;
	if (r >= nreads)
		return;
	if (meta) {
		ReadMeta z;
		z.nex = z.ored = z.zd0 = z.q = z.hdr = z.seclen = z.nlow = 0;
		z.status = 0;
		meta[r] = z;
	}
	const uint32_t klen = KEY2 ? (n + 3) / 4 : (n >> 3) + (((n & 7) + 7) >> 3);
	const uint64_t sbase = slot_off[r];
	// hdr = 4: the slow5 svb-zd stream (u32 sample count in front, values of up to 3 bytes)
	uint32_t ok;
	if (!DEC) {
		ok = (uint64_t) hdr + klen + (hdr ? 3ull : 2ull) * n <= slot_off[r + 1] - sbase;
	} else {
		ok = (uint64_t) hdr + klen <= in_len[r];
		if (ok && hdr) { // slow5_press.c:1086: the count in the stream is what gets decoded
			const uint8_t *h = in + sbase;
			ok = ((uint32_t) h[0] | ((uint32_t) h[1] << 8) | ((uint32_t) h[2] << 16) | ((uint32_t) h[3] << 24)) == n;
		}
	}
	if (n == 0) {
		if (!DEC)
			out_len[r] = 0;
		else
			out_n[r] = 0;
	}
	const uint64_t o = off[r];
	if (first_chunk)
		first_chunk[r] = first;
	for (uint32_t j = 0; j < nch; j++) {
		if (first + j >= max_chunks)
			break; // cannot happen: max_chunks bounds the sum
		ChunkDesc d;
		d.sig_off = o;
		d.out_base = sbase;
		d.n = n;
		d.j = j;
		d.read = r;
		d.cap_ok = ok;
		d.ebefore = 0;
		d.ecnt[0] = d.ecnt[1] = d.ecnt[2] = d.ecnt[3] = 0;
		d.kmask[0] = d.kmask[1] = d.kmask[2] = d.kmask[3] = 0;
		chunks[first + j] = d;
		gran[first + j] = 0;
	}
}

// ------------------------------------------------------------------ encode

// S5 (with KEY2, ZD): slow5lib's "svb-zd" signal codec (slow5_press.c:1054): a u32 sample count
// in front, and the delta taken in 32 bits - a jump of more than 32767 is a 3-byte value.
// HIST (the zstd compositions, press_zstd.hip): the data bytes are counted per read (BatchArgs::zhist) while a lane
// holds them - the Huffman table of the read's frame needs their histogram, and a kernel of its own read the stream again
// for it (0.36 ms of config 3's press call).  16 copies of the counters in LDS (4 per wave, by lane): nanopore deltas
// are peaked, and lanes that hit one counter in one instruction are served one after the other.
template <bool KEY2, bool ZD, bool S5 = false, bool HIST = false>
__global__ __launch_bounds__(CWG) void k_svb_encode_chunked(BatchArgs a)
{
	static_assert(!S5 || (KEY2 && ZD), "the slow5 variant is svb32 of zig-zag deltas");
	static_assert(!HIST || !S5, "the histogram is for the two-byte formats");
	__shared__ uint32_t s_ticket;
	__shared__ uint32_t s_wtot[4];
	__shared__ uint64_t s_excl;
	__shared__ uint32_t s_hist[HIST ? 16 : 1][256];
	__shared__ uint32_t s_knz[4];
	uint32_t *myhist = s_hist[HIST ? 4 * (threadIdx.x >> 6) + (threadIdx.x & 3) : 0];
	if (HIST)
		for (int i = 0; i < 16; i++)
			s_hist[i][threadIdx.x] = 0; // (the ticket barrier orders this in front of every count)

	// persistent workgroups: chunks are handed out in ticket order (what makes the
	// look-back deadlock free).  (Fetching the next ticket early was measured slower.)
	const uint32_t nchunks = uni(a.ctl->nchunks);
	for (;;) {
	if (threadIdx.x == 0)
		s_ticket = atomicAdd(&a.ctl->ticket, 1u);
	__syncthreads();
	const uint32_t t = uni(s_ticket);
	if (t >= nchunks)
		break;
	const ChunkU d = load_chunk(a.chunks + t);
	const uint32_t n = d.n;
	const uint32_t first = d.j * CHUNK;          // first sample of the chunk within the read
	const bool last = first + CHUNK >= n;
	const int lane = threadIdx.x & 63;
	const int w = (int) uni(threadIdx.x >> 6);

	if (!d.cap_ok) { // slot smaller than the worst case of the format: fail the read
		if (threadIdx.x < 64) {
			(void) lookback(a.gran, t, d.j, 0, last);
			if (last && threadIdx.x == 0)
				a.out_len[d.read] = CFAIL64;
			if (HIST && last && threadIdx.x == 0)
				a.zkcnt[t - d.j] = 0;
		}
		__syncthreads(); // every wave has read s_ticket before thread 0 overwrites it
		continue;
	}

	const int16_t *in = a.sig + d.sig_off;
	uint8_t *out = a.out + d.out_base + (S5 ? 4 : 0);
	if (S5 && first == 0 && threadIdx.x == 0) { // slow5_press.c:1046: the sample count
		uint8_t *h = a.out + d.out_base;
		h[0] = (uint8_t) n;
		h[1] = (uint8_t) (n >> 8);
		h[2] = (uint8_t) (n >> 16);
		h[3] = (uint8_t) (n >> 24);
	}
	const uint32_t klen = KEY2 ? (n + 3) / 4 : (n >> 3) + (((n & 7) + 7) >> 3);
	const uint32_t ws = first + w * WAVE_SAMPLES; // first sample of this wave's quarter

	// ---- phase 1: load everything, zig-zag delta in registers, count exceptions
	uint4 z[CK];
#pragma unroll
	for (int k = 0; k < CK; k++) {
		const uint32_t i0 = ws + k * SUB + lane * 8;
		z[k] = make_uint4(0, 0, 0, 0);
		if (i0 < n)
			z[k] = ld16_stream(in + i0);
	}
	uint32_t carry = 0; // dword holding the sample in front of lane 0's first sample
	if (ZD && ws > 0 && ws < n)
		carry = (uint32_t) (uint16_t) in[ws - 1] << 16;

	uint32_t kmask = 0;  // sub-tiles that need the slow path (exceptions or a ragged tail)
	uint32_t omask = 0;  // S5: sub-tiles with a 17-bit value
	uint32_t etot = 0;   // exceptions in this wave's quarter (uniform)
	uint32_t knz1 = 0;   // HIST: key bytes of this wave's quarter that are not zero (uniform)
#pragma unroll
	for (int k = 0; k < CK; k++) {
		const uint32_t i0 = ws + k * SUB + lane * 8;
		const uint32_t raw_w = z[k].w;
		uint32_t ov[4] = { 0, 0, 0, 0 };
		if (ZD) {
			const uint32_t pw = prev_lane(raw_w, carry);
			uint32_t z0, z1, z2, z3;
			if (S5) {
				z0 = zd_pair_ovf(z[k].x, pw, ov[0]);
				z1 = zd_pair_ovf(z[k].y, z[k].x, ov[1]);
				z2 = zd_pair_ovf(z[k].z, z[k].y, ov[2]);
				z3 = zd_pair_ovf(z[k].w, z[k].z, ov[3]);
			} else {
				z0 = zd_pair(z[k].x, pw);
				z1 = zd_pair(z[k].y, z[k].x);
				z2 = zd_pair(z[k].z, z[k].y);
				z3 = zd_pair(z[k].w, z[k].z);
			}
			z[k] = make_uint4(z0, z1, z2, z3);
			carry = (uint32_t) __builtin_amdgcn_readlane((int) raw_w, 63);
		}
		// samples at or beyond n must not count: zero them
		if (i0 + 8 > n) {
			const uint32_t nv = i0 < n ? n - i0 : 0;
			uint32_t zz[4] = { z[k].x, z[k].y, z[k].z, z[k].w };
#pragma unroll
			for (int q = 0; q < 4; q++) {
				if (nv <= (uint32_t) (2 * q))
					zz[q] = 0;
				else if (nv == (uint32_t) (2 * q + 1))
					zz[q] &= 0xFFFFu;
			}
			z[k] = make_uint4(zz[0], zz[1], zz[2], zz[3]);
			if (S5) { // ... and their differences cannot overflow
#pragma unroll
				for (int q = 0; q < 4; q++) {
					if (nv <= (uint32_t) (2 * q))
						ov[q] = 0;
					else if (nv == (uint32_t) (2 * q + 1))
						ov[q] &= 0xFFFFu;
				}
			}
		}
		const uint32_t hi = (z[k].x | z[k].y | z[k].z | z[k].w) & 0xFF00FF00u;
		const bool ragged = i0 < n && i0 + 8 > n;
		const unsigned long long bo = S5 ? __ballot((ov[0] | ov[1] | ov[2] | ov[3]) != 0) : 0ull;
		const unsigned long long bx = __ballot(hi != 0) | bo;
		const unsigned long long br = __ballot(ragged);
		if (bx | br)
			kmask |= 1u << k;
		if (bo)
			omask |= 1u << k;
		if (bx) {
			if (HIST) { // key bytes that will not be zero: a lane's eight values are one (svb16) or two (svb32) of them
				if (KEY2)
					knz1 += (uint32_t) (__popcll(__ballot(((z[k].x | z[k].y) & 0xFF00FF00u) != 0)) +
							     __popcll(__ballot(((z[k].z | z[k].w) & 0xFF00FF00u) != 0)));
				else
					knz1 += (uint32_t) __popcll(__ballot(hi != 0));
			}
			// exact count of the extra bytes: one popcount of a ballot per key bit; a 17-bit
			// value has two of them whatever its low half looks like
			const uint32_t zz[4] = { z[k].x, z[k].y, z[k].z, z[k].w };
#pragma unroll
			for (int q = 0; q < 4; q++) {
				const bool o0 = S5 && (ov[q] & 0x00008000u), o1 = S5 && (ov[q] & 0x80000000u);
				etot += (uint32_t) __popcll(__ballot(!o0 && (zz[q] & 0x0000FF00u) != 0));
				etot += (uint32_t) __popcll(__ballot(!o1 && (zz[q] & 0xFF000000u) != 0));
				if (S5 && bo)
					etot += 2u * (uint32_t) (__popcll(__ballot(o0)) + __popcll(__ballot(o1)));
			}
		}
	}

	// ---- exceptions before this wave: within the chunk (LDS) and before the chunk (look-back)
	if (lane == 0) {
		s_wtot[w] = etot;
		if (HIST)
			s_knz[w] = knz1;
	}
	__syncthreads();
	const uint32_t t0 = uni(s_wtot[0]), t1 = uni(s_wtot[1]), t2 = uni(s_wtot[2]), t3 = uni(s_wtot[3]);
	// HIST: the look-back carries two counts - exceptions below bit 40, key bytes that are not zero above (a read has
	// fewer than 2^32 of the first and 2^22 of the second)
	const uint32_t q0 = HIST ? uni(s_knz[0]) : 0u, q1 = HIST ? uni(s_knz[1]) : 0u, q2 = HIST ? uni(s_knz[2]) : 0u,
		       q3 = HIST ? uni(s_knz[3]) : 0u;
	if (w == 0) {
		const uint64_t e = lookback(a.gran, t, d.j, ((uint64_t) t0 + t1 + t2 + t3) | ((uint64_t) (q0 + q1 + q2 + q3) << 40), last);
		if (lane == 0)
			s_excl = e;
	}
	__syncthreads();
	const uint64_t ebefore = uni64(s_excl) & ((1ull << 40) - 1);
	uint32_t kbefore = (uint32_t) (uni64(s_excl) >> 40) + (w > 0 ? q0 : 0u) + (w > 1 ? q1 : 0u) + (w > 2 ? q2 : 0u); // list slot of the wave's next key byte
	uint64_t ebase = ebefore + (w > 0 ? t0 : 0u) + (w > 1 ? t1 : 0u) + (w > 2 ? t2 : 0u);
	if (last && threadIdx.x == 0)
		a.out_len[d.read] = (uint64_t) (S5 ? 4 : 0) + klen + n + ebefore + t0 + t1 + t2 + t3;

	// ---- phase 2: keys and data
	uint8_t *data = out + klen;

#pragma unroll
	for (int k = 0; k < CK; k++) {
		const uint32_t i0 = ws + k * SUB + lane * 8;
		if (ws + k * SUB >= n)
			continue; // uniform: nothing left in this wave's quarter
		const uint32_t zz[4] = { z[k].x, z[k].y, z[k].z, z[k].w };
		const uint32_t lo0 = __builtin_amdgcn_perm(zz[1], zz[0], 0x06040200);
		const uint32_t lo1 = __builtin_amdgcn_perm(zz[3], zz[2], 0x06040200);
		if (!((kmask >> k) & 1u)) {
			// fast path: no exception, no ragged tail in the whole sub-tile (lanes are
			// either complete or entirely beyond the end of the read)
			if (i0 < n) {
				if (!KEY2)
					out[i0 >> 3] = 0;
				else
					__builtin_memset(out + (i0 >> 2), 0, 2);
				uint2 v;
				v.x = lo0;
				v.y = lo1;
				__builtin_memcpy(data + ebase + i0, &v, 8);
				if (HIST) {
#pragma unroll
					for (int e = 0; e < 4; e++) {
						atomicAdd(&myhist[(lo0 >> (8 * e)) & 0xFFu], 1u);
						atomicAdd(&myhist[(lo1 >> (8 * e)) & 0xFFu], 1u);
					}
				}
			}
			continue;
		}
		// slow path
		uint32_t key = 0;
#pragma unroll
		for (int q = 0; q < 4; q++) {
			key |= ((zz[q] & 0x0000FF00u) ? 1u : 0u) << (2 * q);
			key |= ((zz[q] & 0xFF000000u) ? 1u : 0u) << (2 * q + 1);
		}
		const uint32_t nv = i0 < n ? min(8u, n - i0) : 0u;
		uint32_t wide = 0; // S5: which of the lane's values are 17 bits wide (from the samples again: rare)
		if (S5 && ((omask >> k) & 1u) && nv) {
			const uint4 raw = ld16_stream(in + i0);
			const uint32_t pv = i0 ? (uint32_t) (uint16_t) in[i0 - 1] << 16 : 0u;
			uint32_t ov[4] = { 0, 0, 0, 0 };
			(void) zd_pair_ovf(raw.x, pv, ov[0]);
			(void) zd_pair_ovf(raw.y, raw.x, ov[1]);
			(void) zd_pair_ovf(raw.z, raw.y, ov[2]);
			(void) zd_pair_ovf(raw.w, raw.z, ov[3]);
#pragma unroll
			for (int q = 0; q < 4; q++) {
				wide |= ((ov[q] >> 15) & 1u) << (2 * q);
				wide |= ((ov[q] >> 31) & 1u) << (2 * q + 1);
			}
			wide &= (1u << nv) - 1u;
			key &= ~wide;
		}
		// the lane's key byte(s)
		uint32_t k0 = key, k1 = 0;
		if (KEY2) {
			k0 = 0;
#pragma unroll
			for (int q = 0; q < 4; q++) {
				k0 |= (((key >> q) & 1u) | (((wide >> q) & 1u) << 1)) << (2 * q);
				k1 |= (((key >> (q + 4)) & 1u) | (((wide >> (q + 4)) & 1u) << 1)) << (2 * q);
			}
		}
		if (!nv)
			k0 = 0;
		if (nv <= 4)
			k1 = 0;
		const uint32_t cnt = __popc(key) + 2u * __popc(wide);
		const uint32_t kc = HIST ? (k0 != 0) + (k1 != 0) : 0u; // (one scan for both: a sub-tile has at most 1536 extra bytes)
		const uint32_t inc2 = wave_incl_scan32(cnt | (kc << 16));
		const uint32_t inc = inc2 & 0xFFFFu;
		const uint32_t tot2 = (uint32_t) __builtin_amdgcn_readlane((int) inc2, 63);
		const uint32_t tot = tot2 & 0xFFFFu;
		uint8_t *p = data + ebase + i0 + (inc - cnt);
		if (HIST && kc) { // the read's list of key bytes that are not zero (press_zstd.hip: RLE blocks between them)
			uint32_t slot = kbefore + (inc2 >> 16) - kc;
			const uint32_t kpos = KEY2 ? i0 >> 2 : i0 >> 3;
			if (k0) {
				a.ex_pos[d.sig_off + slot] = kpos;
				a.ex_val[d.sig_off + slot] = k0;
				slot++;
			}
			if (k1) {
				a.ex_pos[d.sig_off + slot] = kpos + 1;
				a.ex_val[d.sig_off + slot] = k1;
			}
		}
		if (HIST)
			kbefore += tot2 >> 16;
		if (nv) {
			if (!KEY2) {
				out[i0 >> 3] = (uint8_t) key;
			} else {
				out[i0 >> 2] = (uint8_t) k0;
				if (nv > 4)
					out[(i0 >> 2) + 1] = (uint8_t) k1;
			}
			if (HIST) {
#pragma unroll
				for (int q = 0; q < 8; q++) {
					if ((uint32_t) q < nv) {
						const uint32_t val = (zz[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
						atomicAdd(&myhist[val & 0xFFu], 1u);
						if (val > 255u)
							atomicAdd(&myhist[val >> 8], 1u);
					}
				}
			}
			if (nv == 8 && cnt == 0) {
				uint2 v;
				v.x = lo0;
				v.y = lo1;
				__builtin_memcpy(p, &v, 8);
			} else {
#pragma unroll
				for (int q = 0; q < 8; q++) {
					if ((uint32_t) q < nv) {
						uint32_t val = (zz[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
						if (S5 && ((wide >> q) & 1u))
							val = (~val & 0xFFFFu) | 0x10000u;
						*p++ = (uint8_t) val;
						if (val > 255u)
							*p++ = (uint8_t) (val >> 8);
						if (val > 65535u)
							*p++ = (uint8_t) (val >> 16);
					}
				}
			}
		}
		ebase += tot;
	}
	if (HIST) { // the chunk's counts to its read's; a thread clears what it has read (the next chunk counts behind
		    // three barriers)
		__syncthreads();
		if (last && threadIdx.x == 0) // the read's count of such key bytes (k_zs_table takes it from its first chunk's slot)
			a.zkcnt[t - d.j] = (uint32_t) (uni64(s_excl) >> 40) + q0 + q1 + q2 + q3;
		uint32_t c = 0;
		for (int i = 0; i < 16; i++) {
			c += s_hist[i][threadIdx.x];
			s_hist[i][threadIdx.x] = 0;
		}
		if (c)
			atomicAdd(&a.zhist[(uint64_t) d.read * 256 + threadIdx.x], c);
	}
	} // ticket loop
}

// ------------------------------------------------------------------ decode
//
// Chunk j of a read needs two things from the chunks before it: the number of exceptions
// (its data bytes start at klen + j*CHUNK + E) and the value of the last sample (the
// deltas are a running sum, trans.c:260).  E comes from the key bytes alone, so it is
// published before any data is touched; the sample value is published after the chunk
// has summed its own deltas.  Two look-back chains over the same ticket order.
//
// The payload a lane keeps between the phases is the COMPRESSED form (8 bytes per 8
// samples, 32 VGPRs per chunk quarter) plus one 16-bit base per sub-tile; values are
// expanded twice (once to sum the deltas, once to store) - ALU is cheap, registers are
// what bounds the number of chunks in flight per CU.

// key bits (svb16) / 2-bit codes (svb32) of the 8 samples at i0, masked to the valid ones
template <bool KEY2>
__device__ __forceinline__ uint32_t load_key(const uint8_t *in, uint32_t i0, uint32_t n)
{
	uint32_t kk = 0;
	if (i0 < n) {
		if (!KEY2) {
			kk = in[i0 >> 3];
		} else {
			kk = in[i0 >> 2];
			if (i0 + 4 < n)
				kk |= (uint32_t) in[(i0 >> 2) + 1] << 8;
		}
		const uint32_t nv = n - i0;
		if (nv < 8)
			kk &= KEY2 ? ((1u << (2 * nv)) - 1u) : ((1u << nv) - 1u);
	}
	return kk;
}

template <bool KEY2>
__device__ __forceinline__ uint32_t key_extra_bytes(uint32_t kk)
{
	if (!KEY2)
		return __popc(kk);
	uint32_t c = 0;
#pragma unroll
	for (int q = 0; q < 8; q++)
		c += (kk >> (2 * q)) & 3u;
	return c;
}

// Slow path of one sub-tile (exceptions, a ragged tail, or the end of the stream is near):
// byte-wise, bounds-checked gather of the lane's values into packed 16-bit pairs.
// `eb` = exceptions in front of the sub-tile; returns the sub-tile's exception count.
// S5: values may be 17 bits wide (slow5lib); bit q of `hib` = bit 16 of the lane's value q.
template <bool KEY2, bool S5 = false>
__device__ __forceinline__ uint32_t gather_slow(const uint8_t *in, const uint8_t *data, uint64_t dlen,
						uint32_t i0, uint32_t n, uint64_t eb, uint32_t v[4], uint32_t &hib)
{
	hib = 0;
	const uint32_t kk = load_key<KEY2>(in, i0, n);
	const uint32_t c = key_extra_bytes<KEY2>(kk);
	const uint32_t inc = wave_incl_scan_dpp(c);
	const uint32_t tot = (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
	uint64_t p = eb + i0 + (inc - c);
	const uint32_t nv = i0 < n ? min(8u, n - i0) : 0u;
	v[0] = v[1] = v[2] = v[3] = 0;
#pragma unroll
	for (int q = 0; q < 8; q++) {
		if ((uint32_t) q < nv) {
			const uint32_t code = KEY2 ? ((kk >> (2 * q)) & 3u) : ((kk >> q) & 1u);
			uint32_t val = p < dlen ? data[p] : 0u;
			if (code >= 1)
				val |= (p + 1 < dlen ? (uint32_t) data[p + 1] : 0u) << 8;
			if (S5 && code >= 2)
				hib |= (p + 2 < dlen ? (uint32_t) data[p + 2] & 1u : 0u) << q;
			p += 1 + code;
			v[q >> 1] |= val << (16 * (q & 1));
		}
	}
	return tot;
}

// inverse zig-zag of pair q of a gathered lane; a value's bit 16 ends up in bit 15 of z >> 1
__device__ __forceinline__ uint32_t unzz_pair_hib(uint32_t z, uint32_t hib, int q)
{
	return unzz_pair(z) ^ ((((hib >> (2 * q)) & 1u) << 15) | (((hib >> (2 * q + 1)) & 1u) << 31));
}

// ---- key scan: the exception counts depend on the key bytes only (n/8 bytes, 4 % of the
// stream), so they are settled by two tiny kernels before any data is touched; the main
// kernel then starts its data loads straight from the chunk descriptor.

// One workgroup per chunk: per wave quarter, which sub-tiles are not plain and how many
// extra data bytes (exceptions) the quarter holds.  A wave quarter's key bytes are contiguous
// (1 KiB for svb16, 2 KiB for svb32): each lane takes 16 of them with one (unaligned) 16-byte
// load, the count is one popcount reduction and the per-sub-tile flags come from one ballot.
template <bool KEY2, bool S5 = false>
__global__ __launch_bounds__(CWG) void k_svb_keyscan(DecodeArgs a)
{
	// one WAVE per chunk, its four quarters one after the other (a workgroup per chunk: four times the waves for
	// one 16-byte load per lane each)
	const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (c >= uni(a.ctl->nchunks))
		return;
	ChunkDesc *dp = a.chunks + c;
	if (!uni(dp->cap_ok))
		return;
	const uint32_t n = uni(dp->n);
	const uint8_t *in = a.in + uni64(dp->out_base) + (S5 ? 4 : 0);
	const uint64_t in_len = uni64(a.in_len[uni(dp->read)]) - (S5 ? 4 : 0);
	const uint32_t klen = KEY2 ? (n + 3) / 4 : (n >> 3) + (((n & 7) + 7) >> 3);
	const int lane = threadIdx.x & 63;
	constexpr int NL = KEY2 ? 2 : 1;            // 16-byte loads per lane
	constexpr uint32_t SPB = KEY2 ? 4 : 8;      // samples per key byte
	const uint32_t cj = uni(dp->j);
#pragma unroll 1
	for (int w = 0; w < 4; w++) {
	const uint32_t ws = cj * CHUNK + w * WAVE_SAMPLES;
	if (ws >= n && w)
		break; // (the read ends in an earlier quarter: nothing here - the descriptor's zeros stand)
	uint32_t cnt = 0;
	uint32_t kmask = 0;
	uint32_t bad = 0;
#pragma unroll
	for (int h = 0; h < NL; h++) {
		// key bytes [kb, kb+16) of the stream = samples [kb*SPB, (kb+16)*SPB)
		const uint32_t kb = ws / SPB + h * 1024 + lane * 16;
		uint32_t q[4] = { 0, 0, 0, 0 };
		if (kb < klen) {
			// (the stream's very last key byte may carry bits of samples beyond n: byte path)
			if ((uint64_t) kb + 16 <= in_len && (kb + 16 < klen || (kb + 16 == klen && n % SPB == 0))) {
				uint4 v;
				__builtin_memcpy(&v, in + kb, 16);
				q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
			} else {
				for (uint32_t b = 0; b < 16 && kb + b < klen; b++) {
					uint32_t kk = in[kb + b];
					// bits of samples at or beyond n do not count
					const uint32_t s0 = (kb + b) * SPB;
					if (s0 + SPB > n)
						kk &= (1u << ((n - s0) * (KEY2 ? 2 : 1))) - 1u;
					q[b >> 2] |= kk << (8 * (b & 3));
				}
			}
		}
		uint32_t any = q[0] | q[1] | q[2] | q[3];
		if (!KEY2) {
			cnt += __popc(q[0]) + __popc(q[1]) + __popc(q[2]) + __popc(q[3]);
		} else {
#pragma unroll
			for (int i = 0; i < 4; i++) {
				// sum of the 2-bit codes: low bits + 2 x high bits
				cnt += __popc(q[i] & 0x55555555u) + 2 * __popc(q[i] & 0xAAAAAAAAu);
				bad |= q[i] & 0xAAAAAAAAu;
			}
		}
		// sub-tile k of this wave = key bytes [k*64/NL', ...): svb16: 64 bytes = 4 lanes;
		// svb32: 128 bytes = 8 lanes of this half... lanes are ordered by key byte
		const unsigned long long nz = __ballot(any != 0);
		if (!KEY2) {
#pragma unroll
			for (int k = 0; k < CK; k++)
				if ((nz >> (4 * k)) & 0xFull)
					kmask |= 1u << k;
		} else {
#pragma unroll
			for (int k = 0; k < CK / 2; k++)
				if ((nz >> (8 * k)) & 0xFFull)
					kmask |= 1u << (k + h * (CK / 2));
		}
	}
	// ragged tail: the sub-tile that holds sample n-1 when n is not a multiple of 8
	if ((n & 7) && n > ws && n - ws <= WAVE_SAMPLES)
		kmask |= 1u << ((n - 1 - ws) / SUB);
	const uint32_t inc = wave_incl_scan_dpp(cnt);
	uint64_t etot = (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
	// 3- and 4-byte codes cannot come from a 16-bit signal: poison the count so that the
	// read fails its length check and its offsets fall out of range
	if (KEY2 && !S5 && __ballot(bad != 0))
		etot += 1u << 30;
	if (lane == 0) {
		dp->ecnt[w] = (uint32_t) (etot > 0xFFFFFFFFull ? 0xFFFFFFFFull : etot);
		dp->kmask[w] = (uint16_t) kmask;
	}
	}
}

// One thread per read: exclusive prefix of the chunk counts, and the verdict on the stream
// length (decode.hpp:23 consumes klen + n + #exceptions bytes).
template <bool KEY2, bool S5 = false>
__global__ __launch_bounds__(256) void k_svb_keyprefix(DecodeArgs a)
{
	// one wave per read (see k_ex_prefix)
	const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
	const uint32_t lane = threadIdx.x & 63;
	if (r >= a.nreads)
		return;
	const uint32_t n = a.nsamp[r];
	if (n == 0)
		return; // out_n[r] = 0 written by k_chunk_prep
	const uint32_t klen = KEY2 ? (n + 3) / 4 : (n >> 3) + (((n & 7) + 7) >> 3);
	const uint64_t in_len = a.in_len[r] - (S5 ? 4 : 0);
	ChunkDesc *dp = a.chunks + a.first_chunk[r];
	if (a.in_len[r] < (S5 ? 4u : 0u) || klen > in_len || !dp->cap_ok) { // cap_ok: k_chunk_prep (S5: the count in the stream)
		if (lane == 0)
			a.out_n[r] = CFAIL32;
		return;
	}
	const uint32_t nch = (n + CHUNK - 1) / CHUNK;
	uint64_t e = 0;
	for (uint32_t j0 = 0; j0 < nch; j0 += 64) {
		const uint32_t j = j0 + lane;
		// (a chunk's count may carry k_svb_keyscan's poison bit 30: the low 20 bits and the rest are summed apart)
		const uint64_t c = j < nch ? (uint64_t) dp[j].ecnt[0] + dp[j].ecnt[1] + dp[j].ecnt[2] + dp[j].ecnt[3] : 0ull;
		const uint32_t lo = wave_incl_scan_dpp((uint32_t) (c & 0xFFFFFu)), hi = wave_incl_scan_dpp((uint32_t) (c >> 20));
		const uint64_t inc = (uint64_t) lo + ((uint64_t) hi << 20);
		if (j < nch)
			dp[j].ebefore = e + inc - c;
		e += (uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) lo, 63) +
		     ((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) hi, 63) << 20);
	}
	// slow5_press.c:1098: the decoder must consume exactly the bytes it was given
	if (lane == 0)
		a.out_n[r] = (S5 ? (uint64_t) klen + n + e == in_len : (uint64_t) klen + n + e <= in_len) ? n : CFAIL32;
}

#ifdef DEC_STAMPS
// diagnostic build only: per-chunk phase timestamps (s_memtime), read back by tools
__device__ uint64_t g_stamps[65536 * 8];
#define STAMP(i)                                                                         \
	do {                                                                             \
		if (threadIdx.x == 0 && t < 65536)                                       \
			g_stamps[t * 8 + (i)] = __builtin_amdgcn_s_memtime();             \
	} while (0)
extern "C" int press_hip_debug_stamps(uint64_t *dst, uint32_t nwords)
{
	return (int) hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), (size_t) nwords * 8);
}
#else
#define STAMP(i)
#endif

// make the LDS writes of this wave visible to its other lanes (DS ops of a wave execute in
// order; this only stops the compiler from moving them)
__device__ __forceinline__ void wave_lds_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Main decode kernel.  Structure per wave quarter (16 sub-tiles):
//   * the plain sub-tiles (no exception, no ragged tail, not at the end of the stream) are
//     handled by fully unrolled loops that contain no load except their own 8-byte data
//     load - so the compiler needs no conservative s_waitcnt between the 16 stores;
//   * the few other sub-tiles go through ROLLED loops over the set bits of `kmask`
//     (byte-wise gather, one copy of the code);
//   * both write their per-lane and per-sub-tile running sums to a small wave-private LDS
//     table, which also frees the registers a per-sub-tile array would take.
template <bool KEY2, bool ZD, bool S5 = false>
__global__ __launch_bounds__(CWG) void k_svb_decode_chunked(DecodeArgs a)
{
#ifdef DEC_STAMPS
	const uint64_t t_start = __builtin_amdgcn_s_memtime();
#endif
	__shared__ uint32_t s_ticket;
	__shared__ uint32_t s_wsum[4];
	__shared__ uint32_t s_sbase;
	__shared__ uint16_t s_xl[4][CK][64];  // delta sum in front of the lane, within its sub-tile
	__shared__ uint32_t s_sub[4][CK];     // per sub-tile: exception count, later delta total / base
	__shared__ uint32_t s_epre[4][CK];    // exceptions of the wave's earlier sub-tiles

	const int lane = threadIdx.x & 63;
	const int w = (int) uni(threadIdx.x >> 6);
	if (blockIdx.x >= uni(a.ctl->nchunks))
		return; // (the grid is an upper bound: surplus workgroups draw no ticket - blockIdx only COUNTS the workgroups)
#ifdef DEC_BLOCKIDX_TICKET
	// DIAGNOSTIC BUILD ONLY (tools/build_variants.sh "tkb:-DDEC_BLOCKIDX_TICKET"): what the ticket in front of every
	// workgroup costs.  Chunk ids from blockIdx rely on workgroups starting in the order of their ids, which HIP does not
	// promise - the look-back can deadlock where they do not.  Measured: 0.58-0.61 instead of 0.66-0.69 ms.
	const uint32_t t = blockIdx.x;
	(void) s_ticket;
#else
	if (threadIdx.x == 0)
		s_ticket = atomicAdd(&a.ctl->ticket, 1u);
	__syncthreads();
	const uint32_t t = uni(s_ticket);
#endif
	if (t >= a.ctl->nchunks)
		return;
#ifdef DEC_STAMPS
	if (threadIdx.x == 0 && t < 65536)
		g_stamps[t * 8 + 0] = t_start;
#endif
	STAMP(1);
	const ChunkDesc *dp = a.chunks + t;
	const ChunkU d = load_chunk(dp);
	const uint32_t n = d.n;
	const uint32_t first = d.j * CHUNK;
	const bool last = first + CHUNK >= n;
	if (!d.cap_ok) { // not even the key bytes are there (out_n: k_svb_keyprefix)
		if (ZD && threadIdx.x < 64)
			(void) lookback(a.gran, t, d.j, 0, last);
		return;
	}
	const uint8_t *in = a.in + d.out_base + (S5 ? 4 : 0);
	const uint64_t in_len = uni64(a.in_len[d.read]) - (S5 ? 4 : 0);
	int16_t *out = a.sig + d.sig_off;
	const uint32_t klen = KEY2 ? (n + 3) / 4 : (n >> 3) + (((n & 7) + 7) >> 3);
	const uint64_t dlen = in_len - klen; // bytes in the data section (cap_ok: klen <= in_len)
	const uint8_t *data = in + klen;
	const uint32_t ws = first + w * WAVE_SAMPLES;
	const uint32_t e0 = uni(dp->ecnt[0]), e1 = uni(dp->ecnt[1]), e2 = uni(dp->ecnt[2]), e3 = uni(dp->ecnt[3]);
	const uint32_t ew = w == 0 ? e0 : w == 1 ? e1 : w == 2 ? e2 : e3;
	const uint64_t ebase = uni64(dp->ebefore) + (w > 0 ? e0 : 0u) + (w > 1 ? e1 : 0u) + (w > 2 ? e2 : 0u);
	// sub-tiles of this wave that exist, and those that are not plain; a sub-tile whose
	// 8-byte windows could reach past the end of the stream is not plain either
	const uint32_t nsub = ws >= n ? 0u : min((uint32_t) CK, (n - ws + SUB - 1) / SUB);
	const uint32_t live = nsub >= 32 ? ~0u : ((1u << nsub) - 1u);
	uint32_t kmask = uni(dp->kmask[w]) & live;
#pragma unroll
	for (int k = 0; k < CK; k++)
		if (((live >> k) & 1u) && ebase + ew + ws + k * SUB + SUB + 8 > dlen)
			kmask |= 1u << k;
	const uint32_t plain = live & ~kmask;
	STAMP(2);

	// ---- exceptions of the non-plain sub-tiles -> data offset of every sub-tile
	if (lane < CK)
		s_sub[w][lane] = 0;
	wave_lds_sync();
	for (uint32_t m = kmask; m; m &= m - 1) {
		const uint32_t k = (uint32_t) __builtin_ctz(m);
		const uint32_t kk = load_key<KEY2>(in, ws + k * SUB + lane * 8, n);
		const uint32_t inc = wave_incl_scan_dpp(key_extra_bytes<KEY2>(kk));
		if (lane == 63)
			s_sub[w][k] = inc;
	}
	wave_lds_sync();
	if (lane < CK) {
		const uint32_t v = s_sub[w][lane];
		const uint32_t inc = wave_incl_scan_dpp(v);
		s_epre[w][lane] = inc - v;
	}
	wave_lds_sync();

	// ---- phase 1: data loads of the plain sub-tiles (8 bytes per lane, any alignment)
	uint2 dat[CK];
#pragma unroll
	for (int k = 0; k < CK; k++) {
		const uint32_t i0 = ws + k * SUB + lane * 8;
		// (a zeroed register pair of its own for every sub-tile: left to itself the compiler kept the two halves of
		// sub-tile 0 in registers that are no pair, loaded into a third place, copied one half over - and waited for
		// that load before it issued the other fifteen: two memory round trips per chunk instead of one)
		unsigned long long d64;
		asm volatile("v_mov_b64 %0, 0" : "=v"(d64)); // (a register PAIR of its own: the 8-byte load lands in it)
		if ((plain >> k) & 1u) { // uniform
			const uint64_t eb = ebase + uni(s_epre[w][k]);
			if (i0 < n)
				d64 = ld8_stream(data + eb + i0);
		}
		const uint2 dd = make_uint2((uint32_t) d64, (uint32_t) (d64 >> 32));
		dat[k] = dd; // (k_low_decode_chunked's unconditional loads were tried here too: 82 instead of 51 VGPRs, 6 % slower)
	}

	// ---- phase 2: delta sums (16-bit wraparound) - per lane inside its sub-tile, per sub-tile
	if (ZD) {
#pragma unroll
		for (int k = 0; k < CK; k++) {
			if ((plain >> k) & 1u) { // uniform
				uint32_t v[4];
				expand8(dat[k], v);
				uint32_t acc = 0;
#pragma unroll
				for (int q = 0; q < 4; q++)
					acc = pk_add16(acc, unzz_pair(v[q]));
				const uint32_t tot16 = (acc + (acc >> 16)) & 0xFFFFu;
				const uint32_t inc = wave_incl_scan_dpp(tot16);
				s_xl[w][k][lane] = (uint16_t) (inc - tot16);
				if (lane == 63)
					s_sub[w][k] = inc;
			}
		}
		for (uint32_t m = kmask; m; m &= m - 1) {
			const uint32_t k = (uint32_t) __builtin_ctz(m);
			const uint32_t i0 = ws + k * SUB + lane * 8;
			uint32_t v[4], hib;
			(void) gather_slow<KEY2, S5>(in, data, dlen, i0, n, ebase + uni(s_epre[w][k]), v, hib);
			uint32_t acc = 0;
#pragma unroll
			for (int q = 0; q < 4; q++)
				acc = pk_add16(acc, unzz_pair_hib(v[q], hib, q));
			const uint32_t tot16 = (acc + (acc >> 16)) & 0xFFFFu;
			const uint32_t inc = wave_incl_scan_dpp(tot16);
			s_xl[w][k][lane] = (uint16_t) (inc - tot16);
			if (lane == 63)
				s_sub[w][k] = inc;
		}
		wave_lds_sync();
		// exclusive prefix over the sub-tiles of the wave
		uint32_t wtot = 0;
		{
			const uint32_t v = (lane < (int) nsub) ? s_sub[w][lane < CK ? lane : 0] : 0u;
			const uint32_t inc = wave_incl_scan_dpp(v);
			wave_lds_sync();
			if (lane < CK)
				s_sub[w][lane] = inc - v;
			wtot = (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
		}
		wave_lds_sync();
		if (lane == 0)
			s_wsum[w] = wtot & 0xFFFFu;
	}
	STAMP(3);

	// ---- sample value in front of this wave: within the chunk (LDS) and before it (look-back)
	uint32_t sb = 0;
	if (ZD) {
		__syncthreads();
		STAMP(4);
		const uint32_t u0 = uni(s_wsum[0]), u1 = uni(s_wsum[1]), u2 = uni(s_wsum[2]), u3 = uni(s_wsum[3]);
		if (w == 0) {
			const uint32_t sv = (uint32_t) lookback(a.gran, t, d.j, (uint64_t) ((u0 + u1 + u2 + u3) & 0xFFFFu), last);
			if (lane == 0)
				s_sbase = sv;
		}
		__syncthreads();
		sb = uni(s_sbase) + (w > 0 ? u0 : 0u) + (w > 1 ? u1 : 0u) + (w > 2 ? u2 : 0u);
	}
	STAMP(5);

	// Every load has completed by now (the barrier above drains vmcnt).  Re-define the payload
	// registers through an empty asm so that the compiler stops associating them with memory
	// operations: otherwise it puts an s_waitcnt vmcnt(0) in front of every sub-tile of the
	// store loop and each 16-byte store is drained before the next one is issued.
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
	for (int k = 0; k < CK; k++)
		asm volatile("" : "+v"(dat[k].x), "+v"(dat[k].y));

	// ---- phase 3: expand again, prefix inside the lane, add the bases, store
#pragma unroll
	for (int k = 0; k < CK; k++) {
		const uint32_t i0 = ws + k * SUB + lane * 8;
		if ((plain >> k) & 1u) { // uniform; plain sub-tiles hold only complete lanes
			uint32_t v[4];
			expand8(dat[k], v);
			if (ZD) {
#pragma unroll
				for (int q = 0; q < 4; q++)
					v[q] = unzz_pair(v[q]);
				(void) lane_prefix8(v);
				const uint32_t b16 = (sb + uni(s_sub[w][k]) + s_xl[w][k][lane]) & 0xFFFFu;
				const uint32_t b2 = b16 | (b16 << 16);
#pragma unroll
				for (int q = 0; q < 4; q++)
					v[q] = pk_add16(v[q], b2);
			}
			if (i0 < n)
				st16_stream(out + i0, make_uint4(v[0], v[1], v[2], v[3]));
		}
	}
	for (uint32_t m = kmask; m; m &= m - 1) {
		const uint32_t k = (uint32_t) __builtin_ctz(m);
		const uint32_t i0 = ws + k * SUB + lane * 8;
		uint32_t v[4], hib;
		(void) gather_slow<KEY2, S5>(in, data, dlen, i0, n, ebase + uni(s_epre[w][k]), v, hib);
		if (ZD) {
#pragma unroll
			for (int q = 0; q < 4; q++)
				v[q] = unzz_pair_hib(v[q], hib, q);
			(void) lane_prefix8(v);
			const uint32_t b16 = (sb + uni(s_sub[w][k]) + s_xl[w][k][lane]) & 0xFFFFu;
			const uint32_t b2 = b16 | (b16 << 16);
#pragma unroll
			for (int q = 0; q < 4; q++)
				v[q] = pk_add16(v[q], b2);
		}
		if (i0 + 8 <= n) {
			st16_stream(out + i0, make_uint4(v[0], v[1], v[2], v[3]));
		} else if (i0 < n) {
#pragma unroll
			for (uint32_t q = 0; q < 8; q++)
				if (q < n - i0)
					out[i0 + q] = (int16_t) (v[q >> 1] >> (16 * (q & 1)));
		}
	}
	STAMP(6);
}

// ------------------------------------------------------------------ exception split, chunked
//
// vb1e2 family / ex-zd (press.c:2679-3580, ex_zd.c:9-172): the stream is
//   header || u32 nex || exception section || one-byte values of the non-exceptions.
// The section's size depends on ALL exceptions of the read, so the input is read twice:
//   k_ex_scan_chunked  (ticket order + look-back on the exception count): exception list
//                      (position, value) at its final rank, per-chunk counts, OR of the samples
//   k_ex_section       (press_sections.hip): header + section, one lane per read
//   k_low_encode_chunked: the one-byte stream; every chunk knows its offset from the counts,
//                      so no ordering is needed: one workgroup per chunk, 8-byte stores for
//                      sub-tiles without exceptions exactly as in k_svb_encode_chunked.

// zig-zag deltas of one wave quarter, as in k_svb_encode_chunked phase 1, with the ex-zd
// shift q applied to the samples first (ex_zd.c:383).  hi[k] != 0 marks lanes with an
// exception in sub-tile k; sample 0 of the read is never one (it is stored raw).
__device__ __forceinline__ void quarter_zd(const int16_t *in, uint32_t n, uint32_t ws, int lane, int q,
					   uint4 (&z)[CK], uint32_t &ored)
{
#pragma unroll
	for (int k = 0; k < CK; k++) {
		const uint32_t i0 = ws + k * SUB + lane * 8;
		z[k] = make_uint4(0, 0, 0, 0);
		if (i0 < n)
			z[k] = ld16_stream(in + i0);
	}
	uint32_t carry = 0;
	if (ws > 0 && ws < n)
		carry = (uint32_t) (uint16_t) in[ws - 1] << 16;
	uint32_t o = 0;
	const s16x2 qq = { (short) q, (short) q };
#pragma unroll
	for (int k = 0; k < CK; k++) {
		const uint32_t i0 = ws + k * SUB + lane * 8;
		uint32_t r[4] = { z[k].x, z[k].y, z[k].z, z[k].w };
		// samples at or beyond n: zero (they neither count nor contribute to the OR)
		if (i0 + 8 > n) {
			const uint32_t nv = i0 < n ? n - i0 : 0;
#pragma unroll
			for (int h = 0; h < 4; h++) {
				if (nv <= (uint32_t) (2 * h))
					r[h] = 0;
				else if (nv == (uint32_t) (2 * h + 1))
					r[h] &= 0xFFFFu;
			}
		}
		o |= r[0] | r[1] | r[2] | r[3];
		const uint32_t raw_w = r[3];
		if (q) {
#pragma unroll
			for (int h = 0; h < 4; h++)
				r[h] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, r[h]) >> qq);
		}
		uint32_t c = carry;
		if (q)
			c = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, c) >> qq);
		const uint32_t pw = prev_lane(r[3], c);
		uint32_t zz[4];
		zz[0] = zd_pair(r[0], pw);
		zz[1] = zd_pair(r[1], r[0]);
		zz[2] = zd_pair(r[2], r[1]);
		zz[3] = zd_pair(r[3], r[2]);
		carry = (uint32_t) __builtin_amdgcn_readlane((int) raw_w, 63);
		if (i0 + 8 > n) { // deltas of samples beyond n are garbage: zero them again
			const uint32_t nv = i0 < n ? n - i0 : 0;
#pragma unroll
			for (int h = 0; h < 4; h++) {
				if (nv <= (uint32_t) (2 * h))
					zz[h] = 0;
				else if (nv == (uint32_t) (2 * h + 1))
					zz[h] &= 0xFFFFu;
			}
		}
		z[k] = make_uint4(zz[0], zz[1], zz[2], zz[3]);
	}
	ored = (o | (o >> 16)) & 0xFFFFu;
}

// 8-bit mask of the samples of a lane that are exceptions (value > 255)
__device__ __forceinline__ uint32_t exc_mask(const uint4 &z, uint32_t i0)
{
	const uint32_t zz[4] = { z.x, z.y, z.z, z.w };
	uint32_t m = 0;
#pragma unroll
	for (int h = 0; h < 4; h++) {
		m |= ((zz[h] & 0x0000FF00u) ? 1u : 0u) << (2 * h);
		m |= ((zz[h] & 0xFF000000u) ? 1u : 0u) << (2 * h + 1);
	}
	if (i0 == 0)
		m &= ~1u; // zd[0] is stored raw in the header
	return m;
}

// raw samples of a lane's 8 samples of a sub-tile (zeros at or beyond n)
__device__ __forceinline__ uint4 sub_load(const int16_t *in, uint32_t n, uint32_t i0)
{
	uint4 r = make_uint4(0, 0, 0, 0);
	if (i0 < n)
		r = ld16_stream(in + i0);
	return r;
}

// their zig-zag deltas (trans.c:75,215), after the ex-zd shift q (ex_zd.c:383).  carry = the
// (raw) sample in front of the sub-tile in bits 16..31; returns the sub-tile's last sample the
// same way for the next one.  ored |= the raw samples below n (qts, ex_zd.c:358).
__device__ __forceinline__ uint4 sub_zd(uint4 raw, uint32_t n, uint32_t i0, uint32_t &carry, int q = 0,
					uint32_t *ored = nullptr)
{
	uint32_t r[4] = { raw.x, raw.y, raw.z, raw.w };
	const uint32_t nv = i0 + 8 <= n ? 8u : (i0 < n ? n - i0 : 0u);
	if (nv < 8) {
#pragma unroll
		for (int h = 0; h < 4; h++) {
			if (nv <= (uint32_t) (2 * h))
				r[h] = 0;
			else if (nv == (uint32_t) (2 * h + 1))
				r[h] &= 0xFFFFu;
		}
	}
	if (ored)
		*ored |= r[0] | r[1] | r[2] | r[3];
	const uint32_t raw_w = r[3];
	uint32_t c = carry;
	if (q) {
		const s16x2 qq = { (short) q, (short) q };
#pragma unroll
		for (int h = 0; h < 4; h++)
			r[h] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, r[h]) >> qq);
		c = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, c) >> qq);
	}
	const uint32_t pw = prev_lane(r[3], c);
	uint32_t zz[4];
	zz[0] = zd_pair(r[0], pw);
	zz[1] = zd_pair(r[1], r[0]);
	zz[2] = zd_pair(r[2], r[1]);
	zz[3] = zd_pair(r[3], r[2]);
	carry = (uint32_t) __builtin_amdgcn_readlane((int) raw_w, 63);
	if (nv < 8) { // deltas of samples beyond n are garbage
#pragma unroll
		for (int h = 0; h < 4; h++) {
			if (nv <= (uint32_t) (2 * h))
				zz[h] = 0;
			else if (nv == (uint32_t) (2 * h + 1))
				zz[h] &= 0xFFFFu;
		}
	}
	return make_uint4(zz[0], zz[1], zz[2], zz[3]);
}

// which of a lane's 8 samples starting at i0 are one-byte values: inside [1, n), no exception
__device__ __forceinline__ uint32_t low_mask(const uint4 &z, uint32_t i0, uint32_t n)
{
	uint32_t lowm = i0 + 8 <= n ? 0xFFu : (i0 < n ? (1u << (n - i0)) - 1u : 0u);
	if (i0 == 0)
		lowm &= ~1u;
	if ((z.x | z.y | z.z | z.w) & 0xFF00FF00u)
		lowm &= ~exc_mask(z, i0);
	return lowm;
}

// ex-zd: does any read of the batch need the second scan (all samples divisible by 2^q, q > 0)?
__global__ __launch_bounds__(256) void k_ex_redo_flag(BatchArgs a)
{
	const uint32_t r = blockIdx.x * 256 + threadIdx.x;
	if (r < a.nreads && a.nsamp[r] && !(a.meta[r].ored & 1u))
		atomicOr(&a.ctl->pad0[0], 1u);
}

// Pass A.  Nothing chains the chunks here: every workgroup streams its chunk once (four sub-tiles
// in flight per wave) and leaves, per wave quarter, the number of exceptions and the sub-tiles that
// hold one in the chunk descriptor; k_ex_prefix turns the counts into ranks and k_ex_list
// writes the (few) exceptions at their final rank from the flagged sub-tiles alone.
// HUFF: also the number of Huffman code bits of every wave quarter (ChunkBits), which lets
// k_huff_encode_chunked place its bits without any chain either.
// Waves per SIMD the register allocator is asked to leave room for (measured: pass A is slower at 8 - a spill -,
// the Huffman encoder 16 % faster - 38 instead of 76 registers; the svb and one-byte kernels are slower with any
// hint: they stream, and the registers hold their loads in flight).
#ifndef SCAN_WAVES
#define SCAN_WAVES 4
#endif
#ifndef HENC_WAVES
#define HENC_WAVES 8
#endif
template <bool REDO, bool HUFF = false>
__global__ __launch_bounds__(CWG, SCAN_WAVES) void k_ex_scan_chunked(BatchArgs a)
{
	// code length of every one-byte value, 1 << 16 for a value without a code; [256] = 0: what a sample that
	// is no one-byte value (exception, sample 0, outside the read) looks up
	__shared__ uint32_t s_len[HUFF ? 257 : 1];
	if (REDO && uni(a.ctl->pad0[0]) == 0)
		return; // no read of this batch has q > 0 (the common case)
	const uint32_t t = blockIdx.x;
	if (t >= uni(a.ctl->nchunks))
		return;
	const int lane = threadIdx.x & 63;
	const int w = (int) uni(threadIdx.x >> 6);
	if (HUFF) {
		const uint32_t l = a.huff->enc[threadIdx.x] >> 24;
		s_len[threadIdx.x] = l ? l : 0x10000u;
		if (threadIdx.x == 0)
			s_len[256] = 0;
		__syncthreads();
	}
	ChunkDesc *dp = a.chunks + t;
	const ChunkU d = load_chunk(dp);
	const uint32_t n = d.n;
	const uint32_t first = d.j * CHUNK;
	ReadMeta *m = a.meta + d.read;
	int q = 0;
	if (REDO) {
		const uint32_t ored = uni(m->ored);
		// ex_zd.c:358-381: largest q <= 5 with every sample divisible by 2^q
		while (q < 5 && !((ored >> q) & 1u))
			q++;
		if (q == 0)
			return; // nothing to redo for this read
	}
	const int16_t *in = a.sig + d.sig_off;
	const uint32_t ws = first + w * WAVE_SAMPLES;
	if (ws >= n) { // the descriptor's counts are zero already
		if (HUFF && lane == 0)
			a.cbits[t].q[w] = 0;
		return;
	}

	uint32_t kmask = 0, etot = 0, ored32 = 0, zd0 = 0, lbits = 0, nocode = 0;
	uint4 raw[4];
#pragma unroll
	for (int k = 0; k < 4; k++)
		raw[k] = sub_load(in, n, ws + k * SUB + lane * 8);
	// the sample in front of the sub-tile: asked for BEHIND the first sub-tiles (in front of them, its shift made the
	// wave wait for this one load before it issued any of the others: one more memory round trip per quarter)
	uint32_t carry = 0, c16 = 0;
	if (ws > 0)
		c16 = (uint32_t) (uint16_t) in[ws - 1]; // (shifted into place behind the next group's loads)
#pragma unroll 1
	for (int kk = 0; kk < CK; kk += 4) {
		uint4 nxt[4];
#pragma unroll
		for (int k = 0; k < 4; k++)
			nxt[k] = kk + 4 + k < CK ? sub_load(in, n, ws + (kk + 4 + k) * SUB + lane * 8) : make_uint4(0, 0, 0, 0);
		if (kk == 0) {
			asm volatile("" : "+v"(c16)); // (keeps the shift - and the wait for the load - here)
			carry = c16 << 16;
		}
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const uint32_t i0 = ws + (kk + k) * SUB + lane * 8;
			const uint4 z = sub_zd(raw[k], n, i0, carry, q, &ored32);
			if (kk + k == 0)
				zd0 = z.x & 0xFFFFu; // lane 0 of the read's first quarter: zd[0]
			uint32_t hi = (z.x | z.y | z.z | z.w) & 0xFF00FF00u;
			if (i0 == 0)
				hi = ((z.x & 0xFF000000u) | ((z.y | z.z | z.w) & 0xFF00FF00u));
			const bool anyex = __ballot(hi != 0) != 0;
			if (anyex) {
				kmask |= 1u << (kk + k);
				const uint32_t inc = wave_incl_scan_dpp(__popc(exc_mask(z, i0)));
				etot += (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
			}
			if (HUFF) { // code lengths of the one-byte values: samples [1, n) that are no exceptions
				const uint32_t zz[4] = { z.x, z.y, z.z, z.w };
				const uint32_t sub0 = ws + (kk + k) * SUB;
				if (!anyex && sub0 != 0 && sub0 + SUB <= n) {
					// a sub-tile of one-byte values only (nearly all are): no mask, no select per sample
#pragma unroll
					for (int h = 0; h < 8; h++)
						lbits += s_len[(zz[h >> 1] >> (16 * (h & 1))) & 0xFFu];
				} else {
					const uint32_t lowm = low_mask(z, i0, n);
#pragma unroll
					for (int h = 0; h < 8; h++) // (a lane sums 128 values of at most 24 bits: below 1 << 16)
						lbits += s_len[((lowm >> h) & 1u) ? ((zz[h >> 1] >> (16 * (h & 1))) & 0xFFu) : 256u];
				}
			}
		}
#pragma unroll
		for (int k = 0; k < 4; k++)
			raw[k] = nxt[k];
	}
	if (HUFF) {
		if (lbits >> 16)
			nocode = 0x80000000u; // a value the table has no code for: the read fails (k_ex_section)
		lbits &= 0xFFFFu;
		const uint32_t inc = wave_incl_scan_dpp(lbits);
		if (lane == 63)
			a.cbits[t].q[w] = inc;
	}
	if (!REDO) { // OR of the raw samples (qts), one atomic per wave; bit 31: a value without a Huffman code
		uint32_t o = ((ored32 | (ored32 >> 16)) & 0xFFFFu) | (HUFF ? nocode : 0u);
#pragma unroll
		for (int dd = 32; dd >= 1; dd >>= 1)
			o |= (uint32_t) __shfl_xor((int) o, dd, 64);
		if (lane == 0)
			atomicOr(&m->ored, o);
	}
	if (lane == 0) {
		dp->ecnt[w] = etot;
		dp->kmask[w] = (uint16_t) kmask;
		if (ws == 0)
			m->zd0 = zd0;
	}
}

// exceptions in front of every chunk (one thread per read); the read's total and its ex-zd shift
__global__ __launch_bounds__(256) void k_ex_prefix(BatchArgs a, int exzd)
{
	// one wave per read (a thread per read walked the 175 chunks of the longest read one after the other)
	const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
	const uint32_t lane = threadIdx.x & 63;
	if (r >= a.nreads)
		return;
	const uint32_t n = a.nsamp[r];
	if (n == 0)
		return;
	ReadMeta *m = a.meta + r;
	ChunkDesc *dp = a.chunks + a.first_chunk[r];
	const uint32_t nch = (n + CHUNK - 1) / CHUNK;
	// exclusive prefix of a 64-bit count over the chunks, 64 at a time; returns the total
	auto scan = [&](auto count_of, auto store) -> uint64_t {
		uint64_t carry = 0;
		for (uint32_t j0 = 0; j0 < nch; j0 += 64) {
			const uint32_t j = j0 + lane;
			const uint64_t c = j < nch ? count_of(j) : 0ull;
			// (two 32-bit scans, of the counts' low 20 bits and of the rest: neither can overflow over 64 chunks)
			const uint32_t lo = wave_incl_scan_dpp((uint32_t) (c & 0xFFFFFu)), hi = wave_incl_scan_dpp((uint32_t) (c >> 20));
			const uint64_t inc = (uint64_t) lo + ((uint64_t) hi << 20);
			if (j < nch)
				store(j, carry + inc - c);
			const uint64_t tot = (uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) lo, 63) +
					     ((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) hi, 63) << 20);
			carry += tot;
		}
		return carry;
	};
	const uint64_t e = scan([&](uint32_t j) { return (uint64_t) dp[j].ecnt[0] + dp[j].ecnt[1] + dp[j].ecnt[2] + dp[j].ecnt[3]; },
				[&](uint32_t j, uint64_t v) { dp[j].ebefore = v; });
	if (a.cbits) { // Huffman code bits in front of every chunk
		ChunkBits *cb = a.cbits + a.first_chunk[r];
		(void) scan([&](uint32_t j) { return (uint64_t) cb[j].q[0] + cb[j].q[1] + cb[j].q[2] + cb[j].q[3]; },
			    [&](uint32_t j, uint64_t v) { cb[j].before = v; });
	}
	if (lane == 0) {
		m->nex = (uint32_t) e;
		uint32_t q = 0;
		if (exzd) { // ex_zd.c:358-381
			const uint32_t ored = m->ored;
			while (q < 5 && !((ored >> q) & 1u))
				q++;
		}
		m->q = q;
	}
}

// the exception list (position, value) at its final rank: only the flagged sub-tiles are read again
__global__ __launch_bounds__(CWG) void k_ex_list(BatchArgs a)
{
	// one WAVE per chunk (its four quarters one after the other: nearly all of them hold no exception) - a
	// workgroup per chunk meant 150 000 waves that read a descriptor and left
	const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (t >= uni(a.ctl->nchunks))
		return;
	const int lane = threadIdx.x & 63;
	const ChunkDesc *dp = a.chunks + t;
	const uint32_t km[4] = { uni(dp->kmask[0]), uni(dp->kmask[1]), uni(dp->kmask[2]), uni(dp->kmask[3]) };
	if (!(km[0] | km[1] | km[2] | km[3]))
		return;
	const ChunkU d = load_chunk(dp);
	const uint32_t n = d.n;
	const int q = (int) uni(a.meta[d.read].q);
	const int16_t *in = a.sig + d.sig_off;
	uint32_t *lpos = a.ex_pos + d.sig_off;
	uint32_t *lval = a.ex_val + d.sig_off;
	const uint32_t cnt[4] = { uni(dp->ecnt[0]), uni(dp->ecnt[1]), uni(dp->ecnt[2]), uni(dp->ecnt[3]) };
	uint32_t rank = (uint32_t) uni64(dp->ebefore);
#pragma unroll
	for (int w = 0; w < 4; w++) {
		const uint32_t ws = d.j * CHUNK + w * WAVE_SAMPLES;
		uint32_t rk = rank;
		for (uint32_t mm = km[w]; mm; mm &= mm - 1) {
			const uint32_t k = (uint32_t) __builtin_ctz(mm);
			const uint32_t sub0 = ws + k * SUB;
			const uint32_t i0 = sub0 + lane * 8;
			uint32_t carry = sub0 ? (uint32_t) (uint16_t) in[sub0 - 1] << 16 : 0u;
			const uint4 z = sub_zd(sub_load(in, n, i0), n, i0, carry, q);
			const uint32_t em = exc_mask(z, i0);
			const uint32_t c = __popc(em);
			const uint32_t inc = wave_incl_scan_dpp(c);
			uint32_t p = rk + inc - c;
			const uint32_t zz[4] = { z.x, z.y, z.z, z.w };
			if (em) {
#pragma unroll
				for (int h = 0; h < 8; h++) {
					if ((em >> h) & 1u) {
						lpos[p] = i0 + h - 1;
						lval[p] = (zz[h >> 1] >> (16 * (h & 1))) & 0xFFFFu;
						p++;
					}
				}
			}
			rk += (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
		}
		rank += cnt[w];
	}
}

// Pass B, plain one-byte stream (press.c:2717-2725): one workgroup per chunk, any order.
// HIST (zstd_hasgam_vbsse21_zdq, press_zstd.hip): the bytes are counted per read while a lane holds them, as in
// k_svb_encode_chunked<.., HIST> (BatchArgs::zhist).
template <bool HIST = false>
__global__ __launch_bounds__(CWG) void k_low_encode_chunked(BatchArgs a)
{
	__shared__ uint32_t s_hist[HIST ? 16 : 1][256];
	const uint32_t t = blockIdx.x;
	if (t >= a.ctl->nchunks)
		return;
	const ChunkDesc *dp = a.chunks + t;
	const ChunkU d = load_chunk(dp);
	const ReadMeta *m = a.meta + d.read;
	if (uni(m->status))
		return; // out_len = FAILED was written by k_ex_section
	const uint32_t n = d.n;
	const int lane = threadIdx.x & 63;
	const int w = (int) uni(threadIdx.x >> 6);
	const uint32_t ws = d.j * CHUNK + w * WAVE_SAMPLES;
	uint32_t *myhist = s_hist[HIST ? 4 * w + (lane & 3) : 0];
	if (HIST) {
		for (int i = 0; i < 16; i++)
			s_hist[i][threadIdx.x] = 0;
		__syncthreads();
	}
	if (ws < n) {
	const int q = (int) uni(m->q);
	const int16_t *in = a.sig + d.sig_off;
	// one-byte value of sample i (i >= 1) goes to stream[(i - 1) - #exceptions before i]
	uint8_t *stream = a.low_tmp ? a.low_tmp + d.sig_off : a.out + d.out_base + uni(m->hdr) + uni(m->seclen);
	const uint32_t e0 = uni(dp->ecnt[0]), e1 = uni(dp->ecnt[1]), e2 = uni(dp->ecnt[2]);
	uint64_t eb = uni64(dp->ebefore) + (w > 0 ? e0 : 0u) + (w > 1 ? e1 : 0u) + (w > 2 ? e2 : 0u);

	uint4 z[CK];
	uint32_t ored;
	quarter_zd(in, n, ws, lane, q, z, ored);
#pragma unroll
	for (int k = 0; k < CK; k++) {
		const uint32_t i0 = ws + k * SUB + lane * 8;
		const uint32_t sub0 = ws + k * SUB;
		if (sub0 >= n)
			continue;
		uint32_t hi = (z[k].x | z[k].y | z[k].z | z[k].w) & 0xFF00FF00u;
		const bool special = (i0 == 0) || (i0 < n && i0 + 8 > n); // sample 0 / ragged tail
		if (!__ballot(hi != 0 || special)) {
			// no exception in the sub-tile: 8 low bytes per lane, one (unaligned) store
			if (i0 < n) {
				uint2 v;
				v.x = __builtin_amdgcn_perm(z[k].y, z[k].x, 0x06040200);
				v.y = __builtin_amdgcn_perm(z[k].w, z[k].z, 0x06040200);
				__builtin_memcpy(stream + (i0 - 1 - eb), &v, 8);
				if (HIST) {
#pragma unroll
					for (int e = 0; e < 4; e++) {
						atomicAdd(&myhist[(v.x >> (8 * e)) & 0xFFu], 1u);
						atomicAdd(&myhist[(v.y >> (8 * e)) & 0xFFu], 1u);
					}
				}
			}
			continue;
		}
		const uint32_t em = exc_mask(z[k], i0);
		const uint32_t c = __popc(em);
		const uint32_t inc = wave_incl_scan_dpp(c);
		const uint32_t nv = i0 < n ? min(8u, n - i0) : 0u;
		const uint32_t zz[4] = { z[k].x, z[k].y, z[k].z, z[k].w };
		// first stream index of this lane: samples before i0 minus sample 0 minus exceptions
		uint64_t p = (uint64_t) (i0 ? i0 - 1 : 0) - (eb + inc - c);
#pragma unroll
		for (int h = 0; h < 8; h++) {
			if ((uint32_t) h < nv && !((em >> h) & 1u) && !(i0 == 0 && h == 0)) {
				const uint32_t b = (zz[h >> 1] >> (16 * (h & 1))) & 0xFFu;
				stream[p++] = (uint8_t) b;
				if (HIST)
					atomicAdd(&myhist[b], 1u);
			}
		}
		eb += (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
	}
	} // (a wave whose quarter lies behind the read's end has nothing to write)
	if (HIST) {
		__syncthreads();
		uint32_t c = 0;
		for (int i = 0; i < 16; i++)
			c += s_hist[i][threadIdx.x];
		if (c)
			atomicAdd(&a.zhist[(uint64_t) d.read * 256 + threadIdx.x], c);
	}
}

// Pass B, static-Huffman stream (huffman.c:1184 + do_memory_encode :848: the codes packed
// from bit 0 of each byte upwards - a little-endian bit stream).  One workgroup per chunk in any
// order, its four waves independent of each other: pass A has counted the code bits of every
// wave quarter (k_ex_scan_chunked<.., true>) and k_ex_prefix has summed them, so a wave knows the
// bit position Bq of its quarter from the start; it stages the codes of one 512-sample sub-tile
// at a time in a private LDS buffer (LDS atomic OR of code << bitpos) and copies the completed
// dwords out.  No ticket, no look-back, one barrier (the table load).
// Who writes a byte shared by two quarters: the quarter that holds the byte's LAST bit; it
// recomputes the few bits its predecessors put into that byte from the samples in front of
// it.  The last, partial byte of a read and out_len are written by wave 0 of its last chunk.
constexpr int HSTG = 400; // dwords per wave: carried bits (< 32) + 512 codes of at most 24 bits + spill

__device__ __forceinline__ uint32_t zz16_dev(int32_t d)
{
	const int32_t s = (int32_t) (int16_t) d;
	return (uint32_t) ((s << 1) ^ (s >> 15)) & 0xFFFFu;
}

// the last nb (1..7) bits of the code stream of samples [1, end) of a read: all 64 lanes call
// it, every lane gets the value.  enc: the code table in LDS.
__device__ __forceinline__ uint32_t huff_tail_bits(const int16_t *in, uint32_t end, uint32_t nb, const uint32_t *enc,
						   int lane)
{
	uint32_t acc = 0, got = 0;
	uint32_t pos = end; // samples [1, pos) are still to be looked at
	while (got < nb && pos > 1) {
		const int64_t i = (int64_t) pos - 64 + lane;
		const bool valid = i >= 1;
		uint32_t zv = 0xFFFFu;
		if (valid)
			zv = zz16_dev((int32_t) in[i] - (int32_t) in[i - 1]);
		unsigned long long mask = __ballot(valid && zv <= 255u);
		while (mask && got < nb) {
			const int l = 63 - __builtin_clzll(mask);
			const uint32_t e = enc[(uint32_t) __builtin_amdgcn_readlane((int) zv, l)];
			const uint32_t len = e >> 24;
			acc = (acc << len) | (e & 0xFFFFFFu); // an earlier value's code sits below the later ones
			got += len;
			mask &= ~(1ull << l);
		}
		pos = pos > 64 ? pos - 64 : 0;
	}
	return got >= nb ? (acc >> (got - nb)) & ((1u << nb) - 1u) : acc;
}


__global__ __launch_bounds__(CWG, HENC_WAVES) void k_huff_encode_chunked(BatchArgs a)
{
	__shared__ uint32_t enc[257]; // [256] = 0: what a sample without a code (exception, outside the read) looks up
	__shared__ uint32_t stg_all[4][HSTG];

	enc[threadIdx.x] = a.huff->enc[threadIdx.x];
	if (threadIdx.x == 0)
		enc[256] = 0;
	for (uint32_t i = threadIdx.x; i < 4u * HSTG; i += CWG)
		(&stg_all[0][0])[i] = 0;
	__syncthreads(); // the only barrier
	const uint32_t t = blockIdx.x;
	if (t >= uni(a.ctl->nchunks))
		return;
	const int lane = threadIdx.x & 63;
	const int w = (int) uni(threadIdx.x >> 6);
	uint32_t *stg = stg_all[w];
	const ChunkDesc *dp = a.chunks + t;
	const ChunkU d = load_chunk(dp);
	const ReadMeta *m = a.meta + d.read;
	if (uni(m->status))
		return; // out_len = FAILED was written by k_ex_section
	const uint32_t n = d.n;
	const uint32_t first = d.j * CHUNK;
	const bool last = first + CHUNK >= n;
	const uint32_t ws = first + w * WAVE_SAMPLES;
	const int16_t *in = a.sig + d.sig_off;
	const uint32_t head = uni(m->hdr) + uni(m->seclen) + 4; // bytes in front of the payload
	uint8_t *payload = a.out + d.out_base + head;
	const uint64_t cap = uni64(a.out_off[d.read + 1]) - d.out_base;
	// code bits of the quarters and in front of the chunk: k_ex_scan_chunked<.., true> + k_ex_prefix
	const ChunkBits *cb = a.cbits + t;
	const uint32_t q0 = uni(cb->q[0]), q1 = uni(cb->q[1]), q2 = uni(cb->q[2]), q3 = uni(cb->q[3]);
	const uint64_t before = uni64(cb->before);
	if (w == 0 && last) { // the read ends here: its length and its last, partial byte (zero padded, huffman.c:878)
		const uint64_t bits = before + q0 + q1 + q2 + q3;
		const uint64_t total = (uint64_t) head + (bits + 7) / 8;
		const uint32_t nbt = (uint32_t) bits & 7u;
		if (total <= cap && nbt) {
			const uint32_t tb = huff_tail_bits(in, n, nbt, enc, lane);
			if (lane == 0)
				payload[bits >> 3] = (uint8_t) tb;
		}
		if (lane == 0)
			a.out_len[d.read] = total <= cap ? total : CFAIL64;
	}
	const uint32_t mybits = w == 0 ? q0 : w == 1 ? q1 : w == 2 ? q2 : q3;
	const uint64_t Bq = before + (w > 0 ? q0 : 0u) + (w > 1 ? q1 : 0u) + (w > 2 ? q2 : 0u);
	// a quarter that would run past the slot writes nothing - the read fails
	if (ws >= n || !mybits || (uint64_t) head + (Bq + mybits + 7) / 8 > cap)
		return;
	// the nb bits in front of Bq that share its byte
	const uint32_t nb = (uint32_t) Bq & 7u;
	if (nb) {
		const uint32_t tb = huff_tail_bits(in, ws, nb, enc, lane);
		if (lane == 0)
			stg[0] = tb;
	}
	uint8_t *g = payload + (Bq >> 3); // byte of staging bit 0
	uint32_t fill = nb;               // bits waiting at the front of the staging buffer
	wave_lds_sync();
	uint4 raw[2];
	raw[0] = sub_load(in, n, ws + lane * 8);
	raw[1] = sub_load(in, n, ws + SUB + lane * 8);
	// the sample in front of the sub-tile (behind the loads above: see k_ex_scan_chunked)
	uint32_t carry = ws > 0 ? (uint32_t) (uint16_t) in[ws - 1] << 16 : 0u;
#pragma unroll 1
	for (int k = 0; k < CK; k++) {
		const uint32_t sub0 = ws + k * SUB;
		if (sub0 >= n)
			break;
		const uint32_t i0 = sub0 + lane * 8;
		const uint4 nxt = k + 2 < CK ? sub_load(in, n, i0 + 2 * SUB) : make_uint4(0, 0, 0, 0);
		const uint4 z = sub_zd(raw[0], n, i0, carry);
		raw[0] = raw[1];
		raw[1] = nxt;
		const uint32_t zz[4] = { z.x, z.y, z.z, z.w };
		uint32_t e[8];
		uint32_t lb = 0;
		if (sub0 != 0 && sub0 + SUB <= n && !__ballot(((z.x | z.y | z.z | z.w) & 0xFF00FF00u) != 0)) {
			// a sub-tile of one-byte values only (nearly all are): no mask, no select per sample
#pragma unroll
			for (int h = 0; h < 8; h++) {
				e[h] = enc[(zz[h >> 1] >> (16 * (h & 1))) & 0xFFu];
				lb += e[h] >> 24;
			}
		} else {
			const uint32_t lowm = low_mask(z, i0, n);
#pragma unroll
			for (int h = 0; h < 8; h++) {
				e[h] = enc[((lowm >> h) & 1u) ? ((zz[h >> 1] >> (16 * (h & 1))) & 0xFFu) : 256u];
				lb += e[h] >> 24;
			}
		}
		const uint32_t inc = wave_incl_scan_dpp(lb);
		const uint32_t tot = (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
		uint32_t pos = fill + inc - lb;
		if (lb <= 64) {
			// the lane's codes side by side in two registers (8 codes of the NA12878 table: 43 bits on
			// average, more than 64 in one lane of 700), then at most three LDS atomics
			uint64_t acc = 0;
			uint32_t o = 0;
#pragma unroll
			for (int h = 0; h < 8; h++) {
				acc |= (uint64_t) (e[h] & 0xFFFFFFu) << o;
				o += e[h] >> 24;
			}
			const uint32_t sh = pos & 31u;
			const uint64_t lo2 = acc << sh;
			const uint32_t w2 = sh ? (uint32_t) (acc >> 32) >> (32u - sh) : 0u;
			if ((uint32_t) lo2)
				atomicOr(&stg[pos >> 5], (uint32_t) lo2);
			if ((uint32_t) (lo2 >> 32))
				atomicOr(&stg[(pos >> 5) + 1], (uint32_t) (lo2 >> 32));
			if (w2)
				atomicOr(&stg[(pos >> 5) + 2], w2);
		} else {
#pragma unroll
			for (int h = 0; h < 8; h++) {
				if (e[h]) {
					const uint64_t wv = (uint64_t) (e[h] & 0xFFFFFFu) << (pos & 31u);
					if ((uint32_t) wv)
						atomicOr(&stg[pos >> 5], (uint32_t) wv);
					if ((uint32_t) (wv >> 32))
						atomicOr(&stg[(pos >> 5) + 1], (uint32_t) (wv >> 32));
					pos += e[h] >> 24;
				}
			}
		}
		wave_lds_sync();
		// completed dwords out (unaligned 4-byte stores), the partial one to the front
		const uint32_t total = fill + tot;
		const uint32_t nfull = total >> 5;
		for (uint32_t c = (uint32_t) lane; c < nfull; c += 64) {
			const uint32_t v = stg[c];
			stg[c] = 0;
			__builtin_memcpy(g + 4ull * c, &v, 4);
		}
		if (nfull && lane == 0) {
			const uint32_t tl = stg[nfull];
			stg[nfull] = 0;
			stg[0] = tl;
		}
		g += 4ull * nfull;
		fill = total & 31u;
		wave_lds_sync();
	}
	// whole bytes that are left; a partial last byte belongs to whoever holds its last bit
	if (lane == 0) {
		const uint32_t v = stg[0];
		for (uint32_t b = 0; b < (fill >> 3); b++)
			g[b] = (uint8_t) (v >> (8 * b));
	}
}

// ------------------------------------------------------------------ exception split decode, chunked
//
// Second half of vbe21_depress and siblings (press.c:2757-2771) fused with unzigdelta_u16_16
// (trans.c:260), after k_ex_parse has produced the sorted exception list (and, for the
// Huffman variants, k_huff_decode_tiles the one-byte stream).  Sample i >= 1 is exception e
// if pos[e] == i-1, otherwise one-byte value number (i-1) - #exceptions before it; so a
// chunk finds its place in the stream by binary search - only the running sample value
// needs the look-back chain.

// chunk table from the parsed streams: n = 1 + nlow + nex samples per read (hread: the Huffman decoder's
// per-read record - reads it has finished itself get no chunks)
__global__ __launch_bounds__(256) void k_chunk_prep_meta(const uint64_t *off, const uint64_t *in_off,
							 const ReadMeta *meta, uint32_t nreads, ChunkDesc *chunks,
							 uint64_t *gran, ChunkCtl *ctl, uint32_t max_chunks,
							 uint32_t *out_n, const uint32_t *hread)
{
	const uint32_t r = blockIdx.x * 256 + threadIdx.x;
	uint32_t n = 0, nch = 0;
	if (r < nreads && !meta[r].status) {
		n = 1u + meta[r].nlow + meta[r].nex;
		nch = (n + CHUNK - 1) / CHUNK;
		if (hread && (hread[2 * r + 1] & HUF_FUSED))
			nch = 0; // k_huf_emit wrote this read's samples already
	}
	const uint32_t inc = wave_incl_scan32(nch);
	uint32_t base = 0;
	if ((threadIdx.x & 63) == 63 && inc)
		base = atomicAdd(&ctl->nchunks, inc);
	base = (uint32_t) __builtin_amdgcn_readlane((int) base, 63);
	const uint32_t first = base + inc - nch;
	if (r >= nreads)
		return;
	out_n[r] = n ? n : CFAIL32;
	for (uint32_t j = 0; j < nch && first + j < max_chunks; j++) {
		ChunkDesc d;
		d.sig_off = off[r];
		d.out_base = in_off[r];
		d.n = n;
		d.j = j;
		d.read = r;
		d.cap_ok = 1;
		d.ebefore = 0;
		d.ecnt[0] = d.ecnt[1] = d.ecnt[2] = d.ecnt[3] = 0;
		d.kmask[0] = d.kmask[1] = d.kmask[2] = d.kmask[3] = 0;
		chunks[first + j] = d;
		gran[first + j] = 0;
	}
}

__device__ __forceinline__ uint32_t lower_bound_dev(const uint32_t *p, uint32_t n, uint32_t key)
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

// Exceptions in front of every wave quarter of every chunk (ChunkDesc::ebefore / ecnt): five
// binary searches of the read's sorted position list per chunk, one lane each - here they
// run a hundred thousand at a time instead of in front of the data loads of
// k_low_decode_chunked (22 dependent round trips per wave).
__global__ __launch_bounds__(256) void k_ex_ranks(DecodeArgs a)
{
	const uint32_t g = blockIdx.x * 256 + threadIdx.x;
	const uint32_t c = g >> 3, l = g & 7u;
	uint32_t b = 0;
	const bool act = c < a.ctl->nchunks && c < a.max_chunks;
	if (act && l < 5) {
		const ChunkDesc *dp = a.chunks + c;
		const uint32_t n = dp->n;
		const uint32_t ws = dp->j * CHUNK + l * WAVE_SAMPLES; // l = 4: the end of the chunk
		const uint32_t upto = ws < n ? ws : n;                // exceptions among samples [1, upto)
		b = lower_bound_dev(a.ex_pos + dp->sig_off, a.meta[dp->read].nex, upto ? upto - 1 : 0);
	}
	const uint32_t nxt = (uint32_t) __shfl_down((int) b, 1, 64);
	if (act && l < 4) {
		ChunkDesc *dp = a.chunks + c;
		dp->ecnt[l] = nxt - b;
		if (l == 0)
			dp->ebefore = b;
	}
}

// slow path of one sub-tile: values of the lane's 8 samples, exceptions merged in.  The
// exceptions of the sub-tile are pos[e_first, e_first + e_cnt): the search stays inside them.
__device__ __forceinline__ void gather_low(const uint8_t *low, uint32_t nlow, const uint32_t *pos,
					   const uint32_t *val, uint32_t nex, uint32_t zd0, uint32_t i0,
					   uint32_t n, uint32_t e_first, uint32_t e_cnt, uint32_t v[4])
{
	v[0] = v[1] = v[2] = v[3] = 0;
	if (i0 >= n)
		return;
	const uint32_t nv = min(8u, n - i0);
	const uint32_t uf = i0 ? i0 - 1 : 0;
	uint32_t e = e_first + lower_bound_dev(pos + e_first, e_cnt, uf);
	uint32_t l = uf - e; // index of the next one-byte value
	uint32_t nextpos = e < nex ? pos[e] : 0xFFFFFFFFu;
#pragma unroll
	for (int h = 0; h < 8; h++) {
		if ((uint32_t) h < nv) {
			const uint32_t i = i0 + h;
			uint32_t z;
			if (i == 0) {
				z = zd0;
			} else if (i - 1 == nextpos) {
				z = val[e] & 0xFFFFu;
				e++;
				nextpos = e < nex ? pos[e] : 0xFFFFFFFFu;
			} else {
				z = l < nlow ? low[l] : 0u;
				l++;
			}
			v[h >> 1] |= z << (16 * (h & 1));
		}
	}
}

// HUFF: the one-byte values come from a.low (entropy decoders); the grid may then be smaller than the
// number of chunks (workgroups keep taking tickets)
template <bool HUFF>
__global__ __launch_bounds__(CWG) void k_low_decode_chunked(DecodeArgs a)
{
	__shared__ uint32_t s_ticket;
	__shared__ uint32_t s_wsum[4];
	__shared__ uint32_t s_sbase;
	__shared__ uint16_t s_xl[4][CK][64];
	__shared__ uint32_t s_sub[4][CK];  // delta total / base per sub-tile
	__shared__ uint32_t s_cnt[4][CK];  // exceptions per sub-tile, then their exclusive prefix
	__shared__ uint32_t s_km[4];

	const int lane = threadIdx.x & 63;
	const int w = (int) uni(threadIdx.x >> 6);
#ifdef DEC_STAMPS
	const uint64_t t_start = __builtin_amdgcn_s_memtime();
#endif
	if (HUFF && uni(a.ctl->nchunks) == 0)
		return; // (every read's samples were written by k_huf_emit: no tickets to draw)
	for (;;) {
	if (threadIdx.x == 0)
		s_ticket = atomicAdd(&a.ctl->ticket, 1u);
	if (lane < CK)
		s_cnt[w][lane] = 0;
	if (lane == 0)
		s_km[w] = 0;
	__syncthreads();
	const uint32_t t = uni(s_ticket);
	if (t >= a.ctl->nchunks)
		return;
#ifdef DEC_STAMPS
	if (threadIdx.x == 0 && t < 65536)
		g_stamps[t * 8 + 0] = t_start;
#endif
	STAMP(1);
	const ChunkU d = load_chunk(a.chunks + t);
	const uint32_t n = d.n;
	const uint32_t first = d.j * CHUNK;
	const bool last = first + CHUNK >= n;
	const ReadMeta *m = a.meta + d.read;
	const uint32_t nex = uni(m->nex), nlow = uni(m->nlow), zd0 = uni(m->zd0);
	const int q = (int) uni(m->q);
	const uint32_t *pos = a.ex_pos + d.sig_off;
	const uint32_t *val = a.ex_val + d.sig_off;
	const uint8_t *low = HUFF ? a.low + d.sig_off : a.in + d.out_base + uni(m->hdr) + uni(m->seclen);
	int16_t *out = a.sig + d.sig_off;
	const uint32_t ws = first + w * WAVE_SAMPLES;
	const uint32_t nsub = ws >= n ? 0u : min((uint32_t) CK, (n - ws + SUB - 1) / SUB);
	const uint32_t live = (1u << nsub) - 1u;

	// ---- exceptions of this wave's quarter (ranks from k_ex_ranks): which sub-tiles hold one,
	// how many before each
	const ChunkDesc *dp = a.chunks + t;
	const uint32_t c0 = uni(dp->ecnt[0]), c1 = uni(dp->ecnt[1]), c2 = uni(dp->ecnt[2]), c3 = uni(dp->ecnt[3]);
	const uint32_t ew = w == 0 ? c0 : w == 1 ? c1 : w == 2 ? c2 : c3;
	const uint32_t e_lo = (uint32_t) uni64(dp->ebefore) + (w > 0 ? c0 : 0u) + (w > 1 ? c1 : 0u) + (w > 2 ? c2 : 0u);
	if (nsub) {
		const uint32_t e_hi = e_lo + ew;
		for (uint32_t e = e_lo + lane; e < e_hi; e += 64) {
			const uint32_t k = (pos[e] + 1 - ws) / SUB;
			atomicAdd(&s_cnt[w][k], 1u);
			atomicOr(&s_km[w], 1u << k);
		}
	}
	wave_lds_sync();
	uint32_t kmask = uni(s_km[w]);
	if (ws == 0)
		kmask |= 1u; // sample 0 comes from the header
	if ((n & 7) && n > ws && n - ws <= WAVE_SAMPLES)
		kmask |= 1u << ((n - 1 - ws) / SUB); // ragged tail
	if (lane < CK) {
		const uint32_t c = s_cnt[w][lane];
		const uint32_t inc = wave_incl_scan_dpp(c);
		s_cnt[w][lane] = inc - c;
	}
	wave_lds_sync();
	// sub-tiles whose 8-byte windows could reach past the one-byte stream are not plain
#pragma unroll
	for (int k = 0; k < CK; k++)
		if (((live >> k) & 1u) && (uint64_t) ws + k * SUB + SUB + 8 > (uint64_t) nlow + 1 + e_lo)
			kmask |= 1u << k;
	kmask &= live;
	const uint32_t plain = live & ~kmask;
	STAMP(2);

	// ---- phase 1: one-byte values of the plain sub-tiles (8 per lane, any alignment)
	const uint8_t *zeros8 = reinterpret_cast<const uint8_t *>(a.ctl->pad2); // (zeroed with the control block)
	uint2 dat[CK];
#pragma unroll
	for (int k = 0; k < CK; k++) {
		const uint32_t i0 = ws + k * SUB + lane * 8;
		// (every sub-tile loads, the ones that are not plain and the lanes behind the read's end zeros from the
		// control block: with the load under a branch or a lane mask the compiler merges the zero and the loaded
		// value in other registers right behind the load and waits for it - these sixteen loads went out one
		// after the other)
		const uint32_t eb = e_lo + uni(s_cnt[w][k]);
		const uint8_t *src = (((plain >> k) & 1u) && i0 < n) ? low + (i0 - 1 - eb) : zeros8;
		const unsigned long long d64 = ld8_stream(src);
		dat[k] = make_uint2((uint32_t) d64, (uint32_t) (d64 >> 32));
	}

	// ---- phase 2: delta sums
#pragma unroll
	for (int k = 0; k < CK; k++) {
		if ((plain >> k) & 1u) {
			uint32_t v[4];
			expand8(dat[k], v);
			uint32_t acc = 0;
#pragma unroll
			for (int h = 0; h < 4; h++)
				acc = pk_add16(acc, unzz_pair(v[h]));
			const uint32_t tot16 = (acc + (acc >> 16)) & 0xFFFFu;
			const uint32_t inc = wave_incl_scan_dpp(tot16);
			s_xl[w][k][lane] = (uint16_t) (inc - tot16);
			if (lane == 63)
				s_sub[w][k] = inc;
		}
	}
	for (uint32_t mm = kmask; mm; mm &= mm - 1) {
		const uint32_t k = (uint32_t) __builtin_ctz(mm);
		uint32_t v[4];
		const uint32_t ef = uni(s_cnt[w][k]);
		const uint32_t ec = (k + 1 < (uint32_t) CK ? uni(s_cnt[w][k + 1 < (uint32_t) CK ? k + 1 : k]) : ew) - ef;
		gather_low(low, nlow, pos, val, nex, zd0, ws + k * SUB + lane * 8, n, e_lo + ef, ec, v);
		uint32_t acc = 0;
#pragma unroll
		for (int h = 0; h < 4; h++)
			acc = pk_add16(acc, unzz_pair(v[h]));
		const uint32_t tot16 = (acc + (acc >> 16)) & 0xFFFFu;
		const uint32_t inc = wave_incl_scan_dpp(tot16);
		s_xl[w][k][lane] = (uint16_t) (inc - tot16);
		if (lane == 63)
			s_sub[w][k] = inc;
	}
	wave_lds_sync();
	uint32_t wtot = 0;
	{
		const uint32_t v = (lane < (int) nsub) ? s_sub[w][lane < CK ? lane : 0] : 0u;
		const uint32_t inc = wave_incl_scan_dpp(v);
		wave_lds_sync();
		if (lane < CK)
			s_sub[w][lane] = inc - v;
		wtot = (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
	}
	wave_lds_sync();
	if (lane == 0)
		s_wsum[w] = wtot & 0xFFFFu;
	STAMP(3);
	__syncthreads();
	STAMP(4);
	const uint32_t u0 = uni(s_wsum[0]), u1 = uni(s_wsum[1]), u2 = uni(s_wsum[2]), u3 = uni(s_wsum[3]);
	if (w == 0) {
		const uint32_t sv = (uint32_t) lookback(a.gran, t, d.j, (uint64_t) ((u0 + u1 + u2 + u3) & 0xFFFFu), last);
		if (lane == 0)
			s_sbase = sv;
	}
	__syncthreads();
	STAMP(5);
	const uint32_t sb = uni(s_sbase) + (w > 0 ? u0 : 0u) + (w > 1 ? u1 : 0u) + (w > 2 ? u2 : 0u);

	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
	for (int k = 0; k < CK; k++)
		asm volatile("" : "+v"(dat[k].x), "+v"(dat[k].y));

	// ---- phase 3: values, prefix inside the lane, bases, ex-zd shift, store
	const u16x2 qq = { (unsigned short) q, (unsigned short) q };
#pragma unroll
	for (int k = 0; k < CK; k++) {
		const uint32_t i0 = ws + k * SUB + lane * 8;
		if ((plain >> k) & 1u) {
			uint32_t v[4];
			expand8(dat[k], v);
#pragma unroll
			for (int h = 0; h < 4; h++)
				v[h] = unzz_pair(v[h]);
			(void) lane_prefix8(v);
			const uint32_t b16 = (sb + uni(s_sub[w][k]) + s_xl[w][k][lane]) & 0xFFFFu;
			const uint32_t b2 = b16 | (b16 << 16);
#pragma unroll
			for (int h = 0; h < 4; h++) {
				v[h] = pk_add16(v[h], b2);
				if (q) // ex_zd.c:396 do_rev_qts_inplace
					v[h] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, v[h]) << qq);
			}
			if (i0 < n)
				st16_stream(out + i0, make_uint4(v[0], v[1], v[2], v[3]));
		}
	}
	for (uint32_t mm = kmask; mm; mm &= mm - 1) {
		const uint32_t k = (uint32_t) __builtin_ctz(mm);
		const uint32_t i0 = ws + k * SUB + lane * 8;
		uint32_t v[4];
		const uint32_t ef = uni(s_cnt[w][k]);
		const uint32_t ec = (k + 1 < (uint32_t) CK ? uni(s_cnt[w][k + 1 < (uint32_t) CK ? k + 1 : k]) : ew) - ef;
		gather_low(low, nlow, pos, val, nex, zd0, i0, n, e_lo + ef, ec, v);
#pragma unroll
		for (int h = 0; h < 4; h++)
			v[h] = unzz_pair(v[h]);
		(void) lane_prefix8(v);
		const uint32_t b16 = (sb + uni(s_sub[w][k]) + s_xl[w][k][lane]) & 0xFFFFu;
		const uint32_t b2 = b16 | (b16 << 16);
#pragma unroll
		for (int h = 0; h < 4; h++) {
			v[h] = pk_add16(v[h], b2);
			if (q)
				v[h] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, v[h]) << qq);
		}
		if (i0 + 8 <= n) {
			st16_stream(out + i0, make_uint4(v[0], v[1], v[2], v[3]));
		} else if (i0 < n) {
#pragma unroll
			for (uint32_t h = 0; h < 8; h++)
				if (h < n - i0)
					out[i0 + h] = (int16_t) (v[h >> 1] >> (16 * (h & 1)));
		}
	}
	STAMP(6);
	if (!HUFF)
		return;
	__syncthreads(); // the ticket and the tables of this chunk are free again
	}
}

// ------------------------------------------------------------------ launchers

template <bool KEY2, bool ZD, bool S5 = false, bool HIST = false>
static void run_encode(const BatchArgs &a, hipStream_t s)
{
	(void) hipMemsetAsync(a.ctl, 0, sizeof(ChunkCtl), s);
	hipLaunchKernelGGL((k_chunk_prep<false, KEY2>), dim3((a.nreads + 255) / 256), dim3(256), 0, s, a.off,
			   a.nsamp, a.out_off, (const uint64_t *) nullptr, a.nreads, a.chunks, a.gran, a.ctl,
			   a.max_chunks, a.out_len, (uint32_t *) nullptr, a.first_chunk, (ReadMeta *) nullptr,
			   (const uint8_t *) nullptr, S5 ? 4u : 0u);
	// persistent grid: as many workgroups as are resident (4 per CU)
	const uint32_t grid = a.max_chunks < PERSISTENT_GRID ? a.max_chunks : PERSISTENT_GRID;
	ktime_begin(0, s);
	hipLaunchKernelGGL((k_svb_encode_chunked<KEY2, ZD, S5, HIST>), dim3(grid), dim3(CWG), 0, s, a);
	ktime_end(0, s);
}

void launch_svb_encode_chunked(const BatchArgs &a, bool key2bit, bool zd, hipStream_t s, bool slow5)
{
	if (!a.nreads || !a.max_chunks)
		return;
	if (slow5)
		run_encode<true, true, true>(a, s);
	else if (a.zhist && key2bit)
		run_encode<true, true, false, true>(a, s);
	else if (a.zhist)
		run_encode<false, true, false, true>(a, s);
	else if (key2bit)
		run_encode<true, true>(a, s);
	else if (zd)
		run_encode<false, true>(a, s);
	else
		run_encode<false, false>(a, s);
}

template <bool KEY2, bool ZD, bool S5 = false>
static void run_decode(const DecodeArgs &a, hipStream_t s)
{
	(void) hipMemsetAsync(a.ctl, 0, sizeof(ChunkCtl), s);
	hipLaunchKernelGGL((k_chunk_prep<true, KEY2>), dim3((a.nreads + 255) / 256), dim3(256), 0, s, a.off,
			   a.nsamp, a.in_off, a.in_len, a.nreads, a.chunks, a.gran, a.ctl, a.max_chunks,
			   (uint64_t *) nullptr, a.out_n, a.first_chunk, (ReadMeta *) nullptr, a.in,
			   S5 ? 4u : 0u);
	// surplus workgroups (max_chunks bounds the real count from above) exit at once
	hipLaunchKernelGGL((k_svb_keyscan<KEY2, S5>), dim3((a.max_chunks + 3) / 4), dim3(CWG), 0, s, a);
	hipLaunchKernelGGL((k_svb_keyprefix<KEY2, S5>), dim3((a.nreads + 3) / 4), dim3(256), 0, s, a);
	ktime_begin(1, s);
	hipLaunchKernelGGL((k_svb_decode_chunked<KEY2, ZD, S5>), dim3(a.max_chunks), dim3(CWG), 0, s, a);
	ktime_end(1, s);
}

// exception-split encode: chunked scan, section per read, chunked pass B
void launch_ex_encode_chunked(const BatchArgs &a, int fmt, int ent, hipStream_t s)
{
	const bool huff = ent == 1;
	if (!a.nreads || !a.max_chunks)
		return;
	(void) hipMemsetAsync(a.ctl, 0, sizeof(ChunkCtl), s);
	hipLaunchKernelGGL((k_chunk_prep<false, false>), dim3((a.nreads + 255) / 256), dim3(256), 0, s, a.off,
			   a.nsamp, a.out_off, (const uint64_t *) nullptr, a.nreads, a.chunks, a.gran, a.ctl,
			   a.max_chunks, a.out_len, (uint32_t *) nullptr, a.first_chunk, a.meta);
	// surplus workgroups (max_chunks bounds the real count from above) exit at once
	if (huff)
		hipLaunchKernelGGL((k_ex_scan_chunked<false, true>), dim3(a.max_chunks), dim3(CWG), 0, s, a);
	else
		hipLaunchKernelGGL((k_ex_scan_chunked<false>), dim3(a.max_chunks), dim3(CWG), 0, s, a);
	if (fmt == EXF_EXZD) {
		// second scan on the shifted samples for reads with q > 0
		hipLaunchKernelGGL(k_ex_redo_flag, dim3((a.nreads + 255) / 256), dim3(256), 0, s, a);
		hipLaunchKernelGGL((k_ex_scan_chunked<true>), dim3(a.max_chunks), dim3(CWG), 0, s, a);
	}
	hipLaunchKernelGGL(k_ex_prefix, dim3((a.nreads + 3) / 4), dim3(256), 0, s, a, fmt == EXF_EXZD ? 1 : 0);
	hipLaunchKernelGGL(k_ex_list, dim3((a.max_chunks + 3) / 4), dim3(CWG), 0, s, a);
	launch_ex_section(a, fmt, ent, s);
	if (ent >= 2) { // range coder: the one-byte values go to a temporary, one lane (order 1: one workgroup) per read codes them
		hipLaunchKernelGGL(k_low_encode_chunked<false>, dim3(a.max_chunks), dim3(CWG), 0, s, a);
		ktime_begin(0, s);
		if (ent == 2)
			launch_rcs_encode(a, s);
		else if (ent == 3)
			launch_rcc_encode(a, s);
		else
			launch_rcm_encode(a, s);
		ktime_end(0, s);
		return;
	}
	ktime_begin(0, s);
	if (huff)
		hipLaunchKernelGGL(k_huff_encode_chunked, dim3(a.max_chunks), dim3(CWG), 0, s, a);
	else if (a.zhist)
		hipLaunchKernelGGL(k_low_encode_chunked<true>, dim3(a.max_chunks), dim3(CWG), 0, s, a);
	else
		hipLaunchKernelGGL(k_low_encode_chunked<false>, dim3(a.max_chunks), dim3(CWG), 0, s, a);
	ktime_end(0, s);
}

// exception-split decode: parse + (Huffman) from press_sections.hip / press_huffman.hip, then the chunked merge
void launch_ex_decode_chunked(const DecodeArgs &a, int fmt, int ent, hipStream_t s)
{
	const bool huff = ent != 0; // the one-byte stream comes from a.low (Huffman or range decoder)
	if (!a.nreads || !a.max_chunks)
		return;
	launch_ex_parse_huff(a, fmt, ent, s); // (k_ex_parse clears both control blocks)
	hipLaunchKernelGGL(k_chunk_prep_meta, dim3((a.nreads + 255) / 256), dim3(256), 0, s, a.off, a.in_off,
			   a.meta, a.nreads, a.chunks, a.gran, a.ctl, a.max_chunks, a.out_n,
			   ent == 1 ? (const uint32_t *) a.hread : (const uint32_t *) nullptr);
	hipLaunchKernelGGL(k_ex_ranks, dim3((a.max_chunks * 8 + 255) / 256), dim3(256), 0, s, a);
	if (!huff)
		ktime_begin(1, s);
	if (ent == 1) {
		// static Huffman: k_huf_emit has written the samples of every read whose lists interleave cleanly;
		// the chunk table holds the others only (normally none)
		const uint32_t grid = a.max_chunks < 1024u ? a.max_chunks : 1024u;
		hipLaunchKernelGGL((k_low_decode_chunked<true>), dim3(grid), dim3(CWG), 0, s, a);
	} else if (huff) {
		hipLaunchKernelGGL((k_low_decode_chunked<true>), dim3(a.max_chunks), dim3(CWG), 0, s, a);
	} else {
		hipLaunchKernelGGL((k_low_decode_chunked<false>), dim3(a.max_chunks), dim3(CWG), 0, s, a);
	}
	if (!huff)
		ktime_end(1, s);
}

void launch_svb_decode_chunked(const DecodeArgs &a, bool key2bit, bool zd, hipStream_t s, bool slow5)
{
	if (!a.nreads || !a.max_chunks)
		return;
	if (slow5)
		run_decode<true, true, true>(a, s);
	else if (key2bit)
		run_decode<true, true>(a, s);
	else if (zd)
		run_decode<false, true>(a, s);
	else
		run_decode<false, false>(a, s);
}

} // namespace ph
