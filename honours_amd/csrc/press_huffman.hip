// press_huffman.hip - static-Huffman stream decode for gfx950 (huffman.c:1219 huffman_decode_memory).
//
// The stream has no synchronisation points, but Huffman codes self-synchronise: a decoder
// started at a wrong bit position falls back onto true code boundaries after a few codes.
// The payload of every read is cut into TILES of HT subsequences of S bits (S = 128 for
// the NA12878 table); one 512-thread workgroup per tile, handed out in ticket order by a
// persistent grid that keeps the lookup tables in LDS (53 KB per workgroup: 3 per CU).
// Thread i owns the codes that START in subsequence i:
//   pass 0     thread i runs up through subsequence i-1 from its first bit (thread 0: the last
//              subsequence of the previous tile) to where the first code of subsequence i
//              starts, then decodes its own subsequence from there;
//   rounds     thread i takes the end E[i-1] of its left neighbour as its start; whoever's
//              start differs from what it used decodes again -
//              the changed threads are compacted so that a round with few changes costs a
//              wave, not eight - until nothing changes (thread i is final after <= i rounds;
//              in practice after 2-3).  These passes leave their symbols in a private
//              36-byte LDS slot per thread;
//   look-back  a tile other than a read's first does not know where its first code starts.
//              It converges from what pass 0 found across the tile boundary (right in ~97 %
//              of the tiles), then publishes an AGGREGATE granule
//              {start it assumed, where its last code ends, codes it holds}.  The end of
//              the predecessor's granule is the tile's real start: if it differs from the
//              assumption the tile converges again from there (one or two lanes decode) and
//              publishes the new aggregate.  A granule whose assumed start is the true start
//              carries true values, whenever it was read; so a tile walks back over its
//              predecessors' granules - 64 per round trip - checking that each one's
//              assumed start equals the end of the one before it, down to a PREFIX granule
//              {true end, codes up to and including that tile} (a read's first tile always
//              publishes one).  The sum is its output offset, and it publishes a prefix
//              itself.  No tile waits for another tile's look-back; ticket order makes the
//              waits deadlock free (a predecessor was always taken by a workgroup that is
//              already running), and a wrong assumption only costs time;
//   output     prefix sum of the counts, then every thread copies its slot to the one-byte
//              stream with (unaligned) 4-byte stores.
// Result == huffman.c:1219 bit for bit, including its behaviour at the end of the input
// (stops when the bytes run out or the symbol count is reached; a code cut off by the end
// of the input is not delivered).

#include "press_internal.h"

namespace ph {

constexpr uint32_t HEND = 0xFFFFFFFFu; // "no further code": end of input or an undecodable prefix
constexpr int HSLOT = 36;              // bytes per private symbol slot: HSYM + spill; 9 dwords = bank-conflict-free stride
constexpr int HLB_DW = 2096;           // LDS dwords of the bit image (one subsequence before the tile + 2048 + reach of the last code, skewed)
// packed per-thread result: position (20 bits) | END << 20 | codes << 24
constexpr uint32_t PE_END = 1u << 20;
constexpr uint32_t ST_END = 0xFEu; // sS: start relative to the subsequence, or this
// look-back granules: state << 62 | assumed start << 56 | end << 48 | codes; positions are
// relative to the tile boundary (0..23), 63 = the stream has ended
constexpr uint64_t GR_AGG = 1ull << 62, GR_PFX = 2ull << 62;
constexpr uint32_t GP_END = 63;

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); }

// LDS image of a tile's bits: dword j lives at j + (j >> 6), and the slot after every 64th
// dword repeats the dword that follows it.  So dwords j and j+1 are always adjacent (one
// ds_read2_b32), and the 64 lanes of a wave - one subsequence apart - hit 64 different banks.
__device__ __forceinline__ uint32_t la(uint32_t j) { return j + (j >> 6); }

// Decode the codes that start in [start, sub_end) of the image (bit positions relative to
// the image); returns where the next code starts, or HEND.  nbits = end of the payload.
// WRITE: symbols go to `slot`, their number to `cnt`.
// The lane keeps a 64-bit window {hi, lo} of the stream in registers and fetches the dword
// after it while the table lookup is in flight, so a step costs one LDS round trip.
template <bool WRITE>
__device__ __forceinline__ uint32_t huff_run(const uint32_t *lbits, const uint32_t *lut, const uint16_t *lut2,
					     const uint16_t *l2off, const uint8_t *l2bits, const HuffDev *hd,
					     uint32_t start, uint32_t sub_end, uint32_t nbits, uint32_t &cnt, uint8_t *slot)
{
	bool bad = start == HEND;
	uint32_t p = bad ? 0u : start;
	uint32_t lim = sub_end < nbits ? sub_end : nbits; // codes must START below this
	if (bad)
		lim = 0;
	uint32_t c = 0;
	uint32_t j = p >> 5;
	uint32_t lo = lbits[la(j)], hi = lbits[la(j) + 1];
	// wave-uniform and predicated: lanes need different numbers of steps, and per-lane
	// branches cost more than the few masked operations
	for (;;) {
		const bool act = p < lim;
		if (!__any(act))
			break;
		if (act) {
			const uint32_t nxt = lbits[la(j + 2)];
			// 32 stream bits from position p (codes are at most 24 bits long)
			const uint32_t wnd = __builtin_amdgcn_alignbit(hi, lo, p & 31);
			uint32_t e = lut[wnd & ((1u << HUF_LUT_BITS) - 1u)];
			if (e >= HUF_LONG) { // rare: a code longer than 12 bits
				uint32_t sym = 0, len = 0;
				bool ok;
				const uint32_t id = e & 0xFFu;
				if (e != 0xFFFFFFFFu && id < (uint32_t) HUF_L2_IDS) {
					const uint32_t e2 = lut2[l2off[id] + ((wnd >> HUF_LUT_BITS) & ((1u << l2bits[id]) - 1u))];
					ok = e2 != 0xFFFFu;
					sym = e2 & 0xFFu;
					len = e2 >> 8;
				} else {
					int node = 0;
					while (node >= 0 && hd->leaf[node] < 0 && len < 32) {
						node = hd->child[node][(wnd >> len) & 1u];
						len++;
					}
					ok = node >= 0 && hd->leaf[node] >= 0;
					sym = ok ? (uint32_t) hd->leaf[node] : 0u;
				}
				if (!ok) { // no such code: the reference stops here
					bad = true;
					lim = 0;
					len = 0;
				}
				e = sym | (len << 16) | (len << 21) | (ok ? (1u << 26) : 0u);
			}
			// e = sym1 | sym2 << 8 | len1 << 16 | (len1 + len2) << 21 | (codes: 1 or 2) << 26; the second
			// code counts only if it starts inside this subsequence
			const uint32_t len1 = (e >> 16) & 0x1Fu;
			const bool both = p + len1 < lim;
			if (WRITE) {
				slot[c] = (uint8_t) e;
				slot[c + 1] = (uint8_t) (e >> 8);
				const uint32_t n2 = e >> 26;
				c += both ? n2 : (n2 ? 1u : 0u);
			}
			p += both ? ((e >> 21) & 0x1Fu) : len1;
			const uint32_t jn = p >> 5; // a step crosses at most one dword
			if (jn != j) {
				lo = hi;
				hi = nxt;
			}
			j = jn;
		}
	}
	if (!bad && p > nbits) { // the last code ran off the end of the input: not delivered
		if (c)
			c -= 1;
		bad = true;
	}
	cnt = c;
	return bad ? HEND : (p >= nbits && p < sub_end ? HEND : p);
}

// inclusive scan over the 64 lanes of a wave (DPP row shifts + row broadcasts)
__device__ __forceinline__ uint32_t wave_scan(uint32_t v)
{
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, true); // row_shr:1
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, true); // row_shr:2
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, true); // row_shr:4
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, true); // row_shr:8
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1,3
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2,3
	return v;
}

__device__ __forceinline__ uint64_t gran_ld(uint64_t *g)
{
	return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gran_st(uint64_t *g, uint64_t v)
{
	__hip_atomic_store(g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v)
{
#pragma unroll
	for (int dd = 32; dd >= 1; dd >>= 1) {
		const uint32_t lo = (uint32_t) __shfl_xor((int) (uint32_t) v, dd, 64);
		const uint32_t hi = (uint32_t) __shfl_xor((int) (uint32_t) (v >> 32), dd, 64);
		v += ((uint64_t) hi << 32) | lo;
	}
	return v;
}

#ifdef HUF_DEBUG
__device__ unsigned long long g_hufdbg[16];
extern "C" int press_hip_debug_huff(unsigned long long *dst)
{
	int rc = (int) hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_hufdbg), sizeof g_hufdbg);
	unsigned long long z[16] = { 0 };
	(void) hipMemcpyToSymbol(HIP_SYMBOL(g_hufdbg), z, sizeof z);
	return rc;
}
// per-workgroup accumulation in registers, one atomicAdd per counter when the workgroup exits
#define HSTAMP(i) do { const unsigned long long now_ = clock64(); acc_[i] += now_ - stamp_; stamp_ = now_; } while (0)
#define HCOUNT(i, v) do { acc_[i] += (unsigned long long) (v); } while (0)
#else
#define HSTAMP(i) do { } while (0)
#define HCOUNT(i, v) do { } while (0)
#endif

// Look-back of tile k (index t >= 1 in its read), executed by one wave.  my_s = the start
// this tile's current result assumed.  Returns LB_DONE | codes in front of the tile once
// the chain of aggregates is validated down to a prefix, or the predecessor's end position
// (0..23, GP_END) if that is not my_s: the tile has to converge again from there.
constexpr uint64_t LB_DONE = 1ull << 63;
__device__ __forceinline__ uint64_t tile_lookback(uint64_t *gran, uint32_t k, uint32_t t, uint32_t my_s)
{
	const uint32_t lane = threadIdx.x & 63;
	for (;;) {
		uint64_t sum = 0;
		uint32_t need = my_s; // the end position the next (farther) granule must show
		uint32_t done = 0;    // predecessors already validated
		for (;;) {
			const uint32_t idx = done + lane; // lane l looks at tile k - 1 - idx
			const bool have = idx < t;
			uint64_t g = 0;
			if (have)
				g = gran_ld(gran + (k - 1 - idx));
			const uint32_t st = (uint32_t) (g >> 62);
			const uint32_t gs = (uint32_t) (g >> 56) & 63u;
			const uint32_t ge = (uint32_t) (g >> 48) & 63u;
			uint32_t prev_s = (uint32_t) __shfl_up((int) gs, 1, 64);
			if (lane == 0)
				prev_s = need;
			const bool pub = have && st != 0;
			const bool linkok = pub && ge == prev_s;
			const unsigned long long badm = __ballot(have && !linkok);
			const unsigned long long pfxm = __ballot(linkok && st == 2);
			const uint32_t fb = badm ? (uint32_t) __builtin_ctzll(badm) : 64u;
			const uint32_t fp = pfxm ? (uint32_t) __builtin_ctzll(pfxm) : 64u;
			if (fp < fb) // validated down to a prefix
				return LB_DONE | (sum + wave_sum64(lane <= fp ? (g & 0xFFFFFFFFull) : 0ull));
			if (fb < 64) {
				const uint32_t pub0 = (uint32_t) __shfl((int) (pub ? 1 : 0), 0, 64);
				const uint32_t ge0 = (uint32_t) __shfl((int) ge, 0, 64);
				if (fb == 0 && done == 0 && pub0)
					return ge0; // the predecessor ends somewhere else than assumed
				break;              // not published yet / being corrected by its owner: poll again
			}
			// 64 consistent aggregates and no prefix among them: keep walking
			sum += wave_sum64(g & 0xFFFFFFFFull);
			need = (uint32_t) __shfl((int) gs, 63, 64);
			done += 64;
		}
#ifdef HUF_DEBUG
		if (lane == 0)
			atomicAdd(&g_hufdbg[14], 1ull); // polls that found the chain incomplete
#endif
		__builtin_amdgcn_s_sleep(4);
	}
}


__global__ __launch_bounds__(HUF_HT, 6) void k_huff_decode_tiles(DecodeArgs a) // 6 waves per SIMD = 3 workgroups per CU
{
#ifdef HUF_DEBUG
	unsigned long long stamp_ = clock64();
	unsigned long long acc_[16] = { 0 };
#endif
	constexpr int HT = HUF_HT;
	__shared__ uint32_t lut[1 << HUF_LUT_BITS];
	__shared__ uint16_t lut2[HUF_L2_ENTRIES];
	__shared__ uint16_t l2off[HUF_L2_IDS];
	__shared__ uint8_t l2bits[HUF_L2_IDS];
	__shared__ uint32_t lbits[HLB_DW];
	__shared__ uint32_t sE[HT + 1]; // sE[i]: where the first code of image subsequence i+1... see below
	__shared__ uint8_t sS[HT];      // the start (relative to its subsequence) thread i's result was decoded from
	__shared__ uint16_t lst[HT];    // compacted list of threads whose start changed: id | start << 9 | END << 14
	__shared__ uint32_t lst_n[2];
	__shared__ uint32_t wtot[2][HT / 64];
	__shared__ uint32_t s_ticket;
	__shared__ uint64_t s_res;
	__shared__ __attribute__((aligned(16))) uint8_t slots[HT * HSLOT];

	// Positions inside the kernel are relative to the LDS image, which starts one subsequence
	// before the tile: image subsequence 0 is the previous tile's last one, thread i owns image
	// subsequence i+1.  sE[i] = where the first code at or after the start of image subsequence
	// i+1 begins (| PE_END), and for i >= 1 | (codes of thread i-1) << 24.
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	for (uint32_t i = tid; i < (1u << HUF_LUT_BITS); i += HT)
		lut[i] = a.huff->lut32[i];
	for (uint32_t i = tid; i < (uint32_t) HUF_L2_ENTRIES; i += HT)
		lut2[i] = a.huff->lut2[i];
	if (tid < (uint32_t) HUF_L2_IDS) {
		l2off[tid] = a.huff->l2off[tid];
		l2bits[tid] = a.huff->l2bits[tid];
	}
	const uint32_t minlen = uniform(a.huff->minlen);
	// bits per subsequence: at most HUF_HSYM = 32 codes start in one, and it holds the longest code (24)
	const uint32_t S = minlen >= 4 ? 128u : minlen >= 2 ? 64u : 32u;
	const uint32_t TB = HT * S;
	const uint32_t nalloc = uniform(a.ctl->nchunks);
	const uint32_t ntiles = nalloc < a.max_htiles ? nalloc : a.max_htiles;
	const uint32_t ndw = (TB + S) / 32 + 4; // dwords of the image
	uint32_t round = 0;                     // parity selects lst_n / wtot
	uint32_t cnt = 0, excl = 0, total = 0;  // own codes, codes of the threads before, codes of the tile

	// converge: repeat "take the left neighbour's end as start, decode again if it changed"
	// until nothing changes; leaves the prefix sums of the counts in cnt / excl / total.
	// Barriers inside; all threads call it.
	auto converge = [&](uint32_t nbits) {
		for (;;) {
			const uint32_t par = round & 1u;
			round++;
			if (tid == 0)
				lst_n[par] = 0;
			cnt = sE[tid + 1] >> 24; // own write, or settled by the barrier that ended the last round
			const uint32_t inc = wave_scan(cnt);
			if (lane == 63)
				wtot[par][tid >> 6] = inc;
			__syncthreads();
			const uint32_t pe = sE[tid];
			const uint32_t code = (pe & PE_END) ? ST_END : ((pe & 0xFFFFFu) - (tid + 1) * S);
			const bool ch = code != sS[tid];
			const unsigned long long m = __ballot(ch);
			uint32_t base = 0;
			if (lane == 0 && m)
				base = atomicAdd(&lst_n[par], (uint32_t) __popcll(m));
			base = (uint32_t) __shfl((int) base, 0, 64);
			if (ch) {
				const uint32_t idx = base + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
				lst[idx] = (uint16_t) (tid | ((code & 31u) << 9) | ((pe & PE_END) ? (1u << 14) : 0u));
			}
			__syncthreads();
			HSTAMP(12); // detection
			const uint32_t nch = lst_n[par];
			if (nch == 0) {
				uint32_t before = 0;
				total = 0;
#pragma unroll
				for (int w2 = 0; w2 < HT / 64; w2++) {
					const uint32_t x = wtot[par][w2];
					if (w2 < (int) (tid >> 6))
						before += x;
					total += x;
				}
				excl = before + inc - cnt;
				break;
			}
			HCOUNT(9, 1);
			HCOUNT(11, nch);
			if ((tid & ~63u) < nch) { // whole waves beyond the list skip
				const bool mine = tid < nch;
				const uint32_t ent = mine ? lst[tid] : (1u << 14);
				const uint32_t u = ent & 511u;
				const bool end = (ent >> 14) & 1u;
				const uint32_t rel = (ent >> 9) & 31u;
				uint32_t c;
				// the whole workgroup waits for this wave: let it win the issue arbitration against the
				// first-pass waves of the other workgroups on its SIMD
				__builtin_amdgcn_s_setprio(3);
				const uint32_t e = huff_run<true>(lbits, lut, lut2, l2off, l2bits, a.huff, end ? HEND : (u + 1) * S + rel,
								  (u + 2) * S, nbits, c, slots + u * HSLOT);
				__builtin_amdgcn_s_setprio(0);
				if (mine) {
					sS[u] = (uint8_t) (end ? ST_END : rel);
					sE[u + 1] = (e == HEND ? PE_END : e) | (c << 24);
				}
			}
			__syncthreads(); // counts settled before the next round's scan
			HSTAMP(13); // re-decode
		}
	};

	// loads of a tile's image dwords tid + q * HT (zeros outside the payload)
	auto load_bits = [&](const HufTile &d, uint32_t v[5]) {
		const uint8_t *src = a.in + d.src;
		const int64_t nby = (int64_t) (d.nbits >> 3); // payload bytes from the tile start on
		const bool first = (d.t_last & 0x7FFFFFFFu) == 0;
#pragma unroll
		for (int q = 0; q < 5; q++) {
			const uint32_t j = tid + q * HT;
			const int64_t rb = 4ll * j - (int64_t) (S >> 3); // byte offset from the tile start
			v[q] = 0;
			if (j < ndw && (rb >= 0 || !first) && rb + 4 <= nby)
				__builtin_memcpy(&v[q], src + rb, 4);
		}
	};

	if (tid == 0)
		s_ticket = atomicAdd(&a.ctl->ticket, 1u);
	__syncthreads();
	uint32_t k = s_ticket;
	HufTile d;
	if (k < ntiles)
		d = a.htiles[k];
	while (k < ntiles) {
		uint32_t v[5];
		load_bits(d, v);
		HSTAMP(0); // ticket
		HCOUNT(8, 1);
		const uint32_t t = uniform(d.t_last) & 0x7FFFFFFFu;
		const bool last = (uniform(d.t_last) >> 31) != 0; // no tile of this read follows
		const uint32_t r = uniform(d.read);
		const uint32_t want = uniform(d.want);
		const uint32_t nbits = uniform(d.nbits) + S; // payload end in image coordinates
		uint8_t *low = a.low + d.low;

		// ---- stage the tile's bits: lbits[la(j)] = image dword j
		{
			const uint8_t *src = a.in + d.src;
			const int64_t nby = (int64_t) (d.nbits >> 3);
#pragma unroll
			for (int q = 0; q < 5; q++) {
				const uint32_t j = tid + q * HT;
				const int64_t rb = 4ll * j - (int64_t) (S >> 3);
				if (j < ndw) {
					if ((rb >= 0 || t) && rb < nby && rb + 4 > nby) { // the dword that straddles the end of the payload
						for (int64_t i = 0; rb + i < nby; i++)
							v[q] |= (uint32_t) src[rb + i] << (8 * i);
					}
					lbits[la(j)] = v[q];
					if ((j & 63u) == 0 && j)
						lbits[la(j) - 1] = v[q];
				}
			}
		}
		__syncthreads();
		HSTAMP(1); // stage

		// ---- first pass: run up through image subsequence tid from its first bit (no output), then the
		// own subsequence tid + 1 with symbols (a read's first tile starts exactly at its bit 0).
		// Two loops rather than one with a mode per lane: each is leaner than the fused form.
		{
			uint32_t c, c0;
			const bool exact = t == 0 && tid == 0;
			uint32_t f = S;
			if (!__all(exact)) {
				f = huff_run<false>(lbits, lut, lut2, l2off, l2bits, a.huff, tid * S, (tid + 1) * S, nbits, c0, nullptr);
				if (f == HEND && (tid + 1) * S < nbits)
					f = (tid + 1) * S; // a guess that ran into a bit pattern that is no code: any guess will do
				if (exact)
					f = S;
			}
			const uint32_t e = huff_run<true>(lbits, lut, lut2, l2off, l2bits, a.huff, f, (tid + 2) * S, nbits, c,
							  slots + tid * HSLOT);
			sS[tid] = (uint8_t) (f == HEND ? ST_END : f - (tid + 1) * S);
			sE[tid + 1] = (e == HEND ? PE_END : e) | (c << 24);
			if (tid == 0)
				sE[0] = f == HEND ? PE_END : f; // the tile's guess of its own start
		}
		HSTAMP(2); // first pass
		converge(nbits);
		HSTAMP(3); // rounds

		// ---- look-back, (rarely) another convergence
		uint64_t cum_prev = 0;
		uint32_t kn = 0xFFFFFFFFu;
		for (;;) {
			if (tid < 64) { // wave 0
				const uint32_t pe = sE[HT], p0 = sE[0];
				const uint64_t ge = (pe & PE_END) ? GP_END : ((pe & 0xFFFFFu) - (TB + S));
				const uint32_t my_s = (p0 & PE_END) ? GP_END : ((p0 & 0xFFFFFu) - S);
				uint64_t res;
				if (t == 0) {
					if (!last && lane == 0)
						gran_st(a.hgran + k, GR_PFX | (ge << 48) | total);
					res = LB_DONE;
				} else {
					if (!last && lane == 0)
						gran_st(a.hgran + k, GR_AGG | ((uint64_t) my_s << 56) | (ge << 48) | total);
					res = tile_lookback(a.hgran, k, t, my_s);
					if ((res & LB_DONE) && !last && lane == 0) {
						const uint64_t cum = (res & 0xFFFFFFFFull) + total;
						gran_st(a.hgran + k, GR_PFX | (ge << 48) | (cum > 0xFFFFFFFFull ? 0xFFFFFFFFull : cum));
					}
				}
				if (lane == 0) {
					s_res = res;
					if (res & LB_DONE) // nothing left to wait for: the next tile (its descriptor arrives during the output)
						s_ticket = atomicAdd(&a.ctl->ticket, 1u);
					else // the predecessor ends elsewhere: that is thread 0's start
						sE[0] = (uint32_t) res == GP_END ? PE_END : S + (uint32_t) res;
				}
			}
			__syncthreads();
			const uint64_t res = s_res;
			HSTAMP(4); // look-back
			if (res & LB_DONE) {
				cum_prev = res & 0xFFFFFFFFull;
				kn = s_ticket;
				break;
			}
			HCOUNT(10, 1);
			if ((uint32_t) res == GP_END) { // the stream ended before this tile
				sS[tid] = (uint8_t) ST_END;
				sE[tid + 1] = PE_END;
				__syncthreads();
			}
			converge(nbits);
			HSTAMP(5); // rounds, corrected start
		}
		HufTile dn;
		if (kn < ntiles)
			dn = a.htiles[kn];

		// ---- output: this tile delivers symbols [cum_prev, cum_prev + total) of the read, cut at `want`
		const uint32_t o0 = cum_prev < want ? (uint32_t) cum_prev : want;
		const uint32_t room = want - o0;
		const uint32_t take = total < room ? total : room;
		{
			const uint32_t nmine = excl < take ? (cnt < take - excl ? cnt : take - excl) : 0u;
			const uint32_t *sl = reinterpret_cast<const uint32_t *>(slots + tid * HSLOT);
			uint8_t *dst = low + o0 + excl;
			uint32_t q = 0;
			for (; q + 4 <= nmine; q += 4) {
				const uint32_t w = sl[q >> 2];
				__builtin_memcpy(dst + q, &w, 4);
			}
			if (q < nmine) {
				const uint32_t w = sl[q >> 2];
				dst[q] = (uint8_t) w;
				if (q + 1 < nmine)
					dst[q + 1] = (uint8_t) (w >> 8);
				if (q + 2 < nmine)
					dst[q + 2] = (uint8_t) (w >> 16);
			}
		}
		if (last && tid == 0)
			a.meta[r].nlow = o0 + take; // what huffman_decode_memory delivered
		HSTAMP(7); // output
		k = kn;
		d = dn;
		__syncthreads(); // this tile's LDS is no longer in use
	}
#ifdef HUF_DEBUG
	if (threadIdx.x == 0)
		for (int i = 0; i < 16; i++)
			atomicAdd(&g_hufdbg[i], acc_[i]);
#endif
}

// Tiles of every read (one thread per read; runs after k_ex_parse): ids of one read are
// consecutive, so a tile's predecessor is id - 1.  Also zeroes the tiles' granules.
__global__ __launch_bounds__(256) void k_huff_tiles(DecodeArgs a)
{
	const uint32_t r = blockIdx.x * 256 + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t minlen = a.huff->minlen, maxlen = a.huff->maxlen;
	const uint32_t S = minlen >= 4 ? 128u : minlen >= 2 ? 64u : 32u;
	const uint64_t TB = (uint64_t) HUF_HT * S;
	uint32_t nt = 0;
	if (r < a.nreads && a.meta[r].status == 0) {
		const ReadMeta *m = a.meta + r;
		const uint32_t hdr = m->hdr + m->seclen + 4;
		const uint64_t nbytes64 = a.in_len[r] - hdr;
		const uint64_t nbits = 8ull * (nbytes64 > 0x1FFFFFFFull ? 0x1FFFFFFFull : nbytes64);
		// the first `nlow` codes end within nlow * maxlen bits: later tiles could not deliver anything
		const uint64_t need = (uint64_t) m->nlow * maxlen;
		const uint64_t span = nbits < need ? nbits : need;
		nt = (uint32_t) ((span + TB - 1) / TB);
	}
	// wave-aggregated allocation
	uint32_t inc = nt;
#pragma unroll
	for (int dd = 1; dd < 64; dd <<= 1) {
		const uint32_t t2 = (uint32_t) __shfl_up((int) inc, dd, 64);
		if ((int) lane >= dd)
			inc += t2;
	}
	const uint32_t wsum = (uint32_t) __shfl((int) inc, 63, 64);
	uint32_t base = 0;
	if (lane == 0 && wsum)
		base = atomicAdd(&a.ctl->nchunks, wsum);
	base = (uint32_t) __shfl((int) base, 0, 64) + inc - nt;
	if (nt) {
		const ReadMeta *m = a.meta + r;
		const uint32_t hdr = m->hdr + m->seclen + 4;
		const uint64_t nbytes64 = a.in_len[r] - hdr;
		const uint64_t nbits = 8ull * (nbytes64 > 0x1FFFFFFFull ? 0x1FFFFFFFull : nbytes64);
		for (uint32_t t = 0; t < nt; t++) {
			const uint32_t k = base + t;
			if (k >= a.max_htiles)
				break; // cannot happen: max_htiles is the same bound summed over the slots
			HufTile d;
			d.src = a.in_off[r] + hdr + t * (TB / 8);
			d.low = a.off[r];
			d.nbits = (uint32_t) (nbits - t * TB);
			d.t_last = t | (t + 1 == nt ? 0x80000000u : 0u);
			d.read = r;
			d.want = m->nlow;
			a.htiles[k] = d;
			a.hgran[k] = 0;
		}
	}
}

// Huffman stage of the exception-split decoders: payload of every read -> a.low
void launch_huff_decode(const DecodeArgs &a, hipStream_t s)
{
	(void) hipMemsetAsync(a.ctl, 0, sizeof(ChunkCtl), s);
	hipLaunchKernelGGL(k_huff_tiles, dim3((a.nreads + 255) / 256), dim3(256), 0, s, a);
	const uint32_t grid = a.max_htiles < HUF_GRID ? a.max_htiles : HUF_GRID;
	hipLaunchKernelGGL(k_huff_decode_tiles, dim3(grid ? grid : 1), dim3(HUF_HT), 0, s, a);
}

} // namespace ph
