// press_huffman.hip - static-Huffman stream decode for gfx950 (huffman.c:1219 shuffman_decode_memory).
//
// The stream has no synchronisation points, but Huffman codes self-synchronise: a decoder
// started at a wrong bit position falls back onto true code boundaries after a few codes.
// The payload of every read is cut into SUBSEQUENCES of OWN bits (256 for the NA12878 table,
// whose shortest code has 4 bits), one lane each; a TILE = 256 subsequences (8 KiB) = one
// workgroup.  Nothing in the two heavy kernels waits for anything: no chain, no look-back,
// no barrier besides the table load - the dependences between subsequences are settled by
// small kernels in between, over 4-byte records:
//
//   k_huf_sync   lane i runs through the RU = OWN/2 bits in front of its subsequence from
//                their first bit (a guess that is right in ~97 % of the cases once it has
//                crossed into its own subsequence), then through its own subsequence, with a
//                LENGTH-ONLY table: one LDS look-up consumes every whole code that fits in 12
//                bits (up to three 4-bit codes).  No symbol is produced.  Record of the
//                subsequence: {f = where its first code starts (as assumed), e = where the first
//                code of the next subsequence starts, c = codes that start in it}.
//                The lane's bits live in a private LDS column (dword j of lane l at j*64 + l:
//                any mix of per-lane positions is bank-conflict free, and nothing is shared,
//                so no barrier).
//   k_huf_links  link i holds iff f[i] == e[i-1].  A record behind a chain of holding links
//   k_huf_fix    that starts at the read's first subsequence is true.  Broken links (3 %) are
//                listed (leftmost of each run) and one lane per run re-decodes from the true
//                start; a fix that moves the end breaks the next link, which the next round
//                picks up: three rounds leave a handful (0.03^3).
//   k_huf_chain  one wave per read walks its records: repairs what is left serially (this
//                alone is enough for ANY table and stream - a code whose lengths share a factor
//                never synchronises - the rounds before it are only faster), sums the counts
//                per tile and settles how many values the read delivers.
//   k_huf_emit   lane i decodes its subsequence once more from its true start, now with the
//                two-symbol table, straight into the one-byte stream at its final position
//                (in-tile scan of the counts + the tile's base): 4-byte stores at byte
//                addresses, no staging.
//
// Result == huffman.c:1219 bit for bit, including its behaviour at the end of the input
// (stops when the bytes run out or the symbol count is reached; a code cut off by the end
// of the input is not delivered).

#include "press_internal.h"

namespace ph {

constexpr uint32_t HEND = 0xFFFFFFFFu; // "no further code": end of input or an undecodable prefix
constexpr uint32_t R_END = 31;         // the same in a record's 5-bit position fields
constexpr uint32_t R_FIRST = 1u << 31; // record flag: first subsequence of its read (its start is exact)
constexpr int HT = HUF_HT;

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); }

// record = f | e << 5 | c << 10 (| R_FIRST): f, e relative to the start of the own / the next
// subsequence (0 .. longest code - 1, or R_END), c <= 64
__device__ __forceinline__ uint32_t rec_pack(uint32_t f, uint32_t e, uint32_t c) { return f | (e << 5) | (c << 10); }
__device__ __forceinline__ uint32_t rec_f(uint32_t r) { return r & 31u; }
__device__ __forceinline__ uint32_t rec_e(uint32_t r) { return (r >> 5) & 31u; }
__device__ __forceinline__ uint32_t rec_c(uint32_t r) { return (r >> 10) & 127u; }

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

// A code the 12-bit tables cannot resolve: longer than HUF_LUT_BITS bits (second-level table, or
// the trie for prefixes beyond the first HUF_L2_IDS), or no code at all.  wnd = 32 stream bits.
// -> length | symbol << 8 | LC_OK, or 0 if the bits are no code.  (Rare: kept out of line.)
constexpr uint32_t LC_OK = 1u << 31;
__device__ __noinline__ uint32_t long_code(const HuffDev *hd, uint32_t wnd)
{
	const uint32_t e1 = hd->lut[wnd & ((1u << HUF_LUT_BITS) - 1u)];
	if (e1 != 0xFFFFu && !(e1 & 0x8000u)) // (callers come here for long codes only; be complete)
		return (e1 >> 8) | ((e1 & 0xFFu) << 8) | LC_OK;
	if (e1 != 0xFFFFu) {
		const uint32_t id = e1 & 0xFFu;
		const uint32_t e2 = hd->lut2[hd->l2off[id] + ((wnd >> HUF_LUT_BITS) & ((1u << hd->l2bits[id]) - 1u))];
		return e2 == 0xFFFFu ? 0u : ((e2 >> 8) | ((e2 & 0xFFu) << 8) | LC_OK);
	}
	int node = 0;
	uint32_t l = 0;
	while (node >= 0 && hd->leaf[node] < 0 && l < 32) {
		node = hd->child[node][(wnd >> l) & 1u];
		l++;
	}
	if (node < 0 || hd->leaf[node] < 0)
		return 0;
	return l | ((uint32_t) hd->leaf[node] << 8) | LC_OK;
}

// ---- the lane's bits: a private LDS column, dword j of lane l at col[j * 64] ----

// Count the codes that start in [start, lim) of the column (bit positions relative to the
// column), stopping at the payload end nb; returns where the next code starts, or HEND.
// One look-up takes every whole code that fits in 12 bits.  Two loops: while 12 bits are left in
// front of the limit nothing can step over it, so the body is a look-up and two additions, fed
// from a register window over the column (the dword after the window is fetched while the look-up
// is in flight: one LDS round trip per step); the last few codes take the careful loop.
// Wave-uniform loops with predicated bodies: lanes need different numbers of steps, and per-lane
// branches cost more than the few masked operations.
__device__ __forceinline__ uint32_t col_scan(const uint32_t *col, const uint16_t *mlut, const HuffDev *hd,
					     uint32_t start, uint32_t lim, uint32_t nb, uint32_t &cnt)
{
	bool bad = start == HEND;
	uint32_t p = bad ? 0u : start;
	uint32_t L = lim < nb ? lim : nb; // codes must START below this
	if (bad)
		L = 0;
	uint32_t c = 0;
	{
		uint32_t j = p >> 5;
		uint32_t w0 = col[j * 64], w1 = col[(j + 1) * 64];
		const uint32_t *pf = col + (j + 2) * 64;
		bool stuck = false; // a long code that needs the careful loop
		for (;;) {
			const bool act = p + HUF_LUT_BITS <= L && !stuck;
			if (!__any(act))
				break;
			const uint32_t w2 = *pf; // (columns have room for this read behind the last code's reach)
			const uint32_t wnd = __builtin_amdgcn_alignbit(w1, w0, p); // shift = p & 31
			const uint32_t e = mlut[wnd & ((1u << HUF_LUT_BITS) - 1u)];
			uint32_t tot = e & 15u, n = (e >> 4) & 15u;
			bool ok = act;
			if (__any(act && e == 0xFFFFu)) { // rare: a long code, or none
				if (act && e == 0xFFFFu) {
					const uint32_t r = long_code(hd, wnd);
					tot = r & 31u;
					n = 1;
					if (!(r & LC_OK) || p + tot > L) {
						ok = false;
						stuck = true;
					}
				}
			}
			p += ok ? tot : 0u;
			c += ok ? n : 0u;
			const uint32_t jn = p >> 5; // a step crosses at most one dword
			const bool st = jn != j;
			w0 = st ? w1 : w0;
			w1 = st ? w2 : w1;
			pf += st ? 64 : 0;
			j = jn;
		}
	}
	for (;;) {
		const bool act = p < L;
		if (!__any(act))
			break;
		const uint32_t pp = act ? p : 0u;
		const uint32_t j = pp >> 5;
		const uint32_t wnd = __builtin_amdgcn_alignbit(col[(j + 1) * 64], col[j * 64], pp);
		const uint32_t e = mlut[wnd & ((1u << HUF_LUT_BITS) - 1u)];
		uint32_t tot = e & 15u, n = (e >> 4) & 15u, len1 = (e >> 8) & 15u;
		bool fail = false;
		if (__any(act && e == 0xFFFFu)) {
			if (act && e == 0xFFFFu) {
				const uint32_t r = long_code(hd, wnd);
				fail = !(r & LC_OK);
				tot = len1 = r & 31u;
				n = 1;
			}
		}
		const bool fits = pp + tot <= L;             // every code of the group starts below L, ends inside the payload
		const bool cut = !fits && pp + len1 > nb;    // a code cut off by the end of the input is not delivered
		const bool go = act && !fail && !cut;
		p += go ? (fits ? tot : len1) : 0u;
		c += go ? (fits ? n : 1u) : 0u;
		if (act && (fail || cut)) {
			bad = true;
			L = 0;
		}
	}
	cnt = c;
	return bad ? HEND : (p >= nb && p < lim ? HEND : p);
}

// The same walk from global memory, one code per step, for the few subsequences that are decoded
// again (k_huf_fix, k_huf_chain): bits [start, lim) of a tile whose payload has nb bits from src on.
__device__ __noinline__ uint32_t slow_scan(const uint8_t *src, const HuffDev *hd, uint32_t start, uint32_t lim,
					  uint64_t nb64, uint32_t &cnt)
{
	cnt = 0;
	if (start == HEND)
		return HEND;
	const uint32_t nb = nb64 > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t) nb64;
	const uint32_t L = lim < nb ? lim : nb;
	const uint32_t nby = (nb + 7) >> 3;
	uint32_t p = start, c = 0;
	while (p < L) {
		uint64_t w = 0;
		const uint32_t b0 = p >> 3;
		for (uint32_t i = 0; i < 5 && b0 + i < nby; i++)
			w |= (uint64_t) src[b0 + i] << (8 * i);
		const uint32_t wnd = (uint32_t) (w >> (p & 7u));
		const uint32_t r = long_code(hd, wnd);
		const uint32_t len = r & 31u;
		if (!(r & LC_OK) || p + len > nb)
			return cnt = c, HEND;
		p += len;
		c++;
	}
	cnt = c;
	return p >= nb && p < lim ? HEND : p;
}

// Stage NDW dwords of the tile from byte offset rb0 (a multiple of 4, possibly negative: the
// run-up of a tile's first lane lies in the previous tile) into the lane's column.  Bytes outside
// the payload [lo, nby) read as zero.  Lanes whose dwords all lie inside issue their loads back
// to back (16-byte loads at any byte address).
template <int NDW>
__device__ __forceinline__ void col_load(uint32_t *col, const uint8_t *src, int32_t rb0, int32_t lo, int32_t nby)
{
	uint32_t v[NDW];
	if (rb0 >= lo && rb0 + 4 * NDW <= nby) {
		const uint8_t *q = src + rb0;
#pragma unroll
		for (int j = 0; j + 4 <= NDW; j += 4) {
			uint4 t;
			__builtin_memcpy(&t, q + 4 * j, 16);
			v[j] = t.x;
			v[j + 1] = t.y;
			v[j + 2] = t.z;
			v[j + 3] = t.w;
		}
#pragma unroll
		for (int j = NDW & ~3; j < NDW; j++)
			__builtin_memcpy(&v[j], q + 4 * j, 4);
	} else {
#pragma unroll
		for (int j = 0; j < NDW; j++) {
			const int32_t r = rb0 + 4 * j;
			uint32_t x = 0;
			if (r + 4 > lo && r < nby) {
				for (int i = 0; i < 4; i++)
					if (r + i >= lo && r + i < nby)
						x |= (uint32_t) src[r + i] << (8 * i);
			}
			v[j] = x;
		}
	}
#pragma unroll
	for (int j = 0; j < NDW; j++)
		col[j * 64] = v[j];
}

// ------------------------------------------------------------------ k_huf_sync

// columns: the subsequence, the reach of a code that starts in its last bit, two dwords for the
// window prefetch of col_scan
template <int RU>
struct HufGeo {
	static constexpr int OWN = 2 * RU;
	static constexpr int NDW = OWN / 32 + 1; // dwords that are loaded
	static constexpr int NCOL = NDW + 2;     // dwords a column has
};

__device__ __forceinline__ uint32_t clamp_nb(int64_t v)
{
	return v <= 0 ? 0u : (v > 0x7FFFFFFF ? 0x7FFFFFFFu : (uint32_t) v);
}

template <int RU> // run-up bits; the subsequence itself has 2 * RU
__global__ __launch_bounds__(HT) void k_huf_sync(DecodeArgs a)
{
	constexpr int OWN = HufGeo<RU>::OWN, NDW = HufGeo<RU>::NDW, NCOL = HufGeo<RU>::NCOL;
	__shared__ uint16_t mlut[1 << HUF_LUT_BITS];
	// column of lane l of wave w: img[w + 1] + l; img[0] + 63 = the last subsequence of the tile in front
	__shared__ uint32_t img[HT / 64 + 1][NCOL * 64];
	__shared__ uint8_t s_e[HT], s_list[HT];
	__shared__ uint32_t s_nl;

	const uint32_t tid = threadIdx.x;
	const uint32_t ntiles = min(uniform(a.ctl->nchunks), a.max_htiles);
	if (blockIdx.x >= ntiles)
		return;
	{
		const uint4 *s4 = reinterpret_cast<const uint4 *>(a.huff->mlut);
		uint4 *d4 = reinterpret_cast<uint4 *>(mlut);
		for (uint32_t i = tid; i < (1u << HUF_LUT_BITS) / 8; i += HT)
			d4[i] = s4[i];
	}
	uint32_t *col = img[(tid >> 6) + 1] + (tid & 63);
	const uint32_t *rcol = img[((tid + 63) >> 6)] + ((tid + 63) & 63); // the left neighbour's column
	// persistent workgroups: the table is loaded once
	for (uint32_t k = blockIdx.x; k < ntiles; k += gridDim.x) {
		const HufTile *dp = a.htiles + k;
		const uint32_t nbits_t = uniform(dp->nbits);
		const uint32_t t = uniform(dp->t_last) & 0x7FFFFFFFu;
		const uint8_t *src = a.in + dp->src;
		const int32_t nby = (int32_t) ((nbits_t + 7) >> 3); // (nbits_t < 2^32: below 2^29 bytes)
		col_load<NDW>(col, src, (int32_t) tid * (OWN / 8), 0, nby);
		if (tid == 0 && t) // the bits in front of the tile belong to the same payload
			col_load<NDW>(img[0] + 63, src, -(OWN / 8), -(OWN / 8), nby);
		if (tid == 0)
			s_nl = 0;
		__syncthreads(); // columns (a lane's run-up reads its neighbour's), the table
		// payload end in the coordinates of the own / the neighbour's column
		const uint32_t nb = clamp_nb((int64_t) nbits_t - (int64_t) tid * OWN);
		const uint32_t nbr = clamp_nb((int64_t) nbits_t - (int64_t) tid * OWN + OWN);

		// ---- run-up through the second half of the subsequence in front, then the own one
		const bool exact = t == 0 && tid == 0; // a read's first subsequence starts at its bit 0
		uint32_t f = 0, c0, c;
		if (!__all(exact)) {
			const uint32_t g = col_scan(rcol, mlut, a.huff, exact ? HEND : (uint32_t) RU, OWN, nbr, c0);
			f = g == HEND ? HEND : g - OWN;
			if (g == HEND && nb > 0)
				f = 0; // the guess ran into a bit pattern that is no code: any guess will do
			if (exact)
				f = 0;
		}
		const uint32_t e = col_scan(col, mlut, a.huff, f, OWN, nb, c);
		const uint32_t fr = f == HEND ? R_END : f, er = e == HEND ? R_END : e - OWN;

		// ---- first repair round inside the tile: a lane whose assumed start is not where its left
		// neighbour ended (3 %) is decoded again from there - by wave 0, which takes the tile's few
		// such lanes together (their columns are still in LDS).  The first lane's link leaves the
		// tile, and a repair that moves an end breaks the next link: the rounds below.
		s_e[tid] = (uint8_t) er;
		__syncthreads();
		const bool broken = tid > 0 && fr != s_e[tid - 1];
		const unsigned long long bm = __ballot(broken);
		if (bm) {
			uint32_t base = 0;
			if ((tid & 63) == 0)
				base = atomicAdd(&s_nl, (uint32_t) __popcll(bm));
			base = (uint32_t) __shfl((int) base, 0, 64);
			if (broken)
				s_list[base + (uint32_t) __popcll(bm & ((1ull << (tid & 63)) - 1ull))] = (uint8_t) tid;
		}
		if (!broken)
			a.hrec[(uint64_t) k * HT + tid] = rec_pack(fr, er, c) | (exact ? R_FIRST : 0u);
		__syncthreads();
		const uint32_t nl = s_nl;
		if (tid < 64) {
			for (uint32_t i0 = 0; i0 < nl; i0 += 64) {
				const uint32_t i = i0 + tid;
				const bool mine = i < nl;
				const uint32_t u = mine ? s_list[i] : 1u;
				const uint32_t pe = s_e[u - 1];
				uint32_t c2;
				const uint32_t e2 = col_scan(img[(u >> 6) + 1] + (u & 63), mlut, a.huff,
							     (!mine || pe == R_END) ? HEND : pe, OWN,
							     clamp_nb((int64_t) nbits_t - (int64_t) u * OWN), c2);
				if (mine)
					a.hrec[(uint64_t) k * HT + u] = rec_pack(pe, e2 == HEND ? R_END : e2 - OWN, c2);
			}
		}
		__syncthreads(); // the columns and lists are free again
	}
}

// ------------------------------------------------------------------ links, fix rounds

// leftmost broken link of every run -> a.hlist (count in ctl->ticket2)
__global__ __launch_bounds__(256) void k_huf_links(DecodeArgs a)
{
	const uint64_t nsub = (uint64_t) min(uniform(a.ctl->nchunks), a.max_htiles) * HT;
	const uint32_t lane = threadIdx.x & 63;
	for (uint64_t g0 = (uint64_t) blockIdx.x * 256; g0 < nsub; g0 += (uint64_t) gridDim.x * 256) {
		const uint64_t g = g0 + threadIdx.x;
		const uint32_t r0 = a.hrec[g];
		const uint32_t r1 = g >= 1 ? a.hrec[g - 1] : R_FIRST;
		const uint32_t r2 = g >= 2 ? a.hrec[g - 2] : R_FIRST;
		const bool br0 = !(r0 & R_FIRST) && rec_f(r0) != rec_e(r1);
		const bool br1 = !(r1 & R_FIRST) && rec_f(r1) != rec_e(r2);
		const bool lead = br0 && !br1;
		const unsigned long long m = __ballot(lead);
		if (m) {
			uint32_t base = 0;
			if (lane == 0)
				base = atomicAdd(&a.ctl->ticket2, (uint32_t) __popcll(m));
			base = (uint32_t) __shfl((int) base, 0, 64);
			const uint32_t idx = base + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
			if (lead && idx < a.hlist_cap)
				a.hlist[idx] = (uint32_t) g; // (g < 2^32: max_htiles * HT is checked on the host)
		}
	}
}

// decode subsequence g again from column-relative start `f` (0 .. / R_END); -> the new record
template <int RU>
__device__ __forceinline__ uint32_t redo_sub(const DecodeArgs &a, uint64_t g, uint32_t f)
{
	constexpr uint32_t OWN = 2 * RU;
	const HufTile *dp = a.htiles + (g / HT);
	const uint32_t tid = (uint32_t) (g % HT);
	uint32_t c = 0;
	const uint32_t e = slow_scan(a.in + dp->src, a.huff, f == R_END ? HEND : tid * OWN + f, (tid + 1) * OWN, dp->nbits, c);
	return rec_pack(f, e == HEND ? R_END : e - (tid + 1) * OWN, c);
}

// one lane per run of broken links: walk it from its left end (see the file header)
template <int RU>
__global__ __launch_bounds__(256) void k_huf_fix(DecodeArgs a)
{
	const uint64_t nsub = (uint64_t) min(a.ctl->nchunks, a.max_htiles) * HT;
	const uint32_t nl = min(a.ctl->ticket2, a.hlist_cap);
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nl; i += gridDim.x * 256) {
		uint64_t g = a.hlist[i];
		uint32_t prev_e = rec_e(a.hrec[g - 1]); // a lead's predecessor link holds: its record stands
		for (;;) {
			const uint32_t old = a.hrec[g];
			uint32_t now = old;
			if (rec_f(old) != prev_e) {
				now = redo_sub<RU>(a, g, prev_e);
				a.hrec[g] = now;
			}
			// on into g + 1 only if that link was broken when the round began (then nobody else owns it)
			if (g + 1 >= nsub)
				break;
			const uint32_t nx = a.hrec[g + 1];
			if ((nx & R_FIRST) || rec_f(nx) == rec_e(old))
				break;
			prev_e = rec_e(now);
			g++;
		}
	}
}

// ------------------------------------------------------------------ k_huf_chain

// One wave per read: what the rounds left (serially, always correct), the codes in front of every
// tile, and what the read delivers (huffman.c:1243: at most `want` values).
template <int RU>
__global__ __launch_bounds__(64) void k_huf_chain(DecodeArgs a)
{
	const uint32_t r = blockIdx.x;
	const uint32_t lane = threadIdx.x;
	const uint32_t k0 = uniform(a.hread[2 * r]), nt = uniform(a.hread[2 * r + 1]);
	if (!nt)
		return;
	const uint64_t g0 = (uint64_t) k0 * HT;
	const uint32_t nsub = nt * HT;
	uint64_t cum = 0;
	uint32_t carry_e = 0; // end of the subsequence in front of the block
	for (uint32_t b = 0; b < nsub; b += 64) {
		uint32_t rec = a.hrec[g0 + b + lane];
		for (;;) { // until every link of the block holds
			uint32_t pe = (uint32_t) __shfl_up((int) rec_e(rec), 1, 64);
			if (lane == 0)
				pe = carry_e;
			const bool br = !(rec & R_FIRST) && rec_f(rec) != pe;
			const unsigned long long m = __ballot(br);
			if (!m)
				break;
			const uint32_t L = (uint32_t) __builtin_ctzll(m);
			// everything left of lane L is true, so is its predecessor's end: decode it again (all lanes
			// compute the same thing - this path is rare)
			const uint32_t st = uniform((uint32_t) __shfl((int) pe, (int) L, 64));
			const uint32_t now = redo_sub<RU>(a, g0 + b + L, st);
			if (lane == L) {
				rec = now;
				a.hrec[g0 + b + lane] = now;
			}
		}
		carry_e = uniform((uint32_t) __shfl((int) rec_e(rec), 63, 64));
		if ((b % HT) == 0 && lane == 0)
			a.htbase[k0 + b / HT] = (uint32_t) (cum > 0xFFFFFFFFull ? 0xFFFFFFFFull : cum);
		const uint32_t inc = wave_scan(rec_c(rec));
		cum += uniform((uint32_t) __shfl((int) inc, 63, 64));
	}
	if (lane == 0) {
		const uint32_t want = a.htiles[k0].want;
		a.meta[r].nlow = cum < want ? (uint32_t) cum : want; // what huffman_decode_memory delivered
	}
}

// ------------------------------------------------------------------ k_huf_emit

// make the LDS writes of this wave visible to its other lanes (DS ops of a wave execute in
// order; this only stops the compiler from moving them)
__device__ __forceinline__ void wave_lds_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr uint32_t EMIT_STG = 3328; // staging bytes per wave: 52 per lane (NA12878: 47.4 on average, 64 at most)

template <int RU>
__global__ __launch_bounds__(HT) void k_huf_emit(DecodeArgs a)
{
	constexpr int OWN = HufGeo<RU>::OWN, NDW = HufGeo<RU>::NDW, NCOL = NDW + 1;
	__shared__ uint32_t lut[1 << HUF_LUT_BITS];
	__shared__ uint32_t img[HT / 64][NCOL * 64];
	__shared__ __attribute__((aligned(16))) uint8_t stg_all[HT / 64][EMIT_STG];
	__shared__ uint32_t wtot[2][HT / 64];

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	const uint32_t ntiles = min(uniform(a.ctl->nchunks), a.max_htiles);
	if (blockIdx.x >= ntiles)
		return;
	{
		const uint4 *s4 = reinterpret_cast<const uint4 *>(a.huff->lut32);
		uint4 *d4 = reinterpret_cast<uint4 *>(lut);
		for (uint32_t i = tid; i < (1u << HUF_LUT_BITS) / 4; i += HT)
			d4[i] = s4[i];
	}
	uint32_t *col = img[tid >> 6] + lane;
	uint8_t *stg = stg_all[tid >> 6];
	uint32_t par = 0;
	for (uint32_t k = blockIdx.x; k < ntiles; k += gridDim.x, par ^= 1u) {
		const HufTile *dp = a.htiles + k;
		const uint32_t nbits_t = uniform(dp->nbits);
		const uint32_t want = uniform(dp->want);
		const uint8_t *src = a.in + dp->src;
		uint8_t *low = a.low + dp->low;
		col_load<NDW>(col, src, (int32_t) tid * (OWN / 8), 0, (int32_t) ((nbits_t + 7) >> 3));
		const uint32_t nb = clamp_nb((int64_t) nbits_t - (int64_t) tid * OWN);
		const uint32_t rec = a.hrec[(uint64_t) k * HT + tid];
		const uint32_t cnt = rec_c(rec);
		const uint32_t inc = wave_scan(cnt);
		if (lane == 63)
			wtot[par][tid >> 6] = inc;
		__syncthreads(); // the wave totals (and, the first time, the table); columns and staging are private
		uint64_t obase = uniform(a.htbase[k]); // codes of the read in front of this wave
#pragma unroll
		for (int w2 = 0; w2 < HT / 64; w2++)
			if (w2 < (int) (tid >> 6))
				obase += uniform(wtot[par][w2]);
		const uint32_t wsum = uniform((uint32_t) __shfl((int) inc, 63, 64));
		// the wave delivers values [obase, obase + wsum) of the read, cut at `want`
		const uint32_t quota = obase >= want ? 0u : (wsum < want - (uint32_t) obase ? wsum : want - (uint32_t) obase);
		const uint32_t ex = inc - cnt; // codes of the wave in front of this lane
		const uint32_t nmine = ex >= quota ? 0u : (cnt < quota - ex ? cnt : quota - ex);
		uint8_t *dst = low + obase;
		// symbols go to the wave's staging buffer at their final order and leave it with 16-byte stores;
		// a wave that holds more codes than the buffer takes (cannot happen with 5.4-bit codes on
		// average) stores them byte by byte instead
		const bool staged = wsum <= EMIT_STG;
		uint8_t *wp = staged ? stg + ex : dst + ex;

		// decode again, now with the symbols: sym1 | sym2 << 8 | len1 << 16 | (len1 + len2) << 21 | codes << 26
		const uint32_t f = rec_f(rec);
		uint32_t p = f == R_END ? 0u : f;
		uint32_t L = (uint32_t) OWN < nb ? (uint32_t) OWN : nb;
		if (f == R_END || nmine == 0)
			L = 0;
		uint32_t q = 0; // symbols written
		if (staged) {
			uint32_t j = p >> 5;
			uint32_t w0 = col[j * 64], w1 = col[(j + 1) * 64];
			const uint32_t *pf = col + (j + 2) * 64;
			bool stuck = false;
			for (;;) { // both codes of a look-up start below L and the quota has room for two
				const bool act = p + HUF_LUT_BITS <= L && q + 2 <= nmine && !stuck;
				if (!__any(act))
					break;
				const uint32_t pfj = j + 2 < (uint32_t) NCOL ? 0u : 64u; // (stay inside the column)
				const uint32_t w2 = *(pf - pfj);
				const uint32_t wnd = __builtin_amdgcn_alignbit(w1, w0, p);
				uint32_t e = lut[wnd & ((1u << HUF_LUT_BITS) - 1u)];
				bool ok = act;
				if (__any(act && e >= HUF_LONG)) { // rare: a code longer than 12 bits
					if (act && e >= HUF_LONG) {
						const uint32_t r = long_code(a.huff, wnd);
						const uint32_t len = r & 31u;
						e = ((r >> 8) & 0xFFu) | (len << 16) | (len << 21) | (1u << 26);
						if (!(r & LC_OK) || p + len > L) {
							ok = false;
							stuck = true;
						}
					}
				}
				const bool two = (e >> 27) != 0;
				if (ok) {
					wp[q] = (uint8_t) e;
					if (two)
						wp[q + 1] = (uint8_t) (e >> 8);
				}
				p += ok ? ((e >> 21) & 0x1Fu) : 0u;
				q += ok ? (two ? 2u : 1u) : 0u;
				const uint32_t jn = p >> 5;
				const bool st = jn != j;
				w0 = st ? w1 : w0;
				w1 = st ? w2 : w1;
				pf += st ? 64 : 0;
				j = jn;
			}
		}
		for (;;) { // the careful loop: the last codes of the subsequence / of the quota
			const bool act = p < L && q < nmine;
			if (!__any(act))
				break;
			const uint32_t pp = act ? p : 0u;
			const uint32_t j = pp >> 5;
			const uint32_t wnd = __builtin_amdgcn_alignbit(col[(j + 1) * 64], col[j * 64], pp);
			uint32_t e = lut[wnd & ((1u << HUF_LUT_BITS) - 1u)];
			if (__any(act && e >= HUF_LONG)) { // (a pattern that is no code cannot come up: k_huf_sync
				if (act && e >= HUF_LONG) {    // stopped counting in front of it)
					const uint32_t r = long_code(a.huff, wnd);
					const uint32_t len = r & 31u;
					e = ((r >> 8) & 0xFFu) | (len << 16) | (len << 21) | (1u << 26);
				}
			}
			const uint32_t len1 = (e >> 16) & 0x1Fu;
			// the second code counts only if it starts inside and the quota has room
			const bool two = (e >> 27) && pp + len1 < L && q + 2 <= nmine;
			if (act) {
				wp[q] = (uint8_t) e;
				if (two)
					wp[q + 1] = (uint8_t) (e >> 8);
			}
			p += act ? (two ? ((e >> 21) & 0x1Fu) : len1) : 0u;
			q += act ? (two ? 2u : 1u) : 0u;
		}
		if (staged) {
			wave_lds_sync();
			for (uint32_t o = lane * 16; o < quota; o += 64 * 16) {
				const uint4 v = *reinterpret_cast<const uint4 *>(stg + o);
				if (o + 16 <= quota) {
					__builtin_memcpy(dst + o, &v, 16); // any byte address
				} else {
					const uint32_t vv[4] = { v.x, v.y, v.z, v.w };
					for (uint32_t b = 0; o + b < quota; b++)
						dst[o + b] = (uint8_t) (vv[b >> 2] >> (8 * (b & 3)));
				}
			}
			wave_lds_sync(); // staging is free again
		}
	}
}

// ------------------------------------------------------------------ tiles

// Tiles of every read (one thread per read; runs after k_ex_parse): ids of one read are
// consecutive.  hread[2r] = first tile, hread[2r + 1] = number of tiles.
__global__ __launch_bounds__(256) void k_huff_tiles(DecodeArgs a)
{
	const uint32_t r = blockIdx.x * 256 + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t minlen = a.huff->minlen, maxlen = a.huff->maxlen;
	const uint32_t OWN = minlen >= 4 ? 256u : minlen >= 2 ? 128u : 64u;
	const uint64_t TB = (uint64_t) HT * OWN;
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
	if (r < a.nreads) {
		if (base + nt > a.max_htiles) // cannot happen: max_htiles is the same bound summed over the slots
			nt = base < a.max_htiles ? a.max_htiles - base : 0;
		a.hread[2 * r] = base;
		a.hread[2 * r + 1] = nt;
	}
	if (nt) {
		const ReadMeta *m = a.meta + r;
		const uint32_t hdr = m->hdr + m->seclen + 4;
		const uint64_t nbytes64 = a.in_len[r] - hdr;
		const uint64_t nbits = 8ull * (nbytes64 > 0x1FFFFFFFull ? 0x1FFFFFFFull : nbytes64);
		for (uint32_t t = 0; t < nt; t++) {
			HufTile d;
			d.src = a.in_off[r] + hdr + t * (TB / 8);
			d.low = a.off[r];
			d.nbits = (uint32_t) (nbits - t * TB);
			d.t_last = t | (t + 1 == nt ? 0x80000000u : 0u);
			d.read = r;
			d.want = m->nlow;
			a.htiles[base + t] = d;
		}
	}
}

template <int RU>
static void run_huff_decode(const DecodeArgs &a, hipStream_t s)
{
	// persistent workgroups (the tables are loaded once per workgroup): what is resident, 7 / 4 per CU
	const uint32_t nt = a.max_htiles ? a.max_htiles : 1;
	const uint32_t gs = nt < 7u * 256u ? nt : 7u * 256u, ge = nt < 4u * 256u ? nt : 4u * 256u;
	hipLaunchKernelGGL((k_huf_sync<RU>), dim3(gs), dim3(HT), 0, s, a);
	for (int round = 1; round < HUF_FIX_ROUNDS; round++) { // (round 0 ran inside k_huf_sync)
		(void) hipMemsetAsync(&a.ctl->ticket2, 0, 4, s);
		hipLaunchKernelGGL(k_huf_links, dim3(2048), dim3(256), 0, s, a);
		hipLaunchKernelGGL((k_huf_fix<RU>), dim3(1024), dim3(256), 0, s, a);
	}
	hipLaunchKernelGGL((k_huf_chain<RU>), dim3(a.nreads), dim3(64), 0, s, a);
	hipLaunchKernelGGL((k_huf_emit<RU>), dim3(ge), dim3(HT), 0, s, a);
}

// Huffman stage of the exception-split decoders: payload of every read -> a.low
void launch_huff_decode(const DecodeArgs &a, uint32_t minlen, hipStream_t s)
{
	(void) hipMemsetAsync(a.ctl, 0, sizeof(ChunkCtl), s);
	hipLaunchKernelGGL(k_huff_tiles, dim3((a.nreads + 255) / 256), dim3(256), 0, s, a);
	if (minlen >= 4)
		run_huff_decode<128>(a, s);
	else if (minlen >= 2)
		run_huff_decode<64>(a, s);
	else
		run_huff_decode<32>(a, s);
}

} // namespace ph
