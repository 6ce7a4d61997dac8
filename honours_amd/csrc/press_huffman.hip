// press_huffman.hip - static-Huffman stream decode for gfx950 (huffman.c:1219 shuffman_decode_memory).
//
// The stream has no synchronisation points, but Huffman codes self-synchronise: a decoder
// started at a wrong bit position falls back onto true code boundaries after a few codes.
// The payload of every read is cut into SUBSEQUENCES of OWN bits (256 for the NA12878 table,
// whose shortest code has 4 bits), one lane each; a TILE = 256 subsequences (8 KiB) is the unit the records are
// kept by, a quarter of it (64 subsequences) what a wave takes at a time.  In the two heavy kernels the waves of a
// workgroup share the tables and nothing else - no barrier, no wave waits for another:
//
//   k_huf_sync   where do codes start?  Lengths (and the sample deltas the symbols stand for - no symbols): one
//                LDS look-up takes every whole code that fits in 12 bits, and its entry is the increment of ONE
//                accumulator (position, codes, delta sum).  Lane i first runs through the last RU = OWN/2 bits of
//                its OWN subsequence: where that run crosses into the next subsequence is lane i + 1's guess of its
//                first code's start (right in ~97 % of the cases; handed over by a lane shuffle); then through its
//                own subsequence from its guess.  Leaves a record per subsequence {start, codes, sum of their
//                deltas}, where it ended, per wave of a tile the totals, and ONE 64-bit word per unit: the lanes
//                whose guess is not where the left neighbour ended (~3 %) and the unit's first lane (no guess: it
//                starts at bit 0).  Nothing is repaired here.
//                The lane's bits live in a private LDS column (dword j of lane l at j*64 + l: any mix of per-lane
//                positions is bank-conflict free).  The lean loops are divergent while-loops (a lane that is through
//                leaves); while 12 bits are left in front of the limit nothing can step over it, so the body is one
//                look-up fed from a register window over the column (the dword behind the window is fetched while
//                the look-up is in flight); the last few codes take a careful loop.
//   k_huf_list   the words of k_huf_sync -> the dense list of k_huf_fix's first round.
//   k_huf_fix    the listed subsequences, 64 to a wave - dense, whatever tile they came from: an entry whose guess
//                was right is dropped; the others are decoded again from where the subsequence in front ended; the
//                wave totals take the difference (one 64-bit atomic); if the subsequence's own end moved, its right
//                neighbour goes on the next round's list.  Four launches; an ordinary table is through after two
//                (1 150 000 -> 30 000 -> 900 -> 20 entries).
//   k_huf_serial what is listed after that (a code whose lengths share a factor never synchronises): one wave
//                per such read walks it serially from the first unsettled subsequence - slow, enough for ANY
//                table and stream; the rounds before it are only faster.
//   k_huf_chain  one workgroup per read: the codes in front of every tile (scan of the wave totals), the sample
//                value there, what the read delivers, and whether k_huf_emit can write its samples.
//   k_huf_emit   lane i decodes its subsequence once more from its true start, now with the two-symbol
//                table, into the wave's LDS staging buffer at its final order (scan of the counts); the
//                count k_huf_sync found ends the loop.  Every wave is on its own (no barrier in the loop), and
//                its loop is a pipeline two units deep: the next unit's payload and records are on their way
//                while this one is decoded.
//                The one-byte values do not leave the chip: k_huf_sync also summed the sample deltas they
//                stand for (per wave, from the same look-up), k_huf_chain made those the sample
//                value in front of every tile, so the wave turns its staging buffer into samples itself
//                (zig-zag, running sum, the few exceptions merged in - trans.c:260) and stores int16.
//                Reads whose lists do not interleave cleanly (a stream that delivers fewer values than
//                its exceptions assume) keep the two-step way: values to DecodeArgs::low, then
//                k_low_decode_chunked.
//
// Result == huffman.c:1219 bit for bit, including its behaviour at the end of the input
// (stops when the bytes run out or the symbol count is reached; a code cut off by the end
// of the input is not delivered).

#include "press_internal.h"
#include "press_packed.h"
#include <cstddef>
#include <type_traits>

namespace ph {

#ifdef HUF_STAMPS
// diagnostic build only (tools/hufstamps.py): where a wave of k_huf_sync / k_huf_emit spends its time - s_memtime
// differences of lane 0, summed per phase over all units
__device__ unsigned long long g_hstamp[32];
#define HSTAMP_DECL                                             \
	unsigned long long hs_t = __builtin_amdgcn_s_memtime(); \
	unsigned long long hs_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }
#define HSTAMP(i)                                                              \
	do { /* (per wave, in scalar registers; flushed once at the end) */    \
		__builtin_amdgcn_sched_barrier(0);                             \
		const unsigned long long hs_n = __builtin_amdgcn_s_memtime(); \
		hs_acc[(i) & 7] += hs_n - hs_t;                                \
		hs_t = hs_n;                                                   \
		__builtin_amdgcn_sched_barrier(0);                             \
	} while (0)
#define HSTAMP_FLUSH(base)                                                     \
	do {                                                                   \
		if ((threadIdx.x & 63) == 0)                                   \
			for (int hs_i = 0; hs_i < 8; hs_i++)                   \
				atomicAdd(&g_hstamp[(base) + hs_i], hs_acc[hs_i]); \
	} while (0)
#else
#define HSTAMP_DECL
#define HSTAMP(i)
#define HSTAMP_FLUSH(base)
#endif

constexpr uint32_t HEND = 0xFFFFFFFFu; // "no further code": end of input or an undecodable prefix
constexpr uint32_t R_END = 31;         // the same in a record's position fields
constexpr int HT = HUF_HT;

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); }
__device__ __forceinline__ bool any64(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0; }

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

// A code beyond the second-level tables (prefixes past the first HUF_L2_IDS, tables that do not
// fit): the trie in global memory.  wnd = 32 stream bits.  -> a one-code lut32 entry, or 0 if the
// bits are no code.  (Very rare: kept out of line.)
__device__ __noinline__ uint32_t trie_code(const HuffDev *hd, uint32_t wnd)
{
	int node = 0;
	uint32_t l = 0;
	while (node >= 0 && hd->leaf[node] < 0 && l < 32) {
		node = hd->child[node][(wnd >> l) & 1u];
		l++;
	}
	if (node < 0 || hd->leaf[node] < 0)
		return 0;
	return (uint32_t) hd->leaf[node] | (l << 8) | (l << 24);
}

// columns: the subsequence and the reach of a code that starts in its last bit, one dword for the
// window prefetch of the scan loops
template <int RU>
struct HufGeo {
	static constexpr int OWN = 2 * RU;
	static constexpr int NDW = OWN / 32 + 1; // dwords that are loaded
	static constexpr int NCOL = NDW + 1;     // dwords a column has
};

__device__ __forceinline__ uint32_t clamp_nb(int64_t v)
{
	return v <= 0 ? 0u : (v > 0x7FFFFFFF ? 0x7FFFFFFFu : (uint32_t) v);
}

// 16 / 4 payload bytes at any byte address; the payload is streamed (k_huf_sync and k_huf_emit each read it once, far
// apart): non-temporal policy (tools/ubench_stream.hip)
#ifndef HUF_NO_NT
__device__ __forceinline__ uint4 ld16_pay(const uint8_t *p)
{
	typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
	const u32x4_u v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_u *>(p));
	return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t ld4_pay(const uint8_t *p)
{
	typedef uint32_t __attribute__((aligned(1))) u32_u;
	return __builtin_nontemporal_load(reinterpret_cast<const u32_u *>(p));
}
#else
__device__ __forceinline__ uint4 ld16_pay(const uint8_t *p)
{
	uint4 v;
	__builtin_memcpy(&v, p, 16);
	return v;
}
__device__ __forceinline__ uint32_t ld4_pay(const uint8_t *p)
{
	uint32_t v;
	__builtin_memcpy(&v, p, 4);
	return v;
}
#endif

// Stage NDW dwords of the tile from byte offset rb0 (a multiple of 4, possibly negative: the
// run-up of a tile's first lane lies in the previous tile) into a column.  Bytes outside
// the payload [lo, nby) read as zero.  Lanes whose dwords all lie inside issue their loads back
// to back (16-byte loads at any byte address).
template <int NDW, int S = 64>
__device__ __forceinline__ void col_load(uint32_t *col, const uint8_t *src, int32_t rb0, int32_t lo, int32_t nby)
{
	if (rb0 >= lo && rb0 + 4 * NDW <= nby) {
		const uint8_t *q = src + rb0;
		uint4 t[NDW / 4 ? NDW / 4 : 1];
		uint32_t x[NDW % 4 ? NDW % 4 : 1];
#pragma unroll
		for (int j = 0; j < NDW / 4; j++)
			t[j] = ld16_pay(q + 16 * j);
#pragma unroll
		for (int j = 0; j < NDW % 4; j++)
			x[j] = ld4_pay(q + 4 * (NDW / 4 * 4 + j));
#pragma unroll
		for (int j = 0; j < NDW / 4; j++) {
			col[(4 * j + 0) * S] = t[j].x;
			col[(4 * j + 1) * S] = t[j].y;
			col[(4 * j + 2) * S] = t[j].z;
			col[(4 * j + 3) * S] = t[j].w;
		}
#pragma unroll
		for (int j = 0; j < NDW % 4; j++)
			col[(NDW / 4 * 4 + j) * S] = x[j];
	} else if (nby > lo) {
		// the ends of a payload: byte by byte, but every load unconditional (at an address clamped into the payload)
		// and all of them in flight together - as conditional loads each waited for the one before it: 36 memory
		// round trips in a row for the whole wave whenever one of its lanes held a payload's first or last bytes
		uint8_t b[4 * NDW];
#pragma unroll
		for (int i = 0; i < 4 * NDW; i++) {
			const int32_t r = rb0 + i;
			b[i] = src[r < lo ? lo : (r >= nby ? nby - 1 : r)];
		}
#pragma unroll
		for (int j = 0; j < NDW; j++) {
			uint32_t x = 0;
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const int32_t r = rb0 + 4 * j + i;
				x |= (r >= lo && r < nby) ? (uint32_t) b[4 * j + i] << (8 * i) : 0u;
			}
			col[j * S] = x;
		}
	} else {
#pragma unroll
		for (int j = 0; j < NDW; j++)
			col[j * S] = 0;
	}
}

// the delta a one-byte value stands for (zig-zag undone)
__device__ __forceinline__ int32_t unzz8(uint32_t z) { return (int32_t) (z >> 1) ^ -(int32_t) (z & 1u); }

// LDS through 32-bit pointers: address arithmetic stays in 32 bits (generic pointers made the compiler compute every
// row address with a 64-bit multiply-add, a quarter-rate instruction in the innermost loop)
typedef __attribute__((address_space(3))) const uint32_t *lds_cu32p;
typedef __attribute__((address_space(3))) const uint16_t *lds_cu16p;
typedef __attribute__((address_space(3))) const uint8_t *lds_cu8p;
typedef __attribute__((address_space(3))) uint8_t *lds_u8p;

// the tables of the length scans, in LDS (k_huf_sync, k_huf_fix, k_huf_serial)
struct LenTabs {
	alignas(16) uint32_t alut[1 << HUF_LUT_BITS];
	alignas(16) uint16_t flut[1 << HUF_LUT_BITS];
	alignas(16) uint16_t l2ld[HUF_L2_ENTRIES];
};

// Count the codes that start in [start, lim) of a column (bit positions relative to the column),
// stopping at the payload end nb, and sum the deltas their symbols stand for (dsum, mod 2^16);
// returns where the next code starts, or HEND.
// The lean loop keeps position, count and sum in ONE register, A = (position + 30) | codes << 9 | sum << 16, and
// the table entry (HuffDev::alut) is the increment of all three for every whole code that fits in 12 bits: a step is
// position test, row address, the two dwords around the position (read from the column at every step - no register
// window to shift), funnel shift, table look-up, one add, the long-code test: 8 vector instructions (the version
// with separate counters, field extraction and a register window over the column took 19).  The position is kept 30
// bits ahead so that the funnel shift leaves the 12 window bits at bits 2 .. 13: masked, they are the table's byte
// offset.  Long codes (13 .. 24 bits: one code in 200 of the NA12878 table, one look-up in four has such a lane)
// through the second-level table; the last codes in front of the limit take the careful loop (first code alone:
// HuffDev::flut).
template <bool TRIE, int S = 64>
__device__ __forceinline__ uint32_t len_scan(const uint32_t *col_, const LenTabs *tabs, const HuffDev *hd, uint32_t start,
					     uint32_t lim, uint32_t nb, uint32_t &cnt, uint32_t &dsum)
{
	const lds_cu32p col = (lds_cu32p) col_;
	const lds_cu32p alut = (lds_cu32p) tabs->alut;
	const lds_cu16p flut = (lds_cu16p) tabs->flut;
	const lds_cu16p l2ld = (lds_cu16p) tabs->l2ld;
	bool bad = start == HEND;
	uint32_t L = lim < nb ? lim : nb; // codes must START below this
	if (bad)
		L = 0;
	// a long code: bits | delta << 8; 0: the bits are no code.  wnd: the stream bits from the code's first on
	// (TRIE: only for tables with codes the second-level tables do not hold - the host knows (HuffDev::needs_trie);
	// without it no entry is 0xFFFFFFFF: prefixes that are no code point at a second-level slot that says so)
	auto long_ld = [&](uint32_t e, uint32_t wnd) -> uint32_t {
		if (TRIE && e == 0xFFFFFFFFu) { // beyond the second level: the trie in global memory
			const uint32_t t = trie_code(hd, wnd);
			return t ? (((t >> 24) & 31u) | (((uint32_t) unzz8(t & 0xFFu) & 0xFFu) << 8)) : 0u;
		}
		const uint32_t v = l2ld[(e & 0xFFFu) + ((wnd >> HUF_LUT_BITS) & ((1u << ((e >> 12) & 15u)) - 1u))];
		return v == 0xFFFFu ? 0u : v;
	};
	constexpr uint32_t PB = 30; // the position's bias
	const lds_cu32p colm = col - S;
	uint32_t A = (bad ? 0u : start) + PB;
	{
		int32_t LmP = (int32_t) L - HUF_LUT_BITS + (int32_t) PB; // every code of a look-up starts below L while p <= L - 12
		// the two dwords around the (biased) position stay in registers; the dword behind them is fetched while the
		// look-up is in flight (one LDS round trip per step on the chain of dependent operations, not two)
		uint32_t jj = (A >> 5) & 15u;
		uint32_t w0 = colm[__umul24(jj, (uint32_t) S)], w1 = colm[__umul24(jj, (uint32_t) S) + S];
		// (a divergent loop: a lane that is through leaves it - the lanes still in it are the execution mask, which
		// only shrinks; written as "all lanes loop while any is active, the body under a condition" the compiler
		// spent a dozen scalar instructions per step on masks)
		while ((int32_t) (A & 0x1FFu) <= LmP) {
			const uint32_t pp = A & 0x1FFu;
			{
				const uint32_t w2 = colm[__umul24(jj, (uint32_t) S) + 2 * S];
				const uint32_t wnd = __builtin_amdgcn_alignbit(w1, w0, A); // shift = pp & 31; window bits at 2 .. 13
				const uint32_t e = *(lds_cu32p) ((lds_cu8p) alut + (wnd & (((1u << HUF_LUT_BITS) - 1u) << 2)));
				uint32_t inc = e;
				if ((int32_t) e < 0) { // rare (one code in 200): a long code, or none
					const uint32_t ld = long_ld(e, wnd >> 2);
					const uint32_t tot = ld & 0xFFu;
					inc = tot | (1u << HUF_A_CNT) | (((uint32_t) (int32_t) (int8_t) (ld >> 8) & 0x7FFFu) << HUF_A_SUM);
					if (tot - 1u >= L + PB - pp) { // no code (tot 0), or it ends behind the limit (tot > L - p):
						inc = 0;               // the careful loop decides
						LmP = -1;
					}
				}
				A += inc;
				const uint32_t jn = (A >> 5) & 15u; // a step crosses at most one dword
				if (jn != jj) {
					w0 = w1;
					w1 = w2;
				}
				jj = jn;
			}
		}
	}
	uint32_t p = (A & 0x1FFu) - PB;
	uint32_t c = (A >> HUF_A_CNT) & 0x7Fu;
	int32_t d = (int32_t) (A << 1) >> 17; // bits 16 .. 30, signed
	for (;;) {
		const bool act = p < L;
		if (!any64(act))
			break;
		const uint32_t pp = act ? p : 0u;
		const uint32_t j = pp >> 5;
		const int r0 = (int) __umul24(j, (uint32_t) S);
		const uint32_t wnd = __builtin_amdgcn_alignbit(col[r0 + S], col[r0], pp);
		const uint32_t e = alut[wnd & ((1u << HUF_LUT_BITS) - 1u)];
		const uint32_t f = flut[wnd & ((1u << HUF_LUT_BITS) - 1u)];
		uint32_t tot = e & 15u, n = (e >> HUF_A_CNT) & 15u, len1 = f & 0xFFu;
		int32_t dall = (int32_t) (e << 1) >> 17, d1 = (int32_t) (int8_t) (f >> 8);
		bool fail = false;
		if (any64(act && (int32_t) e < 0)) {
			if (act && (int32_t) e < 0) {
				const uint32_t ld = long_ld(e, wnd);
				tot = len1 = ld & 0xFFu;
				dall = d1 = (int32_t) (int8_t) (ld >> 8);
				fail = tot == 0;
				n = 1;
			}
		}
		const bool fits = pp + tot <= L;          // every code of the group starts below L, ends inside the payload
		const bool cut = !fits && pp + len1 > nb; // a code cut off by the end of the input is not delivered
		if (act && !fail && !cut) {
			p += fits ? tot : len1;
			c += fits ? n : 1u;
			d += fits ? dall : d1;
		}
		if (act && (fail || cut)) {
			bad = true;
			L = 0;
		}
	}
	cnt = c;
	dsum = (uint32_t) d & 0xFFFFu;
	return bad ? HEND : (p >= nb && p < lim ? HEND : p);
}

// make the LDS writes of this wave visible to its other lanes (DS ops of a wave execute in
// order; this only stops the compiler from moving them)
__device__ __forceinline__ void wave_lds_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------ one tile

// ------------------------------------------------------------------ first pass

// Sixteen waves share the tables and nothing else: a wave takes a quarter of a tile (64 subsequences) per ticket and
// never meets the workgroup's other waves at a barrier, so that the waves of a CU are in different phases - loads,
// run-up, own pass, records - at any time (with the four barriers per round of the tile-per-256-threads version the
// sixteen waves of a workgroup waited for the round's loads together: 0.82 ms for 0.62 ms of instruction issue).
#ifndef HUF_SYNC_PER_CU
#define HUF_SYNC_PER_CU 2
#endif
// the run-up in eighths of a subsequence (4: the half in front - 128 bits of NA12878's 256).  k_huf_sync's cost goes
// with run-up + subsequence; a shorter run-up leaves more wrong guesses to k_huf_fix's dense rounds.
#ifndef HUF_RUNUP_EIGHTHS
#define HUF_RUNUP_EIGHTHS 4
#endif
constexpr int WGS = 1024;
constexpr int SYNC_S = 64; // dwords per row of a wave's image (a lane's column in its own bank: no conflict whatever rows the lanes are at)

// what k_huf_sync / k_huf_fix leave per subsequence: a.hrec = start | codes << 8 | sum of their deltas << 16,
// a.hend = where the next subsequence's first code starts (0 .. 30, R_END: none)
__device__ __forceinline__ uint32_t pack_rec(uint32_t f, uint32_t c, uint32_t d)
{
	return (f == HEND ? R_END : f) | ((c & 0xFFu) << 8) | ((d & 0xFFFFu) << 16);
}

constexpr uint32_t LIST_NONE = 0xFFFFFFFFu; // an unused list slot
constexpr uint32_t SYNC_UC = 16;            // quarter tiles a workgroup takes from the global counter at a time

// append the lanes with `yes` to the list whose counter is *cnt (one atomic per wave)
__device__ __forceinline__ void list_push(uint32_t *list, uint32_t *cnt, uint32_t cap, bool yes, uint32_t value, uint32_t lane)
{
	const unsigned long long m = __ballot(yes);
	if (!m)
		return;
	uint32_t base = 0;
	if (lane == (uint32_t) __builtin_ctzll(m))
		base = atomicAdd(cnt, (uint32_t) __popcll(m));
	base = (uint32_t) __shfl((int) base, __builtin_ctzll(m), 64);
	const uint32_t idx = base + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
	if (yes && idx < cap)
		list[idx] = value;
}

__device__ __forceinline__ void load_len_tables(const HuffDev *hd, LenTabs *t, uint32_t nthr)
{
	const uint4 *s4 = reinterpret_cast<const uint4 *>(hd->alut);
	uint4 *d4 = reinterpret_cast<uint4 *>(t->alut);
	for (uint32_t i = threadIdx.x; i < (1u << HUF_LUT_BITS) / 4; i += nthr)
		d4[i] = s4[i];
	const uint4 *f4 = reinterpret_cast<const uint4 *>(hd->flut);
	uint4 *g4 = reinterpret_cast<uint4 *>(t->flut);
	for (uint32_t i = threadIdx.x; i < (1u << HUF_LUT_BITS) / 8; i += nthr)
		g4[i] = f4[i];
	const uint4 *t4 = reinterpret_cast<const uint4 *>(hd->l2ld);
	uint4 *u4 = reinterpret_cast<uint4 *>(t->l2ld);
	for (uint32_t i = threadIdx.x; i < (uint32_t) HUF_L2_ENTRIES / 8; i += nthr)
		u4[i] = t4[i];
}


// All tiles, first pass.  Persistent workgroups (the tables are loaded once); a wave's unit of work is a quarter of
// a tile: every lane runs up through the half subsequence in front of its own - where that crosses into its
// subsequence is its guess of the first code's start - and then through its own.  Lanes whose guess is not where the
// left neighbour ended go on the list of k_huf_fix, and so does every unit's first lane, unchecked: where the
// subsequence in front of it ended another wave knows (k_huf_fix drops the entry if the guess was right).  Nothing
// is repaired here.  Units are drawn from a ticket (as in k_huf_emit); which lanes are to be looked at again leaves the
// kernel as one 64-bit word per unit (DecodeArgs::hbits), from which k_huf_list makes k_huf_fix's list.
template <int RU, bool TRIE>
__global__ __launch_bounds__(WGS, 2 * HUF_SYNC_PER_CU * 2) void k_huf_sync(DecodeArgs a)
{
	constexpr int OWN = HufGeo<RU>::OWN, NDW = HufGeo<RU>::NDW;
	constexpr int RUNUP = OWN * HUF_RUNUP_EIGHTHS / 8; // bits of the run-up (the last RUNUP bits of the subsequence in front)
	constexpr int S = SYNC_S;
	// the tables, then a wave's image: row j = dword j of [the subsequence in front, lane 0 .. lane 63]; two rows behind
	// the last (len_scan reads the row behind a position's, and a code may reach 23 bits past the subsequence).  One
	// struct: len_scan also reads the row IN FRONT of a position's (the biased position) - for row 0 that is the
	// image in front, for the first image the tables: anything, as long as it is the workgroup's LDS
	struct Lds {
		LenTabs tabs;
		uint32_t imgs[WGS / 64][(NDW + 2) * S];
	};
	__shared__ __attribute__((aligned(16))) Lds lds;
	__shared__ unsigned long long s_chunk[64];
	__shared__ uint32_t s_ticket;
	auto &imgs = lds.imgs;

	const uint32_t n = min(uniform(a.ctl->nchunks), a.max_htiles);
	const uint32_t nunits = n * (HT / 64);
	const uint32_t lane = threadIdx.x & 63;
	if (threadIdx.x == 0)
		s_ticket = 0;
	if (threadIdx.x < 64)
		s_chunk[threadIdx.x] = ~0ull;
	load_len_tables(a.huff, &lds.tabs, WGS);
	__syncthreads(); // the tables: the only barrier
	uint32_t *img = imgs[threadIdx.x >> 6];
	for (uint32_t i = lane; i < 2 * S; i += 64)
		img[NDW * S + i] = 0;
	uint32_t *col = img + lane;
	HSTAMP_DECL;
	for (;;) {
		uint32_t u = 0;
		if (lane == 0) {
			const uint32_t t = atomicAdd(&s_ticket, 1u);
			const uint32_t c = t / SYNC_UC, slot = t % SYNC_UC;
			unsigned long long *cs = &s_chunk[c & 63u];
			if (slot == 0) {
				u = atomicAdd(&a.ctl->sunits, SYNC_UC);
				__hip_atomic_store(cs, ((unsigned long long) c << 32) | u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			} else {
				unsigned long long v;
				do {
					v = __hip_atomic_load(cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				} while ((uint32_t) (v >> 32) != c);
				u = (uint32_t) v + slot;
			}
		}
		u = uniform(u);
		if (u >= nunits)
			break;
		HSTAMP(0); // ticket
		const uint32_t k = u / (HT / 64);
		const uint32_t tid = (u % (HT / 64)) * 64 + lane; // the lane's subsequence of the tile
		const HufTile *dp = a.htiles + k;
		const uint32_t nbits_t = uniform(dp->nbits);
		const uint32_t t = uniform(dp->t_last) & 0x7FFFFFFFu;
		const uint8_t *src = a.in + dp->src;
		const int32_t nby = (int32_t) ((nbits_t + 7) >> 3); // (nbits_t < 2^32: below 2^29 bytes)
		const bool exact = t == 0 && tid < 64; // the unit's first lane starts the read's payload: its start is known

		wave_lds_sync(); // (the image is free: the last unit's scans are through)
		col_load<NDW, S>(col, src, (int32_t) tid * (OWN / 8), 0, nby);
		wave_lds_sync();
		HSTAMP(1); // descriptor + payload loads
		// payload end in the coordinates of the own column
		const uint32_t nb = clamp_nb((int64_t) nbits_t - (int64_t) tid * OWN);

		// ---- every lane runs through the END of its own subsequence: where that run crosses into the next
		// subsequence is the right neighbour's guess of its first code's start.  (No lane reads another's column, and
		// no bits in front of the unit are loaded: the unit's first lane has no guess - it starts at bit 0 and is
		// listed unchecked as before; k_huf_fix decodes it again unless bit 0 happened to be right.  A second payload
		// load for that one lane - a second memory round trip per unit - cost more than those repairs.)
		uint32_t f, c0, c, dz, dv;
		{
			const uint32_t g = len_scan<TRIE, S>(col, &lds.tabs, a.huff, (uint32_t) (OWN - RUNUP), OWN, nb, c0, dz);
			const uint32_t gl = (uint32_t) __shfl_up((int) g, 1, 64); // the left neighbour's run
			f = gl == HEND ? HEND : gl - OWN;
			if (gl == HEND && nb > 0)
				f = 0; // the run ended in the payload's last bits or in a bit pattern that is no code: any guess will do
			if (lane == 0)
				f = nb > 0 ? 0u : HEND;
		}
		HSTAMP(2); // run-up
		const uint32_t e = len_scan<TRIE, S>(col, &lds.tabs, a.huff, f, OWN, nb, c, dv);
		HSTAMP(3); // own pass
		const uint32_t e8 = e == HEND ? R_END : e - OWN;
		const uint32_t cw = wave_scan(c), dw = wave_scan(dv);
		// ---- whose guess is not where the neighbour ended
		const uint32_t f8 = f == HEND ? R_END : f;
		const uint32_t pe8 = (uint32_t) __shfl_up((int) e8, 1, 64);
		const bool broken = lane ? f8 != pe8 : !exact;
		const unsigned long long bm = __ballot(broken);
		a.hrec[(uint64_t) k * HT + tid] = pack_rec(f, c, dv);
		a.hend[(uint64_t) k * HT + tid] = (uint8_t) e8;
		// per unit: the wave's totals (k_huf_fix adds what its repairs change) and the lanes to be looked at again, one
		// 64-bit word from which k_huf_list makes the dense list of k_huf_fix's first round.  (A list written here -
		// slots taken 128 at a time per wave, a few scattered stores per unit - cost this kernel 80 us of its 600.
		// Both stores by lane 63: a store by lane 0 at the end of the loop body was merged by the compiler with the
		// ticket code of lane 0 at its top, across the wave-uniform read of the ticket - the loop never ended.)
		if (lane == 63) {
			a.hwave[u] = make_uint2(cw, dw & 0xFFFFu);
			a.hbits[u] = bm;
		}
		HSTAMP(4); // records, list
	}
	HSTAMP_FLUSH(0);
}

// The list of k_huf_fix's first round out of k_huf_sync's words: a thread per unit (a quarter tile), its set bits
// -> entries tile * HT + subsequence; slots by ONE atomic per workgroup of 1024 units (ctl->ticket2 = the list's
// length; an atomic per wave - 4800 returning atomics on one address - made this kernel 60 us long).
constexpr int LIST_WG = 1024;
__global__ __launch_bounds__(LIST_WG) void k_huf_list(DecodeArgs a)
{
	__shared__ uint32_t s_w[LIST_WG / 64];
	__shared__ uint32_t s_base;
	const uint32_t n = min(uniform(a.ctl->nchunks), a.max_htiles) * (HT / 64);
	const uint32_t u = blockIdx.x * LIST_WG + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	unsigned long long bm = u < n ? a.hbits[u] : 0ull;
	const uint32_t cnt = (uint32_t) __popcll(bm);
	const uint32_t inc = wave_scan(cnt);
	if (lane == 63)
		s_w[w] = inc;
	__syncthreads();
	uint32_t before = 0, total = 0;
#pragma unroll
	for (int i = 0; i < LIST_WG / 64; i++) {
		const uint32_t x = s_w[i];
		if (i < (int) w)
			before += x;
		total += x;
	}
	if (threadIdx.x == 0)
		s_base = total ? atomicAdd(&a.ctl->ticket2, total) : 0u;
	__syncthreads();
	uint32_t at = s_base + before + inc - cnt;
	while (bm) {
		const uint32_t b = (uint32_t) __builtin_ctzll(bm);
		bm &= bm - 1;
		if (at < a.hlist_cap)
			a.hlist[at] = u * 64 + b; // = tile * HT + subsequence
		at++;
	}
}

// Subsequences whose start was guessed wrong, 64 to a wave: decoded again from where the subsequence in front
// ended.  If that moves the subsequence's own end, its right neighbour goes on the next round's list.  Lists:
// a.hlist[0 .. cap) and a.hlist[cap .. 2 cap) take turns (`round` odd: the second is read), their counts in
// ctl->ticket2 (what k_huf_sync listed) and ctl->lists[round] (what this round lists).  `last`: what is still pushed marks its read for k_huf_serial (a.hmin).
#ifndef HUF_FIX_WG
#define HUF_FIX_WG 1024
#endif
constexpr int FIX_WG = HUF_FIX_WG; // waves that are each on their own share the tables
template <int RU, bool TRIE>
__global__ __launch_bounds__(FIX_WG) void k_huf_fix(DecodeArgs a, int round, int last)
{
	constexpr int OWN = HufGeo<RU>::OWN, NDW = HufGeo<RU>::NDW, NCOL = HufGeo<RU>::NCOL;
	struct Lds { // (one struct: see k_huf_sync)
		LenTabs tabs;
		uint32_t imgs[FIX_WG / 64][NCOL * 64 + 64];
	};
	__shared__ __attribute__((aligned(16))) Lds lds;
	auto &imgs = lds.imgs;

	const uint32_t *in_list = a.hlist + (round & 1 ? a.hlist_cap : 0u);
	uint32_t *out_list = a.hlist + (round & 1 ? 0u : a.hlist_cap);
	const uint32_t *in_cnt = round ? &a.ctl->lists[round - 1] : &a.ctl->ticket2;
	uint32_t *out_cnt = &a.ctl->lists[round]; // (zero since the control block was cleared)
	const uint32_t n = min(uniform(*in_cnt), a.hlist_cap);
	if ((uint32_t) FIX_WG * blockIdx.x >= n)
		return;
	const uint32_t lane = threadIdx.x & 63;
	load_len_tables(a.huff, &lds.tabs, FIX_WG);
	__syncthreads(); // the tables are loaded by all waves together (every wave of the workgroup gets here: the early return is per workgroup)
	uint32_t *col = imgs[threadIdx.x >> 6] + lane;
	for (uint32_t i0 = FIX_WG * blockIdx.x + (threadIdx.x & ~63u); i0 < n; i0 += (uint32_t) FIX_WG * gridDim.x) {
		uint32_t ent = i0 + lane < n ? in_list[i0 + lane] : LIST_NONE;
		// (k_huf_sync lists the first lane of every quarter tile unchecked: most of them started where the
		// subsequence in front ended - nothing to do)
		if (ent != LIST_NONE && (a.hrec[ent] & 0xFFu) == a.hend[ent - 1])
			ent = LIST_NONE;
		const bool mine = ent != LIST_NONE;
		const uint64_t g = mine ? ent : 1u;
		const uint32_t k = (uint32_t) (g / HT), tid = (uint32_t) (g % HT);
		const HufTile *dp = a.htiles + k;
		const uint32_t nbits_t = mine ? dp->nbits : 0u;
		wave_lds_sync();
		if (mine)
			col_load<NDW>(col, a.in + dp->src, (int32_t) tid * (OWN / 8), 0, (int32_t) ((nbits_t + 7) >> 3));
		wave_lds_sync();
		// the true start: where the subsequence in front ended (the first of a read never comes here)
		const uint32_t pe = mine ? a.hend[g - 1] : R_END;
		const uint32_t old_e = mine ? a.hend[g] : R_END;
		uint32_t c2, d2;
		const uint32_t e2 = len_scan<TRIE>(col, &lds.tabs, a.huff, (!mine || pe == R_END) ? HEND : pe, OWN,
					     clamp_nb((int64_t) nbits_t - (int64_t) tid * OWN), c2, d2);
		const uint32_t e8 = e2 == HEND ? R_END : e2 - OWN;
		bool next = false;
		if (mine) {
			const uint32_t old = a.hrec[g];
			a.hrec[g] = pack_rec(pe == R_END ? HEND : pe, c2, d2);
			// the wave's totals follow, both in one 64-bit add (the count's borrows and carries cancel)
			const long long dc = (long long) (int32_t) (c2 - ((old >> 8) & 0xFFu));
			const unsigned long long dd = (d2 - (old >> 16)) & 0xFFFFu;
			if (dc != 0 || dd != 0)
				atomicAdd(reinterpret_cast<unsigned long long *>(a.hwave + (uint64_t) k * (HT / 64) + (tid >> 6)),
					  (dd << 32) + (unsigned long long) dc);
			if (e8 != old_e) {
				a.hend[g] = (uint8_t) e8;
				// the right neighbour, if the read has one, started from the old end
				const bool more = tid + 1 < HT || !(dp->t_last >> 31);
				next = more && (uint64_t) tid * OWN + OWN < nbits_t;
			}
		}
		list_push(out_list, out_cnt, a.hlist_cap, next, (uint32_t) g + 1, lane);
		if (last && next)
			atomicMin(&a.hmin[dp->read], (uint32_t) g + 1);
	}
}

// What k_huf_fix's rounds left (a code whose lengths share a factor never synchronises; an ordinary table is
// through after two rounds): one wave per marked read walks its subsequences from the first unsettled one,
// serially - slow and always right.
template <int RU, bool TRIE>
__global__ __launch_bounds__(64) void k_huf_serial(DecodeArgs a)
{
	constexpr int OWN = HufGeo<RU>::OWN, NDW = HufGeo<RU>::NDW, NCOL = HufGeo<RU>::NCOL;
	struct Lds { // (one struct: see k_huf_sync)
		LenTabs tabs;
		uint32_t img[NCOL * 64 + 64];
	};
	__shared__ __attribute__((aligned(16))) Lds lds;
	auto &img = lds.img;

	const uint32_t r = blockIdx.x;
	const uint32_t g0 = uniform(a.hmin[r]);
	if (g0 == 0xFFFFFFFFu)
		return;
	const uint32_t lane = threadIdx.x;
	const uint32_t k0 = uniform(a.hread[2 * r]), nt = uniform(a.hread[2 * r + 1]);
	load_len_tables(a.huff, &lds.tabs, 64);
	uint32_t *col = img + lane;
	const uint64_t gend = (uint64_t) (k0 + nt) * HT;
	for (uint64_t g = g0; g < gend; g++) {
		const uint32_t k = (uint32_t) (g / HT), tid = (uint32_t) (g % HT);
		const HufTile *dp = a.htiles + k;
		const uint32_t nbits_t = uniform(dp->nbits);
		if ((uint64_t) tid * OWN >= nbits_t)
			break; // behind the payload
		const uint32_t pe = uniform((uint32_t) __hip_atomic_load(a.hend + g - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
		const uint32_t rec = uniform(__hip_atomic_load(a.hrec + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
		if ((rec & 0xFFu) == pe)
			continue; // this link holds
		wave_lds_sync();
		if (lane == 0)
			col_load<NDW>(col, a.in + dp->src, (int32_t) tid * (OWN / 8), 0, (int32_t) ((nbits_t + 7) >> 3));
		wave_lds_sync();
		uint32_t c2, d2;
		const uint32_t e2 = len_scan<TRIE>(col, &lds.tabs, a.huff, (lane || pe == R_END) ? HEND : pe, OWN,
					     clamp_nb((int64_t) nbits_t - (int64_t) tid * OWN), c2, d2);
		if (lane == 0) {
			a.hrec[g] = pack_rec(pe == R_END ? HEND : pe, c2, d2);
			atomicAdd(reinterpret_cast<unsigned long long *>(a.hwave + (uint64_t) k * (HT / 64) + (tid >> 6)),
				  ((unsigned long long) ((d2 - (rec >> 16)) & 0xFFFFu) << 32) +
					  (unsigned long long) (long long) (int32_t) (c2 - ((rec >> 8) & 0xFFu)));
			__hip_atomic_store(a.hend + g, (uint8_t) (e2 == HEND ? R_END : e2 - OWN), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
	}
}

// the delta a 16-bit zig-zag value stands for, mod 2^16 (trans.c:80)
__device__ __forceinline__ uint32_t unzz16(uint32_t z) { return ((z >> 1) ^ (0u - (z & 1u))) & 0xFFFFu; }

// One workgroup per read: the codes and the sum of the deltas in front of every tile, what the read delivers
// (huffman.c:1243: at most `want` values), the running sum of the exceptions' deltas (into the upper half of
// ex_val) and whether k_huf_emit may write the samples itself.
__global__ __launch_bounds__(HT) void k_huf_chain(DecodeArgs a)
{
	__shared__ uint32_t s_w[HT / 64], s_wd[HT / 64];

	const uint32_t r = blockIdx.x;
	const uint32_t tid = threadIdx.x;
	const uint32_t k0 = uniform(a.hread[2 * r]), nt = uniform(a.hread[2 * r + 1]);
	if (!nt)
		return;
	// exclusive prefix of the tiles' counts and delta sums
	uint64_t cum = 0;
	uint32_t dcum = 0;
	for (uint32_t b = 0; b < nt; b += HT) {
		const uint32_t u = b + tid;
		uint32_t c = 0, dt = 0;
		if (u < nt) { // the tile's totals: its four waves'
			const uint4 *wp = reinterpret_cast<const uint4 *>(a.hwave + (uint64_t) (k0 + u) * (HT / 64));
			const uint4 w01 = wp[0], w23 = wp[1];
			c = w01.x + w01.z + w23.x + w23.z;
			dt = (w01.y + w01.w + w23.y + w23.w) & 0xFFFFu;
		}
		const uint32_t inc = wave_scan(c);
		const uint32_t dinc = wave_scan(dt);
		if ((tid & 63) == 63) {
			s_w[tid >> 6] = inc;
			s_wd[tid >> 6] = dinc;
		}
		__syncthreads();
		uint32_t before = 0, total = 0, dbefore = 0, dtotal = 0;
#pragma unroll
		for (int w2 = 0; w2 < HT / 64; w2++) {
			const uint32_t x = s_w[w2], y = s_wd[w2];
			if (w2 < (int) (tid >> 6)) {
				before += x;
				dbefore += y;
			}
			total += x;
			dtotal += y;
		}
		if (u < nt) {
			const uint64_t bs = cum + before + inc - c;
			a.htrec[k0 + u].base = bs > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t) bs;
			a.htrec[k0 + u].dbase = (dcum + dbefore + dinc - dt) & 0xFFFFu;
		}
		cum += total;
		dcum += dtotal;
		__syncthreads();
	}
	const uint32_t want = uniform(a.htiles[k0].want);
	const uint32_t nlow = cum < want ? (uint32_t) cum : want; // what huffman_decode_memory delivered
	// running sum of the exceptions' deltas: ex_val[e] = value | (sum of the deltas of exceptions < e) << 16
	const uint32_t nex = uniform(a.meta[r].nex);
	const uint64_t o0 = a.htiles[k0].low;
	uint32_t *val = a.ex_val + o0;
	uint32_t xcum = 0;
	for (uint32_t b = 0; b < nex; b += HT) {
		const uint32_t e = b + tid;
		const uint32_t z = e < nex ? (val[e] & 0xFFFFu) : 0u;
		const uint32_t dl = e < nex ? unzz16(z) : 0u;
		const uint32_t inc = wave_scan(dl);
		if ((tid & 63) == 63)
			s_w[tid >> 6] = inc;
		__syncthreads();
		uint32_t before = 0, total = 0;
#pragma unroll
		for (int w2 = 0; w2 < HT / 64; w2++) {
			const uint32_t x = s_w[w2];
			if (w2 < (int) (tid >> 6))
				before += x;
			total += x;
		}
		if (e < nex)
			val[e] = z | (((xcum + before + inc - dl) & 0xFFFFu) << 16);
		xcum += total;
		__syncthreads();
	}
	// k_huf_emit writes the samples itself if the lists interleave into exactly 1 + nlow + nex samples:
	// every exception sits among (or right behind) the delivered values
	const bool fused = nlow >= 1 && (nex == 0 || a.ex_pos[o0 + nex - 1] < nlow + nex);
	for (uint32_t u = tid; u < nt; u += HT)
		a.htrec[k0 + u].fused = fused ? 1u : 0u;
	static_assert(HT / 64 == 4, "a tile's wave totals are two 16-byte loads");
	if (tid == 0) {
		a.meta[r].nlow = nlow;
		if (fused)
			a.hread[2 * r + 1] = nt | HUF_FUSED;
	}
}

// ------------------------------------------------------------------ k_huf_emit

// staging bytes per wave: 64 per lane - a subsequence holds at most 64 codes (OWN bits / the shortest code), so the buffer
// takes whatever the wave decodes.  (With 52 per lane - NA12878 averages 47.4 - a unit with more symbols went a slow way
// through global memory; low-entropy stretches such as a read's stall, 1 % of the samples with 4-bit codes throughout,
// cost 17 % of the decode time, a whole batch of them 9 x: tools/lowent.py.)
constexpr uint32_t EMIT_STG = 4096;

// The first nmine codes from bit p of the column on: the DELTA each symbol stands for (zig-zag undone, one
// signed byte) to wp[0 ..] (LDS staging or the one-byte stream's place).  k_huf_sync counted the codes that
// start in the subsequence, so the count alone ends the loop: no position is checked against the end.
// lut entry (HuffDev::lut32): d1 | adv << 8 | d2 << 16 | len1 << 24 | HUF_TWO; long codes through lut2.
// A step: room test, row address, the two dwords around the position (read from the column at every step), funnel
// shift, look-up, long-code test, two byte stores, position and output pointer moved on - 10 vector instructions
// (19 + a 64-bit multiply-add with a register window over the column, a conditional second store and a symbol
// counter).  Both bytes of an entry are always stored: while the count has room for two, the second slot is the
// lane's own, and a one-code entry's second byte is overwritten by the lane's next store.  The position runs 30 bits
// ahead, so that the window bits sit at bits 2 .. 13 of the funnel shift: masked, they are the table's byte offset.
template <bool TRIE>
__device__ __forceinline__ void emit_codes(const uint32_t *col_, const uint32_t *lut_, const uint16_t *lut2_,
					   const HuffDev *hd, uint32_t p, uint32_t nmine, uint8_t *wp_)
{
	const lds_cu32p col = (lds_cu32p) col_;
	const lds_cu8p lut = (lds_cu8p) lut_;
	const lds_cu16p lut2 = (lds_cu16p) lut2_;
	// a long code (13 .. 24 bits) as a one-code entry; 0: the bits are no code (cannot come up among
	// the codes k_huf_sync counted).  wnd: the stream bits from the code's first on
	auto long_entry = [&](uint32_t e, uint32_t wnd) -> uint32_t {
		if (TRIE && e == 0xFFFFFFFFu) {
			const uint32_t t = trie_code(hd, wnd);
			return t ? ((t & 0xFFFFFF00u) | ((uint32_t) unzz8(t & 0xFFu) & 0xFFu)) : 0u;
		}
		const uint32_t e2 = lut2[(e & 0xFFFu) + ((wnd >> HUF_LUT_BITS) & ((1u << ((e >> 12) & 15u)) - 1u))];
		return e2 == 0xFFFFu ? 0u : (((uint32_t) unzz8(e2 & 0xFFu) & 0xFFu) | (e2 & 0x1F00u) | ((e2 & 0x1F00u) << 16));
	};
	typedef lds_u8p WP; // (the wave's staging buffer)
	WP w = (WP) wp_;
	const WP wend = w + nmine;
	WP wlim = nmine ? wend - 1 : w; // a step needs room for two: w < wlim
	constexpr uint32_t PB = 30;
	const lds_cu32p colm = col - 64;
	uint32_t pp = p + PB;
	uint32_t jj = pp >> 5;
	uint32_t w0 = colm[__umul24(jj, 64u)], w1 = colm[__umul24(jj, 64u) + 64];
	while (w < wlim) { // while the count has room for two (a divergent loop: the lanes that are through leave it)
		{
			const uint32_t w2 = colm[__umul24(jj, 64u) + 128]; // the dword behind the window: on its way during the look-up
			const uint32_t wnd = __builtin_amdgcn_alignbit(w1, w0, pp);
			uint32_t e = *(lds_cu32p) (lut + (wnd & (((1u << HUF_LUT_BITS) - 1u) << 2)));
			if (e >= HUF_LONG) // rare (one code in 200): longer than 12 bits
				e = long_entry(e, wnd >> 2);
			// (bits that are no code - they cannot come up among the codes k_huf_sync counted with the same
			// tables - give e = 0: a zero is stored and the output moves on by one, so the loop ends anyway)
			w[0] = (uint8_t) e;
			w[1] = (uint8_t) (e >> 16);
			w += 1u + ((e >> 29) & 1u);
			pp += (e >> 8) & 0x1Fu;
			const uint32_t jn = pp >> 5; // a step crosses at most one dword
			if (jn != jj) {
				w0 = w1;
				w1 = w2;
			}
			jj = jn;
		}
	}
	if (w < wend) { // the last code of an odd count
		const uint32_t wnd = __builtin_amdgcn_alignbit(w1, w0, pp);
		uint32_t e = *(lds_cu32p) (lut + (wnd & (((1u << HUF_LUT_BITS) - 1u) << 2)));
		if (e >= HUF_LONG)
			e = long_entry(e, wnd >> 2);
		w[0] = (uint8_t) e;
	}
}

// ---- samples out of the one-byte values (trans.c:260 undone on the fly)
//
// Sample i >= 1 of a read is exception e if pos[e] == i - 1, else one-byte value number (i - 1) -
// #exceptions in front of it; sample 0 is zd[0] from the header.  In the order of the samples,
// exception e therefore sits right in front of value number key(e) = pos[e] - e.  A wave that
// delivers the values [L0, L1) also writes the exceptions with key in [L0, L1) (the one with the
// read's last value: also those behind it), and the one with value 0: sample 0.

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

// exceptions e with pos[e] - e < key (wave-uniform arguments and result)
__device__ __forceinline__ uint32_t keys_below(const uint32_t *pos, uint32_t nex, uint32_t key)
{
	uint32_t lo = 0, hi = nex;
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (uniform(pos[mid]) - mid < key)
			lo = mid + 1;
		else
			hi = mid;
	}
	return lo;
}

// the deltas of the 8 samples at i0 (16 bits each, two per register) with the exceptions
// [e_first, e_first + e_cnt) merged in; samples outside [Ia, Ib) are left zero.  lowat(l) = delta of one-byte
// value number l.
template <typename LOWAT>
__device__ __forceinline__ void gather8(LOWAT lowat, const uint32_t *pos, const uint32_t *val, uint32_t zd0,
					uint32_t i0, uint32_t Ia, uint32_t Ib, uint32_t e_first, uint32_t e_cnt,
					uint32_t v[4])
{
	v[0] = v[1] = v[2] = v[3] = 0;
	const uint32_t lo = i0 > Ia ? i0 : Ia;
	const uint32_t hi = i0 + 8 < Ib ? i0 + 8 : Ib;
	if (lo >= hi)
		return;
	const uint32_t uf = lo ? lo - 1 : 0;
	const uint32_t e_end = e_first + e_cnt;
	uint32_t e = e_first + lower_bound_u32(pos + e_first, e_cnt, uf);
	uint32_t l = uf - e; // number of the next one-byte value
	uint32_t nextpos = e < e_end ? pos[e] : 0xFFFFFFFFu;
#pragma unroll
	for (int h = 0; h < 8; h++) {
		const uint32_t i = i0 + h;
		if (i >= lo && i < hi) {
			uint32_t dl;
			if (i == 0) {
				dl = unzz16(zd0);
			} else if (i - 1 == nextpos) {
				dl = unzz16(val[e] & 0xFFFFu);
				e++;
				nextpos = e < e_end ? pos[e] : 0xFFFFFFFFu;
			} else {
				dl = lowat(l) & 0xFFFFu;
				l++;
			}
			v[h >> 1] |= dl << (16 * (h & 1));
		}
	}
}

struct EmitRead { // what a wave needs of its read to write samples (wave-uniform)
	const uint32_t *pos, *val;
	int16_t *out;
	uint32_t nex, zd0, q, nlow;
};

// emit_samples' groups of 8 samples start at multiples of 8 samples: 16-byte stores at 16-byte addresses
// (measured: groups that start at the wave's first sample - stores at any 2-byte address, one ragged round
// less per wave - make k_huf_emit 4 % slower, at multiples of 2 samples 7 %)
constexpr uint32_t EMIT_ALIGN = 8;

// 8 signed bytes -> 4 packed pairs of 16-bit values
__device__ __forceinline__ void expand8s(uint2 dd, uint32_t v[4])
{
	typedef short i16x2 __attribute__((ext_vector_type(2)));
	const i16x2 eight = { 8, 8 };
	const uint32_t t[4] = { __builtin_amdgcn_perm(0, dd.x, 0x010c000c), __builtin_amdgcn_perm(0, dd.x, 0x030c020c),
				__builtin_amdgcn_perm(0, dd.y, 0x010c000c), __builtin_amdgcn_perm(0, dd.y, 0x030c020c) };
#pragma unroll
	for (int h = 0; h < 4; h++)
		v[h] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(i16x2, t[h]) >> eight);
}

struct EmitPlan { // where a wave's samples lie and what they start from (wave-uniform but for pe)
	uint32_t L0, L1;   // the wave's one-byte values
	uint32_t Ea, ecnt; // its exceptions: [Ea, Ea + ecnt)
	uint32_t Ia, Ib;   // its samples
	uint32_t base;     // value of the sample in front of Ia
	uint32_t pe;       // lane e < ecnt <= 64: position of exception Ea + e
};

// What emit_samples needs besides the values themselves.  B0 = sum of the deltas of the values in front of L0
// (mod 2^16); key = pos[lane] - lane of the read's first 64 exceptions, loaded by the caller BEFORE the wave
// decodes its codes (the load is under way meanwhile).
__device__ __forceinline__ EmitPlan emit_plan(const EmitRead &R, uint32_t L0, uint32_t quota, uint32_t B0, uint32_t lane,
					      uint32_t kraw, uint32_t vraw)
{
	const uint32_t key = kraw == 0xFFFFFFFFu ? kraw : kraw - lane; // kraw: pos[lane] of the read's first 64 exceptions
	EmitPlan P;
	P.L0 = L0;
	P.L1 = L0 + quota;
	const uint32_t nex = R.nex;
	const bool lastw = P.L1 == R.nlow;
	// exceptions in front of the wave's first / behind its last sample
	uint32_t Ea = 0, Eb = 0;
	if (nex && nex <= 64) {
		Ea = (uint32_t) __popcll(__ballot(key < P.L0));
		Eb = (uint32_t) __popcll(__ballot(key < P.L1));
	} else if (nex) {
		Ea = keys_below(R.pos, nex, P.L0);
		Eb = keys_below(R.pos, nex, P.L1);
	}
	if (lastw)
		Eb = nex;
	P.Ea = Ea;
	P.ecnt = Eb - Ea;
	P.Ia = L0 ? L0 + Ea + 1 : 0u;
	P.Ib = P.L1 + Eb + 1;
	P.base = 0;
	if (L0) {
		P.base = unzz16(R.zd0) + B0;
		if (Ea) {
			// (a read of at most 64 exceptions - nearly every read - has them all in kraw / vraw, one per lane,
			// asked for before the decode loop: no load here that the sample phase would have to wait for)
			const uint32_t pv = nex <= 64 ? (uint32_t) __builtin_amdgcn_readlane((int) vraw, (int) (Ea - 1)) : uniform(R.val[Ea - 1]);
			P.base += (pv >> 16) + unzz16(pv & 0xFFFFu);
		}
	}
	// the wave's exceptions, one per lane (more than 64: searched where needed)
	if (nex <= 64) {
		const uint32_t pl = (uint32_t) __shfl((int) kraw, (int) ((Ea + lane) & 63u), 64);
		P.pe = lane < P.ecnt ? pl : 0xFFFFFFFFu;
	} else {
		P.pe = (P.ecnt <= 64 && lane < P.ecnt) ? R.pos[Ea + lane] : 0xFFFFFFFFu;
	}
	return P;
}

// The deltas of the wave's values [L0, L1) are in the LDS staging buffer (src[l - L0]): write the samples they and
// their exceptions make, 16 samples per lane and round (1024 per round; 8 per lane cost twice the wave scans and rounds).
__device__ __forceinline__ void emit_samples16(const uint8_t *src, const EmitRead &R, const EmitPlan &P, uint32_t lane)
{
	const uint32_t L0 = P.L0, L1 = P.L1, Ea = P.Ea, ecnt = P.ecnt, Ia = P.Ia, Ib = P.Ib, pe = P.pe;
	uint32_t base = P.base;
	auto ex_below = [&](uint32_t key) -> uint32_t { // exceptions of the wave with pos < key
		if (ecnt == 0)
			return 0u;
		if (ecnt <= 64)
			return (uint32_t) __popcll(__ballot(pe < key));
		return uniform(lower_bound_u32(R.pos + Ea, ecnt, key));
	};
	const bool shift = uniform(R.q) != 0;
	const u16x2 qq = { (unsigned short) R.q, (unsigned short) R.q };
	const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
	for (uint32_t g = Ia & ~(EMIT_ALIGN - 1); g < Ib; g += 1024) {
		const uint32_t i0 = g + lane * 16;
		// exceptions in front of sample g / of sample g + 1024
		const uint32_t e0 = ex_below(g ? g - 1 : 0u), e1 = ex_below(g + 1023);
		const bool plain = e0 == e1 && g != 0;
		const bool ragged = g < Ia || g + 1024 > Ib;
		uint32_t v[8];
		if (plain) {
			// sample i = value number i - 1 - (Ea + e0)
			const bool any = i0 + 16 > Ia && i0 < Ib;
			const int32_t o = any ? (int32_t) (i0 - 1 - Ea - e0 - L0) : 0; // >= -16 (the staging buffers have room in front)
			const int32_t j = o >> 2;
			const uint32_t sh = (uint32_t) (o & 3) * 8u; // (the same in every lane)
			const uint32_t d0 = s32[j], d1 = s32[j + 1], d2 = s32[j + 2], d3 = s32[j + 3], d4 = s32[j + 4];
			uint2 da = make_uint2(__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh));
			uint2 db = make_uint2(__builtin_amdgcn_alignbit(d3, d2, sh), __builtin_amdgcn_alignbit(d4, d3, sh));
			if (g < Ia) { // the first round: what lies in front of the wave's first sample does not count
				// (behind its last sample nothing needs a mask: those deltas only reach samples that are not stored)
				auto mask8 = [&](uint2 &dd, uint32_t ia) {
					const uint32_t lo = Ia > ia ? (Ia - ia < 8 ? Ia - ia : 8u) : 0u;
					const uint64_t m = lo >= 8 ? 0ull : ~((1ull << (8 * lo)) - 1ull);
					dd.x &= (uint32_t) m;
					dd.y &= (uint32_t) (m >> 32);
				};
				mask8(da, i0);
				mask8(db, i0 + 8);
			}
			expand8s(da, v);
			expand8s(db, v + 4);
		} else {
			auto lowat = [&](uint32_t l) -> uint32_t { return (l >= L0 && l < L1) ? (uint32_t) (int32_t) (int8_t) src[l - L0] : 0u; };
			gather8(lowat, R.pos, R.val, R.zd0, i0, Ia, Ib, Ea, ecnt, v);
			gather8(lowat, R.pos, R.val, R.zd0, i0 + 8, Ia, Ib, Ea, ecnt, v + 4);
			// (this rare path loads exception positions / values it may not use: with such a load possibly pending at
			// the loop's back edge the compiler put a full vector-memory wait into the PLAIN path of every round - a
			// wait for the previous round's sample stores.  Nothing of this path is pending behind this line.)
			__builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
		}
		const uint32_t ta = lane_prefix8(v) & 0xFFFFu;
		const uint32_t tb = lane_prefix8(v + 4) & 0xFFFFu;
		const uint32_t tot = (ta + tb) & 0xFFFFu;
		const uint32_t inc = wave_incl_scan_dpp(tot);
		const uint32_t ba = (base + inc - tot) & 0xFFFFu, bb = (ba + ta) & 0xFFFFu;
		const uint32_t ba2 = ba | (ba << 16), bb2 = bb | (bb << 16);
#pragma unroll
		for (int h = 0; h < 4; h++) {
			v[h] = pk_add16(v[h], ba2);
			v[4 + h] = pk_add16(v[4 + h], bb2);
		}
		if (shift) { // ex_zd.c:396 do_rev_qts_inplace
#pragma unroll
			for (int h = 0; h < 8; h++)
				v[h] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, v[h]) << qq);
		}
#if defined(EMIT_ABL) && EMIT_ABL == 1
		if (R.nlow == 0x7FFFFFFFu)
#endif
#pragma unroll
		for (int hh = 0; hh < 2; hh++) {
			const uint32_t ia = i0 + 8 * hh;
			if (!ragged || (ia >= Ia && ia + 8 <= Ib)) {
				typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
				const u32x4 vv = { v[4 * hh], v[4 * hh + 1], v[4 * hh + 2], v[4 * hh + 3] };
#ifdef EMIT_NT_STORES
				__builtin_nontemporal_store(vv, reinterpret_cast<u32x4 *>(R.out + ia)); // (written once, never read here)
#else
				*reinterpret_cast<u32x4 *>(R.out + ia) = vv;
#endif
			} else if (ia < Ib && ia + 8 > Ia) { // (the one group that straddles the wave's first or last sample)
#pragma unroll
				for (uint32_t h = 0; h < 8; h++)
					if (ia + h >= Ia && ia + h < Ib)
						R.out[ia + h] = (int16_t) (v[4 * hh + (h >> 1)] >> (16 * (h & 1)));
			}
		}
		base += (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
	}
}

// one-byte values out of their deltas, four at a time (trans.c:74 zig-zag of a byte)
__device__ __forceinline__ uint32_t zz_bytes(uint32_t d)
{
	return ((d << 1) & 0xFEFEFEFEu) ^ (((d >> 7) & 0x01010101u) * 0xFFu);
}

// eight waves share the tables (workgroup sizes that are no multiple of 256 threads do not run two to a CU on this
// GPU whatever their registers and LDS: tools/ubench_lds_occ.hip)
#ifndef EMIT_WG
#define EMIT_WG 512
#endif
#ifndef EMIT_PER_CU
#define EMIT_PER_CU 2
#endif
constexpr int WGE = EMIT_WG; // eight waves, each on its own, share the tables
constexpr uint32_t EMIT_UC = 16; // units of work a workgroup takes from the global counter at a time

struct EmitStg { // the waves' staging buffers; emit_samples16 reads up to 16 bytes in front of / 20 behind a buffer's content
	uint8_t pre[16];
	uint8_t s[WGE / 64][EMIT_STG];
	uint8_t post[32];
};

template <int RU, bool TRIE>
__global__ __launch_bounds__(WGE) void k_huf_emit(DecodeArgs a)
{
	// (a column here: the NDW dwords that are loaded; emit_codes' look-ahead reads up to two rows further - never
	// used: the next wave's column, or the pad rows)
	constexpr int OWN = HufGeo<RU>::OWN, NDW = HufGeo<RU>::NDW, NCOL = HufGeo<RU>::NDW;
	constexpr uint32_t EW = WGE / 64;
	// one struct: emit_codes reads the row in front of a position's and the row behind it - for a wave's first / last
	// row that is the image next to it, the tables or the pad row: anything, as long as it is the workgroup's LDS
	struct Lds {
		alignas(16) uint32_t lut[1 << HUF_LUT_BITS];
		alignas(16) uint16_t lut2[HUF_L2_ENTRIES];
		uint32_t img[EW][NCOL * 64];
		uint32_t pad_row[2 * 64];
		alignas(16) EmitStg stg_all;
	};
	__shared__ __attribute__((aligned(16))) Lds lds;
	auto &lut = lds.lut;
	auto &lut2 = lds.lut2;
	auto &img = lds.img;
	auto &pad_row = lds.pad_row;
	auto &stg_all = lds.stg_all;
	// Units (a quarter of a tile each) are handed out one at a time: ten waves do not spread evenly over four SIMDs,
	// and with a fixed share per wave the kernel lasted as long as the waves of the crowded SIMDs (1.64 instead of
	// 1.19 ms).  A wave draws a ticket from the workgroup's LDS counter; ticket t is unit t % EMIT_UC of the
	// workgroup's chunk t / EMIT_UC, and whoever draws a chunk's first ticket fetches the chunk from the global
	// counter and posts it (chunk number << 32 | first unit); the others wait for that post.
	__shared__ unsigned long long s_chunk[64];
	__shared__ uint32_t s_ticket;

	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wv = threadIdx.x >> 6; // wave of the workgroup
	const uint32_t ntiles = min(uniform(a.ctl->nchunks), a.max_htiles);
	const uint32_t nunits = ntiles * (HT / 64); // a wave's unit of work: a quarter of a tile
	if (threadIdx.x == 0) {
		pad_row[0] = 0; // (keeps the rows behind the columns alive)
		s_ticket = 0;
	}
	if (threadIdx.x < 64)
		s_chunk[threadIdx.x] = ~0ull;
	{
		const uint4 *s4 = reinterpret_cast<const uint4 *>(a.huff->lut32);
		uint4 *d4 = reinterpret_cast<uint4 *>(lut);
		for (uint32_t i = threadIdx.x; i < (1u << HUF_LUT_BITS) / 4; i += WGE)
			d4[i] = s4[i];
		const uint4 *t4 = reinterpret_cast<const uint4 *>(a.huff->lut2);
		uint4 *u4 = reinterpret_cast<uint4 *>(lut2);
		for (uint32_t i = threadIdx.x; i < (uint32_t) HUF_L2_ENTRIES / 8; i += WGE)
			u4[i] = t4[i];
	}
	uint32_t *col = img[wv] + lane;
	uint8_t *stg = stg_all.s[wv];
	__syncthreads(); // the tables; from here on every wave is on its own: columns and staging are private, and
	                 // what the other waves of its tile hold in front of it comes with the tile's records
	// persistent workgroups: the tables are loaded once.
	//
	// The loop is a pipeline two units deep, so that no wave waits for a unit's records or payload: while unit N is
	// decoded, the payload / per-lane record / read record of unit N + 1 are on their way into registers (their
	// addresses come from the tile records of N + 1, asked for one unit earlier), and the ticket of unit N + 2 is
	// drawn and its tile records asked for.  What arrives is used BETWEEN the decode loop and the sample phase of unit
	// N - never right behind the sample phase's stores: vector-memory operations complete in order, and a wait for a
	// load issued behind stores is a wait for the stores.  (Stamps before: records + payload 15 %, scan 8 % of a
	// wave's time; an ablation without decode loop and sample phase left 0.33 of the kernel's 0.98 ms.)
	auto draw = [&]() -> uint32_t { // the next unit of this wave (>= nunits: none)
		uint32_t u = 0;
		if (lane == 0) {
			const uint32_t t = atomicAdd(&s_ticket, 1u);
			const uint32_t c = t / EMIT_UC, slot = t % EMIT_UC;
			unsigned long long *cs = &s_chunk[c & 63u];
			if (slot == 0) {
				u = atomicAdd(&a.ctl->units, EMIT_UC);
				__hip_atomic_store(cs, ((unsigned long long) c << 32) | u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			} else {
				unsigned long long v;
				do {
					v = __hip_atomic_load(cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				} while ((uint32_t) (v >> 32) != c);
				u = (uint32_t) v + slot;
			}
		}
		return uniform(u);
	};
	static_assert(sizeof(HufTile) == 32 && sizeof(HufTRec) == 32 && sizeof(ReadMeta) == 32 && offsetof(HufTile, nbits) == 16 &&
			      offsetof(HufTile, read) == 24 && offsetof(HufTRec, base) == 8 && offsetof(HufTRec, fused) == 28,
		      "the records are read as eight dwords each");
	// a unit's tile records (HufTile, HufTRec, the totals of the tile's four waves: 32 bytes each, wave-uniform) are
	// loaded ONE DWORD PER LANE - lanes 0 .. 7 the HufTile, 8 .. 15 the HufTRec, 16 .. 23 the totals - into a single
	// vector register and read out with v_readlane when they have arrived (as 16-byte loads of every lane, held in
	// flight across a unit, they took 48 registers and the kernel no longer fit four waves per SIMD)
	auto fetch_tile = [&](uint32_t u) -> uint32_t {
		uint32_t v = 0;
		if (u < nunits && lane < 24) {
			const uint32_t k = u / (HT / 64);
			const uint32_t *p = lane < 8 ? reinterpret_cast<const uint32_t *>(a.htiles + k)
					  : lane < 16 ? reinterpret_cast<const uint32_t *>(a.htrec + k)
						      : reinterpret_cast<const uint32_t *>(a.hwave + (uint64_t) k * (HT / 64));
			v = p[lane & 7];
		}
		return v;
	};
	struct Unit { // what a unit needs of its tile (wave-uniform)
		uint32_t u, nbits_t, read, want, fused, cb, db, base, dbase;
		uint64_t srco, roff;
	};
	auto resolve = [&](uint32_t u, uint32_t tv) -> Unit {
		auto at = [&](int i) -> uint32_t { return (uint32_t) __builtin_amdgcn_readlane((int) tv, i); };
		Unit U;
		U.u = u;
		U.srco = ((uint64_t) at(1) << 32) | at(0); // HufTile::src
		U.roff = ((uint64_t) at(3) << 32) | at(2); // ::low: the read's slot - samples in a.sig, one-byte values in
		                                          // a.low, exceptions
		U.nbits_t = at(4);                        // ::nbits
		U.read = at(6);                           // ::read
		U.want = at(7);                           // ::want
		U.base = at(8 + 2);                       // HufTRec::base, ::dbase: codes of the read in front of the tile
		U.dbase = at(8 + 3);                      // and the sum of their deltas
		U.fused = at(8 + 7);                      // ::fused
		const uint32_t wq = u % (HT / 64); // what the tile's waves in front of this one hold
		U.cb = (wq > 0 ? at(16) : 0u) + (wq > 1 ? at(18) : 0u) + (wq > 2 ? at(20) : 0u);
		U.db = (wq > 0 ? at(17) : 0u) + (wq > 1 ? at(19) : 0u) + (wq > 2 ? at(21) : 0u);
		return U;
	};
	struct Stage2 { // a unit's per-lane loads: payload (in registers when every lane's 36 bytes lie inside it), record, read
		uint32_t pay[NDW];
		uint32_t rec;
		uint4 m0, m1;
		bool fast;
	};
	auto issue2 = [&](const Unit &U, Stage2 &S) {
		const uint32_t tid = (U.u % (HT / 64)) * 64 + lane; // the lane's subsequence of the tile
		const int32_t rb0 = (int32_t) tid * (OWN / 8), nby = (int32_t) ((U.nbits_t + 7) >> 3);
		S.fast = !any64(rb0 + 4 * NDW > nby);
		const uint8_t *q = a.in + U.srco + rb0;
		if (S.fast) {
			uint4 t[NDW / 4 ? NDW / 4 : 1];
#pragma unroll
			for (int j = 0; j < NDW / 4; j++)
				t[j] = ld16_pay(q + 16 * j);
#pragma unroll
			for (int j = 0; j < NDW % 4; j++)
				S.pay[NDW / 4 * 4 + j] = ld4_pay(q + 4 * (NDW / 4 * 4 + j));
#pragma unroll
			for (int j = 0; j < NDW / 4; j++) {
				S.pay[4 * j + 0] = t[j].x;
				S.pay[4 * j + 1] = t[j].y;
				S.pay[4 * j + 2] = t[j].z;
				S.pay[4 * j + 3] = t[j].w;
			}
		}
		S.rec = a.hrec[(uint64_t) (U.u / (HT / 64)) * HT + tid];
		const uint4 *mp = reinterpret_cast<const uint4 *>(a.meta + U.read);
		S.m0 = mp[0];
		S.m1 = mp[1];
	};
	auto land2 = [&](const Unit &U, const Stage2 &S) { // the payload into the wave's columns (which must be free)
		wave_lds_sync();
		if (S.fast) {
#pragma unroll
			for (int j = 0; j < NDW; j++)
				col[j * 64] = S.pay[j];
		} else { // a payload's last bytes: the careful loader (its loads are waited for here)
			const uint32_t tid = (U.u % (HT / 64)) * 64 + lane;
			col_load<NDW>(col, a.in + U.srco, (int32_t) tid * (OWN / 8), 0, (int32_t) ((U.nbits_t + 7) >> 3));
		}
		wave_lds_sync();
	};
	HSTAMP_DECL;
	const uint32_t u_first = draw();
	if (u_first >= nunits)
		return;
	Unit U = resolve(u_first, fetch_tile(u_first)), Un = {}, Unn = {};
	Stage2 S, Sn = {};
	issue2(U, S);
	land2(U, S);
	uint32_t u_nxt = draw();
	if (u_nxt < nunits)
		Un = resolve(u_nxt, fetch_tile(u_nxt));
	for (;;) {
		// ---- the units behind this one: the ticket of N + 2 and its tile records; the per-lane loads of N + 1
		const uint32_t u_nn = draw();
		const uint32_t tv_nn = fetch_tile(u_nn);
		const bool more = u_nxt < nunits;
		if (more)
			issue2(Un, Sn);
		HSTAMP(0); // tickets, loads of the next units
		// ---- this unit
		const uint32_t tid = (U.u % (HT / 64)) * 64 + lane; // the lane's subsequence of the tile
		const uint32_t rec = S.rec;
		const uint32_t want = U.want;
		const uint64_t roff = U.roff;
		uint8_t *low = a.low + roff;
		const uint32_t cnt = (rec >> 8) & 0xFFu;
		const uint32_t inc = wave_scan(cnt);
		const uint64_t obase = (uint64_t) U.base + U.cb;
		const uint32_t B0 = U.dbase + (U.db & 0xFFFFu);
		const bool fused = U.fused != 0;
		const uint32_t wsum = uniform((uint32_t) __shfl((int) inc, 63, 64));
		// the wave delivers values [obase, obase + wsum) of the read, cut at `want`
		const uint32_t quota = obase >= want ? 0u : (wsum < want - (uint32_t) obase ? wsum : want - (uint32_t) obase);
		const uint32_t ex = inc - cnt; // codes of the wave in front of this lane
		const uint32_t f = rec & 0xFFu;
		const uint32_t p0 = f == R_END ? 0u : f;
		uint32_t nmine = (ex >= quota || f == R_END) ? 0u : (cnt < quota - ex ? cnt : quota - ex);
		// (at most 64 codes per lane: the staging buffer takes them all; the clamp only keeps records that are not
		// this library's own from writing outside it)
		nmine = ex >= EMIT_STG ? 0u : (nmine < EMIT_STG - ex ? nmine : EMIT_STG - ex);
		uint8_t *dst = low + obase;
		EmitRead R = {};
		if (fused) {
			R.pos = a.ex_pos + roff;
			R.val = a.ex_val + roff;
			R.out = a.sig + roff;
			R.nex = uniform(S.m0.x);  // ReadMeta::nex
			R.zd0 = uniform(S.m0.z);  // ::zd0
			R.q = uniform(S.m0.w);    // ::q
			R.nlow = uniform(S.m1.z); // ::nlow
		}
		(void) tid;
		// symbols go to the wave's staging buffer in their final order
		// (the read's first exceptions: asked for now, needed behind the decode loop)
		// (the position as loaded: any arithmetic on it here would put the wait for it - and for every load issued before
		// it, the next unit's among them - in front of the decode loop)
		const uint32_t kraw = (fused && quota && lane < R.nex) ? R.pos[lane] : 0xFFFFFFFFu;
		const uint32_t vraw = (fused && quota && lane < R.nex) ? R.val[lane] : 0u; // (their values and delta prefixes)
		HSTAMP(2); // scan, read record
		{
#if defined(EMIT_ABL) && EMIT_ABL == 3 // (timing experiments only: wrong results)
			if (a.nreads == 0x7FFFFFFFu)
#endif
			emit_codes<TRIE>(col, lut, lut2, a.huff, p0, nmine, stg + ex);
			HSTAMP(3); // decode
			// what was asked for at the top has arrived (the columns are free, and no store of this unit is in the way)
			if (more)
				land2(Un, Sn);
			else
				wave_lds_sync();
			if (u_nn < nunits)
				Unn = resolve(u_nn, tv_nn);
			HSTAMP(1); // the next units' loads land
			if (fused) {
				if (quota) {
					const EmitPlan P = emit_plan(R, (uint32_t) obase, quota, B0, lane, kraw, vraw);
					HSTAMP(4); // plan
#if defined(EMIT_ABL) && EMIT_ABL == 2
					if (a.nreads == 0x7FFFFFFFu)
#endif
					emit_samples16(stg, R, P, lane);
					HSTAMP(5); // samples
				}
			} else {
				// the one-byte stream, for k_low_decode_chunked: 16-byte stores
				for (uint32_t o = lane * 16; o < quota; o += 64 * 16) {
					uint4 v = *reinterpret_cast<const uint4 *>(stg + o);
					v = make_uint4(zz_bytes(v.x), zz_bytes(v.y), zz_bytes(v.z), zz_bytes(v.w));
					if (o + 16 <= quota) {
						__builtin_memcpy(dst + o, &v, 16); // any byte address
					} else {
						const uint32_t vv[4] = { v.x, v.y, v.z, v.w };
						for (uint32_t b = 0; o + b < quota; b++)
							dst[o + b] = (uint8_t) (vv[b >> 2] >> (8 * (b & 3)));
					}
				}
			}
			wave_lds_sync(); // staging is free again
		}
		HSTAMP(6); // (the rest of a unit)
		if (!more)
			break;
		U = Un;
		S = Sn;
		Un = Unn;
		u_nxt = u_nn;
	}
	HSTAMP_FLUSH(8);
}

// ------------------------------------------------------------------ tiles

// Tiles of every read (runs after k_ex_parse): ids of one read are consecutive.  hread[2r] = first tile,
// hread[2r + 1] = number of tiles.  A wave takes TILES_RPW reads (a lane each; the other lanes only help to write
// the descriptors): with 64 reads to a wave the 8192 reads of a batch kept 128 waves busy for 15 us.
constexpr uint32_t TILES_RPW = 16;
__global__ __launch_bounds__(64) void k_huff_tiles(DecodeArgs a)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t r = lane < TILES_RPW ? blockIdx.x * TILES_RPW + lane : 0xFFFFFFFFu;
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
		a.hmin[r] = 0xFFFFFFFFu; // (k_huf_fix marks the reads its rounds leave unsettled)
	}
	// the descriptors: the wave writes the tiles of its 64 reads together, 64 tiles at a time (a lane per read
	// wrote the 420 tiles of the longest read one after the other)
	uint64_t src0 = 0, low0 = 0, nbits = 0;
	uint32_t want = 0;
	if (nt) {
		const ReadMeta *m = a.meta + r;
		const uint32_t hdr = m->hdr + m->seclen + 4;
		const uint64_t nbytes64 = a.in_len[r] - hdr;
		nbits = 8ull * (nbytes64 > 0x1FFFFFFFull ? 0x1FFFFFFFull : nbytes64);
		src0 = a.in_off[r] + hdr;
		low0 = a.off[r];
		want = m->nlow;
	}
	unsigned long long todo = __ballot(nt != 0);
	while (todo) {
		const int rr = __builtin_ctzll(todo);
		todo &= todo - 1;
		const uint32_t r_nt = (uint32_t) __shfl((int) nt, rr, 64), r_base = (uint32_t) __shfl((int) base, rr, 64);
		const uint32_t r_read = (uint32_t) __shfl((int) r, rr, 64), r_want = (uint32_t) __shfl((int) want, rr, 64);
		const uint64_t r_src = ((uint64_t) (uint32_t) __shfl((int) (src0 >> 32), rr, 64) << 32) | (uint32_t) __shfl((int) src0, rr, 64);
		const uint64_t r_low = ((uint64_t) (uint32_t) __shfl((int) (low0 >> 32), rr, 64) << 32) | (uint32_t) __shfl((int) low0, rr, 64);
		const uint64_t r_bits = ((uint64_t) (uint32_t) __shfl((int) (nbits >> 32), rr, 64) << 32) | (uint32_t) __shfl((int) nbits, rr, 64);
		for (uint32_t t = lane; t < r_nt; t += 64) {
			HufTile d;
			d.src = r_src + t * (TB / 8);
			d.low = r_low;
			d.nbits = (uint32_t) (r_bits - t * TB);
			d.t_last = t | (t + 1 == r_nt ? 0x80000000u : 0u);
			d.read = r_read;
			d.want = r_want;
			a.htiles[r_base + t] = d;
		}
	}
}

#ifndef HUF_FIX_ROUNDS
#define HUF_FIX_ROUNDS 4
#endif
#ifdef HUF_STAMPS
} // namespace ph
extern "C" int press_hip_huf_stamps(unsigned long long *dst, uint32_t nwords)
{
	unsigned long long z[32] = { 0 };
	if (hipDeviceSynchronize() != hipSuccess ||
	    hipMemcpyFromSymbol(dst, HIP_SYMBOL(ph::g_hstamp), (nwords < 32 ? nwords : 32) * 8) != hipSuccess ||
	    hipMemcpyToSymbol(HIP_SYMBOL(ph::g_hstamp), z, sizeof z) != hipSuccess)
		return -1;
	return 0;
}
namespace ph {
#endif

constexpr int HUF_FIX_LAUNCHES = HUF_FIX_ROUNDS; // rounds of k_huf_fix (an ordinary table is through after two)

template <int RU, bool TRIE>
static void run_huff_decode(const DecodeArgs &a, hipStream_t s)
{
	// persistent workgroups: what is resident (sync: 2 per CU of four tiles each, emit: 2 of eight waves)
	const uint32_t nt = a.max_htiles ? a.max_htiles : 1;
	const uint32_t ngs = (nt * (HT / 64) + WGS / 64 - 1) / (WGS / 64);
	const uint32_t grid = ngs < HUF_SYNC_PER_CU * 256u ? ngs : HUF_SYNC_PER_CU * 256u;
	const uint32_t nge = (nt * (HT / 64) + WGE / 64 - 1) / (WGE / 64);
	const uint32_t ge = nge < EMIT_PER_CU * 256u ? nge : EMIT_PER_CU * 256u;
	hipLaunchKernelGGL((k_huf_sync<RU, TRIE>), dim3(grid), dim3(WGS), 0, s, a);
	hipLaunchKernelGGL(k_huf_list, dim3((nt * (HT / 64) + LIST_WG - 1) / LIST_WG), dim3(LIST_WG), 0, s, a);
	for (int round = 0; round < HUF_FIX_LAUNCHES; round++)
		hipLaunchKernelGGL((k_huf_fix<RU, TRIE>), dim3(FIX_WG >= 1024 ? 512 : FIX_WG >= 512 ? 768 : 1280), dim3(FIX_WG), 0, s, a, round, round + 1 == HUF_FIX_LAUNCHES ? 1 : 0);
	hipLaunchKernelGGL((k_huf_serial<RU, TRIE>), dim3(a.nreads), dim3(64), 0, s, a);
	hipLaunchKernelGGL(k_huf_chain, dim3(a.nreads), dim3(HT), 0, s, a);
	hipLaunchKernelGGL((k_huf_emit<RU, TRIE>), dim3(ge), dim3(WGE), 0, s, a); // (ctl->units: zero since the control block was cleared)
}

template <int RU>
static void run_huff_decode_t(const DecodeArgs &a, bool trie, hipStream_t s)
{
	if (trie)
		run_huff_decode<RU, true>(a, s);
	else
		run_huff_decode<RU, false>(a, s);
}

// Huffman stage of the exception-split decoders: payload of every read -> a.low.  minlen: the table's shortest code
// (selects the subsequence size) | HUF_NEEDS_TRIE if some code is beyond the second-level tables
void launch_huff_decode(const DecodeArgs &a0, uint32_t minlen, hipStream_t s)
{
	const bool trie = (minlen & HUF_NEEDS_TRIE) != 0;
	minlen &= ~HUF_NEEDS_TRIE;
	DecodeArgs a = a0;
	a.ctl = a0.ctl + 1; // the decoder's own control block (cleared by k_ex_parse, the first kernel of the call)
	hipLaunchKernelGGL(k_huff_tiles, dim3((a.nreads + TILES_RPW - 1) / TILES_RPW), dim3(64), 0, s, a);
	if (minlen >= 4)
		run_huff_decode_t<128>(a, trie, s);
	else if (minlen >= 2)
		run_huff_decode_t<64>(a, trie, s);
	else
		run_huff_decode_t<32>(a, trie, s);
}

} // namespace ph
