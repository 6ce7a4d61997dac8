// press_huffman.hip - static-Huffman stream decode for gfx950 (huffman.c:1219 huffman_decode_memory).
//
// The stream has no synchronisation points, but Huffman codes self-synchronise: a decoder
// started at a wrong bit position falls back onto true code boundaries after a few codes.
// The payload of every read is cut into TILES of HT subsequences of S bits (S = 128 for
// the NA12878 table); one 512-thread workgroup per tile, handed out in ticket order by a
// persistent grid that keeps the lookup tables in LDS.  Thread i owns the codes that
// START in subsequence i:
//   pass 0    every thread decodes from the first bit of its subsequence and records where
//             the first code of the next subsequence starts, E[i];
//   rounds    thread i takes E[i-1] as its start; whoever's start changed decodes again -
//             the changed threads are compacted so that a round with few changes costs a
//             few waves, not eight - until nothing changes (thread i is final after <= i
//             rounds; in practice after 2-3).  These passes write their symbols into a
//             private 36-byte LDS slot per thread;
//   chain     tiles of one read are chained twice.  As soon as a tile has converged under
//             the assumption that its first bit starts a code, it publishes where its last
//             code ends as a HINT; its successor re-converges from that hint (one or two
//             lanes decode again) without waiting for anything else.  The FINAL granule -
//             {true end position, symbols up to and including this tile} - travels down the
//             read behind that: a tile waits for its predecessor's FINAL, checks that the
//             end position is the hint it used (if not it converges once more), and
//             publishes its own.  The wait therefore costs one L2 round trip per tile, and
//             a wrong hint only costs time.  Ticket order makes the waits deadlock free:
//             a predecessor was always taken by a workgroup that is already running;
//   output    prefix sum of the counts, slots -> contiguous LDS image -> 16-byte stores.
// Result == huffman.c:1219 bit for bit, including its behaviour at the end of the input
// (stops when the bytes run out or the symbol count is reached; a code cut off by the end
// of the input is not delivered).

#include "press_internal.h"

namespace ph {

constexpr uint32_t HEND = 0xFFFFFFFFu;  // "no further code": end of input or an undecodable prefix
constexpr uint32_t HNONE = 0xFFFFFFFEu; // never a start position: forces the first round to decode everybody
constexpr int HSLOT = 36;               // bytes per private symbol slot: HSYM + spill; 9 dwords = bank-conflict-free stride
constexpr int HLB_DW = 2176;            // LDS dwords of the bit image (2048 + reach of the last code, skewed)
constexpr uint32_t HINT_READY = 1u << 31;
constexpr uint32_t HINT_END = 1u << 30;
constexpr uint64_t FIN_READY = 1ull << 63;
constexpr uint64_t FIN_END = 1ull << 62;

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); }

// LDS image of a tile's bits: dword j lives at j + (j >> 6), and the slot after every 64th
// dword repeats the dword that follows it.  So dwords j and j+1 are always adjacent (one
// ds_read2_b32), and the 64 lanes of a wave - one subsequence apart - hit 64 different banks.
__device__ __forceinline__ uint32_t la(uint32_t j) { return j + (j >> 6); }

// Decode the codes that start in [start, sub_end) of the tile (bit positions relative to
// the tile); returns where the next code starts, or HEND.  nbits = end of the payload.
// WRITE: symbols go to `slot`, their number to `cnt`.
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
	// wave-uniform and predicated: lanes need different numbers of steps, and per-lane
	// branches cost more than the few masked operations
	for (;;) {
		const bool act = p < lim;
		if (!__any(act))
			break;
		if (act) {
			const uint32_t ad = (p >> 5) + (p >> 11);
			// 32 stream bits from position p (codes are at most 24 bits long)
			const uint32_t wnd = __builtin_amdgcn_alignbit(lbits[ad + 1], lbits[ad], p & 31);
			uint32_t e = lut[wnd & ((1u << HUF_LUT_BITS) - 1u)];
			if (e >= HUF_LONG) { // rare: a code longer than 12 bits
				uint32_t sym = 0, len = 0;
				bool ok;
				if (e != 0xFFFFFFFFu) {
					const uint32_t id = e & 0xFFu;
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
		}
	}
	if (!bad && p > nbits) { // the last code ran off the end of the input: not delivered
		c -= 1;
		bad = true;
	}
	cnt = c;
	return bad ? HEND : (p >= nbits && p < sub_end ? HEND : p);
}

__device__ __forceinline__ uint32_t poll32(uint32_t *p)
{
	uint32_t v;
	while ((v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0)
		__builtin_amdgcn_s_sleep(2);
	return v;
}
__device__ __forceinline__ uint64_t poll64(uint64_t *p)
{
	uint64_t v;
	while ((v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0)
		__builtin_amdgcn_s_sleep(2);
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
#define HSTAMP(i) do { if (tid == 0) { const unsigned long long now_ = clock64(); atomicAdd(&g_hufdbg[i], now_ - stamp_); stamp_ = now_; } } while (0)
#define HCOUNT(i, v) do { if (tid == 0) atomicAdd(&g_hufdbg[i], (unsigned long long) (v)); } while (0)
#else
#define HSTAMP(i) do { } while (0)
#define HCOUNT(i, v) do { } while (0)
#endif

__global__ __launch_bounds__(HUF_HT) void k_huff_decode_tiles(DecodeArgs a)
{
#ifdef HUF_DEBUG
	unsigned long long stamp_ = clock64();
#endif
	constexpr int HT = HUF_HT;
	__shared__ uint32_t lut[1 << HUF_LUT_BITS];
	__shared__ uint16_t lut2[HUF_L2_ENTRIES];
	__shared__ uint16_t l2off[256];
	__shared__ uint8_t l2bits[256];
	__shared__ uint32_t lbits[HLB_DW];
	__shared__ uint32_t sS[HT];      // start position each thread's current result was decoded from
	__shared__ uint32_t sE[HT];      // ... where the code after its subsequence starts
	__shared__ uint32_t sC[HT];      // ... how many codes start in its subsequence
	__shared__ uint16_t lst_id[HT];  // compacted list of threads whose start changed
	__shared__ uint32_t lst_start[HT];
	__shared__ uint32_t lst_n[2];
	__shared__ uint32_t wtot[HT / 64];
	__shared__ uint32_t s_ticket;
	__shared__ uint32_t s_hint;
	__shared__ uint64_t s_fin;
	__shared__ __attribute__((aligned(16))) uint8_t slots[HT * HSLOT];
	__shared__ __attribute__((aligned(16))) uint8_t obuf[16 + HT * HUF_HSYM + 16];

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	for (uint32_t i = tid; i < (1u << HUF_LUT_BITS); i += HT)
		lut[i] = a.huff->lut32[i];
	for (uint32_t i = tid; i < (uint32_t) HUF_L2_ENTRIES; i += HT)
		lut2[i] = a.huff->lut2[i];
	if (tid < 256) {
		l2off[tid] = a.huff->l2off[tid];
		l2bits[tid] = a.huff->l2bits[tid];
	}
	const uint32_t minlen = uniform(a.huff->minlen);
	// bits per subsequence: at most HUF_HSYM = 32 codes start in one, and it holds the longest code (24)
	const uint32_t S = minlen >= 4 ? 128u : minlen >= 2 ? 64u : 32u;
	const uint32_t TB = HT * S;
	const uint32_t nalloc = uniform(a.ctl->nchunks);
	const uint32_t ntiles = nalloc < a.max_htiles ? nalloc : a.max_htiles;
	const uint32_t sub0 = tid * S;
	uint8_t *const myslot = slots + tid * HSLOT;
	uint32_t round = 0; // parity selects lst_n

	// converge: repeat "take the left neighbour's end as start, decode again if it changed"
	// until nothing changes; pos0 = start of thread 0.  Barriers inside; all threads call it.
	auto converge = [&](uint32_t pos0, uint32_t nbits) {
		for (;;) {
			const uint32_t par = round & 1u;
			round++;
			if (tid == 0)
				lst_n[par] = 0;
			__syncthreads();
			const uint32_t ns = tid ? sE[tid - 1] : pos0;
			const bool ch = ns != sS[tid];
			const unsigned long long m = __ballot(ch);
			uint32_t base = 0;
			if (lane == 0 && m)
				base = atomicAdd(&lst_n[par], (uint32_t) __popcll(m));
			base = (uint32_t) __shfl((int) base, 0, 64);
			if (ch) {
				const uint32_t idx = base + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
				lst_id[idx] = (uint16_t) tid;
				lst_start[idx] = ns;
			}
			__syncthreads();
			const uint32_t nch = lst_n[par];
			if (nch == 0)
				break;
			HCOUNT(9, 1);
			HCOUNT(11, nch);
			if ((tid & ~63u) < nch) { // whole waves beyond the list skip
				const bool mine = tid < nch;
				const uint32_t u = mine ? lst_id[tid] : 0u;
				const uint32_t st = mine ? lst_start[tid] : HEND;
				uint32_t c;
				const uint32_t e = huff_run<true>(lbits, lut, lut2, l2off, l2bits, a.huff, st, (u + 1) * S, nbits, c,
								  mine ? slots + u * HSLOT : slots);
				if (mine) {
					sS[u] = st;
					sE[u] = e;
					sC[u] = c;
				}
			}
		}
	};

	for (;;) {
		__syncthreads(); // tables loaded / previous tile's LDS no longer in use
		if (tid == 0)
			s_ticket = atomicAdd(&a.ctl->ticket, 1u);
		__syncthreads();
		const uint32_t k = s_ticket;
		if (k >= ntiles)
			break;
		HSTAMP(0); // ticket
		HCOUNT(8, 1);
		const uint2 desc = a.htiles[k];
		const uint32_t r = uniform(desc.x);
		const uint32_t t = uniform(desc.y) & 0x7FFFFFFFu;
		const bool last = (uniform(desc.y) >> 31) != 0; // no tile of this read follows
		const ReadMeta *m = a.meta + r;
		const uint32_t hdr = uniform(m->hdr) + uniform(m->seclen) + 4;
		const uint8_t *h = a.in + a.in_off[r] + hdr;
		const uint64_t nbytes64 = a.in_len[r] - hdr;
		const uint32_t nbytes = uniform(nbytes64 > 0x1FFFFFFFull ? 0x1FFFFFFFu : (uint32_t) nbytes64);
		const uint32_t want = uniform(m->nlow);
		const uint32_t tile_bit0 = t * TB;                // < 8 * nbytes by construction (k_huff_tiles)
		const uint32_t nbits = nbytes * 8 - tile_bit0;    // payload end, relative to the tile
		uint8_t *low = a.low + a.off[r];

		// ---- stage the tile's bits: lbits[la(j)] = payload bytes [4*(tile_dw0 + j), +4), zeros past the end
		{
			const uint32_t dw0 = tile_bit0 >> 5;
			const uint32_t ndw = TB / 32 + 4;
			for (uint32_t j = tid; j < ndw; j += HT) {
				const uint64_t b = 4ull * (dw0 + j);
				uint32_t v = 0;
				if (b + 4 <= nbytes) {
					__builtin_memcpy(&v, h + b, 4);
				} else {
					for (uint32_t q = 0; q < 4; q++)
						if (b + q < nbytes)
							v |= (uint32_t) h[b + q] << (8 * q);
				}
				lbits[la(j)] = v;
				if ((j & 63u) == 0 && j)
					lbits[la(j) - 1] = v;
			}
		}
		__syncthreads();
		HSTAMP(1); // stage

		// ---- pass 0: from the first bit of the own subsequence (thread 0 of the read's first tile: exact)
		{
			uint32_t c;
			sE[tid] = huff_run<false>(lbits, lut, lut2, l2off, l2bits, a.huff, sub0, sub0 + S, nbits, c, nullptr);
			sS[tid] = HNONE;
			sC[tid] = 0;
		}
		HSTAMP(2); // pass 0
		converge(0, nbits);
		HSTAMP(3); // rounds A

		// ---- hint for the successor; own start from the predecessor's hint
		uint32_t pos0 = 0;
		if (!last && tid == 0) {
			const uint32_t el = sE[HT - 1];
			__hip_atomic_store(a.hhint + k, HINT_READY | (el == HEND ? HINT_END : el - TB), __ATOMIC_RELAXED,
					   __HIP_MEMORY_SCOPE_AGENT);
		}
		if (t) {
			if (tid == 0)
				s_hint = poll32(a.hhint + k - 1);
			__syncthreads();
			HSTAMP(4); // hint wait
			const uint32_t hv = s_hint;
			pos0 = (hv & HINT_END) ? HEND : (hv & 0xFFu);
			if (pos0 == HEND) { // the stream ended before this tile
				sS[tid] = HEND;
				sE[tid] = HEND;
				sC[tid] = 0;
			} else {
				converge(pos0, nbits);
			}
		}

		HSTAMP(5); // rounds B
		uint32_t cnt, excl, total;
		uint64_t cum_prev = 0;
		for (;;) {
			// ---- offsets: exclusive prefix of the counts over the workgroup
			__syncthreads();
			cnt = sC[tid];
			uint32_t inc = cnt;
#pragma unroll
			for (int dd = 1; dd < 64; dd <<= 1) {
				const uint32_t t2 = (uint32_t) __shfl_up((int) inc, dd, 64);
				if ((int) lane >= dd)
					inc += t2;
			}
			if (lane == 63)
				wtot[tid >> 6] = inc;
			__syncthreads();
			uint32_t base = 0;
			total = 0;
#pragma unroll
			for (int w2 = 0; w2 < HT / 64; w2++) {
				const uint32_t x = wtot[w2];
				if (w2 < (int) (tid >> 6))
					base += x;
				total += x;
			}
			excl = base + inc - cnt;

			// ---- FINAL chain: wait for the predecessor, publish {end, count} for the successor
			bool redo = false;
			if (tid == 0) {
				uint64_t fin = FIN_READY; // first tile: starts at bit 0, nothing before it
				if (t)
					fin = poll64(a.hfin + k - 1);
				const uint32_t el = sE[HT - 1];
				const uint32_t ptrue = (fin & FIN_END) ? HEND : (uint32_t) ((fin >> 32) & 0xFFu);
				if (t && ptrue != pos0) {
					fin |= 1ull << 61; // the hint was wrong: converge again from the true start
				} else if (!last) {
					const uint64_t cum = (fin & 0xFFFFFFFFull) + total;
					__hip_atomic_store(a.hfin + k,
							   FIN_READY | (el == HEND ? FIN_END : ((uint64_t) (el - TB) << 32)) |
								   (cum > 0xFFFFFFFFull ? 0xFFFFFFFFull : cum),
							   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				s_fin = fin;
			}
			__syncthreads();
			HSTAMP(6); // prefix + final wait
			const uint64_t fin = s_fin;
			cum_prev = fin & 0xFFFFFFFFull;
			redo = (fin >> 61) & 1ull;
			if (!redo)
				break;
			HCOUNT(10, 1);
			pos0 = (fin & FIN_END) ? HEND : (uint32_t) ((fin >> 32) & 0xFFu);
			if (pos0 == HEND) {
				sS[tid] = HEND;
				sE[tid] = HEND;
				sC[tid] = 0;
			} else {
				converge(pos0, nbits);
			}
		}

		// ---- output: this tile delivers symbols [cum_prev, cum_prev + total) of the read, cut at `want`
		const uint32_t o0 = cum_prev < want ? (uint32_t) cum_prev : want;
		const uint32_t room = want - o0;
		const uint32_t take = total < room ? total : room;
		uint8_t *dst = low + o0;
		const uint32_t skip = (uint32_t) ((uintptr_t) dst & 15);
		uint8_t *g = dst - skip; // 16-byte aligned; obuf[b] <-> g[b]
		{
			const uint32_t nmine = excl < take ? (cnt < take - excl ? cnt : take - excl) : 0u;
			const uint32_t *sl = reinterpret_cast<const uint32_t *>(myslot);
			uint8_t *d = obuf + skip + excl;
			for (uint32_t q = 0; q < nmine; q += 4) {
				const uint32_t w = sl[q >> 2];
				d[q] = (uint8_t) w;
				if (q + 1 < nmine)
					d[q + 1] = (uint8_t) (w >> 8);
				if (q + 2 < nmine)
					d[q + 2] = (uint8_t) (w >> 16);
				if (q + 3 < nmine)
					d[q + 3] = (uint8_t) (w >> 24);
			}
		}
		__syncthreads();
		{
			const uint32_t tot = skip + take;
			for (uint32_t c = tid; c * 16 < tot; c += HT) {
				const uint32_t lo = c * 16;
				if (lo >= skip && lo + 16 <= tot) {
					reinterpret_cast<uint4 *>(g)[c] = reinterpret_cast<const uint4 *>(obuf)[c];
				} else {
					const uint32_t b0 = lo > skip ? lo : skip;
					const uint32_t b1 = lo + 16 < tot ? lo + 16 : tot;
					for (uint32_t b = b0; b < b1; b++)
						g[b] = obuf[b];
				}
			}
		}
		if (last && tid == 0)
			a.meta[r].nlow = o0 + take; // what huffman_decode_memory delivered
		HSTAMP(7); // output
	}
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
	for (uint32_t t = 0; t < nt; t++) {
		const uint32_t k = base + t;
		if (k >= a.max_htiles)
			break; // cannot happen: max_htiles is the same bound summed over the slots
		a.htiles[k] = make_uint2(r, t | (t + 1 == nt ? 0x80000000u : 0u));
		a.hhint[k] = 0;
		a.hfin[k] = 0;
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
