// zs_table.h - the per-read Huffman table of the zstd frames this library writes
// (zstd_svb_zd in the batch API, SURVEY.md 8f-3: "zstd-compatible GPU entropy stage producing
// stock-decodable frames").
//
// The reference hands the svb-zd stream to libzstd (press.c:1865 ZSTD_compress, level 1); what
// must hold is the FORMAT (RFC 8878): any zstd decoder - the reference's zstd_svb_zd_depress_16
// included - has to get the same svb-zd stream back.  A frame made here holds no sequences:
// RLE blocks for the (almost constant) key bytes and Huffman-coded literal blocks of 16 KiB
// for the data bytes, four independent bit streams each, "treeless" after the first - thousands
// of short independent streams per batch, which is what a GPU can encode and decode in parallel.
//
// This header is the serial part: code lengths limited to 11 bits (the limit zstd's own
// encoder uses), the canonical codes in the order the zstd DECODER derives them from the
// weights (RFC 8878 4.2.1), and the tree description (4.2.1.1: direct 4-bit weights, or
// FSE-compressed weights, 4.1).  Plain C++ for both sides: the device runs it on one lane per
// read (press_zstd.hip), tests/ run it on the host against libzstd.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZS_FN __host__ __device__ inline
#else
#define ZS_FN inline
#endif

namespace zs {

constexpr int MAXLEN = 11;          // longest code (zstd's encoder: HUF_TABLELOG_DEFAULT)
constexpr int DESC_MAX = 132;       // header byte + at most 127 bytes (FSE) / 64 bytes (direct)
constexpr uint32_t BLOCK_LITS = 16384; // data bytes per Huffman block
constexpr uint32_t MIN_HUF_LITS = 64;  // shorter tails are stored raw

struct Table {
	uint16_t code[256];
	uint8_t len[256]; // 0: the byte does not occur
	uint8_t desc[DESC_MAX];
	uint32_t desc_len;
	uint32_t table_log;
	uint32_t ok; // 0: no usable table (the data is stored raw)
	uint32_t pad;
};

// scratch of build_table (LDS on the device)
struct Work {
	uint32_t w[512];     // node weights
	uint16_t parent[512];
	uint8_t depth[512];
	uint8_t weight[256];
	uint16_t cum[16];
	uint16_t state_tab[64];
	uint8_t spread[64];
};

ZS_FN uint32_t highbit(uint32_t v)
{
	uint32_t r = 0;
	while (v >>= 1)
		r++;
	return r;
}

// little-endian bit writer into a small byte buffer
struct Bits {
	uint8_t *p;
	uint32_t cap, pos; // pos in bits
	bool over;
};
ZS_FN void bits_init(Bits &b, uint8_t *p, uint32_t cap)
{
	b.p = p;
	b.cap = cap;
	b.pos = 0;
	b.over = false;
	for (uint32_t i = 0; i < cap; i++)
		p[i] = 0;
}
ZS_FN void bits_add(Bits &b, uint32_t v, uint32_t n)
{
	for (uint32_t i = 0; i < n; i++) {
		const uint32_t q = b.pos + i;
		if ((q >> 3) >= b.cap) {
			b.over = true;
			break;
		}
		if ((v >> i) & 1u)
			b.p[q >> 3] |= (uint8_t) (1u << (q & 7u));
	}
	b.pos += n;
}

// cnt[256]: occurrences; order[0..m): the bytes with cnt > 0, ascending by (cnt, byte); m >= 2.
ZS_FN void build_table(const uint32_t *cnt, const uint8_t *order, uint32_t m, Table &t, Work &k)
{
	for (int s = 0; s < 256; s++) {
		t.code[s] = 0;
		t.len[s] = 0;
	}
	t.desc_len = 0;
	t.table_log = 0;
	t.ok = 0;
	if (m < 2 || m > 256)
		return;
	// ---- Huffman tree by the two-queue method: leaves 0..m-1 (sorted), internal nodes m..2m-2
	for (uint32_t i = 0; i < m; i++)
		k.w[i] = cnt[order[i]];
	uint32_t qi = 0, qn = m, next = m;
	while (next < 2 * m - 1) {
		uint32_t pick[2];
		for (int j = 0; j < 2; j++) {
			const bool leaf = qi < m && (qn >= next || k.w[qi] <= k.w[qn]);
			pick[j] = leaf ? qi++ : qn++;
		}
		k.w[next] = k.w[pick[0]] + k.w[pick[1]];
		k.parent[pick[0]] = (uint16_t) next;
		k.parent[pick[1]] = (uint16_t) next;
		next++;
	}
	// ---- depths, clamped to MAXLEN; bl[d] = bytes with a code of d bits.  Lengths are handed
	// out in sorted order below (the rarest byte gets the longest code), so only the counts matter.
	uint32_t bl[MAXLEN + 2];
	for (int i = 0; i <= MAXLEN + 1; i++)
		bl[i] = 0;
	k.depth[2 * m - 2] = 0;
	for (int n = (int) (2 * m - 3); n >= 0; n--) {
		const uint32_t d = (uint32_t) k.depth[k.parent[n]] + 1;
		k.depth[n] = (uint8_t) (d > 255 ? 255 : d);
		if ((uint32_t) n < m)
			bl[d > (uint32_t) MAXLEN ? (uint32_t) MAXLEN : d]++;
	}
	// ---- the clamp oversubscribes the code space by `debt` (units of 2^-MAXLEN).  Pay it back
	// where it is cheapest: making the rarest byte of the d-bit class one bit longer frees
	// 2^(MAXLEN-1-d) units for cnt bits; whatever was freed too much is spent on the most
	// frequent bytes that fit (within 0.05 % of the package-merge optimum on nanopore reads)
	{
		int64_t debt = -(int64_t) (1u << MAXLEN);
		for (int d = 1; d <= MAXLEN; d++)
			debt += (int64_t) bl[d] << (MAXLEN - d);
		while (debt > 0) {
			int best = 0;
			uint64_t bc = 0, bf = 1;
			uint32_t first = 0; // index (sorted order) of the rarest byte of class d
			for (int d = MAXLEN; d >= 1; d--) {
				if (d < MAXLEN && bl[d]) {
					const uint64_t fr = 1ull << (MAXLEN - 1 - d);
					const uint64_t c = k.w[first];
					if ((int64_t) fr <= debt && (!best || c * bf < bc * fr)) {
						best = d;
						bc = c;
						bf = fr;
					}
				}
				first += bl[d];
			}
			if (!best) { // nothing fits: the smallest step there is
				best = MAXLEN - 1;
				while (best > 1 && !bl[best])
					best--;
				if (!bl[best])
					return;
			}
			bl[best]--;
			bl[best + 1]++;
			debt -= (int64_t) 1 << (MAXLEN - 1 - best);
		}
		while (debt < 0) {
			int best = 0;
			uint64_t bc = 0, bn = 1;
			uint32_t end = 0; // one past the most frequent byte of class d
			for (int d = MAXLEN; d >= 2; d--) {
				end += bl[d];
				if (bl[d]) {
					const uint64_t need = 1ull << (MAXLEN - d);
					const uint64_t c = k.w[end - 1];
					if ((int64_t) need <= -debt && (!best || c * bn > bc * need)) {
						best = d;
						bc = c;
						bn = need;
					}
				}
			}
			if (!best)
				return;
			bl[best]--;
			bl[best - 1]++;
			debt += (int64_t) 1 << (MAXLEN - best);
		}
	}
	// ---- lengths: the rarest bytes get the longest codes
	{
		uint32_t idx = 0;
		for (int bits = MAXLEN; bits >= 1; bits--)
			for (uint32_t j = 0; j < bl[bits]; j++)
				t.len[order[idx++]] = (uint8_t) bits;
		if (idx != m)
			return;
	}
	uint32_t tl = MAXLEN;
	while (tl > 1 && bl[tl] == 0)
		tl--;
	t.table_log = tl;
	{ // Kraft equality (a zstd decoder refuses anything else)
		uint32_t kraft = 0;
		for (uint32_t b = 1; b <= tl; b++)
			kraft += bl[b] << (tl - b);
		if (kraft != (1u << tl))
			return;
	}
	// ---- codes as the decoder lays out its table (HUF_readDTableX1): by weight = tl + 1 - len
	// ascending, equal weights by byte value ascending; code = table index >> (weight - 1)
	{
		uint32_t start = 0;
		for (uint32_t wt = 1; wt <= tl; wt++) {
			k.cum[wt] = (uint16_t) start;
			start += bl[tl + 1 - wt] << (wt - 1);
		}
		for (int s = 0; s < 256; s++) {
			if (!t.len[s])
				continue;
			const uint32_t wt = tl + 1 - t.len[s];
			t.code[s] = (uint16_t) (k.cum[wt] >> (wt - 1));
			k.cum[wt] = (uint16_t) (k.cum[wt] + (1u << (wt - 1)));
		}
	}
	// ---- tree description: the weights of bytes 0 .. last-1 (the last one is implied)
	int last = 255;
	while (!t.len[last])
		last--;
	const uint32_t nw = (uint32_t) last;
	for (uint32_t s = 0; s < nw; s++)
		k.weight[s] = t.len[s] ? (uint8_t) (tl + 1 - t.len[s]) : 0;
	if (nw <= 128) { // direct: 4 bits each
		t.desc[0] = (uint8_t) (127 + nw);
		for (uint32_t i = 0; i < nw; i += 2)
			t.desc[1 + i / 2] = (uint8_t) ((k.weight[i] << 4) | (i + 1 < nw ? k.weight[i + 1] : 0));
		t.desc_len = 1 + (nw + 1) / 2;
		t.ok = 1;
		return;
	}
	// ---- FSE-compressed weights (FSE table log 6, the most HUF_readStats accepts)
	constexpr uint32_t FL = 6, FS = 1u << FL;
	uint32_t wc[13], norm[13];
	for (int v = 0; v < 13; v++)
		wc[v] = 0;
	uint32_t maxw = 0;
	for (uint32_t s = 0; s < nw; s++) {
		wc[k.weight[s]]++;
		if (k.weight[s] > maxw)
			maxw = k.weight[s];
	}
	{
		uint32_t sum = 0, big = 0, distinct = 0;
		for (uint32_t v = 0; v <= maxw; v++) {
			norm[v] = wc[v] ? (wc[v] * FS / nw ? wc[v] * FS / nw : 1u) : 0u;
			sum += norm[v];
			distinct += wc[v] != 0;
			if (wc[v] > wc[big])
				big = v;
		}
		if (distinct < 2)
			return; // one weight value only: FSE cannot carry the count (and such data does not shrink)
		if (sum > FS && norm[big] <= sum - FS)
			return;
		norm[big] = norm[big] + FS - sum;
		if (norm[big] >= FS)
			return;
	}
	Bits b;
	bits_init(b, t.desc + 1, 127);
	{ // the normalised counts (FSE_writeNCount)
		bits_add(b, FL - 5, 4);
		int remaining = (int) FS + 1, threshold = (int) FS, nbits = (int) FL + 1;
		uint32_t sym = 0;
		bool prev0 = false;
		while (sym <= maxw && remaining > 1) {
			if (prev0) {
				uint32_t start = sym;
				while (sym <= maxw && !norm[sym])
					sym++;
				if (sym > maxw)
					break;
				while (sym >= start + 24) {
					start += 24;
					bits_add(b, 0xFFFFu, 16);
				}
				while (sym >= start + 3) {
					start += 3;
					bits_add(b, 3, 2);
				}
				bits_add(b, sym - start, 2);
			}
			int count = (int) norm[sym++];
			const int max = (2 * threshold - 1) - remaining;
			remaining -= count;
			count++;
			if (count >= threshold)
				count += max;
			bits_add(b, (uint32_t) count, (uint32_t) (nbits - (count < max ? 1 : 0)));
			prev0 = count == 1;
			if (remaining < 1)
				return;
			while (remaining < threshold) {
				nbits--;
				threshold >>= 1;
			}
		}
		if (remaining != 1)
			return;
		b.pos = (b.pos + 7) & ~7u; // the bit stream starts on the next byte
	}
	// ---- the state table (FSE_buildCTable): spread, then the states of each value in table order
	uint32_t dfs[13], dnb[13]; // deltaFindState (+64 to stay unsigned), deltaNbBits
	{
		const uint32_t step = (FS >> 1) + (FS >> 3) + 3;
		uint32_t pos = 0, total = 0;
		for (uint32_t v = 0; v <= maxw; v++) {
			for (uint32_t i = 0; i < norm[v]; i++) {
				k.spread[pos] = (uint8_t) v;
				pos = (pos + step) & (FS - 1);
			}
			k.cum[v] = (uint16_t) total;
			if (norm[v]) {
				const uint32_t mbo = norm[v] > 1 ? FL - highbit(norm[v] - 1) : FL;
				dnb[v] = (mbo << 16) - (norm[v] > 1 ? norm[v] << mbo : FS);
				dfs[v] = total + FS - norm[v];
			}
			total += norm[v];
		}
		for (uint32_t u = 0; u < FS; u++) {
			const uint32_t v = k.spread[u];
			k.state_tab[k.cum[v]++] = (uint16_t) (FS + u);
		}
	}
	// ---- the weights backwards through two alternating states (FSE_compress_usingCTable):
	// weight i lives in state (i & 1); the last weight of each state costs no bits
	{
		uint32_t st[2];
		bool have[2] = { false, false };
		for (int i = (int) nw - 1; i >= 0; i--) {
			const uint32_t v = k.weight[i];
			const int q = i & 1;
			if (!have[q]) {
				const uint32_t nbo = (dnb[v] + (1u << 15)) >> 16;
				const uint32_t val = (nbo << 16) - dnb[v];
				st[q] = k.state_tab[(val >> nbo) + dfs[v] - FS];
				have[q] = true;
			} else {
				const uint32_t nbo = (st[q] + dnb[v]) >> 16;
				bits_add(b, st[q] & ((1u << nbo) - 1u), nbo);
				st[q] = k.state_tab[(st[q] >> nbo) + dfs[v] - FS];
			}
		}
		bits_add(b, st[1] & (FS - 1), FL); // FSE_flushCState: state 2, then state 1
		bits_add(b, st[0] & (FS - 1), FL);
		bits_add(b, 1, 1); // end mark
	}
	const uint32_t bytes = (b.pos + 7) / 8;
	if (b.over || bytes > 127)
		return;
	t.desc[0] = (uint8_t) bytes;
	t.desc_len = 1 + bytes;
	t.ok = 1;
}

} // namespace zs
