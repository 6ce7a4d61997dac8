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

#ifndef ZS_MAXLEN
#define ZS_MAXLEN 11
#endif
constexpr int MAXLEN = ZS_MAXLEN;   // longest code the writer makes (zstd's encoder: HUF_TABLELOG_DEFAULT = 11)
constexpr int READ_MAXLEN = 11;     // longest code the device reader takes (12, the format's limit: left to libzstd)
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
	uint16_t parent[512]; // (the device keeps the occurrence counts here until the leaves' weights are in w: 8 reads' scratch
	                      // per CU instead of 6 - all 8192 reads of a batch in one round)
	union {               // the depths are done with when the weights are written
		uint8_t depth[512];
		struct {
			uint8_t weight[256];
			uint16_t state_tab[64];
			uint8_t spread[64];
		};
	};
	uint16_t cum[16];
	// small tables (kept here rather than in locals: indexed at run time, and a GPU lane's
	// locals of that kind live in scratch memory)
	uint32_t bl[MAXLEN + 2], wc[13], norm[13], dfs[13], dnb[13];
	uint32_t fail;
};

ZS_FN uint32_t highbit(uint32_t v) // position of the highest set bit; 0 for 0
{
	return v ? 31u - (uint32_t) __builtin_clz(v) : 0u; // (a loop over the bits here was ten steps per cell of every FSE table)
}

// little-endian bit writer into a small byte buffer: bits collect in a register, whole bytes
// are stored (no read-modify-write of the buffer)
struct Bits {
	uint8_t *p;
	uint32_t cap, bytes; // bytes stored
	uint64_t acc;
	uint32_t nacc;       // bits waiting in acc (< 8 between calls)
	uint32_t pos;        // bits written in all
	bool over;
};
ZS_FN void bits_init(Bits &b, uint8_t *p, uint32_t cap)
{
	b.p = p;
	b.cap = cap;
	b.bytes = 0;
	b.acc = 0;
	b.nacc = 0;
	b.pos = 0;
	b.over = false;
}
ZS_FN void bits_add(Bits &b, uint32_t v, uint32_t n) // n <= 32
{
	b.acc |= (uint64_t) (n < 32 ? v & ((1u << n) - 1u) : v) << b.nacc;
	b.nacc += n;
	b.pos += n;
	while (b.nacc >= 8) {
		if (b.bytes < b.cap)
			b.p[b.bytes] = (uint8_t) b.acc;
		else
			b.over = true;
		b.bytes++;
		b.acc >>= 8;
		b.nacc -= 8;
	}
}
ZS_FN void bits_align(Bits &b) // pad the byte in progress with zeros
{
	if (b.nacc)
		bits_add(b, 0, 8 - b.nacc);
}

// How the lanes of a wave share build_table: the host runs it with one "lane".
struct Solo {
	ZS_FN uint32_t lane() const { return 0; }
	ZS_FN uint32_t lanes() const { return 1; }
	ZS_FN void sync() const {}
	ZS_FN void inc(uint32_t *p) const { ++*p; }
	ZS_FN void stamp(int) const {} // (the device's diagnostic build: where a wave of k_zs_table spends its time)
};

// cnt[256]: occurrences; order[0..m): the bytes with cnt > 0, ascending by (cnt, byte); m >= 2.
// Every lane of `par` calls it with the same arguments (t, k shared between them); loops over
// bytes / nodes are dealt out to the lanes, the chains that cannot be (the merge of the tree,
// the length limiter, the FSE state machine) run on lane 0.
template <class Par> ZS_FN void build_table(const uint32_t *cnt, const uint8_t *order, uint32_t m, Table &t, Work &k, const Par &par)
{
	const uint32_t me = par.lane(), nl = par.lanes();
	for (uint32_t s = me; s < 256; s += nl) {
		t.code[s] = 0;
		t.len[s] = 0;
	}
	if (me == 0) {
		t.desc_len = 0;
		t.table_log = 0;
		t.ok = 0;
		k.fail = 0;
	}
	if (m < 2 || m > 256)
		return;
	uint32_t *bl = k.bl;
	for (uint32_t i = me; i < m; i += nl)
		k.w[i] = cnt[order[i]];
	for (uint32_t i = me; i <= (uint32_t) MAXLEN + 1; i += nl)
		bl[i] = 0;
	par.sync();
	// ---- Huffman tree by the two-queue method: leaves 0..m-1 (sorted), internal nodes m..2m-2
	const uint32_t root = 2 * m - 2;
	if (me == 0) {
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
		k.parent[root] = (uint16_t) root;
	}
	par.sync();
	par.stamp(1); // the tree
	// ---- depths.  One lane: from the root down (a parent has a larger index than its children).
	// Many lanes: pointer jumping - every node doubles the distance to its marked ancestor,
	// eight times for up to 255 levels.
	if (nl == 1) {
		k.depth[root] = 0;
		for (int n = (int) root - 1; n >= 0; n--) {
			const uint32_t d = (uint32_t) k.depth[k.parent[n]] + 1;
			k.depth[n] = (uint8_t) (d > 255 ? 255 : d);
		}
	} else {
		for (uint32_t n = me; n <= root; n += nl)
			k.depth[n] = n == root ? 0 : 1;
		par.sync();
		for (int it = 0; it < 8; it++) {
			uint16_t aa[8];
			uint8_t dd[8];
			int c = 0;
			for (uint32_t n = me; n <= root && c < 8; n += nl, c++) {
				const uint32_t a = k.parent[n];
				aa[c] = k.parent[a];
				dd[c] = (uint8_t) (k.depth[n] + k.depth[a]);
			}
			par.sync();
			c = 0;
			for (uint32_t n = me; n <= root && c < 8; n += nl, c++) {
				k.parent[n] = aa[c];
				k.depth[n] = dd[c];
			}
			par.sync();
		}
	}
	// bl[d] = bytes with a code of d bits, clamped to MAXLEN.  Lengths are handed out in sorted
	// order below (the rarest byte gets the longest code), so only the counts matter.
	for (uint32_t n = me; n < m; n += nl) {
		const uint32_t d = k.depth[n];
		par.inc(&bl[d > (uint32_t) MAXLEN ? (uint32_t) MAXLEN : d]);
	}
	par.sync();
	par.stamp(2); // depths
	// ---- the clamp oversubscribes the code space by `debt` (units of 2^-MAXLEN).  Pay it back
	// where it is cheapest: making the rarest byte of the d-bit class one bit longer frees
	// 2^(MAXLEN-1-d) units for cnt bits; whatever was freed too much is spent on the most
	// frequent bytes that fit (within 0.05 % of the package-merge optimum on nanopore reads)
	if (me == 0) {
		int64_t debt = -(int64_t) (1u << MAXLEN);
		for (int d = 1; d <= MAXLEN; d++)
			debt += (int64_t) bl[d] << (MAXLEN - d);
		while (debt > 0 && !k.fail) {
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
				if (!bl[best]) {
					k.fail = 1;
					break;
				}
			}
			bl[best]--;
			bl[best + 1]++;
			debt -= (int64_t) 1 << (MAXLEN - 1 - best);
		}
		while (debt < 0 && !k.fail) {
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
			if (!best) {
				k.fail = 1;
				break;
			}
			bl[best]--;
			bl[best - 1]++;
			debt += (int64_t) 1 << (MAXLEN - best);
		}
		// the table log, Kraft equality (a zstd decoder refuses anything else), and where the
		// codes of each weight start in the decoder's table
		uint32_t tl = MAXLEN, total = 0;
		while (tl > 1 && bl[tl] == 0)
			tl--;
		for (uint32_t b = 1; b <= (uint32_t) MAXLEN; b++)
			total += bl[b];
		uint32_t kraft = 0;
		for (uint32_t b = 1; b <= tl; b++)
			kraft += bl[b] << (tl - b);
		if (kraft != (1u << tl) || total != m)
			k.fail = 1;
		t.table_log = tl;
		uint32_t start = 0;
		for (uint32_t wt = 1; wt <= tl; wt++) {
			k.cum[wt] = (uint16_t) start;
			start += bl[tl + 1 - wt] << (wt - 1);
		}
	}
	par.sync();
	par.stamp(3); // the length limit
	if (k.fail)
		return;
	const uint32_t tl = t.table_log;
	// ---- lengths: the rarest bytes get the longest codes
	for (uint32_t i = me; i < m; i += nl) {
		uint32_t bits = MAXLEN, before = 0;
		while (bits > 1 && before + bl[bits] <= i) {
			before += bl[bits];
			bits--;
		}
		t.len[order[i]] = (uint8_t) bits;
	}
	par.sync();
	// ---- codes as the decoder lays out its table (HUF_readDTableX1): by weight = tl + 1 - len
	// ascending, equal weights by byte value ascending; code = table index >> (weight - 1)
	for (uint32_t s = me; s < 256; s += nl) {
		const uint32_t l = t.len[s];
		if (!l)
			continue;
		uint32_t rank = 0;
		for (uint32_t q = 0; q < s; q++)
			rank += t.len[q] == l;
		const uint32_t wt = tl + 1 - l;
		t.code[s] = (uint16_t) ((k.cum[wt] >> (wt - 1)) + rank);
	}
	par.stamp(4); // lengths, codes
	// ---- tree description: the weights of bytes 0 .. last-1 (the last one is implied)
	int last = 255;
	while (!t.len[last])
		last--;
	const uint32_t nw = (uint32_t) last;
	for (uint32_t s = me; s < nw; s += nl)
		k.weight[s] = t.len[s] ? (uint8_t) (tl + 1 - t.len[s]) : 0;
	for (uint32_t v = me; v < 13; v += nl)
		k.wc[v] = 0;
	par.sync();
	if (nw <= 128) { // direct: 4 bits each
		for (uint32_t i = 2 * me; i < nw; i += 2 * nl)
			t.desc[1 + i / 2] = (uint8_t) ((k.weight[i] << 4) | (i + 1 < nw ? k.weight[i + 1] : 0));
		if (me == 0) {
			t.desc[0] = (uint8_t) (127 + nw);
			t.desc_len = 1 + (nw + 1) / 2;
			t.ok = 1;
		}
		par.sync();
		return;
	}
	// ---- FSE-compressed weights (FSE table log 6, the most HUF_readStats accepts)
	constexpr uint32_t FL = 6, FS = 1u << FL;
	uint32_t *wc = k.wc, *norm = k.norm;
	for (uint32_t s = me; s < nw; s += nl)
		par.inc(&wc[k.weight[s]]);
	par.sync();
	if (me != 0) { // the rest is one chain
		par.sync();
		return;
	}
	uint32_t maxw = 12;
	while (maxw && !wc[maxw])
		maxw--;
	bool fine = true;
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
			fine = false; // one weight value only: FSE cannot carry the count (and such data does not shrink)
		if (fine && sum > FS && norm[big] <= sum - FS)
			fine = false;
		if (fine) {
			norm[big] = norm[big] + FS - sum;
			if (norm[big] >= FS)
				fine = false;
		}
	}
	Bits b;
	bits_init(b, t.desc + 1, 127);
	if (fine) { // the normalised counts (FSE_writeNCount)
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
			if (remaining < 1) {
				fine = false;
				break;
			}
			while (remaining < threshold) {
				nbits--;
				threshold >>= 1;
			}
		}
		if (remaining != 1)
			fine = false;
		bits_align(b); // the bit stream starts on the next byte
	}
	if (fine) {
		// ---- the state table (FSE_buildCTable): spread, then the states of each value in table order
		uint32_t *dfs = k.dfs, *dnb = k.dnb; // deltaFindState (+64 to stay unsigned), deltaNbBits
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
			uint32_t st[2] = { 0, 0 };
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
		bits_align(b);
		const uint32_t bytes = b.bytes;
		if (!b.over && bytes <= 127) {
			t.desc[0] = (uint8_t) bytes;
			t.desc_len = 1 + bytes;
			t.ok = 1;
		}
	}
	par.stamp(5); // the description (FSE)
	par.sync();
}

ZS_FN void build_table(const uint32_t *cnt, const uint8_t *order, uint32_t m, Table &t, Work &k)
{
	build_table(cnt, order, m, t, k, Solo());
}

} // namespace zs

// ====================================================================================
// The reading side: tree descriptions and frames (RFC 8878 3.1, 4.2.1).  Host/device code again:
// the device walks a frame with one lane per read (press_zstd.hip), the tests walk frames on
// the host (oracle/zsframe_model.cpp).
namespace zs {

// ---- forward little-endian bit reader (FSE table description); zeros behind the end
struct FwdBits {
	const uint8_t *p;
	uint32_t len, pos;
};
ZS_FN uint32_t fwd_peek(const FwdBits &b, uint32_t n) // n <= 16
{
	const uint32_t q = b.pos >> 3;
	uint32_t w = 0;
	for (uint32_t i = 0; i < 3; i++) // three bytes hold any 16 bits
		if (q + i < b.len)
			w |= (uint32_t) b.p[q + i] << (8 * i);
	return (w >> (b.pos & 7u)) & ((1u << n) - 1u);
}

// ---- backward bit reader (FSE and Huffman streams): the last byte holds the end mark
struct BackBits {
	const uint8_t *p;
	int64_t pos; // bits left below the mark; negative: read past the start
};
ZS_FN bool back_init(BackBits &b, const uint8_t *p, uint32_t len)
{
	if (!len || !p[len - 1])
		return false;
	b.p = p;
	b.pos = 8ll * (len - 1) + highbit(p[len - 1]);
	return true;
}
ZS_FN uint32_t back_read(BackBits &b, uint32_t n) // the n <= 16 bits below pos, the highest first; zeros below bit 0
{
	uint32_t v = 0;
	if (b.pos > 0 && n) {
		const int64_t top = (b.pos - 1) >> 3; // byte of the highest bit wanted
		uint32_t w = 0;                     // bytes top-2 .. top, top in bits 16..23
		for (int i = 0; i < 3; i++)
			if (top - i >= 0)
				w |= (uint32_t) b.p[top - i] << (8 * (2 - i));
		const uint32_t hi = 16 + (uint32_t) ((b.pos - 1) & 7); // position of that bit in w
		v = hi + 1 >= n ? (w >> (hi + 1 - n)) & ((1u << n) - 1u) : (w << (n - hi - 1)) & ((1u << n) - 1u);
	}
	b.pos -= n;
	return v;
}

// ---- the same reader with the stream's next bits in a register (the FSE state machines are chains of "table entry ->
// that many bits -> next state": with three byte loads per read in that chain the tree description of one frame took a
// device wave 0.3 M cycles).  acc holds the next `have` bits in its low bits, the one read first on top; bytes are taken
// in BEHIND a read, so their loads run next to the table look-up of the following step.
struct BackWin {
	const uint8_t *p;
	int32_t pos;   // bits left below the mark; negative: read past the start
	int32_t nextb; // the next byte to take in (descending); < 0: none left
	uint64_t acc;
	uint32_t have;
};
ZS_FN void backwin_fill(BackWin &b)
{
	while (b.have <= 56 && b.nextb >= 0) {
		b.acc = (b.acc << 8) | b.p[b.nextb--];
		b.have += 8;
	}
}
ZS_FN bool backwin_init(BackWin &b, const uint8_t *p, uint32_t len) // len < 2^28
{
	if (!len || !p[len - 1])
		return false;
	const uint32_t h = highbit(p[len - 1]);
	b.p = p;
	b.pos = (int32_t) (8 * (len - 1) + h);
	b.acc = p[len - 1] & ((1u << h) - 1u);
	b.have = h;
	b.nextb = (int32_t) len - 2;
	backwin_fill(b);
	return true;
}
ZS_FN uint32_t backwin_read(BackWin &b, uint32_t n) // as back_read: n <= 16, zeros below bit 0
{
	if (!n)
		return 0;
	const uint32_t mask = (1u << n) - 1u;
	uint32_t v;
	if (b.have >= n) {
		b.have -= n;
		v = (uint32_t) (b.acc >> b.have) & mask;
	} else { // (backwin_fill keeps more than 56 bits while bytes are left: the stream ends here)
		v = (uint32_t) (b.acc << (n - b.have)) & mask;
		b.have = 0;
		b.acc = 0;
	}
	b.pos -= (int32_t) n;
	backwin_fill(b);
	return v;
}

// scratch of the reading side (LDS on the device, where every lane of a wave runs the same walk)
constexpr uint32_t WIN = 128; // frame bytes fetched at a time by walk_frame (a run of 4-byte RLE blocks in one fetch)
// FSE decoding table of one of the three sequence symbol types (RFC 8878 3.1.1.3.2.1): at most 9 bits
struct SeqTab {
	uint8_t sym[512], nb[512];
	uint16_t nw[512];
	uint32_t log;   // table log; 0 with one entry: an RLE table
	uint32_t valid;
};
// (what a frame WITHOUT sequences needs - this library's own frames: under 1 KB, so that a CU holds a wave for every read of
// a batch; a walk with this scratch leaves frames with sequences to a second walk, W_SEQ)
struct ReadWorkLean {
	uint8_t win[WIN];
	uint8_t w[256];
	uint8_t desc[DESC_MAX + 4];
	uint8_t dsym[64];
	uint32_t dtab[64]; // FSE decoding table of the weights: value | bits << 8 | (next state's base) << 16
	uint16_t next[16];
	int norm[16];
};
struct ReadWork : ReadWorkLean {
	// sequences
	SeqTab st[3];      // literal lengths, offsets, match lengths
	int norm2[64];
	uint16_t next2[64];
	uint8_t sdesc[512]; // the sequence section's table descriptions (all three tables: below 400 bytes)
};

// FSE-compressed weights (FSE_decompress with table log <= 6) -> w[0..count); 0: malformed
ZS_FN uint32_t fse_read_weights(const uint8_t *src, uint32_t len, uint8_t *w, uint32_t max_out, ReadWorkLean &k)
{
	int *norm = k.norm;
	uint8_t *dsym = k.dsym;
	uint32_t *dtab = k.dtab;
	uint16_t *next = k.next;
	FwdBits f{ src, len, 0 };
	const uint32_t tl = fwd_peek(f, 4) + 5;
	f.pos += 4;
	if (tl > 6)
		return 0;
	const int ts = 1 << tl;
	int remaining = ts + 1, threshold = ts, nbits = (int) tl + 1;
	uint32_t charnum = 0;
	bool prev0 = false;
	while (remaining > 1 && charnum < 16) {
		if (prev0) {
			uint32_t n0 = charnum;
			while (fwd_peek(f, 16) == 0xFFFFu) {
				n0 += 24;
				f.pos += 16;
				if (f.pos > 8 * len)
					return 0;
			}
			while (fwd_peek(f, 2) == 3) {
				n0 += 3;
				f.pos += 2;
				if (f.pos > 8 * len)
					return 0;
			}
			n0 += fwd_peek(f, 2);
			f.pos += 2;
			if (n0 >= 16)
				return 0;
			while (charnum < n0)
				norm[charnum++] = 0;
		}
		const int max = (2 * threshold - 1) - remaining;
		const int pk = (int) fwd_peek(f, (uint32_t) nbits);
		int count;
		if ((pk & (threshold - 1)) < max) {
			count = pk & (threshold - 1);
			f.pos += (uint32_t) nbits - 1;
		} else {
			count = pk & (2 * threshold - 1);
			if (count >= threshold)
				count -= max;
			f.pos += (uint32_t) nbits;
		}
		count--;
		remaining -= count < 0 ? -count : count;
		norm[charnum++] = count;
		prev0 = count == 0;
		while (remaining < threshold) {
			nbits--;
			threshold >>= 1;
		}
	}
	if (remaining != 1 || f.pos > 8 * len)
		return 0;
	const uint32_t hbytes = (f.pos + 7) / 8;
	// ---- decoding table (FSE_buildDTable)
	int high = ts - 1;
	for (uint32_t s = 0; s < charnum; s++) {
		if (norm[s] == -1) {
			dsym[high--] = (uint8_t) s;
			next[s] = 1;
		} else {
			next[s] = (uint16_t) norm[s];
		}
	}
	{
		const int step = (ts >> 1) + (ts >> 3) + 3, mask = ts - 1;
		int pos = 0;
		for (uint32_t s = 0; s < charnum; s++)
			for (int i = 0; i < norm[s]; i++) {
				dsym[pos] = (uint8_t) s;
				pos = (pos + step) & mask;
				while (pos > high)
					pos = (pos + step) & mask;
			}
		if (pos != 0)
			return 0;
	}
	for (int u = 0; u < ts; u++) {
		const uint32_t s = dsym[u];
		const uint32_t nx = next[s]++;
		const uint32_t nb = tl - highbit(nx);
		dtab[u] = s | (nb << 8) | (((nx << nb) - (uint32_t) ts) << 16);
	}
	// ---- two interleaved states; whoever reads past the start ends it, the other adds its symbol
	BackWin b;
	if (len <= hbytes || !backwin_init(b, src + hbytes, len - hbytes))
		return 0;
	uint32_t e[2]; // the states' table entries
	e[0] = backwin_read(b, tl);
	e[1] = backwin_read(b, tl);
	if (b.pos < 0)
		return 0;
	e[0] = dtab[e[0]];
	e[1] = dtab[e[1]];
	uint32_t out = 0;
	for (int q = 0;; q ^= 1) {
		if (out + 2 > max_out)
			return 0;
		w[out++] = (uint8_t) e[q];
		e[q] = dtab[(e[q] >> 16) + backwin_read(b, (e[q] >> 8) & 0xFFu)];
		if (b.pos < 0) {
			w[out++] = (uint8_t) e[q ^ 1];
			break;
		}
	}
	return out;
}

// tree description at p -> weights of all 256 bytes, the table log; returns the bytes used,
// 0: malformed, 0xFFFFFFFF: valid but beyond what the device decoder holds (table log 12)
// (par: lane() / lanes() / sum() / sync() - how the lanes that run the walk together share the
// loops over the weights; the host has one lane)
template <class Par>
ZS_FN uint32_t read_tree(const uint8_t *p, uint32_t avail, uint8_t *w, uint32_t *table_log, ReadWorkLean &k, const Par &par)
{
	if (!avail)
		return 0;
	const uint32_t hb = p[0];
	const uint32_t me = par.lane(), nl = par.lanes();
	uint32_t n, used;
	if (hb >= 128) {
		n = hb - 127;
		used = 1 + (n + 1) / 2;
		if (used > avail)
			return 0;
		for (uint32_t i = me; i < n; i += nl)
			w[i] = (i & 1) ? (p[1 + i / 2] & 15) : (p[1 + i / 2] >> 4);
	} else {
		used = 1 + hb;
		if (used > avail)
			return 0;
		n = fse_read_weights(p + 1, hb, w, 255, k); // every lane the same chain, the same stores
		if (!n)
			return 0;
	}
	par.sync();
	uint32_t total = 0, bad = 0, ones = 0;
	for (uint32_t i = me; i < n; i += nl) {
		const uint32_t wi = w[i];
		bad += wi > 11;
		total += (1u << (wi & 15)) >> 1;
		ones += wi == 1;
	}
	total = par.sum(total);
	bad = par.sum(bad);
	ones = par.sum(ones);
	if (bad || !total)
		return 0;
	const uint32_t tl = highbit(total) + 1;
	if (tl > 12)
		return 0;
	const uint32_t rest = (1u << tl) - total;
	if (rest != (1u << highbit(rest)))
		return 0;
	const uint32_t lastw = highbit(rest) + 1;
	ones += lastw == 1;
	if (ones < 2 || (ones & 1))
		return 0;
	par.sync();
	for (uint32_t i = n + me; i < 256; i += nl)
		w[i] = i == n ? (uint8_t) lastw : 0;
	par.sync();
	*table_log = tl;
	return tl > (uint32_t) READ_MAXLEN ? 0xFFFFFFFFu : used;
}

// ---- sequences (RFC 8878 3.1.1.3.2): three FSE-coded symbol streams interleaved in one backward bit stream

ZS_FN uint32_t back_read32(BackBits &b, uint32_t n) // n <= 32
{
	if (n <= 16)
		return back_read(b, n);
	const uint32_t hi = back_read(b, n - 16);
	return (hi << 16) | back_read(b, 16);
}

// normalised counts of an FSE table description (FSE_readNCount) at f -> norm[0..*nsym); returns the bytes
// used, 0: malformed
ZS_FN uint32_t fse_read_ncount(const uint8_t *src, uint32_t len, int *norm, uint32_t maxsym, uint32_t maxlog,
			       uint32_t *tlog, uint32_t *nsym)
{
	FwdBits f{ src, len, 0 };
	const uint32_t tl = fwd_peek(f, 4) + 5;
	f.pos += 4;
	if (tl > maxlog)
		return 0;
	const int ts = 1 << tl;
	int remaining = ts + 1, threshold = ts, nbits = (int) tl + 1;
	uint32_t charnum = 0;
	bool prev0 = false;
	while (remaining > 1 && charnum <= maxsym) {
		if (prev0) {
			uint32_t n0 = charnum;
			while (fwd_peek(f, 16) == 0xFFFFu) {
				n0 += 24;
				f.pos += 16;
				if (f.pos > 8 * len)
					return 0;
			}
			while (fwd_peek(f, 2) == 3) {
				n0 += 3;
				f.pos += 2;
				if (f.pos > 8 * len)
					return 0;
			}
			n0 += fwd_peek(f, 2);
			f.pos += 2;
			if (n0 > maxsym)
				return 0;
			while (charnum < n0)
				norm[charnum++] = 0;
		}
		const int max = (2 * threshold - 1) - remaining;
		const int pk = (int) fwd_peek(f, (uint32_t) nbits);
		int count;
		if ((pk & (threshold - 1)) < max) {
			count = pk & (threshold - 1);
			f.pos += (uint32_t) nbits - 1;
		} else {
			count = pk & (2 * threshold - 1);
			if (count >= threshold)
				count -= max;
			f.pos += (uint32_t) nbits;
		}
		count--;
		remaining -= count < 0 ? -count : count;
		norm[charnum++] = count;
		prev0 = count == 0;
		while (remaining < threshold) {
			nbits--;
			threshold >>= 1;
		}
	}
	if (remaining != 1 || f.pos > 8 * len)
		return 0;
	*tlog = tl;
	*nsym = charnum;
	return (f.pos + 7) / 8;
}

// decoding table from normalised counts (FSE_buildDTable); false: the counts do not fill the table
ZS_FN bool fse_build_seqtab(const int *norm, uint32_t nsym, uint32_t tl, SeqTab &t, uint16_t *next)
{
	const int ts = 1 << tl;
	int high = ts - 1;
	for (uint32_t s = 0; s < nsym; s++) {
		if (norm[s] == -1) {
			t.sym[high--] = (uint8_t) s;
			next[s] = 1;
		} else {
			next[s] = (uint16_t) norm[s];
		}
	}
	const int step = (ts >> 1) + (ts >> 3) + 3, mask = ts - 1;
	int pos = 0;
	for (uint32_t s = 0; s < nsym; s++)
		for (int i = 0; i < norm[s]; i++) {
			t.sym[pos] = (uint8_t) s;
			pos = (pos + step) & mask;
			while (pos > high)
				pos = (pos + step) & mask;
		}
	if (pos != 0)
		return false;
	for (int u = 0; u < ts; u++) {
		const uint32_t sy = t.sym[u];
		const uint32_t nx = next[sy]++;
		const uint32_t nb = tl - highbit(nx);
		t.nb[u] = (uint8_t) nb;
		t.nw[u] = (uint16_t) ((nx << nb) - (uint32_t) ts);
	}
	t.log = tl;
	t.valid = 1;
	return true;
}

// the predefined distributions (RFC 8878 3.1.1.3.2.2.1/2/3)
ZS_FN int seq_default_norm(int which, uint32_t s)
{
	if (which == 0) { // literal lengths, 36 symbols, log 6
		return s == 0 ? 4 : s == 1 ? 3 : s <= 12 ? 2 : s <= 15 ? 1 : s <= 24 ? 2 : s == 25 ? 3 : s == 26 ? 2 : s <= 31 ? 1 : -1;
	}
	if (which == 1) // offsets, 29 symbols, log 5
		return s <= 5 ? 1 : s <= 8 ? 2 : s <= 23 ? 1 : -1;
	// match lengths, 53 symbols, log 6
	return s == 0 ? 1 : s == 1 ? 4 : s == 2 ? 3 : s <= 8 ? 2 : s <= 45 ? 1 : -1;
}

// base value and extra bits of a literal-length / match-length code
ZS_FN uint32_t ll_base(uint32_t c, uint32_t *bits)
{
	if (c < 16) {
		*bits = 0;
		return c;
	}
	if (c < 20) {
		*bits = 1;
		return 16 + 2 * (c - 16);
	}
	if (c < 22) {
		*bits = 2;
		return 24 + 4 * (c - 20);
	}
	if (c < 24) {
		*bits = 3;
		return 32 + 8 * (c - 22);
	}
	if (c == 24) {
		*bits = 4;
		return 48;
	}
	*bits = c - 19; // 25 -> 6 bits, base 64; 26 -> 7, 128; ... 35 -> 16, 65536
	return 1u << (c - 19);
}
ZS_FN uint32_t ml_base(uint32_t c, uint32_t *bits)
{
	if (c < 32) {
		*bits = 0;
		return c + 3;
	}
	if (c < 36) {
		*bits = 1;
		return 35 + 2 * (c - 32);
	}
	if (c < 38) {
		*bits = 2;
		return 43 + 4 * (c - 36);
	}
	if (c < 40) {
		*bits = 3;
		return 51 + 8 * (c - 38);
	}
	if (c < 42) {
		*bits = 4;
		return 67 + 16 * (c - 40);
	}
	if (c == 42) {
		*bits = 5;
		return 99;
	}
	*bits = c - 36; // 43 -> 7 bits, base 131; 44 -> 8, 259; ... 52 -> 16, 65539
	return (1u << (c - 36)) + 3;
}

// One table of a sequence section: mode 0 predefined, 1 RLE, 2 FSE description, 3 the table of the block
// before.  p / avail: the section bytes from here on; returns the bytes used, 0xFFFFFFFF: malformed.
ZS_FN uint32_t seq_table(int which, uint32_t mode, const uint8_t *p, uint32_t avail, ReadWork &k)
{
	const uint32_t maxsym = which == 0 ? 35 : which == 1 ? 31 : 52;
	const uint32_t maxlog = which == 1 ? 8 : 9;
	SeqTab &t = k.st[which];
	if (mode == 0) {
		const uint32_t ns = which == 0 ? 36 : which == 1 ? 29 : 53;
		for (uint32_t s = 0; s < ns; s++)
			k.norm2[s] = seq_default_norm(which, s);
		return fse_build_seqtab(k.norm2, ns, which == 1 ? 5 : 6, t, k.next2) ? 0 : 0xFFFFFFFFu;
	}
	if (mode == 1) {
		if (avail < 1 || p[0] > maxsym)
			return 0xFFFFFFFFu;
		t.sym[0] = p[0];
		t.nb[0] = 0;
		t.nw[0] = 0;
		t.log = 0;
		t.valid = 1;
		return 1;
	}
	if (mode == 2) {
		uint32_t tl, ns;
		const uint32_t used = fse_read_ncount(p, avail, k.norm2, maxsym, maxlog, &tl, &ns);
		if (!used || !fse_build_seqtab(k.norm2, ns, tl, t, k.next2))
			return 0xFFFFFFFFu;
		return used;
	}
	return t.valid ? 0 : 0xFFFFFFFFu; // repeat
}

// ---- frames.  walk_frame() checks a frame and hands its pieces to a sink:
//   sink.fetch(dst, src, n)           n frame bytes for the walk itself (dst: the window / the description buffer)
//   sink.copy(src, dst, n, lit)       n bytes of the frame at offset src are the content at dst (lit: the bytes at
//                                     dst of the frame's LITERALS space instead - a block with sequences)
//   sink.fill(src, dst, n, lit, v)    the byte at src (= v: the walk has it in its window), n times
//   sink.tree(w, tl) -> 0 / W_*       the Huffman table from here on: weights of the 256 bytes, table log
//   sink.huf(src, csize, dst, R, four, lit) -> 0 / W_*   Huffman-coded literals: csize bytes at src (jump table
//                                     first when `four`) are R bytes at dst
//   sink.seq_block(nseq, lit, R, dst) -> 0 / W_*   a block with nseq sequences: its R literals sit at `lit` of the
//                                     literals space, its content starts at dst
//   sink.seq(i, ll, ml, off)          sequence i: ll literals, then ml bytes from `off` bytes back in the content
//   sink.seq_end(tail)                the block's last `tail` literals follow its last match
// Returns the content size, or W_BAD (malformed) / W_HOST (valid zstd this reader leaves to
// libzstd: dictionaries, 12-bit tables, several frames, more sequences than the sink takes).
constexpr int64_t W_BAD = -1, W_HOST = -2, W_SEQ = -3;
#if defined(HUF_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define ZS_STAMP(sink, i) (sink).stamp(i) // diagnostic build: where a wave of k_zs_walk spends its time
#else
#define ZS_STAMP(sink, i) ((void) 0)
#endif

// Frame bytes come through a window of WIN bytes that the sink fills (sink.fetch(dst, src, n):
// on the device a wave's lanes share the loads).  The walk touches a few bytes per block - the
// block header, the literals header behind it and the "no sequences" byte in front of the next
// header - so one fetch per block is the rule.
template <class Sink> struct FrameSrc {
	const uint8_t *f;
	uint64_t len, base;
	bool valid;
	Sink &sink;
	ReadWorkLean &k;
	ZS_FN uint32_t operator[](uint64_t i) // i < len
	{
		if (!valid || i < base || i >= base + WIN) {
			// (a backward reader - the sequences' bit stream - gets the bytes in front of i as well)
			base = valid && i < base ? (i >= WIN - 1 ? i - (WIN - 1) : 0) : i;
			valid = true;
			ZS_STAMP(sink, 7);
			sink.fetch(k.win, f + base, len - base < WIN ? (uint32_t) (len - base) : WIN);
			ZS_STAMP(sink, 0); // window fetches
		}
		return k.win[i - base];
	}
};

// With a ReadWorkLean the walk ends with W_SEQ at the first block that has sequences, before anything of that block is
// handed to the sink (what the blocks in front of it handed over stays valid: a second walk with a ReadWork hands over the
// same pieces again).
template <class Sink, class Work> ZS_FN int64_t walk_frame(const uint8_t *fp, uint64_t len, uint64_t cap, Sink &sink, Work &k)
{
	constexpr bool SEQS = sizeof(Work) > sizeof(ReadWorkLean);
	FrameSrc<Sink> f{ fp, len, 0, false, sink, k };
	if (len < 6 || f[0] != 0x28 || f[1] != 0xB5 || f[2] != 0x2F || f[3] != 0xFD)
		return W_BAD;
	const uint32_t fhd = f[4];
	if (fhd & 8)
		return W_BAD;
	if (fhd & 3)
		return W_HOST; // a dictionary
	const bool single = fhd & 0x20, checksum = fhd & 4;
	uint64_t at = 5;
	if (!single)
		at++; // the window descriptor says nothing a one-pass reader needs
	const uint32_t fcs_flag = fhd >> 6;
	const uint32_t fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : fcs_flag == 1 ? 2 : fcs_flag == 2 ? 4 : 8;
	if (at + fcs_bytes > len)
		return W_BAD;
	uint64_t fcs = 0;
	for (uint32_t i = 0; i < fcs_bytes; i++)
		fcs |= (uint64_t) f[at + i] << (8 * i);
	if (fcs_bytes == 2)
		fcs += 256;
	at += fcs_bytes;
	uint64_t dst = 0, lit_pos = 0; // content so far; literals of the blocks with sequences so far
	uint32_t rep[3] = { 1, 4, 8 };  // repeat offsets (RFC 8878 3.1.1.5)
	bool have_tree = false;
	for (;;) {
		if (at + 3 > len)
			return W_BAD;
		const uint32_t h = (uint32_t) f[at] | ((uint32_t) f[at + 1] << 8) | ((uint32_t) f[at + 2] << 16);
		at += 3;
		const bool last = h & 1;
		const uint32_t type = (h >> 1) & 3, bs = h >> 3;
		if (type == 3 || bs > 131072)
			return W_BAD;
		if (type == 0) {
			if (at + bs > len || dst + bs > cap)
				return W_BAD;
			ZS_STAMP(sink, 7);
			sink.copy(at, dst, bs, false);
			ZS_STAMP(sink, 1); // raw / RLE pieces queued
			at += bs;
			dst += bs;
		} else if (type == 1) {
			if (at + 1 > len || dst + bs > cap)
				return W_BAD;
			ZS_STAMP(sink, 7);
			sink.fill(at, dst, bs, false, f[at]);
			ZS_STAMP(sink, 1);
			at += 1;
			dst += bs;
		} else {
			if (at + bs > len || bs < 2)
				return W_BAD;
			const uint64_t end = at + bs;
			const uint32_t b0 = f[at], lt = b0 & 3, sf = (b0 >> 2) & 3;
			uint32_t lh, R, cs = 0;
			if (lt < 2) {
				lh = sf == 1 ? 2 : sf == 3 ? 3 : 1;
				if (at + lh > end)
					return W_BAD;
				R = lh == 1 ? b0 >> 3 : lh == 2 ? (b0 >> 4) | (f[at + 1] << 4) : (b0 >> 4) | (f[at + 1] << 4) | (f[at + 2] << 12);
				cs = lt == 0 ? R : 1;
			} else {
				lh = sf < 2 ? 3 : sf == 2 ? 4 : 5;
				if (at + lh > end)
					return W_BAD;
				uint64_t v = 0;
				for (uint32_t i = 0; i < lh; i++)
					v |= (uint64_t) f[at + i] << (8 * i);
				const uint32_t kb = sf < 2 ? 10 : sf == 2 ? 14 : 18;
				R = (uint32_t) (v >> 4) & ((1u << kb) - 1);
				cs = (uint32_t) (v >> (4 + kb)) & ((1u << kb) - 1);
			}
			if (R > 131072 || at + lh + cs + 1 > end)
				return W_BAD;
			// ---- the sequences section header behind the literals
			const uint64_t sq = at + lh + cs;
			uint32_t nseq = f[sq], shl = 1;
			if (nseq >= 128) {
				if (sq + 2 > end)
					return W_BAD;
				if (nseq < 255) {
					nseq = ((nseq - 128) << 8) + f[sq + 1];
					shl = 2;
				} else {
					if (sq + 3 > end)
						return W_BAD;
					nseq = f[sq + 1] + (f[sq + 2] << 8) + 0x7F00;
					shl = 3;
				}
			}
			const bool seqs = nseq != 0;
			if (!SEQS && seqs)
				return W_SEQ;
			if (!seqs && sq + 1 != end)
				return W_BAD;
			if (!seqs && dst + R > cap)
				return W_BAD;
			// a block with sequences delivers at least its literals, so the literals of all such blocks fit in
			// `cap` bytes of a valid frame: checked BEFORE the pieces are queued (the literals space is a slot of
			// `cap` bytes and the copy / fill / Huffman kernels carry out whatever was queued)
			if (seqs && lit_pos + R > cap)
				return W_BAD;
			// literals: straight into the content when no sequences follow, else into the frame's
			// literals space (lit = true) from which the sequences copy them
			uint64_t src = at + lh;
			const uint64_t ldst = seqs ? lit_pos : dst;
			if (lt == 0) {
				sink.copy(src, ldst, R, seqs);
			} else if (lt == 1) {
				sink.fill(src, ldst, R, seqs, f[src]);
			} else {
				if (lt == 2) {
					uint8_t *w = k.w;
					uint32_t tl;
					const uint32_t dn = cs < (uint32_t) DESC_MAX ? cs : (uint32_t) DESC_MAX;
					ZS_STAMP(sink, 7);
					sink.fetch(k.desc, fp + src, dn);
					ZS_STAMP(sink, 2); // the tree description fetched
					const uint32_t used = read_tree(k.desc, dn, w, &tl, k, sink);
					ZS_STAMP(sink, 3); // ... read
					if (used == 0)
						return W_BAD;
					if (used == 0xFFFFFFFFu)
						return W_HOST;
					const int64_t e = sink.tree(w, tl);
					ZS_STAMP(sink, 4); // ... stored
					if (e)
						return e;
					src += used;
					cs -= used;
					have_tree = true;
				} else if (!have_tree) {
					return W_BAD;
				}
				const bool four = sf != 0;
				if (four ? (cs < 10 || R < 4) : cs < 1)
					return W_BAD;
				if (R) {
					ZS_STAMP(sink, 7);
					const int64_t e = sink.huf(src, cs, ldst, R, four, seqs);
					ZS_STAMP(sink, 5); // Huffman blocks queued
					if (e)
						return e;
				}
			}
			uint32_t produced = R;
			if constexpr (SEQS) if (seqs) {
				// ---- tables (literal lengths, offsets, match lengths), then the backward bit stream
				uint64_t q = sq + shl;
				if (q + 1 > end)
					return W_BAD;
				const uint32_t modes = f[q++];
				if (modes & 3)
					return W_BAD;
				const uint32_t dn = end - q < sizeof k.sdesc ? (uint32_t) (end - q) : (uint32_t) sizeof k.sdesc;
				ZS_STAMP(sink, 7);
				sink.fetch(k.sdesc, fp + q, dn);
				uint32_t used_all = 0;
				for (int which = 0; which < 3; which++) {
					const uint32_t mode = (modes >> (6 - 2 * which)) & 3;
					const uint32_t used = seq_table(which, mode, k.sdesc + used_all, dn - used_all, k);
					if (used == 0xFFFFFFFFu)
						return W_BAD;
					used_all += used;
					if (used_all > dn)
						return W_BAD;
				}
				q += used_all;
				if (q >= end)
					return W_BAD;
				ZS_STAMP(sink, 1); // (frames with sequences: the three FSE tables)
				const int64_t e0 = sink.seq_block(nseq, lit_pos, R, dst);
				if (e0)
					return e0;
				// the stream is read through the sink's window, byte-wise (sequence sections are small
				// next to the literals); bits come from the top of the last byte down
				const uint64_t sb = q, sl = end - q;
				const uint32_t lastb = f[end - 1];
				if (!lastb)
					return W_BAD;
				int64_t bpos = 8ll * (int64_t) (sl - 1) + highbit(lastb); // bits left below the end mark
				// (the next `have` bits wait in acc, the one read first on top; bytes come in behind a read, through the
				// frame window.  Bit by bit through the window, this reader was most of the walk of a frame with sequences.)
				uint32_t have = highbit(lastb);
				uint64_t acc = lastb & ((1u << have) - 1u);
				int64_t nextb = (int64_t) sl - 2; // the next byte of the section to take in
				auto fill = [&]() {
					while (have <= 56 && nextb >= 0) {
						acc = (acc << 8) | f[sb + (uint64_t) nextb];
						nextb--;
						have += 8;
					}
				};
				fill();
				auto rd = [&](uint32_t n) -> uint32_t { // the n <= 32 bits below bpos, highest first; zeros below bit 0
					if (!n)
						return 0;
					const uint32_t mask = n < 32 ? (1u << n) - 1u : 0xFFFFFFFFu;
					uint32_t v;
					if (have >= n) {
						have -= n;
						v = (uint32_t) (acc >> have) & mask;
					} else { // (fill keeps more than 56 bits while bytes are left: the section ends here)
						v = (uint32_t) (acc << (n - have)) & mask;
						have = 0;
						acc = 0;
					}
					bpos -= n;
					fill();
					return v;
				};
				uint32_t stl = rd(k.st[0].log), sto = rd(k.st[1].log), stm = rd(k.st[2].log);
				if (bpos < 0)
					return W_BAD;
				uint64_t lits_used = 0, out = dst;
				for (uint32_t i = 0; i < nseq; i++) {
					const uint32_t oc = k.st[1].sym[sto], mc = k.st[2].sym[stm], lc = k.st[0].sym[stl];
					if (oc > 31)
						return W_BAD;
					uint32_t ofv = (oc ? (1u << oc) : 1u) + rd(oc);
					uint32_t mb, lb;
					const uint32_t mbase = ml_base(mc, &mb), lbase = ll_base(lc, &lb);
					const uint32_t ml = mbase + rd(mb);
					const uint32_t ll = lbase + rd(lb);
					// repeat offsets (RFC 8878 3.1.1.5)
					uint32_t off;
					if (ofv > 3) {
						off = ofv - 3;
						rep[2] = rep[1];
						rep[1] = rep[0];
						rep[0] = off;
					} else {
						const uint32_t idx = ofv - (ll ? 1 : 0); // 0 .. 3
						if (idx == 0) {
							off = rep[0];
						} else {
							off = idx == 3 ? rep[0] - 1 : rep[idx];
							if (idx != 1)
								rep[2] = rep[1];
							rep[1] = rep[0];
							rep[0] = off;
						}
					}
					if (bpos < 0 || off == 0)
						return W_BAD;
					lits_used += ll;
					if (lits_used > R || off > out + ll || out + ll + ml > cap) // (off reaches at most to the frame's first byte)
						return W_BAD;
					sink.seq(i, ll, ml, off);
					out += (uint64_t) ll + ml;
					if (i + 1 < nseq) { // the states move on: literal lengths, match lengths, offsets
						stl = k.st[0].nw[stl] + rd(k.st[0].nb[stl]);
						stm = k.st[2].nw[stm] + rd(k.st[2].nb[stm]);
						sto = k.st[1].nw[sto] + rd(k.st[1].nb[sto]);
					}
				}
				if (bpos != 0)
					return W_BAD;
				// what is left of the literals follows the last match
				const uint64_t tail = R - lits_used;
				if (out + tail > cap)
					return W_BAD;
				sink.seq_end((uint32_t) tail);
				ZS_STAMP(sink, 6); // (frames with sequences: the sequences read)
				if (out + tail - dst > 131072) // Block_Maximum_Size also bounds what a block delivers (RFC 8878 3.1.1.2)
					return W_BAD;
				produced = (uint32_t) (out + tail - dst);
				lit_pos += R;
			}
			at = end;
			dst += produced;
		}
		if (last)
			break;
	}
	if (checksum)
		at += 4; // XXH64 of the content: not verified here
	if (at > len)
		return W_BAD;
	if (at != len)
		return W_HOST; // more frames behind this one
	if (fcs_bytes && fcs != dst)
		return W_BAD;
	return (int64_t) dst;
}

// one Huffman stream of a block: k bytes from the len bytes at s; dt[i] = byte | bits << 8
// for the tl-bit prefix i.  false: the stream does not end where it must.
ZS_FN bool huf_decode_stream(const uint8_t *s, uint32_t len, const uint16_t *dt, uint32_t tl, uint8_t *out, uint32_t k)
{
	BackBits b;
	if (!back_init(b, s, len))
		return false;
	for (uint32_t i = 0; i < k; i++) {
		BackBits pk = b;
		const uint32_t e = dt[back_read(pk, tl)];
		out[i] = (uint8_t) e;
		b.pos -= e >> 8;
	}
	return b.pos == 0;
}

// decoding table from the weights (HUF_readDTableX1's order): dt[0 .. 1 << tl)
ZS_FN void huf_build_dtable(const uint8_t *w, uint32_t tl, uint16_t *dt)
{
	uint32_t start[13], cnt[13];
	for (int i = 0; i < 13; i++)
		cnt[i] = 0;
	for (int s = 0; s < 256; s++)
		cnt[w[s]]++;
	uint32_t at = 0;
	for (uint32_t x = 1; x <= tl; x++) {
		start[x] = at;
		at += cnt[x] << (x - 1);
	}
	for (int s = 0; s < 256; s++) {
		const uint32_t x = w[s];
		if (!x)
			continue;
		const uint32_t n = 1u << (x - 1), e = (uint32_t) s | ((tl + 1 - x) << 8);
		for (uint32_t i = 0; i < n; i++)
			dt[start[x] + i] = (uint16_t) e;
		start[x] += n;
	}
}

} // namespace zs
