// zsframe_model.cpp - TEST INFRASTRUCTURE ONLY.  A serial CPU model of the zstd frame writer of
// honours_amd/csrc/press_zstd.hip.  Not a restatement of the reference - the
// reference hands the svb-zd stream to libzstd (press.c:1865) and any valid zstd frame of that
// stream is a correct result.  The tests use this model two ways: libzstd must decode what it
// writes back to the stream (here, without a GPU), and the device must write the same bytes.
// The table construction is the product's own header (zs_table.h), compiled for the host.
#include "../honours_amd/csrc/zs_table.h"
#include <algorithm>
#include <cstring>
#include <vector>

namespace {

struct Out {
	uint8_t *p;
	uint64_t cap, pos;
	bool over;
	void put(const void *s, uint64_t n)
	{
		if (pos + n > cap)
			over = true;
		else
			memcpy(p + pos, s, n);
		pos += n;
	}
	void block_header(bool last, uint32_t type, uint32_t size)
	{
		const uint32_t h = (last ? 1u : 0u) | (type << 1) | (size << 3);
		put(&h, 3);
	}
};

void frame_header(Out &o, uint64_t L)
{
	const uint8_t h[5] = { 0x28, 0xB5, 0x2F, 0xFD, 0xA0 }; // magic; single segment, 4-byte content size
	o.put(h, 5);
	const uint32_t l = (uint32_t) L;
	o.put(&l, 4);
}

uint64_t raw_frame(const uint8_t *S, uint64_t L, uint8_t *out, uint64_t cap)
{
	Out o{ out, cap, 0, false };
	frame_header(o, L);
	uint64_t at = 0;
	do {
		const uint64_t len = std::min<uint64_t>(L - at, 131072);
		o.block_header(at + len == L, 0, (uint32_t) len);
		o.put(S + at, len);
		at += len;
	} while (at < L);
	return o.over ? 0 : o.pos;
}

// one bit stream: the bytes back to front, the first byte's code on top, then the end mark
uint32_t stream_bits(const zs::Table &t, const uint8_t *s, uint32_t k)
{
	uint32_t bits = 0;
	for (uint32_t i = 0; i < k; i++)
		bits += t.len[s[i]];
	return bits;
}
void stream_write(const zs::Table &t, const uint8_t *s, uint32_t k, std::vector<uint8_t> &dst)
{
	const uint32_t bits = stream_bits(t, s, k);
	const size_t base = dst.size();
	dst.resize(base + bits / 8 + 1, 0);
	uint32_t pos = 0;
	for (uint32_t i = k; i-- > 0;) {
		const uint32_t c = t.code[s[i]], l = t.len[s[i]];
		for (uint32_t b = 0; b < l; b++, pos++)
			if ((c >> b) & 1u)
				dst[base + (pos >> 3)] |= (uint8_t) (1u << (pos & 7));
	}
	dst[base + (pos >> 3)] |= (uint8_t) (1u << (pos & 7));
}

} // namespace

extern "C" {

// cnt[256] -> table (for direct tests of zs_table.h); returns t.ok
int zsm_table(const uint32_t *cnt, zs::Table *t)
{
	uint8_t order[256];
	uint32_t m = 0;
	for (int s = 0; s < 256; s++)
		if (cnt[s])
			order[m++] = (uint8_t) s;
	std::sort(order, order + m, [&](uint8_t a, uint8_t b) { return cnt[a] != cnt[b] ? cnt[a] < cnt[b] : a < b; });
	zs::Work k;
	zs::build_table(cnt, order, m, *t, k);
	return (int) t->ok;
}

// S = [u32 n][keys (n+3)/4][data] (the buffer the reference gives to ZSTD_compress, press.c:1860)
// -> frame; returns its size, 0 when it does not fit
uint64_t zsm_frame_p(const uint8_t *S, uint64_t L, uint8_t *out, uint64_t cap, uint64_t plen, uint64_t nk);
// S = [u32 n][keys][data]; kdiv: samples per key byte - 4 for the svb-zd stream (zstd_svb_zd), 8 for
// svb16-zd (zstd_svb12_zd)
uint64_t zsm_frame_k(const uint8_t *S, uint64_t L, uint8_t *out, uint64_t cap, uint32_t kdiv)
{
	if (L < 4)
		return raw_frame(S, L, out, cap);
	uint32_t n;
	memcpy(&n, S, 4);
	return zsm_frame_p(S, L, out, cap, 4, ((uint64_t) n + kdiv - 1) / kdiv);
}
uint64_t zsm_frame(const uint8_t *S, uint64_t L, uint8_t *out, uint64_t cap) { return zsm_frame_k(S, L, out, cap, 4); }

// The general shape: S = [prefix: plen bytes, stored raw][keys: nk bytes, zero runs as RLE blocks]
// [data: Huffman-coded].  zstd_hasgam_vbsse21_zdq (press.c:8554): prefix = the ex-zd header and
// exception section, no keys, data = the one-byte values.
uint64_t zsm_frame_p(const uint8_t *S, uint64_t L, uint8_t *out, uint64_t cap, uint64_t plen, uint64_t nk)
{
	if (plen + nk > L || plen == 0)
		return raw_frame(S, L, out, cap);
	const uint8_t *keys = S + plen, *data = S + plen + nk;
	const uint64_t nd = L - plen - nk;

	// the table of this read: every data byte counts; a lone byte value gets a partner
	uint32_t cnt[256] = { 0 };
	for (uint64_t i = 0; i < nd; i++)
		cnt[data[i]]++;
	uint32_t distinct = 0;
	for (int s = 0; s < 256; s++)
		distinct += cnt[s] != 0;
	if (distinct == 1)
		cnt[cnt[0] ? 1 : 0] = 1;
	zs::Table t;
	memset(&t, 0, sizeof t);
	if (distinct)
		zsm_table(cnt, &t);

	std::vector<uint8_t> buf(L + L / 1000 + 4096);
	Out o{ buf.data(), buf.size(), 0, false };
	frame_header(o, L);
	const uint64_t nblocks = (nd + zs::BLOCK_LITS - 1) / zs::BLOCK_LITS;
	// the prefix
	for (uint64_t at = 0; at < plen; at += 131072) {
		const uint64_t len = std::min<uint64_t>(plen - at, 131072);
		o.block_header(at + len == plen && nk == 0 && nd == 0, 0, (uint32_t) len);
		o.put(S + at, len);
	}
	// the keys: runs of zeros as RLE blocks of at most 128 KiB, every other key byte on its own
	for (uint64_t i = 0; i < nk;) {
		uint64_t j = i;
		uint8_t v = keys[i];
		if (v == 0) {
			while (j < nk && keys[j] == 0 && j - i < 131072)
				j++;
		} else {
			j = i + 1;
		}
		o.block_header(j == nk && nd == 0, 1, (uint32_t) (j - i));
		o.put(&v, 1);
		i = j;
	}
	// the data: 16 KiB blocks, four streams each
	bool have_tree = false;
	for (uint64_t bidx = 0; bidx < nblocks; bidx++) {
		const uint8_t *lit = data + bidx * zs::BLOCK_LITS;
		const uint32_t R = (uint32_t) std::min<uint64_t>(zs::BLOCK_LITS, nd - bidx * zs::BLOCK_LITS);
		const bool last = bidx + 1 == nblocks;
		bool huf = t.ok && R >= zs::MIN_HUF_LITS;
		std::vector<uint8_t> body;
		if (huf) {
			const uint32_t seg = (R + 3) / 4;
			uint32_t sz[4];
			if (!have_tree)
				body.insert(body.end(), t.desc, t.desc + t.desc_len);
			const size_t jump = body.size();
			body.resize(jump + 6);
			for (int q = 0; q < 4; q++) {
				const uint32_t k = q < 3 ? seg : R - 3 * seg;
				const size_t before = body.size();
				stream_write(t, lit + q * seg, k, body);
				sz[q] = (uint32_t) (body.size() - before);
			}
			for (int q = 0; q < 3; q++) {
				body[jump + 2 * q] = (uint8_t) sz[q];
				body[jump + 2 * q + 1] = (uint8_t) (sz[q] >> 8);
			}
			huf = 6 + body.size() < R && sz[0] < 65536 && sz[1] < 65536 && sz[2] < 65536;
		}
		if (huf) {
			o.block_header(last, 2, (uint32_t) (5 + body.size() + 1));
			// literals section header: type (2: with tree, 3: treeless), size format 3 (18 + 18 bits)
			const uint64_t lh = (have_tree ? 3u : 2u) | (3u << 2) | ((uint64_t) R << 4) | ((uint64_t) body.size() << 22);
			o.put(&lh, 5);
			o.put(body.data(), body.size());
			const uint8_t noseq = 0;
			o.put(&noseq, 1);
			have_tree = true;
		} else {
			o.block_header(last, 0, R);
			o.put(lit, R);
		}
	}
	const uint64_t raw = 9 + L + 3 * ((L + 131071) / 131072);
	if (o.over || o.pos >= raw)
		return raw_frame(S, L, out, cap);
	if (o.pos > cap)
		return 0;
	memcpy(out, buf.data(), o.pos);
	return o.pos;
}

// the reading side on the host: walk_frame (the product's) with a sink that decodes at once.
// Returns the content size, -1 malformed, -2 left to libzstd.
struct HostSink {
	const uint8_t *f;
	uint8_t *out;
	uint16_t dt[4096];
	uint32_t tl;
	uint32_t lane() const { return 0; }
	uint32_t lanes() const { return 1; }
	uint32_t sum(uint32_t v) const { return v; }
	void sync() const {}
	void fetch(uint8_t *dst, const uint8_t *src, uint32_t n) { memcpy(dst, src, n); }
	// blocks with sequences: their literals go to a buffer of their own, the sequences copy from it
	// (exactly `cap` bytes on the heap, like the device's literals slot: ASan sees a piece that runs past it)
	uint8_t *lits = nullptr;
	uint64_t cap = 0;
	uint64_t blk_lit = 0, blk_out = 0;
	~HostSink() { free(lits); }
	uint8_t *where(uint64_t dst, uint32_t, bool lit)
	{
		if (!lit)
			return out + dst;
		if (!lits)
			lits = (uint8_t *) malloc(cap ? cap : 1);
		return lits + dst;
	}
	void copy(uint64_t src, uint64_t dst, uint32_t n, bool lit) { memcpy(where(dst, n, lit), f + src, n); }
	void fill(uint64_t src, uint64_t dst, uint32_t n, bool lit, uint32_t v) { (void) src; memset(where(dst, n, lit), (int) v, n); } // (v: the byte at src, handed over by the walk)
	int64_t seq_block(uint32_t, uint64_t lit, uint32_t R, uint64_t dst)
	{
		(void) where(lit, R, true);
		blk_lit = lit;
		blk_out = dst;
		return 0;
	}
	void seq(uint32_t, uint32_t ll, uint32_t ml, uint32_t off)
	{
		memcpy(out + blk_out, lits + blk_lit, ll);
		blk_lit += ll;
		blk_out += ll;
		for (uint32_t i = 0; i < ml; i++, blk_out++) // (byte by byte: a match may overlap itself)
			out[blk_out] = out[blk_out - off];
	}
	void seq_end(uint32_t tail)
	{
		memcpy(out + blk_out, lits + blk_lit, tail);
		blk_out += tail;
	}
	int64_t tree(const uint8_t *w, uint32_t t)
	{
		tl = t;
		zs::huf_build_dtable(w, t, dt);
		return 0;
	}
	int64_t huf(uint64_t src, uint32_t cs, uint64_t dst, uint32_t R, bool four, bool lit)
	{
		const uint8_t *p = f + src;
		uint8_t *out = where(dst, R, lit) - dst; // (so that out + dst is where the literals go)
		if (!four)
			return zs::huf_decode_stream(p, cs, dt, tl, out + dst, R) ? 0 : zs::W_BAD;
		const uint32_t s1 = p[0] | (p[1] << 8), s2 = p[2] | (p[3] << 8), s3 = p[4] | (p[5] << 8);
		if (6ull + s1 + s2 + s3 >= cs)
			return zs::W_BAD;
		const uint32_t s4 = cs - 6 - s1 - s2 - s3, seg = (R + 3) / 4;
		if (3 * seg > R)
			return zs::W_BAD;
		const uint32_t sz[4] = { s1, s2, s3, s4 };
		p += 6;
		for (int q = 0; q < 4; q++) {
			if (!zs::huf_decode_stream(p, sz[q], dt, tl, out + dst + q * seg, q < 3 ? seg : R - 3 * seg))
				return zs::W_BAD;
			p += sz[q];
		}
		return 0;
	}
};

int64_t zsm_decode(const uint8_t *frame, uint64_t len, uint8_t *out, uint64_t cap)
{
	HostSink s;
	s.f = frame;
	s.out = out;
	s.tl = 0;
	s.cap = cap;
	zs::ReadWork k;
	return zs::walk_frame(frame, len, cap, s, k);
}

} // extern "C"
