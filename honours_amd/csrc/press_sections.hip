// press_sections.hip - the per-read control kernels of the exception-split methods for gfx950:
// the header and "u32 nex || exception section" of a read on the encode side, and the parse of
// the same bytes back into a sorted exception list on the decode side.  Both are a few dozen
// bytes per read (NA12878: 4.85 exceptions per read, thesis/plots/ex-tab.tex:14) - one lane per
// read; the sample streams themselves are press_chunked.hip's business.
// All arithmetic is integer (u8/u16/u32).

#include <stdlib.h>

#include "press_internal.h"

namespace ph {

constexpr uint64_t FAIL64 = ~0ull;

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
	if (huff == 2 && (m->ored >> 31))
		return; // a one-byte value the table has no code for (huffman.c:860 dereferences a NULL code there)

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
	if (huff && fmt != EXF_VBE21 && seclen > 65535) // (the entropy-coded b/sb/ss forms keep this length in a uint16_t: press.c:4520, 6917)
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

// ------------------------------------------------------------------ launchers

void launch_ex_section(const BatchArgs &a, int fmt, int ent, hipStream_t s)
{
	hipLaunchKernelGGL(k_ex_section, dim3(a.nreads), dim3(64), 0, s, a, fmt, ent == 1 ? 2 : ent >= 2 ? 3 : 0);
	if (fmt == EXF_VBE21)
		hipLaunchKernelGGL(k_ex_fill_vbe21, dim3(a.nreads), dim3(64), 0, s, a);
}

// k_ex_parse and, for the Huffman variants, the stream decode into a.low (timed as the
// dominant kernel of those methods)
void launch_ex_parse_huff(const DecodeArgs &a, int fmt, int ent, hipStream_t s)
{
	hipLaunchKernelGGL(k_ex_parse, dim3(a.nreads), dim3(64), 0, s, a, fmt, ent >= 2 ? 3 : ent);
	if (fmt == EXF_VBE21)
		hipLaunchKernelGGL(k_ex_parse_fill_vbe21, dim3(a.nreads), dim3(64), 0, s, a);
	if (ent) {
		ktime_begin(1, s);
		if (ent == 2)
			launch_rcs_decode(a, s);
		else if (ent == 3)
			launch_rcc_decode(a, s);
		else if (ent == 4)
			launch_rcm_decode(a, s);
		else
			launch_huff_decode(a, a.huf_minlen, s);
		ktime_end(1, s);
	}
}

} // namespace ph
