// press_sections.hip - the per-read control kernels of the exception-split methods for gfx950:
// the header and "u32 nex || exception section" of a read on the encode side, and the parse of
// the same bytes back into a sorted exception list on the decode side.  Both are a few dozen
// bytes per read on NA12878 (4.85 exceptions per read, thesis/plots/ex-tab.tex:14) - a wave per
// read, 64 exceptions per round, so that reads with thousands of exceptions cost microseconds too;
// the sample streams themselves are press_chunked.hip's business.
// All arithmetic is integer (u8/u16/u32).

#include <stdlib.h>

#include "press_internal.h"
#include "press_packed.h"

namespace ph {

constexpr uint64_t FAIL64 = ~0ull;

// ------------------------------------------------------------------ exception split: section builder
//
// Writes the header and "u32 nex || section" of one read and decides whether the read
// fits its slot.  One wave per read: lane 0 writes the header fields, the whole wave the two
// coded blocks.
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

// ---- the blocks of a section, written by a whole wave, 64 exceptions per round (one lane per read kept a lane busy
// for 14 ms here and 28 ms in k_ex_parse on reads with 30 % exceptions: tools/exc_heavy.py)

__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1)
		v = max(v, (uint32_t) __shfl_xor((int) v, d, 64));
	return v;
}

__device__ __forceinline__ uint32_t wave_total(uint32_t v) // sum over the wave, in every lane
{
	return (uint32_t) __builtin_amdgcn_readlane((int) wave_incl_scan_dpp(v), 63);
}

__device__ void w_exsec_sizes(int fmt, const uint32_t *pos, const uint32_t *val, uint32_t nex, uint32_t lane,
			      uint32_t &len_pos, uint32_t &len_val, uint32_t &bits_pos, uint32_t &bits_val)
{
	uint32_t maxd = 0, maxv = 0, dbytes = 0, vbytes = 0;
	for (uint32_t k = lane; k < nex; k += 64) {
		const uint32_t d = dpos(pos, k), v = val[k] - 256u;
		maxd = max(maxd, d);
		maxv = max(maxv, v);
		dbytes += svb32_nbytes(d);
		vbytes += (fmt == EXF_EXZD) ? svb32_nbytes(v) : (v > 255u ? 2u : 1u);
	}
	maxd = wave_max(maxd);
	maxv = wave_max(maxv);
	dbytes = wave_total(dbytes);
	vbytes = wave_total(vbytes);
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

// [bits][values, `bits` bits each, most significant bit first]: 64 values = 8 * bits whole bytes per round, put
// together in LDS (neighbours share bytes) and copied out
__device__ void w_bitpack_write(uint8_t *o, uint32_t nex, uint32_t bits, const uint32_t *pos, const uint32_t *val, uint32_t lane,
				uint32_t *buf /* LDS, 66 dwords */)
{
	if (lane == 0)
		o[0] = (uint8_t) bits;
	if (!bits)
		return;
	uint8_t *p = o + 1;
	for (uint32_t k0 = 0; k0 < nex; k0 += 64) {
		const uint32_t k = k0 + lane;
		const uint32_t cnt = nex - k0 < 64 ? nex - k0 : 64;
		for (uint32_t i = lane; i < 66; i += 64)
			buf[i] = 0;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		if (k < nex) {
			const uint32_t v = pos ? dpos(pos, k) : val[k] - 256u;
			const uint32_t B = lane * bits; // bit offset in the round's bytes
			// the value's bits in a 64-bit word whose most significant bit is bit 8 * (B / 8) of the stream
			const uint64_t w = (uint64_t) v << (64 - bits - (B & 7u));
			const uint32_t nb = ((B & 7u) + bits + 7) / 8; // bytes it touches (at most 5)
			for (uint32_t j = 0; j < nb; j++) {
				const uint32_t byte = (B >> 3) + j;
				const uint32_t x = (uint32_t) (w >> (56 - 8 * j)) & 0xFFu;
				if (x)
					atomicOr(&buf[byte >> 2], x << (8 * (byte & 3u)));
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		const uint32_t nbytes = (cnt * bits + 7) / 8;
		for (uint32_t i = lane; i < nbytes; i += 64)
			p[i] = (uint8_t) (buf[i >> 2] >> (8 * (i & 3u)));
		p += 8 * bits; // (a full round; the last round ends the block)
		__builtin_amdgcn_wave_barrier();
	}
}

// streamvbyte_encode.c:36 over position deltas (pos != NULL) or value-256
__device__ void w_svb32_write(uint8_t *o, uint32_t nex, const uint32_t *pos, const uint32_t *val, uint32_t lane)
{
	uint8_t *d = o + (nex + 3) / 4;
	for (uint32_t k0 = 0; k0 < nex; k0 += 64) {
		const uint32_t k = k0 + lane;
		const bool act = k < nex;
		const uint32_t v = act ? (pos ? dpos(pos, k) : val[k] - 256u) : 0u;
		const uint32_t nb = act ? svb32_nbytes(v) : 0u;
		const uint32_t inc = wave_incl_scan_dpp(nb);
		uint8_t *q = d + (inc - nb);
		for (uint32_t b = 0; b < nb; b++)
			q[b] = (uint8_t) (v >> (8 * b));
		// the four 2-bit codes of a key byte sit in four neighbouring lanes
		uint32_t key = act ? (nb - 1) << (2 * (k & 3u)) : 0u;
		key |= (uint32_t) __shfl_xor((int) key, 1, 64);
		key |= (uint32_t) __shfl_xor((int) key, 2, 64);
		if (act && (k & 3u) == 0)
			o[k >> 2] = (uint8_t) key;
		d += (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
	}
}

// svb16/encode_scalar.hpp:14 without delta/zigzag over value-256 (press.c:3262: svb12_press)
__device__ void w_svb16_write(uint8_t *o, uint32_t nex, const uint32_t *val, uint32_t lane)
{
	uint8_t *d = o + (nex + 7) / 8;
	for (uint32_t k0 = 0; k0 < nex; k0 += 64) {
		const uint32_t k = k0 + lane;
		const bool act = k < nex;
		const uint32_t v = act ? (val[k] - 256u) & 0xFFFFu : 0u;
		const uint32_t nb = act ? (v > 255u ? 2u : 1u) : 0u;
		const uint32_t inc = wave_incl_scan_dpp(nb);
		uint8_t *q = d + (inc - nb);
		if (act) {
			q[0] = (uint8_t) v;
			if (nb == 2)
				q[1] = (uint8_t) (v >> 8);
		}
		const unsigned long long two = __ballot(act && nb == 2);
		if (act && (k & 7u) == 0)
			o[k >> 3] = (uint8_t) (two >> lane);
		d += (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
	}
}

__global__ __launch_bounds__(64) void k_ex_section(BatchArgs a, int fmt, int huff)
{
	__shared__ uint32_t s_buf[66];
	const uint32_t r = blockIdx.x;
	const uint32_t lane = threadIdx.x;
	const bool l0 = lane == 0; // what is not a block of the section is lane 0's
	const uint64_t o0 = a.off[r];
	const uint32_t n = a.nsamp[r];
	const uint32_t *pos = a.ex_pos + o0;
	const uint32_t *val = a.ex_val + o0;
	uint8_t *out = a.out + a.out_off[r];
	const uint64_t cap = a.out_off[r + 1] - a.out_off[r];
	ReadMeta *m = a.meta + r;
	const uint32_t nex = m->nex;
	const uint32_t zd0 = m->zd0, mq = m->q, ored = m->ored;
	const uint32_t hdr = (fmt == EXF_EXZD) ? 12u : 2u;

	if (l0) {
		m->hdr = hdr;
		m->status = 1;
		m->seclen = 0;
		a.out_len[r] = FAIL64; // until proven to fit
	}
	if (n == 0)
		return; // the reference reads zd[0] of an empty array: outside its domain
	if (huff == 2 && (ored >> 31))
		return; // a one-byte value the table has no code for (huffman.c:860 dereferences a NULL code there)

	uint32_t len_pos = 0, len_val = 0, bits_pos = 0, bits_val = 0;
	uint64_t seclen = 4;
	if (fmt == EXF_VBE21) {
		seclen += 6ull * nex;
	} else if (nex == 1) {
		seclen += (fmt == EXF_EXZD) ? 8 : 6;
	} else if (nex > 1) {
		w_exsec_sizes(fmt, pos, val, nex, lane, len_pos, len_val, bits_pos, bits_val);
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
	if (l0) {
		if (fmt == EXF_EXZD) {
			p[0] = 0; // version
			put32(p + 1, n);
			put32(p + 5, 0);
			p[9] = (uint8_t) mq;
			put16(p + 10, zd0);
		} else {
			put16(p, zd0);
		}
		put32(p + hdr, nex);
	}
	p += hdr + 4;
	if (fmt == EXF_VBE21) {
		// nex x u32 positions, nex x u16 values: copied by k_ex_fill_vbe21, a whole wave per read
	} else if (nex == 1) {
		if (l0) {
			put32(p, pos[0]);
			if (fmt == EXF_EXZD)
				put32(p + 4, val[0] - 256u);
			else
				put16(p + 4, val[0] - 256u);
		}
	} else if (nex > 1) { // the two coded blocks: the whole wave, 64 exceptions per round
		if (l0)
			put32(p, len_pos);
		p += 4;
		if (fmt == EXF_VBBE21)
			w_bitpack_write(p, nex, bits_pos, pos, nullptr, lane, s_buf);
		else
			w_svb32_write(p, nex, pos, nullptr, lane);
		p += len_pos;
		if (l0)
			put32(p, len_val);
		p += 4;
		if (fmt == EXF_VBBE21 || fmt == EXF_VBSBE21)
			w_bitpack_write(p, nex, bits_val, nullptr, val, lane, s_buf);
		else if (fmt == EXF_VBSSE21)
			w_svb16_write(p, nex, val, lane);
		else
			w_svb32_write(p, nex, nullptr, val, lane);
	}
	if (!l0)
		return;
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

// one coded block of a section read by a whole wave, 64 values per round (streamvbyte_decode.c:62,
// svb16/decode_scalar.hpp, the bit-packed form of press.c:397)
struct BlockReader {
	const uint8_t *in;
	uint32_t len;
	int kind;      // 0 bit-packed, 1 svb32 (2-bit keys), 2 svb16 (1-bit keys)
	uint32_t bits; // bit-packed: bits per value
	uint32_t dptr; // svb: offset of the round's first data byte
	bool bad;
};

__device__ __forceinline__ BlockReader block_open(const uint8_t *in, uint32_t len, uint32_t nex, int kind)
{
	BlockReader b = { in, len, kind, 0, 0, false };
	if (kind == 0) {
		if (len < 1) {
			b.bad = true;
		} else {
			b.bits = in[0];
			b.bad = b.bits > 32 || 1ull + ((uint64_t) nex * b.bits + 7) / 8 > len;
		}
	} else {
		b.dptr = kind == 1 ? (nex + 3) / 4 : (nex + 7) / 8;
		b.bad = b.dptr > len;
	}
	return b;
}

// value number k of the block (k = round's first + lane; act: k < nex); all lanes call
__device__ __forceinline__ uint32_t block_round(BlockReader &b, uint32_t k, bool act)
{
	if (b.kind == 0) {
		if (!act || !b.bits)
			return 0u;
		const uint64_t B = (uint64_t) k * b.bits;
		const uint8_t *q = b.in + 1 + (B >> 3);
		const uint32_t sh = (uint32_t) (B & 7u), nb = (sh + b.bits + 7) / 8;
		uint64_t w = 0;
		for (uint32_t j = 0; j < nb; j++) // (inside the block: its size was checked when it was opened)
			w |= (uint64_t) q[j] << (56 - 8 * j);
		return (uint32_t) ((w << sh) >> (64 - b.bits));
	}
	uint32_t nb = 0;
	if (act)
		nb = b.kind == 1 ? ((b.in[k >> 2] >> (2 * (k & 3u))) & 3u) + 1u : ((b.in[k >> 3] >> (k & 7u)) & 1u) + 1u;
	const uint32_t inc = wave_incl_scan_dpp(nb);
	const uint32_t at = b.dptr + inc - nb;
	uint32_t v = 0;
	if (act) {
		if ((uint64_t) at + nb > b.len) {
			b.bad = true;
		} else {
			for (uint32_t j = 0; j < nb; j++)
				v |= (uint32_t) b.in[at + j] << (8 * j);
		}
	}
	b.dptr += (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
	return v;
}

__global__ __launch_bounds__(64) void k_ex_parse(DecodeArgs a, int fmt, int huff)
{
	const uint32_t r = blockIdx.x;
	const uint32_t lane = threadIdx.x;
	const bool l0 = lane == 0; // what is not a block of the section is lane 0's (every lane reads the same header)
	const uint64_t o0 = a.off[r];
	const uint32_t cap = a.nsamp[r]; // samples the caller has room for
	uint32_t *pos = a.ex_pos + o0;
	uint32_t *val = a.ex_val + o0;
	const uint8_t *in = a.in + a.in_off[r];
	const uint64_t len = a.in_len[r];
	ReadMeta *m = a.meta + r;
	const uint32_t hdr = (fmt == EXF_EXZD) ? 12u : 2u;

	if (l0) {
		m->status = 1;
		m->hdr = hdr;
		m->nex = 0;
		m->q = 0;
	}
	// the two control blocks of the decode call (a.ctl[0]: the chunk table of k_chunk_prep_meta .. k_low_decode_chunked,
	// a.ctl[1]: the Huffman decoder's tiles, lists and tickets) are cleared here, by the first kernel of the call - two
	// hipMemsetAsync between the kernels were 10 us each
	if (r == 0) {
		uint32_t *c = reinterpret_cast<uint32_t *>(a.ctl);
		for (uint32_t i = lane; i < 2 * sizeof(ChunkCtl) / 4; i += 64)
			c[i] = 0;
	}
	if (len < (uint64_t) hdr + 4 || cap == 0)
		return;
	uint32_t mq = 0, zd0;
	if (fmt == EXF_EXZD) {
		// ex_zd.c:495-519: version 0, u64 n, q <= 5
		if (in[0] != 0 || get32(in + 5) != 0 || in[9] > 5)
			return;
		const uint32_t n = get32(in + 1);
		if (n == 0 || n > cap)
			return;
		mq = in[9];
		zd0 = get16(in + 10);
	} else {
		zd0 = get16(in);
	}
	if (l0) {
		m->q = mq;
		m->zd0 = zd0;
	}
	const uint8_t *p = in + hdr;
	uint64_t left = len - hdr - 4;
	const uint32_t nex = get32(p);
	p += 4;
	if ((uint64_t) nex >= cap)
		return; // more exceptions than zd[1..] can hold
	uint64_t seclen = 4;
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
		const uint32_t p0 = get32(p);
		if (l0) {
			pos[0] = p0;
			val[0] = (((fmt == EXF_EXZD) ? get32(p + 4) : get16(p + 4)) + 256u) & 0xFFFFu;
		}
		if (p0 >= cap - 1)
			return; // positions lie inside zd[1..cap)
		seclen += need;
	} else {
		if (left < 4)
			return;
		const uint32_t lp = get32(p);
		if (left < 8ull + lp)
			return;
		const uint32_t lv = get32(p + 4 + lp);
		if (left < 8ull + lp + lv)
			return;
		BlockReader bp = block_open(p + 4, lp, nex, fmt == EXF_VBBE21 ? 0 : 1);
		BlockReader bv = block_open(p + 8 + lp, lv, nex, (fmt == EXF_VBBE21 || fmt == EXF_VBSBE21) ? 0 : fmt == EXF_VBSSE21 ? 2 : 1);
		if (bp.bad || bv.bad)
			return;
		// 64 exceptions per round: delta and value out of the two blocks, the position by a running sum
		// (trans.c:186), "value + 256" (press.c:3345: 16 bits), and the check of the positions: strictly increasing,
		// inside zd[1..cap)
		uint32_t carry = 0, prev_last = 0; // sum of (delta + 1) so far; the position in front of the round
		bool wrong = false;
		for (uint32_t k0 = 0; k0 < nex; k0 += 64) {
			const uint32_t k = k0 + lane;
			const bool act = k < nex;
			const uint32_t d = block_round(bp, k, act), v = block_round(bv, k, act);
			const uint32_t step = act ? d + 1u : 0u;
			const uint32_t inc = wave_incl_scan_dpp(step);
			const uint32_t pk = carry + inc - 1u; // pos[0] = d0; pos[k] = pos[k - 1] + d + 1
			uint32_t before = (uint32_t) __shfl_up((int) pk, 1, 64);
			if (lane == 0)
				before = prev_last;
			if (act) {
				pos[k] = pk;
				val[k] = (v + 256u) & 0xFFFFu;
				if (pk >= cap - 1 || (k && pk <= before))
					wrong = true;
			}
			carry += (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
			prev_last = (uint32_t) __builtin_amdgcn_readlane((int) pk, 63);
		}
		if (__ballot(wrong || bp.bad || bv.bad))
			return;
		seclen += 8ull + lp + lv;
	}
	if (!l0)
		return;
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
