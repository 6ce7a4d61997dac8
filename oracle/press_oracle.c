/*
 * press_oracle.c - plain-C restatement of the reference's per-read codecs.
 *
 * TEST INFRASTRUCTURE ONLY (see press_oracle.h).  Written from the stream formats
 * and behaviour of /root/reference/press (cited per function as file:line), not
 * from its code: one writer/reader per format, explicit bounds checks, no
 * realloc-growing lists.  Bit-exact with the reference inside the reference's
 * valid domain (SURVEY.md "Reference quirks"); outside it this file returns -1
 * where the reference would overflow a buffer.
 *
 * All multi-byte fields are little endian except the two Huffman headers
 * (big endian, huffman.c:507,1203).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "press_oracle.h"

/* ------------------------------------------------------------------ helpers */

static void put_u16(uint8_t *p, uint16_t v) { p[0] = (uint8_t) v; p[1] = (uint8_t) (v >> 8); }
static void put_u32(uint8_t *p, uint32_t v) { int i; for (i = 0; i < 4; i++) p[i] = (uint8_t) (v >> (8 * i)); }
static void put_u64(uint8_t *p, uint64_t v) { int i; for (i = 0; i < 8; i++) p[i] = (uint8_t) (v >> (8 * i)); }
static uint16_t get_u16(const uint8_t *p) { return (uint16_t) (p[0] | (p[1] << 8)); }
static uint32_t get_u32(const uint8_t *p) { return (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16) | ((uint32_t) p[3] << 24); }
static uint64_t get_u64(const uint8_t *p) { return (uint64_t) get_u32(p) | ((uint64_t) get_u32(p + 4) << 32); }

/* trans.c:75 - zigzag of a 16-bit difference, result truncated to 16 bits */
static uint16_t zz16(int16_t d)
{
	return (uint16_t) (((uint16_t) d << 1) ^ (uint16_t) (d >> 15));
}

/* trans.c:80 */
static int16_t unzz16(uint16_t z)
{
	return (int16_t) ((z >> 1) ^ (uint16_t) (0 - (z & 1)));
}

/* trans.c:215 (and :233 for the u32 flavour: same values, wider cells) */
void po_zigdelta_u16(const int16_t *in, uint64_t n, uint16_t *out)
{
	uint16_t prev = 0;
	uint64_t i;
	for (i = 0; i < n; i++) {
		uint16_t cur = (uint16_t) in[i];
		out[i] = zz16((int16_t) (uint16_t) (cur - prev));
		prev = cur;
	}
}

/* trans.c:260 - running sum in int16 wraparound */
void po_unzigdelta_u16(const uint16_t *in, uint64_t n, int16_t *out)
{
	uint16_t acc = 0;
	uint64_t i;
	for (i = 0; i < n; i++) {
		acc = (uint16_t) (acc + (uint16_t) unzz16(in[i]));
		out[i] = (int16_t) acc;
	}
}

/* ------------------------------------------------------------------ svb16 (a4,a5) */

/* svb16.h:11 */
static uint32_t svb16_keylen(uint32_t n) { return (n >> 3) + (((n & 7) + 7) >> 3); }

/*
 * encode.hpp:11 / encode_scalar.hpp:14.  Layout: ceil(n/8) key bytes (bit i%8 of
 * byte i/8 set iff value i needs 2 bytes), then the values, 1 byte if < 256 else
 * 2 bytes LE.  zd != 0: value = zigzag(in[i]-in[i-1]) with in[-1] = 0; zd == 0:
 * value = in[i] reinterpreted as u16.
 */
uint64_t po_svb16_encode(const int16_t *in, uint32_t n, int zd, uint8_t *out)
{
	uint32_t klen = svb16_keylen(n);
	uint8_t *data = out + klen;
	uint16_t prev = 0;
	uint32_t i;
	memset(out, 0, klen);
	for (i = 0; i < n; i++) {
		uint16_t v = (uint16_t) in[i];
		if (zd) {
			uint16_t d = (uint16_t) (v - prev);
			prev = v;
			v = zz16((int16_t) d);
		}
		if (v < 256) {
			*data++ = (uint8_t) v;
		} else {
			put_u16(data, v);
			data += 2;
			out[i >> 3] |= (uint8_t) (1u << (i & 7));
		}
	}
	return (uint64_t) (data - out);
}

/* decode.hpp:23 / decode_scalar.hpp:31.  n = number of samples. returns bytes consumed */
uint64_t po_svb16_decode(const uint8_t *in, uint32_t n, int zd, int16_t *out)
{
	const uint8_t *data = in + svb16_keylen(n);
	uint16_t acc = 0;
	uint32_t i;
	for (i = 0; i < n; i++) {
		uint16_t v;
		if ((in[i >> 3] >> (i & 7)) & 1) {
			v = get_u16(data);
			data += 2;
		} else {
			v = *data++;
		}
		if (zd) {
			acc = (uint16_t) (acc + (uint16_t) unzz16(v));
			v = acc;
		}
		out[i] = (int16_t) v;
	}
	return (uint64_t) (data - in);
}

/* ------------------------------------------------------------------ svb32 (a6) */

/* streamvbyte_encode.c:10-34: code = bytes-1 of the value */
static unsigned svb32_code(uint32_t v)
{
	return v < (1u << 8) ? 0 : v < (1u << 16) ? 1 : v < (1u << 24) ? 2 : 3;
}

/* streamvbyte_encode.c:70: ceil(n/4) key bytes (2 bits per value, value i in bits
 * 2*(i%4)), then 1..4 data bytes per value, LE */
uint64_t po_svb32_encode(const uint32_t *in, uint32_t n, uint8_t *out)
{
	uint32_t klen = (n + 3) / 4;
	uint8_t *data = out + klen;
	uint32_t i;
	memset(out, 0, klen);
	for (i = 0; i < n; i++) {
		unsigned c = svb32_code(in[i]), b;
		for (b = 0; b <= c; b++)
			*data++ = (uint8_t) (in[i] >> (8 * b));
		out[i >> 2] |= (uint8_t) (c << (2 * (i & 3)));
	}
	return (uint64_t) (data - out);
}

/* streamvbyte_decode.c:62 */
uint64_t po_svb32_decode(const uint8_t *in, uint32_t n, uint32_t *out)
{
	const uint8_t *data = in + (n + 3) / 4;
	uint32_t i;
	if (n == 0)
		return 0;
	for (i = 0; i < n; i++) {
		unsigned c = (in[i >> 2] >> (2 * (i & 3))) & 3, b;
		uint32_t v = 0;
		for (b = 0; b <= c; b++)
			v |= (uint32_t) *data++ << (8 * b);
		out[i] = v;
	}
	return (uint64_t) (data - in);
}

/* streamvbyte.h:35 / :42 (STREAMVBYTE_PADDING = 16); both take a uint32_t count */
static uint64_t svb32_bound(uint32_t n) { return (uint64_t) (n + 3) / 4 + (uint64_t) n * 4 + 16; }
static uint64_t svb16_bound(uint32_t n) { return (uint64_t) svb16_keylen(n) + (uint64_t) n * 4 + 16; }

/* ------------------------------------------------------------------ bit-pack (a8: vbbe21/vbsbe21) */

/* press.c:463-473: smallest b with max < 2^b */
static unsigned minbits_u64(uint64_t max)
{
	unsigned b = 0;
	while (b < 64 && (max >> b))
		b++;
	return b;
}

/*
 * press.c:486-585 over the core at :285-397: one header byte b = minbits(max),
 * then the low b bits of every value, most significant bit first, bytes filled
 * from their most significant bit.  b == 0 => header only.
 */
uint64_t po_uint_pack(const void *in, uint32_t n, int width, uint8_t *out)
{
	const uint16_t *in16 = in;
	const uint32_t *in32 = in;
	uint64_t max = 0, nbits = 0;
	uint32_t i;
	unsigned b;
	int k;
	for (i = 0; i < n; i++) {
		uint64_t v = width == 16 ? in16[i] : in32[i];
		if (v > max)
			max = v;
	}
	b = minbits_u64(max);
	out[0] = (uint8_t) b;
	if (b == 0)
		return 1;
	memset(out + 1, 0, (size_t) (((uint64_t) n * b + 7) / 8));
	for (i = 0; i < n; i++) {
		uint64_t v = width == 16 ? in16[i] : in32[i];
		for (k = (int) b - 1; k >= 0; k--, nbits++)
			if ((v >> k) & 1)
				out[1 + (nbits >> 3)] |= (uint8_t) (0x80u >> (nbits & 7));
	}
	return 1 + (nbits + 7) / 8;
}

/* press.c:508-528 / :565-585.  returns bytes consumed */
uint64_t po_uint_unpack(const uint8_t *in, uint32_t n, int width, void *out)
{
	uint16_t *o16 = out;
	uint32_t *o32 = out;
	unsigned b = in[0];
	uint64_t nbits = 0;
	uint32_t i;
	unsigned k;
	for (i = 0; i < n; i++) {
		uint64_t v = 0;
		for (k = 0; k < b; k++, nbits++)
			v = (v << 1) | ((in[1 + (nbits >> 3)] >> (7 - (nbits & 7))) & 1);
		if (width == 16)
			o16[i] = (uint16_t) v;
		else
			o32[i] = (uint32_t) v;
	}
	return 1 + (nbits + 7) / 8;
}

/* ------------------------------------------------------------------ exception split (a8, a9) */

enum exfmt {
	EX_VBE21,   /* nex x u32 pos, nex x u16 raw value                press.c:2679 */
	EX_VBBE21,  /* bit-packed pos deltas, bit-packed (value-256)      press.c:2780 */
	EX_VBSBE21, /* svb32 pos deltas, bit-packed (value-256)           press.c:2985 */
	EX_VBSSE21, /* svb32 pos deltas, svb16 (value-256)                press.c:3191 */
	EX_EXZD     /* svb32 pos deltas, svb32 (value-256); 1 exc = 2xu32 ex_zd.c:9 */
};

struct exlist {
	uint32_t n;
	uint32_t *pos;
	uint32_t *val; /* raw value (> 255) */
};

static int exlist_scan(const uint16_t *z, uint32_t m, struct exlist *e)
{
	uint32_t i, k = 0;
	e->n = 0;
	for (i = 0; i < m; i++)
		if (z[i] > 255)
			e->n++;
	e->pos = malloc(((size_t) e->n + 1) * sizeof *e->pos);
	e->val = malloc(((size_t) e->n + 1) * sizeof *e->val);
	if (!e->pos || !e->val)
		return -1;
	for (i = 0; i < m; i++)
		if (z[i] > 255) {
			e->pos[k] = i;
			e->val[k] = z[i];
			k++;
		}
	return 0;
}

static void exlist_free(struct exlist *e)
{
	free(e->pos);
	free(e->val);
}

/* worst-case size of an exception section, used to size scratch */
static uint64_t exsec_cap(uint32_t nex) { return 64 + (uint64_t) nex * 10; }

/*
 * Write "u32 nex || section" for the m values z[0..m) (the zd stream without its
 * first element).  Returns the number of bytes written.  trans.c:129 gives the
 * position coding p0, p[i]-p[i-1]-1.
 */
static uint64_t exsec_write(enum exfmt f, const struct exlist *e, uint8_t *out)
{
	uint64_t o = 0;
	uint32_t nex = e->n, i;
	uint32_t *dpos;
	put_u32(out, nex);
	o = 4;
	if (nex == 0)
		return o;
	if (f == EX_VBE21) {
		for (i = 0; i < nex; i++, o += 4)
			put_u32(out + o, e->pos[i]);
		for (i = 0; i < nex; i++, o += 2)
			put_u16(out + o, (uint16_t) e->val[i]);
		return o;
	}
	if (nex == 1) {
		put_u32(out + o, e->pos[0]);
		o += 4;
		if (f == EX_EXZD) {
			put_u32(out + o, e->val[0] - 256);
			o += 4;
		} else {
			put_u16(out + o, (uint16_t) (e->val[0] - 256));
			o += 2;
		}
		return o;
	}
	dpos = malloc((size_t) nex * sizeof *dpos);
	dpos[0] = e->pos[0];
	for (i = 1; i < nex; i++)
		dpos[i] = e->pos[i] - e->pos[i - 1] - 1;
	{
		uint64_t len;
		if (f == EX_VBBE21)
			len = po_uint_pack(dpos, nex, 32, out + o + 4);
		else
			len = po_svb32_encode(dpos, nex, out + o + 4);
		put_u32(out + o, (uint32_t) len);
		o += 4 + len;
	}
	free(dpos);
	{
		uint64_t len;
		if (f == EX_EXZD) {
			uint32_t *v = malloc((size_t) nex * sizeof *v);
			for (i = 0; i < nex; i++)
				v[i] = e->val[i] - 256;
			len = po_svb32_encode(v, nex, out + o + 4);
			free(v);
		} else {
			uint16_t *v = malloc((size_t) nex * sizeof *v);
			for (i = 0; i < nex; i++)
				v[i] = (uint16_t) (e->val[i] - 256);
			if (f == EX_VBSSE21)
				len = po_svb16_encode((const int16_t *) v, nex, 0, out + o + 4);
			else
				len = po_uint_pack(v, nex, 16, out + o + 4);
			free(v);
		}
		put_u32(out + o, (uint32_t) len);
		o += 4 + len;
	}
	return o;
}

/*
 * Parse "u32 nex || section" at in[0..nin).  Fills e (raw values) and returns
 * the section length including the nex field, or 0 on a malformed stream.
 */
static uint64_t exsec_read(enum exfmt f, const uint8_t *in, uint64_t nin, struct exlist *e)
{
	uint64_t o = 4;
	uint32_t nex, i;
	e->n = 0;
	e->pos = e->val = NULL;
	if (nin < 4)
		return 0;
	nex = get_u32(in);
	if ((uint64_t) nex > nin) /* every exception costs at least a byte */
		return 0;
	e->pos = malloc(((size_t) nex + 1) * sizeof *e->pos);
	e->val = malloc(((size_t) nex + 1) * sizeof *e->val);
	if (!e->pos || !e->val)
		return 0;
	e->n = nex;
	if (nex == 0)
		return o;
	if (f == EX_VBE21) {
		if (nin < 4 + (uint64_t) nex * 6)
			return 0;
		for (i = 0; i < nex; i++, o += 4)
			e->pos[i] = get_u32(in + o);
		for (i = 0; i < nex; i++, o += 2)
			e->val[i] = get_u16(in + o);
		return o;
	}
	if (nex == 1) {
		if (nin < o + 4 + (f == EX_EXZD ? 4 : 2))
			return 0;
		e->pos[0] = get_u32(in + o);
		o += 4;
		if (f == EX_EXZD) {
			e->val[0] = get_u32(in + o) + 256;
			o += 4;
		} else {
			e->val[0] = (uint32_t) get_u16(in + o) + 256;
			o += 2;
		}
		return o;
	}
	{
		uint32_t len;
		if (nin < o + 4)
			return 0;
		len = get_u32(in + o);
		o += 4;
		if (nin < o + len)
			return 0;
		if (f == EX_VBBE21)
			(void) po_uint_unpack(in + o, nex, 32, e->pos);
		else
			(void) po_svb32_decode(in + o, nex, e->pos);
		o += len;
		for (i = 1; i < nex; i++) /* trans.c:186 */
			e->pos[i] += e->pos[i - 1] + 1;
	}
	{
		uint32_t len;
		if (nin < o + 4)
			return 0;
		len = get_u32(in + o);
		o += 4;
		if (nin < o + len)
			return 0;
		if (f == EX_EXZD) {
			(void) po_svb32_decode(in + o, nex, e->val);
			for (i = 0; i < nex; i++)
				e->val[i] += 256;
		} else {
			uint16_t *v = malloc((size_t) nex * sizeof *v);
			if (f == EX_VBSSE21)
				(void) po_svb16_decode(in + o, nex, 0, (int16_t *) v);
			else
				(void) po_uint_unpack(in + o, nex, 16, v);
			for (i = 0; i < nex; i++)
				e->val[i] = (uint32_t) (uint16_t) (v[i] + 256);
			free(v);
		}
		o += len;
	}
	return o;
}

/* the one-byte stream: every non-exception value of z[0..m), in order */
static uint64_t lowbytes_write(const uint16_t *z, uint32_t m, uint8_t *out)
{
	uint64_t o = 0;
	uint32_t i;
	for (i = 0; i < m; i++)
		if (z[i] <= 255)
			out[o++] = (uint8_t) z[i];
	return o;
}

/*
 * Inverse of the split (press.c:2731 and siblings): interleave nlow one-byte
 * values with the exceptions.  Produces nlow + nex values, or fails when an
 * exception position is not reachable.
 */
static int split_merge(const struct exlist *e, const uint8_t *low, uint64_t nlow,
		       uint16_t *z, uint64_t cap, uint64_t *m)
{
	uint64_t i = 0, l = 0;
	uint32_t j = 0;
	while (l < nlow || j < e->n) {
		if (i >= cap)
			return -1;
		if (j < e->n && i == e->pos[j]) {
			z[i] = (uint16_t) e->val[j];
			j++;
		} else {
			if (l >= nlow)
				return -1;
			z[i] = low[l++];
		}
		i++;
	}
	*m = i;
	return 0;
}

/* ------------------------------------------------------------------ static Huffman (a10) */

#define HMAXBITS 64

static struct {
	int loaded;
	uint32_t len[256];
	uint64_t bits[256]; /* bit k = k-th emitted bit */
	/* binary trie for decoding: node 0 is the root */
	int32_t child[2 * 256 * HMAXBITS / 8][2];
	int32_t leaf[2 * 256 * HMAXBITS / 8];
	int32_t nnodes;
} g_tab;

/*
 * huffman.c:549 read_code_table: u32 BE entry count, u32 BE byte count (unused),
 * then per entry {u8 symbol, u8 numbits, ceil(numbits/8) code bytes} where bit k
 * of the code (k = 0 is the bit next to the root, emitted first) sits at bit k%8
 * of byte k/8 (huffman.c:427-439).
 */
int po_load_table_mem(const uint8_t *buf, uint64_t len)
{
	uint64_t o = 8;
	uint32_t count, i;
	memset(&g_tab, 0, sizeof g_tab);
	if (len < 8)
		return -1;
	count = ((uint32_t) buf[0] << 24) | ((uint32_t) buf[1] << 16) | ((uint32_t) buf[2] << 8) | buf[3];
	if (count > 256)
		return -1;
	g_tab.nnodes = 1;
	g_tab.leaf[0] = -1;
	g_tab.child[0][0] = g_tab.child[0][1] = -1;
	for (i = 0; i < count; i++) {
		unsigned sym, nb, k;
		int32_t p = 0;
		if (o + 2 > len)
			return -1;
		sym = buf[o];
		nb = buf[o + 1];
		o += 2;
		if (nb == 0 || nb > HMAXBITS || o + (nb + 7) / 8 > len)
			return -1;
		g_tab.len[sym] = nb;
		g_tab.bits[sym] = 0;
		for (k = 0; k < nb; k++) {
			unsigned bit = (buf[o + k / 8] >> (k % 8)) & 1;
			if (bit)
				g_tab.bits[sym] |= (uint64_t) 1 << k;
			if (g_tab.leaf[p] >= 0)
				return -1; /* code runs through a leaf */
			if (g_tab.child[p][bit] < 0) {
				int32_t q = g_tab.nnodes++;
				g_tab.child[q][0] = g_tab.child[q][1] = -1;
				g_tab.leaf[q] = -1;
				g_tab.child[p][bit] = q;
			}
			p = g_tab.child[p][bit];
		}
		g_tab.leaf[p] = (int32_t) sym;
		o += (nb + 7) / 8;
	}
	g_tab.loaded = 1;
	return 0;
}

int po_load_table(const char *path)
{
	uint8_t buf[8192];
	size_t n;
	FILE *fp = fopen(path, "rb");
	if (!fp)
		return -1;
	n = fread(buf, 1, sizeof buf, fp);
	fclose(fp);
	return po_load_table_mem(buf, n);
}

int po_table_code(int sym, uint32_t *len, uint64_t *bits)
{
	if (!g_tab.loaded || sym < 0 || sym > 255 || !g_tab.len[sym])
		return -1;
	*len = g_tab.len[sym];
	*bits = g_tab.bits[sym];
	return 0;
}

/*
 * huffman.c:1184 + :848: u32 BE symbol count, then the codes bit by bit, each
 * output byte filled from bit 0 upwards, last byte zero padded.
 */
int po_shuff_encode(const uint8_t *in, uint32_t n, uint8_t *out, uint64_t *nout)
{
	uint64_t cap = *nout, nbits = 0, nbytes;
	uint32_t i;
	if (!g_tab.loaded)
		return 1;
	for (i = 0; i < n; i++) {
		if (!g_tab.len[in[i]])
			return 1;
		nbits += g_tab.len[in[i]];
	}
	nbytes = 4 + (nbits + 7) / 8;
	if (nbytes > cap)
		return 1;
	memset(out, 0, nbytes);
	out[0] = (uint8_t) (n >> 24);
	out[1] = (uint8_t) (n >> 16);
	out[2] = (uint8_t) (n >> 8);
	out[3] = (uint8_t) n;
	nbits = 0;
	for (i = 0; i < n; i++) {
		uint32_t l = g_tab.len[in[i]], k;
		uint64_t c = g_tab.bits[in[i]];
		for (k = 0; k < l; k++, nbits++)
			if ((c >> k) & 1)
				out[4 + (nbits >> 3)] |= (uint8_t) (1u << (nbits & 7));
	}
	*nout = nbytes;
	return 0;
}

/*
 * huffman.c:1219.  Fails (1) when the stream is not longer than its 4-byte header
 * (memread's ">=" test, huffman.c:704) - the reference's wrappers then crash, so
 * "at least one payload byte" bounds the valid domain.
 */
int po_shuff_decode(const uint8_t *in, uint64_t nin, uint8_t *out, uint32_t cap, uint32_t *nout)
{
	uint32_t want, got = 0;
	uint64_t i;
	int32_t p = 0;
	if (!g_tab.loaded || nin <= 4)
		return 1;
	want = ((uint32_t) in[0] << 24) | ((uint32_t) in[1] << 16) | ((uint32_t) in[2] << 8) | in[3];
	for (i = 4; i < nin && got < want; i++) {
		unsigned b;
		for (b = 0; b < 8 && got < want; b++) {
			p = g_tab.child[p][(in[i] >> b) & 1];
			if (p < 0)
				return 1;
			if (g_tab.leaf[p] >= 0) {
				if (got >= cap)
					return 1;
				out[got++] = (uint8_t) g_tab.leaf[p];
				p = 0;
			}
		}
	}
	*nout = got;
	return 0;
}

/* ------------------------------------------------------------------ zstd (third party, a7) */

static struct {
	int tried;
	size_t (*compress)(void *, size_t, const void *, size_t, int);
	size_t (*decompress)(void *, size_t, const void *, size_t);
	size_t (*bound)(size_t);
	unsigned (*is_error)(size_t);
} g_zstd;

static int zstd_open(void)
{
	static const char *names[] = { "/opt/conda/lib/libzstd.so.1", "libzstd.so.1", "libzstd.so", NULL };
	int i;
	if (g_zstd.tried)
		return g_zstd.compress ? 0 : -1;
	g_zstd.tried = 1;
	for (i = 0; names[i]; i++) {
		void *h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
		if (!h)
			continue;
		g_zstd.compress = (size_t (*)(void *, size_t, const void *, size_t, int)) dlsym(h, "ZSTD_compress");
		g_zstd.decompress = (size_t (*)(void *, size_t, const void *, size_t)) dlsym(h, "ZSTD_decompress");
		g_zstd.bound = (size_t (*)(size_t)) dlsym(h, "ZSTD_compressBound");
		g_zstd.is_error = (unsigned (*)(size_t)) dlsym(h, "ZSTD_isError");
		if (g_zstd.compress && g_zstd.decompress && g_zstd.bound && g_zstd.is_error)
			return 0;
		g_zstd.compress = NULL;
	}
	return -1;
}

/* ZSTD_compressBound as a formula (zstd.h ZSTD_COMPRESSBOUND), for when libzstd is absent */
static uint64_t zstd_bound_formula(uint64_t n)
{
	return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0);
}

static uint64_t zstd_bound_(uint64_t n)
{
	return zstd_open() == 0 ? g_zstd.bound(n) : zstd_bound_formula(n);
}

/* ------------------------------------------------------------------ bounds (a12) */

/* press.c:2575 - double arithmetic, truncated */
static uint64_t vb1e2_bound_(uint32_t m)
{
	return (uint64_t) (1 + m * 0.2 * 6 + m * 0.8);
}

static uint64_t vb_zd_bound(uint32_t n) { return 2 + vb1e2_bound_(n - 1); } /* press.c:3411 */

uint64_t po_bound(int method, uint32_t n)
{
	switch (method) {
	case PM_SVB12:
	case PM_SVB12_ZD:         return svb16_bound(n);                           /* press.c:1573,1678 */
	case PM_SVB_ZD:           return svb32_bound(n);                           /* press.c:1585 */
	case PM_ZSTD_SVB_ZD:      return zstd_bound_(4 + svb32_bound(n));          /* press.c:1860 */
	case PM_ZSTD_SVB12_ZD:    return zstd_bound_(4 + svb16_bound(n));          /* press.c:2020 */
	case PM_VBE21_ZD: case PM_VBBE21_ZD: case PM_VBSBE21_ZD: case PM_VBSSE21_ZD:
	case PM_SHUFF_VBE21_ZD: case PM_SHUFF_VBBE21_ZD: case PM_SHUFF_VBSBE21_ZD:
	case PM_SHUFF_VBSSE21_ZD: return vb_zd_bound(n);                           /* press.c:3411,4409 */
	case PM_RC_VBE21_ZD:      return vb_zd_bound(n);                           /* press.c:5422 */
	case PM_RCC_VBE21_ZD:     return vb_zd_bound(n);                           /* press.c:5510 */
	case PM_RCCM_VBBE21_ZD:   return vb_zd_bound(n);                           /* press.c:6901 */
	case PM_HASGAM_ZDQ:       return svb32_bound((uint32_t) vb_zd_bound(n));   /* press.c:8461 */
	case PM_ZSTD_HASGAM_ZDQ:  return zstd_bound_(svb32_bound((uint32_t) vb_zd_bound(n)));
	/* slow5_press.c:1037: __slow5_streamvbyte_max_compressedbytes(n) (streamvbyte.h:31, no padding) + the u32 count */
	case PM_SLOW5_SVB_ZD:     return (uint64_t) (n + 3) / 4 + (uint64_t) n * 4 + 4;
	}
	return 0;
}

/* ------------------------------------------------------------------ method bodies */

/* ------------------------------------------------------------------ order-0 binary range coder (8f-1)
 *
 * What the reference calls rcsenc / rcsdec (Turbo-Range-Coder, "simple predictor", press.c:5456,
 * 5489): a 64-bit range coder that emits 32-bit little-endian words, 15-bit probabilities of a
 * ONE bit kept in 255 contexts (the bits of the byte already coded, MSB first), all starting at
 * one half.  Restated from its behaviour (the compiled reference is the check, tests/
 * test_oracle_golden.py):
 *   bit b with probability p:  x = (range >> 15) * p;  b ? range = x : (range -= x, low += x);
 *                              a carry out of `low` increments the words already written;
 *   update:  b ? p += ceil((32768 - p) / 32) - 1  :  p -= p >> 5;
 *   renormalisation (range < 2^32: emit the top word of low, shift both by 32) is looked at only
 *   before the bits 7, 5, 3 and 1 of a byte - two bits cost at most 30 bits of range;
 *   end: renormalise, then low += 2^32 and one word if range > 2^33, else low += 1 and two words;
 *   give-up rule (rcutil_.h:161): as soon as the output reaches n*255/256 - 8 bytes the input is
 *   stored raw instead - for which the reference has no decoder, so such streams (in practice:
 *   reads of fewer than ~40 samples) are outside its lossless domain.
 */
struct rc_enc {
	uint64_t low, range;
	uint8_t *out;
	uint64_t pos; /* bytes written */
};

static void rc_put32(struct rc_enc *c, uint32_t w)
{
	c->out[c->pos++] = (uint8_t) w;
	c->out[c->pos++] = (uint8_t) (w >> 8);
	c->out[c->pos++] = (uint8_t) (w >> 16);
	c->out[c->pos++] = (uint8_t) (w >> 24);
}

static void rc_carry(struct rc_enc *c)
{
	uint64_t q = c->pos;
	while (q >= 4) { /* the reference would walk past the start of its buffer; it never has to */
		uint32_t w;
		q -= 4;
		w = get_u32(c->out + q) + 1;
		put_u32(c->out + q, w);
		if (w)
			break;
	}
}

static void rc_norm(struct rc_enc *c)
{
	if (c->range < ((uint64_t) 1 << 32)) {
		c->range <<= 32;
		rc_put32(c, (uint32_t) (c->low >> 32));
		c->low <<= 32;
	}
}

static void rc_bit(struct rc_enc *c, uint16_t *p, unsigned bit)
{
	const uint64_t x = (c->range >> 15) * *p, before = c->low;
	if (bit) {
		c->range = x;
		*p = (uint16_t) (*p + (32768u - *p + 31u) / 32u - 1u);
	} else {
		c->range -= x;
		c->low += x;
		*p = (uint16_t) (*p - (*p >> 5));
	}
	if (before > c->low)
		rc_carry(c);
}

/* out must have room for n + n/2 + 64 bytes; returns the stream length.
 * order1 (rccsenc, rc_.c:181-193): the same bit coder, the 255 probabilities of a byte chosen by the
 * byte in front of it (256 x 256 predictors, the first byte under context 0). */
static uint64_t rc_encode(const uint8_t *in, uint64_t n, uint8_t *out, int order1)
{
	struct rc_enc c = { 0, ~(uint64_t) 0, out, 0 };
	uint16_t *mb = malloc((order1 ? 65536u : 256u) * sizeof *mb);
	uint64_t i;
	const int64_t giveup = (int64_t) (n * 255 / 256) - 8;
	int k;
	unsigned cx = 0;
	for (k = 0; k < (order1 ? 65536 : 256); k++)
		mb[k] = 1u << 14;
	for (i = 0; i < n; i++) {
		const unsigned x = 0x100u | in[i];
		uint16_t *row = mb + (order1 ? cx * 256u : 0u);
		for (k = 7; k >= 0; k--) {
			if (k & 1)
				rc_norm(&c);
			rc_bit(&c, &row[x >> (k + 1)], (x >> k) & 1u);
		}
		cx = in[i];
		if ((int64_t) c.pos >= giveup) {
			memcpy(out, in, n);
			free(mb);
			return n;
		}
	}
	free(mb);
	rc_norm(&c);
	{
		const uint64_t before = c.low;
		if (c.range > ((uint64_t) 1 << 33)) {
			c.low += (uint64_t) 1 << 32;
			if (before > c.low)
				rc_carry(&c);
			rc_put32(&c, (uint32_t) (c.low >> 32));
		} else {
			c.low += 1;
			if (before > c.low)
				rc_carry(&c);
			rc_put32(&c, (uint32_t) (c.low >> 32));
			rc_put32(&c, (uint32_t) c.low);
		}
	}
	return c.pos;
}
uint64_t po_rcs_encode(const uint8_t *in, uint64_t n, uint8_t *out) { return rc_encode(in, n, out, 0); }
uint64_t po_rccs_encode(const uint8_t *in, uint64_t n, uint8_t *out) { return rc_encode(in, n, out, 1); }

static uint32_t rc_get32(const uint8_t *in, uint64_t len, uint64_t pos)
{
	uint32_t w = 0;
	int b;
	for (b = 0; b < 4; b++)
		if (pos + (uint64_t) b < len)
			w |= (uint32_t) in[pos + b] << (8 * b);
	return w;
}

/* decodes n bytes; bytes past `len` read as zeros (the reference reads whatever follows) */
static void rc_decode(const uint8_t *in, uint64_t len, uint64_t n, uint8_t *out, int order1)
{
	uint64_t range = ~(uint64_t) 0, code = 0, pos = 0, i;
	uint16_t *mb0 = malloc((order1 ? 65536u : 256u) * sizeof *mb0), *mb = mb0;
	int k;
#define RC_GET32() rc_get32(in, len, pos); pos += 4
	for (k = 0; k < (order1 ? 65536 : 256); k++)
		mb0[k] = 1u << 14;
	for (k = 0; k < 2; k++) {
		const uint32_t w = RC_GET32();
		code = (code << 32) | w;
	}
	for (i = 0; i < n; i++) {
		unsigned x = 1;
		for (k = 7; k >= 0; k--) {
			uint16_t *p = &mb[x];
			uint64_t t;
			if ((k & 1) && range < ((uint64_t) 1 << 32)) {
				const uint32_t w = RC_GET32();
				range <<= 32;
				code = (code << 32) | w;
			}
			t = (range >> 15) * *p;
			if (code < t) {
				range = t;
				*p = (uint16_t) (*p + (32768u - *p + 31u) / 32u - 1u);
				x = 2 * x + 1;
			} else {
				range -= t;
				code -= t;
				*p = (uint16_t) (*p - (*p >> 5));
				x = 2 * x;
			}
		}
		out[i] = (uint8_t) x;
		if (order1)
			mb = mb0 + 256u * (uint8_t) x; /* rc_.c:199: the byte just decoded is the next context */
	}
	free(mb0);
#undef RC_GET32
}
void po_rcs_decode(const uint8_t *in, uint64_t len, uint64_t n, uint8_t *out) { rc_decode(in, len, n, out, 0); }
void po_rccs_decode(const uint8_t *in, uint64_t len, uint64_t n, uint8_t *out) { rc_decode(in, len, n, out, 1); }

/*
 * rcmsenc / rcmsdec (Turbo-Range-Coder rccm_.c:79-121 instantiated by rccm_s.c:46-48: 16-bit
 * probabilities, simple predictor with the rate as parameter): the same range coder, every bit coded
 * under a mix of two predictions with a secondary estimation on top (mbc.h:185-193 mbum_p):
 *   p0 = order-0 predictor of the bit's node, p1 = the same node under the byte in front (order 1)
 *   p  = (p0 + 15 p1) / 16
 *   s  = value at p of the 17-point curve of the node (linear between the two points around p)
 *   P(bit = 1) = (p + 3 s) / 4
 * then p0 moves towards the bit by 1/4, p1 by 1/16, the two points of the curve by 1/64
 * (mbc.h:195-209, 230-235; mbc_s.h:40; mbu_updates mbc.h:94).  The range is renormalised in front of
 * every bit (turborc_.h:431-434).
 */
struct rcm_model {
	uint16_t mb0[256];
	uint16_t mb1[256][256];
	uint16_t sse[256][17];
};

static struct rcm_model *rcm_new(void)
{
	struct rcm_model *m = malloc(sizeof *m);
	int i, j;
	for (i = 0; i < 256; i++) {
		m->mb0[i] = 1u << 15;
		for (j = 0; j < 256; j++)
			m->mb1[i][j] = 1u << 15;
		for (j = 0; j <= 16; j++) /* rccm_s.c:40-44 */
			m->sse[i][j] = (uint16_t) ((j << 12) - (j == 16));
	}
	return m;
}

/* probability of a 1 under node x with `cx` the byte in front; *pp0, *pp1: the two predictions; *cell:
 * the lower one of the curve's two points */
static uint32_t rcm_p(struct rcm_model *m, unsigned cx, unsigned x, uint32_t *pp0, uint32_t *pp1, uint16_t **cell)
{
	const uint32_t p0 = m->mb0[x], p1 = m->mb1[cx][x];
	const int32_t p = (int32_t) ((p0 + 15u * p1) >> 4);
	uint16_t *c = &m->sse[x][p >> 12];
	const int32_t x1 = c[0];
	const int32_t sp = x1 + ((((int32_t) c[1] - x1) * (p & 4095)) >> 12);
	*pp0 = p0;
	*pp1 = p1;
	*cell = c;
	return (uint32_t) ((p + 3 * sp) >> 2);
}

static uint16_t rcm_step(uint32_t p, unsigned rate, unsigned bit)
{
	/* mbc_s.h:40 with the bit as a 64-bit unsigned (turborc_.h:421): the low 16 bits of
	 * p - (((p - (bit ? 65536 : 0)) >> rate) + bit) */
	const uint64_t t = (uint64_t) p - (bit ? 65536u : 0u);
	return (uint16_t) (p - ((t >> rate) + bit));
}

static void rcm_update(struct rcm_model *m, unsigned cx, unsigned x, uint32_t p0, uint32_t p1, uint16_t *cell, unsigned bit)
{
	m->mb0[x] = rcm_step(p0, 2, bit);
	m->mb1[cx][x] = rcm_step(p1, 4, bit);
	cell[0] = rcm_step(cell[0], 6, bit);
	cell[1] = rcm_step(cell[1], 6, bit);
}

/* out must have room for n + n/2 + 64 bytes; returns the stream length */
uint64_t po_rcms_encode(const uint8_t *in, uint64_t n, uint8_t *out)
{
	struct rc_enc c = { 0, ~(uint64_t) 0, out, 0 };
	struct rcm_model *m = rcm_new();
	const int64_t giveup = (int64_t) (n * 255 / 256) - 8; /* rcutil_.h:161 */
	unsigned cx = 0;
	uint64_t i;
	int k;
	for (i = 0; i < n; i++) {
		const unsigned x = 0x100u | in[i];
		for (k = 7; k >= 0; k--) {
			const unsigned node = x >> (k + 1), bit = (x >> k) & 1u;
			uint32_t p0, p1;
			uint16_t *cell;
			const uint32_t pm = rcm_p(m, cx, node, &p0, &p1, &cell);
			uint64_t t, before;
			rc_norm(&c);
			t = (c.range >> 16) * pm;
			before = c.low;
			if (bit) {
				c.range = t;
			} else {
				c.range -= t;
				c.low += t;
			}
			if (before > c.low)
				rc_carry(&c);
			rcm_update(m, cx, node, p0, p1, cell, bit);
		}
		cx = in[i];
		if ((int64_t) c.pos >= giveup) {
			memcpy(out, in, n);
			free(m);
			return n;
		}
	}
	free(m);
	rc_norm(&c);
	{
		const uint64_t before = c.low;
		if (c.range > ((uint64_t) 1 << 33)) {
			c.low += (uint64_t) 1 << 32;
			if (before > c.low)
				rc_carry(&c);
			rc_put32(&c, (uint32_t) (c.low >> 32));
		} else {
			c.low += 1;
			if (before > c.low)
				rc_carry(&c);
			rc_put32(&c, (uint32_t) (c.low >> 32));
			rc_put32(&c, (uint32_t) c.low);
		}
	}
	return c.pos;
}

/* decodes n bytes; bytes past `len` read as zeros (the reference reads whatever follows) */
void po_rcms_decode(const uint8_t *in, uint64_t len, uint64_t n, uint8_t *out)
{
	uint64_t range = ~(uint64_t) 0, code = 0, pos = 0, i;
	struct rcm_model *m = rcm_new();
	unsigned cx = 0;
	int k;
	for (k = 0; k < 2; k++) {
		const uint32_t w = rc_get32(in, len, pos);
		pos += 4;
		code = (code << 32) | w;
	}
	for (i = 0; i < n; i++) {
		unsigned x = 1;
		for (k = 7; k >= 0; k--) {
			uint32_t p0, p1;
			uint16_t *cell;
			const uint32_t pm = rcm_p(m, cx, x, &p0, &p1, &cell);
			uint64_t t;
			unsigned bit;
			if (range < ((uint64_t) 1 << 32)) {
				const uint32_t w = rc_get32(in, len, pos);
				pos += 4;
				range <<= 32;
				code = (code << 32) | w;
			}
			t = (range >> 16) * pm;
			bit = code < t;
			if (bit) {
				range = t;
			} else {
				range -= t;
				code -= t;
			}
			rcm_update(m, cx, x, p0, p1, cell, bit);
			x = 2 * x + bit;
		}
		out[i] = (uint8_t) x;
		cx = (uint8_t) x;
	}
	free(m);
}

static enum exfmt exfmt_of(int method)
{
	switch (method) {
	case PM_VBE21_ZD: case PM_SHUFF_VBE21_ZD: case PM_RC_VBE21_ZD: case PM_RCC_VBE21_ZD: return EX_VBE21;
	case PM_VBBE21_ZD: case PM_SHUFF_VBBE21_ZD: case PM_RCCM_VBBE21_ZD: return EX_VBBE21;
	case PM_VBSBE21_ZD: case PM_SHUFF_VBSBE21_ZD: return EX_VBSBE21;
	case PM_VBSSE21_ZD: case PM_SHUFF_VBSSE21_ZD: return EX_VBSSE21;
	default:                                      return EX_EXZD;
	}
}

static int is_shuff(int method)
{
	return method >= PM_SHUFF_VBE21_ZD && method <= PM_SHUFF_VBSSE21_ZD;
}

/*
 * u16 zd[0] || u32 nex || section || low bytes (plain, press.c:3411-3580) or their
 * static-Huffman stream (press.c:4414-4846).
 */
static int vb_family_press(int method, const int16_t *in, uint32_t n, uint8_t *out, uint64_t *nout)
{
	enum exfmt f = exfmt_of(method);
	uint64_t cap = *nout, o = 0, seclen, nlow;
	uint16_t *z;
	uint8_t *sec, *low;
	struct exlist e;
	int ret = -1;
	if (n == 0)
		return -1;
	z = malloc((size_t) n * sizeof *z);
	po_zigdelta_u16(in, n, z);
	if (exlist_scan(z + 1, n - 1, &e)) {
		free(z);
		return -1;
	}
	sec = malloc(exsec_cap(e.n));
	low = malloc(n);
	seclen = exsec_write(f, &e, sec);
	nlow = lowbytes_write(z + 1, n - 1, low);
	/* press.c:4520 etc: the b/sb/ss Huffman variants keep the section length in a uint16_t */
	if ((is_shuff(method) || method == PM_RCCM_VBBE21_ZD) && f != EX_VBE21 && seclen > 65535) /* (:6917 for rccm) */
		goto done;
	if (2 + seclen > cap)
		goto done;
	put_u16(out, z[0]);
	memcpy(out + 2, sec, seclen);
	o = 2 + seclen;
	if (method == PM_RC_VBE21_ZD || method == PM_RCC_VBE21_ZD || method == PM_RCCM_VBBE21_ZD) { /* press.c:5427 / :5547 / :6946: the one-byte values through rcsenc / rccsenc / rcmsenc */
		uint8_t *tmp = malloc((size_t) nlow + nlow / 2 + 64);
		const uint64_t rl = method == PM_RCCM_VBBE21_ZD ? po_rcms_encode(low, nlow, tmp) : rc_encode(low, nlow, tmp, method == PM_RCC_VBE21_ZD);
		if (o + rl > cap) {
			free(tmp);
			goto done;
		}
		memcpy(out + o, tmp, rl);
		free(tmp);
		o += rl;
	} else if (is_shuff(method)) {
		uint64_t hl = cap - o;
		if (po_shuff_encode(low, (uint32_t) nlow, out + o, &hl))
			goto done;
		o += hl;
	} else {
		if (o + nlow > cap)
			goto done;
		memcpy(out + o, low, nlow);
		o += nlow;
	}
	*nout = o;
	ret = 0;
done:
	free(sec);
	free(low);
	exlist_free(&e);
	free(z);
	return ret;
}

static int vb_family_depress(int method, const uint8_t *in, uint64_t nbytes, uint32_t n,
			     int16_t *out, uint32_t *nout)
{
	enum exfmt f = exfmt_of(method);
	struct exlist e;
	uint64_t seclen, m = 0, nlow;
	uint16_t *z;
	uint8_t *low = NULL;
	const uint8_t *lowp;
	int ret = -1;
	if (nbytes < 6 || n == 0)
		return -1;
	seclen = exsec_read(f, in + 2, nbytes - 2, &e);
	if (!seclen) {
		exlist_free(&e);
		return -1;
	}
	z = malloc(((size_t) n + 1) * sizeof *z);
	z[0] = get_u16(in);
	if (method == PM_RC_VBE21_ZD || method == PM_RCC_VBE21_ZD || method == PM_RCCM_VBBE21_ZD) { /* press.c:5465 / :5573 / :6987: n is the exact sample count, so the byte count is known */
		if ((uint64_t) e.n + 1 > n)
			goto done;
		nlow = (uint64_t) n - 1 - e.n;
		low = malloc((size_t) nlow + 1);
		if (method == PM_RCCM_VBBE21_ZD)
			po_rcms_decode(in + 2 + seclen, nbytes - 2 - seclen, nlow, low);
		else
			rc_decode(in + 2 + seclen, nbytes - 2 - seclen, nlow, low, method == PM_RCC_VBE21_ZD);
		lowp = low;
	} else if (is_shuff(method)) {
		uint32_t got = 0;
		low = malloc((size_t) n + 1);
		ret = po_shuff_decode(in + 2 + seclen, nbytes - 2 - seclen, low, n, &got);
		if (ret)
			goto done;
		ret = -1;
		lowp = low;
		nlow = got;
	} else {
		lowp = in + 2 + seclen;
		nlow = nbytes - 2 - seclen;
	}
	if (split_merge(&e, lowp, nlow, z + 1, (uint64_t) n - 1, &m))
		goto done;
	po_unzigdelta_u16(z, m + 1, out);
	*nout = (uint32_t) (m + 1);
	ret = 0;
done:
	free(low);
	free(z);
	exlist_free(&e);
	return ret;
}

/*
 * ex-zd v0 (ex_zd.c:403): u8 version 0 || u64 n || u8 q || u16 zd[0] || u32 nex ||
 * section || low bytes, computed on in >> q with q the largest shift <= 5 that loses
 * no bits in any sample (ex_zd.c:358-381).  The reference works in a 2n+1024-byte
 * buffer and fails when it does not fit (ex_zd.c:411).
 */
static int exzd_press(const int16_t *in, uint32_t n, uint8_t *out, uint64_t *nout)
{
	uint64_t cap = *nout, o, seclen, nlow, total;
	uint16_t ored = 0;
	unsigned q = 0;
	int16_t *s;
	uint16_t *z;
	uint8_t *sec;
	struct exlist e;
	uint32_t i;
	int ret = -1;
	if (n == 0)
		return -1;
	for (i = 0; i < n; i++)
		ored |= (uint16_t) in[i];
	while (q < 5 && !((ored >> q) & 1))
		q++;
	/* all-zero signal: every sample is divisible by 32 */
	s = malloc((size_t) n * sizeof *s);
	for (i = 0; i < n; i++)
		s[i] = (int16_t) (in[i] >> q);
	z = malloc((size_t) n * sizeof *z);
	po_zigdelta_u16(s, n, z);
	free(s);
	if (exlist_scan(z + 1, n - 1, &e)) {
		free(z);
		return -1;
	}
	sec = malloc(exsec_cap(e.n));
	seclen = exsec_write(EX_EXZD, &e, sec);
	nlow = (uint64_t) (n - 1) - e.n;
	total = 10 + 2 + seclen + nlow;
	if (total > (uint64_t) n * 2 + 1024 || total > cap)
		goto done;
	out[0] = 0;
	put_u64(out + 1, n);
	out[9] = (uint8_t) q;
	put_u16(out + 10, z[0]);
	memcpy(out + 12, sec, seclen);
	o = 12 + seclen;
	o += lowbytes_write(z + 1, n - 1, out + o);
	*nout = o;
	ret = 0;
done:
	free(sec);
	exlist_free(&e);
	free(z);
	return ret;
}

/* ex_zd.c:495 -> :459 -> :329 -> :174 */
static int exzd_depress(const uint8_t *in, uint64_t nbytes, uint32_t cap, int16_t *out, uint32_t *nout)
{
	struct exlist e;
	uint64_t n64, seclen, m = 0;
	unsigned q;
	uint16_t *z;
	uint64_t i;
	int ret = -1;
	if (nbytes < 16 || in[0] != 0)
		return -1;
	n64 = get_u64(in + 1);
	q = in[9];
	if (q > 5 || n64 == 0 || n64 > cap)
		return -1;
	seclen = exsec_read(EX_EXZD, in + 12, nbytes - 12, &e);
	if (!seclen) {
		exlist_free(&e);
		return -1;
	}
	z = malloc(((size_t) n64 + 1) * sizeof *z);
	z[0] = get_u16(in + 10);
	if (split_merge(&e, in + 12 + seclen, nbytes - 12 - seclen, z + 1, n64 - 1, &m))
		goto done;
	po_unzigdelta_u16(z, m + 1, out);
	for (i = 0; i < m + 1; i++)
		out[i] = (int16_t) ((uint16_t) out[i] << q);
	*nout = (uint32_t) (m + 1);
	ret = 0;
done:
	free(z);
	exlist_free(&e);
	return ret;
}

/* press.c:1865 / :2025 / :8554: zstd level 1 (press.h:275) over an inner stream */
static int zstd_wrap_press(int inner, const int16_t *in, uint32_t n, uint8_t *out, uint64_t *nout)
{
	uint64_t cap_in = 4 + po_bound(inner, n), len;
	uint8_t *buf;
	size_t r;
	int ret;
	if (zstd_open())
		return -1;
	buf = malloc(cap_in + 64);
	if (inner == PM_HASGAM_ZDQ) {
		len = cap_in;
		ret = po_press(inner, in, n, buf, &len);
	} else {
		put_u32(buf, n);
		len = cap_in - 4;
		ret = po_press(inner, in, n, buf + 4, &len);
		len += 4;
	}
	if (ret) {
		free(buf);
		return ret;
	}
	r = g_zstd.compress(out, *nout, buf, len, 1);
	free(buf);
	if (g_zstd.is_error(r))
		return -1;
	*nout = r;
	return 0;
}

static int zstd_wrap_depress(int inner, const uint8_t *in, uint64_t nbytes, uint32_t n,
			     int16_t *out, uint32_t *nout)
{
	uint64_t cap = zstd_bound_((uint64_t) n * 2); /* press.c:1897: the reference's own capacity */
	uint8_t *buf;
	size_t r;
	int ret;
	if (zstd_open())
		return -1;
	buf = malloc(cap + 64);
	r = g_zstd.decompress(buf, cap, in, nbytes);
	if (g_zstd.is_error(r)) {
		free(buf);
		return -1;
	}
	if (inner == PM_HASGAM_ZDQ) {
		ret = po_depress(inner, buf, r, n, out, nout);
	} else {
		uint32_t cnt;
		if (r < 4 || (cnt = get_u32(buf)) > n) {
			free(buf);
			return -1;
		}
		ret = po_depress(inner, buf + 4, r - 4, cnt, out, nout);
	}
	free(buf);
	return ret;
}

int po_press(int method, const int16_t *in, uint32_t n, uint8_t *out, uint64_t *nout)
{
	switch (method) {
	case PM_SVB12:
	case PM_SVB12_ZD: {
		/* worst case of the format itself: keys + 2 bytes per value */
		if ((uint64_t) svb16_keylen(n) + (uint64_t) n * 2 > *nout)
			return -1;
		*nout = po_svb16_encode(in, n, method == PM_SVB12_ZD, out);
		return 0;
	}
	case PM_SVB_ZD: { /* press.c:1590: zd widened to u32, then svb32 */
		uint32_t *z32;
		uint16_t *z;
		uint32_t i;
		if ((uint64_t) (n + 3) / 4 + (uint64_t) n * 2 > *nout)
			return -1;
		z = malloc(((size_t) n + 1) * sizeof *z);
		z32 = malloc(((size_t) n + 1) * sizeof *z32);
		po_zigdelta_u16(in, n, z);
		for (i = 0; i < n; i++)
			z32[i] = z[i];
		*nout = po_svb32_encode(z32, n, out);
		free(z);
		free(z32);
		return 0;
	}
	case PM_ZSTD_SVB_ZD:      return zstd_wrap_press(PM_SVB_ZD, in, n, out, nout);
	case PM_ZSTD_SVB12_ZD:    return zstd_wrap_press(PM_SVB12_ZD, in, n, out, nout);
	case PM_ZSTD_HASGAM_ZDQ:  return zstd_wrap_press(PM_HASGAM_ZDQ, in, n, out, nout);
	case PM_HASGAM_ZDQ:       return exzd_press(in, n, out, nout);
	case PM_VBE21_ZD: case PM_VBBE21_ZD: case PM_VBSBE21_ZD: case PM_VBSSE21_ZD:
	case PM_SHUFF_VBE21_ZD: case PM_SHUFF_VBBE21_ZD: case PM_SHUFF_VBSBE21_ZD:
	case PM_SHUFF_VBSSE21_ZD:
	case PM_RC_VBE21_ZD: case PM_RCC_VBE21_ZD: case PM_RCCM_VBBE21_ZD:
		return vb_family_press(method, in, n, out, nout);
	case PM_SLOW5_SVB_ZD: {
		/* slow5_press.c:1054 ptr_compress_svb_zd: samples widened to int32, zig-zag delta in 32 bits
		 * (streamvbyte_zigzag.c:15, prev = 0: a jump of more than 32767 takes 17 bits - unlike
		 * trans.c:233, which wraps at 16), svb32 of that behind the u32 sample count (:1034) */
		uint32_t *z32;
		int32_t prev = 0;
		uint32_t i;
		if (4 + (uint64_t) (n + 3) / 4 + (uint64_t) n * 3 > *nout)
			return -1;
		z32 = malloc(((size_t) n + 1) * sizeof *z32);
		for (i = 0; i < n; i++) {
			const int32_t d = (int32_t) in[i] - prev;
			z32[i] = ((uint32_t) d << 1) ^ (uint32_t) (d >> 31);
			prev = in[i];
		}
		put_u32(out, n);
		*nout = 4 + po_svb32_encode(z32, n, out + 4);
		free(z32);
		return 0;
	}
	}
	return -2;
}

int po_depress(int method, const uint8_t *in, uint64_t nbytes, uint32_t n,
	       int16_t *out, uint32_t *nout)
{
	switch (method) {
	case PM_SVB12:
	case PM_SVB12_ZD:
		(void) po_svb16_decode(in, n, method == PM_SVB12_ZD, out);
		*nout = n;
		return 0;
	case PM_SVB_ZD: { /* press.c:1600 */
		uint32_t *z32 = malloc(((size_t) n + 1) * sizeof *z32);
		uint16_t *z = malloc(((size_t) n + 1) * sizeof *z);
		uint32_t i;
		(void) po_svb32_decode(in, n, z32);
		for (i = 0; i < n; i++)
			z[i] = (uint16_t) z32[i];
		po_unzigdelta_u16(z, n, out);
		free(z);
		free(z32);
		*nout = n;
		return 0;
	}
	case PM_ZSTD_SVB_ZD:      return zstd_wrap_depress(PM_SVB_ZD, in, nbytes, n, out, nout);
	case PM_ZSTD_SVB12_ZD:    return zstd_wrap_depress(PM_SVB12_ZD, in, nbytes, n, out, nout);
	case PM_ZSTD_HASGAM_ZDQ:  return zstd_wrap_depress(PM_HASGAM_ZDQ, in, nbytes, n, out, nout);
	case PM_HASGAM_ZDQ:       return exzd_depress(in, nbytes, n, out, nout);
	case PM_VBE21_ZD: case PM_VBBE21_ZD: case PM_VBSBE21_ZD: case PM_VBSSE21_ZD:
	case PM_SHUFF_VBE21_ZD: case PM_SHUFF_VBBE21_ZD: case PM_SHUFF_VBSBE21_ZD:
	case PM_SHUFF_VBSSE21_ZD:
	case PM_RC_VBE21_ZD: case PM_RCC_VBE21_ZD: case PM_RCCM_VBBE21_ZD:
		return vb_family_depress(method, in, nbytes, n, out, nout);
	case PM_SLOW5_SVB_ZD: {
		/* slow5_press.c:1110 ptr_depress_svb_zd -> :1085 ptr_depress_svb: the count comes from
		 * the stream and the decoder must consume exactly the bytes it was given (:1098);
		 * streamvbyte_zigzag.c:34: 32-bit running sum, stored as int16.  n = room in `out`. */
		uint32_t cnt, i;
		uint32_t *z32;
		uint64_t need, used;
		int32_t prev = 0;
		if (nbytes < 4)
			return -1;
		cnt = get_u32(in);
		if (cnt > n)
			return -1;
		/* does the stream hold the bytes its keys announce?  (the reference would read past it) */
		need = (uint64_t) (cnt + 3) / 4;
		if (4 + need > nbytes)
			return -1;
		for (i = 0; i < cnt; i++)
			need += 1 + ((in[4 + (i >> 2)] >> (2 * (i & 3))) & 3);
		if (4 + need != nbytes)
			return -1;
		z32 = malloc(((size_t) cnt + 1) * sizeof *z32);
		used = po_svb32_decode(in + 4, cnt, z32);
		(void) used;
		for (i = 0; i < cnt; i++) {
			const int32_t val = (int32_t) (z32[i] >> 1) ^ -(int32_t) (z32[i] & 1);
			out[i] = (int16_t) (val + prev);
			prev += val;
		}
		free(z32);
		*nout = cnt;
		return 0;
	}
	}
	return -2;
}

/* ------------------------------------------------------------------ timing */

int po_time_batch(int method, const int16_t *sig, const uint64_t *off,
		  uint32_t nreads, double *press_s, double *depress_s,
		  uint64_t *press_bytes, int check)
{
	uint32_t r;
	*press_s = *depress_s = 0.0;
	*press_bytes = 0;
	for (r = 0; r < nreads; r++) {
		const int16_t *in = sig + off[r];
		uint32_t n = (uint32_t) (off[r + 1] - off[r]), nd = 0;
		uint64_t bound = po_bound(method, n), len = bound;
		uint8_t *out = malloc(bound + 64);
		int16_t *dec = malloc((uint64_t) n * 2 + 64);
		clock_t t0, t1;
		int ret;
		if (!out || !dec)
			return -1;
		t0 = clock();
		ret = po_press(method, in, n, out, &len);
		t1 = clock();
		if (ret)
			return ret;
		*press_s += (double) (t1 - t0) / CLOCKS_PER_SEC;
		t0 = clock();
		ret = po_depress(method, out, len, n, dec, &nd);
		t1 = clock();
		if (ret)
			return ret;
		*depress_s += (double) (t1 - t0) / CLOCKS_PER_SEC;
		*press_bytes += len;
		if (check && (nd != n || memcmp(dec, in, (size_t) n * 2)))
			return -3;
		free(out);
		free(dec);
	}
	return 0;
}
