/*
 * ref_shim.c - thin uniform entry points over the REAL reference implementation.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is compiled together with the reference's
 * own sources, in place under /root/reference/press (never copied), into
 * oracle/_ref/libpress_ref.so by oracle/Makefile.  It exists so that tests and
 * tools can (1) validate the CPU restatement in oracle/press_oracle.c against the
 * reference itself, (2) generate the golden vectors committed under tests/golden/
 * and (3) serve as bench.py's cpu_baseline of kind "reference".
 *
 * Only ref_* symbols are exported (oracle/ref_exports.map); everything the hot
 * path does not reference is dropped by --gc-sections, so none of the vendored
 * bzip2/FLAC/TurboPFor/TurboRC/lzma2/sigtk libraries is needed.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "press.h"            /* the reference's press/press.h (via -I) */
#include "huffman/huffman.h"  /* the reference's press/huffman/huffman.h */
#include "press_methods.h"
#include <slow5/slow5_press.h> /* the reference's slow5lib (sigtk/slow5lib/include), for PM_SLOW5_SVB_ZD */

static huffman_node *g_root;
static SymbolEncoder *g_se;

/* press/test.c:3786-3791 does exactly this per read */
int ref_load_table(const char *path)
{
	unsigned int data_bytes = 0;
	FILE *fp = fopen(path, "r");
	if (!fp)
		return -1;
	if (g_root) {
		free_encoder(g_se);
		free_huffman_tree(g_root);
		g_root = NULL;
		g_se = NULL;
	}
	if (!read_code_table(fp, &g_root, &data_bytes)) {
		fclose(fp);
		return -1;
	}
	fclose(fp);
	g_se = calloc(1, sizeof(SymbolEncoder));
	build_symbol_encoder(g_root, g_se);
	return 0;
}

/* expose the code table the reference built: len[i], and code bits (bit k of
 * the code in bit k of the returned word, i.e. first-emitted bit in bit 0) */
int ref_table_code(int sym, uint32_t *len, uint64_t *bits)
{
	unsigned long k;
	huffman_code *c;
	if (!g_se || sym < 0 || sym > 255 || !(*g_se)[sym])
		return -1;
	c = (*g_se)[sym];
	*len = (uint32_t) c->numbits;
	*bits = 0;
	for (k = 0; k < c->numbits && k < 64; k++)
		if (c->bits[k / 8] & (1u << (k % 8)))
			*bits |= (uint64_t) 1 << k;
	return 0;
}

uint64_t ref_bound(int method, uint32_t n)
{
	switch (method) {
	case PM_SVB12:            return svb12_bound(n);
	case PM_SVB12_ZD:         return svb12_zd_bound(n);
	case PM_SVB_ZD:           return svb_zd_bound_16(n);
	case PM_ZSTD_SVB_ZD:      return zstd_svb_zd_bound_16(n);
	case PM_ZSTD_SVB12_ZD:    return zstd_svb12_zd_bound(n);
	case PM_VBE21_ZD:         return vbe21_zd_bound_16(n);
	case PM_VBBE21_ZD:        return vbbe21_zd_bound_16(n);
	case PM_VBSBE21_ZD:       return vbsbe21_zd_bound_16(n);
	case PM_VBSSE21_ZD:       return vbsse21_zd_bound_16(n);
	case PM_SHUFF_VBE21_ZD:   return shuffman_vbe21_zd_bound_16(n);
	case PM_SHUFF_VBBE21_ZD:  return shuffman_vbbe21_zd_bound_16(n);
	case PM_SHUFF_VBSBE21_ZD: return shuffman_vbsbe21_zd_bound_16(n);
	case PM_SHUFF_VBSSE21_ZD: return shuffman_vbsse21_zd_bound_16(n);
	case PM_HASGAM_ZDQ:       return hasgam_vbsse21_zdq_bound_16(n);
	case PM_ZSTD_HASGAM_ZDQ:  return zstd_hasgam_vbsse21_zdq_bound_16(n);
	case PM_SLOW5_SVB_ZD:     return (uint64_t) (n + 3) / 4 + (uint64_t) n * 4 + 4; /* slow5_press.c:1037 */
	case PM_RC_VBE21_ZD:      return rc_vbe21_zd_bound_16(n);
	case PM_RCC_VBE21_ZD:     return rcc_vbe21_zd_bound_16(n);
	case PM_RCCM_VBBE21_ZD:   return rccm_vbbe21_zd_bound_16(n);
	}
	return 0;
}

/* *nout: capacity in, produced bytes out - as the harness passes it (test.c:1782) */
int ref_press(int method, const int16_t *in, uint32_t n, uint8_t *out,
	      uint64_t *nout)
{
	switch (method) {
	case PM_SVB12:            svb12_press(in, n, out, nout); return 0;
	case PM_SVB12_ZD:         svb12_zd_press(in, n, out, nout); return 0;
	case PM_SVB_ZD:           svb_zd_press_16(in, n, out, nout); return 0;
	case PM_ZSTD_SVB_ZD:      return zstd_svb_zd_press_16(in, n, out, nout);
	case PM_ZSTD_SVB12_ZD:    return zstd_svb12_zd_press(in, n, out, nout);
	case PM_VBE21_ZD:         vbe21_zd_press_16(in, n, out, nout); return 0;
	case PM_VBBE21_ZD:        vbbe21_zd_press_16(in, n, out, nout); return 0;
	case PM_VBSBE21_ZD:       vbsbe21_zd_press_16(in, n, out, nout); return 0;
	case PM_VBSSE21_ZD:       vbsse21_zd_press_16(in, n, out, nout); return 0;
	case PM_SHUFF_VBE21_ZD:   return shuffman_vbe21_zd_press_16(g_se, in, n, out, nout);
	case PM_SHUFF_VBBE21_ZD:  return shuffman_vbbe21_zd_press_16(g_se, in, n, out, nout);
	case PM_SHUFF_VBSBE21_ZD: return shuffman_vbsbe21_zd_press_16(g_se, in, n, out, nout);
	case PM_SHUFF_VBSSE21_ZD: return shuffman_vbsse21_zd_press_16(g_se, in, n, out, nout);
	case PM_HASGAM_ZDQ:       return hasgam_vbsse21_zdq_press_16(in, n, out, nout);
	case PM_ZSTD_HASGAM_ZDQ:  return zstd_hasgam_vbsse21_zdq_press_16(in, n, out, nout);
	case PM_RC_VBE21_ZD:      rc_vbe21_zd_press_16(in, n, out, nout); return 0;
	case PM_RCC_VBE21_ZD:     rcc_vbe21_zd_press_16(in, n, out, nout); return 0;
	case PM_RCCM_VBBE21_ZD:   rccm_vbbe21_zd_press_16(in, n, out, nout); return 0;
	case PM_SLOW5_SVB_ZD: { /* what slow5_rec_to_mem does with a read's signal (slow5.c:3948) */
		size_t len = 0;
		void *buf = slow5_ptr_compress_solo(SLOW5_COMPRESS_SVB_ZD, in, (size_t) n * sizeof *in, &len);
		if (!buf)
			return -1;
		if (len > *nout) {
			free(buf);
			return -1;
		}
		memcpy(out, buf, len);
		free(buf);
		*nout = len;
		return 0;
	}
	}
	return -2;
}

/* n: the true sample count (what the harness knows: svb* take it as `nin`,
 * the others as the capacity in *nout).  nbytes: compressed length. */
int ref_depress(int method, uint8_t *in, uint64_t nbytes, uint32_t n,
		int16_t *out, uint32_t *nout)
{
	uint64_t n64 = n;
	int ret = 0;
	*nout = n;
	switch (method) {
	case PM_SVB12:            svb12_depress(in, n, out); break;
	case PM_SVB12_ZD:         svb12_zd_depress(in, n, out, &n64); *nout = (uint32_t) n64; break;
	case PM_SVB_ZD:           svb_zd_depress_16(in, n, out, &n64); *nout = (uint32_t) n64; break;
	case PM_ZSTD_SVB_ZD:      ret = zstd_svb_zd_depress_16(in, nbytes, out, nout); break;
	case PM_ZSTD_SVB12_ZD:    ret = zstd_svb12_zd_depress(in, nbytes, out, nout); break;
	case PM_VBE21_ZD:         vbe21_zd_depress_16(in, nbytes, out, nout); break;
	case PM_VBBE21_ZD:        vbbe21_zd_depress_16(in, nbytes, out, nout); break;
	case PM_VBSBE21_ZD:       vbsbe21_zd_depress_16(in, nbytes, out, nout); break;
	case PM_VBSSE21_ZD:       vbsse21_zd_depress_16(in, nbytes, out, nout); break;
	case PM_SHUFF_VBE21_ZD:   ret = shuffman_vbe21_zd_depress_16(g_root, in, nbytes, out, nout); break;
	case PM_SHUFF_VBBE21_ZD:  ret = shuffman_vbbe21_zd_depress_16(g_root, in, nbytes, out, nout); break;
	case PM_SHUFF_VBSBE21_ZD: ret = shuffman_vbsbe21_zd_depress_16(g_root, in, nbytes, out, nout); break;
	case PM_SHUFF_VBSSE21_ZD: ret = shuffman_vbsse21_zd_depress_16(g_root, in, nbytes, out, nout); break;
	case PM_HASGAM_ZDQ:       ret = hasgam_vbsse21_zdq_depress_16(in, nbytes, out, nout); break;
	case PM_ZSTD_HASGAM_ZDQ:  ret = zstd_hasgam_vbsse21_zdq_depress_16(in, nbytes, out, nout); break;
	case PM_RC_VBE21_ZD:      rc_vbe21_zd_depress_16(in, nbytes, out, nout); break;
	case PM_RCC_VBE21_ZD:     rcc_vbe21_zd_depress_16(in, nbytes, out, nout); break;
	case PM_RCCM_VBBE21_ZD:   rccm_vbbe21_zd_depress_16(in, nbytes, out, nout); break;
	case PM_SLOW5_SVB_ZD: {
		size_t len = 0;
		void *buf = slow5_ptr_depress_solo(SLOW5_COMPRESS_SVB_ZD, in, (size_t) nbytes, &len);
		if (!buf)
			return -1;
		if (len / 2 > n) {
			free(buf);
			return -1;
		}
		memcpy(out, buf, len);
		free(buf);
		*nout = (uint32_t) (len / 2);
		break;
	}
	default: return -2;
	}
	return ret;
}

/*
 * Timed pass over a batch of reads with the harness's semantics
 * (press/test.c:1756-1815): fresh malloc(X_bound(n)) per read, clock() around
 * X_press and X_depress only.  Returns 0 on success and the summed seconds /
 * bytes through the out-parameters.  Used by bench.py's cpu_baseline leg.
 */
int ref_time_batch(int method, const int16_t *sig, const uint64_t *off,
		   uint32_t nreads, double *press_s, double *depress_s,
		   uint64_t *press_bytes, int check)
{
	uint32_t r;
	*press_s = *depress_s = 0.0;
	*press_bytes = 0;
	for (r = 0; r < nreads; r++) {
		const int16_t *in = sig + off[r];
		uint32_t n = (uint32_t) (off[r + 1] - off[r]);
		uint64_t bound = ref_bound(method, n);
		uint8_t *out = malloc(bound + 64);
		int16_t *dec = malloc((uint64_t) n * 2 + 64);
		uint64_t len = bound;
		uint32_t nd = n;
		clock_t t0, t1;
		int ret;
		if (!out || !dec)
			return -1;
		t0 = clock();
		ret = ref_press(method, in, n, out, &len);
		t1 = clock();
		if (ret)
			return ret;
		*press_s += (double) (t1 - t0) / CLOCKS_PER_SEC;
		t0 = clock();
		ret = ref_depress(method, out, len, n, dec, &nd);
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
