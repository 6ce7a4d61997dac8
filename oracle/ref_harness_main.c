/*
 * ref_harness_main.c - drive the REFERENCE's own benchmark harness against libpress_hip.so.
 *
 * TEST INFRASTRUCTURE / integration proof.  oracle/Makefile (target `harness`) compiles
 * the reference's unmodified press/test.c in place (its main() renamed away with
 * -Dmain=...), the reference's slow5lib loader, and this file into
 * oracle/_ref/press_test_hip, linked ONLY against honours_amd/libpress_hip.so - none of
 * the reference's codec objects (press.o, trans.o, ex_zd.o, huffman.o, streamvbyte*) is
 * linked, so every X_bound / X_press / X_depress and every huffman.h call made by the
 * reference's test_X functions below resolves to the HIP library.  Unreferenced test_X
 * functions (bzip2, FLAC, ...) are dropped by --gc-sections.
 *
 * The TEST() macro, struct result and the TSV writers are the reference's
 * (press/test.h:10-37, press/test.c:21-47), used through #include, not copied.
 *
 *   cd <dir with NA12878_zd.huffman> && press_test_hip FILE.blow5     (test.c:3786)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "test.h" /* the reference's press/test.h */

/* defined in the reference's test.c */
void init_res(struct result *res);
void fwrite_res_hdr(FILE *fp);
void fwrite_res(FILE *fp, struct result *res);
#define DECL(m) int test_##m(const int16_t *sigs, const uint32_t nr_sigs, struct result *res)
DECL(svb_zd);
DECL(svb12_zd);
DECL(zstd_svb_zd);
DECL(zstd_svb12_zd);
DECL(vbe21_zd);
DECL(vbbe21_zd);
DECL(vbsbe21_zd);
DECL(vbsse21_zd);
DECL(shuffman_vbe21_zd);
DECL(shuffman_vbbe21_zd);
DECL(shuffman_vbsbe21_zd);
DECL(shuffman_vbsse21_zd);
DECL(hasgam_vbsse21_zdq);
DECL(zstd_hasgam_vbsse21_zdq);

int main(int argc, char **argv)
{
	FILE *fp = stdout;
	struct result res;

	if (argc != 2) {
		fprintf(stderr, "usage: %s (S|B)LOW5_FILE\n", argv[0]);
		return 1;
	}
	fwrite_res_hdr(fp);
	/* the hot-path subset of press/test.c:6110-6198, same order */
	TEST(svb_zd, &res, fp);
	TEST(svb12_zd, &res, fp);
	TEST(vbe21_zd, &res, fp);
	TEST(vbbe21_zd, &res, fp);
	TEST(zstd_svb_zd, &res, fp);
	TEST(zstd_svb12_zd, &res, fp);
	TEST(vbsbe21_zd, &res, fp);
	TEST(vbsse21_zd, &res, fp);
	TEST(shuffman_vbe21_zd, &res, fp);
	TEST(shuffman_vbbe21_zd, &res, fp);
	TEST(shuffman_vbsbe21_zd, &res, fp);
	TEST(shuffman_vbsse21_zd, &res, fp);
	TEST(hasgam_vbsse21_zdq, &res, fp);
	TEST(zstd_hasgam_vbsse21_zdq, &res, fp);
	return 0;
}
