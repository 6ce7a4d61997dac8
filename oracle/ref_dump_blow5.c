/*
 * ref_dump_blow5.c - dump the raw signals of a SLOW5/BLOW5 file with the
 * reference's own loader (press/sigtk/slow5lib, slow5.h:345,446,454,354).
 *
 * TEST INFRASTRUCTURE ONLY: built into oracle/_ref/ref_dump_blow5 by
 * oracle/Makefile from slow5lib sources in place; used once by
 * tests/golden/make_golden.py to turn data/three-reads.blow5 into the
 * committed fixture tests/golden/three_reads.* (signals are data, not code).
 *
 * With read ids on the command line: those reads through the index (slow5_idx_load / slow5_get).
 *
 * output (little endian): u32 nreads | per read: u32 idlen, id bytes, u64 n, n x int16
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <slow5/slow5.h>

int main(int argc, char **argv)
{
	struct slow5_file *sp;
	struct slow5_rec *rec = NULL;
	FILE *out;
	uint32_t nreads = 0;
	long pos;

	if (argc < 3) {
		fprintf(stderr, "usage: %s FILE.blow5 OUT.bin [READ_ID ...]\n", argv[0]);
		return 2;
	}
	sp = slow5_open(argv[1], "r");
	if (!sp)
		return 1;
	out = fopen(argv[2], "wb");
	if (!out)
		return 1;
	fwrite(&nreads, 4, 1, out);
	if (argc > 3) {
		/* random access through the index (slow5_idx_load slow5.h:375 reads FILE.blow5.idx if it is there and
		 * builds it otherwise; slow5_get :423): the reads named on the command line, in that order */
		int a;
		if (slow5_idx_load(sp) != 0)
			return 3;
		for (a = 3; a < argc; a++) {
			uint32_t idlen;
			uint64_t n;
			if (slow5_get(argv[a], &rec, sp) != 0)
				return 4;
			idlen = (uint32_t) strlen(rec->read_id);
			n = rec->len_raw_signal;
			fwrite(&idlen, 4, 1, out);
			fwrite(rec->read_id, 1, idlen, out);
			fwrite(&n, 8, 1, out);
			fwrite(rec->raw_signal, 2, n, out);
			nreads++;
		}
		slow5_idx_unload(sp);
	} else
	while (slow5_get_next(&rec, sp) >= 0) {
		uint32_t idlen = (uint32_t) strlen(rec->read_id);
		uint64_t n = rec->len_raw_signal;
		fwrite(&idlen, 4, 1, out);
		fwrite(rec->read_id, 1, idlen, out);
		fwrite(&n, 8, 1, out);
		fwrite(rec->raw_signal, 2, n, out);
		nreads++;
	}
	slow5_rec_free(rec);
	slow5_close(sp);
	pos = ftell(out);
	fseek(out, 0, SEEK_SET);
	fwrite(&nreads, 4, 1, out);
	fseek(out, pos, SEEK_SET);
	fclose(out);
	fprintf(stderr, "%u reads\n", nreads);
	return 0;
}
