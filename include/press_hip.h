/*
 * press_hip.h - C ABI of libpress_hip.so: the MI355X (gfx950) implementation of the
 * per-read transform -> pack -> entropy hot path of sashajenner/honours `press/`.
 *
 * Three groups of entry points:
 *
 *  (1) DROP-IN SYMBOLS.  Exactly the X_bound / X_press / X_depress triples that
 *      press/press.h declares for the hot-path methods (file:line cited on each),
 *      plus the four huffman.h functions press/test.c calls for the static table.
 *      Same names, prototypes, argument meaning and return codes, so the
 *      reference's unmodified harness (press/test.c, TEST() in press/test.h:10-37)
 *      links against this library for these methods.  Host pointers in, host
 *      pointers out; each call is H2D copy -> kernels -> D2H copy on a private
 *      stream, synchronous like the reference.
 *
 *  (2) BATCH API (press_hip_*).  Not in the reference: one call per batch of reads
 *      with device-resident buffers - the throughput path (one read per call is
 *      launch/PCIe-latency bound).  Reads are independent (thesis/probspace/
 *      focus.tex:177-183), so a batch is the natural device unit.
 *
 *  (3) BLOW5 FILES (press_hip_blow5_*, slow5_svb_zd_*).  The data format on the input
 *      side of the path: slow5lib's "svb-zd" signal codec on the
 *      device and a host reader / writer for the record framing, so that signals
 *      travel to and from the GPU as stored.
 *
 * Plain C: pointers and sizes only, no C++ or torch types.
 * Deviation from the reference, on purpose: X_press never writes past the capacity
 * passed in *nout (the reference ignores it and relies on X_bound being large
 * enough, press/test.c:1788).  When the stream does not fit, int methods return -1
 * and void methods set *nout = 0.
 */
#ifndef PRESS_HIP_H
#define PRESS_HIP_H

#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ======================================================================= (1) drop-in symbols */

/* ---- static Huffman table objects: press/huffman/huffman.h:22-51 (layout kept) ---- */
#ifndef HUFFMAN_HUFFMAN_H
#define MAX_SYMBOLS 256
typedef struct huffman_node_tag {
	unsigned char isLeaf;
	unsigned long count;
	struct huffman_node_tag *parent;
	union {
		struct {
			struct huffman_node_tag *zero, *one;
		};
		unsigned char symbol;
	};
} huffman_node;
typedef struct huffman_code_tag {
	unsigned long numbits; /* code length in bits */
	unsigned char *bits;   /* bit k of the code at bit k%8 of bits[k/8]; bit 0 is emitted first */
} huffman_code;
typedef huffman_code *SymbolEncoder[MAX_SYMBOLS];
#endif

/* huffman.h:58 / huffman.c:549 - parse the table file (format: SURVEY.md 8(a) a10) */
bool read_code_table(FILE *in, huffman_node **rootOut, unsigned int *dataBytesOut);
/* huffman.h:65 / huffman.c:353 */
void build_symbol_encoder(huffman_node *subtree, SymbolEncoder *pSF);
/* huffman.h:66 / huffman.c:145 - frees the codes AND *pSE itself, as the reference does */
void free_encoder(SymbolEncoder *pSE);
/* huffman.h:67 / huffman.c:117 */
void free_huffman_tree(huffman_node *subtree);

/* ---- svb16 without zd: press.h:317-320 (press.c:1568-1581) ---- */
uint64_t svb12_bound(uint64_t nin);
void svb12_press(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
void svb12_depress(const uint8_t *in, uint64_t nin /* samples */, int16_t *out);

/* ---- svb16-zd: press.h:348-352 (press.c:1678-1694).  nin of depress = samples ---- */
uint64_t svb12_zd_bound(uint64_t nin);
void svb12_zd_press(const int16_t *in, uint64_t nin, uint8_t *out, uint64_t *nout);
void svb12_zd_depress(const uint8_t *in, uint64_t nin, int16_t *out, uint64_t *nout);

/* ---- svb-zd (svb32): press.h:324-328 (press.c:1585-1611) ---- */
uint64_t svb_zd_bound_16(uint64_t nin);
void svb_zd_press_16(const int16_t *in, uint64_t nin, uint8_t *out, uint64_t *nout);
void svb_zd_depress_16(const uint8_t *in, uint64_t nin, int16_t *out, uint64_t *nout);

/* ---- VBZ = zstd-svb-zd: press.h:380-384 (press.c:1860-1910).  The inner stream is
 * produced on the GPU; the zstd stage is the third-party libzstd the reference itself
 * calls (press.c:1464), loaded at run time.  -1 when libzstd is not present. ---- */
uint64_t zstd_svb_zd_bound_16(uint32_t nin);
int zstd_svb_zd_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
int zstd_svb_zd_depress_16(const uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout);

/* ---- zstd-svb16-zd: press.h:404-408 (press.c:2020-2070) ---- */
uint64_t zstd_svb12_zd_bound(uint32_t nin);
int zstd_svb12_zd_press(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
int zstd_svb12_zd_depress(const uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout);

/* ---- exception split: press.h:498-526 (press.c:3411-3580 over :2679-3405).
 * depress: nin = compressed bytes; *nout = decoded samples ---- */
uint64_t vbe21_zd_bound_16(uint32_t nin);
void vbe21_zd_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
void vbe21_zd_depress_16(uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout);
uint64_t vbbe21_zd_bound_16(uint32_t nin);
void vbbe21_zd_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
void vbbe21_zd_depress_16(uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout);
uint64_t vbsbe21_zd_bound_16(uint32_t nin);
void vbsbe21_zd_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
void vbsbe21_zd_depress_16(uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout);
uint64_t vbsse21_zd_bound_16(uint32_t nin);
void vbsse21_zd_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
void vbsse21_zd_depress_16(uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout);

/* ---- static Huffman over the one-byte stream: press.h:634-662 (press.c:4409-4846).
 * se / root are the caller's table objects (read_code_table / build_symbol_encoder);
 * their codes are flattened and uploaded to the device (cached by content).
 * press returns 0 ok, -1 capacity, 1 Huffman-layer failure (as huffman.c does). ---- */
uint64_t shuffman_vbe21_zd_bound_16(uint32_t nin);
int shuffman_vbe21_zd_press_16(SymbolEncoder *se, const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
int shuffman_vbe21_zd_depress_16(huffman_node *root, uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout);
uint64_t shuffman_vbbe21_zd_bound_16(uint32_t nin);
int shuffman_vbbe21_zd_press_16(SymbolEncoder *se, const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
int shuffman_vbbe21_zd_depress_16(huffman_node *root, uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout);
uint64_t shuffman_vbsbe21_zd_bound_16(uint32_t nin);
int shuffman_vbsbe21_zd_press_16(SymbolEncoder *se, const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
int shuffman_vbsbe21_zd_depress_16(huffman_node *root, uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout);
uint64_t shuffman_vbsse21_zd_bound_16(uint32_t nin);
int shuffman_vbsse21_zd_press_16(SymbolEncoder *se, const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
int shuffman_vbsse21_zd_depress_16(huffman_node *root, uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout);

/* ---- vbe21 + order-0 adaptive range coder (TurboRC rcsenc / rcsdec): press.h:712-716 (press.c:5422-5500).
 * SURVEY 8f-1.  depress: *nout in = the exact sample count (press.c:5464). ---- */
uint64_t rc_vbe21_zd_bound_16(uint32_t nin);
void rc_vbe21_zd_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
void rc_vbe21_zd_depress_16(uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout);
/* ---- vbe21 + order-1 adaptive range coder (TurboRC rccsenc / rccsdec, rc_.c:181): press.h (press.c:5510-5580).
 * Same calling shape as rc_vbe21_zd. ---- */
uint64_t rcc_vbe21_zd_bound_16(uint32_t nin);
void rcc_vbe21_zd_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
void rcc_vbe21_zd_depress_16(uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout);
/* ---- vbbe21 + order 1-0 context mixing with secondary estimation (TurboRC rcmsenc / rcmsdec, rccm_.c:79,
 * instantiated by rccm_s.c): press.c:6901-7000 (the thesis's "rc01s-vbbe21-zd").  Same calling
 * shape as rc_vbe21_zd: *nout of depress in = the exact sample count (press.c:6986). ---- */
uint64_t rccm_vbbe21_zd_bound_16(uint32_t nin);
void rccm_vbbe21_zd_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
void rccm_vbbe21_zd_depress_16(uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout);

/* ---- ex-zd v0: press.h:960-964 (press.c:8461-8500 over ex_zd.c:403,495) ---- */
uint64_t hasgam_vbsse21_zdq_bound_16(uint32_t nin);
int hasgam_vbsse21_zdq_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
int hasgam_vbsse21_zdq_depress_16(uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout);

/* ---- zstd over ex-zd: press.h:976-980 (press.c:8549-8589) ---- */
uint64_t zstd_hasgam_vbsse21_zdq_bound_16(uint32_t nin);
int zstd_hasgam_vbsse21_zdq_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
int zstd_hasgam_vbsse21_zdq_depress_16(uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout);

/* ---- BLOW5's signal codec "svb-zd" (SURVEY 8f-2): slow5lib slow5_press.c:1054 ptr_compress_svb_zd /
 * :1110 ptr_depress_svb_zd, reached through slow5_ptr_compress_solo / slow5_ptr_depress_solo
 * (slow5_press.h:103-105) from slow5_rec_to_mem (slow5.c:3948) and the record parser.
 * Stream: u32 sample count, then streamvbyte (2-bit keys) of the 32-bit zig-zag deltas - the same
 * bytes as the pre-zstd buffer of zstd_svb_zd for signals whose jumps fit 16 bits.
 * press: *nout capacity in / bytes out; depress: *nout room in samples in / samples out. */
uint64_t slow5_svb_zd_bound(uint32_t nin);
int slow5_svb_zd_press(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout);
int slow5_svb_zd_depress(const uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout);
/* slow5lib's own calling shape: malloc'd result (free() it), counts in bytes; NULL on failure */
void *press_hip_slow5_ptr_compress_svb_zd(const int16_t *ptr, size_t count, size_t *n);
void *press_hip_slow5_ptr_depress_svb_zd(const uint8_t *ptr, size_t count, size_t *n);

/* ======================================================================= (2) batch API */

/* method ids (== oracle/press_methods.h); names are the reference's code names */
enum press_hip_method {
	PRESS_HIP_SVB12            = 0,
	PRESS_HIP_SVB12_ZD         = 1,
	PRESS_HIP_SVB_ZD           = 2,
	PRESS_HIP_ZSTD_SVB_ZD      = 3,  /* "VBZ".  Per-read symbols: libzstd on the host, the reference's bytes.
	                                    Batch API: the zstd frames are made and read ON THE DEVICE (below) */
	PRESS_HIP_ZSTD_SVB12_ZD    = 4,  /* zstd(svb16-zd): as method 3 (per-read: libzstd on the host; batch: frames on the device) */
	PRESS_HIP_VBE21_ZD         = 5,
	PRESS_HIP_VBBE21_ZD        = 6,
	PRESS_HIP_VBSBE21_ZD       = 7,
	PRESS_HIP_VBSSE21_ZD       = 8,
	PRESS_HIP_SHUFF_VBE21_ZD   = 9,
	PRESS_HIP_SHUFF_VBBE21_ZD  = 10,
	PRESS_HIP_SHUFF_VBSBE21_ZD = 11,
	PRESS_HIP_SHUFF_VBSSE21_ZD = 12,
	PRESS_HIP_HASGAM_ZDQ       = 13,
	PRESS_HIP_ZSTD_HASGAM_ZDQ  = 14, /* zstd(ex-zd): as method 3 */
	PRESS_HIP_SLOW5_SVB_ZD     = 15, /* BLOW5's signal codec (section 3) */
	PRESS_HIP_RC_VBE21_ZD      = 16, /* vbe21 + order-0 range coder: one read per lane (serial format) */
	PRESS_HIP_RCC_VBE21_ZD     = 17, /* vbe21 + order-1 range coder: one read per workgroup, its 128 KiB of state in LDS */
	PRESS_HIP_RCCM_VBBE21_ZD   = 18, /* vbbe21 + order 1-0 context mixing + SSE: one read per workgroup, 137 KiB of state in LDS */
	PRESS_HIP_NMETHODS         = 19
};

#define PRESS_HIP_OK        0
#define PRESS_HIP_EARG     (-2)  /* bad argument (method, alignment, NULL) */
#define PRESS_HIP_EHIP     (-3)  /* a HIP runtime call failed */
#define PRESS_HIP_ENOTABLE (-4)  /* static-Huffman method without a table */
#define PRESS_HIP_FAILED   UINT64_MAX /* per-read length on failure (capacity, malformed stream) */

/* last HIP / library error as text (thread local) */
const char *press_hip_last_error(void);

/* select the device for this process (default: current device).  One process per GPU: the
 * library keeps ONE context (device, stream, scratch).  Calls may come from any host thread -
 * every entry point re-selects the device for the calling thread and entry points are
 * serialised by a lock - but two threads cannot run batches concurrently. */
int press_hip_set_device(int device);
/* stream (hipStream_t, as void*) the batch calls enqueue on; NULL is HIP's default
 * (null) stream.  Until this is called the library uses a private non-blocking stream;
 * press_hip_reset_stream() returns to it. */
int press_hip_set_stream(void *stream);
int press_hip_reset_stream(void);
void *press_hip_get_stream(void);
/* block until everything enqueued by batch calls has finished */
int press_hip_synchronize(void);

/* static Huffman table for the batch API: file in the format of press/NA12878_zd.huffman
 * (huffman.c:549; it may list fewer than 256 symbols) */
int press_hip_load_table_file(const char *path);
/* or 256 {length, code bits} pairs; bit k of bits[s] is the k-th emitted bit.
 * len[s] == 0: symbol s has no code - a read fails (press: PRESS_HIP_FAILED / return 1) only if that
 * value occurs in it (the reference dereferences a NULL code there, huffman.c:860).
 * Limit of the device tables: codes of at most 24 bits (huffman.c takes up to 255); longer -> EARG,
 * and the shuffman_* drop-in symbols return 1 for such a table. */
int press_hip_set_table(const uint32_t len[256], const uint64_t bits[256]);

/* X_bound of the reference for `method` (what press/test.c allocates) */
uint64_t press_hip_bound(int method, uint32_t n);

/*
 * Compress nreads reads.
 *   sig      int16 samples; read r = sig[off[r] .. off[r]+n[r]).  off (nreads entries)
 *            must be multiples of 8 samples and sig 16-byte aligned: every sample load
 *            is a coalesced 16-byte access
 *   n        samples per read (nreads entries)
 *   total_samples  extent of sig in samples: max(off[r]+n[r]) rounded up as the caller
 *            likes; sizes the library's scratch (with device-resident offsets the
 *            library cannot read them without a synchronisation)
 *   out      output arena; read r may use out[out_off[r] .. out_off[r+1]) - its
 *            capacity; out_off has nreads+1 entries
 *   out_len  per read: bytes produced, or PRESS_HIP_FAILED (capacity too small, ...)
 * device_resident != 0: every pointer is a device pointer, the call only enqueues
 * work on the current stream (no synchronisation; scratch is (re)allocated only when a
 * batch is larger than any before it); == 0: host pointers, synchronous.
 */
int press_hip_press_batch(int method, const int16_t *sig, const uint64_t *off, const uint32_t *n,
			  uint32_t nreads, uint64_t total_samples, uint8_t *out,
			  const uint64_t *out_off, uint64_t *out_len, int device_resident);

/*
 * Decompress nreads streams.
 *   in/in_off/in_len  stream r = in[in_off[r] .. in_off[r]+in_len[r])
 *   sig/off/n         output layout as above; n[r] is the room for read r in samples
 *                     AND, for the svb methods, the sample count (their streams do
 *                     not carry it, press.c:1689)
 *   out_n             per read: samples decoded, or UINT32_MAX on a malformed stream
 */
int press_hip_depress_batch(int method, const uint8_t *in, const uint64_t *in_off,
			    const uint64_t *in_len, uint32_t nreads, int16_t *sig,
			    const uint64_t *off, const uint32_t *n, uint64_t total_samples,
			    uint32_t *out_n, int device_resident);

/*
 * PRESS_HIP_ZSTD_SVB_ZD, _ZSTD_SVB12_ZD and _ZSTD_HASGAM_ZDQ in the two calls above (SURVEY.md 8f-3,
 * replaces the ZSTD_compress / ZSTD_decompress calls of press.c:1860-1910, 2020-2070, 8549-8589 for batches):
 *   press    writes one standard zstd frame (RFC 8878) per read whose content is the buffer
 *            the reference hands to ZSTD_compress ([u32 n][svb stream], or the ex-zd stream): a raw
 *            block with the count (ex-zd: header + exception section), RLE blocks for the svb key
 *            bytes, the one-byte values in Huffman-coded literal blocks of 16 KiB
 *            (one table per read, no sequences).  Any zstd decoder reads it - the reference's
 *            zstd_svb_zd_depress_16 included; the BYTES are not libzstd's (they never were
 *            pinned: they depend on the libzstd version).  Size <= 9 + L + 3 * ceil(L / 128 KiB)
 *            for a content of L bytes, well inside press_hip_bound().
 *   depress  reads any single zstd frame of such a buffer on the device: the frames above and
 *            ZSTD_compress's own (the reference's streams, press.c:1462-1469) - Huffman literals,
 *            FSE-coded sequences with all four table modes, repeat offsets - one wave per frame walks
 *            the blocks, the literals are decoded in parallel, one wave per frame carries out its
 *            sequences.  Only frames with a dictionary, a 12-bit Huffman table, more sequences than the
 *            scratch takes, or several frames in one stream are decompressed by libzstd on the host
 *            inside the call.  NOTE: a zstd depress call waits on the host until the frames have been
 *            walked (the count of such frames comes back through page-locked memory behind an event), also
 *            when device_resident != 0 - the device meanwhile goes on with the rest of the batch, and the
 *            call does not wait for that; only a batch WITH such frames synchronises the stream.
 *            n[r] is the room in samples; the count in the stream decides (press.c:1901).  Content
 *            checksums are not verified.
 */
/* frames the last zstd depress batch left to libzstd on the host (0 for this library's frames and for
 * ZSTD_compress's) */
uint32_t press_hip_zstd_host_frames(void);

/*
 * Host buffers (device_resident == 0).  Ordinary (pageable) memory is copied through two page-locked
 * staging buffers of 32 MiB (the DMA of one overlaps the host's memcpy into / out of the other);
 * compressed streams cross the link packed back to back, whatever the slots' sizes.  Buffers from
 * press_hip_host_alloc() are page-locked: `sig` is then copied by ONE DMA in press, and in depress the
 * decoded samples are written straight into it (the alignment padding between reads is overwritten).
 * Both calls return when the caller's buffers are complete.
 */
void *press_hip_host_alloc(uint64_t bytes); /* hipHostMalloc; NULL on failure */
void press_hip_host_free(void *p);

/* bytes of device scratch the two calls above keep for a batch of this shape (informational) */
uint64_t press_hip_workspace_bytes(int method, uint64_t total_samples, uint32_t nreads);

/* Measurement aid (bench.py): when enabled, every batch call brackets its dominant kernel
 * (the svb encode / decode kernel, the one-byte-stream kernels of the exception methods)
 * with HIP events on the launch stream.  press_hip_kernel_times(which, ms, max) returns the
 * elapsed times recorded since timing was (re-)enabled; which: 0 = press, 1 = depress. */
int press_hip_kernel_timing(int enable);
int press_hip_kernel_times(int which, float *ms, int max);

/* release every device and host resource held by the library (all scratch buffers, the device
 * copy of the Huffman table, the private stream).  press_hip_set_device(other) goes through it. */
void press_hip_shutdown(void);
/* number of scratch buffers the library keeps; *bytes (may be NULL) = device bytes they hold now
 * (0 after press_hip_shutdown) */
uint32_t press_hip_scratch_buffers(uint64_t *bytes);

/* ======================================================================= (3) BLOW5 files (host) */

/*
 * Reader for BLOW5 files that hands out the signal fields AS STORED: what the
 * reference's loader decodes on the CPU (slow5_open slow5.h:345, slow5_get_next :446,
 * slow5_rec.raw_signal :274) goes to the device compressed and is decoded there with
 * press_hip_depress_batch(PRESS_HIP_SLOW5_SVB_ZD, ...).  Record compression none / zlib / zstd,
 * signal compression none / svb-zd.  Host code, usable without a GPU.
 */
typedef struct press_hip_blow5 press_hip_blow5;
#define PRESS_HIP_BLOW5_ID_LEN 64 /* bytes per read id handed out (NUL padded) */
int press_hip_blow5_open(const char *path, press_hip_blow5 **out);
void press_hip_blow5_close(press_hip_blow5 *f);
/* record method: 0 none, 1 zlib, 2 zstd; signal method: 0 none (int16 samples), 1 svb-zd */
int press_hip_blow5_methods(const press_hip_blow5 *f, int *record_method, int *signal_method);
/* Next batch: up to max_reads signal fields copied back to back into arena (each starts on a
 * 16-byte boundary), sig_off / sig_len in bytes, n_samples per read, read_ids (may be NULL)
 * max_reads x PRESS_HIP_BLOW5_ID_LEN chars.  *got = reads delivered; 0 at the end of the file. */
int press_hip_blow5_next(press_hip_blow5 *f, uint32_t max_reads, uint8_t *arena, uint64_t arena_cap,
			 uint64_t *sig_off, uint64_t *sig_len, uint32_t *n_samples, char *read_ids, uint32_t *got);
const char *press_hip_blow5_last_error(void);
/* Whole inflated records (for a transcoder): sig_pos / sig_len locate the signal field in
 * record k, the u64 length field sits 8 bytes in front of it. */
int press_hip_blow5_next_records(press_hip_blow5 *f, uint32_t max_reads, uint8_t *arena, uint64_t arena_cap,
				 uint64_t *rec_off, uint64_t *rec_len, uint64_t *sig_pos, uint64_t *sig_len,
				 uint32_t *n_samples, uint32_t *got);
/* Writer (slow5_write / slow5_rec_to_mem, slow5.c:3903-4010): header copied from `like`; records
 * framed and - record_method 1 - deflated; signal_method says what `sig` holds (0: int16 samples,
 * 1: svb-zd, e.g. the output of press_hip_press_batch(PRESS_HIP_SLOW5_SVB_ZD, ...)). */
typedef struct press_hip_blow5_writer press_hip_blow5_writer;
int press_hip_blow5_create(const char *path, const press_hip_blow5 *like, int record_method, int signal_method,
			   press_hip_blow5_writer **out);
int press_hip_blow5_write(press_hip_blow5_writer *w, const uint8_t *pre, uint64_t pre_len, const uint8_t *sig,
			  uint64_t sig_len, const uint8_t *post, uint64_t post_len);
/* n records at once: framed and deflated by a pool of host threads, written in the order given */
int press_hip_blow5_write_batch(press_hip_blow5_writer *w, uint32_t n, const uint8_t *const *pre, const uint64_t *pre_len,
				const uint8_t *const *sig, const uint64_t *sig_len, const uint8_t *const *post,
				const uint64_t *post_len);
/* enable != 0: keep slow5lib's index (read id -> record offset and size, slow5_idx.c:269 slow5_idx_write) while
 * writing; press_hip_blow5_finish leaves it as <path>.idx, so that the reference's slow5_idx_load / slow5_get
 * (slow5.h:375, 423) find the reads of the file without building the index themselves */
int press_hip_blow5_index(press_hip_blow5_writer *w, int enable);
/* host threads that inflate records (reader; 0 = as many as the host offers, at most 32) */
int press_hip_blow5_threads(press_hip_blow5 *f, int threads);
int press_hip_blow5_finish(press_hip_blow5_writer *w); /* end marker, the index if asked for, close, free */

#ifdef __cplusplus
}
#endif
#endif /* PRESS_HIP_H */
