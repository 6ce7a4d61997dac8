// press_internal.h - shared between the kernel files (press_*.hip) and the C-ABI
// host layer (press_abi.hip).  Not installed; the public interface is include/press_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ph {

// Exception-section encodings of the "vb1e2" family (press.c:2679-3405) and ex-zd (ex_zd.c:9).
enum ExFmt : int {
	EXF_VBE21 = 0,   // nex x u32 pos, nex x u16 raw value
	EXF_VBBE21 = 1,  // bit-packed position deltas, bit-packed (value-256)
	EXF_VBSBE21 = 2, // svb32 position deltas, bit-packed (value-256)
	EXF_VBSSE21 = 3, // svb32 position deltas, svb16 (value-256)
	EXF_EXZD = 4     // svb32 position deltas, svb32 (value-256); one exception = 2 x u32
};

// Per-read record shared by the passes of the exception-split methods (device memory).
struct ReadMeta {
	uint32_t nex;     // exceptions among zd[1..n)
	uint32_t ored;    // OR of all samples (ex-zd qts, ex_zd.c:358-381)
	uint32_t zd0;     // zd[0], stored raw in the header
	uint32_t q;       // ex-zd shift (0 for the other methods)
	uint32_t hdr;     // bytes in front of the "u32 nex" field: 2, or 12 for ex-zd
	uint32_t seclen;  // bytes of "u32 nex || exception section"
	uint32_t nlow;    // one-byte values in the stream (decode side)
	uint32_t status;  // 0 = ok, anything else = this read failed
};

// Static-Huffman tables on the device (built on the host from the 256 {len,bits} pairs).
constexpr int HUF_LUT_BITS = 12;
constexpr int HUF_L2_ENTRIES = 3072; // the NA12878 table needs 2848
constexpr int HUF_L2_IDS = 64;       // long-code prefixes with a second-level table (the rest walk the trie)
struct HuffDev {
	uint32_t enc[256];                 // code bits (bit k = k-th emitted bit) | len << 24
	uint16_t lut[1 << HUF_LUT_BITS];   // sym | len << 8 for codes <= 12 bits, 0xFFFF: walk the trie
	int16_t child[1024][2];            // binary trie, node 0 = root, -1 = none
	int16_t leaf[1024];                // symbol at a leaf, -1 otherwise
	uint32_t minlen, maxlen;           // shortest / longest code
	// second level for codes longer than HUF_LUT_BITS: lut[prefix] = 0x8000 | id, then
	// lut2[l2off[id] + next l2bits[id] stream bits] = sym | len << 8 (0xFFFF: no such code)
	uint16_t lut2[HUF_L2_ENTRIES];
	uint16_t l2off[256];
	uint8_t l2bits[256];
	// two-symbol first level for the parallel decoder:
	//   d1 | adv << 8 | d2 << 16 | len1 << 24 | HUF_TWO,  adv = len1 + len2 with HUF_TWO (two whole
	//   codes fit in HUF_LUT_BITS bits), else adv = len1 and d2 = 0;  d = the delta the symbol stands for
	//   (zig-zag undone, a signed byte)
	// (the two deltas sit in the low bytes of the two register halves: ds_write_b8 / _d16_hi store
	// them without a shift); a long code's prefix: HUF_LONG | l2bits << 12 | l2off of its
	// second-level table; 0xFFFFFFFF: walk the trie
	alignas(16) uint32_t lut32[1 << HUF_LUT_BITS];
	// first level of k_huf_sync (lengths and sample deltas, no symbols): every whole code that fits in
	// HUF_LUT_BITS bits at once (at most HUF_M_MAXN of them):
	//   total bits | codes << 4 | bits of the first code << 8 | d1 << 12 | dsum << 20
	//   d1 = delta the first code's symbol stands for (8 bits, signed: zig-zag undone), dsum = sum of the
	//   deltas of all codes of the entry (11 bits, signed);
	// a long code's prefix: HUF_MLONG | l2off | l2bits << 12 (length and delta of the code in
	// l2ld[l2off + the next l2bits stream bits]); 0xFFFFFFFF: walk the trie
	alignas(16) uint32_t mlut[1 << HUF_LUT_BITS];
	alignas(16) uint16_t l2ld[HUF_L2_ENTRIES]; // bits of the code | delta << 8 (signed), 0xFFFF: no such code
	// the same first level laid out as an INCREMENT of the scan's one accumulator (position | codes << 9 |
	// delta sum << 16): total bits | codes << HUF_A_CNT | (dsum & 0x7FFF) << HUF_A_SUM - one add per look-up moves
	// position, count and sum (a scan covers at most 64 codes: 7 bits of count, 15 bits of signed sum).  Long
	// codes / no code: as in mlut (the sign bit)
	alignas(16) uint32_t alut[1 << HUF_LUT_BITS];
	// ... and the first code alone, for the last steps in front of a limit: its bits | its delta << 8; 0: none
	alignas(16) uint16_t flut[1 << HUF_LUT_BITS];
};
constexpr uint32_t HUF_A_CNT = 9, HUF_A_SUM = 16;
constexpr uint32_t HUF_NEEDS_TRIE = 0x80000000u; // in DecodeArgs::huf_minlen: some code is beyond the second-level tables
constexpr uint32_t HUF_L2_NONE = HUF_L2_ENTRIES - 1; // second-level slot that says "no code" (tables that need no trie)
constexpr uint32_t HUF_MLONG = 1u << 31; // (the sign bit: one compare)
constexpr uint32_t HUF_M_MAXN = 8; // 8 deltas of -128 .. 127 fit the 11-bit sum
constexpr uint32_t HUF_LONG = 1u << 30;
constexpr uint32_t HUF_TWO = 1u << 29;

// parallel Huffman decode (press_huffman.hip): tiles of HUF_HT subsequences
constexpr int HUF_HT = 256;        // subsequences per tile (a lane each)
struct HufTRec {             // what k_huf_chain leaves per tile (32 bytes: k_huf_emit takes it in two 16-byte loads)
	uint32_t rsv0[2];
	uint32_t base;       // codes of the read in front of the tile
	uint32_t dbase;      // sum of their deltas, mod 2^16
	uint32_t rsv1[3];
	uint32_t fused;      // 1 = k_huf_emit writes the read's samples itself, 0 = its one-byte
	                     // values go to DecodeArgs::low and k_low_decode_chunked merges them
};
static_assert(sizeof(HufTRec) == 32, "HufTRec");
constexpr uint32_t HUF_FUSED = 0x80000000u; // in DecodeArgs::hread[2r + 1]: the read needs no k_low_decode_chunked

// ---- chunked (v2) svb kernels: a read is cut into chunks of CHUNK samples, one workgroup
// per chunk; chunks of a read are chained by a decoupled look-back over 8-byte granules.
constexpr uint32_t CHUNK = 32768;           // samples per chunk: 4 waves x 16 sub-tiles x 512
struct ChunkDesc {                          // written by k_chunk_prep, one per chunk (64 bytes)
	uint64_t sig_off;   // sample offset of the READ in sig
	uint64_t out_base;  // byte offset of the read's slot (encode) / stream (decode) in the arena
	uint32_t n;         // samples in the read
	uint32_t j;         // chunk index within the read
	uint32_t read;      // read index
	uint32_t cap_ok;    // slot large enough for the worst case of the format
	// decode only, filled by k_svb_keyscan / k_svb_keyprefix:
	uint64_t ebefore;   // exceptions (extra data bytes) in front of this chunk
	uint32_t ecnt[4];   // exceptions in each wave's quarter of the chunk
	uint16_t kmask[4];  // per wave: sub-tiles that are not plain (exception or ragged tail)
};
static_assert(sizeof(ChunkDesc) == 64, "ChunkDesc is one 64-byte line");
struct ChunkBits {                          // Huffman encode: code bits of a chunk (k_ex_scan_chunked<.., true>, k_ex_prefix)
	uint64_t before;    // bits of the read's payload in front of this chunk
	uint32_t q[4];      // bits of each wave's quarter
	uint32_t pad[2];
};
struct ChunkCtl {                           // device control block; every word on its own 128-B line
	uint32_t ticket;    // next chunk to hand out (atomic)
	uint32_t pad0[31];
	uint32_t nchunks;   // written by k_chunk_prep, read by every workgroup
	uint32_t pad1[31];
	uint32_t ticket2;   // spare second ticket
	uint32_t pad2[31];  // (stays zero: k_low_decode_chunked loads its zeros from here)
	uint32_t lists[32]; // Huffman decode: entries of the list repair round r leaves (the first list's: ticket2)
	uint32_t units;     // Huffman decode: next unit of work k_huf_emit hands out
	uint32_t pad4[31];
	uint32_t sunits;    // ... and k_huf_sync
	uint32_t pad5[31];
};

// Arguments common to every batch kernel.
struct BatchArgs {
	const int16_t *sig;       // samples of all reads
	const uint64_t *off;      // [nreads] read r starts at sig[off[r]] (multiple of 8 samples)
	const uint32_t *nsamp;    // [nreads] samples in read r
	uint8_t *out;             // compressed arena
	const uint64_t *out_off;  // [nreads+1] slot of read r
	uint64_t *out_len;        // [nreads] bytes produced / UINT64_MAX
	ReadMeta *meta;           // [nreads]
	uint32_t *ex_pos;         // [total samples] exception positions of read r at ex_pos[off[r]..]
	uint32_t *ex_val;         // [total samples] raw exception values
	const HuffDev *huff;
	uint32_t nreads;
	// chunked kernels
	ChunkDesc *chunks;        // [max_chunks]
	uint64_t *gran;           // [max_chunks] look-back granules (zeroed per launch)
	ChunkCtl *ctl;            // zeroed per launch
	uint32_t max_chunks;      // >= sum over reads of ceil(n / CHUNK)
	ChunkBits *cbits;         // [max_chunks] Huffman code-bit counts (NULL for the other methods)
	uint8_t *low_tmp;         // range-coder methods: the one-byte values of read r at low_tmp[off[r]..] (else NULL)
	uint32_t *first_chunk;    // [nreads] id of the first chunk of read r (exception-split encode)
	uint32_t *zhist;          // zstd compositions: [nreads][256] occurrences of the data bytes, counted where the svb
	                          // encoder has them in registers (zeroed by the caller; NULL for the other methods)
	uint32_t *zkcnt;          // ... and [max_chunks] at a read's FIRST chunk: the read's key bytes that are not zero (with zhist;
	                          // their positions and values are listed in ex_pos / ex_val of the read, in order)
};

struct HufTile {             // one tile of a read's Huffman payload (press_huffman.hip), 32 bytes
	uint64_t src;        // byte offset in the compressed arena of the tile's first payload byte
	uint64_t low;        // offset in DecodeArgs::low of the read's one-byte stream
	uint32_t nbits;      // payload bits from the tile's first bit to the end of the payload
	uint32_t t_last;     // tile index in the read | (no tile follows) << 31
	uint32_t read;
	uint32_t want;       // codes the read's header announces
};

struct DecodeArgs {
	const uint8_t *in;        // compressed arena
	const uint64_t *in_off;   // [nreads]
	const uint64_t *in_len;   // [nreads]
	int16_t *sig;             // decoded samples
	const uint64_t *off;      // [nreads] slot of read r starts at sig[off[r]] (multiple of 8 samples)
	const uint32_t *nsamp;    // [nreads] slot capacity in samples (= the sample count for svb streams)
	uint32_t *out_n;          // [nreads] samples decoded / UINT32_MAX
	ReadMeta *meta;
	uint32_t *ex_pos;
	uint32_t *ex_val;
	uint8_t *low;             // [total samples] Huffman- / range-decoded one-byte stream of read r at low[off[r]..]
	const HuffDev *huff;
	uint32_t nreads;
	// chunked kernels
	ChunkDesc *chunks;
	uint64_t *gran;           // [max_chunks] look-back granules of the sample-value chain
	ChunkCtl *ctl;
	uint32_t *first_chunk;    // [nreads] id of the first chunk of read r
	uint32_t max_chunks;
	// Huffman tiles (press_huffman.hip)
	HufTile *htiles;          // [max_htiles]
	HufTRec *htrec;           // [max_htiles]
	uint32_t *hrec;           // [max_htiles * HUF_HT] one record per subsequence: start | codes << 8 | sum of their deltas << 16
	uint32_t *hlist;          // [2 * hlist_cap] two lists of subsequences to decode again (k_huf_fix)
	uint8_t *hend;            // [max_htiles * HUF_HT] per subsequence: where the next one's first code starts (0 .. 30, 31: none)
	uint32_t *hmin;           // [nreads] first subsequence k_huf_fix's rounds left unsettled (0xFFFFFFFF: none)
	uint32_t *hread;          // [2 * nreads] first tile, number of tiles of read r
	uint2 *hwave;             // [max_htiles * 4] per wave of a tile: its codes, the sum of their deltas (mod 2^16 in the low half)
	unsigned long long *hbits; // [max_htiles * 4] per wave of a tile: the lanes whose guess k_huf_sync found wrong or could not check
	uint32_t max_htiles;
	uint32_t hlist_cap;
	uint32_t huf_minlen;      // shortest code of the table (selects the subsequence size on the host)
};

// zstd frames on the device (press_zstd.hip, zs_table.h): scratch of one batch
struct ZsRead {               // per read
	uint32_t nk, nd;      // key bytes / data bytes of the svb-zd stream
	uint32_t knz;         // key bytes that are not zero
	uint32_t mode;        // 0: RLE keys + Huffman data, 1: raw blocks, 2: failed
	uint32_t dbase;       // frame offset of the first data block
	uint32_t plen;        // bytes in front of the keys: the count / the ex-zd header and exception section
	uint32_t pad[2];
};
// ZsRead on the decode side: nd = content size of the frame, mode 0 ok / 2 malformed / 3 left to libzstd,
// pad[0] = its first block with sequences + 1 (0: none)
struct ZsCopy {               // content bytes [dst, dst + n) of ztmp: a copy of n bytes at src of the arena, or its byte n times
	uint64_t src, dst;
	uint32_t n, fill;
};
struct ZsHuf {                // Huffman-coded literals of one block
	uint64_t src, dst;    // arena offset of the streams (jump table first) / ztmp offset of the literals
	uint32_t cs, R;       // bytes of the streams, literals
	uint32_t four;        // 4 streams (else 1)
	uint32_t pad;
};
struct ZsUnit {               // up to 8 Huffman blocks of a read with one table: half a wave's work
	uint32_t read, tree, count, pad;
};
struct ZsLong {               // a block with four LONG streams (ZSTD_compress's 128-KiB blocks): a wave of its own, 16 lanes a stream
	ZsHuf h;
	uint32_t tree, read, pad[2];
};
struct ZsTree {
	uint8_t w[256];       // weights (RFC 8878 4.2.1.1)
	uint32_t tl, pad[3];
};
struct ZsSeq {               // one sequence: ll literals, then ml bytes from off bytes back
	uint32_t ll, ml, off;
};
struct ZsXBlk {              // a block with sequences, in the order of its frame
	uint64_t lit;        // ztmp offset of its literals
	uint64_t dst;        // ztmp offset of its content
	uint32_t seq0, nseq; // its sequences in dseq
	uint32_t tail;       // literals behind the last match
	uint32_t next;       // the frame's next such block + 1, 0: none
};
struct ZsDCtl { // every counter on a 4-KB page of its own: they are bumped by all waves of k_zs_walk with returning atomics,
                // which are carried out one after the other per address - and on one cache line, all of them together
	uint32_t ncopy, pad0[1023];
	uint32_t nunits, pad1[1023];
	uint32_t ntrees, pad2[1023]; // trees beyond a read's first (that one is slot r of the tree list)
	uint32_t nseq, pad3[1023];
	uint32_t nxblk, pad4[1023];
	uint32_t nhost, pad5[1023];
	uint32_t nlong, pad6[1023];
};
struct ZsBufs {
	uint8_t *ztmp;        // the svb-zd streams between the two stages: [u32 n][keys][data] of read r at zoff[r]
	uint64_t *zoff;       // [nreads + 1]
	uint64_t *zoff4;      // [nreads + 1] zoff + 4: where the svb kernels write / read
	uint64_t *zlen;       // [nreads] bytes of keys + data
	uint32_t *hist;       // [nreads][256]
	void *tab;            // zs::Table [nreads]
	uint32_t *first_blk;  // [nreads + 1] id of the first data block of read r
	uint32_t *blk_read;   // [max_blocks]
	uint4 *sbits;         // [max_blocks] code bits of the four streams of a block
	uint32_t *bpos;       // [max_blocks] frame offset of the block
	uint8_t *bflag;       // [max_blocks] 1: Huffman, 2: carries the tree, 4: last block of the frame
	uint32_t *kcnt;       // [max_chunks] at a read's first chunk: its key bytes that are not zero (BatchArgs::zkcnt)
	ZsRead *rd;           // [nreads]
	uint32_t *nblocks;    // [1]
	uint32_t max_blocks;
	// decode: what k_zs_walk finds in the frames
	ZsCopy *dcopy;        // [cap_copy] raw / RLE pieces
	ZsHuf *dhuf;          // [cap_units * 8] Huffman blocks, 8 slots per unit
	ZsUnit *dunit;        // [cap_units]
	ZsTree *dtree;        // [cap_trees]
	ZsLong *dlong;        // [cap_long] blocks whose streams are decoded in segments (k_zs_hdecode_long)
	ZsDCtl *dctl;
	uint32_t *zn;         // [nreads] sample count found in the stream
	ZsSeq *dseq;          // [cap_seq] sequences of the frames' blocks
	ZsXBlk *dxblk;        // [cap_xblk]
	uint64_t lit_base;    // ztmp offset of the literals space (read r's at lit_base + zoff[r])
	uint32_t cap_seq, cap_xblk;
	uint32_t cap_copy, cap_units, cap_trees, cap_long;
	uint32_t kdiv;        // samples per key byte of the inner stream: 4 (svb-zd), 8 (svb16-zd), 0: ex-zd (no keys)
};
void launch_zstd_encode(const BatchArgs &a, const ZsBufs &z, hipStream_t s); // press_zstd.hip
// decode in two steps: frames -> svb-zd streams in ztmp (reads the device leaves to libzstd
// are counted in dctl->nhost and patched in by the caller), then the svb-zd decode
void launch_zstd_decode_frames(const DecodeArgs &a, const ZsBufs &z, hipStream_t s);
void launch_zstd_decode_streams(const DecodeArgs &a, const ZsBufs &z, hipStream_t s);

// Optional timing of the dominant kernel of a batch call with HIP events recorded on the
// launch stream (bench.py's roofline figure): launchers call these around that kernel.
void ktime_begin(int which, hipStream_t s); // which: 0 = press, 1 = depress
void ktime_end(int which, hipStream_t s);
void ktime_mute(bool m); // a composite launcher times its own kernel instead of the inner one's

// launchers.  All asynchronous on `s`.
void launch_svb_encode_chunked(const BatchArgs &a, bool key2bit, bool zd, hipStream_t s, bool slow5 = false); // chunks + look-back
void launch_svb_decode_chunked(const DecodeArgs &a, bool key2bit, bool zd, hipStream_t s, bool slow5 = false);
void launch_ex_encode_chunked(const BatchArgs &a, int fmt, int ent, hipStream_t s); // ent: 0 plain, 1 Huffman, 2 / 3 / 4 range coder of order 0 / 1 / 1-0 mixing
void launch_ex_decode_chunked(const DecodeArgs &a, int fmt, int ent, hipStream_t s);
void launch_ex_parse_huff(const DecodeArgs &a, int fmt, int ent, hipStream_t s); // press_sections.hip
void launch_huff_decode(const DecodeArgs &a, uint32_t minlen, hipStream_t s);      // press_huffman.hip
void launch_ex_section(const BatchArgs &a, int fmt, int ent, hipStream_t s);        // press_sections.hip
void launch_rcs_encode(const BatchArgs &a, hipStream_t s);  // press_rc.hip
void launch_rcs_decode(const DecodeArgs &a, hipStream_t s);
void launch_rcc_encode(const BatchArgs &a, hipStream_t s);  // order 1 (rcc_vbe21_zd)
void launch_rcc_decode(const DecodeArgs &a, hipStream_t s);
void launch_rcm_encode(const BatchArgs &a, hipStream_t s);  // order 1-0 context mixing (rccm_vbbe21_zd)
void launch_rcm_decode(const DecodeArgs &a, hipStream_t s);

} // namespace ph
