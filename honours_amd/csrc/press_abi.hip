// press_abi.hip - host side of libpress_hip.so: the C ABI of include/press_hip.h.
//
// Mirrors the reference's per-method C interface (press/press.h) on top of the batch
// kernels in press_*.hip.  There is NO CPU implementation of any codec in this
// file: every X_press / X_depress runs the HIP kernels and fails (-1 / *nout = 0)
// when the device is unavailable.  The only third-party stage is libzstd for the
// zstd_* compositions, which the reference itself delegates to libzstd (press.c:1464).

#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <thread>
#include <algorithm>
#include <vector>

#include "../../include/press_hip.h"
#include "press_internal.h"
#include "zs_table.h"

using namespace ph;

// ------------------------------------------------------------------ errors

static thread_local char g_err[256] = "";

static int fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return code;
}

#define HIPCHK(call)                                                                        \
	do {                                                                                \
		hipError_t e_ = (call);                                                     \
		if (e_ != hipSuccess)                                                       \
			return fail(PRESS_HIP_EHIP, "%s: %s", #call, hipGetErrorString(e_)); \
	} while (0)

extern "C" const char *press_hip_last_error(void) { return g_err; }

// ------------------------------------------------------------------ context

namespace {

struct DevBuf;
DevBuf *g_bufs = nullptr; // every DevBuf links itself in here: press_hip_shutdown() cannot forget one

struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	DevBuf *next;
	DevBuf() : next(g_bufs) { g_bufs = this; }
	DevBuf(const DevBuf &) = delete;
	DevBuf &operator=(const DevBuf &) = delete;
	// grow-only; contents are not preserved
	int reserve(size_t n)
	{
		if (n <= cap)
			return 0;
		if (p)
			(void) hipFree(p);
		p = nullptr;
		cap = 0;
		size_t want = n + n / 8 + 4096;
		hipError_t e = hipMalloc(&p, want);
		if (e != hipSuccess)
			return fail(PRESS_HIP_EHIP, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
		cap = want;
		return 0;
	}
	void release()
	{
		if (p)
			(void) hipFree(p);
		p = nullptr;
		cap = 0;
	}
};

struct Ctx {
	bool ready = false;
	int device = -1;
	hipStream_t own = nullptr;
	hipStream_t user = nullptr;
	bool use_user = false;
	// scratch shared by both modes
	DevBuf meta, ex_pos, ex_val, low, huff, chunks, gran, ctl, first_chunk, htiles, htrec, hrec, hlist, hread, hwave, hbits, hend, hmin, cbits;
	DevBuf ztmp, zoff, zoff4, zlen, zhist, ztab, zfirst, zblk, zsbits, zbpos, zbflag, zkcnt, zrd, znb, zn, zdcopy, zdhuf, zdunit, zdtree, zdlong, zdctl, zdseq, zdxblk; // zstd frames
	// staging for host-pointer calls
	DevBuf sig, off, nsamp, arena, arena_off, lens, lens2, outn, dense, dense_off;
	uint64_t zs_total = 0; // total_samples of the batch in flight (sizes of the zstd scratch)
	uint32_t zs_nhost = 0; // frames the last zstd depress batch left to libzstd
	uint32_t *zs_pin = nullptr; // page-locked: that count comes back while the device goes on with the batch
	hipEvent_t zs_ev = nullptr;
	// static Huffman table currently on the device
	bool have_table = false;
	bool table_trie = false; // some code of the table is beyond the second-level tables (HUF_NEEDS_TRIE)
	uint32_t tlen[256];
	uint64_t tbits[256];

	hipStream_t stream() const { return use_user ? user : own; }
};

Ctx g;
// One context per process (one process per GPU): calls from several host threads are serialised.
std::recursive_mutex g_mu;
#define API_LOCK std::lock_guard<std::recursive_mutex> api_lock_(g_mu)
void staging_release(); // page-locked staging buffers of the host-pointer calls (below)

// Every entry point that touches the device comes through here: HIP's current device is per
// host thread, so it is selected again on every call (cheap), not only by the first caller.
int ctx_init()
{
	if (g.ready) {
		HIPCHK(hipSetDevice(g.device));
		return 0;
	}
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev == 0)
		return fail(PRESS_HIP_EHIP, "no HIP device: %s", hipGetErrorString(e));
	if (g.device < 0)
		HIPCHK(hipGetDevice(&g.device));
	HIPCHK(hipSetDevice(g.device));
	HIPCHK(hipStreamCreateWithFlags(&g.own, hipStreamNonBlocking));
	g.ready = true;
	return 0;
}

// ---- zstd, loaded lazily (third party; the reference links -lzstd, press/Makefile:3) ----
struct Zstd {
	bool tried = false;
	size_t (*compress)(void *, size_t, const void *, size_t, int) = nullptr;
	size_t (*decompress)(void *, size_t, const void *, size_t) = nullptr;
	size_t (*bound)(size_t) = nullptr;
	unsigned (*is_error)(size_t) = nullptr;
} zstd_fn;

bool zstd_open()
{
	if (zstd_fn.tried)
		return zstd_fn.compress != nullptr;
	zstd_fn.tried = true;
	// Bind to the libzstd the process already holds, if any: a SECOND copy of the library (another version opened by
	// absolute path) resolves its internal calls through the first one's symbols when that one is in the global
	// scope, and frees what the other allocated - glibc aborts with "free(): invalid pointer" / "munmap_chunk()"
	// (round 2: rocprofv3's tool library links the system's 1.4.8, this opened conda's 1.4.9; RTLD_LOCAL does not
	// prevent it).  So: what is loaded, then the soname, and an absolute path only last and bound to itself.
	struct { const char *name; int flags; } names[] = {
		{ "libzstd.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD },
		{ "libzstd.so.1", RTLD_NOW | RTLD_LOCAL },
		{ "libzstd.so", RTLD_NOW | RTLD_LOCAL },
		{ "/opt/conda/lib/libzstd.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND },
		{ nullptr, 0 } };
	for (int i = 0; names[i].name; i++) {
		void *h = dlopen(names[i].name, names[i].flags);
		if (!h)
			continue;
		zstd_fn.compress = (decltype(zstd_fn.compress)) dlsym(h, "ZSTD_compress");
		zstd_fn.decompress = (decltype(zstd_fn.decompress)) dlsym(h, "ZSTD_decompress");
		zstd_fn.bound = (decltype(zstd_fn.bound)) dlsym(h, "ZSTD_compressBound");
		zstd_fn.is_error = (decltype(zstd_fn.is_error)) dlsym(h, "ZSTD_isError");
		if (zstd_fn.compress && zstd_fn.decompress && zstd_fn.bound && zstd_fn.is_error)
			return true;
		zstd_fn.compress = nullptr;
	}
	return false;
}

uint64_t zstd_bound_(uint64_t n)
{
	if (zstd_open())
		return zstd_fn.bound(n);
	// zstd.h ZSTD_COMPRESSBOUND
	return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0);
}

// ---- bounds: the reference's formulas ----
uint32_t svb16_keylen(uint32_t n) { return (n >> 3) + (((n & 7) + 7) >> 3); }
uint64_t bound_svb32(uint32_t n) { return (uint64_t) (n + 3) / 4 + (uint64_t) n * 4 + 16; }    // streamvbyte.h:35
uint64_t bound_svb16(uint32_t n) { return (uint64_t) svb16_keylen(n) + (uint64_t) n * 4 + 16; } // streamvbyte.h:42
uint64_t bound_vb1e2(uint32_t m) { return (uint64_t) (1 + m * 0.2 * 6 + m * 0.8); }            // press.c:2575
uint64_t bound_vbzd(uint32_t n) { return 2 + bound_vb1e2(n - 1); }                             // press.c:3411

bool is_shuff(int m) { return m >= PRESS_HIP_SHUFF_VBE21_ZD && m <= PRESS_HIP_SHUFF_VBSSE21_ZD; }
bool is_svb(int m) { return m == PRESS_HIP_SVB12 || m == PRESS_HIP_SVB12_ZD || m == PRESS_HIP_SVB_ZD || m == PRESS_HIP_SLOW5_SVB_ZD; }
bool is_zstd(int m)
{
	return m == PRESS_HIP_ZSTD_SVB_ZD || m == PRESS_HIP_ZSTD_SVB12_ZD || m == PRESS_HIP_ZSTD_HASGAM_ZDQ;
}
bool is_rc(int m) { return m == PRESS_HIP_RC_VBE21_ZD || m == PRESS_HIP_RCC_VBE21_ZD || m == PRESS_HIP_RCCM_VBBE21_ZD; }
bool is_zs(int m) { return m == PRESS_HIP_ZSTD_SVB_ZD || m == PRESS_HIP_ZSTD_SVB12_ZD || m == PRESS_HIP_ZSTD_HASGAM_ZDQ; } // zstd frames made on the device (batch API)
int zs_inner(int m) { return m == PRESS_HIP_ZSTD_SVB_ZD ? PRESS_HIP_SVB_ZD : m == PRESS_HIP_ZSTD_SVB12_ZD ? PRESS_HIP_SVB12_ZD : PRESS_HIP_HASGAM_ZDQ; }
uint32_t zs_kdiv(int m) { return m == PRESS_HIP_ZSTD_SVB_ZD ? 4 : m == PRESS_HIP_ZSTD_SVB12_ZD ? 8 : 0; }
// bytes per sample the inner stream can take at most (device: zs_content_max)
uint64_t zs_tmp_bytes(int m, uint64_t total_samples, uint32_t nreads)
{
	return (zs_kdiv(m) ? total_samples * 9 / 4 : total_samples * 9) + ((uint64_t) nreads + 1) * 128 + 64;
}
bool is_ex(int m) { return (m >= PRESS_HIP_VBE21_ZD && m <= PRESS_HIP_HASGAM_ZDQ) || is_rc(m); }
int entropy_of(int m) { return is_shuff(m) ? 1 : m == PRESS_HIP_RC_VBE21_ZD ? 2 : m == PRESS_HIP_RCC_VBE21_ZD ? 3 : m == PRESS_HIP_RCCM_VBBE21_ZD ? 4 : 0; }

int exfmt_of(int m)
{
	switch (m) {
	case PRESS_HIP_VBE21_ZD: case PRESS_HIP_SHUFF_VBE21_ZD: case PRESS_HIP_RC_VBE21_ZD: case PRESS_HIP_RCC_VBE21_ZD: return EXF_VBE21;
	case PRESS_HIP_VBBE21_ZD: case PRESS_HIP_SHUFF_VBBE21_ZD: case PRESS_HIP_RCCM_VBBE21_ZD: return EXF_VBBE21;
	case PRESS_HIP_VBSBE21_ZD: case PRESS_HIP_SHUFF_VBSBE21_ZD: return EXF_VBSBE21;
	case PRESS_HIP_VBSSE21_ZD: case PRESS_HIP_SHUFF_VBSSE21_ZD: return EXF_VBSSE21;
	default:                                                    return EXF_EXZD;
	}
}

// ---- static Huffman table -> device form ----
// the delta a one-byte value stands for (zig-zag undone): 0, -1, 1, -2, ... -128
static inline int32_t unzz8(uint32_t z) { return (int32_t) (z >> 1) ^ -(int32_t) (z & 1u); }

int upload_table(const uint32_t len[256], const uint64_t bits[256])
{
	if (g.have_table && !memcmp(len, g.tlen, sizeof g.tlen) && !memcmp(bits, g.tbits, sizeof g.tbits))
		return 0;
	std::vector<HuffDev> hv(1);
	HuffDev &h = hv[0];
	memset(&h, 0xFF, sizeof h); // lut = 0xFFFF, child/leaf = -1
	int nnodes = 1, ncoded = 0;
	bool needs_trie = false;
	for (int s = 0; s < 256; s++) {
		const uint32_t l = len[s];
		if (l > 24) // huffman.c takes codes of up to 255 bits; the device tables stop at 24 (press_hip.h)
			return fail(PRESS_HIP_EARG, "Huffman table: symbol %d has a code of %u bits (at most 24)", s, l);
		if (l == 0) { // a table file may list fewer than 256 symbols (huffman.c:549): such a symbol
			h.enc[s] = 0; // has no code, and a read fails only if the symbol occurs (k_ex_scan_chunked)
			continue;
		}
		ncoded++;
		h.enc[s] = (uint32_t) (bits[s] & 0xFFFFFFu) | (l << 24);
		int p = 0;
		for (uint32_t k = 0; k < l; k++) {
			const int b = (int) ((bits[s] >> k) & 1);
			if (h.leaf[p] >= 0)
				return fail(PRESS_HIP_EARG, "Huffman table: not a prefix code");
			if (h.child[p][b] < 0) {
				if (nnodes >= 1024)
					return fail(PRESS_HIP_EARG, "Huffman table: too many nodes");
				h.child[p][b] = (int16_t) nnodes++;
			}
			p = h.child[p][b];
		}
		if (h.child[p][0] >= 0 || h.child[p][1] >= 0)
			return fail(PRESS_HIP_EARG, "Huffman table: not a prefix code");
		h.leaf[p] = (int16_t) s;
		if (l <= (uint32_t) HUF_LUT_BITS) {
			const uint32_t step = 1u << l;
			for (uint32_t i = (uint32_t) bits[s] & (step - 1); i < (1u << HUF_LUT_BITS); i += step)
				h.lut[i] = (uint16_t) (s | (l << 8));
		}
	}
	// second-level tables for the long codes, grouped by their first HUF_LUT_BITS bits
	{
		int nid = 0;
		uint32_t used = 0;
		int id_of[1 << HUF_LUT_BITS];
		for (int i = 0; i < (1 << HUF_LUT_BITS); i++)
			id_of[i] = -1;
		uint32_t depth[256] = { 0 };
		int prefix_of[256];
		for (int s = 0; s < 256; s++) {
			if (len[s] <= (uint32_t) HUF_LUT_BITS)
				continue;
			const int pfx = (int) (bits[s] & ((1u << HUF_LUT_BITS) - 1));
			if (id_of[pfx] < 0 && nid < HUF_L2_IDS) {
				prefix_of[nid] = pfx;
				id_of[pfx] = nid++;
			}
			if (id_of[pfx] >= 0 && len[s] - HUF_LUT_BITS > depth[id_of[pfx]])
				depth[id_of[pfx]] = len[s] - HUF_LUT_BITS;
		}
		// deepest first: every table then starts at a multiple of its own size (the first-level
		// entries of the parallel decoder keep offset / 2 in 11 bits)
		for (uint32_t d = 32; d >= 1; d--) {
			for (int id = 0; id < nid; id++) {
				if (depth[id] != d)
					continue;
				if (d > 12 || used + (1u << d) > (uint32_t) HUF_L2_ENTRIES)
					continue; // does not fit: this prefix keeps 0xFFFF and walks the trie
				h.l2off[id] = (uint16_t) used;
				h.l2bits[id] = (uint8_t) d;
				h.lut[prefix_of[id]] = (uint16_t) (0x8000u | (uint32_t) id);
				used += 1u << d;
			}
		}
		for (int s = 0; s < 256; s++) {
			if (len[s] <= (uint32_t) HUF_LUT_BITS)
				continue;
			const int pfx = (int) (bits[s] & ((1u << HUF_LUT_BITS) - 1));
			const int id = id_of[pfx];
			if (id < 0 || h.lut[pfx] != (uint16_t) (0x8000u | (uint32_t) id))
				continue;
			const uint32_t rest = (uint32_t) (bits[s] >> HUF_LUT_BITS), rl = len[s] - HUF_LUT_BITS;
			for (uint32_t i = rest; i < (1u << depth[id]); i += 1u << rl)
				h.lut2[h.l2off[id] + i] = (uint16_t) (s | (len[s] << 8));
		}
		// does any code need the trie (a long code whose prefix got no second-level table)?  If none does, the
		// kernels run without it (their TRIE = false variants) and a prefix that is no code at all points at a
		// second-level slot that says so
		needs_trie = used > HUF_L2_NONE; // (the slot must be free)
		for (int s = 0; s < 256; s++) {
			if (len[s] <= (uint32_t) HUF_LUT_BITS)
				continue;
			const int pfx = (int) (bits[s] & ((1u << HUF_LUT_BITS) - 1));
			const int id = id_of[pfx];
			if (id < 0 || h.lut[pfx] != (uint16_t) (0x8000u | (uint32_t) id))
				needs_trie = true;
		}
		for (int i = 0; i < HUF_L2_ENTRIES; i++) {
			const uint32_t e2 = h.lut2[i];
			h.l2ld[i] = e2 == 0xFFFFu ? (uint16_t) 0xFFFFu : (uint16_t) ((e2 >> 8) | ((unzz8(e2 & 0xFFu) & 0xFFu) << 8));
		}
	}
	// two-symbol table
	for (uint32_t i = 0; i < (1u << HUF_LUT_BITS); i++) {
		const uint16_t e1 = h.lut[i];
		if (e1 == 0xFFFFu) {
			h.lut32[i] = needs_trie ? 0xFFFFFFFFu : (HUF_LONG | HUF_L2_NONE);
		} else if (e1 & 0x8000u) { // a long code's prefix: where its second-level table sits
			const uint32_t id = e1 & 0xFFu;
			h.lut32[i] = HUF_LONG | ((uint32_t) h.l2bits[id] << 12) | h.l2off[id];
		} else {
			// (the parallel decoder stages the delta a symbol stands for, not the symbol)
			const uint32_t s1 = (uint32_t) unzz8(e1 & 0xFFu) & 0xFFu, l1 = e1 >> 8;
			uint32_t v = s1 | (l1 << 8) | (l1 << 24);
			const uint32_t rest = (uint32_t) HUF_LUT_BITS - l1;
			const uint16_t e2 = h.lut[i >> l1]; // the upper bits are zeros, not stream bits:
			if (e2 != 0xFFFFu && !(e2 & 0x8000u) && (uint32_t) (e2 >> 8) <= rest) // only a code that fits counts
				v = s1 | ((l1 + (e2 >> 8)) << 8) | (((uint32_t) unzz8(e2 & 0xFFu) & 0xFFu) << 16) | (l1 << 24) | HUF_TWO;
			h.lut32[i] = v;
		}
	}
	// every whole code that fits in the first HUF_LUT_BITS bits, lengths and deltas (k_huf_sync)
	for (uint32_t i = 0; i < (1u << HUF_LUT_BITS); i++) {
		uint32_t pos = 0, n = 0, len1 = 0;
		int32_t d1 = 0, dsum = 0;
		for (;;) {
			const uint16_t e1 = h.lut[i >> pos]; // the upper bits are zeros, not stream bits:
			if (e1 == 0xFFFFu || (e1 & 0x8000u) || pos + (uint32_t) (e1 >> 8) > (uint32_t) HUF_LUT_BITS)
				break; // only a code that fits counts
			const int32_t d = (int32_t) unzz8(e1 & 0xFFu);
			if (!n) {
				len1 = e1 >> 8;
				d1 = d;
			}
			dsum += d;
			pos += e1 >> 8;
			n++;
			if (pos == (uint32_t) HUF_LUT_BITS || n == HUF_M_MAXN)
				break;
		}
		const uint16_t e0 = h.lut[i];
		if (n)
			h.mlut[i] = pos | (n << 4) | (len1 << 8) | (((uint32_t) d1 & 0xFFu) << 12) | (((uint32_t) dsum & 0x7FFu) << 20);
		else if (e0 != 0xFFFFu && (e0 & 0x8000u)) // long code: second-level table of lengths and deltas
			h.mlut[i] = HUF_MLONG | (uint32_t) h.l2off[e0 & 0xFFu] | ((uint32_t) h.l2bits[e0 & 0xFFu] << 12);
		else
			h.mlut[i] = needs_trie ? 0xFFFFFFFFu : (HUF_MLONG | HUF_L2_NONE);
		// the accumulator form (k_huf_sync's lean loop) and the first code alone (its careful loop)
		h.alut[i] = n ? (pos | (n << HUF_A_CNT) | (((uint32_t) dsum & 0x7FFFu) << HUF_A_SUM)) : h.mlut[i];
		h.flut[i] = n ? (uint16_t) (len1 | (((uint32_t) d1 & 0xFFu) << 8)) : (uint16_t) 0;
	}
	if (!ncoded)
		return fail(PRESS_HIP_EARG, "Huffman table: no symbol has a code");
	h.minlen = 64;
	h.maxlen = 0;
	for (int s = 0; s < 256; s++) {
		if (!len[s])
			continue;
		h.minlen = len[s] < h.minlen ? len[s] : h.minlen;
		h.maxlen = len[s] > h.maxlen ? len[s] : h.maxlen;
	}
	if (g.huff.reserve(sizeof(HuffDev)))
		return PRESS_HIP_EHIP;
	HIPCHK(hipMemcpy(g.huff.p, &h, sizeof h, hipMemcpyHostToDevice));
	memcpy(g.tlen, len, sizeof g.tlen);
	memcpy(g.tbits, bits, sizeof g.tbits);
	g.table_trie = needs_trie;
	g.have_table = true;
	return 0;
}

int parse_table_file(FILE *fp, uint32_t len[256], uint64_t bits[256]);

uint32_t max_chunks_of(uint64_t total_samples, uint32_t nreads)
{
	return (uint32_t) (total_samples / CHUNK + nreads + 1);
}

// Huffman tiles of a batch (press_huffman.hip, k_huff_tiles): a read of n samples has at most
// n - 1 codes of at most maxlen bits, cut into tiles of HUF_HT subsequences
uint32_t max_htiles_of(uint64_t total_samples, uint32_t nreads)
{
	uint32_t minlen = 64, maxlen = 1;
	for (int s = 0; s < 256; s++) {
		if (!g.tlen[s])
			continue;
		minlen = g.tlen[s] < minlen ? g.tlen[s] : minlen;
		maxlen = g.tlen[s] > maxlen ? g.tlen[s] : maxlen;
	}
	const uint64_t tb = (uint64_t) HUF_HT * (minlen >= 4 ? 256u : minlen >= 2 ? 128u : 64u);
	const uint64_t mt = total_samples * maxlen / tb + nreads + 1;
	return mt > 0xFFFFFFull ? 0xFFFFFFu : (uint32_t) mt;
}

// broken links a repair round can list (3 % of the subsequences on signal data; what does not fit
// is left to the serial pass)
size_t hlist_cap_of(size_t max_htiles) { return max_htiles * HUF_HT + (1u << 20); } // (a table that never synchronises lists every subsequence; k_huf_sync's workgroups take slots 1024 at a time)
uint32_t table_minlen()
{
	uint32_t minlen = 64;
	for (int s = 0; s < 256; s++)
		if (g.tlen[s] && g.tlen[s] < minlen)
			minlen = g.tlen[s];
	return minlen;
}

uint32_t max_zblocks_of(uint64_t total_samples, uint32_t nreads)
{
	return (uint32_t) (total_samples * 2 / zs::BLOCK_LITS + nreads + 1);
}

void zs_bufs(ZsBufs &z, uint64_t total_samples, uint32_t nreads, int method = PRESS_HIP_ZSTD_SVB_ZD)
{
	z.kdiv = zs_kdiv(method);
	z.ztmp = (uint8_t *) g.ztmp.p;
	z.zoff = (uint64_t *) g.zoff.p;
	z.zoff4 = (uint64_t *) g.zoff4.p;
	z.zlen = (uint64_t *) g.zlen.p;
	z.hist = (uint32_t *) g.zhist.p;
	z.tab = g.ztab.p;
	z.first_blk = (uint32_t *) g.zfirst.p;
	z.blk_read = (uint32_t *) g.zblk.p;
	z.sbits = (uint4 *) g.zsbits.p;
	z.bpos = (uint32_t *) g.zbpos.p;
	z.bflag = (uint8_t *) g.zbflag.p;
	z.kcnt = (uint32_t *) g.zkcnt.p;
	z.rd = (ZsRead *) g.zrd.p;
	z.nblocks = (uint32_t *) g.znb.p;
	z.max_blocks = max_zblocks_of(total_samples, nreads);
	z.dcopy = (ZsCopy *) g.zdcopy.p;
	z.dhuf = (ZsHuf *) g.zdhuf.p;
	z.dunit = (ZsUnit *) g.zdunit.p;
	z.dtree = (ZsTree *) g.zdtree.p;
	z.dlong = (ZsLong *) g.zdlong.p;
	z.dctl = (ZsDCtl *) g.zdctl.p;
	z.zn = (uint32_t *) g.zn.p;
	// what a batch of this library's frames needs, with room for others; a frame that does not
	// fit (tiny blocks, thousands of trees) goes to libzstd on the host
	z.cap_copy = z.max_blocks + (uint32_t) (total_samples / 256) + 16 * nreads + 64;
	z.cap_units = z.max_blocks / 8 + 2 * nreads + 64;
	z.cap_trees = 4 * nreads + 64;
	z.cap_long = (uint32_t) (total_samples / 8192) + nreads + 64; // (blocks of at least 32 KiB of literals)
	// frames with sequences (libzstd's own): their literals in the second half of ztmp
	z.dseq = (ZsSeq *) g.zdseq.p;
	z.dxblk = (ZsXBlk *) g.zdxblk.p;
	z.lit_base = zs_tmp_bytes(method, total_samples, nreads);
	// (level 1 on signal data: a handful per block; higher levels: one per ~20 content bytes)
	const uint64_t cs = total_samples / 4 + 64ull * nreads + 1024;
	z.cap_seq = cs > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t) cs;
	z.cap_xblk = z.max_blocks / 4 + 4 * nreads + 64;
}

int reserve_scratch(int method, uint64_t total_samples, uint32_t nreads, bool decode)
{
	if (g.meta.reserve(((size_t) nreads + 1) * sizeof(ReadMeta)))
		return PRESS_HIP_EHIP;
	if (is_zs(method)) { // whatever the inner stream's kernels need
		const int rc = reserve_scratch(zs_inner(method), total_samples, nreads, decode);
		if (rc)
			return rc;
	}
	if (is_zs(method) && decode) {
		ZsBufs z;
		zs_bufs(z, total_samples, nreads, method);
		const size_t nr = (size_t) nreads + 1;
		if (g.ztmp.reserve(2 * zs_tmp_bytes(method, total_samples, nreads)) || g.zoff.reserve(nr * 8) || g.zoff4.reserve(nr * 8) ||
		    g.zlen.reserve(nr * 8) || g.zrd.reserve(nr * sizeof(ZsRead)) || g.zn.reserve(nr * 4) ||
		    g.zdseq.reserve((size_t) z.cap_seq * sizeof(ZsSeq)) || g.zdxblk.reserve((size_t) z.cap_xblk * sizeof(ZsXBlk)) ||
		    g.zdcopy.reserve((size_t) z.cap_copy * sizeof(ZsCopy)) || g.zdhuf.reserve((size_t) z.cap_units * 8 * sizeof(ZsHuf)) ||
		    g.zdunit.reserve((size_t) z.cap_units * sizeof(ZsUnit)) || g.zdtree.reserve((size_t) z.cap_trees * sizeof(ZsTree)) || g.zdlong.reserve((size_t) z.cap_long * sizeof(ZsLong)) ||
		    g.zdctl.reserve(sizeof(ZsDCtl)))
			return PRESS_HIP_EHIP;
	}
	if (is_zs(method) && !decode) {
		const size_t mc = max_chunks_of(total_samples, nreads), mb = max_zblocks_of(total_samples, nreads);
		const size_t nr = (size_t) nreads + 1;
		if (g.ztmp.reserve(zs_tmp_bytes(method, total_samples, nreads)) || g.zoff.reserve(nr * 8) || g.zoff4.reserve(nr * 8) ||
		    g.zlen.reserve(nr * 8) || g.zhist.reserve(nr * 1024) || g.ztab.reserve(nr * sizeof(zs::Table)) ||
		    g.zfirst.reserve(nr * 4) || g.zblk.reserve(mb * 4) || g.zsbits.reserve(mb * 16) || g.zbpos.reserve(mb * 4) ||
		    g.zbflag.reserve(mb) || g.zkcnt.reserve(mc * 4) ||
		    g.zrd.reserve(nr * sizeof(ZsRead)) || g.znb.reserve(64) ||
		    g.ex_pos.reserve((total_samples + 64) * 4) || g.ex_val.reserve((total_samples + 64) * 4))
			return PRESS_HIP_EHIP;
	}
	if (is_svb(method) || is_ex(method)) {
		const size_t mc = max_chunks_of(total_samples, nreads);
		if (g.chunks.reserve(mc * sizeof(ChunkDesc)) || g.gran.reserve(2 * mc * sizeof(uint64_t)) ||
		    g.ctl.reserve(2 * sizeof(ChunkCtl)) || g.first_chunk.reserve(((size_t) nreads + 1) * 4))
			return PRESS_HIP_EHIP;
	}
	if (is_ex(method)) {
		if (g.ex_pos.reserve((total_samples + 64) * 4) || g.ex_val.reserve((total_samples + 64) * 4))
			return PRESS_HIP_EHIP;
		if (!decode && is_shuff(method)) {
			const size_t mc = max_chunks_of(total_samples, nreads);
			if (g.cbits.reserve(mc * sizeof(ChunkBits)))
				return PRESS_HIP_EHIP;
		}
		if (is_rc(method) && g.low.reserve(total_samples + 64)) // the one-byte values between the two stages
			return PRESS_HIP_EHIP;
		if (decode && is_shuff(method)) {
			const size_t mt = max_htiles_of(total_samples, nreads);
			if (g.low.reserve(total_samples + 64) || g.htiles.reserve(mt * sizeof(HufTile)) ||
			    g.htrec.reserve(mt * sizeof(HufTRec)) || g.hrec.reserve(mt * HUF_HT * 4) ||
			    g.hlist.reserve(hlist_cap_of(mt) * 8) || g.hread.reserve(((size_t) nreads + 1) * 8) ||
			    g.hwave.reserve(mt * (HUF_HT / 64) * 8) || g.hbits.reserve(mt * (HUF_HT / 64) * 8) ||
			    g.hend.reserve(mt * HUF_HT + 64) ||
			    g.hmin.reserve(((size_t) nreads + 1) * 4))
				return PRESS_HIP_EHIP;
		}
	}
	return 0;
}

} // namespace

// ------------------------------------------------------------------ dominant-kernel timing

namespace {
constexpr int KT_RING = 128;
struct KTime {
	bool on = false;
	hipEvent_t ev[2][KT_RING][2];
	bool made = false;
	int n[2] = { 0, 0 };
} kt;
} // namespace

namespace ph {
static bool kt_muted = false;
void ktime_mute(bool m) { kt_muted = m; }
void ktime_begin(int which, hipStream_t s)
{
	if (kt.on && !kt_muted && kt.n[which] < KT_RING)
		(void) hipEventRecord(kt.ev[which][kt.n[which]][0], s);
}
void ktime_end(int which, hipStream_t s)
{
	if (kt.on && !kt_muted && kt.n[which] < KT_RING) {
		(void) hipEventRecord(kt.ev[which][kt.n[which]][1], s);
		kt.n[which]++;
	}
}
} // namespace ph

extern "C" int press_hip_kernel_timing(int enable)
{
	API_LOCK;
	int rc = ctx_init();
	if (rc)
		return rc;
	if (enable && !kt.made) {
		for (int w = 0; w < 2; w++)
			for (int i = 0; i < KT_RING; i++)
				for (int e = 0; e < 2; e++)
					HIPCHK(hipEventCreate(&kt.ev[w][i][e]));
		kt.made = true;
	}
	kt.on = enable != 0;
	kt.n[0] = kt.n[1] = 0;
	return 0;
}

extern "C" int press_hip_kernel_times(int which, float *ms, int max)
{
	API_LOCK;
	if (which < 0 || which > 1 || !kt.made)
		return 0;
	int n = kt.n[which] < max ? kt.n[which] : max;
	for (int i = 0; i < n; i++) {
		if (hipEventSynchronize(kt.ev[which][i][1]) != hipSuccess ||
		    hipEventElapsedTime(&ms[i], kt.ev[which][i][0], kt.ev[which][i][1]) != hipSuccess)
			return i;
	}
	return n;
}

// ------------------------------------------------------------------ library control

extern "C" int press_hip_set_device(int device)
{
	API_LOCK;
	if (g.ready && device != g.device)
		press_hip_shutdown();
	g.device = device;
	return ctx_init();
}

extern "C" int press_hip_set_stream(void *stream)
{
	API_LOCK;
	int rc = ctx_init();
	if (rc)
		return rc;
	g.user = (hipStream_t) stream; // nullptr = the default (null) stream
	g.use_user = true;
	return 0;
}

extern "C" int press_hip_reset_stream(void)
{
	API_LOCK;
	int rc = ctx_init();
	if (rc)
		return rc;
	g.user = nullptr;
	g.use_user = false;
	return 0;
}

extern "C" void *press_hip_get_stream(void)
{
	API_LOCK;
	if (ctx_init())
		return nullptr;
	return (void *) g.stream();
}

extern "C" int press_hip_synchronize(void)
{
	API_LOCK;
	int rc = ctx_init();
	if (rc)
		return rc;
	HIPCHK(hipStreamSynchronize(g.stream()));
	return 0;
}

extern "C" void press_hip_shutdown(void)
{
	API_LOCK;
	if (!g.ready)
		return;
	(void) hipSetDevice(g.device);
	(void) hipStreamSynchronize(g.own);
	for (DevBuf *b = g_bufs; b; b = b->next)
		b->release();
	staging_release();
	if (g.zs_pin)
		(void) hipHostFree(g.zs_pin);
	if (g.zs_ev)
		(void) hipEventDestroy(g.zs_ev);
	g.zs_pin = nullptr;
	g.zs_ev = nullptr;
	g.zs_total = 0;
	(void) hipStreamDestroy(g.own);
	g.own = nullptr;
	g.user = nullptr;
	g.use_user = false;
	g.have_table = false;
	g.ready = false;
}

extern "C" uint32_t press_hip_zstd_host_frames(void)
{
	API_LOCK;
	return g.zs_nhost;
}

extern "C" uint32_t press_hip_scratch_buffers(uint64_t *bytes)
{
	API_LOCK;
	uint32_t nb = 0;
	uint64_t b = 0;
	for (DevBuf *d = g_bufs; d; d = d->next) {
		nb++;
		b += d->cap;
	}
	if (bytes)
		*bytes = b;
	return nb;
}

extern "C" int press_hip_set_table(const uint32_t len[256], const uint64_t bits[256])
{
	API_LOCK;
	int rc = ctx_init();
	if (rc)
		return rc;
	return upload_table(len, bits);
}

extern "C" int press_hip_load_table_file(const char *path)
{
	uint32_t len[256];
	uint64_t bits[256];
	FILE *fp = fopen(path, "rb");
	if (!fp)
		return fail(PRESS_HIP_EARG, "cannot open %s", path);
	int rc = parse_table_file(fp, len, bits);
	fclose(fp);
	if (rc)
		return rc;
	return press_hip_set_table(len, bits);
}

extern "C" uint64_t press_hip_bound(int method, uint32_t n)
{
	switch (method) {
	case PRESS_HIP_SVB12:
	case PRESS_HIP_SVB12_ZD:        return bound_svb16(n);                          // press.c:1568,1678
	case PRESS_HIP_SVB_ZD:          return bound_svb32(n);                          // press.c:1585
	case PRESS_HIP_ZSTD_SVB_ZD:     return zstd_bound_(4 + bound_svb32(n));         // press.c:1860
	case PRESS_HIP_ZSTD_SVB12_ZD:   return zstd_bound_(4 + bound_svb16(n));         // press.c:2020
	case PRESS_HIP_HASGAM_ZDQ:      return bound_svb32((uint32_t) bound_vbzd(n));   // press.c:8461
	case PRESS_HIP_ZSTD_HASGAM_ZDQ: return zstd_bound_(bound_svb32((uint32_t) bound_vbzd(n)));
	case PRESS_HIP_SLOW5_SVB_ZD:    return (uint64_t) (n + 3) / 4 + (uint64_t) n * 4 + 4; // slow5_press.c:1037 (streamvbyte.h:31) + u32 count
	default:
		if (is_ex(method))
			return bound_vbzd(n);                                           // press.c:3411,4409
	}
	return 0;
}

extern "C" uint64_t press_hip_workspace_bytes(int method, uint64_t total_samples, uint32_t nreads)
{
	uint64_t b = ((uint64_t) nreads + 1) * sizeof(ReadMeta);
	if (is_zs(method)) { // the inner stream between the two stages dominates
		b += press_hip_workspace_bytes(zs_inner(method), total_samples, nreads) + zs_tmp_bytes(method, total_samples, nreads) +
		     (uint64_t) max_zblocks_of(total_samples, nreads) * 64 + ((uint64_t) nreads + 1) * (1024 + sizeof(zs::Table) + 64);
		if (zs_kdiv(method))
			b += 2 * (total_samples + 64) * 4; // the lists of non-zero key bytes
		return b;
	}
	if (is_ex(method))
		b += 2 * (total_samples + 64) * 4;
	if (is_shuff(method))
		b += total_samples + 64 + sizeof(HuffDev);
	return b;
}

// ------------------------------------------------------------------ host <-> device staging
//
// The host-pointer form of the batch calls (device_resident = 0) is what a caller like
// press/test.c uses: its buffers are ordinary (pageable) memory.  Copies go through two
// page-locked staging buffers of STAGE_BYTES: while the DMA engine moves one, the host fills
// (or drains) the other, several threads sharing the memcpy.  Buffers obtained from
// press_hip_host_alloc() are page-locked themselves and are copied by ONE DMA, no staging.
// Compressed streams travel densely: a gather kernel packs the slots' contents before the
// D2H (slots are sized by X_bound, several times their content), and the decoder's input is
// packed on the host while it is staged.

namespace {

constexpr size_t STAGE_BYTES = 32u << 20;
constexpr size_t DIRECT_MAX = 256u << 10; // below this a plain hipMemcpyAsync (HIP's own staging) is cheaper

struct Staging {
	void *buf[2] = { nullptr, nullptr };
	hipEvent_t ev[2];
	bool busy[2] = { false, false };
	bool made = false;
} stg;

int staging_init()
{
	if (stg.made)
		return 0;
	for (int k = 0; k < 2; k++) {
		HIPCHK(hipHostMalloc(&stg.buf[k], STAGE_BYTES, hipHostMallocDefault));
		HIPCHK(hipEventCreateWithFlags(&stg.ev[k], hipEventDisableTiming));
	}
	stg.made = true;
	return 0;
}

void staging_release()
{
	if (!stg.made)
		return;
	for (int k = 0; k < 2; k++) {
		(void) hipEventDestroy(stg.ev[k]);
		(void) hipHostFree(stg.buf[k]);
		stg.buf[k] = nullptr;
		stg.busy[k] = false;
	}
	stg.made = false;
}

int staging_wait(int k)
{
	if (stg.busy[k]) {
		HIPCHK(hipEventSynchronize(stg.ev[k]));
		stg.busy[k] = false;
	}
	return 0;
}

bool is_pinned(const void *p)
{
	hipPointerAttribute_t a;
	if (hipPointerGetAttributes(&a, p) != hipSuccess) {
		(void) hipGetLastError(); // ordinary memory is reported as an error: not one of ours
		return false;
	}
	return a.type == hipMemoryTypeHost;
}

// memcpy shared by a few threads (one core moves ~10 GB/s, the link 50+)
void par_memcpy(void *dst, const void *src, size_t n)
{
	constexpr size_t MIN_PART = 2u << 20;
	unsigned nt = (unsigned) (n / MIN_PART);
	if (nt > 6)
		nt = 6;
	if (nt < 2) {
		memcpy(dst, src, n);
		return;
	}
	const size_t part = (n / nt + 63) & ~(size_t) 63;
	std::vector<std::thread> th;
	for (unsigned t = 1; t < nt; t++) {
		const size_t o = (size_t) t * part;
		if (o >= n)
			break;
		const size_t l = o + part > n ? n - o : part;
		th.emplace_back([=] { memcpy((char *) dst + o, (const char *) src + o, l); });
	}
	memcpy(dst, src, part < n ? part : n);
	for (auto &t : th)
		t.join();
}

// host -> device, asynchronous on s as far as the source allows (returns when src may be reused
// unless src is page-locked)
int h2d(void *dst, const void *src, size_t n, hipStream_t s)
{
	if (!n)
		return 0;
	if (n <= DIRECT_MAX || is_pinned(src)) {
		HIPCHK(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, s));
		return 0;
	}
	int rc = staging_init();
	if (rc)
		return rc;
	int k = 0;
	for (size_t o = 0; o < n; o += STAGE_BYTES, k ^= 1) {
		const size_t l = n - o < STAGE_BYTES ? n - o : STAGE_BYTES;
		if ((rc = staging_wait(k)))
			return rc;
		par_memcpy(stg.buf[k], (const char *) src + o, l);
		HIPCHK(hipMemcpyAsync((char *) dst + o, stg.buf[k], l, hipMemcpyHostToDevice, s));
		HIPCHK(hipEventRecord(stg.ev[k], s));
		stg.busy[k] = true;
	}
	return 0;
}

// Pieces of host memory <-> one dense device range, through the staging buffers.
struct Piece {
	uint8_t *host;   // where the piece lives on the host
	uint64_t dense;  // its offset in the dense range
	uint64_t len;
};

// pieces must be sorted by `dense` and must not overlap.  TO_DEV: host pieces -> dev[0, total);
// else dev[0, total) -> host pieces.  Synchronous for the host memory involved.
template <bool TO_DEV>
int staged_pieces(uint8_t *dev, uint64_t total, const std::vector<Piece> &pc, hipStream_t s)
{
	if (!total)
		return 0;
	int rc = staging_init();
	if (rc)
		return rc;
	size_t ip = 0; // first piece that may reach into the current chunk
	auto host_side = [&](int k, uint64_t o, uint64_t l) { // move the pieces' bytes of chunk [o, o + l)
		while (ip < pc.size() && pc[ip].dense + pc[ip].len <= o)
			ip++;
		for (size_t i = ip; i < pc.size() && pc[i].dense < o + l; i++) {
			const uint64_t a = pc[i].dense > o ? pc[i].dense : o;
			const uint64_t b = pc[i].dense + pc[i].len < o + l ? pc[i].dense + pc[i].len : o + l;
			if (b <= a)
				continue;
			uint8_t *h = pc[i].host + (a - pc[i].dense);
			uint8_t *g = (uint8_t *) stg.buf[k] + (a - o);
			if (TO_DEV)
				par_memcpy(g, h, b - a);
			else
				par_memcpy(h, g, b - a);
		}
	};
	int k = 0;
	if (TO_DEV) {
		for (uint64_t o = 0; o < total; o += STAGE_BYTES, k ^= 1) {
			const uint64_t l = total - o < STAGE_BYTES ? total - o : STAGE_BYTES;
			if ((rc = staging_wait(k)))
				return rc;
			host_side(k, o, l);
			HIPCHK(hipMemcpyAsync(dev + o, stg.buf[k], l, hipMemcpyHostToDevice, s));
			HIPCHK(hipEventRecord(stg.ev[k], s));
			stg.busy[k] = true;
		}
		return 0;
	}
	// device -> host: the DMA of chunk i+1 runs while the host drains chunk i
	uint64_t po = 0, pl = 0;
	int pk = -1;
	for (uint64_t o = 0; o < total; o += STAGE_BYTES, k ^= 1) {
		const uint64_t l = total - o < STAGE_BYTES ? total - o : STAGE_BYTES;
		if ((rc = staging_wait(k)))
			return rc;
		HIPCHK(hipMemcpyAsync(stg.buf[k], dev + o, l, hipMemcpyDeviceToHost, s));
		HIPCHK(hipEventRecord(stg.ev[k], s));
		stg.busy[k] = true;
		if (pk >= 0) {
			if ((rc = staging_wait(pk)))
				return rc;
			host_side(pk, po, pl);
		}
		pk = k;
		po = o;
		pl = l;
	}
	if (pk >= 0) {
		if ((rc = staging_wait(pk)))
			return rc;
		host_side(pk, po, pl);
	}
	return 0;
}

// dense[dense_off[r] ..) = arena[slot_off[r] .. + len[r]) - the streams of a batch packed back to back
// (16-byte aligned) for ONE copy to the host.  One workgroup per (read, 1/8 of its 4-KiB pieces).
__global__ __launch_bounds__(256) void k_gather_streams(const uint8_t *arena, const uint64_t *slot_off,
							const uint64_t *len, const uint64_t *dense_off, uint8_t *dense)
{
	const uint32_t r = blockIdx.x;
	const uint64_t l = len[r];
	if (l == PRESS_HIP_FAILED || l == 0)
		return;
	const uint8_t *src = arena + slot_off[r];
	uint8_t *dst = dense + dense_off[r];
	const uint64_t n16 = l / 16;
	for (uint64_t c = (uint64_t) blockIdx.y * 256 + threadIdx.x; c < n16; c += 256ull * gridDim.y) {
		uint4 v;
		__builtin_memcpy(&v, src + 16 * c, 16); // the slot may sit at any byte address
		*reinterpret_cast<uint4 *>(dst + 16 * c) = v;
	}
	if (blockIdx.y == 0 && threadIdx.x < (l & 15))
		dst[16 * n16 + threadIdx.x] = src[16 * n16 + threadIdx.x];
}

} // namespace

extern "C" void *press_hip_host_alloc(uint64_t bytes)
{
	API_LOCK;
	if (ctx_init())
		return nullptr;
	void *p = nullptr;
	hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
	if (e != hipSuccess) {
		fail(PRESS_HIP_EHIP, "hipHostMalloc(%llu): %s", (unsigned long long) bytes, hipGetErrorString(e));
		return nullptr;
	}
	return p;
}

extern "C" void press_hip_host_free(void *p)
{
	API_LOCK;
	if (p)
		(void) hipHostFree(p);
}

// ------------------------------------------------------------------ batch API

static int launch_press(int method, const BatchArgs &a, hipStream_t s)
{
	switch (method) {
	case PRESS_HIP_SVB12:    launch_svb_encode_chunked(a, false, false, s); break;
	case PRESS_HIP_SVB12_ZD: launch_svb_encode_chunked(a, false, true, s); break;
	case PRESS_HIP_SVB_ZD:   launch_svb_encode_chunked(a, true, true, s); break;
	case PRESS_HIP_SLOW5_SVB_ZD: launch_svb_encode_chunked(a, true, true, s, true); break;
	case PRESS_HIP_ZSTD_SVB_ZD:
	case PRESS_HIP_ZSTD_SVB12_ZD:
	case PRESS_HIP_ZSTD_HASGAM_ZDQ: {
		ZsBufs z;
		zs_bufs(z, g.zs_total, a.nreads, method);
		launch_zstd_encode(a, z, s);
		break;
	}
	default:
		launch_ex_encode_chunked(a, exfmt_of(method), entropy_of(method), s);
	}
	hipError_t e = hipGetLastError();
	if (e != hipSuccess)
		return fail(PRESS_HIP_EHIP, "kernel launch: %s", hipGetErrorString(e));
	return 0;
}

// Frames the device walk leaves to libzstd (dictionaries, 12-bit tables, several frames in one stream ...): their
// content is made on the host and put where the device would have put it.  How many there are comes back through a
// page-locked word behind an event (zs_count_host_frames, queued between the two stages of a batch): the host waits for
// that word only, while the device already runs the second stage - a batch of this library's own frames, or of
// ZSTD_compress's, has no such frame and is never waited for; with one, the second stage is run again (*patched).
static int zs_count_host_frames(const ZsBufs &z, hipStream_t s)
{
	if (!g.zs_pin) {
		HIPCHK(hipHostMalloc((void **) &g.zs_pin, 64, hipHostMallocDefault));
		HIPCHK(hipEventCreateWithFlags(&g.zs_ev, hipEventDisableTiming));
	}
	HIPCHK(hipMemcpyAsync(g.zs_pin, &z.dctl->nhost, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
	HIPCHK(hipEventRecord(g.zs_ev, s));
	return 0;
}
static int zs_host_frames(const DecodeArgs &a, const ZsBufs &z, hipStream_t s, bool *patched)
{
	*patched = false;
	HIPCHK(hipEventSynchronize(g.zs_ev));
	const uint32_t nhost = *g.zs_pin;
	g.zs_nhost = nhost;
	if (!nhost || !zstd_open())
		return 0; // without libzstd those reads fail
	HIPCHK(hipStreamSynchronize(s)); // (the second stage is running on what the device had)
	*patched = true;
	const uint32_t nr = a.nreads;
	std::vector<ZsRead> rd(nr);
	std::vector<uint64_t> ioff(nr), ilen(nr), zoff(nr + 1);
	std::vector<uint32_t> caps(nr);
	HIPCHK(hipMemcpy(rd.data(), z.rd, (size_t) nr * sizeof(ZsRead), hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(ioff.data(), a.in_off, (size_t) nr * 8, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(ilen.data(), a.in_len, (size_t) nr * 8, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(zoff.data(), z.zoff, ((size_t) nr + 1) * 8, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(caps.data(), a.nsamp, (size_t) nr * 4, hipMemcpyDeviceToHost));
	std::vector<uint8_t> frame, buf;
	for (uint32_t r = 0; r < nr; r++) {
		if (rd[r].mode != 3)
			continue;
		uint64_t cap;
		if (z.kdiv) {
			cap = 4ull + (caps[r] + (uint64_t) z.kdiv - 1) / z.kdiv + 2ull * caps[r];
		} else { // device: zs_content_max
			const uint64_t vb = bound_vbzd(caps[r] ? caps[r] : 1);
			cap = (vb + 3) / 4 + vb * 4 + 16;
		}
		frame.resize(ilen[r] + 8);
		buf.resize(cap + 8);
		HIPCHK(hipMemcpy(frame.data(), a.in + ioff[r], ilen[r], hipMemcpyDeviceToHost));
		const size_t got = zstd_fn.decompress(buf.data(), cap, frame.data(), ilen[r]);
		if (zstd_fn.is_error(got)) {
			rd[r].mode = 2;
			continue;
		}
		HIPCHK(hipMemcpy(z.ztmp + zoff[r], buf.data(), got, hipMemcpyHostToDevice));
		rd[r].mode = 0;
		rd[r].nd = (uint32_t) got;
	}
	HIPCHK(hipMemcpy(z.rd, rd.data(), (size_t) nr * sizeof(ZsRead), hipMemcpyHostToDevice));
	return 0;
}

static int launch_depress(int method, const DecodeArgs &a, hipStream_t s)
{
	switch (method) {
	case PRESS_HIP_SVB12:    launch_svb_decode_chunked(a, false, false, s); break;
	case PRESS_HIP_SVB12_ZD: launch_svb_decode_chunked(a, false, true, s); break;
	case PRESS_HIP_SVB_ZD:   launch_svb_decode_chunked(a, true, true, s); break;
	case PRESS_HIP_SLOW5_SVB_ZD: launch_svb_decode_chunked(a, true, true, s, true); break;
	case PRESS_HIP_ZSTD_SVB_ZD:
	case PRESS_HIP_ZSTD_SVB12_ZD:
	case PRESS_HIP_ZSTD_HASGAM_ZDQ: {
		ZsBufs z;
		zs_bufs(z, g.zs_total, a.nreads, method);
		launch_zstd_decode_frames(a, z, s);
		int rc = zs_count_host_frames(z, s);
		if (rc)
			return rc;
		launch_zstd_decode_streams(a, z, s);
		bool patched;
		if ((rc = zs_host_frames(a, z, s, &patched)))
			return rc;
		if (patched)
			launch_zstd_decode_streams(a, z, s);
		break;
	}
	default:
		launch_ex_decode_chunked(a, exfmt_of(method), entropy_of(method), s);
	}
	hipError_t e = hipGetLastError();
	if (e != hipSuccess)
		return fail(PRESS_HIP_EHIP, "kernel launch: %s", hipGetErrorString(e));
	return 0;
}

static int check_method(int method)
{
	if (method < 0 || method >= PRESS_HIP_NMETHODS || (is_zstd(method) && !is_zs(method)))
		return fail(PRESS_HIP_EARG, "method %d is not available in the batch API", method);
	if (is_shuff(method) && !g.have_table)
		return fail(PRESS_HIP_ENOTABLE, "static-Huffman method without a table (press_hip_load_table_file)");
	return 0;
}

extern "C" int press_hip_press_batch(int method, const int16_t *sig, const uint64_t *off, const uint32_t *n,
				     uint32_t nreads, uint64_t total_samples, uint8_t *out,
				     const uint64_t *out_off, uint64_t *out_len, int device_resident)
{
	API_LOCK;
	int rc = ctx_init();
	if (rc)
		return rc;
	if ((rc = check_method(method)))
		return rc;
	if (nreads == 0)
		return 0;
	if (!sig || !off || !n || !out || !out_off || !out_len)
		return fail(PRESS_HIP_EARG, "NULL argument");
	hipStream_t s = g.stream();
	if ((rc = reserve_scratch(method, total_samples, nreads, false)))
		return rc;
	g.zs_total = total_samples;

	BatchArgs a;
	memset(&a, 0, sizeof a);
	a.nreads = nreads;
	a.meta = (ReadMeta *) g.meta.p;
	a.ex_pos = (uint32_t *) g.ex_pos.p;
	a.ex_val = (uint32_t *) g.ex_val.p;
	a.huff = (const HuffDev *) g.huff.p;
	a.chunks = (ChunkDesc *) g.chunks.p;
	a.gran = (uint64_t *) g.gran.p;
	a.ctl = (ChunkCtl *) g.ctl.p;
	a.max_chunks = max_chunks_of(total_samples, nreads);
	if (is_shuff(method))
		a.cbits = (ChunkBits *) g.cbits.p;
	if (is_rc(method))
		a.low_tmp = (uint8_t *) g.low.p;
	a.first_chunk = (uint32_t *) g.first_chunk.p;

	if (device_resident) {
		if ((uintptr_t) sig & 15)
			return fail(PRESS_HIP_EARG, "sig must be 16-byte aligned");
		a.sig = sig;
		a.off = off;
		a.nsamp = n;
		a.out = out;
		a.out_off = out_off;
		a.out_len = out_len;
		return launch_press(method, a, s);
	}

	// host pointers: stage, run, copy back, synchronise
	for (uint32_t r = 0; r < nreads; r++) {
		if (off[r] & 7)
			return fail(PRESS_HIP_EARG, "off[%u] = %llu is not a multiple of 8 samples", r,
				    (unsigned long long) off[r]);
		if (off[r] + n[r] > total_samples)
			return fail(PRESS_HIP_EARG, "read %u ends beyond total_samples", r);
		if (out_off[r + 1] < out_off[r])
			return fail(PRESS_HIP_EARG, "out_off must be non-decreasing");
	}
	const uint64_t a0 = out_off[0], a1 = out_off[nreads];
	if (g.sig.reserve(total_samples * 2 + 64) || g.off.reserve((size_t) nreads * 8) ||
	    g.nsamp.reserve((size_t) nreads * 4) ||
	    g.arena.reserve(a1 - a0 + 64) || g.arena_off.reserve(((size_t) nreads + 1) * 8) ||
	    g.lens.reserve((size_t) nreads * 8))
		return PRESS_HIP_EHIP;
	std::vector<uint64_t> rel(nreads + 1);
	for (uint32_t r = 0; r <= nreads; r++)
		rel[r] = out_off[r] - a0;
	HIPCHK(hipMemcpyAsync(g.off.p, off, (size_t) nreads * 8, hipMemcpyHostToDevice, s));
	HIPCHK(hipMemcpyAsync(g.nsamp.p, n, (size_t) nreads * 4, hipMemcpyHostToDevice, s));
	HIPCHK(hipMemcpyAsync(g.arena_off.p, rel.data(), ((size_t) nreads + 1) * 8, hipMemcpyHostToDevice, s));
	if ((rc = h2d(g.sig.p, sig, total_samples * 2, s)))
		return rc;
	a.sig = (const int16_t *) g.sig.p;
	a.off = (const uint64_t *) g.off.p;
	a.nsamp = (const uint32_t *) g.nsamp.p;
	a.out = (uint8_t *) g.arena.p;
	a.out_off = (const uint64_t *) g.arena_off.p;
	a.out_len = (uint64_t *) g.lens.p;
	if ((rc = launch_press(method, a, s)))
		return rc;
	HIPCHK(hipMemcpyAsync(out_len, g.lens.p, (size_t) nreads * 8, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	if (nreads <= 4) { // per-read calls: one small copy each
		for (uint32_t r = 0; r < nreads; r++) {
			if (out_len[r] == PRESS_HIP_FAILED || out_len[r] == 0)
				continue;
			HIPCHK(hipMemcpyAsync(out + out_off[r], (uint8_t *) g.arena.p + rel[r], out_len[r],
					      hipMemcpyDeviceToHost, s));
		}
		HIPCHK(hipStreamSynchronize(s));
		return 0;
	}
	// the streams packed back to back on the device, ONE pass over the link, scattered into the
	// caller's slots by the host
	std::vector<uint64_t> doff(nreads);
	std::vector<Piece> pc;
	pc.reserve(nreads);
	uint64_t dense = 0;
	for (uint32_t r = 0; r < nreads; r++) {
		doff[r] = dense;
		if (out_len[r] == PRESS_HIP_FAILED || out_len[r] == 0)
			continue;
		pc.push_back({ out + out_off[r], dense, out_len[r] });
		dense += (out_len[r] + 15) & ~15ull;
	}
	if (!dense)
		return 0;
	if (g.dense.reserve(dense + 64) || g.dense_off.reserve((size_t) nreads * 8))
		return PRESS_HIP_EHIP;
	HIPCHK(hipMemcpyAsync(g.dense_off.p, doff.data(), (size_t) nreads * 8, hipMemcpyHostToDevice, s));
	hipLaunchKernelGGL(k_gather_streams, dim3(nreads, 8), dim3(256), 0, s, (const uint8_t *) g.arena.p,
			   (const uint64_t *) g.arena_off.p, (const uint64_t *) g.lens.p, (const uint64_t *) g.dense_off.p,
			   (uint8_t *) g.dense.p);
	if ((rc = staged_pieces<false>((uint8_t *) g.dense.p, dense, pc, s)))
		return rc;
	HIPCHK(hipStreamSynchronize(s));
	return 0;
}

extern "C" int press_hip_depress_batch(int method, const uint8_t *in, const uint64_t *in_off,
				       const uint64_t *in_len, uint32_t nreads, int16_t *sig,
				       const uint64_t *off, const uint32_t *n, uint64_t total_samples,
				       uint32_t *out_n, int device_resident)
{
	API_LOCK;
	int rc = ctx_init();
	if (rc)
		return rc;
	if ((rc = check_method(method)))
		return rc;
	if (nreads == 0)
		return 0;
	if (!in || !in_off || !in_len || !sig || !off || !n || !out_n)
		return fail(PRESS_HIP_EARG, "NULL argument");
	hipStream_t s = g.stream();
	if ((rc = reserve_scratch(method, total_samples, nreads, true)))
		return rc;
	g.zs_total = total_samples;

	DecodeArgs a;
	memset(&a, 0, sizeof a);
	a.nreads = nreads;
	a.meta = (ReadMeta *) g.meta.p;
	a.ex_pos = (uint32_t *) g.ex_pos.p;
	a.ex_val = (uint32_t *) g.ex_val.p;
	a.low = (uint8_t *) g.low.p;
	a.huff = (const HuffDev *) g.huff.p;
	a.chunks = (ChunkDesc *) g.chunks.p;
	a.gran = (uint64_t *) g.gran.p;
	a.ctl = (ChunkCtl *) g.ctl.p;
	a.first_chunk = (uint32_t *) g.first_chunk.p;
	a.max_chunks = max_chunks_of(total_samples, nreads);
	if (is_shuff(method)) {
		a.htiles = (HufTile *) g.htiles.p;
		a.htrec = (HufTRec *) g.htrec.p;
		a.hrec = (uint32_t *) g.hrec.p;
		a.hlist = (uint32_t *) g.hlist.p;
		a.hread = (uint32_t *) g.hread.p;
		a.hwave = (uint2 *) g.hwave.p;
		a.hbits = (unsigned long long *) g.hbits.p;
		a.hend = (uint8_t *) g.hend.p;
		a.hmin = (uint32_t *) g.hmin.p;
		a.max_htiles = max_htiles_of(total_samples, nreads);
		a.hlist_cap = (uint32_t) hlist_cap_of(a.max_htiles);
		a.huf_minlen = table_minlen() | (g.table_trie ? HUF_NEEDS_TRIE : 0u);
	}

	if (device_resident) {
		if ((uintptr_t) sig & 15)
			return fail(PRESS_HIP_EARG, "sig must be 16-byte aligned");
		a.in = in;
		a.in_off = in_off;
		a.in_len = in_len;
		a.sig = sig;
		a.off = off;
		a.nsamp = n;
		a.out_n = out_n;
		return launch_depress(method, a, s);
	}

	uint64_t dense = 0;
	std::vector<uint64_t> doff(nreads);
	std::vector<Piece> pc;
	pc.reserve(nreads);
	for (uint32_t r = 0; r < nreads; r++) {
		if (off[r] & 7)
			return fail(PRESS_HIP_EARG, "off[%u] is not a multiple of 8 samples", r);
		if (off[r] + n[r] > total_samples)
			return fail(PRESS_HIP_EARG, "slot %u ends beyond total_samples", r);
		doff[r] = dense;
		if (in_len[r])
			pc.push_back({ const_cast<uint8_t *>(in) + in_off[r], dense, in_len[r] });
		dense += in_len[r];
	}
	if (g.sig.reserve(total_samples * 2 + 64) || g.off.reserve((size_t) nreads * 8) ||
	    g.nsamp.reserve((size_t) nreads * 4) || g.arena.reserve(dense + 64) || g.arena_off.reserve((size_t) nreads * 8) ||
	    g.lens2.reserve((size_t) nreads * 8) || g.outn.reserve((size_t) nreads * 4))
		return PRESS_HIP_EHIP;
	HIPCHK(hipMemcpyAsync(g.arena_off.p, doff.data(), (size_t) nreads * 8, hipMemcpyHostToDevice, s));
	HIPCHK(hipMemcpyAsync(g.lens2.p, in_len, (size_t) nreads * 8, hipMemcpyHostToDevice, s));
	HIPCHK(hipMemcpyAsync(g.off.p, off, (size_t) nreads * 8, hipMemcpyHostToDevice, s));
	HIPCHK(hipMemcpyAsync(g.nsamp.p, n, (size_t) nreads * 4, hipMemcpyHostToDevice, s));
	// the streams, packed back to back while they are staged (the caller's slots may be far apart)
	if (nreads <= 4) {
		for (const Piece &q : pc)
			HIPCHK(hipMemcpyAsync((uint8_t *) g.arena.p + q.dense, q.host, q.len, hipMemcpyHostToDevice, s));
	} else if ((rc = staged_pieces<true>((uint8_t *) g.arena.p, dense, pc, s))) {
		return rc;
	}
	a.in = (const uint8_t *) g.arena.p;
	a.in_off = (const uint64_t *) g.arena_off.p;
	a.in_len = (const uint64_t *) g.lens2.p;
	a.sig = (int16_t *) g.sig.p;
	a.off = (const uint64_t *) g.off.p;
	a.nsamp = (const uint32_t *) g.nsamp.p;
	a.out_n = (uint32_t *) g.outn.p;
	if ((rc = launch_depress(method, a, s)))
		return rc;
	HIPCHK(hipMemcpyAsync(out_n, g.outn.p, (size_t) nreads * 4, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	if (nreads <= 4) {
		for (uint32_t r = 0; r < nreads; r++) {
			if (out_n[r] == UINT32_MAX || out_n[r] == 0)
				continue;
			HIPCHK(hipMemcpyAsync(sig + off[r], (int16_t *) g.sig.p + off[r], (size_t) out_n[r] * 2,
					      hipMemcpyDeviceToHost, s));
		}
		HIPCHK(hipStreamSynchronize(s));
		return 0;
	}
	// only the decoded samples of every read reach the caller's buffer (its padding between the
	// reads is left alone); reads in ascending slot order for the staged copy
	std::vector<uint32_t> order(nreads);
	for (uint32_t r = 0; r < nreads; r++)
		order[r] = r;
	bool sorted = true;
	for (uint32_t r = 1; r < nreads && sorted; r++)
		sorted = off[r] >= off[r - 1];
	if (!sorted)
		std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return off[x] < off[y]; });
	pc.clear();
	uint64_t end = 0;
	for (uint32_t i = 0; i < nreads; i++) {
		const uint32_t r = order[i];
		if (out_n[r] == UINT32_MAX || out_n[r] == 0)
			continue;
		if (off[r] * 2 < end)
			return fail(PRESS_HIP_EARG, "sample slots overlap");
		pc.push_back({ (uint8_t *) (sig + off[r]), off[r] * 2, (uint64_t) out_n[r] * 2 });
		end = (off[r] + out_n[r]) * 2;
	}
	if (is_pinned(sig)) { // page-locked: the decoded ranges go straight to the caller, one DMA per run of reads
		size_t i = 0;
		while (i < pc.size()) {
			size_t k = i;
			// reads whose gaps are only the alignment padding travel together (the padding is overwritten)
			while (k + 1 < pc.size() && pc[k + 1].dense - (pc[k].dense + pc[k].len) < 128)
				k++;
			const uint64_t b0 = pc[i].dense, b1 = pc[k].dense + pc[k].len;
			HIPCHK(hipMemcpyAsync((uint8_t *) sig + b0, (uint8_t *) g.sig.p + b0, b1 - b0, hipMemcpyDeviceToHost, s));
			i = k + 1;
		}
		HIPCHK(hipStreamSynchronize(s));
		return 0;
	}
	if ((rc = staged_pieces<false>((uint8_t *) g.sig.p, end, pc, s)))
		return rc;
	HIPCHK(hipStreamSynchronize(s));
	return 0;
}

// ------------------------------------------------------------------ huffman.h objects (press/huffman/huffman.h)

namespace {

huffman_node *new_node(bool leaf, unsigned char sym)
{
	huffman_node *p = (huffman_node *) calloc(1, sizeof *p);
	if (!p)
		return nullptr;
	p->isLeaf = leaf ? 1 : 0;
	if (leaf)
		p->symbol = sym;
	return p;
}

// File format (huffman.c:427-439, :549): u32 BE entry count, u32 BE byte count, then per
// entry {u8 symbol, u8 numbits, ceil(numbits/8) code bytes; bit k of the code at bit k%8
// of byte k/8, bit 0 next to the root}.
int parse_table_file(FILE *fp, uint32_t len[256], uint64_t bits[256])
{
	unsigned char hd[8];
	memset(len, 0, 256 * sizeof len[0]);
	memset(bits, 0, 256 * sizeof bits[0]);
	if (fread(hd, 1, 8, fp) != 8)
		return fail(PRESS_HIP_EARG, "Huffman table: short header");
	uint32_t count = ((uint32_t) hd[0] << 24) | ((uint32_t) hd[1] << 16) | ((uint32_t) hd[2] << 8) | hd[3];
	if (count > 256)
		return fail(PRESS_HIP_EARG, "Huffman table: %u entries", count);
	for (uint32_t i = 0; i < count; i++) {
		int sym = fgetc(fp), nb = fgetc(fp);
		if (sym == EOF || nb == EOF || nb == 0 || nb > 64)
			return fail(PRESS_HIP_EARG, "Huffman table: bad entry %u", i);
		unsigned char code[8] = { 0 };
		if (fread(code, 1, (size_t) (nb + 7) / 8, fp) != (size_t) (nb + 7) / 8)
			return fail(PRESS_HIP_EARG, "Huffman table: truncated");
		uint64_t b = 0;
		for (int k = 0; k < nb; k++)
			if (code[k / 8] & (1u << (k % 8)))
				b |= 1ull << k;
		len[sym] = (uint32_t) nb;
		bits[sym] = b;
	}
	return 0;
}

void collect_codes(const huffman_node *p, uint64_t code, uint32_t depth, uint32_t len[256], uint64_t bits[256])
{
	if (!p || depth > 64)
		return;
	if (p->isLeaf) {
		len[p->symbol] = depth;
		bits[p->symbol] = code;
		return;
	}
	collect_codes(p->zero, code, depth + 1, len, bits);
	collect_codes(p->one, code | (depth < 64 ? 1ull << depth : 0), depth + 1, len, bits);
}

int table_from_encoder(SymbolEncoder *se)
{
	API_LOCK;
	uint32_t len[256];
	uint64_t bits[256];
	if (!se)
		return fail(PRESS_HIP_EARG, "NULL SymbolEncoder");
	for (int s = 0; s < 256; s++) {
		const huffman_code *c = (*se)[s];
		len[s] = 0;
		bits[s] = 0;
		if (!c)
			continue;
		len[s] = (uint32_t) c->numbits;
		for (unsigned long k = 0; k < c->numbits && k < 64; k++)
			if (c->bits[k / 8] & (1u << (k % 8)))
				bits[s] |= 1ull << k;
	}
	int rc = ctx_init();
	return rc ? rc : upload_table(len, bits);
}

int table_from_tree(huffman_node *root)
{
	API_LOCK;
	uint32_t len[256];
	uint64_t bits[256];
	if (!root)
		return fail(PRESS_HIP_EARG, "NULL Huffman tree");
	memset(len, 0, sizeof len);
	memset(bits, 0, sizeof bits);
	collect_codes(root, 0, 0, len, bits);
	int rc = ctx_init();
	return rc ? rc : upload_table(len, bits);
}

} // namespace

extern "C" bool read_code_table(FILE *in, huffman_node **rootOut, unsigned int *dataBytesOut)
{
	uint32_t len[256];
	uint64_t bits[256];
	if (!in || !rootOut)
		return false;
	long at = ftell(in);
	unsigned char hd[8];
	if (fread(hd, 1, 8, in) != 8)
		return false;
	if (dataBytesOut)
		*dataBytesOut = ((uint32_t) hd[4] << 24) | ((uint32_t) hd[5] << 16) | ((uint32_t) hd[6] << 8) | hd[7];
	if (fseek(in, at, SEEK_SET) || parse_table_file(in, len, bits))
		return false;
	huffman_node *root = new_node(false, 0);
	if (!root)
		return false;
	for (int s = 0; s < 256; s++) {
		if (!len[s])
			continue;
		huffman_node *p = root;
		for (uint32_t k = 0; k < len[s]; k++) {
			const bool one = (bits[s] >> k) & 1;
			if (p->isLeaf) { // a code runs through another one (huffman.c:643)
				free_huffman_tree(root);
				return false;
			}
			huffman_node **slot = one ? &p->one : &p->zero;
			if (!*slot) {
				*slot = new_node(k + 1 == len[s], (unsigned char) s);
				if (!*slot) {
					free_huffman_tree(root);
					return false;
				}
				(*slot)->parent = p;
			}
			p = *slot;
		}
	}
	*rootOut = root;
	return true;
}

extern "C" void build_symbol_encoder(huffman_node *subtree, SymbolEncoder *pSF)
{
	if (!subtree || !pSF)
		return;
	if (!subtree->isLeaf) {
		build_symbol_encoder(subtree->zero, pSF);
		build_symbol_encoder(subtree->one, pSF);
		return;
	}
	// walk up to the root to get the length, then fill the bits from the root down
	unsigned long nb = 0;
	for (const huffman_node *p = subtree; p->parent; p = p->parent)
		nb++;
	huffman_code *c = (huffman_code *) malloc(sizeof *c);
	c->numbits = nb;
	c->bits = (unsigned char *) calloc((nb + 7) / 8 + 1, 1);
	unsigned long k = nb;
	for (const huffman_node *p = subtree; p->parent; p = p->parent) {
		k--;
		if (p == p->parent->one)
			c->bits[k / 8] |= (unsigned char) (1u << (k % 8));
	}
	(*pSF)[subtree->symbol] = c;
}

extern "C" void free_encoder(SymbolEncoder *pSE)
{
	if (!pSE)
		return;
	for (int s = 0; s < 256; s++) {
		huffman_code *c = (*pSE)[s];
		if (c) {
			free(c->bits);
			free(c);
		}
	}
	free(pSE);
}

extern "C" void free_huffman_tree(huffman_node *subtree)
{
	if (!subtree)
		return;
	if (!subtree->isLeaf) {
		free_huffman_tree(subtree->zero);
		free_huffman_tree(subtree->one);
	}
	free(subtree);
}

// ------------------------------------------------------------------ drop-in per-read symbols

namespace {

// one read through the batch path; returns 0 and the length, or an error code
int press_one(int method, const int16_t *in, uint32_t n, uint8_t *out, uint64_t cap, uint64_t *len)
{
	const uint64_t off[1] = { 0 };
	const uint64_t ooff[2] = { 0, cap };
	uint64_t l = PRESS_HIP_FAILED;
	if (n == 0)
		return fail(PRESS_HIP_EARG, "empty read");
	int rc = press_hip_press_batch(method, in, off, &n, 1, n, out, ooff, &l, 0);
	if (rc)
		return rc;
	if (l == PRESS_HIP_FAILED)
		return fail(-1, "stream does not fit %llu bytes", (unsigned long long) cap);
	*len = l;
	return 0;
}

int depress_one(int method, const uint8_t *in, uint64_t nbytes, int16_t *out, uint32_t cap, uint32_t *n)
{
	const uint64_t off[1] = { 0 };
	const uint64_t ioff[1] = { 0 };
	const uint64_t ilen[1] = { nbytes };
	uint32_t got = UINT32_MAX;
	if (cap == 0)
		return fail(PRESS_HIP_EARG, "no room for samples");
	int rc = press_hip_depress_batch(method, in, ioff, ilen, 1, out, off, &cap, cap, &got, 0);
	if (rc)
		return rc;
	if (got == UINT32_MAX)
		return fail(-1, "malformed stream");
	*n = got;
	return 0;
}

// exact length of an svb16 / svb32 stream of n values (its callers do not pass it)
uint64_t svb_stream_len(const uint8_t *in, uint32_t n, bool key2)
{
	uint64_t len;
	if (!key2) {
		const uint32_t klen = svb16_keylen(n);
		len = (uint64_t) klen + n;
		for (uint32_t i = 0; i < n / 8; i++)
			len += (uint64_t) __builtin_popcount(in[i]);
		if (n & 7)
			len += (uint64_t) __builtin_popcount(in[n / 8] & ((1u << (n & 7)) - 1));
	} else {
		len = (uint64_t) (n + 3) / 4 + n;
		for (uint32_t i = 0; i < n; i++)
			len += (in[i >> 2] >> (2 * (i & 3))) & 3u;
	}
	return len;
}

void void_press(int method, const int16_t *in, uint64_t n, uint8_t *out, uint64_t *nout)
{
	uint64_t len = 0;
	if (press_one(method, in, (uint32_t) n, out, *nout, &len)) {
		fprintf(stderr, "press_hip: %s\n", g_err);
		len = 0;
	}
	*nout = len;
}

int int_press_inner(int method, const int16_t *in, uint32_t n, uint8_t *out, uint64_t *nout)
{
	uint64_t len = 0;
	int rc = press_one(method, in, n, out, *nout, &len);
	if (rc)
		return -1;
	*nout = len;
	return 0;
}

void void_depress(int method, const uint8_t *in, uint64_t nbytes, int16_t *out, uint32_t *nout)
{
	uint32_t n = 0;
	if (depress_one(method, in, nbytes, out, *nout, &n)) {
		fprintf(stderr, "press_hip: %s\n", g_err);
		n = 0;
	}
	*nout = n;
}

// zstd level 1 (press.h:275) around a GPU-made inner stream (press.c:1865, 2025, 8554)
int zstd_press_(int inner, bool prefix_n, const int16_t *in, uint32_t n, uint8_t *out, uint64_t *nout)
{
	if (!zstd_open())
		return fail(-1, "libzstd not found"), -1;
	const uint64_t cap = (prefix_n ? 4 : 0) + press_hip_bound(inner, n);
	std::vector<uint8_t> buf(cap + 64);
	uint64_t len = 0;
	if (prefix_n)
		memcpy(buf.data(), &n, 4);
	if (press_one(inner, in, n, buf.data() + (prefix_n ? 4 : 0), cap - (prefix_n ? 4 : 0), &len))
		return -1;
	len += prefix_n ? 4 : 0;
	const size_t r = zstd_fn.compress(out, *nout, buf.data(), len, 1);
	if (zstd_fn.is_error(r))
		return -1;
	*nout = r;
	return 0;
}

int zstd_depress_(int inner, bool prefix_n, const uint8_t *in, uint64_t nbytes, int16_t *out, uint32_t *nout)
{
	if (!zstd_open())
		return fail(-1, "libzstd not found"), -1;
	const uint64_t cap = zstd_bound_((uint64_t) *nout * 2); // press.c:1897
	std::vector<uint8_t> buf(cap + 64);
	const size_t r = zstd_fn.decompress(buf.data(), cap, in, nbytes);
	if (zstd_fn.is_error(r))
		return -1;
	uint32_t n = 0;
	if (prefix_n) {
		uint32_t cnt;
		if (r < 4)
			return -1;
		memcpy(&cnt, buf.data(), 4);
		if (cnt > *nout)
			return -1;
		if (cnt && depress_one(inner, buf.data() + 4, r - 4, out, cnt, &n))
			return -1;
	} else if (depress_one(inner, buf.data(), r, out, *nout, &n)) {
		return -1;
	}
	*nout = n;
	return 0;
}

} // namespace

extern "C" {

uint64_t svb12_bound(uint64_t nin) { return bound_svb16((uint32_t) nin); }
void svb12_press(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout)
{
	void_press(PRESS_HIP_SVB12, in, nin, out, nout);
}
void svb12_depress(const uint8_t *in, uint64_t nin, int16_t *out)
{
	uint32_t n = (uint32_t) nin;
	void_depress(PRESS_HIP_SVB12, in, svb_stream_len(in, (uint32_t) nin, false), out, &n);
}

uint64_t svb12_zd_bound(uint64_t nin) { return bound_svb16((uint32_t) nin); }
void svb12_zd_press(const int16_t *in, uint64_t nin, uint8_t *out, uint64_t *nout)
{
	void_press(PRESS_HIP_SVB12_ZD, in, nin, out, nout);
}
void svb12_zd_depress(const uint8_t *in, uint64_t nin, int16_t *out, uint64_t *nout)
{
	uint32_t n = (uint32_t) nin;
	void_depress(PRESS_HIP_SVB12_ZD, in, svb_stream_len(in, (uint32_t) nin, false), out, &n);
	*nout = n;
}

uint64_t svb_zd_bound_16(uint64_t nin) { return bound_svb32((uint32_t) nin); }
void svb_zd_press_16(const int16_t *in, uint64_t nin, uint8_t *out, uint64_t *nout)
{
	void_press(PRESS_HIP_SVB_ZD, in, nin, out, nout);
}
void svb_zd_depress_16(const uint8_t *in, uint64_t nin, int16_t *out, uint64_t *nout)
{
	uint32_t n = (uint32_t) nin;
	void_depress(PRESS_HIP_SVB_ZD, in, svb_stream_len(in, (uint32_t) nin, true), out, &n);
	*nout = n;
}

uint64_t zstd_svb_zd_bound_16(uint32_t nin) { return press_hip_bound(PRESS_HIP_ZSTD_SVB_ZD, nin); }
int zstd_svb_zd_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout)
{
	return zstd_press_(PRESS_HIP_SVB_ZD, true, in, nin, out, nout);
}
int zstd_svb_zd_depress_16(const uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout)
{
	return zstd_depress_(PRESS_HIP_SVB_ZD, true, in, nin, out, nout);
}

uint64_t zstd_svb12_zd_bound(uint32_t nin) { return press_hip_bound(PRESS_HIP_ZSTD_SVB12_ZD, nin); }
int zstd_svb12_zd_press(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout)
{
	return zstd_press_(PRESS_HIP_SVB12_ZD, true, in, nin, out, nout);
}
int zstd_svb12_zd_depress(const uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout)
{
	return zstd_depress_(PRESS_HIP_SVB12_ZD, true, in, nin, out, nout);
}

#define VB_FAMILY(name, id)                                                                      \
	uint64_t name##_zd_bound_16(uint32_t nin) { return bound_vbzd(nin); }                    \
	void name##_zd_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout)   \
	{                                                                                        \
		void_press(id, in, nin, out, nout);                                              \
	}                                                                                        \
	void name##_zd_depress_16(uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout)       \
	{                                                                                        \
		void_depress(id, in, nin, out, nout);                                            \
	}
VB_FAMILY(vbe21, PRESS_HIP_VBE21_ZD)
VB_FAMILY(vbbe21, PRESS_HIP_VBBE21_ZD)
VB_FAMILY(vbsbe21, PRESS_HIP_VBSBE21_ZD)
VB_FAMILY(vbsse21, PRESS_HIP_VBSSE21_ZD)

// rc_vbe21_zd (press.h:712-716): same shape; *nout of depress must be the exact sample count (press.c:5464)
uint64_t rc_vbe21_zd_bound_16(uint32_t nin) { return press_hip_bound(PRESS_HIP_RC_VBE21_ZD, nin); }
void rc_vbe21_zd_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout)
{
	void_press(PRESS_HIP_RC_VBE21_ZD, in, nin, out, nout);
}
void rc_vbe21_zd_depress_16(uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout)
{
	void_depress(PRESS_HIP_RC_VBE21_ZD, in, nin, out, nout);
}
// rcc_vbe21_zd (press.c:5510-5580): the order-1 coder
uint64_t rcc_vbe21_zd_bound_16(uint32_t nin) { return press_hip_bound(PRESS_HIP_RCC_VBE21_ZD, nin); }
void rcc_vbe21_zd_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout)
{
	void_press(PRESS_HIP_RCC_VBE21_ZD, in, nin, out, nout);
}
void rcc_vbe21_zd_depress_16(uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout)
{
	void_depress(PRESS_HIP_RCC_VBE21_ZD, in, nin, out, nout);
}
// rccm_vbbe21_zd (press.c:6901-7000): vbbe21 + the order 1-0 context-mixing coder
uint64_t rccm_vbbe21_zd_bound_16(uint32_t nin) { return press_hip_bound(PRESS_HIP_RCCM_VBBE21_ZD, nin); }
void rccm_vbbe21_zd_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout)
{
	void_press(PRESS_HIP_RCCM_VBBE21_ZD, in, nin, out, nout);
}
void rccm_vbbe21_zd_depress_16(uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout)
{
	void_depress(PRESS_HIP_RCCM_VBBE21_ZD, in, nin, out, nout);
}

#define SHUFF_FAMILY(name, id)                                                                           \
	uint64_t shuffman_##name##_zd_bound_16(uint32_t nin) { return bound_vbzd(nin); }                 \
	int shuffman_##name##_zd_press_16(SymbolEncoder *se, const int16_t *in, uint32_t nin,            \
					  uint8_t *out, uint64_t *nout)                                  \
	{                                                                                                \
		if (table_from_encoder(se))                                                              \
			return 1;                                                                        \
		return int_press_inner(id, in, nin, out, nout);                                          \
	}                                                                                                \
	int shuffman_##name##_zd_depress_16(huffman_node *root, uint8_t *in, uint64_t nin, int16_t *out, \
					    uint32_t *nout)                                              \
	{                                                                                                \
		if (table_from_tree(root))                                                               \
			return 1;                                                                        \
		uint32_t n = 0;                                                                          \
		if (depress_one(id, in, nin, out, *nout, &n))                                            \
			return 1;                                                                        \
		*nout = n;                                                                               \
		return 0;                                                                                \
	}
SHUFF_FAMILY(vbe21, PRESS_HIP_SHUFF_VBE21_ZD)
SHUFF_FAMILY(vbbe21, PRESS_HIP_SHUFF_VBBE21_ZD)
SHUFF_FAMILY(vbsbe21, PRESS_HIP_SHUFF_VBSBE21_ZD)
SHUFF_FAMILY(vbsse21, PRESS_HIP_SHUFF_VBSSE21_ZD)

uint64_t hasgam_vbsse21_zdq_bound_16(uint32_t nin) { return press_hip_bound(PRESS_HIP_HASGAM_ZDQ, nin); }
int hasgam_vbsse21_zdq_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout)
{
	return int_press_inner(PRESS_HIP_HASGAM_ZDQ, in, nin, out, nout);
}
int hasgam_vbsse21_zdq_depress_16(uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout)
{
	uint32_t n = 0;
	if (depress_one(PRESS_HIP_HASGAM_ZDQ, in, nin, out, *nout, &n))
		return -1;
	*nout = n;
	return 0;
}

uint64_t zstd_hasgam_vbsse21_zdq_bound_16(uint32_t nin) { return press_hip_bound(PRESS_HIP_ZSTD_HASGAM_ZDQ, nin); }
int zstd_hasgam_vbsse21_zdq_press_16(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout)
{
	return zstd_press_(PRESS_HIP_HASGAM_ZDQ, false, in, nin, out, nout);
}
int zstd_hasgam_vbsse21_zdq_depress_16(uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout)
{
	return zstd_depress_(PRESS_HIP_HASGAM_ZDQ, false, in, nin, out, nout);
}

// ---- BLOW5's signal codec "svb-zd" (slow5lib slow5_press.c:1054,1110), SURVEY 8f-2
uint64_t slow5_svb_zd_bound(uint32_t nin) { return press_hip_bound(PRESS_HIP_SLOW5_SVB_ZD, nin); }
int slow5_svb_zd_press(const int16_t *in, uint32_t nin, uint8_t *out, uint64_t *nout)
{
	if (nin == 0) { // an empty signal is just its count
		if (*nout < 4)
			return -1;
		memset(out, 0, 4);
		*nout = 4;
		return 0;
	}
	return int_press_inner(PRESS_HIP_SLOW5_SVB_ZD, in, nin, out, nout);
}
int slow5_svb_zd_depress(const uint8_t *in, uint64_t nin, int16_t *out, uint32_t *nout)
{
	uint32_t cnt, n = 0;
	if (nin < 4)
		return -1;
	memcpy(&cnt, in, 4); // slow5_press.c:1086: the count travels in the stream
	if (cnt > *nout)
		return -1;
	if (cnt == 0) {
		*nout = 0;
		return nin == 4 ? 0 : -1;
	}
	if (depress_one(PRESS_HIP_SLOW5_SVB_ZD, in, nin, out, cnt, &n))
		return -1;
	*nout = n;
	return 0;
}
// the shape slow5lib itself uses (slow5_ptr_compress_solo / slow5_ptr_depress_solo with
// SLOW5_COMPRESS_SVB_ZD, slow5_press.h:103-105): malloc'd result, byte counts in and out
void *press_hip_slow5_ptr_compress_svb_zd(const int16_t *ptr, size_t count, size_t *n)
{
	const uint32_t ns = (uint32_t) (count / sizeof *ptr);
	uint64_t len = slow5_svb_zd_bound(ns);
	uint8_t *out = (uint8_t *) malloc(len + 16);
	if (!out || slow5_svb_zd_press(ptr, ns, out, &len)) {
		free(out);
		return nullptr;
	}
	*n = (size_t) len;
	return out;
}
void *press_hip_slow5_ptr_depress_svb_zd(const uint8_t *ptr, size_t count, size_t *n)
{
	uint32_t cnt;
	if (count < 4)
		return nullptr;
	memcpy(&cnt, ptr, 4);
	int16_t *out = (int16_t *) malloc(((size_t) cnt + 8) * sizeof *out);
	uint32_t got = cnt;
	if (!out || slow5_svb_zd_depress(ptr, count, out, &got)) {
		free(out);
		return nullptr;
	}
	*n = (size_t) got * sizeof *out;
	return out;
}

} // extern "C"
