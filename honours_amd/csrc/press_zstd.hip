// press_zstd.hip - zstd frames made on the device (BASELINE config 3: the
// "VBZ" pipeline zstd(svb-zd), press.c:1860 zstd_svb_zd_press_16).
//
// The reference gives the buffer [u32 n][svb32 keys][svb32 data] to libzstd.  What a drop-in
// has to keep is the format (RFC 8878): every zstd decoder must return that buffer.  A frame
// written here is laid out for a GPU on both sides - no sequences, only
//   * one raw block with the count,
//   * RLE blocks for the key bytes (they are zero except where a delta needs two bytes),
//   * the data bytes in blocks of 16 KiB, Huffman-coded literals in four independent bit
//     streams each, with ONE table per read (the first block carries the tree, the others are
//     "treeless"), or raw where that does not pay,
// i.e. ~60 independent streams of <= 4096 bytes per mean read.  zs_table.h builds the table.
// tests/: libzstd decodes these frames to the reference's buffer; oracle/zsframe_model.cpp is a
// serial model that must give the same bytes.
//
// Kernels (encode): k_zs_layout -> [the inner stream into ztmp; its encoder counts the data bytes per read and (svb) counts
// and lists the key bytes that are not zero: press_chunked.hip, HIST] -> k_zs_blocks -> k_zs_blockmap -> k_zs_table ->
// k_zs_bits -> k_zs_plan -> k_zs_encode (+ k_zs_rawframes for reads that do not shrink).

#include "press_internal.h"
#include "zs_table.h"
#include <type_traits>

namespace ph {

namespace {

constexpr uint32_t ZB = zs::BLOCK_LITS;
constexpr uint32_t RLE_MAX = 131072; // Block_Maximum_Size
constexpr uint64_t ZFAIL = ~0ull;

__device__ __forceinline__ uint64_t wave_incl64(uint64_t v, int lane)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint64_t o = __shfl_up(v, d);
		if (lane >= d)
			v += o;
	}
	return v;
}
__device__ __forceinline__ uint32_t wave_incl32(uint32_t v, int lane)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t o = __shfl_up(v, d);
		if (lane >= d)
			v += o;
	}
	return v;
}
// 16 bytes at any byte address with the non-temporal policy (streams that are read once: press_chunked.hip's ld16_stream
// wants alignment)
__device__ __forceinline__ uint4 ld16_nt_any(const uint8_t *p)
{
	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	typedef u32x4 u32x4_u __attribute__((aligned(1)));
	const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_u *>(p));
	return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t wave_sum32(uint32_t v)
{
#pragma unroll
	for (int d = 32; d; d >>= 1)
		v += __shfl_xor(v, d);
	return v;
}

// exclusive scan over a 1024-thread workgroup; *total = sum
__device__ __forceinline__ uint64_t wg_excl_scan64(uint64_t v, uint64_t *wsum, uint64_t *total)
{
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint64_t inc = wave_incl64(v, lane);
	__syncthreads();
	if (lane == 63)
		wsum[w] = inc;
	__syncthreads();
	uint64_t before = 0, all = 0;
	for (int i = 0; i < (int) (blockDim.x >> 6); i++) {
		const uint64_t x = wsum[i];
		if (i < w)
			before += x;
		all += x;
	}
	*total = all;
	return before + inc - v;
}

// key bytes of n samples: kdiv = 4 samples per key byte (svb32, zstd_svb_zd) or 8 (svb16, zstd_svb12_zd)
__device__ __forceinline__ uint32_t zs_nk(uint32_t n, uint32_t kdiv) { return kdiv ? (uint32_t) (((uint64_t) n + kdiv - 1) / kdiv) : 0u; }
// most bytes the inner stream of n samples can have
__device__ __forceinline__ uint64_t zs_content_max(uint32_t n, uint32_t kdiv)
{
	if (kdiv) // [u32 n][keys][data]: a 16-bit zig-zag delta is at most two bytes
		return 4ull + zs_nk(n, kdiv) + 2ull * n;
	// ex-zd (hasgam_vbsse21_zdq): the reference's own bound, press.c:2575, 3411, 8461
	const uint32_t m1 = n ? n - 1 : 0;
	const uint64_t vb = 2 + (uint64_t) (1 + m1 * 0.2 * 6 + m1 * 0.8);
	return (vb + 3) / 4 + vb * 4 + 16;
}
__device__ __forceinline__ uint64_t zs_slot(uint32_t n, uint32_t kdiv)
{
	return ((zs_content_max(n, kdiv) + 16) + 15) & ~15ull;
}

// where the svb-zd stream of every read goes in ztmp (one workgroup)
__global__ __launch_bounds__(1024) void k_zs_layout(const uint32_t *nsamp, uint32_t nreads, uint64_t *zoff, uint64_t *zoff4,
						    uint32_t kdiv)
{
	__shared__ uint64_t wsum[16];
	uint64_t carry = 0;
	for (uint32_t base = 0; base <= nreads; base += 1024) {
		const uint32_t r = base + threadIdx.x;
		const uint64_t v = r < nreads ? zs_slot(nsamp[r], kdiv) : 0;
		uint64_t total;
		const uint64_t ex = wg_excl_scan64(v, wsum, &total);
		if (r <= nreads) {
			zoff[r] = carry + ex;
			zoff4[r] = carry + ex + 4;
		}
		carry += total;
	}
}

// data blocks of every read (one workgroup): first_blk, the per-read record, the total
__global__ __launch_bounds__(1024) void k_zs_blocks(const uint32_t *nsamp, const uint64_t *zlen, uint32_t nreads,
						    uint32_t *first_blk, ZsRead *rd, uint32_t *nblocks, uint32_t max_blocks, uint32_t kdiv,
						    uint8_t *ztmp, const uint64_t *zoff, const ReadMeta *meta)
{
	__shared__ uint64_t wsum[16];
	uint64_t carry = 0;
	for (uint32_t base = 0; base <= nreads; base += 1024) {
		const uint32_t r = base + threadIdx.x;
		uint64_t nb = 0;
		if (r < nreads) {
			const uint32_t n = nsamp[r];
			const uint64_t l = zlen[r];
			ZsRead z;
			z.nk = zs_nk(n, kdiv);
			z.mode = 0;
			z.knz = 0;
			z.dbase = 0;
			z.pad[0] = z.pad[1] = 0;
			// the prefix: the count (svb streams; it is put in front of the stream here) or the
			// ex-zd header + exception section (whatever is not a one-byte value)
			uint64_t body = l; // keys + data
			if (kdiv) {
				z.plen = 4;
				uint8_t *c = ztmp + zoff[r];
				c[0] = (uint8_t) n;
				c[1] = (uint8_t) (n >> 8);
				c[2] = (uint8_t) (n >> 16);
				c[3] = (uint8_t) (n >> 24);
			} else {
				z.plen = meta[r].hdr + meta[r].seclen;
				if (l != ZFAIL && (meta[r].status || l < z.plen || n == 0))
					body = ZFAIL;
				else if (l != ZFAIL)
					body = l - z.plen;
			}
			if (body == ZFAIL || body < z.nk) {
				z.nd = 0;
				z.mode = 2;
			} else {
				z.nd = (uint32_t) (body - z.nk);
			}
			nb = z.mode == 0 ? (z.nd + ZB - 1) / ZB : 0;
			rd[r] = z;
		}
		uint64_t total;
		const uint64_t ex = wg_excl_scan64(nb, wsum, &total);
		if (r <= nreads)
			first_blk[r] = (uint32_t) (carry + ex < max_blocks ? carry + ex : max_blocks);
		carry += total;
	}
	if (threadIdx.x == 0)
		*nblocks = (uint32_t) (carry < max_blocks ? carry : max_blocks);
}

__global__ __launch_bounds__(256) void k_zs_blockmap(const uint32_t *first_blk, uint32_t nreads, const uint32_t *nblocks,
						     uint32_t *blk_read)
{
	const uint32_t b = blockIdx.x * 256 + threadIdx.x;
	if (b >= *nblocks)
		return;
	uint32_t lo = 0, hi = nreads; // the last r with first_blk[r] <= b
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) / 2;
		if (first_blk[mid] <= b)
			lo = mid;
		else
			hi = mid;
	}
	blk_read[b] = lo;
}

struct BlkU {
	uint32_t r, j, R;
	const uint8_t *data; // the block's bytes in ztmp
};
__device__ __forceinline__ BlkU load_blk(const ZsBufs &z, uint32_t b)
{
	BlkU u;
	u.r = z.blk_read[b];
	u.j = b - z.first_blk[u.r];
	const ZsRead *rd = z.rd + u.r;
	const uint32_t nd = rd->nd;
	u.R = nd - u.j * ZB < ZB ? nd - u.j * ZB : ZB;
	u.data = z.ztmp + z.zoff[u.r] + rd->plen + rd->nk + (uint64_t) u.j * ZB;
	return u;
}

// the lanes of a wave inside zs::build_table
struct WavePar {
	uint32_t l;
	__device__ uint32_t lane() const { return l; }
	__device__ uint32_t lanes() const { return 64; }
	__device__ void sync() const
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
	__device__ void inc(uint32_t *p) const { atomicAdd(p, 1u); }
#ifdef HUF_STAMPS
	unsigned long long *last; // (LDS, per wave)
	__device__ void stamp(int i) const;
#else
	__device__ void stamp(int) const {}
#endif
};
#ifdef HUF_STAMPS
__device__ unsigned long long g_tstamp[8];
__device__ void WavePar::stamp(int i) const
{
	if (l == 0) {
		const unsigned long long n = __builtin_amdgcn_s_memtime();
		atomicAdd(&g_tstamp[i], n - *last);
		*last = n;
	}
}
#endif

// one wave per read: the Huffman table from the histogram, and the ranks of the key lists
struct TabLds {
	uint8_t order[256];
	zs::Table t;
	zs::Work k;
};
__global__ __launch_bounds__(256) void k_zs_table(BatchArgs a, ZsBufs z)
{
	__shared__ TabLds lds[4];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#ifdef HUF_STAMPS
	__shared__ unsigned long long tlast[4];
	if (lane == 0)
		tlast[w] = __builtin_amdgcn_s_memtime();
	const WavePar wpar{ (uint32_t) lane, &tlast[w] };
#else
	const WavePar wpar{ (uint32_t) lane };
#endif
	const uint32_t r = blockIdx.x * 4 + w;
	if (r >= a.nreads)
		return;
	ZsRead *rd = z.rd + r;
	if (rd->mode)
		return;
	TabLds &L = lds[w];
	// the counts live where build_table keeps the nodes' parents: it reads them (into Work::w) before it writes those
	static_assert(sizeof(L.k.parent) >= 256 * sizeof(uint32_t) && offsetof(zs::Work, parent) % 4 == 0, "counts in Work::parent");
	uint32_t *cnt = reinterpret_cast<uint32_t *>(L.k.parent);
	// ---- the key bytes that are not zero: counted and listed by the svb encoder (the read's count in its first chunk's
	// slot); the exception-split stream has no keys
	if (lane == 0) {
		const uint32_t n = a.nsamp[r];
		rd->knz = z.kdiv && n ? z.kcnt[a.first_chunk[r]] : 0u;
	}
	// ---- the table
	uint32_t mine[4], present = 0;
	for (int i = 0; i < 4; i++) {
		mine[i] = z.hist[(uint64_t) r * 256 + lane + 64 * i];
		present += mine[i] != 0;
	}
	present = wave_sum32(present);
	zs::Table *gt = (zs::Table *) z.tab + r;
	if (present == 0) {
		if (lane == 0)
			gt->ok = 0;
		return;
	}
	if (present == 1) { // a lone byte value gets a partner (zsframe_model.cpp)
		const uint32_t zero_has = __shfl(mine[0], 0);
		if (lane == (zero_has ? 1 : 0))
			mine[0] = 1;
		present = 2;
	}
	for (int i = 0; i < 4; i++)
		cnt[lane + 64 * i] = mine[i];
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	uint32_t rank[4] = { 0, 0, 0, 0 };
	for (uint32_t s = 0; s < 256; s++) {
		const uint32_t c = cnt[s];
		if (!c)
			continue;
		for (int i = 0; i < 4; i++) {
			const uint32_t me = lane + 64 * i;
			rank[i] += c < mine[i] || (c == mine[i] && s < me);
		}
	}
	for (int i = 0; i < 4; i++)
		if (mine[i])
			L.order[rank[i]] = (uint8_t) (lane + 64 * i);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	wpar.stamp(0); // key ranks, counts, order
	zs::build_table(cnt, L.order, present, L.t, L.k, wpar);
	wpar.stamp(6); // (direct description / the rest)
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	const uint32_t *src = (const uint32_t *) &L.t;
	uint32_t *dst = (uint32_t *) gt;
	for (uint32_t i = lane; i < sizeof(zs::Table) / 4; i += 64)
		dst[i] = src[i];
}

// code bits of the four streams of a data block
__global__ __launch_bounds__(256) void k_zs_bits(ZsBufs z)
{
	__shared__ uint8_t len[256];
	const uint32_t b = blockIdx.x;
	if (b >= *z.nblocks)
		return;
	const BlkU u = load_blk(z, b);
	const zs::Table *t = (const zs::Table *) z.tab + u.r;
	if (!t->ok || u.R < zs::MIN_HUF_LITS) {
		if (threadIdx.x == 0)
			z.sbits[b] = make_uint4(0, 0, 0, 0);
		return;
	}
	len[threadIdx.x] = t->len[threadIdx.x];
	__syncthreads();
	const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
	const uint32_t seg = (u.R + 3) / 4;
	const uint32_t k = q < 3 ? seg : u.R - 3 * seg;
	const uint8_t *s = u.data + (uint64_t) q * seg;
	uint32_t bits = 0;
	for (uint32_t i0 = lane * 16; i0 < k; i0 += 1024) {
		if (i0 + 16 <= k) {
			uint4 v;
			v = ld16_nt_any(s + i0); // (read once here, once by k_zs_encode: streamed, 0.23 -> 0.20 ms)
			const uint32_t x[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
			for (int d = 0; d < 4; d++)
#pragma unroll
				for (int e = 0; e < 4; e++)
					bits += len[(x[d] >> (8 * e)) & 0xFFu];
		} else {
			for (uint32_t e = i0; e < k; e++)
				bits += len[s[e]];
		}
	}
	bits = wave_sum32(bits);
	__shared__ uint32_t qb[4];
	if (lane == 0)
		qb[q] = bits;
	__syncthreads();
	if (threadIdx.x == 0)
		z.sbits[b] = make_uint4(qb[0], qb[1], qb[2], qb[3]);
}

__device__ __forceinline__ void put_block_header(uint8_t *p, bool last, uint32_t type, uint32_t size)
{
	const uint32_t h = (last ? 1u : 0u) | (type << 1) | (size << 3);
	p[0] = (uint8_t) h;
	p[1] = (uint8_t) (h >> 8);
	p[2] = (uint8_t) (h >> 16);
}
__device__ __forceinline__ void put_rle(uint8_t *p, uint32_t len, uint8_t v)
{
	put_block_header(p, false, 1, len);
	p[3] = v;
}

// bytes of the Huffman streams of a block (without the tree): jump table + four streams
__device__ __forceinline__ uint32_t huf_size(uint4 sb)
{
	return 6 + (sb.x / 8 + 1) + (sb.y / 8 + 1) + (sb.z / 8 + 1) + (sb.w / 8 + 1);
}

// one wave per read: the plan of the frame - key blocks written, data blocks placed
__global__ __launch_bounds__(256) void k_zs_plan(BatchArgs a, ZsBufs z)
{
	const int lane = threadIdx.x & 63;
	const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= a.nreads)
		return;
	ZsRead *rd = z.rd + r;
	if (rd->mode) {
		if (lane == 0)
			a.out_len[r] = ZFAIL;
		return;
	}
	const uint32_t nk = rd->nk, nd = rd->nd, knz = rd->knz, plen = rd->plen;
	const uint64_t L = (uint64_t) plen + nk + nd;
	const uint32_t np = (plen + RLE_MAX - 1) / RLE_MAX; // raw blocks of the prefix
	const uint8_t *S = z.ztmp + z.zoff[r];
	const uint64_t cap = a.out_off[r + 1] - a.out_off[r];
	uint8_t *out = a.out + a.out_off[r];
	const uint32_t *kpos = a.ex_pos + a.off[r];
	const uint32_t *kval = a.ex_val + a.off[r];
	const zs::Table *t = (const zs::Table *) z.tab + r;
	const uint32_t b0 = z.first_blk[r], nblk = (nd + ZB - 1) / ZB;
	const uint32_t desc_len = t->ok ? t->desc_len : 0;

	// ---- key blocks: a run of zeros (pieces of at most 128 KiB) in front of every other key byte
	uint32_t nkb = 0;
	for (uint32_t i = 0; i < knz; i += 64) {
		uint32_t blocks = 0;
		if (i + lane < knz) {
			const uint32_t p = kpos[i + lane];
			const uint32_t gap = i + lane ? p - kpos[i + lane - 1] - 1 : p;
			blocks = (gap + RLE_MAX - 1) / RLE_MAX + 1;
		}
		nkb += wave_sum32(blocks);
	}
	const uint32_t tail = knz ? nk - 1 - kpos[knz - 1] : nk;
	nkb += (tail + RLE_MAX - 1) / RLE_MAX;
	const uint32_t kbase0 = 9 + 3 * np + plen; // frame offset of the first key block
	const uint32_t dbase = kbase0 + 4 * nkb;

	// ---- data blocks: the first one that pays WITH the tree carries it
	uint32_t F = 0xFFFFFFFFu;
	if (t->ok) {
		for (uint32_t i = 0; i < nblk && F == 0xFFFFFFFFu; i += 64) {
			bool cand = false;
			if (i + lane < nblk) {
				const uint32_t R = nd - (i + lane) * ZB < ZB ? nd - (i + lane) * ZB : ZB;
				const uint4 sb = z.sbits[b0 + i + lane];
				const uint32_t hs = huf_size(sb);
				cand = R >= zs::MIN_HUF_LITS && 6 + desc_len + hs < R && sb.x / 8 + 1 < 65536 && sb.y / 8 + 1 < 65536 &&
				       sb.z / 8 + 1 < 65536;
			}
			const unsigned long long m = __ballot(cand);
			if (m)
				F = i + (uint32_t) __builtin_ctzll(m);
		}
	}
	uint64_t total = dbase;
	for (uint32_t i = 0; i < nblk; i += 64) {
		uint32_t size = 0, flag = 0;
		if (i + lane < nblk) {
			const uint32_t bi = i + lane;
			const uint32_t R = nd - bi * ZB < ZB ? nd - bi * ZB : ZB;
			const uint4 sb = z.sbits[b0 + bi];
			const uint32_t hs = huf_size(sb);
			bool huf = false;
			if (bi == F)
				huf = true;
			else if (bi > F && F != 0xFFFFFFFFu)
				huf = R >= zs::MIN_HUF_LITS && 6 + hs < R;
			if (huf) {
				flag = 1u | (bi == F ? 2u : 0u);
				size = 3 + 5 + (bi == F ? desc_len : 0) + hs + 1;
			} else {
				size = 3 + R;
			}
			if (bi + 1 == nblk)
				flag |= 4u;
		}
		const uint64_t inc = wave_incl64(size, lane);
		if (i + lane < nblk) {
			z.bpos[b0 + i + lane] = (uint32_t) (total + inc - size);
			z.bflag[b0 + i + lane] = (uint8_t) flag;
		}
		total += __shfl(inc, 63);
	}
	const uint64_t raw_total = 9 + L + 3 * ((L + RLE_MAX - 1) / RLE_MAX);
	uint32_t mode = 0;
	uint64_t size = total;
	if (total >= raw_total) {
		mode = 1;
		size = raw_total;
	}
	if (size > cap) {
		mode = 2;
		size = ZFAIL;
	}
	if (lane == 0) {
		rd->mode = mode;
		rd->dbase = dbase;
		a.out_len[r] = size;
	}
	if (mode)
		return;
	// ---- frame header, the count, the key blocks
	if (lane == 0) {
		out[0] = 0x28;
		out[1] = 0xB5;
		out[2] = 0x2F;
		out[3] = 0xFD;
		out[4] = 0xA0; // single segment, 4-byte content size
		out[5] = (uint8_t) L;
		out[6] = (uint8_t) (L >> 8);
		out[7] = (uint8_t) (L >> 16);
		out[8] = (uint8_t) (L >> 24);
		for (uint32_t at = 0; at < plen; at += RLE_MAX) {
			const uint32_t len = plen - at < RLE_MAX ? plen - at : RLE_MAX;
			put_block_header(out + 9 + at + 3 * (at / RLE_MAX), at + len == plen && nk == 0 && nd == 0, 0, len);
		}
	}
	for (uint32_t i = lane; i < plen; i += 64) // the prefix itself
		out[9 + 3 * (i / RLE_MAX + 1) + i] = S[i];
	uint32_t kb = 0; // key blocks in front
	for (uint32_t i = 0; i < knz; i += 64) {
		uint32_t blocks = 0, gap = 0;
		if (i + lane < knz) {
			const uint32_t p = kpos[i + lane];
			gap = i + lane ? p - kpos[i + lane - 1] - 1 : p;
			blocks = (gap + RLE_MAX - 1) / RLE_MAX + 1;
		}
		const uint32_t inc = wave_incl32(blocks, lane);
		if (i + lane < knz) {
			uint8_t *p = out + kbase0 + 4ull * (kb + inc - blocks);
			for (uint32_t at = 0; at < gap; at += RLE_MAX, p += 4)
				put_rle(p, gap - at < RLE_MAX ? gap - at : RLE_MAX, 0);
			put_rle(p, 1, (uint8_t) kval[i + lane]);
		}
		kb += __shfl(inc, 63);
	}
	if (lane == 0) {
		uint8_t *p = out + kbase0 + 4ull * kb;
		for (uint32_t at = 0; at < tail; at += RLE_MAX, p += 4)
			put_rle(p, tail - at < RLE_MAX ? tail - at : RLE_MAX, 0);
	}
}

// one workgroup per data block, one wave per bit stream
constexpr int ZSTG = 1416; // dwords: 4096 codes of at most 11 bits + the end mark
__global__ __launch_bounds__(256) void k_zs_encode(BatchArgs a, ZsBufs z)
{
	__shared__ uint32_t enc[256];       // code | len << 16
	__shared__ uint32_t stg_all[4][ZSTG];
	const uint32_t b = blockIdx.x;
	if (b >= *z.nblocks)
		return;
	const BlkU u = load_blk(z, b);
	if (z.rd[u.r].mode)
		return;
	const uint32_t flag = z.bflag[b];
	uint8_t *blk = a.out + a.out_off[u.r] + z.bpos[b];
	const bool last = flag & 4u;
	if (!(flag & 1u)) { // raw block
		if (threadIdx.x == 0)
			put_block_header(blk, last, 0, u.R);
		for (uint32_t i = threadIdx.x; i < u.R; i += 256)
			blk[3 + i] = u.data[i];
		return;
	}
	const zs::Table *t = (const zs::Table *) z.tab + u.r;
	enc[threadIdx.x] = (uint32_t) t->code[threadIdx.x] | ((uint32_t) t->len[threadIdx.x] << 16);
	const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
	uint32_t *stg = stg_all[q];
	for (int i = lane; i < ZSTG; i += 64)
		stg[i] = 0;
	const uint32_t seg = (u.R + 3) / 4;
	const uint32_t k = q < 3 ? seg : u.R - 3 * seg;
	// ---- a lane's run of the stream: 64 bytes (the lanes behind a short stream's end have none), loaded straight into
	// registers.  (Through LDS, a lane's 64-byte run 64 bytes from its neighbour's: every dword read of the runs was a
	// 32-way bank conflict - two thirds of the kernel's LDS cycles - and the 16 KB cost a third of the resident waves.)
	const uint32_t lo = 64u * lane < k ? 64u * lane : k, hi = lo + 64 < k ? lo + 64 : k;
	const uint32_t nmine = hi - lo;
	uint32_t run[16];
	{
		const uint8_t *s = u.data + (uint64_t) q * seg + lo;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			uint4 v = make_uint4(0, 0, 0, 0);
			if (16u * j + 16u <= nmine) {
				__builtin_memcpy(&v, s + 16 * j, 16); // (non-temporal here: 0.05 ms slower - what k_zs_bits left in the caches helps)
			} else if (16u * j < nmine) { // the stream's last, ragged 16 bytes (one lane)
				uint32_t x[4] = { 0, 0, 0, 0 };
				for (uint32_t e = 16u * j; e < nmine; e++)
					x[(e >> 2) & 3] |= (uint32_t) s[e] << (8 * (e & 3));
				v = make_uint4(x[0], x[1], x[2], x[3]);
			}
			run[4 * j] = v.x;
			run[4 * j + 1] = v.y;
			run[4 * j + 2] = v.z;
			run[4 * j + 3] = v.w;
		}
	}
	__syncthreads();
	const uint4 sb = z.sbits[b];
	const uint32_t sbq[4] = { sb.x, sb.y, sb.z, sb.w };
	const uint32_t desc_len = (flag & 2u) ? t->desc_len : 0;
	const uint32_t body = desc_len + huf_size(sb);
	uint8_t *p = blk + 3 + 5 + desc_len; // the jump table
	if (threadIdx.x == 0) {
		put_block_header(blk, last, 2, 5 + body + 1);
		const uint64_t lh = ((flag & 2u) ? 2u : 3u) | (3u << 2) | ((uint64_t) u.R << 4) | ((uint64_t) body << 22);
		for (int i = 0; i < 5; i++)
			blk[3 + i] = (uint8_t) (lh >> (8 * i));
		for (int i = 0; i < 3; i++) {
			const uint32_t sz = sbq[i] / 8 + 1;
			p[2 * i] = (uint8_t) sz;
			p[2 * i + 1] = (uint8_t) (sz >> 8);
		}
		blk[3 + 5 + body] = 0; // no sequences
	}
	if (desc_len)
		for (uint32_t i = threadIdx.x; i < desc_len; i += 256)
			blk[8 + i] = t->desc[i];
	// ---- the stream: the bytes back to front, the first byte's code on top, then the end mark.
	// A lane packs its contiguous run of codes in a register and ORs whole dwords into LDS.
	// (Four bytes at a time, so that the look-ups of four codes are in flight together.)
	const bool whole = !__any(nmine != 0 && nmine != 64); // no lane of the wave with a ragged run
	uint32_t mybits = 0;
	if (whole) {
		if (nmine) {
#pragma unroll
			for (int w4 = 0; w4 < 16; w4++) {
				const uint32_t four = run[w4];
				mybits += (enc[four & 0xFFu] >> 16) + (enc[(four >> 8) & 0xFFu] >> 16) + (enc[(four >> 16) & 0xFFu] >> 16) +
					  (enc[four >> 24] >> 16);
			}
		}
	} else {
#pragma unroll
		for (int w4 = 0; w4 < 16; w4++)
#pragma unroll
			for (int e = 0; e < 4; e++)
				if ((uint32_t) (4 * w4 + e) < nmine)
					mybits += enc[(run[w4] >> (8 * e)) & 0xFFu] >> 16;
	}
	const uint32_t inc = wave_incl32(mybits, lane);
	const uint32_t totalb = __shfl(inc, 63);
	uint32_t pos = totalb - inc; // bits of the lanes behind this one
	{
		uint64_t acc = 0;
		uint32_t nb = pos & 31u, wd = pos >> 5;
		auto put = [&](uint32_t e) {
			acc |= (uint64_t) (e & 0xFFFFu) << nb;
			nb += e >> 16;
			if (nb >= 32) {
				atomicOr(&stg[wd++], (uint32_t) acc);
				acc >>= 32;
				nb -= 32;
			}
		};
		if (whole) {
			if (nmine) {
#pragma unroll
				for (int w4 = 15; w4 >= 0; w4--) {
					const uint32_t four = run[w4];
					const uint32_t e3 = enc[four >> 24], e2 = enc[(four >> 16) & 0xFFu], e1 = enc[(four >> 8) & 0xFFu],
						       e0 = enc[four & 0xFFu];
					put(e3);
					put(e2);
					put(e1);
					put(e0);
				}
			}
		} else {
#pragma unroll
			for (int w4 = 15; w4 >= 0; w4--)
#pragma unroll
				for (int e = 3; e >= 0; e--)
					if ((uint32_t) (4 * w4 + e) < nmine)
						put(enc[(run[w4] >> (8 * e)) & 0xFFu]);
		}
		if (lane == 0) { // the end mark sits on top of the first byte's code
			acc |= 1ull << nb;
			nb++;
		}
		if (nb)
			atomicOr(&stg[wd], (uint32_t) acc);
		if (nb > 32)
			atomicOr(&stg[wd + 1], (uint32_t) (acc >> 32));
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	uint8_t *dst = p + 6;
	for (int i = 0; i < q; i++)
		dst += sbq[i] / 8 + 1;
	const uint32_t bytes = totalb / 8 + 1;
	for (uint32_t i = lane; i < bytes / 4; i += 64) {
		const uint32_t v = stg[i];
		__builtin_memcpy(dst + 4ull * i, &v, 4);
	}
	if (lane < (int) (bytes & 3u)) {
		const uint32_t at = (bytes & ~3u) + lane;
		dst[at] = (uint8_t) (stg[at >> 2] >> (8 * (at & 3)));
	}
}

// reads that do not shrink: the stream in raw blocks of 128 KiB
__global__ __launch_bounds__(256) void k_zs_rawframes(BatchArgs a, ZsBufs z)
{
	const uint32_t r = blockIdx.x;
	if (z.rd[r].mode != 1)
		return;
	const uint64_t L = (uint64_t) z.rd[r].plen + z.rd[r].nk + z.rd[r].nd;
	uint8_t *out = a.out + a.out_off[r];
	const uint8_t *S = z.ztmp + z.zoff[r];
	if (threadIdx.x == 0) {
		out[0] = 0x28;
		out[1] = 0xB5;
		out[2] = 0x2F;
		out[3] = 0xFD;
		out[4] = 0xA0;
		for (int i = 0; i < 4; i++)
			out[5 + i] = (uint8_t) (L >> (8 * i));
		for (uint64_t at = 0; at < L; at += RLE_MAX) {
			const uint64_t len = L - at < RLE_MAX ? L - at : RLE_MAX;
			put_block_header(out + 9 + at + 3 * (at / RLE_MAX), at + len == L, 0, (uint32_t) len);
		}
	}
	for (uint64_t i = threadIdx.x; i < L; i += 256) {
		out[9 + 3 * (i / RLE_MAX + 1) + i] = S[i];
	}
}

} // namespace

void launch_zstd_encode(const BatchArgs &a, const ZsBufs &z, hipStream_t s)
{
	if (!a.nreads)
		return;
	hipLaunchKernelGGL(k_zs_layout, dim3(1), dim3(1024), 0, s, a.nsamp, a.nreads, z.zoff, z.zoff4, z.kdiv);
	BatchArgs sv = a; // the inner stream of every read into ztmp (svb: behind the place of its count)
	sv.out = z.ztmp;
	sv.out_off = z.kdiv ? z.zoff4 : z.zoff;
	sv.out_len = z.zlen;
	(void) hipMemsetAsync(z.hist, 0, (size_t) a.nreads * 1024, s);
	sv.zhist = z.hist; // the data bytes - and (svb) the key bytes that are not zero - are counted where they are made
	sv.zkcnt = z.kcnt;
	ktime_mute(true);
	if (z.kdiv)
		launch_svb_encode_chunked(sv, z.kdiv == 4, true, s);
	else
		launch_ex_encode_chunked(sv, EXF_EXZD, 0, s);
	ktime_mute(false);
	hipLaunchKernelGGL(k_zs_blocks, dim3(1), dim3(1024), 0, s, a.nsamp, z.zlen, a.nreads, z.first_blk, z.rd, z.nblocks,
			   z.max_blocks, z.kdiv, z.ztmp, z.zoff, a.meta);
	hipLaunchKernelGGL(k_zs_blockmap, dim3((z.max_blocks + 255) / 256), dim3(256), 0, s, z.first_blk, a.nreads, z.nblocks,
			   z.blk_read);
	hipLaunchKernelGGL(k_zs_table, dim3((a.nreads + 3) / 4), dim3(256), 0, s, a, z);
	hipLaunchKernelGGL(k_zs_bits, dim3(z.max_blocks), dim3(256), 0, s, z);
	hipLaunchKernelGGL(k_zs_plan, dim3((a.nreads + 3) / 4), dim3(256), 0, s, a, z);
	ktime_begin(0, s);
	hipLaunchKernelGGL(k_zs_encode, dim3(z.max_blocks), dim3(256), 0, s, a, z);
	ktime_end(0, s);
	hipLaunchKernelGGL(k_zs_rawframes, dim3(a.nreads), dim3(256), 0, s, a, z);
}

// ==================================================================== decode
//
// k_zs_layout -> k_zs_walk<lean> (one wave per read: zs::walk_frame checks the frame, carries out fills and short copies
// and lists the other pieces; frames with sequences or long blocks are left to) -> k_zs_walk<full> -> k_zs_copy (long raw
// blocks) + k_zs_hdecode (one lane per Huffman stream - hd_units - and, for blocks with more literals than this library
// writes, 32 lanes per stream - hd_long -, side by side in one launch) -> k_zs_exec (one wave per frame: the sequences of
// libzstd's own frames) -> [the caller lets libzstd do the frames the walk left to it: dictionaries, 12-bit Huffman tables,
// several frames in one stream] -> k_zs_finish -> svb-zd decode.  Neither this library's frames nor ZSTD_compress's need libzstd.

namespace {

// Every lane of the wave runs the same walk on the same bytes (uniform control flow, no
// divergence between reads); lane 0 writes what is found.
constexpr uint32_t ZU = 8; // Huffman blocks per unit (32 streams: half a wave)
constexpr uint32_t ZCOPY_INLINE = 2048; // raw blocks up to this size are copied by the walking wave itself
constexpr uint32_t ZLONG_R = zs::BLOCK_LITS + 1; // literals of a block from which each of its four streams is decoded by 16 lanes
                                                 // (more than a block of this library's frames holds)
struct DevSink {
	const ZsBufs &z;
	uint64_t in_base, out_base; // arena offset of the frame, ztmp offset of the content
	uint32_t read;
	uint32_t cur_tree;          // index of the table in force
	uint32_t unit, ucount;      // the unit being filled
	bool overflow;

	uint32_t cbase, cleft;      // copy slots taken eight at a time (one atomic), unused ones are cleared
	uint64_t lit_abs = 0;       // ztmp offset of the frame's literals space
	uint32_t ntree_mine = 0;    // trees of this frame so far
	uint32_t nreads = 0;        // reads of the batch
	const uint8_t *in = nullptr; // the compressed arena
	bool lean = false;           // the walk that leaves frames with sequences or long blocks to the one behind it
	uint32_t first_xblk = 0, cur_xblk = 0, seq0 = 0; // blocks with sequences: the frame's chain, the one being filled

	__device__ uint32_t take(uint32_t *ctr, uint32_t n = 1)
	{
		uint32_t i = 0;
		if ((threadIdx.x & 63) == 0)
			i = atomicAdd(ctr, n);
		return (uint32_t) __shfl((int) i, 0);
	}
	// how the lanes share the loops of zs::read_tree
	__device__ uint32_t lane() const { return threadIdx.x & 63; }
	__device__ uint32_t lanes() const { return 64; }
	__device__ uint32_t sum(uint32_t v) const { return wave_sum32(v); }
	__device__ void sync() const
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
#ifdef HUF_STAMPS
	unsigned long long st_t = 0, st_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	__device__ void stamp(int i)
	{
		__builtin_amdgcn_sched_barrier(0);
		const unsigned long long n = __builtin_amdgcn_s_memtime();
		st_acc[i] += n - st_t;
		st_t = n;
		__builtin_amdgcn_sched_barrier(0);
	}
#endif
	// frame bytes for the walk: the lanes share the loads
	__device__ void fetch(uint8_t *dst, const uint8_t *src, uint32_t n)
	{
		for (uint32_t j = threadIdx.x & 63; j < n; j += 64)
			dst[j] = src[j];
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
	__device__ void close_copies()
	{
		if ((threadIdx.x & 63) == 0)
			for (uint32_t j = 0; j < cleft; j++)
				if (cbase + j < z.cap_copy)
					z.dcopy[cbase + j].n = 0;
		cleft = 0;
	}
	__device__ void push_copy(uint64_t src, uint64_t dst, uint32_t n, uint32_t fill, bool lit)
	{
		if (!n)
			return;
		if (!cleft) {
			cbase = take(&z.dctl->ncopy, 8);
			cleft = 8;
		}
		const uint32_t i = cbase++;
		cleft--;
		if (i < z.cap_copy) {
			ZsCopy c;
			c.src = in_base + src;
			c.dst = (lit ? lit_abs : out_base) + dst;
			c.n = n;
			c.fill = fill;
			if ((threadIdx.x & 63) == 0)
				z.dcopy[i] = c;
		} else {
			overflow = true;
		}
	}
	// Fills and short copies are carried out by the walking wave itself (the key blocks of one of this library's frames are a
	// dozen fills; through the piece list they were 130 000 mostly empty entries for k_zs_copy); only long raw blocks are
	// queued for k_zs_copy, where a workgroup takes each.  (walk_frame has checked that dst + n stays in the read's slot.)
	__device__ void copy(uint64_t src, uint64_t dst, uint32_t n, bool lit)
	{
		if (n > ZCOPY_INLINE) {
			push_copy(src, dst, n, 0, lit);
			return;
		}
		uint8_t *d = z.ztmp + (lit ? lit_abs : out_base) + dst;
		const uint8_t *sp = in + in_base + src;
		const uint32_t lane = threadIdx.x & 63;
		for (uint32_t k = lane * 16; k < n; k += 64 * 16) {
			if (k + 16 <= n) {
				uint4 v;
				__builtin_memcpy(&v, sp + k, 16);
				__builtin_memcpy(d + k, &v, 16);
			} else {
				for (uint32_t e = k; e < n; e++)
					d[e] = sp[e];
			}
		}
	}
	__device__ void fill(uint64_t, uint64_t dst, uint32_t n, bool lit, uint32_t value)
	{
		if (!n)
			return;
		uint8_t *d = z.ztmp + (lit ? lit_abs : out_base) + dst;
		const uint8_t v = (uint8_t) value; // (from the walk's window: a load from the arena here was a memory round trip per block)
		const uint32_t v4 = v * 0x01010101u, lane = threadIdx.x & 63;
		// bytes up to a 16-byte boundary, 16 at a time, the rest
		const uint32_t head = (uint32_t) ((16 - ((uintptr_t) d & 15)) & 15);
		const uint32_t h = head < n ? head : n;
		if (lane < h)
			d[lane] = v;
		const uint32_t mid = (n - h) / 16;
		for (uint32_t k = lane; k < mid; k += 64)
			*reinterpret_cast<uint4 *>(d + h + 16ull * k) = make_uint4(v4, v4, v4, v4);
		const uint32_t rest = h + 16 * mid + lane;
		if (rest < n)
			d[rest] = v;
	}
	// a block with sequences: a record of its own, chained to the frame's earlier ones, and room for
	// its sequences (k_zs_exec carries them out)
	__device__ int64_t seq_block(uint32_t nseq, uint64_t lit, uint32_t R, uint64_t dst)
	{
		(void) R;
		const uint32_t b = take(&z.dctl->nxblk);
		const uint32_t q = take(&z.dctl->nseq, nseq);
		if (b >= z.cap_xblk || (uint64_t) q + nseq > z.cap_seq)
			return zs::W_HOST;
		if ((threadIdx.x & 63) == 0) {
			ZsXBlk x;
			x.lit = lit_abs + lit;
			x.dst = out_base + dst;
			x.seq0 = q;
			x.nseq = nseq;
			x.tail = 0;
			x.next = 0;
			z.dxblk[b] = x;
			if (cur_xblk)
				z.dxblk[cur_xblk - 1].next = b + 1;
		}
		if (!first_xblk)
			first_xblk = b + 1;
		cur_xblk = b + 1;
		seq0 = q;
		return 0;
	}
	__device__ void seq(uint32_t i, uint32_t ll, uint32_t ml, uint32_t off)
	{
		if ((threadIdx.x & 63) == 0) {
			ZsSeq q;
			q.ll = ll;
			q.ml = ml;
			q.off = off;
			z.dseq[seq0 + i] = q;
		}
	}
	__device__ void seq_end(uint32_t tail)
	{
		if ((threadIdx.x & 63) == 0)
			z.dxblk[cur_xblk - 1].tail = tail;
	}
	__device__ void close_unit()
	{
		if (unit != 0xFFFFFFFFu && (threadIdx.x & 63) == 0) {
			ZsUnit u;
			u.read = read;
			u.tree = cur_tree;
			u.count = ucount;
			u.pad = 0;
			z.dunit[unit] = u;
		}
		unit = 0xFFFFFFFFu;
		ucount = 0;
	}
	__device__ int64_t tree(const uint8_t *w, uint32_t tl)
	{
		close_unit();
		// a read's first tree is slot `read` of the list (no counter), the others follow the reads' slots
		uint32_t i = read;
		if (ntree_mine++)
			i = nreads + take(&z.dctl->ntrees);
		if (i >= z.cap_trees)
			return zs::W_HOST;
		ZsTree *t = z.dtree + i;
		for (int s = threadIdx.x & 63; s < 256; s += 64)
			t->w[s] = w[s];
		if ((threadIdx.x & 63) == 0)
			t->tl = tl;
		cur_tree = i;
		return 0;
	}
	__device__ int64_t huf(uint64_t src, uint32_t cs, uint64_t dst, uint32_t R, bool four, bool lit)
	{
		if (four && R >= ZLONG_R) { // four long streams: a wave of its own (hd_long)
			if (lean) // (no frame of this library has such a block: the lean walk stays lean)
				return zs::W_SEQ;
			const uint32_t i = take(&z.dctl->nlong);
			if (i >= z.cap_long)
				return zs::W_HOST;
			if ((threadIdx.x & 63) == 0) {
				ZsLong L;
				L.h.src = in_base + src;
				L.h.dst = (lit ? lit_abs : out_base) + dst;
				L.h.cs = cs;
				L.h.R = R;
				L.h.four = 1;
				L.h.pad = 0;
				L.tree = cur_tree;
				L.read = read;
				L.pad[0] = L.pad[1] = 0;
				z.dlong[i] = L;
			}
			return 0;
		}
		if (unit == 0xFFFFFFFFu || ucount == ZU) {
			close_unit();
			const uint32_t u = take(&z.dctl->nunits);
			if (u >= z.cap_units)
				return zs::W_HOST;
			unit = u;
		}
		ZsHuf h;
		h.src = in_base + src;
		h.dst = (lit ? lit_abs : out_base) + dst;
		h.cs = cs;
		h.R = R;
		h.four = four;
		h.pad = 0;
		if ((threadIdx.x & 63) == 0)
			z.dhuf[(uint64_t) unit * ZU + ucount] = h;
		ucount++;
		return 0;
	}
};

#ifdef HUF_STAMPS
// diagnostic build only (tools/zsstamps.py): s_memtime differences of a wave of k_zs_hdecode, summed per phase
__device__ unsigned long long g_zstamp[8];
__device__ unsigned long long g_wstamp[8];
__device__ uint32_t g_walk_ticks[8192]; // k_zs_walk: s_memtime ticks of read r's wave
#define ZSTAMP_DECL                                             \
	unsigned long long zs_t = __builtin_amdgcn_s_memtime(); \
	unsigned long long zs_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }
#define ZSTAMP(i)                                                              \
	do {                                                                   \
		__builtin_amdgcn_sched_barrier(0);                             \
		const unsigned long long zs_n = __builtin_amdgcn_s_memtime(); \
		zs_acc[(i) & 7] += zs_n - zs_t;                                \
		zs_t = zs_n;                                                   \
		__builtin_amdgcn_sched_barrier(0);                             \
	} while (0)
#define ZSTAMP_FLUSH()                                                   \
	do {                                                             \
		if ((threadIdx.x & 63) == 0)                             \
			for (int zs_i = 0; zs_i < 8; zs_i++)             \
				atomicAdd(&g_zstamp[zs_i], zs_acc[zs_i]); \
	} while (0)
#else
#define ZSTAMP_DECL
#define ZSTAMP(i)
#define ZSTAMP_FLUSH()
#endif

// (four reads = four waves per workgroup: a CU holds twice as many waves that way)
// (room for 4 waves per SIMD: 128 instead of 147 registers, 0.71 -> 0.69 ms; 5: slower)
#ifndef ZSWALK_WAVES
#define ZSWALK_WAVES 4
#endif
// LEAN: the walk of frames without sequences (this library's own) - under 1 KB of scratch per read and half the
// registers, so that every read of a batch has its wave resident at once (the walk is a chain of memory round trips:
// with 16 waves per CU a batch of 8192 reads went through the chip in two rounds).  It leaves a frame at its first block
// with sequences (mode 4) to the full walk behind it, whose waves end at once for the reads that are done.
#ifndef ZSWALK_LEAN_WAVES
#define ZSWALK_LEAN_WAVES 6 // (80 registers; 8 = 64 registers spills: 0.75 ms against 0.55)
#endif
template <bool LEAN>
__global__ __launch_bounds__(256, LEAN ? ZSWALK_LEAN_WAVES : ZSWALK_WAVES) void k_zs_walk(DecodeArgs a, ZsBufs z)
{
	using Work = std::conditional_t<LEAN, zs::ReadWorkLean, zs::ReadWork>;
	__shared__ Work works[4];
	Work &work = works[threadIdx.x >> 6];
	const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= a.nreads)
		return;
	if (!LEAN && z.rd[r].mode != 4)
		return;
#ifdef HUF_STAMPS
	const unsigned long long wt0 = __builtin_amdgcn_s_memtime();
#endif
	const uint32_t cap_n = a.nsamp[r];
	const uint64_t cap = zs_content_max(cap_n, z.kdiv); // what zs_slot() leaves room for
	DevSink sink{ z, a.in_off[r], z.zoff[r], r, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, false, 0, 0 };
	sink.lit_abs = z.lit_base + z.zoff[r];
	sink.nreads = a.nreads;
	sink.in = a.in;
	sink.lean = LEAN;
#ifdef HUF_STAMPS
	sink.st_t = wt0;
#endif
	int64_t L = zs::walk_frame(a.in + a.in_off[r], a.in_len[r], cap, sink, work);
	sink.close_unit();
	sink.close_copies();
	if (L >= 0 && sink.overflow)
		L = zs::W_HOST;
#ifdef HUF_STAMPS
	sink.stamp(6); // the end of the walk
	if ((threadIdx.x & 63) == 0 && r < 8192)
		g_walk_ticks[r] = (uint32_t) (__builtin_amdgcn_s_memtime() - wt0);
	if ((threadIdx.x & 63) == 0)
		for (int i = 0; i < 8; i++)
			atomicAdd(&g_wstamp[i], sink.st_acc[i]);
#endif
	if (threadIdx.x & 63)
		return;
	ZsRead rd;
	rd.nk = rd.knz = rd.dbase = rd.plen = 0;
	rd.pad[0] = L >= 0 ? sink.first_xblk : 0;
	rd.pad[1] = 0;
	rd.nd = L >= 0 ? (uint32_t) L : 0;
	rd.mode = L >= 0 ? 0 : L == zs::W_HOST ? 3 : L == zs::W_SEQ ? 4 : 2;
	z.rd[r] = rd;
	if (rd.mode == 3)
		atomicAdd(&z.dctl->nhost, 1u);
}

__device__ __forceinline__ void copy_piece(const DecodeArgs &a, const ZsBufs &z, const ZsCopy c)
{
	if (!c.n)
		return;
	uint8_t *d = z.ztmp + c.dst;
	const uint8_t *s = a.in + c.src;
	if (c.fill) {
		const uint8_t v = s[0];
		const uint32_t v4 = v * 0x01010101u;
		// bytes up to a 16-byte boundary, 16 at a time, the rest
		const uint32_t head = (uint32_t) ((16 - ((uintptr_t) d & 15)) & 15);
		const uint32_t h = head < c.n ? head : c.n;
		if (threadIdx.x < h)
			d[threadIdx.x] = v;
		const uint32_t mid = (c.n - h) / 16;
		for (uint32_t k = threadIdx.x; k < mid; k += 256)
			*reinterpret_cast<uint4 *>(d + h + 16ull * k) = make_uint4(v4, v4, v4, v4);
		for (uint32_t k = h + 16 * mid + threadIdx.x; k < c.n; k += 256)
			d[k] = v;
	} else {
		for (uint32_t k = threadIdx.x * 16; k < c.n; k += 256 * 16) {
			if (k + 16 <= c.n) {
				uint4 v;
				__builtin_memcpy(&v, s + k, 16);
				__builtin_memcpy(d + k, &v, 16);
			} else {
				for (uint32_t e = k; e < c.n; e++)
					d[e] = s[e];
			}
		}
	}
}

__global__ __launch_bounds__(256) void k_zs_copy(DecodeArgs a, ZsBufs z)
{
	const uint32_t total = z.dctl->ncopy < z.cap_copy ? z.dctl->ncopy : z.cap_copy;
	for (uint32_t i = blockIdx.x; i < total; i += gridDim.x)
		copy_piece(a, z, z.dcopy[i]);
}

#ifndef HD_SYMS_N
#define HD_SYMS_N 32
#endif
constexpr uint32_t HD_SYMS = HD_SYMS_N; // codes per round: at most 11 bits each + 12 bits of look-ahead - less than a chunk
                                        // (k_zs_hdecode on config 3 with 16 / 32 / 64: 1.35 / 1.14-1.19 / 1.27 ms - 12 / 9 / 6 waves per CU)
constexpr uint32_t HD_CH = 2 * HD_SYMS; // bytes of a chunk of the ring (on a boundary of as many)
constexpr int HD_PC = HD_CH / 16;       // 16-byte pieces of a chunk
constexpr uint32_t HD_RD = HD_CH / 2;   // dwords of a lane's ring (two chunks)
static_assert(HD_SYMS % 16 == 0 && 11 * HD_SYMS + 12 < 8 * HD_CH, "a round stays inside the lower chunk");
// The decoding table of a tree in LDS: bytes in the order of the weights (zs::huf_build_dtable), always indexed by 11 stream
// bits (a shorter table log: every entry 2^(11 - log) times); dt[i] = byte | bits << 8.  All 64 lanes of a wave; false: the
// weights are not a tree (cannot happen: read_tree checked them).
__device__ __forceinline__ bool hd_build_table(uint16_t *dt, const ZsTree *t, int lane)
{
	const uint32_t tl = t->tl;
	// ---- table: bytes in the order of the weights (zs::huf_build_dtable), all lanes fill
	uint32_t w4[4], rank[4];
	{
		const uint32_t four = reinterpret_cast<const uint32_t *>(t->w)[lane];
		for (int i = 0; i < 4; i++)
			w4[i] = (four >> (8 * i)) & 0xFFu;
	}
	uint32_t start_x[12];
	uint32_t at = 0;
	for (uint32_t x = 1; x <= 11; x++) {
		uint32_t mine = 0;
		for (int i = 0; i < 4; i++) {
			if (w4[i] == x)
				rank[i] = mine;
			mine += w4[i] == x;
		}
		const uint32_t inc = wave_incl32(mine, lane);
		for (int i = 0; i < 4; i++)
			if (w4[i] == x)
				rank[i] += inc - mine;
		start_x[x] = at;
		at += __shfl(inc, 63) << (x - 1);
	}
	uint32_t st[4], nn[4];
	for (int i = 0; i < 4; i++) {
		const uint32_t x = w4[i];
		nn[i] = x ? 1u << (x - 1) : 0;
		uint32_t base = 0;
		for (uint32_t y = 1; y <= 11; y++)
			base = x == y ? start_x[y] : base;
		st[i] = base + rank[i] * nn[i];
	}
	if (at != (1u << tl) || tl > 11) // cannot happen: read_tree checked the weights
		return false;
	// a byte's run of entries: up to 16 by the lane that owns the byte (most bytes have long codes: short runs), the
	// longer ones by all lanes, one run after the other (one run after the other for all 256 bytes: 10 % of the
	// kernel's wave time)
	for (int i = 0; i < 4; i++) {
		const uint32_t s0 = st[i] << (11 - tl), n0 = nn[i] << (11 - tl);
		const uint32_t e = (uint32_t) (4 * lane + i) | ((tl + 1 - w4[i]) << 8);
		if (n0 <= 16)
			for (uint32_t k = 0; k < n0; k++)
				dt[s0 + k] = (uint16_t) e;
		unsigned long long big = __ballot(n0 > 16);
		while (big) {
			const int sl = __builtin_ctzll(big);
			big &= big - 1;
			const uint32_t s1 = (uint32_t) __builtin_amdgcn_readlane((int) s0, sl), n1 = (uint32_t) __builtin_amdgcn_readlane((int) n0, sl);
			const uint32_t e1 = (uint32_t) __builtin_amdgcn_readlane((int) e, sl);
			for (uint32_t k = lane; k < n1; k += 64)
				dt[s1 + k] = (uint16_t) e1;
		}
	}
	return true;
}

// one wave per PAIR of units (a unit = up to ZU blocks of one read = 32 streams; the mean read has
// 7 blocks, so whole waves per read would leave more than half of the lanes idle - and the
// kernel is bound by the instructions per decoded byte, not by latency): two tables in LDS,
// one lane per bit stream
// (LDS of k_zs_hdecode: the two tables, always indexed by 11 stream bits - a shorter table log: every entry 2^(11 - log)
// times -, each on a 4096-byte boundary: a look-up's LDS address is (window & 0xFFE) | base, one instruction; and a lane's
// 128 stream bytes, odd stride: a lane's ring starts in its own bank)
typedef uint16_t HdTables[2][2048];
typedef uint32_t HdRings[64][HD_RD + 1];
__device__ __forceinline__ void hd_units(const DecodeArgs &a, const ZsBufs &z, HdTables &dt2, HdRings &ring, uint32_t wg)
{
	const uint32_t total = z.dctl->nunits < z.cap_units ? z.dctl->nunits : z.cap_units;
	if (2 * wg >= total)
		return;
	const int lane = threadIdx.x;
	ZSTAMP_DECL;
	uint32_t cnth[2] = { 0, 0 }, readh[2] = { 0, 0 };
	for (int hh = 0; hh < 2; hh++) {
		const uint32_t uu = 2 * wg + hh;
		if (uu >= total)
			continue;
		const ZsUnit un = z.dunit[uu];
		if (un.tree >= z.cap_trees || un.count == 0 || un.count > ZU)
			continue;
		if (!hd_build_table(dt2[hh], z.dtree + un.tree, lane))
			continue;
		cnth[hh] = un.count;
		readh[hh] = un.read;
	}
	__syncthreads();
	ZSTAMP(0); // tables
	const int half = lane >> 5;
	const uint32_t u = 2 * wg + half;
	typedef __attribute__((address_space(3))) const uint16_t *lds_cu16p;
	const uint32_t dtb = (uint32_t) (uintptr_t) (lds_cu16p) dt2[half]; // (a multiple of 4096)
	ZsUnit un;
	un.count = half ? cnth[1] : cnth[0];
	un.read = half ? readh[1] : readh[0];
	const uint32_t bi = (lane & 31) >> 2, q = lane & 3;
	bool ok = true;
	// ---- one lane per stream.  Global memory is touched in bulk only: per round of HD_SYMS
	// bytes the lanes stage the HD_IN stream bytes below each lane's position in its LDS slot,
	// a lane decodes from there into its output slot, and the lanes store the slots (loads and
	// stores issued from inside the decoding loop wait on each other - they share one counter on
	// this part - and scattered dword stores cost as much as the decoding itself).
	const uint8_t *p = nullptr;
	uint8_t *out = nullptr;
	uint32_t len = 0, k = 0;
	bool active = false;
	if (bi < un.count) {
		const ZsHuf h = z.dhuf[(uint64_t) u * ZU + bi];
		p = a.in + h.src;
		out = z.ztmp + h.dst;
		active = true;
		if (h.four) {
			const uint32_t s1 = p[0] | (p[1] << 8), s2 = p[2] | (p[3] << 8), s3 = p[4] | (p[5] << 8);
			const uint32_t seg = (h.R + 3) / 4;
			if (6ull + s1 + s2 + s3 >= h.cs || 3 * seg > h.R) {
				ok = false;
				active = false;
			} else {
				const uint32_t sz[4] = { s1, s2, s3, h.cs - 6 - s1 - s2 - s3 };
				p += 6;
				for (uint32_t i = 0; i < q; i++)
					p += sz[i];
				len = sz[q];
				k = q < 3 ? seg : h.R - 3 * seg;
				out += (uint64_t) q * seg;
			}
		} else {
			active = q == 0;
			len = h.cs;
			k = h.R;
		}
		if (active && (len == 0 || p[len - 1] == 0)) {
			ok = false;
			active = false;
		}
	}
	// The stream is read backwards from its end mark: bp = stream bits not yet used.  No 64-bit bit window (its shifts run at
	// a quarter of the rate): the 11 bits below bp are cut out of two dwords that stay in registers.
	// (Measured and dropped, round 2: a two-symbol table over 10 index bits, as much LDS as this one - 1.55
	// symbols per step, but with its longer step 2.05 .. 2.3 ms against 1.83 ms.)
	//
	// Stream bytes come through a RING of 128 bytes per lane: the byte at (the low bits of) global address A sits at ring
	// byte A & 127 - two 64-byte chunks on 64-byte boundaries, each fetched ONCE, whole, and a round ahead of its use.
	// (Windows of 96 bytes at any byte address, staged again every round, fetched 2.2 GB for 0.43 GB of streams and put a
	// memory round trip in front of every round: without them the kernel took 0.93 instead of 1.22 - 1.45 ms.)
	// A round is HD_SYMS = 32 codes: at most 44 bytes and 12 bits of look-ahead - from anywhere in the upper chunk that stays
	// inside the lower one.  When the position has moved into the lower chunk, the chunk below it (asked for at the round's
	// start, in registers since) replaces the upper one; the round's bytes are stored BEHIND that, so that the wait for
	// the chunk is not a wait for those stores (the memory counter is in order).
	int32_t bp = 0;
	bool overrun = false;
	if (active)
		bp = (int32_t) (8 * (len - 1)) + (31 - __builtin_clz((uint32_t) p[len - 1]));
	uint32_t *myring = ring[lane];
	const uint32_t ap = (uint32_t) (uintptr_t) p & 0x07FFFFFFu; // (low address bits: 8 ap + the stream's bits stay below 2^31)
	// 16 bytes of chunk address ca (low bits; a multiple of 16) -> registers: whole pieces inside the stream by one load,
	// the stream's first and last piece byte by byte, what lies outside reads as zeros
	// (whole = inside the stream's FRAME: what a stream's first and last piece hold of its neighbours is never used, and
	// byte by byte those two pieces of every stream cost a quarter of a wave's time)
	int32_t flo = 0, fhi = (int32_t) len; // the frame's bytes, as offsets in the stream
	if (active && un.read < a.nreads) {
		const int64_t f0 = (int64_t) a.in_off[un.read] - (int64_t) (p - a.in);
		const int64_t f1 = f0 + (int64_t) a.in_len[un.read];
		if (f0 <= 0 && f1 >= (int64_t) len && f0 > -0x40000000ll && f1 < 0x40000000ll) {
			flo = (int32_t) f0;
			fhi = (int32_t) f1;
		}
	}
	auto piece = [&](uint32_t ca) -> uint4 {
		const int32_t o = (int32_t) (ca - ap); // offset of the piece in the stream
		uint4 v = make_uint4(0, 0, 0, 0);
		if (o >= flo && o + 16 <= fhi) {
			__builtin_memcpy(&v, p + o, 16);
		} else if (o > -16 && o < (int32_t) len) {
			uint32_t x[4] = { 0, 0, 0, 0 };
			for (int e = 0; e < 16; e++)
				if (o + e >= 0 && o + e < (int32_t) len)
					x[e >> 2] |= (uint32_t) p[o + e] << (8 * (e & 3));
			v = make_uint4(x[0], x[1], x[2], x[3]);
		}
		return v;
	};
	auto chunk_to_ring = [&](uint32_t ca, const uint4 *c4) { // chunk address ca (a multiple of 64)
		uint32_t *d = myring + ((ca >> 2) & (HD_RD / 2));
#pragma unroll
		for (int j = 0; j < HD_PC; j++) {
			d[4 * j] = c4[j].x;
			d[4 * j + 1] = c4[j].y;
			d[4 * j + 2] = c4[j].z;
			d[4 * j + 3] = c4[j].w;
		}
	};
	// the chunk that holds the stream's highest bit still to come, and the one below it
	uint32_t cc = 0; // (address of the upper chunk)
	if (active) {
		cc = (ap + (uint32_t) ((bp > 0 ? bp - 1 : 0) >> 3)) & ~(HD_CH - 1u);
		uint4 c4[HD_PC];
#pragma unroll
		for (int j = 0; j < HD_PC; j++)
			c4[j] = piece(cc + 16 * j);
		chunk_to_ring(cc, c4);
#pragma unroll
		for (int j = 0; j < HD_PC; j++)
			c4[j] = piece(cc - HD_CH + 16 * j);
		chunk_to_ring(cc - HD_CH, c4);
	}
	// sp = the (address-based) bit position ONE BELOW the eleven bits a look-up wants: the window cut at sp has the table
	// index in bits 1 .. 11 - twice the index, the entry's byte offset.  The chain from one code to the next is cut -
	// mask/or - look-up - subtract - clamp - dword test - select: the clamp at the stream's bit 0 replaces a compare and a
	// select, the bits are also summed next to the chain and compared with what the stream had when the round is over.
	const int32_t c0 = (int32_t) (8 * ap) - 12;
	int32_t sp = bp + c0;
	uint32_t w0 = (uint32_t) sp >> 5;
	uint32_t lo = myring[w0 & (HD_RD - 1u)], hi = myring[(w0 + 1) & (HD_RD - 1u)];
	uint4 pre[HD_PC]; // the chunk below the ring's two
	bool have_pre = false;
#pragma unroll
	for (int j = 0; j < HD_PC; j++)
		pre[j] = make_uint4(0, 0, 0, 0);
	ZSTAMP(1); // stream headers, first chunks
	for (uint32_t done = 0; __any(active && done < k); done += HD_SYMS) {
		const bool go = active && done < k;
		const uint32_t cnt = go ? (k - done < HD_SYMS ? k - done : HD_SYMS) : 0;
		if (go && !have_pre) { // asked for now, wanted a round or more from now
			const int32_t o = (int32_t) (cc - 2 * HD_CH - ap);
			if (o >= flo && o + (int32_t) HD_CH <= fhi) { // (the rule: the whole chunk is inside the frame)
#pragma unroll
				for (int j = 0; j < HD_PC; j++)
					__builtin_memcpy(&pre[j], p + o + 16 * j, 16);
			} else {
#pragma unroll
				for (int j = 0; j < HD_PC; j++)
					pre[j] = piece(cc - 2 * HD_CH + 16 * j);
			}
			have_pre = true;
		}
		ZSTAMP(2); // the next chunk asked for
		const int32_t bp0 = bp;
		uint32_t used = 0;
		auto step = [&]() -> uint32_t { // one code: its byte
			const uint32_t below = myring[(w0 - 1) & (HD_RD - 1u)];
			const uint32_t y = __builtin_amdgcn_alignbit(hi, lo, (uint32_t) sp); // shift = low 5 bits
			const uint32_t e = *(lds_cu16p) (uintptr_t) ((y & 0xFFEu) | dtb);
			const uint32_t nb = e >> 8;
			used += nb;
			sp = max(sp - (int32_t) nb, c0); // (c0: the stream's bit 0 - only a damaged stream gets there before its last code)
			const uint32_t wn = (uint32_t) sp >> 5;
			if (wn != w0) {
				hi = lo;
				lo = below;
			}
			w0 = wn;
			return e & 0xFFu;
		};
		// ---- sixteen codes at a time into registers
		uint32_t all[HD_SYMS / 4];
#pragma unroll
		for (int i = 0; i < (int) HD_SYMS / 4; i++)
			all[i] = 0;
		const bool whole = !__any(go && cnt != HD_SYMS); // every stream of the wave that is still running has a whole round
		if (whole) {
			if (go) {
#pragma unroll
				for (int j = 0; j < (int) HD_SYMS; j++)
					all[j >> 2] |= step() << (8 * (j & 3));
			}
		} else { // a stream's last, short round
#pragma unroll
			for (int j = 0; j < (int) HD_SYMS; j++)
				if ((uint32_t) j < cnt)
					all[j >> 2] |= step() << (8 * (j & 3));
		}
		if (go) {
			overrun |= used > (uint32_t) bp0; // more code bits than the stream has
			bp = sp - c0;
		}
		ZSTAMP(3); // decode
		// ---- has the position left the upper chunk?  (at most 44 bytes a round: by one chunk)
		if (go && bp > 0) {
			const uint32_t cn = (ap + (uint32_t) ((bp - 1) >> 3)) & ~(HD_CH - 1u);
			if (cn != cc) {
				chunk_to_ring(cc - 2 * HD_CH, pre);
				cc -= HD_CH;
				have_pre = false;
			}
		}
		// ---- the round's bytes out (behind the chunk: see above)
		if (go) {
			uint8_t *o = out + done;
#pragma unroll
			for (int g = 0; g < (int) HD_SYMS / 16; g++) {
				if (cnt >= 16u * g + 16u) {
					const uint4 v = make_uint4(all[4 * g], all[4 * g + 1], all[4 * g + 2], all[4 * g + 3]);
					__builtin_memcpy(o + 16 * g, &v, 16);
				} else if (cnt > 16u * g) {
#pragma unroll
					for (int j = 0; j < 16; j++)
						if (16u * g + j < cnt)
							o[16 * g + j] = (uint8_t) (all[4 * g + (j >> 2)] >> (8 * (j & 3)));
				}
			}
		}
		ZSTAMP(4); // ring, stores
	}
	ZSTAMP_FLUSH();
	if (active)
		ok = bp == 0 && !overrun; // the stream ends exactly here
	{
		const unsigned long long bad = __ballot(!ok);
		if ((bad & 0xFFFFFFFFull) && lane == 0)
			z.rd[un.read].mode = 2;
		if ((bad >> 32) && lane == 32)
			z.rd[un.read].mode = 2;
	}
}

// Blocks with four LONG streams - ZSTD_compress's own frames, the reference's VBZ streams: 128-KiB blocks, 32 768 codes a
// stream, and a lane per stream was a chain of 32 768 dependent look-ups (2.5 ms for 2048 reads, most lanes of the chip
// without work).  Here ZSEG lanes share a stream: Huffman codes synchronise themselves - a decoder
// started at ANY bit falls into step with the true code boundaries after a few codes - so lane j starts at bit B (ZSEG - j) / ZSEG
// (lane 0: at the stream's true start), runs down to where lane j + 1 started and notes where it got out; then every lane
// runs its segment again from where the lane above got out, until no start moves any more (as a rule: once).  The codes of
// the segments, counted on the way, give every lane its place in the output; a last run writes the bytes.  Same ring, same
// step as k_zs_hdecode; a step is done under "not yet at the segment's end".
#ifndef ZSEG_N
#define ZSEG_N 32
#endif
constexpr int ZSEG = ZSEG_N;       // lanes per stream.  2048 / 1024 reads of ZSTD_compress's frames, whole depress call: 16 (a wave takes a
                                   // block's four streams) 2.17 / 1.46 ms, 32 (two waves a block) 2.02 / 1.32, 64: 2.07 / 1.39
constexpr int ZSPW = 64 / ZSEG;    // streams per wave
__device__ __forceinline__ void hd_long(const DecodeArgs &a, const ZsBufs &z, uint16_t *dt, HdRings &ring, uint32_t first, uint32_t stride)
{
	const uint32_t total = z.dctl->nlong < z.cap_long ? z.dctl->nlong : z.cap_long;
	const int lane = threadIdx.x;
	typedef __attribute__((address_space(3))) const uint16_t *lds_cu16p;
	const uint32_t dtb = (uint32_t) (uintptr_t) (lds_cu16p) dt;
	uint32_t *myring = ring[lane];
	for (uint32_t bi = first; bi < total * (4 / ZSPW); bi += stride) {
		if (bi != first)
			__syncthreads(); // the table of the block before is done with
		const ZsLong L = z.dlong[bi / (4 / ZSPW)];
		const uint32_t q0 = (bi % (4 / ZSPW)) * ZSPW; // the wave's first stream of the block
		const bool tab = L.tree < z.cap_trees && hd_build_table(dt, z.dtree + L.tree, lane);
		__syncthreads();
		const uint32_t qw = (uint32_t) lane / ZSEG, q = q0 + qw, j = lane & (ZSEG - 1); // stream in the wave, in the block; segment
		const uint8_t *p = a.in + L.h.src;
		uint8_t *out = z.ztmp + L.h.dst;
		uint32_t len = 0, k = 0;
		bool ok = tab;
		{
			const uint32_t s1 = p[0] | (p[1] << 8), s2 = p[2] | (p[3] << 8), s3 = p[4] | (p[5] << 8);
			const uint32_t seg = (L.h.R + 3) / 4;
			if (6ull + s1 + s2 + s3 >= L.h.cs || 3 * seg > L.h.R) {
				ok = false;
			} else {
				const uint32_t sz[4] = { s1, s2, s3, L.h.cs - 6 - s1 - s2 - s3 };
				p += 6;
				for (uint32_t i = 0; i < q; i++)
					p += sz[i];
				len = sz[q];
				k = q < 3 ? seg : L.h.R - 3 * seg;
				out += (uint64_t) q * seg;
			}
			if (ok && (len == 0 || p[len - 1] == 0))
				ok = false;
		}
		const int32_t B = ok ? (int32_t) (8 * (len - 1)) + (31 - __builtin_clz((uint32_t) p[len - 1])) : 0; // the stream's bits
		const uint32_t ap = (uint32_t) (uintptr_t) p & 0x07FFFFFFu;
		const int32_t c0 = (int32_t) (8 * ap) - 12;
		int32_t flo = 0, fhi = (int32_t) len; // the frame's bytes, as offsets in the stream (k_zs_hdecode)
		if (ok && L.read < a.nreads) {
			const int64_t f0 = (int64_t) a.in_off[L.read] - (int64_t) (p - a.in);
			const int64_t f1 = f0 + (int64_t) a.in_len[L.read];
			if (f0 <= 0 && f1 >= (int64_t) len && f0 > -0x40000000ll && f1 < 0x40000000ll) {
				flo = (int32_t) f0;
				fhi = (int32_t) f1;
			}
		}
		auto piece = [&](uint32_t ca) -> uint4 {
			const int32_t o = (int32_t) (ca - ap);
			uint4 v = make_uint4(0, 0, 0, 0);
			if (o >= flo && o + 16 <= fhi) {
				__builtin_memcpy(&v, p + o, 16);
			} else if (o > -16 && o < (int32_t) len) {
				uint32_t x[4] = { 0, 0, 0, 0 };
				for (int e = 0; e < 16; e++)
					if (o + e >= 0 && o + e < (int32_t) len)
						x[e >> 2] |= (uint32_t) p[o + e] << (8 * (e & 3));
				v = make_uint4(x[0], x[1], x[2], x[3]);
			}
			return v;
		};
		auto chunk_to_ring = [&](uint32_t ca, const uint4 *c4) {
			uint32_t *d = myring + ((ca >> 2) & (HD_RD / 2));
#pragma unroll
			for (int i = 0; i < HD_PC; i++) {
				d[4 * i] = c4[i].x;
				d[4 * i + 1] = c4[i].y;
				d[4 * i + 2] = c4[i].z;
				d[4 * i + 3] = c4[i].w;
			}
		};
		// the lane's segment: from `start` (bits of the stream not yet used) down to `lim`
		const int32_t lim = (int32_t) (((int64_t) B * (ZSEG - 1 - (int) j)) / ZSEG);
		int32_t start = j ? (int32_t) (((int64_t) B * (ZSEG - (int) j)) / ZSEG) : B;
		int32_t exit_bp = start;  // where the lane's last code ended (<= lim)
		uint32_t ncodes = 0;      // codes of the segment
		bool over = false;        // the run wanted bits below the stream's first
		// one run of the lanes with `on`; with `o`: the bytes are written there
		auto run = [&](bool on, uint8_t *o) {
			int32_t sp = start + c0;
			const int32_t lsp = lim + c0;
			uint32_t cc = (ap + (uint32_t) ((start > 0 ? start - 1 : 0) >> 3)) & ~(HD_CH - 1u);
			if (on) {
				uint4 c4[HD_PC];
#pragma unroll
				for (int i = 0; i < HD_PC; i++)
					c4[i] = piece(cc + 16 * i);
				chunk_to_ring(cc, c4);
#pragma unroll
				for (int i = 0; i < HD_PC; i++)
					c4[i] = piece(cc - HD_CH + 16 * i);
				chunk_to_ring(cc - HD_CH, c4);
			}
			uint32_t w0 = (uint32_t) sp >> 5;
			uint32_t lo = myring[w0 & (HD_RD - 1u)], hi = myring[(w0 + 1) & (HD_RD - 1u)];
			uint4 pre[HD_PC];
#pragma unroll
			for (int i = 0; i < HD_PC; i++)
				pre[i] = make_uint4(0, 0, 0, 0);
			bool have_pre = false;
			uint32_t n = 0, used = 0;
			while (__any(on && sp > lsp)) {
				const bool go = on && sp > lsp;
				if (go && !have_pre) {
					const int32_t oo = (int32_t) (cc - 2 * HD_CH - ap);
					if (oo >= flo && oo + (int32_t) HD_CH <= fhi) {
#pragma unroll
						for (int i = 0; i < HD_PC; i++)
							__builtin_memcpy(&pre[i], p + oo + 16 * i, 16);
					} else {
#pragma unroll
						for (int i = 0; i < HD_PC; i++)
							pre[i] = piece(cc - 2 * HD_CH + 16 * i);
					}
					have_pre = true;
				}
				uint32_t all[HD_SYMS / 4];
#pragma unroll
				for (int i = 0; i < (int) HD_SYMS / 4; i++)
					all[i] = 0;
				uint32_t c = 0;
#pragma unroll
				for (int i = 0; i < (int) HD_SYMS; i++) {
					if (go && sp > lsp) {
						const uint32_t below = myring[(w0 - 1) & (HD_RD - 1u)];
						const uint32_t y = __builtin_amdgcn_alignbit(hi, lo, (uint32_t) sp);
						const uint32_t e = *(lds_cu16p) (uintptr_t) ((y & 0xFFEu) | dtb);
						const uint32_t nb = e >> 8;
						used += nb;
						sp = max(sp - (int32_t) nb, c0);
						const uint32_t wn = (uint32_t) sp >> 5;
						if (wn != w0) {
							hi = lo;
							lo = below;
						}
						w0 = wn;
						all[i >> 2] |= (e & 0xFFu) << (8 * (i & 3));
						c++;
					}
				}
				if (go) {
					const int32_t bp = sp - c0;
					if (bp > 0) {
						const uint32_t cn = (ap + (uint32_t) ((bp - 1) >> 3)) & ~(HD_CH - 1u);
						if (cn != cc) {
							chunk_to_ring(cc - 2 * HD_CH, pre);
							cc -= HD_CH;
							have_pre = false;
						}
					}
					if (o) {
#pragma unroll
						for (int g = 0; g < (int) HD_SYMS / 16; g++) {
							if (c >= 16u * g + 16u) {
								const uint4 v = make_uint4(all[4 * g], all[4 * g + 1], all[4 * g + 2], all[4 * g + 3]);
								__builtin_memcpy(o + n + 16 * g, &v, 16);
							} else if (c > 16u * g) {
#pragma unroll
								for (int i = 0; i < 16; i++)
									if (16u * g + i < c)
										o[n + 16 * g + i] = (uint8_t) (all[4 * g + (i >> 2)] >> (8 * (i & 3)));
							}
						}
					}
					n += c;
				}
			}
			if (on) {
				exit_bp = sp - c0;
				ncodes = n;
				over = used > (uint32_t) (start - exit_bp); // (the clamp at bit 0 held the position back)
			}
		};
		run(ok, nullptr);
		for (int it = 0; it < ZSEG; it++) { // (every round fixes at least one more lane's start)
			const int32_t above = __shfl_up(exit_bp, 1);
			const int32_t want = j ? above : B;
			const bool redo = ok && want != start;
			if (!__any(redo))
				break;
			if (redo)
				start = want;
			run(redo, nullptr);
		}
		// ---- the codes in front of the lane's, in its stream; the stream's codes must be k and its last code end at bit 0
		const uint32_t inc = wave_incl32(ok ? ncodes : 0u, lane);
		const uint32_t excl = inc - (ok ? ncodes : 0u);
		const uint32_t base = (uint32_t) __shfl((int) excl, (int) (qw * ZSEG));
		const uint32_t sum = (uint32_t) __shfl((int) inc, (int) (qw * ZSEG + ZSEG - 1)) - base;
		const int32_t last_exit = __shfl(exit_bp, (int) (qw * ZSEG + ZSEG - 1));
		const int32_t above = __shfl_up(exit_bp, 1);
		bool good = ok && sum == k && last_exit == 0 && !over && (j ? above : B) == start;
		// (one lane's verdict is its stream's: every lane of the stream must agree)
		{
			const unsigned long long g64 = __ballot(good);
			const unsigned long long mask = (ZSEG == 64 ? ~0ull : (1ull << (ZSEG & 63)) - 1ull) << (qw * (ZSEG & 63));
			good = (g64 & mask) == mask;
		}
		run(good, out + (excl - base));
		if (!good && j == 0 && L.read < a.nreads)
			z.rd[L.read].mode = 2;
	}
}

// One launch for both kinds of work, the long blocks' workgroups first (they run longest): a lane-per-stream wave and a
// 16-lanes-per-stream wave are each a chain of look-ups with most of the chip's issue slots free - side by side they
// take as long as the longer of the two (ZSTD_compress's frames: 0.70 + 0.59 ms one after the other).
__global__ __launch_bounds__(64) void k_zs_hdecode(DecodeArgs a, ZsBufs z, uint32_t nlong_wgs)
{
	__shared__ __attribute__((aligned(4096))) HdTables dt2;
	__shared__ HdRings ring;
	if (blockIdx.x < nlong_wgs)
		hd_long(a, z, dt2[0], ring, blockIdx.x, nlong_wgs);
	else
		hd_units(a, z, dt2, ring, blockIdx.x - nlong_wgs);
}

// Blocks with sequences (libzstd's own frames: the reference's streams, press.c:1462-1469): one wave
// per frame carries them out in order - ll literals from the literals space, then ml bytes from off
// bytes back in the content, which may be bytes the same wave has just written (a match may even
// overlap itself): the wave waits for its stores before the next copy and reads the match source past
// the CU's L1.  Runs behind k_zs_copy / k_zs_hdecode (literals and the other blocks are in place).
__device__ __forceinline__ void wave_copy_plain(uint8_t *d, const uint8_t *s, uint32_t n)
{
	const uint32_t lane = threadIdx.x & 63;
	for (uint32_t k = lane * 16; k < n; k += 64 * 16) {
		if (k + 16 <= n) {
			uint4 v;
			__builtin_memcpy(&v, s + k, 16);
			__builtin_memcpy(d + k, &v, 16);
		} else {
			for (uint32_t e = k; e < n; e++)
				d[e] = s[e];
		}
	}
}
__device__ __forceinline__ uint8_t ld_l2(const uint8_t *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wave_match(uint8_t *d, uint32_t off, uint32_t n)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint8_t *s = d - off;
	if (off >= n) { // no overlap
		for (uint32_t k = lane * 8; k < n; k += 64 * 8) {
			if (k + 8 <= n) {
				uint64_t v = 0;
				for (int b = 0; b < 8; b++)
					v |= (uint64_t) ld_l2(s + k + b) << (8 * b);
				__builtin_memcpy(d + k, &v, 8);
			} else {
				for (uint32_t e = k; e < n; e++)
					d[e] = ld_l2(s + e);
			}
		}
		return;
	}
	// the match overlaps itself: the off bytes in front of it, repeated
	if (off == 1) {
		const uint32_t v4 = ld_l2(s) * 0x01010101u;
		for (uint32_t k = lane * 16; k < n; k += 64 * 16) {
			if (k + 16 <= n) {
				const uint4 v = make_uint4(v4, v4, v4, v4);
				__builtin_memcpy(d + k, &v, 16);
			} else {
				for (uint32_t e = k; e < n; e++)
					d[e] = (uint8_t) v4;
			}
		}
		return;
	}
	for (uint32_t k = lane; k < n; k += 64)
		d[k] = ld_l2(s + k % off);
}

__global__ __launch_bounds__(256) void k_zs_exec(DecodeArgs a, ZsBufs z)
{
	const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= a.nreads)
		return;
	uint32_t b = z.rd[r].mode == 0 ? z.rd[r].pad[0] : 0;
	while (b) {
		const ZsXBlk x = z.dxblk[b - 1];
		const uint8_t *lit = z.ztmp + x.lit;
		uint8_t *out = z.ztmp + x.dst;
		for (uint32_t i = 0; i < x.nseq; i++) {
			const ZsSeq q = z.dseq[x.seq0 + i];
			wave_copy_plain(out, lit, q.ll);
			lit += q.ll;
			out += q.ll;
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the bytes written so far have reached L2
			wave_match(out, q.off, q.ml);
			out += q.ml;
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		wave_copy_plain(out, lit, x.tail);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		b = x.next;
	}
}

__global__ __launch_bounds__(256) void k_zs_finish(DecodeArgs a, ZsBufs z)
{
	const uint32_t r = blockIdx.x * 256 + threadIdx.x;
	if (r >= a.nreads)
		return;
	const ZsRead rd = z.rd[r];
	const uint32_t cap_n = a.nsamp[r];
	uint32_t n = cap_n ? cap_n : 1;
	uint64_t len = 0; // too short for any read: the svb-zd decode reports the failure
	if (!z.kdiv) { // ex-zd carries its own sample count; n[r] is the room
		n = cap_n;
		len = rd.mode == 0 ? rd.nd : 0;
	} else if (rd.mode == 0 && rd.nd >= 4) {
		const uint8_t *S = z.ztmp + z.zoff[r];
		const uint32_t cnt = (uint32_t) S[0] | ((uint32_t) S[1] << 8) | ((uint32_t) S[2] << 16) | ((uint32_t) S[3] << 24);
		if (cnt <= cap_n) { // press.c:1901: the count in the stream is what gets decoded
			n = cnt;
			len = rd.nd - 4;
		}
	}
	z.zn[r] = n;
	z.zlen[r] = len;
}

} // namespace

void launch_zstd_decode_frames(const DecodeArgs &a, const ZsBufs &z, hipStream_t s)
{
	if (!a.nreads)
		return;
	(void) hipMemsetAsync(z.dctl, 0, sizeof(ZsDCtl), s);
	hipLaunchKernelGGL(k_zs_layout, dim3(1), dim3(1024), 0, s, a.nsamp, a.nreads, z.zoff, z.zoff4, z.kdiv);
	hipLaunchKernelGGL(k_zs_walk<true>, dim3((a.nreads + 3) / 4), dim3(256), 0, s, a, z);
	hipLaunchKernelGGL(k_zs_walk<false>, dim3((a.nreads + 3) / 4), dim3(256), 0, s, a, z);
	hipLaunchKernelGGL(k_zs_copy, dim3(z.cap_copy < 8192 ? z.cap_copy : 8192), dim3(256), 0, s, a, z);
	ktime_begin(1, s);
	const uint32_t nlong_wgs = z.cap_long < 4096 ? z.cap_long : 4096;
	hipLaunchKernelGGL(k_zs_hdecode, dim3(nlong_wgs + (z.cap_units + 1) / 2), dim3(64), 0, s, a, z, nlong_wgs);
	ktime_end(1, s);
	hipLaunchKernelGGL(k_zs_exec, dim3((a.nreads + 3) / 4), dim3(256), 0, s, a, z);
}

void launch_zstd_decode_streams(const DecodeArgs &a, const ZsBufs &z, hipStream_t s)
{
	if (!a.nreads)
		return;
	hipLaunchKernelGGL(k_zs_finish, dim3((a.nreads + 255) / 256), dim3(256), 0, s, a, z);
	DecodeArgs sv = a; // the inner streams are in ztmp (svb: behind their counts)
	sv.in = z.ztmp;
	sv.in_off = z.kdiv ? z.zoff4 : z.zoff;
	sv.in_len = z.zlen;
	sv.nsamp = z.zn;
	ktime_mute(true);
	if (z.kdiv)
		launch_svb_decode_chunked(sv, z.kdiv == 4, true, s);
	else
		launch_ex_decode_chunked(sv, EXF_EXZD, 0, s);
	ktime_mute(false);
}

} // namespace ph

#ifdef HUF_STAMPS
extern "C" int press_hip_zs_table_stamps(unsigned long long *dst)
{
	return hipDeviceSynchronize() == hipSuccess && hipMemcpyFromSymbol(dst, HIP_SYMBOL(ph::g_tstamp), 64) == hipSuccess ? 0 : -1;
}
extern "C" int press_hip_zs_walk_stamps(unsigned long long *dst)
{
	return hipDeviceSynchronize() == hipSuccess && hipMemcpyFromSymbol(dst, HIP_SYMBOL(ph::g_wstamp), 64) == hipSuccess ? 0 : -1;
}
extern "C" int press_hip_zs_walk_ticks(uint32_t *dst)
{
	return hipDeviceSynchronize() == hipSuccess && hipMemcpyFromSymbol(dst, HIP_SYMBOL(ph::g_walk_ticks), 8192 * 4) == hipSuccess ? 0 : -1;
}
extern "C" int press_hip_zs_stamps(unsigned long long *dst)
{
	unsigned long long z[8] = { 0 };
	if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(dst, HIP_SYMBOL(ph::g_zstamp), sizeof z) != hipSuccess ||
	    hipMemcpyToSymbol(HIP_SYMBOL(ph::g_zstamp), z, sizeof z) != hipSuccess)
		return -1;
	return 0;
}
#endif
