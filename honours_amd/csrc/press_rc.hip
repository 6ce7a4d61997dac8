// press_rc.hip - the adaptive binary range coders of rc_vbe21_zd, rcc_vbe21_zd and
// rccm_vbbe21_zd (8f-4).  First the order-0 coder: TurboRC's rcsenc / rcsdec as the reference calls them
// (press.c:5456, 5489) on the one-byte values of the vbe21 exception split.
//
// The format leaves a GPU nothing to parallelise inside a read: the 255 adaptive probabilities
// and the 64-bit interval run through the whole stream, bit by bit (~1 M dependent steps for a
// mean NA12878 read), and nothing marks a position where a second coder could start.  So this
// is ONE READ PER LANE: a wave codes 64 reads side by side, its probability tables in LDS
// (256 x 64 dwords, the lane is the bank - conflict free whatever the contexts), and a batch
// lasts as long as its longest read.  It is here for coverage of the reference's entropy
// stages, bit-exact like everything else; its throughput is what the format allows
// (DESIGN.md section 6).
//
// Coder (restated from its behaviour, pinned against the compiled reference by
// oracle/press_oracle.c:po_rcs_encode and tests/test_oracle_golden.py):
//   state   range = 2^64 - 1, low = 0; output in 32-bit little-endian words;
//   bit b, probability p of a ONE (15 bits, context = bits of the byte coded so far):
//           x = (range >> 15) * p;  b ? range = x : (range -= x, low += x); a carry out of low
//           increments the words already written;  b ? p += ceil((32768-p)/32) - 1 : p -= p >> 5;
//   renormalisation (range < 2^32: emit the top word of low, shift both by 32) only before
//           bits 7, 5, 3, 1 of a byte;
//   end     renormalise; range > 2^33 ? (low += 2^32, one word) : (low += 1, two words);
//   give-up (rcutil_.h:161) once the output reaches n*255/256 - 8 bytes the input is stored raw.

#include "press_internal.h"

namespace ph {

namespace {

constexpr uint64_t RC_TOP = 1ull << 32;

struct RcOut {
	uint8_t *out;
	uint64_t cap;  // bytes the stream may take
	uint64_t pos;  // bytes written
	bool failed;   // the slot is too small
};

__device__ __forceinline__ void rc_put32(RcOut &o, uint32_t w)
{
	if (o.pos + 4 > o.cap) {
		o.failed = true;
	} else {
		__builtin_memcpy(o.out + o.pos, &w, 4);
	}
	o.pos += 4;
}

__device__ __forceinline__ void rc_carry(RcOut &o)
{
	uint64_t q = o.pos;
	while (q >= 4 && !o.failed) {
		q -= 4;
		uint32_t w;
		__builtin_memcpy(&w, o.out + q, 4);
		w += 1;
		__builtin_memcpy(o.out + q, &w, 4);
		if (w)
			break;
	}
}

} // namespace

// one read per lane; the one-byte values of read r are a.low_tmp[off[r] ..) (k_low_encode_chunked)
__global__ __launch_bounds__(64) void k_rcs_encode(BatchArgs a)
{
	__shared__ uint32_t mb[256][64];
	const uint32_t lane = threadIdx.x;
	const uint32_t r = blockIdx.x * 64 + lane;
	for (int c = 0; c < 256; c++)
		mb[c][lane] = 1u << 14;
	bool alive = r < a.nreads;
	const ReadMeta *m = a.meta + (alive ? r : 0);
	if (alive && m->status)
		alive = false; // out_len = FAILED was written by k_ex_section
	const uint64_t n = alive ? m->nlow : 0;
	const uint32_t head = alive ? m->hdr + m->seclen : 0;
	const uint8_t *in = a.low_tmp + (alive ? a.off[r] : 0);
	RcOut o;
	o.out = a.out + (alive ? a.out_off[r] + head : 0);
	o.cap = alive ? a.out_off[r + 1] - a.out_off[r] - head : 0;
	o.pos = 0;
	o.failed = false;
	uint64_t low = 0, range = ~0ull;
	const long long giveup = (long long) (n * 255 / 256) - 8;
	bool raw = false;
	// the input is fetched 16 bytes at a time, one fetch ahead: a load per byte would put an L1
	// round trip (~0.1 us) on every byte's critical path.  (Reads past n stay inside the buffer's slack.)
	uint4 cur = make_uint4(0, 0, 0, 0), nxt = make_uint4(0, 0, 0, 0);
	if (alive && n)
		__builtin_memcpy(&nxt, in, 16);
	for (uint64_t i = 0;; i++) {
		const bool act = alive && !raw && i < n;
		if (!__any(act))
			break;
		if ((i & 15) == 0) { // i is wave uniform
			cur = nxt;
			if (alive && i + 16 < n)
				__builtin_memcpy(&nxt, in + i + 16, 16);
		}
		if (act) {
			const uint32_t sel = (uint32_t) (i >> 2) & 3u;
			const uint32_t dw = sel == 0 ? cur.x : sel == 1 ? cur.y : sel == 2 ? cur.z : cur.w;
			const uint32_t x = 0x100u | ((dw >> (8 * ((uint32_t) i & 3u))) & 0xFFu);
			// the eight contexts of a byte are known in advance (and distinct): fetch their
			// probabilities together, so that the dependent chain is arithmetic only
			uint32_t pp[8];
#pragma unroll
			for (int k = 7; k >= 0; k--)
				pp[k] = mb[x >> (k + 1)][lane];
#pragma unroll
			for (int k = 7; k >= 0; k--) {
				if ((k & 1) && range < RC_TOP) {
					range <<= 32;
					rc_put32(o, (uint32_t) (low >> 32));
					low <<= 32;
				}
				const uint32_t p = pp[k];
				const uint64_t t = (range >> 15) * p, before = low;
				if ((x >> k) & 1u) {
					range = t;
					mb[x >> (k + 1)][lane] = p + (32768u - p + 31u) / 32u - 1u;
				} else {
					range -= t;
					low += t;
					mb[x >> (k + 1)][lane] = p - (p >> 5);
				}
				if (before > low)
					rc_carry(o);
			}
			if ((long long) o.pos >= giveup)
				raw = true; // rcutil_.h:161: stored instead
		}
	}
	if (alive) {
		if (raw) {
			o.failed = n > o.cap;
			if (!o.failed)
				for (uint64_t i = 0; i < n; i++)
					o.out[i] = in[i];
			o.pos = n;
		} else {
			if (range < RC_TOP) {
				range <<= 32;
				rc_put32(o, (uint32_t) (low >> 32));
				low <<= 32;
			}
			const uint64_t before = low;
			if (range > (1ull << 33)) {
				low += 1ull << 32;
				if (before > low)
					rc_carry(o);
				rc_put32(o, (uint32_t) (low >> 32));
			} else {
				low += 1;
				if (before > low)
					rc_carry(o);
				rc_put32(o, (uint32_t) (low >> 32));
				rc_put32(o, (uint32_t) low);
			}
		}
		a.out_len[r] = o.failed ? ~0ull : (uint64_t) head + o.pos;
	}
}

// one read per lane: the rc stream behind the exception section -> a.low[off[r] ..)
__global__ __launch_bounds__(64) void k_rcs_decode(DecodeArgs a)
{
	__shared__ uint32_t mb[256][64];
	const uint32_t lane = threadIdx.x;
	const uint32_t r = blockIdx.x * 64 + lane;
	for (int c = 0; c < 256; c++)
		mb[c][lane] = 1u << 14;
	bool alive = r < a.nreads;
	const ReadMeta *m = a.meta + (alive ? r : 0);
	if (alive && m->status)
		alive = false;
	const uint64_t n = alive ? m->nlow : 0;
	const uint32_t head = alive ? m->hdr + m->seclen : 0;
	const uint8_t *in = a.in + (alive ? a.in_off[r] + head : 0);
	const uint64_t len = alive ? a.in_len[r] - head : 0;
	uint8_t *out = a.low + (alive ? a.off[r] : 0);
	uint64_t pos = 0, range = ~0ull, code = 0;
	// bytes past the end of the stream read as zeros (the reference reads whatever follows)
	auto get32 = [&]() -> uint32_t {
		uint32_t w = 0;
		if (pos + 4 <= len) {
			__builtin_memcpy(&w, in + pos, 4);
		} else {
			for (int b = 0; b < 4; b++)
				if (pos + b < len)
					w |= (uint32_t) in[pos + b] << (8 * b);
		}
		pos += 4;
		return w;
	};
	// the next word is always in flight before it is needed
	uint32_t wn = 0;
	auto next32 = [&]() -> uint32_t {
		const uint32_t w = wn;
		wn = get32();
		return w;
	};
	if (alive) {
		wn = get32();
		code = next32();
		code = (code << 32) | next32();
	}
	for (uint64_t i = 0;; i++) {
		const bool act = alive && i < n;
		if (!__any(act))
			break;
		if (act) {
			uint32_t x = 1;
			uint32_t p = mb[1][lane];
#pragma unroll
			for (int k = 7; k >= 0; k--) {
				// both candidates for the next context while this bit is being decided
				uint32_t c0 = 0, c1 = 0;
				if (k) {
					c0 = mb[2 * x][lane];
					c1 = mb[2 * x + 1][lane];
				}
				if ((k & 1) && range < RC_TOP) {
					range <<= 32;
					code = (code << 32) | next32();
				}
				const uint64_t t = (range >> 15) * p;
				if (code < t) {
					range = t;
					mb[x][lane] = p + (32768u - p + 31u) / 32u - 1u;
					x = 2 * x + 1;
					p = c1;
				} else {
					range -= t;
					code -= t;
					mb[x][lane] = p - (p >> 5);
					x = 2 * x;
					p = c0;
				}
			}
			out[i] = (uint8_t) x;
		}
	}
}

// ------------------------------------------------------------------ order 1: rcc_vbe21_zd
//
// rccsenc / rccsdec (rc_.c:181-205; press.c:5510-5580): the same bit coder, the 255 probabilities of a
// byte chosen by the byte in front of it - 256 x 256 adaptive 16-bit predictors = 128 KiB of state per
// read, which fits nowhere but a whole CU's LDS.  So: ONE READ PER WORKGROUP (one wave), the table in
// LDS, reads handed out by a ticket to as many workgroups as there are CUs.  Every lane runs the same
// interval arithmetic (uniform values); encoding, the eight predictors of a byte - their contexts are
// known in advance - are fetched and updated by eight lanes at once, so the dependent chain is
// arithmetic only; decoding, the wave fetches the whole row in front of a byte (four predictors per lane) and
// the eight dependent steps pick theirs with v_readlane (the interval arithmetic runs on the scalar unit).
// What the format leaves: 256 reads in flight on the chip, each at one byte per few hundred cycles.

namespace {
constexpr uint32_t RCC_TAB = 65536; // predictors
constexpr uint32_t RCC_WG = 64;

__device__ __forceinline__ void rcc_init(uint16_t *mb)
{
	uint4 *m4 = reinterpret_cast<uint4 *>(mb);
	const uint32_t v = (1u << 14) | (1u << 30);
	for (uint32_t i = threadIdx.x; i < RCC_TAB / 8; i += RCC_WG)
		m4[i] = make_uint4(v, v, v, v);
	__syncthreads();
}
} // namespace

__global__ __launch_bounds__(RCC_WG) void k_rcc_encode(BatchArgs a)
{
	__shared__ __attribute__((aligned(16))) uint16_t mb[RCC_TAB];
	__shared__ uint32_t s_r;
	const uint32_t lane = threadIdx.x;
	for (;;) {
		if (lane == 0)
			s_r = atomicAdd(&a.ctl->ticket2, 1u);
		__syncthreads();
		const uint32_t r = s_r;
		__syncthreads();
		if (r >= a.nreads)
			return;
		const ReadMeta *m = a.meta + r;
		if (m->status)
			continue; // out_len = FAILED was written by k_ex_section
		rcc_init(mb);
		const uint64_t n = m->nlow;
		const uint32_t head = m->hdr + m->seclen;
		const uint8_t *in = a.low_tmp + a.off[r];
		RcOut o;
		o.out = a.out + a.out_off[r] + head;
		o.cap = a.out_off[r + 1] - a.out_off[r] - head;
		o.pos = 0;
		o.failed = false;
		uint64_t low = 0, range = ~0ull;
		const long long giveup = (long long) (n * 255 / 256) - 8;
		bool raw = false;
		uint32_t cx = 0;
		uint32_t grp = 0; // 64 input bytes, one per lane
		for (uint64_t i = 0; i < n && !raw; i++) {
			if ((i & 63) == 0)
				grp = i + lane < n ? in[i + lane] : 0u;
			const uint32_t byte = (uint32_t) __builtin_amdgcn_readlane((int) grp, __builtin_amdgcn_readfirstlane((int) (i & 63)));
			const uint32_t x = 0x100u | byte;
			uint16_t *row = mb + cx * 256u;
			// lane k < 8: the predictor of bit k
			const uint32_t node = x >> ((lane & 7u) + 1u);
			const uint32_t pk = row[node];
#pragma unroll
			for (int k = 7; k >= 0; k--) {
				if ((k & 1) && range < RC_TOP) {
					range <<= 32;
					if (lane == 0)
						rc_put32(o, (uint32_t) (low >> 32));
					else
						o.pos += 4;
					low <<= 32;
				}
				const uint32_t p = (uint32_t) __builtin_amdgcn_readlane((int) pk, k);
				const uint64_t t = (range >> 15) * p, before = low;
				if ((x >> k) & 1u) {
					range = t;
				} else {
					range -= t;
					low += t;
				}
				if (before > low && lane == 0)
					rc_carry(o);
			}
			if (lane < 8) {
				const uint32_t bit = (x >> lane) & 1u;
				row[node] = (uint16_t) (bit ? pk + (32768u - pk + 31u) / 32u - 1u : pk - (pk >> 5));
			}
			cx = byte;
			if ((long long) o.pos >= giveup)
				raw = true; // rcutil_.h:161: stored instead
		}
		const bool failed0 = (bool) __shfl((int) o.failed, 0, 64);
		if (raw) {
			const bool fail = n > o.cap;
			if (!fail)
				for (uint64_t i = lane; i < n; i += RCC_WG)
					o.out[i] = in[i];
			if (lane == 0)
				a.out_len[r] = fail ? ~0ull : (uint64_t) head + n;
		} else if (lane == 0) {
			o.failed = failed0;
			if (range < RC_TOP) {
				range <<= 32;
				rc_put32(o, (uint32_t) (low >> 32));
				low <<= 32;
			}
			const uint64_t before = low;
			if (range > (1ull << 33)) {
				low += 1ull << 32;
				if (before > low)
					rc_carry(o);
				rc_put32(o, (uint32_t) (low >> 32));
			} else {
				low += 1;
				if (before > low)
					rc_carry(o);
				rc_put32(o, (uint32_t) (low >> 32));
				rc_put32(o, (uint32_t) low);
			}
			a.out_len[r] = o.failed ? ~0ull : (uint64_t) head + o.pos;
		}
		__syncthreads(); // the table is free again
	}
}

__global__ __launch_bounds__(RCC_WG) void k_rcc_decode(DecodeArgs a)
{
	__shared__ __attribute__((aligned(16))) uint16_t mb[RCC_TAB];
	__shared__ uint32_t s_r;
	const uint32_t lane = threadIdx.x;
	for (;;) {
		if (lane == 0)
			s_r = atomicAdd(&a.ctl->ticket2, 1u);
		__syncthreads();
		const uint32_t r = s_r;
		__syncthreads();
		if (r >= a.nreads)
			return;
		const ReadMeta *m = a.meta + r;
		if (m->status)
			continue;
		rcc_init(mb);
		const uint64_t n = m->nlow;
		const uint32_t head = m->hdr + m->seclen;
		const uint8_t *in = a.in + a.in_off[r] + head;
		const uint64_t len = a.in_len[r] - head;
		uint8_t *out = a.low + a.off[r];
		uint64_t pos = 0, range = ~0ull, code = 0;
		// bytes past the end of the stream read as zeros (the reference reads whatever follows)
		auto get32 = [&]() -> uint32_t {
			uint32_t w = 0;
			if (pos + 4 <= len) {
				__builtin_memcpy(&w, in + pos, 4);
			} else {
				for (int b = 0; b < 4; b++)
					if (pos + b < len)
						w |= (uint32_t) in[pos + b] << (8 * b);
			}
			pos += 4;
			return w;
		};
		uint32_t wn = get32(); // the next word is always in flight before it is needed
		// (readfirstlane: every lane holds the same word - said so, the interval arithmetic runs on the scalar unit)
		auto next32 = [&]() -> uint32_t {
			const uint32_t w = (uint32_t) __builtin_amdgcn_readfirstlane((int) wn);
			wn = get32();
			return w;
		};
		code = next32();
		code = (code << 32) | next32();
		uint16_t *row = mb;
		uint32_t acc = 0; // four decoded bytes
		for (uint64_t i = 0; i < n; i++) {
			// nothing but the eight nodes on a byte's path changes while it is decoded: the whole row (255
			// probabilities, four per lane in one 8-byte LDS read) is fetched in front of the byte and the
			// eight dependent steps pick theirs with v_readlane - no memory access on the chain
			uint2 pq;
			__builtin_memcpy(&pq, __builtin_assume_aligned(row + 4u * lane, 8), 8);
			uint32_t x = 1;
#pragma unroll
			for (int k = 7; k >= 0; k--) {
				const int sl = __builtin_amdgcn_readfirstlane((int) (x >> 2));
				const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) pq.x, sl);
				const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) pq.y, sl);
				const uint32_t pair = (x & 2u) ? hi : lo;
				const uint32_t p = (x & 1u) ? pair >> 16 : pair & 0xFFFFu;
				if ((k & 1) && range < RC_TOP) {
					range <<= 32;
					code = (code << 32) | next32();
				}
				const uint64_t t = (range >> 15) * p;
				if (code < t) {
					range = t;
					x = 2 * x + 1;
				} else {
					range -= t;
					code -= t;
					x = 2 * x;
				}
			}
			if (lane < 8) { // the eight nodes of the path move towards their bits (as in the encoder)
				const uint32_t node = x >> (lane + 1u), bit = (x >> lane) & 1u;
				const uint32_t pk = row[node];
				row[node] = (uint16_t) (bit ? pk + (32768u - pk + 31u) / 32u - 1u : pk - (pk >> 5));
			}
			acc |= (x & 0xFFu) << (8 * ((uint32_t) i & 3u));
			if (((uint32_t) i & 3u) == 3u || i + 1 == n) {
				if (lane == 0)
					for (uint32_t b = 0; b <= ((uint32_t) i & 3u); b++)
						out[(i & ~3ull) + b] = (uint8_t) (acc >> (8 * b));
				acc = 0;
			}
			row = mb + 256u * (x & 0xFFu); // rc_.c:199: the byte just decoded is the next context
			// (lane 0's stores to the table and every lane's later loads are DS operations of one wave: in order)
		}
		__syncthreads();
	}
}

// ------------------------------------------------------------------ order 1-0 context mixing: rccm_vbbe21_zd
//
// rcmsenc / rcmsdec (Turbo-Range-Coder rccm_.c:79-121 as rccm_s.c:46-48 instantiates it: 16-bit
// probabilities, simple predictor with the rate as parameter; press.c:6946, 6987).  The same interval
// arithmetic, renormalised in front of EVERY bit; the probability of a bit is a mix (mbc.h:185-193):
//   p0 = order-0 predictor of the bit's node, p1 = the node's predictor under the byte in front
//   p  = (p0 + 15 p1) / 16
//   s  = value at p of the node's 17-point curve (linear between the two points around p)
//   P(bit = 1) = (p + 3 s) / 4
// and then p0 moves towards the bit by 1/4, p1 by 1/16, the two points of the curve by 1/64.
// State per read: 256 + 256 x 256 predictors and 256 x 17 curve points = 137 KiB: ONE READ PER WORKGROUP
// like the order-1 coder above, everything in LDS.  Encoding, the eight nodes of a byte are known in
// advance and distinct (so are their curves): eight lanes fetch, mix and update them side by side and the
// dependent chain is the interval arithmetic alone.  Decoding, the wave mixes the probabilities of ALL 255
// nodes in front of every byte (four nodes per lane) and the eight dependent steps pick theirs with a
// v_readlane: no memory access on the chain either (first version, one LDS round trip per bit for the
// curve: 9.9 s for the 8192-read batch).
// (restated from its behaviour, pinned against the compiled reference by oracle/press_oracle.c:
// po_rcms_encode / po_rcms_decode and tests/test_oracle_golden.py)

namespace {
struct RcmLds {
	uint16_t mb1[65536];
	uint16_t mb0[256];
	uint16_t sse[256][17];
};

__device__ __forceinline__ void rcm_init(RcmLds &L)
{
	uint4 *m4 = reinterpret_cast<uint4 *>(L.mb1);
	const uint32_t v = (1u << 15) | (1u << 31);
	for (uint32_t i = threadIdx.x; i < 65536 / 8; i += RCC_WG)
		m4[i] = make_uint4(v, v, v, v);
	for (uint32_t i = threadIdx.x; i < 256; i += RCC_WG)
		L.mb0[i] = 1u << 15;
	for (uint32_t i = threadIdx.x; i < 256 * 17; i += RCC_WG) { // rccm_s.c:40-44
		const uint32_t k = i % 17;
		(&L.sse[0][0])[i] = (uint16_t) ((k << 12) - (k == 16));
	}
	__syncthreads();
}

// mbc_s.h:40 with the bit as a 64-bit unsigned (turborc_.h:421): the low 16 bits of
// p - (((p - (bit ? 65536 : 0)) >> rate) + bit); for bit = 1 that is p + ceil((65536 - p) / 2^rate) - 1
__device__ __forceinline__ uint32_t rcm_step(uint32_t p, uint32_t rate, uint32_t bit)
{
	return bit ? p + ((65536u - p + (1u << rate) - 1u) >> rate) - 1u : p - (p >> rate);
}

// P(bit = 1) out of the two predictions and the two points of the curve around their mix
__device__ __forceinline__ uint32_t rcm_mix(uint32_t p, uint32_t x1, uint32_t x2)
{
	const int32_t sp = (int32_t) x1 + ((((int32_t) x2 - (int32_t) x1) * (int32_t) (p & 4095u)) >> 12);
	return (uint32_t) (((int32_t) p + 3 * sp) >> 2);
}
} // namespace

__global__ __launch_bounds__(RCC_WG) void k_rcm_encode(BatchArgs a)
{
	__shared__ __attribute__((aligned(16))) RcmLds L;
	__shared__ uint32_t s_r;
	const uint32_t lane = threadIdx.x;
	for (;;) {
		if (lane == 0)
			s_r = atomicAdd(&a.ctl->ticket2, 1u);
		__syncthreads();
		const uint32_t r = s_r;
		__syncthreads();
		if (r >= a.nreads)
			return;
		const ReadMeta *m = a.meta + r;
		if (m->status)
			continue; // out_len = FAILED was written by k_ex_section
		rcm_init(L);
		const uint64_t n = m->nlow;
		const uint32_t head = m->hdr + m->seclen;
		const uint8_t *in = a.low_tmp + a.off[r];
		RcOut o;
		o.out = a.out + a.out_off[r] + head;
		o.cap = a.out_off[r + 1] - a.out_off[r] - head;
		o.pos = 0;
		o.failed = false;
		uint64_t low = 0, range = ~0ull;
		const long long giveup = (long long) (n * 255 / 256) - 8;
		bool raw = false;
		uint32_t cx = 0;
		uint32_t grp = 0; // 64 input bytes, one per lane
		for (uint64_t i = 0; i < n && !raw; i++) {
			if ((i & 63) == 0)
				grp = i + lane < n ? in[i + lane] : 0u;
			const uint32_t byte = (uint32_t) __builtin_amdgcn_readlane((int) grp, __builtin_amdgcn_readfirstlane((int) (i & 63)));
			const uint32_t x = 0x100u | byte;
			uint16_t *row = L.mb1 + cx * 256u;
			// lane k < 8: the node of bit k, its two predictions, its curve
			const uint32_t node = x >> ((lane & 7u) + 1u);
			const uint32_t p0 = L.mb0[node], p1 = row[node];
			const uint32_t p = (p0 + 15u * p1) >> 4;
			uint16_t *cell = &L.sse[node][p >> 12];
			const uint32_t x1 = cell[0], x2 = cell[1];
			const uint32_t pm = rcm_mix(p, x1, x2);
#pragma unroll
			for (int k = 7; k >= 0; k--) {
				if (range < RC_TOP) {
					range <<= 32;
					if (lane == 0)
						rc_put32(o, (uint32_t) (low >> 32));
					else
						o.pos += 4;
					low <<= 32;
				}
				const uint32_t pk = (uint32_t) __builtin_amdgcn_readlane((int) pm, k);
				const uint64_t t = (range >> 16) * pk, before = low;
				if ((x >> k) & 1u) {
					range = t;
				} else {
					range -= t;
					low += t;
				}
				if (before > low && lane == 0)
					rc_carry(o);
			}
			if (lane < 8) {
				const uint32_t bit = (x >> lane) & 1u;
				L.mb0[node] = (uint16_t) rcm_step(p0, 2, bit);
				row[node] = (uint16_t) rcm_step(p1, 4, bit);
				cell[0] = (uint16_t) rcm_step(x1, 6, bit);
				cell[1] = (uint16_t) rcm_step(x2, 6, bit);
			}
			cx = byte;
			if ((long long) o.pos >= giveup)
				raw = true; // rcutil_.h:161: stored instead
		}
		const bool failed0 = (bool) __shfl((int) o.failed, 0, 64);
		if (raw) {
			const bool fail = n > o.cap;
			if (!fail)
				for (uint64_t i = lane; i < n; i += RCC_WG)
					o.out[i] = in[i];
			if (lane == 0)
				a.out_len[r] = fail ? ~0ull : (uint64_t) head + n;
		} else if (lane == 0) {
			o.failed = failed0;
			if (range < RC_TOP) {
				range <<= 32;
				rc_put32(o, (uint32_t) (low >> 32));
				low <<= 32;
			}
			const uint64_t before = low;
			if (range > (1ull << 33)) {
				low += 1ull << 32;
				if (before > low)
					rc_carry(o);
				rc_put32(o, (uint32_t) (low >> 32));
			} else {
				low += 1;
				if (before > low)
					rc_carry(o);
				rc_put32(o, (uint32_t) (low >> 32));
				rc_put32(o, (uint32_t) low);
			}
			a.out_len[r] = o.failed ? ~0ull : (uint64_t) head + o.pos;
		}
		__syncthreads(); // the tables are free again
	}
}

__global__ __launch_bounds__(RCC_WG) void k_rcm_decode(DecodeArgs a)
{
	__shared__ __attribute__((aligned(16))) RcmLds L;
	__shared__ uint32_t s_r;
	const uint32_t lane = threadIdx.x;
	for (;;) {
		if (lane == 0)
			s_r = atomicAdd(&a.ctl->ticket2, 1u);
		__syncthreads();
		const uint32_t r = s_r;
		__syncthreads();
		if (r >= a.nreads)
			return;
		const ReadMeta *m = a.meta + r;
		if (m->status)
			continue;
		rcm_init(L);
		const uint64_t n = m->nlow;
		const uint32_t head = m->hdr + m->seclen;
		const uint8_t *in = a.in + a.in_off[r] + head;
		const uint64_t len = a.in_len[r] - head;
		uint8_t *out = a.low + a.off[r];
		uint64_t pos = 0, range = ~0ull, code = 0;
		// bytes past the end of the stream read as zeros (the reference reads whatever follows)
		auto get32 = [&]() -> uint32_t {
			uint32_t w = 0;
			if (pos + 4 <= len) {
				__builtin_memcpy(&w, in + pos, 4);
			} else {
				for (int b = 0; b < 4; b++)
					if (pos + b < len)
						w |= (uint32_t) in[pos + b] << (8 * b);
			}
			pos += 4;
			return w;
		};
		uint32_t wn = get32(); // the next word is always in flight before it is needed
		// (readfirstlane: every lane holds the same word - said so, the interval arithmetic runs on the scalar unit)
		auto next32 = [&]() -> uint32_t {
			const uint32_t w = (uint32_t) __builtin_amdgcn_readfirstlane((int) wn);
			wn = get32();
			return w;
		};
		code = next32();
		code = (code << 32) | next32();
		uint16_t *row = L.mb1;
		uint32_t acc = 0; // four decoded bytes
		for (uint64_t i = 0; i < n; i++) {
			// The nodes on a byte's path are distinct and nothing else changes while the byte is decoded: the
			// probabilities of ALL 255 nodes are known in front of it.  Lane l takes nodes l, l + 64, l + 128,
			// l + 192 (two LDS round trips for the lot), and the eight dependent steps fetch theirs with a
			// v_readlane - no memory access on the chain.
			uint32_t pm[4];
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const uint32_t node = lane + 64u * j;
				const uint32_t q0 = L.mb0[node], q1 = row[node];
				const uint32_t p = (q0 + 15u * q1) >> 4;
				const uint16_t *cell = &L.sse[node][p >> 12];
				pm[j] = rcm_mix(p, cell[0], cell[1]);
			}
			uint32_t x = 1;
#pragma unroll
			for (int k = 7; k >= 0; k--) {
				// depth 7 - k: nodes 2^(7-k) .. 2^(8-k) - 1
				const int sl = __builtin_amdgcn_readfirstlane((int) (x & 63u));
				uint32_t pk;
				if (k >= 2)
					pk = (uint32_t) __builtin_amdgcn_readlane((int) pm[0], sl);
				else if (k == 1)
					pk = (uint32_t) __builtin_amdgcn_readlane((int) pm[1], sl);
				else
					pk = (x & 64u) ? (uint32_t) __builtin_amdgcn_readlane((int) pm[3], sl)
						       : (uint32_t) __builtin_amdgcn_readlane((int) pm[2], sl);
				if (range < RC_TOP) {
					range <<= 32;
					code = (code << 32) | next32();
				}
				const uint64_t t = (range >> 16) * pk;
				const uint32_t bit = code < t ? 1u : 0u;
				if (bit) {
					range = t;
				} else {
					range -= t;
					code -= t;
				}
				x = 2 * x + bit;
			}
			if (lane < 8) { // the eight nodes of the path move towards their bits (as in the encoder)
				const uint32_t xx = x; // 0x100 | byte
				const uint32_t node = xx >> (lane + 1u), bit = (xx >> lane) & 1u;
				const uint32_t q0 = L.mb0[node], q1 = row[node];
				const uint32_t p = (q0 + 15u * q1) >> 4;
				uint16_t *cell = &L.sse[node][p >> 12];
				const uint32_t x1 = cell[0], x2 = cell[1];
				L.mb0[node] = (uint16_t) rcm_step(q0, 2, bit);
				row[node] = (uint16_t) rcm_step(q1, 4, bit);
				cell[0] = (uint16_t) rcm_step(x1, 6, bit);
				cell[1] = (uint16_t) rcm_step(x2, 6, bit);
			}
			acc |= (x & 0xFFu) << (8 * ((uint32_t) i & 3u));
			if (((uint32_t) i & 3u) == 3u || i + 1 == n) {
				if (lane == 0)
					for (uint32_t b = 0; b <= ((uint32_t) i & 3u); b++)
						out[(i & ~3ull) + b] = (uint8_t) (acc >> (8 * b));
				acc = 0;
			}
			row = L.mb1 + 256u * (x & 0xFFu); // the byte just decoded is the next context (rccm_.c:119)
			// (lane 0's stores to the tables and every lane's later loads are DS operations of one wave: in order)
		}
		__syncthreads();
	}
}

void launch_rcm_encode(const BatchArgs &a, hipStream_t s)
{
	(void) hipMemsetAsync(&a.ctl->ticket2, 0, 4, s);
	hipLaunchKernelGGL(k_rcm_encode, dim3(a.nreads < 256u ? a.nreads : 256u), dim3(RCC_WG), 0, s, a);
}

void launch_rcm_decode(const DecodeArgs &a, hipStream_t s)
{
	(void) hipMemsetAsync(&a.ctl->ticket2, 0, 4, s);
	hipLaunchKernelGGL(k_rcm_decode, dim3(a.nreads < 256u ? a.nreads : 256u), dim3(RCC_WG), 0, s, a);
}

void launch_rcc_encode(const BatchArgs &a, hipStream_t s)
{
	(void) hipMemsetAsync(&a.ctl->ticket2, 0, 4, s);
	hipLaunchKernelGGL(k_rcc_encode, dim3(a.nreads < 256u ? a.nreads : 256u), dim3(RCC_WG), 0, s, a);
}

void launch_rcc_decode(const DecodeArgs &a, hipStream_t s)
{
	(void) hipMemsetAsync(&a.ctl->ticket2, 0, 4, s);
	hipLaunchKernelGGL(k_rcc_decode, dim3(a.nreads < 256u ? a.nreads : 256u), dim3(RCC_WG), 0, s, a);
}

void launch_rcs_encode(const BatchArgs &a, hipStream_t s)
{
	hipLaunchKernelGGL(k_rcs_encode, dim3((a.nreads + 63) / 64), dim3(64), 0, s, a);
}

void launch_rcs_decode(const DecodeArgs &a, hipStream_t s)
{
	hipLaunchKernelGGL(k_rcs_decode, dim3((a.nreads + 63) / 64), dim3(64), 0, s, a);
}

} // namespace ph
