// zsframe_fuzz.cpp - TEST INFRASTRUCTURE ONLY.  The frame reader of the device (zs::walk_frame,
// zs::read_tree, the stream decoder - honours_amd/csrc/zs_table.h) on damaged frames, built
// for the host with AddressSanitizer + UBSan (GPU sanitizers are not available): whatever
// the bytes are, the reader stays inside the frame and inside the output it was given.
//   usage: zsframe_fuzz [iterations]     exit code 0 = no finding
#include "zsframe_model.cpp"
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd()
{
	rng_state ^= rng_state << 13;
	rng_state ^= rng_state >> 7;
	rng_state ^= rng_state << 17;
	return (uint32_t) (rng_state >> 16);
}

int main(int argc, char **argv)
{
	const int iters = argc > 1 ? atoi(argv[1]) : 20000;
	int decoded = 0, refused = 0, host = 0;
	for (int round = 0; round < 24; round++) {
		// a buffer shaped like the reference's: [u32 n][keys][data]
		const uint32_t n = round < 4 ? 1 + rnd() % 200 : 1000 + rnd() % 60000;
		const uint32_t nk = (n + 3) / 4;
		std::vector<uint8_t> S(4 + nk);
		memcpy(S.data(), &n, 4);
		const int alphabet = round % 3 == 0 ? 256 : 8 + rnd() % 120;
		for (uint32_t i = 0; i < n; i++) {
			const bool two = rnd() % 997 == 0;
			if (two)
				S[4 + i / 4] |= (uint8_t) (1u << (2 * (i & 3)));
			uint32_t v = rnd() % alphabet;
			v = v * v / alphabet; // skewed
			S.push_back((uint8_t) v);
			if (two)
				S.push_back((uint8_t) rnd());
		}
		std::vector<uint8_t> frame(S.size() + S.size() / 100 + 65536);
		const uint64_t flen = zsm_frame(S.data(), S.size(), frame.data(), frame.size());
		if (!flen)
			return fprintf(stderr, "frame\n"), 2;
		{ // the undamaged frame reads back
			std::vector<uint8_t> out(S.size());
			std::vector<uint8_t> exact(frame.begin(), frame.begin() + flen);
			const int64_t r = zsm_decode(exact.data(), flen, out.data(), out.size());
			if (r != (int64_t) S.size() || memcmp(out.data(), S.data(), S.size()))
				return fprintf(stderr, "round trip (round %d)\n", round), 3;
		}
		for (int it = 0; it < iters / 24; it++) {
			// exact-size heap copies: ASan sees any access one byte outside
			uint64_t len = flen;
			const int kind = rnd() % 8;
			if (kind == 0)
				len = rnd() % (flen + 1); // truncated
			std::vector<uint8_t> f(frame.begin(), frame.begin() + len);
			if (kind == 1)
				for (int e = 0; e < 1 + (int) (rnd() % 40); e++)
					f.push_back((uint8_t) rnd()); // trailing bytes
			const int flips = kind < 2 ? 0 : 1 + rnd() % 4;
			for (int e = 0; e < flips && !f.empty(); e++) {
				// headers live in the first bytes and at block starts: damage those more often
				const uint64_t at = rnd() % 3 == 0 ? rnd() % (f.size() < 64 ? f.size() : 64) : rnd() % f.size();
				f[at] = rnd() % 2 ? (uint8_t) rnd() : (uint8_t) (f[at] ^ (1u << (rnd() % 8)));
			}
			const uint64_t cap = rnd() % 4 == 0 ? rnd() % (S.size() + 1) : S.size();
			std::vector<uint8_t> out(cap);
			const int64_t r = zsm_decode(f.data(), f.size(), out.data(), cap);
			if (r > (int64_t) cap)
				return fprintf(stderr, "content larger than the room\n"), 4;
			decoded += r >= 0;
			refused += r == -1;
			host += r == -2;
		}
	}
	// ---- frames WITH sequences (their literals go to the literals slot of `cap` bytes)
	{ // forged: RLE literals of 131072 bytes, one sequence, in a frame for a ten-sample read
		const uint8_t forged[] = { 0x28, 0xB5, 0x2F, 0xFD, 0x20, 24, /* block: last, compressed, 9 bytes */ 0x4D, 0x00, 0x00,
					   /* literals: RLE, 3-byte header, R = 131072 */ 0x0D, 0x00, 0x20, 0xAA,
					   /* one sequence, predefined tables, stream */ 0x01, 0x00, 0x01, 0x00, 0x80 };
		for (uint64_t cap = 0; cap < 200; cap += 7) {
			std::vector<uint8_t> f(forged, forged + sizeof forged), out(cap);
			const int64_t r = zsm_decode(f.data(), f.size(), out.data(), cap);
			if (r >= 0)
				return fprintf(stderr, "forged literals accepted\n"), 5;
			refused++;
		}
	}
	// libzstd's own frames (levels 1, 3, 9, 19) of repetitive buffers, damaged: only when a libzstd is around
	{
		void *h = dlopen("libzstd.so.1", RTLD_NOW);
		if (!h)
			h = dlopen("/opt/conda/lib/libzstd.so.1", RTLD_NOW);
		typedef size_t (*comp_t)(void *, size_t, const void *, size_t, int);
		comp_t comp = h ? (comp_t) dlsym(h, "ZSTD_compress") : nullptr;
		typedef unsigned (*iserr_t)(size_t);
		iserr_t iserr = h ? (iserr_t) dlsym(h, "ZSTD_isError") : nullptr;
		int with_seq = 0;
		for (int round = 0; comp && iserr && round < 16; round++) {
			const uint32_t n = 2000 + rnd() % 150000;
			std::vector<uint8_t> S(n);
			const uint32_t period = 3 + rnd() % 500;
			for (uint32_t i = 0; i < n; i++)
				S[i] = (i >= period && rnd() % 16) ? S[i - period] : (uint8_t) (rnd() % 23);
			std::vector<uint8_t> frame(n + n / 50 + 1024);
			const int level = round % 4 == 0 ? 1 : round % 4 == 1 ? 3 : round % 4 == 2 ? 9 : 19;
			const size_t flen = comp(frame.data(), frame.size(), S.data(), n, level);
			if (iserr(flen))
				return fprintf(stderr, "ZSTD_compress\n"), 6;
			{
				std::vector<uint8_t> out(n), exact(frame.begin(), frame.begin() + flen);
				const int64_t r = zsm_decode(exact.data(), flen, out.data(), n);
				if (r == zs::W_HOST)
					continue; // (a frame the reader leaves to libzstd)
				if (r != (int64_t) n || memcmp(out.data(), S.data(), n))
					return fprintf(stderr, "libzstd frame (level %d) does not read back\n", level), 7;
				with_seq++;
			}
			for (int it = 0; it < iters / 40; it++) {
				uint64_t len = flen;
				const int kind = rnd() % 8;
				if (kind == 0)
					len = rnd() % (flen + 1);
				std::vector<uint8_t> f(frame.begin(), frame.begin() + len);
				const int flips = kind < 1 ? 0 : 1 + rnd() % 4;
				for (int e = 0; e < flips && !f.empty(); e++) {
					const uint64_t at = rnd() % 3 == 0 ? rnd() % (f.size() < 64 ? f.size() : 64) : rnd() % f.size();
					f[at] = rnd() % 2 ? (uint8_t) rnd() : (uint8_t) (f[at] ^ (1u << (rnd() % 8)));
				}
				const uint64_t cap = rnd() % 4 == 0 ? rnd() % (n + 1) : n;
				std::vector<uint8_t> out(cap);
				const int64_t r = zsm_decode(f.data(), f.size(), out.data(), cap);
				if (r > (int64_t) cap)
					return fprintf(stderr, "content larger than the room\n"), 4;
				decoded += r >= 0;
				refused += r == -1;
				host += r == -2;
			}
		}
		printf("zsframe_fuzz: %d libzstd frames with sequences damaged\n", with_seq);
	}
	printf("zsframe_fuzz: %d decoded, %d refused, %d left to libzstd - no finding\n", decoded, refused, host);
	return 0;
}
