// micro-benchmark: what streaming rates does an MI355X reach with the access shapes of the press kernels?
// read-only (pass A), copy (svb encode / decode move 2n in, ~n out or the reverse), with 1 / 4 / 8 16-byte loads
// in flight per lane, default and non-temporal policy, grid-stride and one-shot workgroups.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_stream.hip -o tools/bin/ubench_stream
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int K, bool NT, bool WRITE>
__global__ void stream(const u32x4 *in, u32x4 *out, uint32_t *sink, size_t n16)
{
	// a workgroup takes tiles of K x blockDim 16-byte elements, grid-stride over tiles
	const size_t tile = (size_t) K * blockDim.x;
	uint32_t acc = 0;
	for (size_t t0 = (size_t) blockIdx.x * tile; t0 < n16; t0 += (size_t) gridDim.x * tile) {
		u32x4 v[K];
#pragma unroll
		for (int k = 0; k < K; k++) {
			const size_t i = t0 + (size_t) k * blockDim.x + threadIdx.x;
			if (i < n16)
				v[k] = NT ? __builtin_nontemporal_load(in + i) : in[i];
			else
				v[k] = (u32x4){ 0, 0, 0, 0 };
		}
#pragma unroll
		for (int k = 0; k < K; k++) {
			const size_t i = t0 + (size_t) k * blockDim.x + threadIdx.x;
			if (WRITE) {
				if (i < n16) {
					if (NT)
						__builtin_nontemporal_store(v[k], out + i);
					else
						out[i] = v[k];
				}
			} else {
				acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
			}
		}
	}
	if (!WRITE && acc == 0x12345678u)
		sink[0] = acc;
}

template <int K, bool NT, bool WRITE>
static void run(const char *name, const u32x4 *a, u32x4 *b, uint32_t *sink, size_t n16, int wg, int grid)
{
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	float best = 1e9;
	for (int rep = 0; rep < 5; rep++) {
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL((stream<K, NT, WRITE>), dim3(grid), dim3(wg), 0, 0, a, b, sink, n16);
		CK(hipEventRecord(e1));
		CK(hipEventSynchronize(e1));
		float ms;
		CK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best)
			best = ms;
	}
	const double moved = (WRITE ? 2.0 : 1.0) * 16.0 * n16;
	printf("%-5s K=%d %s wg %4d grid %6d : %.3f ms  %7.1f GB/s\n", name, K, NT ? "nt " : "def", wg, grid, best, moved / best / 1e6);
}

int main()
{
	const size_t bytes = (size_t) 2 << 30;
	const size_t n16 = bytes / 16;
	u32x4 *a, *b;
	uint32_t *sink;
	CK(hipMalloc(&a, bytes));
	CK(hipMalloc(&b, bytes));
	CK(hipMalloc(&sink, 64));
	CK(hipMemset(a, 1, bytes));
	const int wgs[] = { 256, 1024 };
	for (int wi = 0; wi < 2; wi++) {
		const int wg = wgs[wi];
		const int grids[] = { 256 * (2048 / wg), 8 * 256 * (2048 / wg) / 2, 65536 };
		for (int gi = 0; gi < 3; gi++) {
			const int g = grids[gi];
			run<1, false, false>("read", a, b, sink, n16, wg, g);
			run<4, false, false>("read", a, b, sink, n16, wg, g);
			run<8, false, false>("read", a, b, sink, n16, wg, g);
			run<4, true, false>("read", a, b, sink, n16, wg, g);
			run<1, false, true>("copy", a, b, sink, n16, wg, g);
			run<4, false, true>("copy", a, b, sink, n16, wg, g);
			run<8, false, true>("copy", a, b, sink, n16, wg, g);
			run<4, true, true>("copy", a, b, sink, n16, wg, g);
		}
	}
	return 0;
}
