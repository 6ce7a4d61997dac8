// How many workgroups of a given size and LDS allocation does a CU of this GPU hold at once?
// Every workgroup spins for a fixed time; 256 * m of them finish in one spin time only if m fit on a CU.
//   hipcc --offload-arch=gfx950 -O2 -o build/ubench_lds_occ tools/ubench_lds_occ.hip && build/ubench_lds_occ
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void spin(unsigned long long ticks, unsigned *sink)
{
	extern __shared__ unsigned lds[];
	lds[threadIdx.x] = threadIdx.x;
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks)
		;
	if (lds[threadIdx.x] == 0xFFFFFFFFu)
		*sink = 1;
}

template <int LDS, int THR, bool BAR = true>
__global__ __launch_bounds__(THR) void spin_static(unsigned long long ticks, unsigned *sink)
{
	__shared__ unsigned lds[LDS / 4];
	for (int i = threadIdx.x; i < LDS / 4; i += THR)
		lds[i] = i;
	if (BAR)
		__syncthreads();
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks)
		;
	if (lds[(threadIdx.x * 7) % (LDS / 4)] == 0xFFFFFFFFu)
		*sink = 1;
}

template <int LDS, int THR, bool BAR = true>
static void run_static(int m, unsigned long long ticks, unsigned *sink)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	int occ = -1;
	hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin_static<LDS, THR, BAR>, THR, 0);
	hipLaunchKernelGGL((spin_static<LDS, THR, BAR>), dim3(256 * m), dim3(THR), 0, 0, ticks, sink);
	hipDeviceSynchronize();
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL((spin_static<LDS, THR, BAR>), dim3(256 * m), dim3(THR), 0, 0, ticks, sink);
	hipEventRecord(e1, 0);
	hipDeviceSynchronize();
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	printf("static: threads %4d lds %6d barrier %d x%d per CU: %.2f ms, runtime says %d per CU (%s)\n", THR, LDS, (int) BAR, m, ms, occ, hipGetErrorString(hipGetLastError()));
}

int main()
{
	unsigned *sink;
	hipMalloc(&sink, 4);
	hipFuncSetAttribute((const void *) spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	const struct { int threads, lds, m; } cases[] = {
		{ 512, 49204, 3 }, { 512, 51200, 3 }, { 512, 53248, 3 }, { 512, 53840, 3 }, { 512, 54528, 3 },
		{ 512, 69920, 2 }, { 576, 73760, 2 }, { 640, 79392, 2 }, { 512, 73760, 2 }, { 512, 79392, 2 }, { 512, 81920, 2 },
		{ 640, 60000, 2 }, { 576, 60000, 2 }, { 768, 60000, 2 }, { 1024, 60000, 2 }, { 640, 30000, 3 }, { 576, 30000, 3 },
	};
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const unsigned long long ticks = 100000; // 1 ms at 100 MHz
	for (const auto &c : cases) {
		hipLaunchKernelGGL(spin, dim3(256 * c.m), dim3(c.threads), c.lds, 0, ticks, sink);
		hipDeviceSynchronize();
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(spin, dim3(256 * c.m), dim3(c.threads), c.lds, 0, ticks, sink);
		hipEventRecord(e1, 0);
		hipDeviceSynchronize();
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		printf("threads %4d lds %6d x%d per CU: %.2f ms (%s)\n", c.threads, c.lds, c.m, ms, hipGetErrorString(hipGetLastError()));
	}
	run_static<79392, 640>(2, ticks, sink);
	run_static<73760, 576>(2, ticks, sink);
	run_static<69920, 512>(2, ticks, sink);
	run_static<65536, 640>(2, ticks, sink);
	run_static<65024, 640>(2, ticks, sink);
	run_static<60000, 640>(2, ticks, sink);
	run_static<79392, 512>(2, ticks, sink);
	run_static<79392, 640, false>(2, ticks, sink);
	run_static<49204, 512>(3, ticks, sink);
	run_static<49204, 512, false>(3, ticks, sink);
	run_static<30000, 256>(5, ticks, sink);
	run_static<30000, 256>(8, ticks, sink);
	run_static<60000, 1024>(2, ticks, sink);
	run_static<40000, 768>(2, ticks, sink);
	run_static<40000, 384>(4, ticks, sink);
	run_static<40000, 320>(4, ticks, sink);
	run_static<50000, 320>(3, ticks, sink);
	run_static<33000, 512>(4, ticks, sink);
	return 0;
}
