// lds_occupancy_probe.hip -- how many 64-thread workgroups (one wave each) a CU REALLY holds at once for a given dynamic LDS size:
// the allocation granularity of LDS decides whether k_chain_twin keeps 6 waves per SIMD when its LDS layout changes, and
// hipOccupancyMaxActiveBlocksPerMultiprocessor does not know it (it answers 24 blocks for 6816 bytes; the chip holds 21).
// Every workgroup stamps the constant-rate clock, waits ~1 ms (bounded) and stamps again; the workgroups that start within
// the first half millisecond are the ones that were resident together.
//   hipcc --offload-arch=gfx950 -O2 -o tools/lds_occupancy_probe tools/lds_occupancy_probe.hip && tools/lds_occupancy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(64) void k_probe(unsigned long long *t, unsigned long long hold)
{
	extern __shared__ int lds[];
	lds[threadIdx.x] = (int)blockIdx.x;
	__syncthreads();
	if (threadIdx.x == 0) {
		const unsigned long long t0 = wall_clock64();
		t[2 * blockIdx.x] = t0;
		while (wall_clock64() - t0 < hold) __builtin_amdgcn_s_sleep(32);      // bounded: every workgroup leaves after `hold` ticks
		t[2 * blockIdx.x + 1] = wall_clock64() + (unsigned long long)(lds[0] & 0);
	}
}
int main()
{
	const int sizes[] = {5120, 5124, 6304, 6400, 6404, 6656, 6816, 7680, 7684, 8192};
	hipDeviceProp_t pr;
	if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
	const int cus = pr.multiProcessorCount, blocks = cus * 40;
	int rate_khz = 100000;
	(void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
	const unsigned long long hold = (unsigned long long)rate_khz;           // 1 ms
	unsigned long long *d = nullptr;
	if (hipMalloc((void**)&d, (size_t)blocks * 16) != hipSuccess) return 1;
	std::vector<unsigned long long> h((size_t)blocks * 2);
	printf("{\"device\": \"%s\", \"cus\": %d, \"workgroups_of_64_threads_resident_per_cu\": {", pr.gcnArchName, cus);
	for (size_t i = 0; i < sizeof(sizes) / sizeof(sizes[0]); ++i) {
		int api = 0;
		(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k_probe, 64, sizes[i]);
		hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(64), sizes[i], 0, d, hold);
		if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
		unsigned long long t0 = ~0ull;
		for (int b = 0; b < blocks; ++b) t0 = std::min(t0, h[2 * (size_t)b]);
		int first = 0;
		for (int b = 0; b < blocks; ++b) if (h[2 * (size_t)b] - t0 < hold / 2) ++first;
		printf("%s\"%d\": {\"measured\": %.2f, \"occupancy_api\": %d}", i ? ", " : "", sizes[i], (double)first / cus, api);
	}
	printf("}}\n");
	(void)hipFree(d);
	return 0;
}
