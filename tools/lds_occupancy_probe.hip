// lds_occupancy_probe.hip -- how many 64-thread workgroups (one wave each) a CU holds for a given dynamic LDS size and VGPR
// count: the allocation granularity of LDS decides whether k_chain_twin keeps 6 waves per SIMD when its LDS layout changes.
//   hipcc --offload-arch=gfx950 -O2 -o tools/lds_occupancy_probe tools/lds_occupancy_probe.hip && tools/lds_occupancy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int R> __global__ __launch_bounds__(64) void k_probe(int *out)
{
	extern __shared__ int lds[];
	int acc[R];
	for (int i = 0; i < R; ++i) acc[i] = lds[(threadIdx.x + i) & 1023] * (i + 1);
	__syncthreads();
	int s = 0;
	for (int i = 0; i < R; ++i) s += acc[i] * out[i];
	out[threadIdx.x] = s;
}
int main()
{
	const int sizes[] = {5120, 6144, 6304, 6400, 6656, 6752, 6816, 6826, 6912, 7168, 7328, 7680, 8192, 9216, 10240};
	hipDeviceProp_t pr;
	if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
	printf("{\"device\": \"%s\", \"cus\": %d, \"lds_per_cu\": %zu, \"lds_per_block_max\": %zu, \"blocks_per_cu_by_lds\": {", pr.gcnArchName, pr.multiProcessorCount,
	       (size_t)pr.maxSharedMemoryPerMultiProcessor, (size_t)pr.sharedMemPerBlock);
	for (size_t i = 0; i < sizeof(sizes) / sizeof(sizes[0]); ++i) {
		int nb = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_probe<8>, 64, sizes[i]) != hipSuccess) nb = -1;
		printf("%s\"%d\": %d", i ? ", " : "", sizes[i], nb);
	}
	printf("}}\n");
	return 0;
}
