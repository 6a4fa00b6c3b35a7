// chaindp_io.hip -- zero-copy movement between pinned host packet buffers and HBM.
//
// The packet shim's reads sit in separate pinned (hipHostMalloc, device-visible) buffers.  Issuing one
// hipMemcpyAsync per read costs ~5 us each (thousands per batch), and copying results once more on the host
// costs a core; instead one kernel per direction moves all reads of a batch: each workgroup streams one read
// between its host buffer and its CSR slot in HBM with 16-byte (anchors) / 8-byte (24-byte records) lanes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "chaindp_kernels.h"

namespace chaindp {

// d_a[off[r] .. off[r+1]) <- src[r][0 .. n_r)      (src[r]: device-visible host pointer)
__global__ __launch_bounds__(256) void k_gather_reads(int64_t n_reads, const int64_t *__restrict__ off,
                                                      const ulonglong2 *const *__restrict__ src, ulonglong2 *__restrict__ d_a)
{
	for (int64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
		const int64_t o = off[r], n = off[r + 1] - o;
		const ulonglong2 *s = src[r];
		for (int64_t k = threadIdx.x; k < n; k += blockDim.x) d_a[o + k] = s[k];
	}
}

// dst[r][0 .. n_r*3) (8-byte words) <- seeds[seeds_off[r] .. seeds_off[r+1]); then zero up to the next 64-byte boundary.
// The destination (a slot of a pinned result packet, 64-byte aligned) is written in 16-byte stores: the link takes them better than
// 8-byte ones (the kernel trace of a packet replay showed this mover alone on 83 % of the wall clock at 46 GB/s).  The source is
// 8-byte aligned only (24-byte records), so a 16-byte store is fed by two 8-byte loads.
typedef unsigned long long io_u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void scatter_words16(unsigned long long *__restrict__ d, const unsigned long long *__restrict__ s, int64_t nw)
{
	const int64_t n16 = (((nw * 8 + 63) & ~(int64_t)63) >> 4);                                // 16-byte words incl. the zero padding
	for (int64_t k = threadIdx.x; k < n16; k += blockDim.x) {
		io_u64x2 t;
		t.x = 2 * k < nw ? s[2 * k] : 0ull;
		t.y = 2 * k + 1 < nw ? s[2 * k + 1] : 0ull;
		__builtin_nontemporal_store(t, (io_u64x2*)d + k);
	}
}

__global__ __launch_bounds__(256) void k_scatter_seeds(int64_t n_reads, const int64_t *__restrict__ seeds_off,
                                                       unsigned long long *const *__restrict__ dst,
                                                       const unsigned long long *__restrict__ seeds)
{
	for (int64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
		unsigned long long *d = dst[r];
		if (!d) continue;
		scatter_words16(d, seeds + seeds_off[r] * 3, (seeds_off[r + 1] - seeds_off[r]) * 3);   // 24 B = 3 words
	}
}

// dst[r][0 .. n_r) <- words[woff[r] .. woff[r+1]) (8-byte words, e.g. mini_pos[]); then zero up to the next 64-byte boundary
__global__ __launch_bounds__(256) void k_scatter_words(int64_t n_reads, const int64_t *__restrict__ woff,
                                                       unsigned long long *const *__restrict__ dst,
                                                       const unsigned long long *__restrict__ words)
{
	for (int64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
		unsigned long long *d = dst[r];
		if (!d) continue;
		scatter_words16(d, words + woff[r], woff[r + 1] - woff[r]);
	}
}

// dst[0 .. n16) <- src[0 .. n16) in 16-byte words; dst is device-visible pinned host memory.  The streaming pipe downloads with
// this kernel instead of a DMA copy: on this platform a download and an upload issued as two DMA copies take turns, while a
// shader that stores over PCIe runs beside the upload's DMA engine (measured: 56.7 -> 3x ms per 76 M-anchor batch).
typedef uint32_t io_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy_out(int64_t n16, const io_u32x4 *__restrict__ src, io_u32x4 *__restrict__ dst)
{
	for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (int64_t)gridDim.x * blockDim.x)
		__builtin_nontemporal_store(src[k], &dst[k]);
}

// Workgroups of the per-read movers.  They are PCIe-bound: a workgroup per read (thousands) fills every wave slot of the chip with
// waves that wait for the link, and the DP kernels of the other contexts of a pipeline cannot start beside them; 64 workgroups of
// 256 threads keep ~260 KB in flight, which covers the link's bandwidth-delay product several times over.
static int io_blocks(int64_t n_reads)
{
	static const int cap = getenv("CHAINDP_IO_BLOCKS") ? atoi(getenv("CHAINDP_IO_BLOCKS")) : 64;     // (read once; tuning only)
	const int64_t c = cap > 0 ? cap : 64;
	return (int)(n_reads < c ? n_reads : c);
}

hipError_t launch_copy_out(hipStream_t st, void *h_dst, const void *d_src, size_t bytes, int blocks)
{
	const int64_t n16 = (int64_t)((bytes + 15) / 16);
	if (n16 <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_copy_out, dim3((unsigned)blocks), dim3(256), 0, st, n16, (const io_u32x4*)d_src, (io_u32x4*)h_dst);
	return hipGetLastError();
}

hipError_t launch_scatter_words(hipStream_t st, int64_t n_reads, const int64_t *d_woff, void *const *d_dst, const void *d_words)
{
	if (n_reads <= 0) return hipSuccess;
	const int64_t blocks = io_blocks(n_reads);
	hipLaunchKernelGGL(k_scatter_words, dim3((unsigned)blocks), dim3(256), 0, st, n_reads, d_woff,
	                   (unsigned long long *const *)d_dst, (const unsigned long long*)d_words);
	return hipGetLastError();
}

hipError_t launch_gather_reads(hipStream_t st, int64_t n_reads, const int64_t *d_off, const void *const *d_src, void *d_a)
{
	if (n_reads <= 0) return hipSuccess;
	const int64_t blocks = io_blocks(n_reads);
	hipLaunchKernelGGL(k_gather_reads, dim3((unsigned)blocks), dim3(256), 0, st, n_reads, d_off,
	                   (const ulonglong2 *const *)d_src, (ulonglong2*)d_a);
	return hipGetLastError();
}

hipError_t launch_scatter_seeds(hipStream_t st, int64_t n_reads, const int64_t *d_seeds_off, void *const *d_dst, const void *d_seeds)
{
	if (n_reads <= 0) return hipSuccess;
	const int64_t blocks = io_blocks(n_reads);
	hipLaunchKernelGGL(k_scatter_seeds, dim3((unsigned)blocks), dim3(256), 0, st, n_reads, d_seeds_off,
	                   (unsigned long long *const *)d_dst, (const unsigned long long*)d_seeds);
	return hipGetLastError();
}

} // namespace chaindp
