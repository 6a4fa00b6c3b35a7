// chaindp_compact.hip -- f/p/v -> new_seed[] (reference chain.c:286-317) as parallel kernels.
//
// The reference compacts while it runs the recurrence, in anchor order:
//   at step k: if p[k] >= 0 and p[k] has not been emitted yet, emit p[k] first ("late" emission,
//   chain.c:287-303); then emit k itself iff v[k] >= min_sc || p[k] >= 0 (chain.c:304-316).
// An anchor i is emitted at its own step iff self(i) = v[i] >= min_sc || p[i] >= 0; otherwise it is
// emitted late, just before the FIRST later k with p[k] == i (if any).  Hence, per read:
//   first_child[i] = min{k : p[k] == i}                      (only needed where !self(i))
//   late(k)  = p[k] >= 0 && !self(p[k]) && first_child[p[k]] == k
//   count(k) = late(k) + self(k)   in {0,1,2};   pos = exclusive prefix sum of count over the read
//   id[p[k]] = pos[k] if late(k);  id[k] = pos[k] + late(k) if self(k)
//   record of anchor i = { a[i], (id[p[i]] << 2 or -4) | (v[i] >= min_sc) | (f[i] < v[i]) << 1, f[i] }
// (a late-emitted anchor has p < 0, so its record carries -4 | flags).  oracle/chain_oracle.c:co_compact
// is the sequential statement of the same thing; tests compare the bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "chaindp_kernels.h"

namespace chaindp {

#define NO_CHILD 0x7f7f7f7f   // hipMemsetAsync(0x7f) pattern; larger than any read-relative index in use

struct SeedRec { uint64_t x, y; int32_t p, f; };   // == struct new_seed (minimap.h:51-55)

__device__ __forceinline__ bool self_emit(int32_t vi, int32_t pi, int min_sc) { return vi >= min_sc || pi >= 0; }

// C2: first_child via atomicMin, one wave per read (grid-stride over reads)
__global__ __launch_bounds__(256) void k_first_child(Params par, int64_t n_reads, const int64_t *__restrict__ off,
                                                     const int32_t *__restrict__ p, const int32_t *__restrict__ v,
                                                     int32_t *__restrict__ first_child)
{
	const int lane = threadIdx.x & 63;
	const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
	for (int64_t r = wave0; r < n_reads; r += n_waves) {
		const int64_t rs = off[r], n = off[r + 1] - rs;
		for (int64_t k = lane; k < n; k += 64) {
			const int32_t q = p[rs + k];
			if (q >= 0 && !self_emit(v[rs + q], p[rs + q], par.min_sc))
				atomicMin(&first_child[rs + q], (int32_t)k);
		}
	}
}

// C3: per-read positions.  Writes id[] and new_i[r] (into seeds_off[r+1], scanned by k_scan_reads).
__global__ __launch_bounds__(256) void k_positions(Params par, int64_t n_reads, const int64_t *__restrict__ off,
                                                   const int32_t *__restrict__ p, const int32_t *__restrict__ v,
                                                   const int32_t *__restrict__ first_child, int32_t *__restrict__ id,
                                                   int64_t *__restrict__ seeds_off)
{
	const int lane = threadIdx.x & 63;
	const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
	for (int64_t r = wave0; r < n_reads; r += n_waves) {
		const int64_t rs = off[r], n = off[r + 1] - rs;
		int carry = 0;
		for (int64_t t0 = 0; t0 < n; t0 += 64) {
			const int64_t k = t0 + lane;
			bool late = false, self = false;
			int32_t q = -1;
			if (k < n) {
				q = p[rs + k];
				self = self_emit(v[rs + k], q, par.min_sc);
				late = q >= 0 && !self_emit(v[rs + q], p[rs + q], par.min_sc) && first_child[rs + q] == (int32_t)k;
			}
			const int c = (int)late + (int)self;
			// wave exclusive prefix sum of c (values 0..2): two ballots
			const uint64_t b0 = __builtin_amdgcn_ballot_w64(c & 1), b1 = __builtin_amdgcn_ballot_w64(c & 2);
			const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
			const int pos = carry + __builtin_popcountll(b0 & below) + 2 * __builtin_popcountll(b1 & below);
			if (late) id[rs + q] = pos;
			if (self) id[rs + k] = pos + (int)late;
			carry += __builtin_popcountll(b0) + 2 * __builtin_popcountll(b1);
		}
		if (lane == 0) seeds_off[r + 1] = carry;
	}
}

// exclusive scan of new_i over reads, in place on seeds_off[1..n_reads] (seeds_off[0] = 0); single block
__global__ __launch_bounds__(1024) void k_scan_reads(int64_t n_reads, int64_t *__restrict__ seeds_off)
{
	__shared__ int64_t part[1024];
	const int tid = threadIdx.x;
	const int64_t per = (n_reads + 1023) / 1024;
	const int64_t lo = (int64_t)tid * per, hi = lo + per < n_reads ? lo + per : n_reads;
	int64_t s = 0;
	for (int64_t r = lo; r < hi; ++r) s += seeds_off[r + 1];
	part[tid] = s;
	__syncthreads();
	if (tid == 0) {
		int64_t acc = 0;
		for (int k = 0; k < 1024; ++k) { const int64_t t = part[k]; part[k] = acc; acc += t; }
		seeds_off[0] = 0;
	}
	__syncthreads();
	int64_t acc = part[tid];
	for (int64_t r = lo; r < hi; ++r) { acc += seeds_off[r + 1]; seeds_off[r + 1] = acc; }
}

// C4: records
__global__ __launch_bounds__(256) void k_write_seeds(Params par, int64_t n_reads, const int64_t *__restrict__ off,
                                                     const ulonglong2 *__restrict__ a, const int32_t *__restrict__ f,
                                                     const int32_t *__restrict__ p, const int32_t *__restrict__ v,
                                                     const int32_t *__restrict__ first_child, const int32_t *__restrict__ id,
                                                     const int64_t *__restrict__ seeds_off, SeedRec *__restrict__ seeds)
{
	const int lane = threadIdx.x & 63;
	const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
	const int min_sc = par.min_sc;
	for (int64_t r = wave0; r < n_reads; r += n_waves) {
		const int64_t rs = off[r], n = off[r + 1] - rs;
		SeedRec *out = seeds + seeds_off[r];
		for (int64_t k = lane; k < n; k += 64) {
			const int32_t q = p[rs + k], vk = v[rs + k], fk = f[rs + k];
			if (!self_emit(vk, q, min_sc)) continue;
			const int32_t idk = id[rs + k];
			int32_t pfield = (int32_t)(0xfffffffcu);                                         // (-1)<<2
			if (q >= 0) {
				const int32_t vq = v[rs + q], pq = p[rs + q], fq = f[rs + q];
				if (!self_emit(vq, pq, min_sc) && first_child[rs + q] == (int32_t)k) {           // late emission of q, chain.c:292-302
					const ulonglong2 aq = a[rs + q];
					SeedRec rec;
					rec.x = aq.x; rec.y = aq.y; rec.f = fq;
					rec.p = (int32_t)(0xfffffffcu | (uint32_t)(vq >= min_sc) | ((uint32_t)(fq < vq) << 1));
					out[idk - 1] = rec;
				}
				pfield = (int32_t)((uint32_t)id[rs + q] << 2);                                   // chain.c:310
			}
			const ulonglong2 ak = a[rs + k];
			SeedRec rec;
			rec.x = ak.x; rec.y = ak.y; rec.f = fk;
			rec.p = pfield | (int32_t)(vk >= min_sc) | ((int32_t)(fk < vk) << 1);                // chain.c:313-314
			out[idk] = rec;
		}
	}
}

static inline unsigned grid_for_reads(int64_t n_reads)
{
	int64_t blocks = (n_reads + 3) / 4;
	if (blocks > 256 * 8 * 4) blocks = 256 * 8 * 4;
	return (unsigned)(blocks < 1 ? 1 : blocks);
}

hipError_t launch_compact(hipStream_t st, const Params &par, int64_t n_reads, int64_t total, const int64_t *d_off,
                          const void *d_a, const int32_t *d_f, const int32_t *d_p, const int32_t *d_v,
                          int32_t *d_first_child, int32_t *d_id, int64_t *d_seeds_off, void *d_seeds, void *)
{
	hipError_t e;
	if (n_reads <= 0) return hipMemsetAsync(d_seeds_off, 0, sizeof(int64_t), st);
	if (total > 0 && (e = hipMemsetAsync(d_first_child, 0x7f, (size_t)total * 4, st)) != hipSuccess) return e;
	const unsigned g = grid_for_reads(n_reads);
	hipLaunchKernelGGL(k_first_child, dim3(g), dim3(256), 0, st, par, n_reads, d_off, d_p, d_v, d_first_child);
	hipLaunchKernelGGL(k_positions, dim3(g), dim3(256), 0, st, par, n_reads, d_off, d_p, d_v, d_first_child, d_id, d_seeds_off);
	hipLaunchKernelGGL(k_scan_reads, dim3(1), dim3(1024), 0, st, n_reads, d_seeds_off);
	hipLaunchKernelGGL(k_write_seeds, dim3(g), dim3(256), 0, st, par, n_reads, d_off, (const ulonglong2*)d_a, d_f, d_p, d_v,
	                   d_first_child, d_id, d_seeds_off, (SeedRec*)d_seeds);
	return hipGetLastError();
}

} // namespace chaindp
