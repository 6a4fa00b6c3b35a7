// chaindp_prepass.hip -- everything that runs before the chain DP kernel: the per-block read table, the
// anchor-parallel prepass (q_span sums, unit starts, singletons), the unit list in longest-first order, and the
// per-read gap-cost table.  See chaindp_kernels.hip for the overall scheme.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "chaindp_kernels.h"
#include "chaindp_wave.h"

namespace chaindp {

// ---------------------------------------------------------------- K0: prepass (anchor-parallel, no hot atomics)
// k_prepass:    one thread per anchor, PRE_PER_BLOCK consecutive anchors of the batch per block.  The
//               block finds the reads its range touches by binary search in off[]; each thread derives
//               its unit-start flag and whether it is a singleton (resolved on the spot), zeroes its
//               global mark, and the block writes: a 64-bit unit-start mask per wave-tile, its unit and
//               singleton counts, and ONE integer atomic per (block, read) for the q_span sum
//               (order-independent, so deterministic).
// launch_scan_u64: exclusive scan of the per-block (units | singletons << 32) counts -> counters[0].
// k_emit_units: one thread per mask word; writes the Unit records in anchor order (deterministic).
// A single same-address atomic per wave would cap this stage at ~90 atomics/us (measured: 9 ms for
// 76 M anchors), hence count -> scan -> emit.

#define PRE_BLOCK 256
#define PRE_PER_BLOCK CHAINDP_BLOCK_ANCHORS
#define PRE_WORDS (PRE_PER_BLOCK / 64)

// largest r in [lo, hi] with off[r] <= g   (off is non-decreasing; empty reads are skipped over)
__device__ __forceinline__ int64_t read_of(const int64_t *__restrict__ off, int64_t lo, int64_t hi, int64_t g)
{
	if (hi - lo == 1) return g >= off[hi] ? hi : lo;               // a block that straddles one read boundary: the usual case
	while (lo < hi) {
		const int64_t mid = (lo + hi + 1) >> 1;
		if (off[mid] <= g) lo = mid; else hi = mid - 1;
	}
	return lo;
}

// Reads that the first and the last anchor of every 1024-anchor block belong to.  The anchor-parallel kernels
// of the prepass and of the compaction all cut the batch into the same blocks; a per-block binary search by one
// thread (28 dependent loads before the block can start) was most of their run time.
__global__ __launch_bounds__(256) void k_block_reads(int64_t n_reads, int64_t total, const int64_t *__restrict__ off, int2 *__restrict__ block_reads)
{
	const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t g0 = b * PRE_PER_BLOCK;
	if (g0 >= total) return;
	const int64_t g1 = g0 + PRE_PER_BLOCK < total ? g0 + PRE_PER_BLOCK : total;
	const int64_t rlo = read_of(off, 0, n_reads - 1, g0);
	block_reads[b] = make_int2((int)rlo, (int)read_of(off, rlo, n_reads - 1, g1 - 1));
}

__global__ __launch_bounds__(PRE_BLOCK) void k_prepass(Params par, int64_t n_reads, int64_t total,
                                                       const int64_t *__restrict__ off, const ulonglong2 *__restrict__ a,
                                                       unsigned long long *__restrict__ sumq, uint64_t *__restrict__ start_mask,
                                                       unsigned long long *__restrict__ block_cnt,
                                                       int32_t *__restrict__ f, int32_t *__restrict__ p, int32_t *__restrict__ v,
                                                       uint8_t *__restrict__ flags,
                                                       const int2 *__restrict__ block_reads)
{
	__shared__ unsigned int s_sum, s_units, s_singles;
	const int lane = threadIdx.x & 63;
	const uint64_t maxx = (uint64_t)(int64_t)par.max_dist_x;
	const int64_t g0 = (int64_t)blockIdx.x * PRE_PER_BLOCK;
	const int64_t g1 = g0 + PRE_PER_BLOCK < total ? g0 + PRE_PER_BLOCK : total;
	if (threadIdx.x == 0) { s_sum = 0; s_units = 0; s_singles = 0; }
	__syncthreads();
	const int2 rr = block_reads[blockIdx.x];                       // reads of the block's first and last anchor (k_block_reads)
	const int64_t rlo = rr.x, rhi = rr.y;
	const bool one_read = rlo == rhi;
	unsigned int w_sum = 0, w_units = 0, w_singles = 0;
	// all loads of the block's four passes are issued before the first is used (one anchor per thread and pass would
	// leave a single 16-byte load in flight per thread); neighbours come from the adjacent lanes, and from memory
	// only at the two ends of a wave's 64 anchors
	constexpr int PASSES = PRE_PER_BLOCK / PRE_BLOCK;
	ulonglong2 an_[PASSES];
	uint64_t xb_[PASSES], xe_[PASSES];
#pragma unroll
	for (int k = 0; k < PASSES; ++k) {
		const int64_t g = g0 + (int64_t)k * PRE_BLOCK + threadIdx.x;
		an_[k] = make_ulonglong2(0, 0); xb_[k] = 0; xe_[k] = 0;
		if (g < g1) an_[k] = a[g];
		if (g < g1 && lane == 0 && g > 0) xb_[k] = a[g - 1].x;
		if (g < g1 && (lane == 63 || g + 1 == g1) && g + 1 < total) xe_[k] = a[g + 1].x;
	}
	if (one_read) {
		// ---- the usual block: inside one read.  Same arithmetic as the general loop below with what is block-uniform hoisted (the
		// read's bounds), neighbours by DPP shifts instead of LDS permutes, the rare per-anchor events (a segment id, a zero q_span)
		// found by ballots, and the singleton stores through block-relative pointers: about half the instructions per anchor --
		// the kernel was issue-bound, not bandwidth-bound (80 vector + 60 scalar instructions per 64 anchors at 3 TB/s).
		const int64_t rs = off[rlo], re = off[rlo + 1];
		int32_t *const fb = f + g0, *const pb = p + g0, *const vb = v + g0;
		uint8_t *const flb = flags + g0;
		unsigned int any_seg = 0, any_span0 = 0;
#pragma unroll
		for (int k = 0; k < PASSES; ++k) {
			const int64_t gb = g0 + (int64_t)k * PRE_BLOCK;
			if (gb >= g1) break;
			const int t = k * PRE_BLOCK + (int)threadIdx.x;              // the anchor's place in the block
			const int64_t g = g0 + t;
			const bool have = g < g1;
			const ulonglong2 an = an_[k];
			const uint32_t xlo = (uint32_t)an.x, xhi = (uint32_t)(an.x >> 32);
			uint32_t plo = (uint32_t)wave_shift_up1((int)xlo, 0), phi = (uint32_t)wave_shift_up1((int)xhi, 0);
			uint32_t nlo = (uint32_t)dpp_or_old<DPP_WAVE_SHL1, 0xf>(0, (int)xlo), nhi = (uint32_t)dpp_or_old<DPP_WAVE_SHL1, 0xf>(0, (int)xhi);
			if (lane == 0) { plo = (uint32_t)xb_[k]; phi = (uint32_t)(xb_[k] >> 32); }
			if (lane == 63 || g + 1 == g1) { nlo = (uint32_t)xe_[k]; nhi = (uint32_t)(xe_[k] >> 32); }
			const uint64_t xprev = (uint64_t)phi << 32 | plo, xnext = (uint64_t)nhi << 32 | nlo;
			const uint32_t yhi = (uint32_t)(an.y >> 32);
			const int span = have ? (int)(yhi & 0xffu) : 0;
			const bool start = have && (g == rs || an.x - xprev > maxx);
			const bool single = start && (g + 1 >= re || xnext - an.x > maxx);
			any_seg |= have ? (yhi & 0x00ff0000u) : 0u;
			any_span0 |= (have && span == 0) ? 1u : 0u;
			if (single) {                                              // chain.c:251,283-284 with an empty window
				fb[t] = span; pb[t] = -1; vb[t] = span;
				flb[t] = (uint8_t)(span >= par.min_sc ? 2 | 8 : 0);       // emitted at its own step iff v >= min_sc (chain.c:304); bit3 = v >= min_sc
			}
			w_sum += (unsigned int)span;
			const uint64_t em = __builtin_amdgcn_ballot_w64(start && !single);
			const uint64_t sm = __builtin_amdgcn_ballot_w64(single);
			if (lane == 0 && gb + (threadIdx.x & ~63) < g1) start_mask[(gb + (threadIdx.x & ~63)) >> 6] = em;
			w_units += (unsigned int)__builtin_popcountll(em);
			w_singles += (unsigned int)__builtin_popcountll(sm);
		}
		if (__builtin_amdgcn_ballot_w64(any_seg != 0) && lane == 0) atomicOr(&sumq[rlo], SUMQ_SEG_FLAG);      // rare: multi-segment reads only
		if (__builtin_amdgcn_ballot_w64(any_span0 != 0) && lane == 0) atomicOr(&sumq[rlo], SUMQ_SPAN0_FLAG);  // (never, in minimap2's own anchors)
	} else
#pragma unroll
	for (int k = 0; k < PASSES; ++k) {
		const int64_t gb = g0 + (int64_t)k * PRE_BLOCK;
		if (gb >= g1) break;
		const int64_t g = gb + threadIdx.x;
		const bool have = g < g1;
		const ulonglong2 an = an_[k];
		uint64_t xprev = (uint64_t)__shfl_up((unsigned long long)an.x, 1, 64), xnext = (uint64_t)__shfl_down((unsigned long long)an.x, 1, 64);
		if (lane == 0) xprev = xb_[k];
		if (lane == 63 || g + 1 == g1) xnext = xe_[k];
		bool start = false, single = false;
		int span = 0;
		int64_t r = rlo;
		if (have) {
			if (!one_read) r = read_of(off, rlo, rhi, g);
			const int64_t rs = off[r], re = off[r + 1];
			span = span_of_hi((uint32_t)(an.y >> 32));
			if (seg_of_hi((uint32_t)(an.y >> 32)) != 0) atomicOr(&sumq[r], SUMQ_SEG_FLAG);   // rare: multi-segment reads only
			if (span == 0) atomicOr(&sumq[r], SUMQ_SPAN0_FLAG);                                // (never, in minimap2's own anchors: q_span is the k-mer span)
			start = g == rs || an.x - xprev > maxx;
			const bool next_starts = g + 1 >= re || xnext - an.x > maxx;
			single = start && next_starts;
			if (single) {                                          // chain.c:251,283-284 with an empty window
				f[g] = span; p[g] = -1; v[g] = span;
				flags[g] = (uint8_t)(span >= par.min_sc ? 2 | 8 : 0);  // emitted at its own step iff v >= min_sc (chain.c:304); bit3 = v >= min_sc
			}
		}
		// q_span sum (chain.c:240): per block when the block sits inside one read, else per wave when the
		// wave does, else (the one wave that straddles a read boundary) per lane
		if (one_read) w_sum += (unsigned int)span;
		else {
			const int64_t r_first = (int64_t)readlane_u64((uint64_t)r, 0);
			if (__builtin_amdgcn_ballot_w64(have && r != r_first) == 0) {
				int sw = have ? span : 0;
				for (int d = 32; d; d >>= 1) sw += __shfl_xor(sw, d, 64);
				if (lane == 0 && sw) atomicAdd(&sumq[r_first], (unsigned long long)sw);
			} else if (have) atomicAdd(&sumq[r], (unsigned long long)span);
		}
		const uint64_t em = __builtin_amdgcn_ballot_w64(start && !single);
		const uint64_t sm = __builtin_amdgcn_ballot_w64(single);
		if (lane == 0 && gb + (threadIdx.x & ~63) < g1) start_mask[(gb + (threadIdx.x & ~63)) >> 6] = em;
		w_units += (unsigned int)__builtin_popcountll(em);
		w_singles += (unsigned int)__builtin_popcountll(sm);
	}
	if (one_read) {
		for (int d = 32; d; d >>= 1) w_sum += __shfl_xor(w_sum, d, 64);
		if (lane == 0 && w_sum) atomicAdd(&s_sum, w_sum);
	}
	if (lane == 0) { atomicAdd(&s_units, w_units); atomicAdd(&s_singles, w_singles); }
	__syncthreads();
	if (threadIdx.x == 0) {
		if (one_read && s_sum) atomicAdd(&sumq[rlo], (unsigned long long)s_sum);
		block_cnt[blockIdx.x] = (unsigned long long)s_singles << 32 | s_units;   // two counters, one scan
	}
}

// Units are scheduled longest first (a unit is one wave's serial work, so a long one started last would be the
// kernel's tail): 128 length classes, class-descending order, order inside a class immaterial.
#define UNIT_CLASSES 128
__device__ __forceinline__ int unit_class(int32_t len)
{
	if (len < 4096) return len >> 6;                         // 0..63: 64-anchor steps
	const int c = 64 + (len >> 12);                          // 65..: 4096-anchor steps
	return c < UNIT_CLASSES ? c : UNIT_CLASSES - 1;
}

__global__ __launch_bounds__(256) void k_emit_units(int64_t n_reads, int64_t n_words, const int64_t *__restrict__ off,
                                                    const uint64_t *__restrict__ start_mask,
                                                    const unsigned long long *__restrict__ block_base, Unit *__restrict__ units,
                                                    unsigned int *__restrict__ hist, const int2 *__restrict__ block_reads)
{
	__shared__ unsigned int s_hist[UNIT_CLASSES];
	if (threadIdx.x < UNIT_CLASSES) s_hist[threadIdx.x] = 0;
	__syncthreads();
	const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t m = w < n_words ? start_mask[w] : 0;
	if (m) {
		const int64_t b = w / PRE_WORDS;
		uint64_t pos = (uint32_t)block_base[b];             // low word: units before this block
		for (int64_t k = b * PRE_WORDS; k < w; ++k) pos += (uint64_t)__builtin_popcountll(start_mask[k]);
		// the read of the word's first anchor: from the block's read range (k_block_reads) instead of a search over all reads
		// (fourteen dependent loads for 12 500 reads; a block usually lies in one read or straddles two)
		const int2 rr = block_reads[b];
		int64_t r = read_of(off, rr.x, rr.y, w << 6);
		while (m) {
			const int bit = __builtin_ctzll(m);
			m &= m - 1;
			const int64_t g = (w << 6) + bit;
			while (g >= off[r + 1]) ++r;                    // units of one word are in anchor order; reads only move forward
			const int64_t re = off[r + 1];
			// upper bound of the unit: the next unit's start or the end of the read (singletons in between are
			// not units, so this can overshoot the true end; the DP kernel finds the true end itself)
			int64_t next = -1;
			if (m) next = (w << 6) + __builtin_ctzll(m);
			else for (int64_t k = w + 1; k < n_words && (k << 6) < re; ++k) {
				const uint64_t mm = start_mask[k];
				if (mm) { next = (k << 6) + __builtin_ctzll(mm); break; }
			}
			const int64_t end = next >= 0 && next < re ? next : re;
			Unit u;
			u.start = g; u.read = (int32_t)r; u.len = (int32_t)(end - g);
			units[pos++] = u;
			atomicAdd(&s_hist[unit_class(u.len)], 1u);
		}
	}
	__syncthreads();
	if (threadIdx.x < UNIT_CLASSES && s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_hist[threadIdx.x]);
}

// hist[c] -> first position of class c in the longest-first order; cursor[c] = 0
__global__ void k_unit_bases(unsigned int *__restrict__ hist, unsigned int *__restrict__ cursor, unsigned int *__restrict__ key_range)
{
	if (threadIdx.x == 0) {
		if (key_range) { key_range[0] = 0xffffffffu; key_range[1] = 0u; }
		unsigned int acc = 0;
		for (int c = UNIT_CLASSES - 1; c >= 0; --c) { const unsigned int n = hist[c]; hist[c] = acc; acc += n; cursor[c] = 0; }
	}
}

// scatter into class order: ranks inside a block come from LDS atomics, one global atomic per (block, class)
// reserves the block's range (a global atomic per wave and class on the handful of hot classes cost 0.45 ms)
#define SCAT_PER_THREAD 4
__global__ __launch_bounds__(256) void k_unit_scatter(const unsigned long long *__restrict__ counters, const Unit *__restrict__ in,
                                                      const unsigned int *__restrict__ base, unsigned int *__restrict__ cursor,
                                                      Unit *__restrict__ out, Params par, const int64_t *__restrict__ off,
                                                      const unsigned long long *__restrict__ sumq, UnitAux *__restrict__ out_aux,
                                                      unsigned int *__restrict__ key_range)
{
	__shared__ unsigned int s_cnt[UNIT_CLASSES], s_base[UNIT_CLASSES];
	__shared__ unsigned int s_kmin, s_kmax;
	if (threadIdx.x == 0) { s_kmin = 0xffffffffu; s_kmax = 0u; }
	unsigned int kmin = 0xffffffffu, kmax = 0u;                     // table keys of this thread's units that the two / four-per-wave kernels may take
	const int64_t n = (int64_t)(uint32_t)counters[0];
	const int64_t per_block = 256 * SCAT_PER_THREAD;
	for (int64_t b0 = (int64_t)blockIdx.x * per_block; b0 < n; b0 += (int64_t)gridDim.x * per_block) {
		if (threadIdx.x < UNIT_CLASSES) s_cnt[threadIdx.x] = 0;
		__syncthreads();
		Unit u[SCAT_PER_THREAD];
		int cls[SCAT_PER_THREAD];
		unsigned int rank[SCAT_PER_THREAD];
		for (int k = 0; k < SCAT_PER_THREAD; ++k) {
			const int64_t i = b0 + k * 256 + threadIdx.x;
			cls[k] = -1;
			if (i < n) { u[k] = in[i]; cls[k] = unit_class(u[k].len); rank[k] = atomicAdd(&s_cnt[cls[k]], 1u); }
		}
		__syncthreads();
		if (threadIdx.x < UNIT_CLASSES && s_cnt[threadIdx.x])
			s_base[threadIdx.x] = base[threadIdx.x] + atomicAdd(&cursor[threadIdx.x], s_cnt[threadIdx.x]);
		__syncthreads();
		for (int k = 0; k < SCAT_PER_THREAD; ++k) if (cls[k] >= 0) {
			const unsigned int pos = s_base[cls[k]] + rank[k];
			out[pos] = u[k];
			if (out_aux) {
				// what k_chain_twin would otherwise fetch per unit (sumq[], off[]): the unit's place in its read, the key of the read's
				// cost table (same f32 divide as k_build_lut, chain.c:241) and whether the read is one for the general kernel
				const int64_t rs = off[u[k].read], n = off[u[k].read + 1] - rs;
				const unsigned long long sq = sumq[u[k].read];
				const float avg = (float)(uint64_t)(sq & ~SUMQ_FLAGS) / (float)n;
				const int lg = par.bw ? 31 - __builtin_clz((unsigned)par.bw) : 0;
				const bool lut16 = 1 - ((int)((double)par.bw * .01 * (double)avg) + (lg >> 1)) < -128;   // the table's last entry (k_build_lut)
				UnitAux ax;
				ax.rel0 = (int32_t)(u[k].start - rs); ax.lutkey = __float_as_uint(avg);
				ax.flags = ((sq & (SUMQ_SEG_FLAG | SUMQ_SPAN0_FLAG)) || lut16) ? 1u : 0u; ax.pad = 0;
				out_aux[pos] = ax;
				if (ax.flags == 0) { kmin = ax.lutkey < kmin ? ax.lutkey : kmin; kmax = ax.lutkey > kmax ? ax.lutkey : kmax; }   // (avg_qspan > 0: its bits order like integers)
			}
		}
		__syncthreads();
	}
	if (key_range) {                                               // one pair of atomics per block (they all hit the same two words)
		for (int d = 32; d; d >>= 1) { const unsigned int a = __shfl_xor(kmin, d, 64), b = __shfl_xor(kmax, d, 64); kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax; }
		if ((threadIdx.x & 63) == 0) { atomicMin(&s_kmin, kmin); atomicMax(&s_kmax, kmax); }
		__syncthreads();
		if (threadIdx.x == 0 && s_kmin <= s_kmax) { atomicMin(&key_range[0], s_kmin); atomicMax(&key_range[1], s_kmax); }
	}
}

// ---------------------------------------------------------------- K0b: per-read gap-cost table
// For a pair of anchors of the same segment in a non-cDNA run the gap cost depends only on
// dd = |dr - dq| <= bw and on the read's avg_qspan (chain.c:264,272):
//     cost(dd) = (int)(dd * .01 * avg_qspan) + (ilog2(dd) >> 1)
// so it is tabulated once per read (bw+1 entries, uint16) with exactly the
// reference's f32/f64 operations, and the hot loop does an LDS lookup instead of f64 arithmetic.
__global__ __launch_bounds__(256) void k_build_lut(Params par, int64_t n_reads, const int64_t *__restrict__ off,
                                                   unsigned long long *__restrict__ sumq, int lut_stride,
                                                   uint16_t *__restrict__ lut)
{
	for (int64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
		const int64_t n = off[r + 1] - off[r];
		if (n <= 0) continue;
		const float avg = (float)(uint64_t)(sumq[r] & ~SUMQ_FLAGS) / (float)n;   // chain.c:241
		const double avgd = (double)avg;
		for (int dd = threadIdx.x; dd <= par.bw; dd += blockDim.x) {
			const int lg = dd ? 31 - __builtin_clz((unsigned)dd) : 0;
			const int lin = (int)((double)dd * .01 * avgd);
			lut[r * lut_stride + dd] = (uint16_t)(int16_t)(1 - (lin + (lg >> 1)));  // stored as 1 - cost (see fast_masks): |.| < 2^15 for bw <= 4095, q_span <= 255
			// the cost grows with dd: the last entry tells whether the whole table fits a signed byte (k_chain_twin keeps it as bytes)
			if (dd == par.bw && 1 - (lin + (lg >> 1)) < -128) atomicOr(&sumq[r], SUMQ_LUT16_FLAG);
		}
	}
}

// ---------------------------------------------------------------- launchers

hipError_t launch_prepass(hipStream_t st, const Params &par, int64_t n_reads, int64_t total, const int64_t *d_off, const void *d_a,
                          unsigned long long *d_sumq, Unit *d_units, unsigned long long *d_counters, PrepassScratch sc,
                          int32_t *d_f, int32_t *d_p, int32_t *d_v, uint8_t *d_flags, UnitAux *d_unit_aux)
{
	hipError_t e = hipMemsetAsync(d_counters, 0, 2 * sizeof(unsigned long long), st);
	if (e != hipSuccess || n_reads <= 0 || total <= 0) return e;
	if ((e = hipMemsetAsync(d_sumq, 0, (size_t)n_reads * sizeof(unsigned long long), st)) != hipSuccess) return e;
	const int64_t blocks = (total + PRE_PER_BLOCK - 1) / PRE_PER_BLOCK;
	const int64_t words = (total + 63) / 64;
	hipLaunchKernelGGL(k_block_reads, dim3((unsigned)((blocks + 255) / 256)), dim3(256), 0, st, n_reads, total, d_off, sc.block_reads);
	hipLaunchKernelGGL(k_prepass, dim3((unsigned)blocks), dim3(PRE_BLOCK), 0, st, par, n_reads, total, d_off, (const ulonglong2*)d_a,
	                   d_sumq, sc.start_mask, sc.block_cnt, d_f, d_p, d_v, d_flags, sc.block_reads);
	if ((e = launch_scan_u64(st, blocks, sc.block_cnt, sc.tile_tmp, d_counters)) != hipSuccess) return e;
	if ((e = hipMemsetAsync(sc.hist, 0, 2 * UNIT_CLASSES * sizeof(unsigned int), st)) != hipSuccess) return e;
	hipLaunchKernelGGL(k_emit_units, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, n_reads, words, d_off,
	                   sc.start_mask, sc.block_cnt, sc.units_tmp, sc.hist, sc.block_reads);
	hipLaunchKernelGGL(k_unit_bases, dim3(1), dim3(64), 0, st, sc.hist, sc.hist + UNIT_CLASSES, d_unit_aux ? sc.key_range : nullptr);
	hipLaunchKernelGGL(k_unit_scatter, dim3((unsigned)(blocks < 1024 ? (blocks > 0 ? blocks : 1) : 1024)), dim3(256), 0, st, d_counters, sc.units_tmp,
	                   sc.hist, sc.hist + UNIT_CLASSES, d_units, par, d_off, d_sumq, d_unit_aux, d_unit_aux ? sc.key_range : nullptr);
	return hipGetLastError();
}

size_t prepass_scratch_bytes(int64_t max_anchors, size_t *mask_bytes, size_t *blocks_bytes)
{
	const size_t words = (size_t)(max_anchors + 63) / 64, blocks = (size_t)(max_anchors + PRE_PER_BLOCK - 1) / PRE_PER_BLOCK;
	*mask_bytes = (words + 1) * 8;
	*blocks_bytes = (blocks + 1) * 8;
	return *mask_bytes + 2 * *blocks_bytes;
}
hipError_t launch_lut(hipStream_t st, const Params &par, int64_t n_reads, const int64_t *d_off,
                      unsigned long long *d_sumq, int lut_stride, uint16_t *d_lut)
{
	if (n_reads <= 0) return hipSuccess;
	int64_t blocks = n_reads < 65536 ? n_reads : 65536;
	hipLaunchKernelGGL(k_build_lut, dim3((unsigned)blocks), dim3(256), 0, st, par, n_reads, d_off, d_sumq, lut_stride, d_lut);
	return hipGetLastError();
}

} // namespace chaindp
