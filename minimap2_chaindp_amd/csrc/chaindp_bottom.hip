// chaindp_bottom.hip -- the host half of the reference's chaining split, mm_chain_dp_bottom
// (reference chain.c:329-431), for every read of a resident batch, as GPU kernels.
//
// Input: the batch's new_seed[] records (compaction output) with their per-read offsets.  Per read the
// reference does, sequentially:
//   1. chain ends = records with the "v >= min_sc" flag that nobody points at          (chain.c:346-354)
//   2. per end, walk to the peak of f along the "f < v" flags; key = f[peak]<<32 | peak   (chain.c:362-370)
//   3. sort the keys, best first                                                          (chain.c:371-375)
//   4. in that order, backtrack each chain until an already visited record; keep it if long enough and, when
//      it ran into an older chain, if it still gained min_sc                              (chain.c:378-393)
//   5. emit the kept chains' anchors (ascending), then re-order the chains by the x of their first anchor
//      with the reference's unstable radix sort                                           (chain.c:401-426)
// Step 4 is the only order-dependent part.  A record is "visited" by the best-ranked chain whose full path to
// the root contains it (a chain stops at the first record a better chain reached, and that better chain, or a
// still better one, continues along the same path), so with owner[x] = min rank over the chains whose path
// contains x, chain k consists of its peak (always taken: chain.c:381 is a do-while) and the records below it while
// owner == k, and it stopped at the first record with owner < k.  owner[] is built with atomicMin by one walker per chain that stops as soon as it
// meets a smaller rank (whoever owns that record keeps walking).  Discarded chains still own their records,
// exactly as the reference leaves its t[] marks set (chain.c:392).
// Step 5's sort is unstable; its output order for equal keys is reproduced by running the reference's exact
// procedure (ksort.h:101-151) sequentially, one thread per read -- reads have tens of chains.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "chaindp_kernels.h"
#include "chaindp_rsort.h"

namespace chaindp {

struct SeedRec { uint64_t x, y; int32_t p, f; };   // == struct new_seed (minimap.h:51-55)

#define BT_BLOCK 256
#define BT_PER_BLOCK 1024

// largest r in [lo, hi] with off[r] <= g
__device__ __forceinline__ int64_t bt_read_of(const int64_t *__restrict__ off, int64_t lo, int64_t hi, int64_t g)
{
	while (lo < hi) {
		const int64_t mid = (lo + hi + 1) >> 1;
		if (off[mid] <= g) lo = mid; else hi = mid - 1;
	}
	return lo;
}

// B0: reads of the first and last record of every 1024-record block, so that the record-parallel kernels resolve
// "which read is record g in" with one compare instead of a 14-step binary search per thread
__global__ __launch_bounds__(256) void k_bt_block_reads(int64_t n_reads, int64_t m, const int64_t *__restrict__ soff, int2 *__restrict__ blk)
{
	const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t g0 = b * BT_PER_BLOCK;
	if (g0 >= m) return;
	const int64_t g1 = g0 + BT_PER_BLOCK < m ? g0 + BT_PER_BLOCK : m;
	const int64_t rlo = bt_read_of(soff, 0, n_reads - 1, g0);
	blk[b] = make_int2((int)rlo, (int)bt_read_of(soff, rlo, n_reads - 1, g1 - 1));
}

__device__ __forceinline__ int64_t bt_read_of_blk(const int64_t *__restrict__ soff, const int2 *__restrict__ blk, int64_t g)
{
	const int2 rr = blk[g / BT_PER_BLOCK];
	if (rr.x == rr.y) return rr.x;
	if (rr.y - rr.x == 1) return g >= soff[rr.y] ? rr.y : rr.x;
	return bt_read_of(soff, rr.x, rr.y, g);
}

// B1: has[j] = 1 if some record of the same read points at j (chain.c:347-349)
__global__ __launch_bounds__(BT_BLOCK) void k_bt_children(int64_t n_reads, int64_t m, const int64_t *__restrict__ soff,
                                                          const SeedRec *__restrict__ s, uint8_t *__restrict__ has, const int2 *__restrict__ blk,
                                                          int32_t *__restrict__ pdense)
{
	for (int64_t g = (int64_t)blockIdx.x * BT_BLOCK + threadIdx.x; g < m; g += (int64_t)gridDim.x * BT_BLOCK) {
		const int32_t p = s[g].p;
		pdense[g] = p;                                              // the later passes read 4 bytes per record instead of a 24-byte stride
		if (p >= 0) has[soff[bt_read_of_blk(soff, blk, g)] + (p >> 2)] = 1;
	}
}

// B2: chain ends per 1024-record block (chain.c:350-354)
__global__ __launch_bounds__(BT_BLOCK) void k_bt_end_count(int64_t m, const int32_t *__restrict__ pdense, const uint8_t *__restrict__ has,
                                                           unsigned long long *__restrict__ block_cnt)
{
	__shared__ unsigned int s_cnt;
	if (threadIdx.x == 0) s_cnt = 0;
	__syncthreads();
	const int64_t g0 = (int64_t)blockIdx.x * BT_PER_BLOCK;
	const int64_t g1 = g0 + BT_PER_BLOCK < m ? g0 + BT_PER_BLOCK : m;
	unsigned int mine = 0;
	for (int64_t g = g0 + threadIdx.x; g < g1; g += BT_BLOCK) mine += (unsigned int)((pdense[g] & 1) && !has[g]);
	for (int d = 32; d; d >>= 1) mine += __shfl_xor(mine, d, 64);
	if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_cnt, mine);
	__syncthreads();
	if (threadIdx.x == 0) block_cnt[blockIdx.x] = s_cnt;
}

// B3: the ends, in record order, as a flat list with per-read offsets (ends_off[r] = list position at the read's
// first record; reads without records share their successor's value, the tail is closed by k_bt_close_offsets)
__global__ __launch_bounds__(BT_BLOCK) void k_bt_end_list(int64_t n_reads, int64_t m, const int64_t *__restrict__ soff,
                                                          const int32_t *__restrict__ pdense, const uint8_t *__restrict__ has,
                                                          const unsigned long long *__restrict__ block_base,
                                                          int32_t *__restrict__ end_rec, int64_t *__restrict__ ends_off, const int2 *__restrict__ blk)
{
	__shared__ unsigned int s_w[4];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int64_t g0 = (int64_t)blockIdx.x * BT_PER_BLOCK;
	const int64_t g1 = g0 + BT_PER_BLOCK < m ? g0 + BT_PER_BLOCK : m;
	unsigned int carry = (unsigned int)block_base[blockIdx.x];
	for (int64_t gb = g0; gb < g1; gb += BT_BLOCK) {
		const int64_t g = gb + threadIdx.x;
		const bool is_end = g < g1 && (pdense[g] & 1) && !has[g];
		const uint64_t bm = __builtin_amdgcn_ballot_w64(is_end);
		const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
		if (lane == 0) s_w[wave] = __builtin_popcountll(bm);
		__syncthreads();
		unsigned int woff = 0, tot = 0;
		for (int w = 0; w < 4; ++w) { const unsigned int t = s_w[w]; if (w < wave) woff += t; tot += t; }
		const unsigned int pos = carry + woff + __builtin_popcountll(bm & below);
		if (g < g1) {
			const int64_t r = bt_read_of_blk(soff, blk, g);
			const int64_t so = soff[r];
			if (is_end) end_rec[pos] = (int32_t)(g - so);
			if (g == so) for (int64_t rr = r; rr >= 0 && soff[rr] == so; --rr) ends_off[rr] = (int64_t)pos;
		}
		carry += tot;
		__syncthreads();
	}
}

__global__ void k_bt_close_offsets(int64_t n_reads, int64_t m, const int64_t *__restrict__ soff,
                                   const unsigned long long *__restrict__ total, int64_t *__restrict__ offs)
{
	const int64_t t = (int64_t)(uint32_t)*total;
	offs[n_reads] = t;
	for (int64_t r = n_reads - 1; r >= 0 && soff[r] == m; --r) offs[r] = t;
}

// B4: walk every end to the peak of f (chain.c:362-370)
__global__ __launch_bounds__(BT_BLOCK) void k_bt_peaks(int64_t n_reads, const int64_t *__restrict__ soff, const SeedRec *__restrict__ s,
                                                       const int64_t *__restrict__ ends_off, const int32_t *__restrict__ end_rec,
                                                       unsigned long long *__restrict__ key, int32_t *__restrict__ end_read)
{
	const int64_t n_e = ends_off[n_reads];
	for (int64_t e = (int64_t)blockIdx.x * BT_BLOCK + threadIdx.x; e < n_e; e += (int64_t)gridDim.x * BT_BLOCK) {
		const int64_t r = bt_read_of(ends_off, 0, n_reads - 1, e);
		end_read[e] = (int32_t)r;                                   // the later per-chain kernels (same list positions) reuse it
		const SeedRec *sr = s + soff[r];
		const int32_t i = end_rec[e];
		int32_t j = i;
		while (j >= 0 && (sr[j].p & 2)) j = sr[j].p >> 2;
		if (j < 0) j = i;
		key[e] = (unsigned long long)(long long)sr[j].f << 32 | (unsigned long long)(uint32_t)j;
	}
}

// B5: per read, keys in descending order (chain.c:371-375), by counting.  Two ends can share a peak and then have
// identical keys; equal keys are interchangeable (whichever comes second finds its peak visited), so ties are
// simply broken by list position.
__global__ __launch_bounds__(BT_BLOCK) void k_bt_rank(int64_t n_reads, const int64_t *__restrict__ ends_off,
                                                      const unsigned long long *__restrict__ key, unsigned long long *__restrict__ skey)
{
	__shared__ unsigned long long tile[1024];
	for (int64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
		const int64_t b = ends_off[r], n = ends_off[r + 1] - b;
		if (n <= 0) continue;
		for (int64_t e0 = 0; e0 < n; e0 += BT_BLOCK) {              // every thread ranks one key per round
			const int64_t e = e0 + threadIdx.x;
			const unsigned long long mine = e < n ? key[b + e] : 0;
			int64_t rank = 0;
			for (int64_t t0 = 0; t0 < n; t0 += 1024) {
				__syncthreads();
				for (int k = threadIdx.x; k < 1024; k += BT_BLOCK) tile[k] = t0 + k < n ? key[b + t0 + k] : 0;
				__syncthreads();
				const int lim = n - t0 < 1024 ? (int)(n - t0) : 1024;
				for (int k = 0; k < lim; ++k) rank += tile[k] > mine || (tile[k] == mine && t0 + k < e);
			}
			if (e < n) skey[b + rank] = mine;
		}
		__syncthreads();
	}
}

// Reads whose records fit the LDS (p and owner, 8 B per record) get B6+B7 from one workgroup working in LDS
// (k_bt_read_lds, launched twice: reads of up to BT_LDS_RECS records at two workgroups per CU, reads of up to
// BT_LDS_RECS_MAX at one); the global-memory kernels below only serve the still longer reads.
#define BT_LDS_RECS 10112   // 2 x (8 B x 10112 + 1 KB) = the CU's 160 KB: still two workgroups per CU
#define BT_LDS_RECS_MAX 20000

// B6+B7 for one read per workgroup, in LDS: owner by LDS atomicMin walkers (one thread per chain), then length,
// score and fate of every chain.  Scattered returning atomics in HBM are ~9 G/s chip-wide; in LDS they cost ~100 cycles.
__global__ __launch_bounds__(BT_BLOCK) void k_bt_read_lds(int64_t n_reads, int min_recs, int max_recs, int min_cnt, int min_sc, const int64_t *__restrict__ soff,
                                                          const SeedRec *__restrict__ s, const int64_t *__restrict__ ends_off,
                                                          const unsigned long long *__restrict__ skey,
                                                          int32_t *__restrict__ ccnt, unsigned long long *__restrict__ cu, const int32_t *__restrict__ pdense)
{
	extern __shared__ int32_t bt_lds[];
	int32_t *s_p = bt_lds, *s_own = bt_lds + max_recs;
	for (int64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
		const int64_t so = soff[r];
		const int32_t m = (int32_t)(soff[r + 1] - so);
		const int64_t cb = ends_off[r];
		const int32_t nc = (int32_t)(ends_off[r + 1] - cb);
		if (m <= min_recs || m > max_recs || nc <= 0) continue;
		const SeedRec *sr = s + so;
		__syncthreads();
		for (int32_t j = threadIdx.x; j < m; j += BT_BLOCK) { s_p[j] = pdense[so + j]; s_own[j] = 0x7fffffff; }
		__syncthreads();
		for (int32_t k = threadIdx.x; k < nc; k += BT_BLOCK) {          // walkers: rank k claims its path until a better rank owns it
			int32_t j = (int32_t)(uint32_t)skey[cb + k];
			while (j >= 0) {
				if (atomicMin(&s_own[j], k) < k) break;
				const int32_t p = s_p[j];
				j = p >= 0 ? p >> 2 : -1;
			}
		}
		__syncthreads();
		for (int32_t k = threadIdx.x; k < nc; k += BT_BLOCK) {          // chain.c:380-392
			const unsigned long long ky = skey[cb + k];
			int32_t j = (int32_t)(uint32_t)ky, cnt = 1;                  // the peak is taken unconditionally (do-while)
			{ const int32_t p = s_p[j]; j = p >= 0 ? p >> 2 : -1; }
			while (j >= 0 && s_own[j] == k) {
				++cnt;
				const int32_t p = s_p[j];
				j = p >= 0 ? p >> 2 : -1;
			}
			unsigned long long sc = ky >> 32;
			bool keep = true;
			if (j >= 0) {
				const int32_t fj = sr[j].f;
				keep = (int32_t)(ky >> 32) - fj >= min_sc;
				sc = (ky >> 32) - (unsigned long long)(long long)fj;
			}
			keep = keep && cnt >= min_cnt;
			ccnt[cb + k] = keep ? cnt : 0;
			cu[cb + k] = sc << 32 | (unsigned long long)(uint32_t)cnt;
		}
	}
}

// B6: owner[x] = best (smallest) rank among the chains whose path contains x
__global__ __launch_bounds__(BT_BLOCK) void k_bt_own(int64_t n_reads, const int64_t *__restrict__ soff, const SeedRec *__restrict__ s,
                                                     const int64_t *__restrict__ ends_off, const unsigned long long *__restrict__ skey,
                                                     int32_t *__restrict__ owner, const int32_t *__restrict__ end_read)
{
	const int64_t n_e = ends_off[n_reads];
	for (int64_t c = (int64_t)blockIdx.x * BT_BLOCK + threadIdx.x; c < n_e; c += (int64_t)gridDim.x * BT_BLOCK) {
		const int64_t r = end_read[c];
		if (soff[r + 1] - soff[r] <= BT_LDS_RECS_MAX) continue;     // done in LDS by k_bt_read_lds
		const int32_t k = (int32_t)(c - ends_off[r]);
		const SeedRec *sr = s + soff[r];
		int32_t *ow = owner + soff[r];
		int32_t j = (int32_t)(uint32_t)skey[c];
		while (j >= 0) {
			if (atomicMin(&ow[j], k) < k) break;                    // a better chain owns it and walks on from here
			const int32_t p = sr[j].p;
			j = p >= 0 ? p >> 2 : -1;
		}
	}
}

// B7: length, score and fate of every chain (chain.c:380-392)
__global__ __launch_bounds__(BT_BLOCK) void k_bt_score(int64_t n_reads, int min_cnt, int min_sc, const int64_t *__restrict__ soff,
                                                       const SeedRec *__restrict__ s, const int64_t *__restrict__ ends_off,
                                                       const unsigned long long *__restrict__ skey, const int32_t *__restrict__ owner,
                                                       int32_t *__restrict__ ccnt, unsigned long long *__restrict__ cu, const int32_t *__restrict__ end_read)
{
	const int64_t n_e = ends_off[n_reads];
	for (int64_t c = (int64_t)blockIdx.x * BT_BLOCK + threadIdx.x; c < n_e; c += (int64_t)gridDim.x * BT_BLOCK) {
		const int64_t r = end_read[c];
		if (soff[r + 1] - soff[r] <= BT_LDS_RECS_MAX) continue;     // done in LDS by k_bt_read_lds
		const int32_t k = (int32_t)(c - ends_off[r]);
		const SeedRec *sr = s + soff[r];
		const int32_t *ow = owner + soff[r];
		const unsigned long long ky = skey[c];
		// chain.c:381-386 is a do-while: the peak is taken unconditionally (even when a better chain already
		// passed through it), the following records only while nobody visited them, i.e. while owner == k
		int32_t j = (int32_t)(uint32_t)ky, cnt = 1;
		{ const int32_t p = sr[j].p; j = p >= 0 ? p >> 2 : -1; }
		while (j >= 0 && ow[j] == k) {
			++cnt;
			const int32_t p = sr[j].p;
			j = p >= 0 ? p >> 2 : -1;
		}
		unsigned long long sc = ky >> 32;
		bool keep = true;
		if (j >= 0) {                                               // ran into an older chain: must still gain min_sc
			keep = (int32_t)(ky >> 32) - sr[j].f >= min_sc;
			sc = (ky >> 32) - (unsigned long long)(long long)sr[j].f;
		}
		keep = keep && cnt >= min_cnt;
		ccnt[c] = keep ? cnt : 0;
		cu[c] = sc << 32 | (unsigned long long)(uint32_t)cnt;
	}
}

// B8: per read (one wave each), positions of the kept chains in rank order and of their anchors
__global__ __launch_bounds__(BT_BLOCK) void k_bt_layout(int64_t n_reads, const int64_t *__restrict__ ends_off,
                                                        const int32_t *__restrict__ ccnt, int32_t *__restrict__ kpos,
                                                        int32_t *__restrict__ bpos, unsigned long long *__restrict__ read_tot)
{
	const int lane = threadIdx.x & 63;
	const int64_t wave0 = (int64_t)blockIdx.x * (BT_BLOCK >> 6) + (threadIdx.x >> 6);
	const int64_t n_waves = (int64_t)gridDim.x * (BT_BLOCK >> 6);
	for (int64_t r = wave0; r < n_reads; r += n_waves) {
		const int64_t b = ends_off[r], n = ends_off[r + 1] - b;
		unsigned int kc = 0, ac = 0;
		for (int64_t t0 = 0; t0 < n; t0 += 64) {
			const int64_t e = t0 + lane;
			const int cnt = e < n ? ccnt[b + e] : 0;
			const uint64_t km = __builtin_amdgcn_ballot_w64(cnt > 0);
			const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
			unsigned int incl = (unsigned int)cnt;
			for (int d = 1; d < 64; d <<= 1) { const unsigned int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
			if (cnt > 0) { kpos[b + e] = (int32_t)(kc + __builtin_popcountll(km & below)); bpos[b + e] = (int32_t)(ac + incl - cnt); }
			kc += __builtin_popcountll(km);
			ac += __shfl(incl, 63, 64);
		}
		if (lane == 0) read_tot[r] = (unsigned long long)ac << 32 | kc;
	}
}

// after the scan of read_tot: chains_off / b_off (int64, n_reads + 1 entries)
__global__ void k_bt_offsets(int64_t n_reads, const unsigned long long *__restrict__ read_base, const unsigned long long *__restrict__ total,
                             int64_t *__restrict__ chains_off, int64_t *__restrict__ b_off)
{
	for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_reads; r += (int64_t)gridDim.x * blockDim.x) {
		const unsigned long long v = r < n_reads ? read_base[r] : *total;
		chains_off[r] = (int64_t)(uint32_t)v;
		b_off[r] = (int64_t)(v >> 32);
	}
}

// B9-B11 for one read per workgroup, in LDS (reads of up to max_recs records; the three kernels below serve the longer
// ones): every kept chain is walked in LDS (p staged densely) into a list of record indices in rank order, the read's
// chains are put into the reference's final order (chain.c:410-426: radix_sort_128x on the first anchor's x, by one
// thread -- a read has tens of chains), and the anchors are gathered straight into their final places.  Replaces a
// pointer chase through HBM per chain, an intermediate copy of all chain anchors and a second pass over them.
__global__ __launch_bounds__(BT_BLOCK) void k_bt_emit_lds(int64_t n_reads, int min_recs, int max_recs, const int64_t *__restrict__ soff,
                                                          const SeedRec *__restrict__ s, const int32_t *__restrict__ pdense,
                                                          const int64_t *__restrict__ ends_off, const unsigned long long *__restrict__ skey,
                                                          const int32_t *__restrict__ ccnt, const unsigned long long *__restrict__ cu,
                                                          const int32_t *__restrict__ kpos, const int32_t *__restrict__ bpos,
                                                          const int64_t *__restrict__ chains_off, const int64_t *__restrict__ b_off,
                                                          unsigned long long *__restrict__ u_tmp, unsigned long long *__restrict__ u_out,
                                                          ulonglong2 *__restrict__ w, BtRange *__restrict__ stacks,
                                                          int32_t *__restrict__ chain_read, ulonglong2 *__restrict__ b_out)
{
	extern __shared__ int32_t bt_lds[];
	int32_t *s_p = bt_lds, *s_list = bt_lds + max_recs;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (int64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
		const int64_t so = soff[r];
		const int32_t m = (int32_t)(soff[r + 1] - so);
		const int64_t cb = ends_off[r], kb = chains_off[r], bb = b_off[r];
		const int32_t nc = (int32_t)(ends_off[r + 1] - cb), nk = (int32_t)(chains_off[r + 1] - kb);
		if (m <= min_recs || m > max_recs || nk <= 0) continue;
		const SeedRec *sr = s + so;
		__syncthreads();
		for (int32_t j = threadIdx.x; j < m; j += BT_BLOCK) s_p[j] = pdense[so + j];
		__syncthreads();
		for (int32_t k = threadIdx.x; k < nc; k += BT_BLOCK) {          // chain.c:404-406: reversed walk order, as record indices
			const int32_t cnt = ccnt[cb + k];
			if (cnt <= 0) continue;
			const int32_t pos = bpos[cb + k], ck = kpos[cb + k];
			int32_t j = (int32_t)(uint32_t)skey[cb + k];
			for (int32_t t = cnt - 1; t >= 0; --t) {
				s_list[pos + t] = j;
				const int32_t p = s_p[j];
				j = p >= 0 ? p >> 2 : -1;
			}
			u_tmp[kb + ck] = cu[cb + k];
			w[kb + ck] = make_ulonglong2(sr[s_list[pos]].x, (unsigned long long)(uint32_t)pos << 32 | (unsigned long long)(uint32_t)ck);
		}
		__threadfence_block();
		__syncthreads();
		if (nk <= 64) {
			// chain.c:410-426 for up to 64 chains is the reference's insertion sort, which is stable: the final place of a
			// chain is the number of chains with a smaller first x, or an equal one and a smaller index -- by one wave
			if (wave == 0) {
				uint64_t *f_u = (uint64_t*)(bt_lds + 2 * max_recs);
				int32_t *f_pos = (int32_t*)(f_u + 64);
				const ulonglong2 me = lane < nk ? w[kb + lane] : make_ulonglong2(~0ull, 0);
				int32_t rank = 0;
				for (int j = 0; j < nk; ++j) {
					const uint64_t xj = (uint64_t)__shfl((unsigned long long)me.x, j, 64);
					rank += xj < me.x || (xj == me.x && j < lane);
				}
				if (lane < nk) { f_u[rank] = u_tmp[kb + (uint32_t)me.y]; f_pos[rank] = (int32_t)(me.y >> 32); }
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				const uint64_t u = lane < nk ? f_u[lane] : 0;
				const uint32_t cnt = (uint32_t)u;
				uint32_t incl = cnt;
				for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
				if (lane < nk) {
					u_out[kb + lane] = u;
					chain_read[kb + lane] = (int32_t)r;
					w[kb + lane] = make_ulonglong2((unsigned long long)(uint32_t)f_pos[lane], (unsigned long long)(incl - cnt) << 32 | cnt);
				}
			}
		} else if (threadIdx.x == 0) {                                   // chain.c:410-426, more than 64 chains: the radix procedure
			bt_radix_128x(w + kb, nk, stacks + kb / 64 + 2 * r);
			int32_t k = 0;
			for (int32_t i = 0; i < nk; ++i) {
				const unsigned long long y = w[kb + i].y;
				const unsigned long long u = u_tmp[kb + (uint32_t)y];
				u_out[kb + i] = u;
				chain_read[kb + i] = (int32_t)r;
				w[kb + i] = make_ulonglong2(y >> 32, (unsigned long long)(uint32_t)k << 32 | (uint32_t)u);   // list position, final offset | count
				k += (int32_t)(uint32_t)u;
			}
		}
		__threadfence_block();
		__syncthreads();
		for (int32_t i = wave; i < nk; i += BT_BLOCK >> 6) {             // one wave per chain: gather into the final place
			const ulonglong2 e = w[kb + i];
			const int32_t src = (int32_t)e.x, dst = (int32_t)(e.y >> 32), n = (int32_t)(uint32_t)e.y;
			for (int32_t t = lane; t < n; t += 64) {
				const SeedRec rec = sr[s_list[src + t]];
				b_out[bb + dst + t] = make_ulonglong2(rec.x, rec.y);
			}
		}
	}
}

// B9: kept chains, still in rank order: anchors (ascending along the chain), u, and the sort keys of chain.c:412-416
// (B9-B11 below: only for reads too long for k_bt_emit_lds)
__global__ __launch_bounds__(BT_BLOCK) void k_bt_emit(int64_t n_reads, const int64_t *__restrict__ soff, const SeedRec *__restrict__ s,
                                                      const int64_t *__restrict__ ends_off, const unsigned long long *__restrict__ skey,
                                                      const int32_t *__restrict__ ccnt, const unsigned long long *__restrict__ cu,
                                                      const int32_t *__restrict__ kpos, const int32_t *__restrict__ bpos,
                                                      const int64_t *__restrict__ chains_off, const int64_t *__restrict__ b_off,
                                                      ulonglong2 *__restrict__ b_tmp, unsigned long long *__restrict__ u_tmp,
                                                      ulonglong2 *__restrict__ w, const int32_t *__restrict__ end_read)
{
	const int64_t n_e = ends_off[n_reads];
	for (int64_t c = (int64_t)blockIdx.x * BT_BLOCK + threadIdx.x; c < n_e; c += (int64_t)gridDim.x * BT_BLOCK) {
		const int32_t cnt = ccnt[c];
		if (cnt <= 0) continue;
		const int64_t r = end_read[c];
		if (soff[r + 1] - soff[r] <= BT_LDS_RECS_MAX) continue;     // done by k_bt_emit_lds
		const SeedRec *sr = s + soff[r];
		ulonglong2 *dst = b_tmp + b_off[r] + bpos[c];
		int32_t j = (int32_t)(uint32_t)skey[c];
		unsigned long long first_x = 0;
		for (int32_t t = cnt - 1; t >= 0; --t) {                    // chain.c:404-406: reversed walk order
			const SeedRec rec = sr[j];
			dst[t] = make_ulonglong2(rec.x, rec.y);
			first_x = rec.x;
			j = rec.p >= 0 ? rec.p >> 2 : -1;
		}
		const int64_t ck = chains_off[r] + kpos[c];
		u_tmp[ck] = cu[c];
		w[ck] = make_ulonglong2(first_x, (unsigned long long)(uint32_t)bpos[c] << 32 | (unsigned long long)(uint32_t)kpos[c]);
	}
}

// B10: chains of a read in the reference's final order (chain.c:410-426); one thread per read
__global__ __launch_bounds__(64) void k_bt_xsort(int64_t n_reads, const int64_t *__restrict__ soff, const int64_t *__restrict__ chains_off, ulonglong2 *__restrict__ w,
                                                 const unsigned long long *__restrict__ u_tmp, unsigned long long *__restrict__ u_out,
                                                 int32_t *__restrict__ c_src, int32_t *__restrict__ c_dst, BtRange *__restrict__ stacks,
                                                 int32_t *__restrict__ chain_read)
{
	for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (int64_t)gridDim.x * blockDim.x) {
		const int64_t b = chains_off[r];
		const int32_t n = (int32_t)(chains_off[r + 1] - b);
		if (n <= 0 || soff[r + 1] - soff[r] <= BT_LDS_RECS_MAX) continue;   // short reads: k_bt_emit_lds
		bt_radix_128x(w + b, n, stacks + b / 64 + 2 * r);           // at most n/65 pending ranges
		int32_t k = 0;
		for (int32_t i = 0; i < n; ++i) {
			const unsigned long long y = w[b + i].y;
			const unsigned long long u = u_tmp[b + (uint32_t)y];
			u_out[b + i] = u;
			c_src[b + i] = (int32_t)(y >> 32);
			c_dst[b + i] = k;
			chain_read[b + i] = (int32_t)r;                         // for k_bt_copy, instead of a search per chain
			k += (int32_t)(uint32_t)u;
		}
	}
}

// B11: anchors into their final place, one wave per chain
__global__ __launch_bounds__(BT_BLOCK) void k_bt_copy(int64_t n_reads, const int64_t *__restrict__ soff, const int64_t *__restrict__ chains_off, const int64_t *__restrict__ b_off,
                                                      const unsigned long long *__restrict__ u_out, const int32_t *__restrict__ c_src,
                                                      const int32_t *__restrict__ c_dst, const ulonglong2 *__restrict__ b_tmp,
                                                      ulonglong2 *__restrict__ b_out, const int32_t *__restrict__ chain_read)
{
	const int lane = threadIdx.x & 63;
	const int64_t n_c = chains_off[n_reads];
	const int64_t wave0 = (int64_t)blockIdx.x * (BT_BLOCK >> 6) + (threadIdx.x >> 6);
	const int64_t n_waves = (int64_t)gridDim.x * (BT_BLOCK >> 6);
	for (int64_t c = wave0; c < n_c; c += n_waves) {
		const int64_t r = chain_read[c];
		if (soff[r + 1] - soff[r] <= BT_LDS_RECS_MAX) continue;     // short reads: k_bt_emit_lds
		const int64_t bb = b_off[r];
		const int32_t n = (int32_t)(uint32_t)u_out[c];
		const ulonglong2 *src = b_tmp + bb + c_src[c];
		ulonglong2 *dst = b_out + bb + c_dst[c];
		for (int32_t k = lane; k < n; k += 64) dst[k] = src[k];
	}
}

static inline unsigned bt_grid(int64_t n, int per)
{
	int64_t g = (n + per - 1) / per;
	if (g < 1) g = 1;
	if (g > 65535) g = 65535;
	return (unsigned)g;
}

hipError_t launch_backtrack(hipStream_t st, int min_cnt, int min_sc, int64_t n_reads, int64_t m_cap, const int64_t *d_soff, const void *d_seeds,
                            const unsigned long long *d_n_seeds, BottomScratch sc, int64_t n_seeds_host)
{
	hipError_t e;
	const int64_t m = n_seeds_host;
	(void)m_cap; (void)d_n_seeds;
	if (n_reads <= 0 || m <= 0) {
		if ((e = hipMemsetAsync(sc.chains_off, 0, (size_t)(n_reads > 0 ? n_reads + 1 : 1) * 8, st)) != hipSuccess) return e;
		return hipMemsetAsync(sc.b_off, 0, (size_t)(n_reads > 0 ? n_reads + 1 : 1) * 8, st);
	}
	const SeedRec *s = (const SeedRec*)d_seeds;
	const int64_t blocks = (m + BT_PER_BLOCK - 1) / BT_PER_BLOCK;
	if ((e = hipMemsetAsync(sc.has, 0, (size_t)m, st)) != hipSuccess) return e;
	if ((e = hipMemsetAsync(sc.owner, 0x7f, (size_t)m * 4, st)) != hipSuccess) return e;
	int2 *blk = (int2*)sc.c_src;                                    // c_src is not needed before k_bt_xsort: 2 ints per 1024 records fit
	int32_t *pdense = (int32_t*)sc.b_tmp;                           // b_tmp is only written by the long reads' k_bt_emit, at the very end: the records' p fields, densely
	int32_t *chain_read = (int32_t*)sc.key;                         // key is dead after k_bt_rank: final chain -> read
	hipLaunchKernelGGL(k_bt_block_reads, dim3((unsigned)((blocks + 255) / 256)), dim3(256), 0, st, n_reads, m, d_soff, blk);
	hipLaunchKernelGGL(k_bt_children, dim3(bt_grid(m, BT_BLOCK)), dim3(BT_BLOCK), 0, st, n_reads, m, d_soff, s, sc.has, blk, pdense);
	hipLaunchKernelGGL(k_bt_end_count, dim3((unsigned)blocks), dim3(BT_BLOCK), 0, st, m, pdense, sc.has, sc.block_cnt);
	if ((e = launch_scan_u64(st, blocks, sc.block_cnt, sc.tile_tmp, sc.total)) != hipSuccess) return e;
	hipLaunchKernelGGL(k_bt_end_list, dim3((unsigned)blocks), dim3(BT_BLOCK), 0, st, n_reads, m, d_soff, pdense, sc.has, sc.block_cnt, sc.end_rec, sc.ends_off, blk);
	hipLaunchKernelGGL(k_bt_close_offsets, dim3(1), dim3(1), 0, st, n_reads, m, d_soff, sc.total, sc.ends_off);
	const unsigned gE = bt_grid(m, BT_BLOCK) < 4096 ? bt_grid(m, BT_BLOCK) : 4096;   // per-chain kernels: ends <= records, usually far fewer; grid-stride loops
	hipLaunchKernelGGL(k_bt_peaks, dim3(gE), dim3(BT_BLOCK), 0, st, n_reads, d_soff, s, sc.ends_off, sc.end_rec, sc.key, sc.c_dst);   // c_dst doubles as the end -> read table until k_bt_xsort
	hipLaunchKernelGGL(k_bt_rank, dim3(bt_grid(n_reads, 1)), dim3(BT_BLOCK), 0, st, n_reads, sc.ends_off, sc.key, sc.skey);
	hipLaunchKernelGGL(k_bt_read_lds, dim3(bt_grid(n_reads, 1)), dim3(BT_BLOCK), (size_t)BT_LDS_RECS * 8, st, n_reads, 0, BT_LDS_RECS, min_cnt, min_sc, d_soff, s,
	                   sc.ends_off, sc.skey, sc.ccnt, sc.cu, pdense);
	hipLaunchKernelGGL(k_bt_read_lds, dim3(bt_grid(n_reads, 1) < 512 ? bt_grid(n_reads, 1) : 512), dim3(BT_BLOCK), (size_t)BT_LDS_RECS_MAX * 8, st, n_reads, BT_LDS_RECS, BT_LDS_RECS_MAX, min_cnt, min_sc, d_soff, s,
	                   sc.ends_off, sc.skey, sc.ccnt, sc.cu, pdense);
	hipLaunchKernelGGL(k_bt_own, dim3(gE), dim3(BT_BLOCK), 0, st, n_reads, d_soff, s, sc.ends_off, sc.skey, sc.owner, sc.c_dst);
	hipLaunchKernelGGL(k_bt_score, dim3(gE), dim3(BT_BLOCK), 0, st, n_reads, min_cnt, min_sc, d_soff, s, sc.ends_off, sc.skey, sc.owner, sc.ccnt, sc.cu, sc.c_dst);
	hipLaunchKernelGGL(k_bt_layout, dim3(bt_grid(n_reads, 4)), dim3(BT_BLOCK), 0, st, n_reads, sc.ends_off, sc.ccnt, sc.kpos, sc.bpos, sc.read_tot);
	if ((e = launch_scan_u64(st, n_reads, sc.read_tot, sc.tile_tmp, sc.total)) != hipSuccess) return e;
	hipLaunchKernelGGL(k_bt_offsets, dim3(bt_grid(n_reads + 1, 256)), dim3(256), 0, st, n_reads, sc.read_tot, sc.total, sc.chains_off, sc.b_off);
	hipLaunchKernelGGL(k_bt_emit_lds, dim3(bt_grid(n_reads, 1)), dim3(BT_BLOCK), (size_t)BT_LDS_RECS * 8 + 1024, st, n_reads, 0, BT_LDS_RECS, d_soff, s, pdense,
	                   sc.ends_off, sc.skey, sc.ccnt, sc.cu, sc.kpos, sc.bpos, sc.chains_off, sc.b_off, sc.u_tmp, sc.u_out, (ulonglong2*)sc.w,
	                   (BtRange*)sc.stacks, chain_read, (ulonglong2*)sc.b_out);
	hipLaunchKernelGGL(k_bt_emit_lds, dim3(bt_grid(n_reads, 1) < 512 ? bt_grid(n_reads, 1) : 512), dim3(BT_BLOCK), (size_t)BT_LDS_RECS_MAX * 8 + 1024, st, n_reads,
	                   BT_LDS_RECS, BT_LDS_RECS_MAX, d_soff, s, pdense, sc.ends_off, sc.skey, sc.ccnt, sc.cu, sc.kpos, sc.bpos, sc.chains_off, sc.b_off,
	                   sc.u_tmp, sc.u_out, (ulonglong2*)sc.w, (BtRange*)sc.stacks, chain_read, (ulonglong2*)sc.b_out);
	// reads with more than BT_LDS_RECS_MAX records: the same three steps through global memory
	hipLaunchKernelGGL(k_bt_emit, dim3(gE), dim3(BT_BLOCK), 0, st, n_reads, d_soff, s, sc.ends_off, sc.skey, sc.ccnt, sc.cu, sc.kpos, sc.bpos,
	                   sc.chains_off, sc.b_off, (ulonglong2*)sc.b_tmp, sc.u_tmp, (ulonglong2*)sc.w, sc.c_dst);
	hipLaunchKernelGGL(k_bt_xsort, dim3(bt_grid(n_reads, 64)), dim3(64), 0, st, n_reads, d_soff, sc.chains_off, (ulonglong2*)sc.w, sc.u_tmp, sc.u_out,
	                   sc.c_src, sc.c_dst, (BtRange*)sc.stacks, chain_read);
	hipLaunchKernelGGL(k_bt_copy, dim3(bt_grid(m, 64) < 8192 ? bt_grid(m, 64) : 8192), dim3(BT_BLOCK), 0, st, n_reads, d_soff, sc.chains_off, sc.b_off, sc.u_out, sc.c_src, sc.c_dst,
	                   (const ulonglong2*)sc.b_tmp, (ulonglong2*)sc.b_out, chain_read);
	return hipGetLastError();
}

} // namespace chaindp
