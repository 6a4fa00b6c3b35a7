// chaindp_regs.hip -- chains to hits on the GPU: mm_gen_regs (hit.c:52-95, with mm_reg_set_coor and
// mm_cal_fuzzy_len, hit.c:8-38) and mm_est_err (esterr.c:30-64), over the chains chaindp_backtrack left in HBM.
//
//   k_regs_keys   wave per read: for every chain its offset in the read's chain anchors (a running sum of the counts),
//                 the sort key of hit.c:61-68 (score and count with the low 32 bits scrambled by hash64 of the chain's
//                 first anchor and the read's hash), and the reference's radix_sort_128x of those keys -- its
//                 insertion sort (stable: a rank by key and index, one lane per chain) for up to 64 chains, its
//                 procedure step by step (chaindp_rsort.h) by one lane above
//   k_regs_fill   wave per read, one hit at a time: the record of hit.c:74-86 from the reversed key order, coordinates
//                 from the chain's first and last anchor, mlen / blen as a sum over its anchors (a lane per anchor)
//   k_regs_div    thread per hit: esterr.c:42-62 (binary search of the chain's first minimizer in mini_pos, a linear
//                 match of the following ones, logf of the ratio)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "chaindp_kernels.h"
#include "chaindp_rsort.h"
#include "chaindp_wave.h"

namespace chaindp {

__device__ __forceinline__ uint64_t regs_hash64(uint64_t key)       // hit.c:40-50
{
	key = (~key + (key << 21));
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8));
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4));
	key = key ^ key >> 28;
	key = (key + (key << 31));
	return key;
}

__global__ __launch_bounds__(256) void k_regs_keys(int64_t n_reads, const int64_t *__restrict__ chains_off, const int64_t *__restrict__ b_off,
                                                   const unsigned long long *__restrict__ u, const ulonglong2 *__restrict__ b,
                                                   const uint32_t *__restrict__ hash, ulonglong2 *__restrict__ z, BtRange *__restrict__ stacks)
{
	const int lane = threadIdx.x & 63;
	const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (r >= n_reads) return;
	const int64_t c0 = chains_off[r];
	const int n_u = (int)(chains_off[r + 1] - c0);
	if (n_u <= 0) return;
	const ulonglong2 *a = b + b_off[r];
	const uint32_t hs = hash[r];
	int carry = 0;
	ulonglong2 mine = make_ulonglong2(0, 0);
	for (int base = 0; base < n_u; base += 64) {
		const int i = base + lane;
		const unsigned long long ui = i < n_u ? u[c0 + i] : 0;
		const int cnt = (int)(int32_t)ui;
		int incl = cnt;
		for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
		const int k = carry + incl - cnt;                                   // hit.c:63,67: the chain's first anchor
		carry += __shfl(incl, 63);
		if (i < n_u) {
			const ulonglong2 f = a[k];
			const uint32_t h = (uint32_t)regs_hash64((regs_hash64(f.x) + regs_hash64(f.y)) ^ hs);
			mine.x = ui ^ h;
			mine.y = (unsigned long long)(uint32_t)k << 32 | (uint32_t)cnt;
			if (n_u > 64) z[c0 + i] = mine;
		}
	}
	if (n_u <= 64) {                                                        // ksort.h:107-117: stable
		int rank = 0;
		for (int j = 0; j < n_u; ++j) {
			const unsigned long long xj = readlane_u64(mine.x, j);
			rank += (xj < mine.x) | (xj == mine.x & j < lane);
		}
		if (lane < n_u) z[c0 + rank] = mine;
	} else {
		wave_global_fence();
		if (lane == 0) bt_radix_128x(z + c0, n_u, stacks + c0 / 64 + 2 * r);
	}
}

__global__ __launch_bounds__(256) void k_regs_fill(int64_t n_reads, const int64_t *__restrict__ chains_off, const int64_t *__restrict__ b_off,
                                                   const ulonglong2 *__restrict__ b, const int32_t *__restrict__ qlen,
                                                   const ulonglong2 *__restrict__ z, int32_t *__restrict__ regs)
{
	const int lane = threadIdx.x & 63;
	const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (r >= n_reads) return;
	const int64_t c0 = chains_off[r];
	const int n_u = (int)(chains_off[r + 1] - c0);
	const ulonglong2 *a = b + b_off[r];
	const int ql = qlen[r];
	for (int i = 0; i < n_u; ++i) {                                         // the wave takes the read's hits one at a time, its lanes the anchors
		const ulonglong2 zi = z[c0 + (n_u - 1 - i)];                        // hit.c:70-71: larger score first
		const int cnt = (int)(int32_t)zi.y, as = (int)(zi.y >> 32), score = (int)(zi.x >> 32);
		const ulonglong2 f = a[as], l = a[as + cnt - 1];
		const int q_span = (int)(f.y >> 32 & 0xff), rev = (int)(f.x >> 63);
		int mlen = 0, blen = 0;                                             // hit.c:8-22, one term per anchor after the first
		for (int k = as + 1 + lane; k < as + cnt; k += 64) {
			const ulonglong2 cur = a[k], prev = a[k - 1];
			const int span = (int)(cur.y >> 32 & 0xff);
			const int tl = (int)(int32_t)cur.x - (int)(int32_t)prev.x, qd = (int)(int32_t)cur.y - (int)(int32_t)prev.y;
			blen += tl > qd ? tl : qd;
			mlen += tl > span && qd > span ? span : tl < qd ? tl : qd;
		}
		for (int d = 32; d > 0; d >>= 1) { mlen += __shfl_xor(mlen, d); blen += __shfl_xor(blen, d); }
		mlen += q_span; blen += q_span;
		if (lane != 0) continue;
		int32_t *o = regs + (c0 + i) * 20;                                  // mm_reg1_t, minimap.h:100-115 (80 B)
		o[0] = i; o[1] = cnt; o[2] = (int32_t)(f.x << 1 >> 33); o[3] = score;
		if (!rev) { o[4] = (int32_t)f.y + 1 - q_span; o[5] = (int32_t)l.y + 1; }                       // hit.c:32-34
		else { o[4] = ql - ((int32_t)l.y + 1); o[5] = ql - ((int32_t)f.y + 1 - q_span); }               // hit.c:35-37
		o[6] = (int32_t)f.x + 1 > q_span ? (int32_t)f.x + 1 - q_span : 0; o[7] = (int32_t)l.x + 1;     // hit.c:29-30
		o[8] = -1; o[9] = 0; o[10] = as; o[11] = mlen; o[12] = blen; o[13] = 0; o[14] = score;         // parent = MM_PARENT_UNSET
		o[15] = rev << 10; o[16] = (int32_t)(uint32_t)zi.x; o[17] = __float_as_int(-1.0f); o[18] = 0; o[19] = 0;
	}
}

// esterr.c:7-14
__device__ __forceinline__ int regs_for_qpos(int qlen, ulonglong2 a)
{
	int x = (int)(int32_t)a.y;
	const int q_span = (int)(a.y >> 32 & 0xff);
	if (a.x >> 63) x = qlen - 1 - (x + 1 - q_span);
	return x;
}

// sum of the minimizers' spans per read (esterr.c:38-40), a wave per read
__global__ __launch_bounds__(256) void k_regs_span_sum(int64_t n_reads, const int64_t *__restrict__ mp_off, const unsigned long long *__restrict__ mini_pos,
                                                       unsigned long long *__restrict__ sum_k)
{
	const int lane = threadIdx.x & 63;
	const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (r >= n_reads) return;
	unsigned long long s = 0;
	for (int64_t i = mp_off[r] + lane; i < mp_off[r + 1]; i += 64) s += mini_pos[i] >> 32 & 0xff;
	for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
	if (lane == 0) sum_k[r] = s;
}

__global__ __launch_bounds__(256) void k_regs_div(int64_t n_reads, int64_t n_regs, const int64_t *__restrict__ regs_off, const int64_t *__restrict__ b_off,
                                                  const ulonglong2 *__restrict__ b, const int32_t *__restrict__ qlen, const int32_t *__restrict__ ref_len,
                                                  int32_t n_ref, const int64_t *__restrict__ mp_off, const unsigned long long *__restrict__ mini_pos,
                                                  const unsigned long long *__restrict__ sum_k, int32_t *__restrict__ regs, int32_t *__restrict__ counts)
{
	const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_regs) return;
	int64_t lo = 0, hi = n_reads - 1;                                       // the read of hit g
	while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (regs_off[mid] <= g) lo = mid; else hi = mid - 1; }
	const int64_t r = lo;
	int32_t *o = regs + g * 20;
	const int n = (int)(mp_off[r + 1] - mp_off[r]);
	if (counts) { counts[2 * g] = 0; counts[2 * g + 1] = 0; }
	if (n == 0) return;                                                     // esterr.c:37: the hits keep the div they came with
	o[17] = __float_as_int(-1.0f);
	const int cnt = o[1];
	if (cnt == 0) return;
	const unsigned long long *mp = mini_pos + mp_off[r];
	const ulonglong2 *a = b + b_off[r];
	const int ql = qlen[r], as = o[10], rev = o[15] >> 10 & 1;
	const float avg_k = (float)sum_k[r] / n;
	int x = regs_for_qpos(ql, rev ? a[as + cnt - 1] : a[as]);
	int st = -1, L = 0, R = n - 1;
	while (L <= R) {                                                        // esterr.c:16-28
		const int m = (int)(((unsigned long long)L + R) >> 1), y = (int)(int32_t)mp[m];
		if (y < x) L = m + 1;
		else if (y > x) R = m - 1;
		else { st = m; break; }
	}
	if (st < 0) return;
	int en = st, k = 1, n_match = 1;
	x = k < cnt ? regs_for_qpos(ql, rev ? a[as + cnt - 1 - k] : a[as + k]) : 0;
	for (int j = st + 1; j < n && k < cnt; ++j) {                           // esterr.c:53-58
		if (x == (int)(int32_t)mp[j]) {
			++k; en = j; ++n_match;
			if (k < cnt) x = regs_for_qpos(ql, rev ? a[as + cnt - 1 - k] : a[as + k]);
		}
	}
	int n_tot = en - st + 1;
	const int rid = o[2], l_ref = rid >= 0 && rid < n_ref ? ref_len[rid] : 0;
	if (o[4] > avg_k && o[6] > avg_k) ++n_tot;
	if (ql - o[4] > avg_k && l_ref - o[7] > avg_k) ++n_tot;
	o[17] = __float_as_int(logf((float)n_tot / n_match) / avg_k);
	if (counts) { counts[2 * g] = n_match; counts[2 * g + 1] = n_tot; }
}

hipError_t launch_gen_regs(hipStream_t st, int64_t n_reads, const int64_t *d_chains_off, const int64_t *d_b_off, const unsigned long long *d_u,
                           const void *d_b, const uint32_t *d_hash, const int32_t *d_qlen, void *d_z, void *d_stacks, void *d_regs)
{
	if (n_reads <= 0) return hipSuccess;
	const unsigned grid = (unsigned)((n_reads * 64 + 255) / 256);
	hipLaunchKernelGGL(k_regs_keys, dim3(grid), dim3(256), 0, st, n_reads, d_chains_off, d_b_off, d_u, (const ulonglong2*)d_b, d_hash,
	                   (ulonglong2*)d_z, (BtRange*)d_stacks);
	hipLaunchKernelGGL(k_regs_fill, dim3(grid), dim3(256), 0, st, n_reads, d_chains_off, d_b_off, (const ulonglong2*)d_b, d_qlen,
	                   (const ulonglong2*)d_z, (int32_t*)d_regs);
	return hipGetLastError();
}

hipError_t launch_est_err(hipStream_t st, int64_t n_reads, int64_t n_regs, const int64_t *d_regs_off, const int64_t *d_b_off, const void *d_b,
                          const int32_t *d_qlen, const int32_t *d_ref_len, int32_t n_ref, const int64_t *d_mp_off, const unsigned long long *d_mini_pos,
                          unsigned long long *d_sum_k, void *d_regs, int32_t *d_counts)
{
	if (n_reads <= 0 || n_regs <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_regs_span_sum, dim3((unsigned)((n_reads * 64 + 255) / 256)), dim3(256), 0, st, n_reads, d_mp_off, d_mini_pos, d_sum_k);
	hipLaunchKernelGGL(k_regs_div, dim3((unsigned)((n_regs + 255) / 256)), dim3(256), 0, st, n_reads, n_regs, d_regs_off, d_b_off, (const ulonglong2*)d_b,
	                   d_qlen, d_ref_len, n_ref, d_mp_off, d_mini_pos, d_sum_k, (int32_t*)d_regs, d_counts);
	return hipGetLastError();
}

} // namespace chaindp
