// chaindp_seed.hip -- seed collection on the GPU: what the reference's device did with a read's
// minimizers before chaining (collect_seed_hits, map.c:187-236, called from fpga_work, map.c:523), over the index
// image the host streams through fpga_load_index (index.c:603-720).
//
//   k_seed_probe   thread per minimizer: mm_idx_get over the image (khash probe, khash.h:218-231), the
//                  "too frequent" test and tandem flag of collect_matches (map.c:112-146), and the number of its
//                  hits that survive skip_seed (map.c:148-185)
//   (two exclusive scans: anchors before each minimizer, used minimizers before each minimizer)
//   k_seed_expand  thread per minimizer: writes its surviving hits as anchors (map.c:197-231), in the reference's
//                  generation order, and its mini_pos entry
//   k_seed_reads   wave per read: anchor / mini_pos offsets of the read and rep_len (the interval merge of
//                  map.c:127-133, one term per skipped minimizer)
//   k_seed_sort    workgroup per read, in LDS: the order radix_sort_128x (ksort.h:101-151) gives equal x is input to the
//                  chaining DP, so a read with equal keys is sorted by the reference's procedure step by step; a
//                  read without (the common case) by a bitonic network.
//   k_seed_sort_huge  workgroup per read too large for LDS: top levels in global memory, buckets back to k_seed_sort
// Image layout: see SeedIndex in chaindp_kernels.h.  A CPU restatement of the same lookup, pinned against the reference, is
// the checker of these kernels (oracle/seed_oracle.cpp; test infrastructure, not part of this library).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "chaindp_kernels.h"
#include "chaindp_rsort.h"
#include "chaindp_wave.h"

namespace chaindp {

#define SEED_P_STRAND (1ull << 21)              // mmpriv.h:20
#define SEED_TANDEM_BIT (1ull << 42)            // mmpriv.h:18
#define SEED_SELF_BIT (1ull << 43)              // mmpriv.h:19
#define SEED_SEG_SHIFT 48                       // mmpriv.h:22
#define SEED_F_NO_DIAG 0x001
#define SEED_F_NO_DUAL 0x002
#define SEED_F_FOR_ONLY 0x100000
#define SEED_F_REV_ONLY 0x200000

__device__ __forceinline__ uint64_t ld_u64(const uint8_t *p) { return *(const uint64_t*)p; }      // all blobs are 8-byte aligned at these offsets

// 6-byte key at an even address (group base is 64-aligned, keys start at +4, 6 bytes each)
__device__ __forceinline__ uint64_t ld_u48(const uint8_t *p)
{
	const uint16_t *q = (const uint16_t*)p;
	return (uint64_t)q[0] | (uint64_t)q[1] << 16 | (uint64_t)q[2] << 32;
}

// mm_idx_get over the image (index.c:221-238): *n hits; returns their location as an index into V (single hit,
// bit 63 set) or into P
__device__ __forceinline__ uint64_t seed_lookup(const SeedIndex &ix, uint64_t minier, int *n)
{
	*n = 0;
	const uint64_t mask = (1ull << ix.b_bits) - 1;
	const uint8_t *be = ix.B + (minier & mask) * 16;
	const uint64_t w0 = ld_u64(be), w1 = ld_u64(be + 8);
	const uint32_t n_buckets = (uint32_t)(w0 >> 24);
	if (n_buckets == 0) return 0;
	const uint64_t h_off = w1 >> 28, p_off = (w1 & ((1ull << 28) - 1)) << 8 | w0 >> 56;
	const uint64_t key = minier >> ix.b_bits << 1;
	const uint32_t m = n_buckets - 1;
	uint32_t i = (uint32_t)(key >> 1) & m, step = 0;
	const uint32_t last = i;
	for (;;) {
		const uint64_t slot = h_off + i;
		if ((slot >> 3) * 64 + 64 > ix.nH) return 0;
		const uint8_t *grp = ix.H + (slot >> 3) * 64;
		const uint32_t fl = (*(const uint32_t*)grp >> ((i & 0xfu) << 1)) & 3u;
		if (fl & 2u) return 0;                                            // empty: absent
		const uint64_t k48 = ld_u48(grp + 4 + (slot & 7) * 6);
		if (!(fl & 1u) && (k48 >> 1) == ((key & 0xffffffffffffull) >> 1)) {
			if ((slot + 1) * 8 > ix.nV) return 0;
			if (k48 & 1) { *n = 1; return 1ull << 63 | slot; }
			const uint64_t v = ld_u64(ix.V + slot * 8);
			const uint64_t first = p_off + (v >> 32);
			const uint32_t cnt = (uint32_t)v;
			if ((first + cnt) * 8 > ix.nP) return 0;
			*n = (int)cnt;
			return first;
		}
		i = (i + (++step)) & m;
		if (i == last) return 0;
	}
}

__device__ __forceinline__ const uint64_t *seed_hits(const SeedIndex &ix, uint64_t src)
{
	return (src >> 63) ? (const uint64_t*)(ix.V + (src & ~(1ull << 63)) * 8) : (const uint64_t*)(ix.P + src * 8);
}

// skip_seed, map.c:148-185 (the block at :152 opens on bit 0 of flag only, as written)
__device__ __forceinline__ bool seed_skip(int flag, uint64_t r, uint32_t q_pos, uint32_t bid, bool *is_self)
{
	*is_self = false;
	if (1 & flag & (SEED_F_NO_DIAG | SEED_F_NO_DUAL)) {
		const uint32_t rank_id = (uint32_t)r & 0x1FFFFFu, val = bid & 0x7fffffffu;
		const int cmp = val > rank_id ? 1 : val < rank_id ? -1 : (bid >> 31) ? 0 : -1;
		if ((flag & SEED_F_NO_DIAG) && cmp == 0) {
			if (((r >> 22) & 0x1fffff) == (q_pos >> 1)) return true;
			if (((r & SEED_P_STRAND) >> 21) == (q_pos & 1)) *is_self = true;
		}
		if ((flag & SEED_F_NO_DUAL) && cmp > 0) return true;
	}
	if (flag & (SEED_F_FOR_ONLY | SEED_F_REV_ONLY)) {
		if (((r & SEED_P_STRAND) >> 21) == (q_pos & 1)) { if (flag & SEED_F_REV_ONLY) return true; }
		else { if (flag & SEED_F_FOR_ONLY) return true; }
	}
	return false;
}

// largest r in [lo, hi] with mini_off[r] <= i (the answer is known to lie there)
__device__ __forceinline__ int64_t seed_read_between(const int64_t *__restrict__ mini_off, int64_t lo, int64_t hi, int64_t i)
{
	while (lo < hi) {
		const int64_t mid = (lo + hi + 1) >> 1;
		if (mini_off[mid] <= i) lo = mid; else hi = mid - 1;
	}
	return lo;
}

// The read of minimizer i, for a workgroup of 256 consecutive minimizers: one thread searches all reads for the
// workgroup's first and last minimizer, the others only between those two (a workgroup spans one or two reads, so
// the thirteen dependent loads of a full search become none or one).  Every thread of the workgroup must call it.
__device__ __forceinline__ int64_t seed_read_of_block(const int64_t *__restrict__ mini_off, int64_t n_reads, int64_t n_mini, int64_t i)
{
	__shared__ int64_t s_first, s_last;
	if (threadIdx.x == 0) {
		const int64_t i0 = (int64_t)blockIdx.x * blockDim.x, i1 = i0 + blockDim.x - 1 < n_mini - 1 ? i0 + blockDim.x - 1 : n_mini - 1;
		s_first = seed_read_between(mini_off, 0, n_reads - 1, i0);
		s_last = seed_read_between(mini_off, s_first, n_reads - 1, i1);
	}
	__syncthreads();
	return seed_read_between(mini_off, s_first, s_last, i < n_mini ? i : n_mini - 1);
}

// per minimizer: mstate = hits (low 32 bits) | used << 32 | tandem << 33; src = where its hits are
__global__ __launch_bounds__(256) void k_seed_probe(SeedIndex ix, int flag, int max_occ, int64_t n_reads, int64_t n_mini,
                                                    const int64_t *__restrict__ mini_off, const ulonglong2 *__restrict__ mini,
                                                    const uint32_t *__restrict__ bid, unsigned long long *__restrict__ kept,
                                                    unsigned long long *__restrict__ used, unsigned long long *__restrict__ src,
                                                    unsigned long long *__restrict__ mstate)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t r = seed_read_of_block(mini_off, n_reads, n_mini, i);
	if (i >= n_mini) return;
	const ulonglong2 p = mini[i];
	int t;
	const uint64_t where = seed_lookup(ix, p.x >> 8, &t);
	unsigned long long k = 0, st = 0;
	if (t < max_occ) {                                                    // map.c:125: a used minimizer (also when it has no hit)
		const bool tandem = (i > mini_off[r] && p.x >> 8 == mini[i - 1].x >> 8) || (i + 1 < mini_off[r + 1] && p.x >> 8 == mini[i + 1].x >> 8);
		const uint64_t *cr = t ? seed_hits(ix, where) : nullptr;
		const uint32_t q_pos = (uint32_t)p.y, b = bid[r];
		for (int h = 0; h < t; ++h) { bool self; if (!seed_skip(flag, cr[h], q_pos, b, &self)) ++k; }
		st = (unsigned long long)(uint32_t)t | 1ull << 32 | (unsigned long long)tandem << 33;
	}
	kept[i] = k; used[i] = st >> 32 & 1; src[i] = where; mstate[i] = st;
}

__global__ __launch_bounds__(256) void k_seed_expand(SeedIndex ix, int flag, int64_t n_reads, int64_t n_mini,
                                                     const int64_t *__restrict__ mini_off, const ulonglong2 *__restrict__ mini,
                                                     const uint32_t *__restrict__ bid, const int32_t *__restrict__ qlen,
                                                     const unsigned long long *__restrict__ kept_pos, const unsigned long long *__restrict__ used_pos,
                                                     const unsigned long long *__restrict__ src, const unsigned long long *__restrict__ mstate,
                                                     ulonglong2 *__restrict__ a, unsigned long long *__restrict__ mini_pos)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t r = seed_read_of_block(mini_off, n_reads, n_mini, i);
	if (i >= n_mini) return;
	const unsigned long long st = mstate[i];
	if (!(st >> 32 & 1)) return;
	const ulonglong2 p = mini[i];
	const uint32_t q_pos = (uint32_t)p.y, q_span = (uint32_t)(p.x & 0xff);
	mini_pos[used_pos[i]] = (unsigned long long)q_span << 32 | q_pos >> 1;               // map.c:141
	const int t = (int)(uint32_t)st;
	if (t == 0) return;
	const uint32_t b = bid[r], ql = (uint32_t)qlen[r];
	const uint64_t seg = (uint64_t)((uint32_t)(p.y >> 32) & 0x7fffffffu) << SEED_SEG_SHIFT;
	const uint64_t extra = seg | ((st >> 33 & 1) ? SEED_TANDEM_BIT : 0);
	const uint64_t *cr = seed_hits(ix, src[i]);
	unsigned long long o = kept_pos[i];
	for (int h = 0; h < t; ++h) {
		const uint64_t rr = cr[h];
		bool self;
		if (seed_skip(flag, rr, q_pos, b, &self)) continue;
		const uint64_t rpos = (rr >> 22) & 0x1fffff;
		ulonglong2 s;
		if (((rr & SEED_P_STRAND) >> 21) == (q_pos & 1)) {                               // forward strand, map.c:216-218
			s.x = ((rr & 0xfffff80000000000ull) >> 11) | rpos;
			s.y = (uint64_t)q_span << 32 | q_pos >> 1;
		} else {                                                                         // reverse strand, map.c:220-222 (32-bit unsigned arithmetic)
			s.x = 1ull << 63 | ((rr & 0xfffff80000000000ull) >> 11) | rpos;
			s.y = (uint64_t)q_span << 32 | (uint32_t)(ql - ((q_pos >> 1) + 1 - q_span) - 1);
		}
		s.y |= extra;
		if (self) s.y |= SEED_SELF_BIT;
		a[o++] = s;
	}
}

// One wave per read.  rep_len (map.c:116,127-133,143) is the length of the union of the skipped minimizers' intervals,
// kept by the reference as (rep_st, rep_en) with rep_en always the end of the last skipped minimizer: a minimizer
// starts a new interval when its start lies beyond the previous skipped minimizer's end, and the sum telescopes to
// sum_i (en_i - (st_i > en_prev ? st_i : en_prev)) with en_prev = 0 before the first -- one term per minimizer.
__global__ __launch_bounds__(256) void k_seed_reads(int64_t n_reads, int64_t n_mini, const int64_t *__restrict__ mini_off,
                                                    const ulonglong2 *__restrict__ mini, const unsigned long long *__restrict__ kept_pos,
                                                    const unsigned long long *__restrict__ used_pos, const unsigned long long *__restrict__ mstate,
                                                    const unsigned long long *__restrict__ totals, int64_t *__restrict__ off,
                                                    int64_t *__restrict__ mp_off, int32_t *__restrict__ rep_len)
{
	const int lane = threadIdx.x & 63;
	const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (r > n_reads) return;
	if (r == n_reads) { if (lane == 0) { off[r] = (int64_t)totals[0]; mp_off[r] = (int64_t)totals[1]; } return; }
	const int64_t b = mini_off[r], e = mini_off[r + 1];
	if (lane == 0) {
		off[r] = b < n_mini ? (int64_t)kept_pos[b] : (int64_t)totals[0];
		mp_off[r] = b < n_mini ? (int64_t)used_pos[b] : (int64_t)totals[1];
	}
	int prev_en = 0, sum = 0;
	for (int64_t base = b; base < e; base += 64) {
		const int64_t i = base + lane;
		bool skipped = false;
		int en = 0, st = 0;
		if (i < e && !(mstate[i] >> 32 & 1)) {
			const ulonglong2 p = mini[i];
			skipped = true;
			en = (int)((uint32_t)p.y >> 1) + 1; st = en - (int)(p.x & 0xff);
		}
		const unsigned long long m = __ballot(skipped);
		if (m == 0) continue;
		const unsigned long long below = m & ((1ull << lane) - 1);
		int pe = __shfl(en, below ? 63 - __clzll((long long)below) : 0);
		if (!below) pe = prev_en;
		if (skipped) sum += st > pe ? en - st : en - pe;
		prev_en = __shfl(en, 63 - __clzll((long long)m));
	}
	for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
	if (lane == 0) rep_len[r] = sum;
}

// Reads too large even for k_seed_sort_huge below: the same procedure by one thread in global memory (slow; millions of anchors in one read).
__global__ __launch_bounds__(64) void k_seed_sort_big(int64_t n_reads, int max_n, const int64_t *__restrict__ off, const ulonglong2 *__restrict__ src,
                                                      ulonglong2 *__restrict__ a, BtRange *__restrict__ stacks)
{
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	const int64_t b = off[r];
	const int64_t n = off[r + 1] - b;
	if (n <= max_n) return;
	for (int64_t i = 0; i < n; ++i) a[b + i] = src[b + i];
	bt_radix_128x(a + b, (int32_t)n, stacks + b / 64 + 2 * r);                           // map.c:233
}

// Reads above what the LDS sort takes (more than max_n2 anchors): their top levels, one workgroup per read.
// A level of the reference's sort (ksort.h:126-141) reads every element exactly once, at a bucket's head pointer, before
// anything was stored there -- so the sequence of swaps depends on the DIGITS of the elements at their original places
// only.  The workgroup writes those digits to LDS (one byte per anchor) and counts them, one lane replays the
// reference's swap loop over the bytes alone and notes for every position which element ends there, and the
// workgroup then moves the 16-byte anchors in one parallel pass.  Buckets that fit the LDS sort become work items for
// it (k_seed_sort below takes them after the reads, at their shift), buckets of up to 64 anchors are insertion-sorted
// by the thread that owns them (ksort.h:148), larger ones go round again.  Pending large ranges are disjoint and
// each longer than min_n, so SEED_HUGE_STACK slots hold them for reads of up to SEED_HUGE_STACK x min_n anchors;
// k_seed_sort_big keeps the rest.  A range with more anchors than LDS has bytes keeps its digits in global memory
// (packed eight to a word in the scratch copy's y fields): same walk, global-memory latency per step.
struct SeedItem { uint32_t beg_lo, len, beg_hi_shift; };           // anchors [beg, beg + len) of d_unsorted, to be sorted from `shift` down
#define SEED_HUGE_STACK 320

template <bool GLOBAL_DIGITS>
__device__ __forceinline__ void seed_huge_walk(const uint8_t *lab, int *head, const int *tail, ulonglong2 *a_read, int rb)
{
	auto digit = [&](int pos) -> int {
		const int i = pos - rb;
		if (GLOBAL_DIGITS) return (int)(a_read[rb + (i >> 3)].y >> ((i & 7) * 8) & 0xff);
		return (int)lab[i];
	};
	for (int d = 0; d < 256; ++d) {                                     // ksort.h:129-141 on digits; a_read[q].x = source of position q
		int hd = head[d];
		const int td = tail[d];
		while (hd != td) {
			int l = digit(hd);
			if (l != d) {
				int carry = hd;
				do {
					const int hp = head[l];
					a_read[hp].x = (uint64_t)carry;
					carry = hp;
					head[l] = hp + 1;
					l = digit(hp);
				} while (l != d);
				a_read[hd].x = (uint64_t)carry;
			} else a_read[hd].x = (uint64_t)hd;
			++hd;
		}
	}
}

__global__ __launch_bounds__(256) void k_seed_sort_huge(int64_t n_reads, int min_n, int max_n, int lab_cap, const int64_t *__restrict__ off,
                                                        ulonglong2 *__restrict__ w, ulonglong2 *__restrict__ a, SeedItem *__restrict__ items,
                                                        unsigned long long *__restrict__ n_items)
{
	extern __shared__ uint8_t huge_lds[];
	uint8_t *lab = huge_lds;
	int *head = (int*)(huge_lds + lab_cap);
	int *tail = head + 256, *start = tail + 256;
	unsigned int *hist = (unsigned int*)(start + 256);
	int *stack = (int*)(hist + 256);                                   // SEED_HUGE_STACK x (beg, end, shift)
	int *sp = stack + 3 * SEED_HUGE_STACK;
	const int tid = threadIdx.x;
	for (int64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
		const int64_t b = off[r];
		const int64_t n64 = off[r + 1] - b;
		if (n64 <= min_n || n64 > max_n) continue;
		const int n = (int)n64;
		ulonglong2 *wr = w + b, *ar = a + b;                            // wr: the read being sorted; ar: scratch during a level, a copy of wr between levels
		for (int i = tid; i < n; i += 256) ar[i] = wr[i];
		__syncthreads();
		if (tid == 0) { stack[0] = 0; stack[1] = n; stack[2] = 56; sp[0] = 1; }
		__syncthreads();
		for (;;) {
			const int top = sp[0];
			if (top == 0) break;
			const int rb = stack[3 * (top - 1)], re = stack[3 * (top - 1) + 1], sh = stack[3 * (top - 1) + 2];
			const int len = re - rb;
			const bool global_digits = len > lab_cap;
			__syncthreads();
			if (tid == 0) sp[0] = top - 1;
			hist[tid] = 0;
			__syncthreads();
			if (!global_digits) {
				for (int i = tid; i < len; i += 256) {
					const int d = (int)(wr[rb + i].x >> sh & 0xff);
					lab[i] = (uint8_t)d;
					atomicAdd(&hist[d], 1u);                            // ksort.h:126
				}
			} else {
				for (int g = tid; g * 8 < len; g += 256) {
					uint64_t pack = 0;
					for (int j = 0; j < 8 && g * 8 + j < len; ++j) {
						const uint64_t d = wr[rb + g * 8 + j].x >> sh & 0xff;
						pack |= d << (8 * j);
						atomicAdd(&hist[(int)d], 1u);
					}
					ar[rb + g].y = pack;
				}
			}
			__syncthreads();
			const int next = sh > 8 ? sh - 8 : 0;
			if (hist[(int)(wr[rb].x >> sh & 0xff)] == (unsigned int)len) {  // one bucket: the pass moves nothing
				if (global_digits) for (int g = tid; g * 8 < len; g += 256) ar[rb + g].y = wr[rb + g].y;
				if (tid == 0 && sh) { const int k = sp[0]; stack[3 * k] = rb; stack[3 * k + 1] = re; stack[3 * k + 2] = next; sp[0] = k + 1; }
				__syncthreads();
				continue;
			}
			if (tid == 0) {
				int acc = rb, n_digits = 0, d_lo = 256, d_hi = -1;
				for (int d = 0; d < 256; ++d) {
					head[d] = acc; start[d] = acc; acc += (int)hist[d]; tail[d] = acc;
					if (hist[d]) { ++n_digits; if (d_lo == 256) d_lo = d; d_hi = d; }
				}
				sp[1] = n_digits; sp[2] = d_lo; sp[3] = d_hi;
			}
			__syncthreads();
			if (sp[1] == 2 && !global_digits) {
				// two buckets (the strand level): closed form, see k_seed_sort; lists in the .y fields, by the first wave
				if (tid < 64) {
					const int d_lo = sp[2], d_hi = sp[3], mid = tail[d_lo];
					const unsigned long long below = (1ull << tid) - 1;
					int run = 0;
					for (int base = rb; base < mid; base += 64) {
						const int q = base + tid;
						const bool f = q < mid && lab[q - rb] == d_hi;
						const unsigned long long m = __ballot(f);
						if (f) ar[rb + run + __popcll(m & below)].y = (uint64_t)q;
						run += __popcll(m);
					}
					const int n_cycles = run;
					run = 0;
					for (int base = mid; base < re; base += 64) {
						const int q = base + tid;
						const bool f = q < re && lab[q - rb] == d_lo;
						const unsigned long long m = __ballot(f);
						if (f) ar[mid + run + __popcll(m & below)].y = (uint64_t)q;
						run += __popcll(m);
					}
					wave_global_fence();
					run = 0;
					for (int base = rb; base < mid; base += 64) {
						const int q = base + tid;
						const bool f = q < mid && lab[q - rb] == d_hi;
						const unsigned long long m = __ballot(f);
						if (q < mid) ar[q].x = f ? ar[mid + run + __popcll(m & below)].y : (uint64_t)q;
						run += __popcll(m);
					}
					run = 0;
					for (int base = mid; base < re; base += 64) {
						const int q = base + tid;
						const bool f = q < re && lab[q - rb] == d_lo;
						const unsigned long long m = __ballot(f);
						const int before = run + __popcll(m & below);
						if (q < re) {
							uint64_t from = (uint64_t)q;
							if (before < n_cycles) from = (q == mid || lab[q - 1 - rb] == d_lo) ? ar[rb + before].y : (uint64_t)(q - 1);
							ar[q].x = from;
						}
						run += __popcll(m);
					}
				}
			} else if (tid == 0) {
				if (global_digits) seed_huge_walk<true>(lab, head, tail, ar, rb);
				else seed_huge_walk<false>(lab, head, tail, ar, rb);
			}
			__syncthreads();
			for (int i = rb + tid; i < re; i += 256) ar[i] = wr[(int)ar[i].x];
			__syncthreads();
			for (int i = rb + tid; i < re; i += 256) wr[i] = ar[i];
			if (sh) {                                                       // ksort.h:143-149
				const int sb = start[tid], se = tail[tid], l = se - sb;
				if (l > min_n) {
					const int k = atomicAdd(&sp[0], 1);
					stack[3 * k] = sb; stack[3 * k + 1] = se; stack[3 * k + 2] = next;
				} else if (l > 64) {
					const unsigned long long k = atomicAdd(n_items, 1ull);
					const uint64_t gb = (uint64_t)(b + sb);
					items[k] = SeedItem{(uint32_t)gb, (uint32_t)l, (uint32_t)(gb >> 32) << 8 | (uint32_t)next};
				}
			}
			__syncthreads();
			if (sh) {                                                       // small buckets after the moves above are visible
				const int sb = start[tid], se = tail[tid], l = se - sb;
				if (l > 1 && l <= 64) {
					bt_insertion(wr + sb, wr + se);
					for (int i = sb; i < se; ++i) ar[i] = wr[i];
				}
			}
			__syncthreads();
		}
	}
}

// radix_sort_128x (ksort.h:101-151) of one read (or one bucket handed over by k_seed_sort_huge) by one workgroup of sixteen
// waves, in LDS, on (key, original index) pairs.  A read whose keys all differ -- almost every read -- has a unique
// sorted order and is done by a bitonic network.  Otherwise the reference's procedure is followed level by level:
// ranges of more than 64 anchors go to the waves one at a time (digit counts by the wave, the bucket permutation
// replayed over the digits by one lane, keys and indices moved by the wave), ranges of up to 64 are sorted a wave each
// (the insertion sort is stable: a rank by key and place), and a level on which every key of a range has the same digit is skipped (the reference's pass
// over it moves nothing).  Positions are 16-bit: a read here has at most max_n2 <= 65535 anchors.
// LDS: keys[n] u64 | idx[n] u16 | W x (head, tail, start)[256] u16 | W x counts[256] u32 | two queues of ranges |
// 4 counters | digits[n] u8, with n = max_n (<= 8192, W = 16 waves with tables, 1024 queue slots for small ranges) or
// max_n2 (~13 k, W = 4, 256 slots),
// chosen per read inside one launch.
// A queue keeps its first n / 65 + 2 slots for ranges of more than 64 anchors (they are disjoint, so they always fit);
// small ranges that find the rest full are insertion-sorted on the spot by the lane that made them.
#define SEED_TPB_C 1024                 // threads of the per-read sort's workgroup (SEED_TPB below)
struct SeedRange { uint16_t beg, end; uint16_t shift, pad; };
__host__ __device__ inline int seed_big_slots(int max_n) { return max_n / 65 + 2; }                    // queue slots for ranges of > 64 anchors: they are disjoint
__host__ __device__ inline int seed_small_slots(int workers) { return workers >= 32 ? 1024 : 256; }   // queue slots for ranges of <= 64 anchors

__device__ __forceinline__ void seed_isort(uint64_t *key, uint16_t *idx, int beg, int end)    // ksort.h:107-117
{
	for (int i = beg + 1; i < end; ++i) {
		if (key[i] < key[i - 1]) {
			const uint64_t tk = key[i]; const uint16_t ti = idx[i];
			int j = i;
			while (j > beg && tk < key[j - 1]) { key[j] = key[j - 1]; idx[j] = idx[j - 1]; --j; }
			key[j] = tk; idx[j] = ti;
		}
	}
}

// Least-significant-digit radix sort of the n <= I * 1024 words in key[] (LDS) by their bits [lo_bit, lo_bit + bits), ascending and
// stable, by the whole 1024-thread workgroup.  Four bits a pass.  Thread t holds the I consecutive words t * I .. (those below n):
// it counts its words per digit in a register of sixteen nibbles (a word's rank among the thread's own words of that digit is the
// nibble before the increment), the counts are scanned over the workgroup -- two 16-bit fields per register, wave prefix by DPP
// adds, wave totals through LDS, then every wave works out its own sixteen bases from the 16 x 16 totals (sixteen lanes, a row scan,
// the results wave-uniform) --, every word goes to its place, and the threads read their next I consecutive words back.  No atomics;
// the order of equal digits is the order of places, as a stable sort needs.  Two barriers a pass.  tbl: 16 x 1024 u16 (a thread's
// sixteen bases, digit-major), tot: 16 x 16 u16.  All 1024 threads call it.
#define SEED_RDX_BITS 4
// (inlined on purpose: as a function of its own it gets key / tbl / tot as generic pointers and accesses LDS with FLAT instructions,
// and that build now and then left two equal neighbours in a tie-free read of more than 8192 anchors -- harmless, the read then
// went through the reference's procedure, but wrong)
template <int I>
__device__ __forceinline__ void seed_radix_words(uint64_t *key, const int n, const int lo_bit, const int bits, uint16_t *tbl, uint16_t *tot)
{
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	uint64_t k[I];
#pragma unroll
	for (int e = 0; e < I; ++e) k[e] = t * I + e < n ? key[t * I + e] : 0;
	const int mine = n - t * I < 0 ? 0 : n - t * I < I ? n - t * I : I;   // words this thread holds
	for (int sh = lo_bit; sh < lo_bit + bits; sh += SEED_RDX_BITS) {
		uint64_t cnt = 0;                                                // sixteen nibbles: this thread's words per digit (I <= 15)
		uint32_t dg[I], lr[I];
#pragma unroll
		for (int e = 0; e < I; ++e) {
			dg[e] = (uint32_t)(k[e] >> sh) & 15u;
			lr[e] = (uint32_t)(cnt >> (4 * dg[e])) & 15u;
			cnt += (e < mine ? 1ull : 0ull) << (4 * dg[e]);
		}
		uint32_t c2[8], own[8];                                          // digits 2 j (low half) and 2 j + 1 (high half)
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const uint32_t b = (uint32_t)(cnt >> (8 * j)) & 0xffu;
			own[j] = (b & 15u) | (b >> 4) << 16;
			uint32_t v = own[j];                                         // inclusive prefix over the wave (no field passes 13 312)
			v += (uint32_t)dpp_or_old<DPP_ROW_SHR(1), 0xf>(0, (int)v);
			v += (uint32_t)dpp_or_old<DPP_ROW_SHR(2), 0xf>(0, (int)v);
			v += (uint32_t)dpp_or_old<DPP_ROW_SHR(4), 0xf>(0, (int)v);
			v += (uint32_t)dpp_or_old<DPP_ROW_SHR(8), 0xf>(0, (int)v);
			v += (uint32_t)dpp_or_old<DPP_ROW_BCAST15, 0xa>(0, (int)v);
			v += (uint32_t)dpp_or_old<DPP_ROW_BCAST31, 0xc>(0, (int)v);
			c2[j] = v;
		}
		if (lane == 63) {
#pragma unroll
			for (int j = 0; j < 8; ++j) ((uint32_t*)tot)[wave * 8 + j] = c2[j];
		}
		__syncthreads();
		// lane d < 16 of every wave: words of digit d in front of this wave's, words of smaller digits in front of those
		uint32_t bv = 0;
		{
			const int d = lane & 15;
			uint32_t all = 0, before = 0;
#pragma unroll
			for (int x = 0; x < 16; ++x) { const uint32_t v = tot[x * 16 + d]; all += v; before += x < wave ? v : 0u; }
			uint32_t inc = all;                                          // words of digits <= d: prefix over the sixteen lanes of the row
			inc += (uint32_t)dpp_or_old<DPP_ROW_SHR(1), 0xf>(0, (int)inc);
			inc += (uint32_t)dpp_or_old<DPP_ROW_SHR(2), 0xf>(0, (int)inc);
			inc += (uint32_t)dpp_or_old<DPP_ROW_SHR(4), 0xf>(0, (int)inc);
			inc += (uint32_t)dpp_or_old<DPP_ROW_SHR(8), 0xf>(0, (int)inc);
			bv = inc - all + before;
		}
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const uint32_t bj = (uint32_t)__builtin_amdgcn_readlane((int)bv, 2 * j) | (uint32_t)__builtin_amdgcn_readlane((int)bv, 2 * j + 1) << 16;
			const uint32_t v = c2[j] - own[j] + bj;                      // exclusive over the workgroup, both fields at once
			tbl[(2 * j) * SEED_TPB_C + t] = (uint16_t)v;
			tbl[(2 * j + 1) * SEED_TPB_C + t] = (uint16_t)(v >> 16);
		}
		// (a thread reads its own sixteen entries back: LDS operations of one wave are performed in order)
#pragma unroll
		for (int e = 0; e < I; ++e) if (e < mine) key[tbl[dg[e] * SEED_TPB_C + t] + lr[e]] = k[e];
		__syncthreads();
		if (sh + SEED_RDX_BITS < lo_bit + bits) {
#pragma unroll
			for (int e = 0; e < I; ++e) if (e < mine) k[e] = key[t * I + e];
		}
	}
}

// takes the reads (and work items) of up to max_n2 anchors: sixteen waves with bucket tables up to max_n anchors, four
// above (the LDS layout is chosen per read; the launch reserves the larger of the two)
#define SEED_TPB 1024
// waves that have bucket tables (and so take large ranges): all sixteen for reads of the first configuration, four for the second
__host__ __device__ inline int seed_table_waves(int workers) { return workers >= 32 ? SEED_TPB / 64 : 4; }
__global__ __launch_bounds__(SEED_TPB) void k_seed_sort(int64_t n_reads, int max_n, int max_n2, int try_network, const int64_t *__restrict__ off,
                                                  const ulonglong2 *__restrict__ src, ulonglong2 *__restrict__ a,
                                                  const SeedItem *__restrict__ items, const unsigned long long *__restrict__ n_items,
                                                  int phase, uint32_t *__restrict__ tied, unsigned long long *__restrict__ n_tied)
{
	// phase 0: every unit start to end.  phase 1: the network only -- a unit with equal x is put on the list `tied` instead of going
	// through the reference's procedure here; phase 2: that procedure for the units on the list.  A read with equal x takes ~1 ms
	// (serial digit walks), a tie-free one ~60 us: in one launch the few tied reads are the kernel's tail with the rest of the chip
	// idle; as a launch of their own they leave the chip to whatever the process' other streams have to run.
	extern __shared__ uint64_t seed_lds[];
	const int lane = threadIdx.x;
	const int64_t n_units = phase == 2 ? (int64_t)*n_tied : n_reads + (int64_t)*n_items;   // whole reads, then the buckets k_seed_sort_huge left
	if (phase == 2) try_network = 0;
	for (int64_t r0 = blockIdx.x; r0 < n_units; r0 += gridDim.x) {
		const int64_t r = phase == 2 ? (int64_t)tied[r0] : r0;
		int64_t b, n64;
		int shift0 = 56;
		if (r < n_reads) { b = off[r]; n64 = off[r + 1] - b; }
		else {
			const SeedItem it = items[r - n_reads];
			b = (int64_t)((uint64_t)(it.beg_hi_shift >> 8) << 32 | it.beg_lo); n64 = it.len; shift0 = (int)(it.beg_hi_shift & 0xff);
		}
		if (n64 > (max_n2 > max_n ? max_n2 : max_n)) continue;
		const int n = (int)n64;
		const int cap_n = n <= max_n ? max_n : max_n2;
		const int q_big = seed_big_slots(cap_n), q_slots = q_big + seed_small_slots(n <= max_n ? 32 : 4);
		const int wave = lane >> 6, wl = lane & 63;
		const int n_tab = seed_table_waves(n <= max_n ? 32 : 4), tw = wave < n_tab ? wave : 0;
		uint64_t *key = seed_lds;
		uint16_t *idx = (uint16_t*)(key + cap_n);
		uint16_t *head = idx + cap_n + tw * 768, *tail = head + 256, *start = tail + 256;      // one set of bucket tables per wave that takes ranges
		unsigned int *cnt = (unsigned int*)(idx + cap_n + n_tab * 768) + tw * 256;              // and one set of digit counts
		SeedRange *qbase = (SeedRange*)(cnt - tw * 256 + n_tab * 256);
		int *qn = (int*)(qbase + 2 * q_slots);                             // [parity][0 = big ranges, 1 = small ranges]
		uint8_t *lab = (uint8_t*)(qn + 4);                                 // cap_n: the current digit of every position
		ulonglong2 *ag = a + b;                                            // this read's output range doubles as scratch until the final gather
		__syncthreads();
		// A sorted order is unique when all keys differ, and then any sort will do: sort first (the whole workgroup busy) and keep
		// the result unless two neighbours are equal; only reads with equal x go through the reference's procedure below (whose
		// top levels are serial walks over thousands of digits).
		int pow2 = 64;
		while (pow2 < n) pow2 <<= 1;
		// First attempt: key and place packed into one 64-bit word -- strand, reference id and position in as many bits as the read's
		// largest id and position need, 14 bits of place below them; the order of the words is the order of x, the place only separates
		// equal x, which sends the read to the reference's procedure anyway -- and the words sorted by a radix sort over the key bits
		// (seed_radix_words: ~24 key bits for an 8 kb read against a few hundred references, i.e. six passes of ~3 us) or, where the
		// LDS behind the keys has no room for its tables (other LDS sizes than this device's), by a bitonic network on neighbouring
		// PAIRS of words (ds_read_b128 / ds_write_b128: both partners of two adjacent comparators are adjacent in LDS, ascending, or
		// descending on a merge's mirror step).  The version further below -- 64-bit key plus 16-bit place, one comparator at a time,
		// 91 steps for 4 800 anchors -- issued eight LDS instructions per comparator and took 97-137 us per read; it remains for
		// reads whose ids and positions do not fit 50 bits.
		int packed_state = try_network ? 0 : 2;                           // 0: sorted and tie-free, 1: equal x found, 2: not attempted / keys too wide
#ifdef SEED_STAMPS
		unsigned long long stp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SEED_T(k) stp[k] = __builtin_amdgcn_s_memtime()
#else
#define SEED_T(k)
#endif
		SEED_T(0);
		if (try_network) {
			typedef unsigned long long sd_u64x2 __attribute__((ext_vector_type(2)));
			constexpr int PLACE_BITS = 14;
			if (lane == 0) { qn[0] = 0; qn[1] = 0; qn[2] = 0; }
			__syncthreads();
			uint32_t or_lo = 0, or_hi = 0;
			for (int i = lane; i < n; i += SEED_TPB) {
				const uint64_t x = src[b + i].x;
				key[i] = x;
				or_lo |= (uint32_t)x; or_hi |= (uint32_t)(x >> 32) & 0x7fffffffu;
			}
			for (int d = 32; d; d >>= 1) { or_lo |= (uint32_t)__shfl_xor((int)or_lo, d, 64); or_hi |= (uint32_t)__shfl_xor((int)or_hi, d, 64); }
			SEED_T(1);
			if (wl == 0) { atomicOr((unsigned int*)&qn[1], or_lo); atomicOr((unsigned int*)&qn[2], or_hi); }
			__syncthreads();
			const int pbits = qn[1] ? 32 - __builtin_clz((unsigned)qn[1]) : 0, rbits = qn[2] ? 32 - __builtin_clz((unsigned)qn[2]) : 0;
			const int kbits = 1 + rbits + pbits;                         // strand | id | position
			if (kbits + PLACE_BITS > 64) packed_state = 2;
			__syncthreads();
			if (packed_state == 0) {
				for (int i = lane; i < n; i += SEED_TPB) {
					const uint64_t x = key[i];
					key[i] = (((x >> 63) << (rbits + pbits)) | (((x >> 32) & 0x7fffffffull) << pbits) | (x & 0xffffffffull)) << PLACE_BITS | (uint64_t)i;
				}
				__syncthreads();
			}
			SEED_T(2);
			const int items = (n + SEED_TPB - 1) / SEED_TPB;
			// (the radix sort's tables lie over idx[], the bucket tables and the digit counts, all unused until the procedure below)
			const bool radix = packed_state == 0 && items <= 13 && (size_t)cap_n * 2 + (size_t)n_tab * (1536 + 1024) >= 2 * 16 * SEED_TPB + 512;
			if (radix) {
				uint16_t *tbl = (uint16_t*)(key + cap_n), *tot = tbl + 16 * SEED_TPB;
#if defined(SEED_EXP_PASSES)
				const int rb = SEED_EXP_PASSES * SEED_RDX_BITS;                 // (timing experiment: wrong order)
#else
				const int rb = (kbits + SEED_RDX_BITS - 1) / SEED_RDX_BITS * SEED_RDX_BITS;
#endif
				switch (items) {
				case 1: seed_radix_words<1>(key, n, PLACE_BITS, rb, tbl, tot); break;
				case 2: seed_radix_words<2>(key, n, PLACE_BITS, rb, tbl, tot); break;
				case 3: seed_radix_words<3>(key, n, PLACE_BITS, rb, tbl, tot); break;
				case 4: seed_radix_words<4>(key, n, PLACE_BITS, rb, tbl, tot); break;
				case 5: seed_radix_words<5>(key, n, PLACE_BITS, rb, tbl, tot); break;
				case 6: seed_radix_words<6>(key, n, PLACE_BITS, rb, tbl, tot); break;
				case 7: seed_radix_words<7>(key, n, PLACE_BITS, rb, tbl, tot); break;
				case 8: seed_radix_words<8>(key, n, PLACE_BITS, rb, tbl, tot); break;
				case 9: case 10: seed_radix_words<10>(key, n, PLACE_BITS, rb, tbl, tot); break;
				default: seed_radix_words<13>(key, n, PLACE_BITS, rb, tbl, tot); break;
				}
			}
			if (packed_state == 0) {
				auto cex = [](unsigned long long &lo, unsigned long long &hi) { if (lo > hi) { const unsigned long long t = lo; lo = hi; hi = t; } };
				for (int k = 2; k <= (radix ? 0 : pow2); k <<= 1) {
					for (int j = k >> 1; j > 0; j >>= 1) {
						if (j == 1) {                                             // partners are neighbours: a pair per load
							for (int t0 = lane; 2 * t0 + 1 < n; t0 += 2 * SEED_TPB) {
								const int t1 = t0 + SEED_TPB;
								const bool h1 = 2 * t1 + 1 < n;
								sd_u64x2 v0 = *(const sd_u64x2*)(key + 2 * t0), v1 = h1 ? *(const sd_u64x2*)(key + 2 * t1) : v0;
								if (v0.x > v0.y) { const sd_u64x2 s2 = {v0.y, v0.x}; *(sd_u64x2*)(key + 2 * t0) = s2; }
								if (h1 && v1.x > v1.y) { const sd_u64x2 s2 = {v1.y, v1.x}; *(sd_u64x2*)(key + 2 * t1) = s2; }
							}
						} else {
							const bool mirror = j == k >> 1;
							for (int d0 = lane; d0 < pow2 / 4; d0 += 2 * SEED_TPB) {  // two double comparators in flight per thread
								int pi[2], plo[2], kind[2];                           // kind 0: nothing, 1: both comparators, 2: only the one whose partner is plo
								unsigned long long ax[2], ay[2], bx[2], by[2];
#pragma unroll
								for (int u = 0; u < 2; ++u) {
									const int t = 2 * (d0 + u * SEED_TPB);
									pi[u] = ((t & ~(j - 1)) << 1) | (t & (j - 1));
									plo[u] = mirror ? ((pi[u] | (k - 1)) - (pi[u] & (j - 1))) - 1 : pi[u] | j;   // the lower of the two partners' places
									kind[u] = (d0 + u * SEED_TPB >= pow2 / 4 || plo[u] >= n) ? 0 : plo[u] + 1 < n ? 1 : 2;
									ax[u] = ay[u] = bx[u] = 0; by[u] = ~0ull;
									if (kind[u] == 1) {
										const sd_u64x2 A = *(const sd_u64x2*)(key + pi[u]), B = *(const sd_u64x2*)(key + plo[u]);
										ax[u] = A.x; ay[u] = A.y; bx[u] = B.x; by[u] = B.y;
									} else if (kind[u] == 2) { ax[u] = key[pi[u]]; ay[u] = key[pi[u] + 1]; bx[u] = key[plo[u]]; }   // (place plo + 1 is padding: +inf)
								}
#pragma unroll
								for (int u = 0; u < 2; ++u) {
									if (kind[u] == 0) continue;
									const unsigned long long ax0 = ax[u], ay0 = ay[u];
									if (mirror) { cex(ax[u], by[u]); cex(ay[u], bx[u]); }         // pi ~ plo + 1, pi + 1 ~ plo
									else { cex(ax[u], bx[u]); cex(ay[u], by[u]); }
									if (ax[u] == ax0 && ay[u] == ay0) continue;                   // nothing moved
									if (kind[u] == 1) {
										const sd_u64x2 A = {ax[u], ay[u]}, B = {bx[u], by[u]};
										*(sd_u64x2*)(key + pi[u]) = A; *(sd_u64x2*)(key + plo[u]) = B;
									} else { key[pi[u]] = ax[u]; key[pi[u] + 1] = ay[u]; key[plo[u]] = bx[u]; }
								}
							}
						}
						__syncthreads();
					}
				}
				SEED_T(3);
				int ties = 0;
				for (int i = lane; i + 1 < n; i += SEED_TPB) ties |= (key[i] >> PLACE_BITS) == (key[i + 1] >> PLACE_BITS);
				if (ties) qn[0] = 1;
				__syncthreads();
				packed_state = qn[0];
				__syncthreads();
				if (packed_state == 0) {
					SEED_T(4);
					for (int i = lane; i < n; i += SEED_TPB) a[b + i] = src[b + (int)(key[i] & ((1u << PLACE_BITS) - 1u))];
#ifdef SEED_STAMPS
					__syncthreads();
					SEED_T(5);
					if (blockIdx.x == 7 && lane == 0) printf("[seed stamp] n=%d load %llu, bits+repack %llu, sort %llu, ties %llu, gather %llu (ticks of 10 ns)\n", n, stp[1] - stp[0], stp[2] - stp[1], stp[3] - stp[2], stp[4] - stp[3], stp[5] - stp[4]);
#endif
					continue;
				}
				if (phase == 1) { if (lane == 0) tied[atomicAdd(n_tied, 1ull)] = (uint32_t)r; continue; }
			}
		}
		if (packed_state == 2 && try_network) {
			// bitonic network in its all-ascending form (first step of a merge compares mirror positions), so that the
			// virtual +inf padding behind the n real keys never has to move and needs no storage
			for (int i = lane; i < n; i += SEED_TPB) { key[i] = src[b + i].x; idx[i] = (uint16_t)i; }
			if (lane == 0) qn[0] = 0;
			__syncthreads();
			for (int k = 2; k <= pow2; k <<= 1) {
				for (int j = k >> 1; j > 0; j >>= 1) {
					const bool mirror = j == k >> 1;
					for (int t0 = lane; t0 < pow2 / 2; t0 += 4 * SEED_TPB) {              // four disjoint pairs in flight per thread
						int pi[4], pl[4];
						uint64_t ki[4], kl[4];
#pragma unroll
						for (int u = 0; u < 4; ++u) {
							const int t = t0 + u * SEED_TPB;
							pi[u] = ((t & ~(j - 1)) << 1) | (t & (j - 1));
							pl[u] = mirror ? (pi[u] | (k - 1)) - (pi[u] & (j - 1)) : pi[u] | j;   // mirror: block end minus offset
							if (t >= pow2 / 2 || pl[u] >= n) pl[u] = -1;
							else { ki[u] = key[pi[u]]; kl[u] = key[pl[u]]; }
						}
#pragma unroll
						for (int u = 0; u < 4; ++u) {
							if (pl[u] >= 0 && ki[u] > kl[u]) {
								const uint16_t ti = idx[pi[u]]; idx[pi[u]] = idx[pl[u]]; idx[pl[u]] = ti;
								key[pi[u]] = kl[u]; key[pl[u]] = ki[u];
							}
						}
					}
					__syncthreads();
				}
			}
			int ties = 0;
			for (int i = lane; i + 1 < n; i += SEED_TPB) ties |= key[i] == key[i + 1];
			if (ties) qn[0] = 1;
			__syncthreads();
			const bool has_ties = qn[0] != 0;
			__syncthreads();
			if (!has_ties) {
				for (int i = lane; i < n; i += SEED_TPB) a[b + i] = src[b + idx[i]];
				continue;
			}
			if (phase == 1) { if (lane == 0) tied[atomicAdd(n_tied, 1ull)] = (uint32_t)r; continue; }
		}
		for (int i = lane; i < n; i += SEED_TPB) { key[i] = src[b + i].x; idx[i] = (uint16_t)i; }
		if (lane == 0) { qbase[0] = SeedRange{0, (uint16_t)n, (uint16_t)shift0, 0}; qn[0] = 1; qn[1] = 0; qn[2] = 0; qn[3] = 0; }
		__syncthreads();
		for (int which = 0;; which ^= 1) {
			SeedRange *cur = qbase + which * q_slots, *nxt = qbase + (which ^ 1) * q_slots;
			int *ncnt = qn + 2 * (which ^ 1);
			const int n_big = qn[2 * which], n_small = min(qn[2 * which + 1], q_slots - q_big);
			if (n_big + n_small == 0) break;
			for (int w = wave; w < n_small; w += SEED_TPB / 64) {                    // ksort.h:148, a wave per small range: the insertion sort of
				const SeedRange rg = cur[q_big + w];                                // ksort.h:107-117 is stable, i.e. a rank by (key, place), a lane per anchor
				const int len = rg.end - rg.beg;
				const uint64_t k = wl < len ? key[rg.beg + wl] : ~0ull;
				const uint16_t ix = wl < len ? idx[rg.beg + wl] : (uint16_t)0;
				int rank = 0;
				for (int j = 0; j < len; ++j) {
					const uint64_t kj = readlane_u64(k, j);
					rank += (kj < k) | (kj == k & j < wl);
				}
				if (wl < len) { key[rg.beg + rank] = k; idx[rg.beg + rank] = ix; }
			}
			for (int w = wave; w < n_big && wave < n_tab; w += n_tab) {              // a wave per large range
				const SeedRange rg = cur[w];
				const int rb = rg.beg, re = rg.end, len = re - rb, sh = rg.shift, next = sh > 8 ? sh - 8 : 0;
				if (len <= 64) { if (wl == 0) seed_isort(key, idx, rb, re); continue; }       // ksort.h:143 (a whole read of <= 64 anchors)
				for (int d = wl; d < 256; d += 64) cnt[d] = 0;
				wave_global_fence();
				for (int q = rb + wl; q < re; q += 64) {                             // ksort.h:126, and the digits written down
					const int dg = (int)(key[q] >> sh & 0xff);
					lab[q] = (uint8_t)dg;
					atomicAdd(&cnt[dg], 1u);
				}
				wave_global_fence();
				if ((int)cnt[(int)(key[rb] >> sh & 0xff)] == len) {                   // one bucket: the pass moves nothing
					if (wl == 0 && sh) nxt[atomicAdd(&ncnt[0], 1)] = SeedRange{(uint16_t)rb, (uint16_t)re, (uint16_t)next, 0};
					wave_global_fence();
					continue;
				}
				int n_digits, d_lo, d_hi;                                            // digits in use; the lowest and the highest of them
				{                                                                    // ksort.h:127-128: lane wl owns digits 4 wl .. 4 wl + 3
					const int c0 = (int)cnt[4 * wl], c1 = (int)cnt[4 * wl + 1], c2 = (int)cnt[4 * wl + 2], c3 = (int)cnt[4 * wl + 3];
					int incl = c0 + c1 + c2 + c3;
					for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (wl >= d) incl += t; }
					int acc = rb + incl - (c0 + c1 + c2 + c3);
					head[4 * wl] = start[4 * wl] = (uint16_t)acc; acc += c0; tail[4 * wl] = (uint16_t)acc;
					head[4 * wl + 1] = start[4 * wl + 1] = (uint16_t)acc; acc += c1; tail[4 * wl + 1] = (uint16_t)acc;
					head[4 * wl + 2] = start[4 * wl + 2] = (uint16_t)acc; acc += c2; tail[4 * wl + 2] = (uint16_t)acc;
					head[4 * wl + 3] = start[4 * wl + 3] = (uint16_t)acc; acc += c3; tail[4 * wl + 3] = (uint16_t)acc;
					n_digits = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
					{                                                                // the digits in use, ascending, where the counts were
						int at = n_digits;
						for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(at, d); if (wl >= d) at += t; }
						at -= n_digits;
						if (c0) cnt[at++] = 4 * wl;
						if (c1) cnt[at++] = 4 * wl + 1;
						if (c2) cnt[at++] = 4 * wl + 2;
						if (c3) cnt[at++] = 4 * wl + 3;
					}
					d_lo = c0 ? 4 * wl : c1 ? 4 * wl + 1 : c2 ? 4 * wl + 2 : c3 ? 4 * wl + 3 : 256;
					d_hi = c3 ? 4 * wl + 3 : c2 ? 4 * wl + 2 : c1 ? 4 * wl + 1 : c0 ? 4 * wl : -1;
					for (int d = 32; d > 0; d >>= 1) {
						n_digits += __shfl_xor(n_digits, d);
						d_lo = min(d_lo, __shfl_xor(d_lo, d)); d_hi = max(d_hi, __shfl_xor(d_hi, d));
					}
				}
				wave_global_fence();
				// ksort.h:129-141.  The loop reads every element once, at a bucket's head, before anything was stored there: its
				// course depends on the digits at the original places only (see k_seed_sort_huge).  One lane replays it over the
				// digit bytes and notes the source of every position (in the read's output range, free until the final gather);
				// the wave then moves keys and indices.
				if (n_digits == 2) {
					// Two buckets, A = [rb, mid) for d_lo and B = [mid, re) for d_hi (the strand level always; often the next one):
					// the loop has a closed form.  With a_1 < a_2 < .. the places in A that hold a B element and b_1 < b_2 < ..
					// the places in B that hold an A element, cycle j carries a_j to the head of B, which stands just behind
					// b_(j-1); every B element from there on moves up one place until b_j is reached, whose A element closes
					// the cycle at a_j.  So a_j receives b_j; in B a place q <= b_m receives a_j if it starts segment j
					// (q = mid or q - 1 holds an A element) and q - 1 otherwise; everything else stays.  Lists in the .x fields.
					const int mid = tail[d_lo];
					const unsigned long long below = (1ull << wl) - 1;
					int run = 0;
					for (int base = rb; base < mid; base += 64) {
						const int q = base + wl;
						const bool f = q < mid && lab[q] == d_hi;
						const unsigned long long m = __ballot(f);
						if (f) ag[rb + run + __popcll(m & below)].x = (uint64_t)q;
						run += __popcll(m);
					}
					const int n_cycles = run;
					run = 0;
					for (int base = mid; base < re; base += 64) {
						const int q = base + wl;
						const bool f = q < re && lab[q] == d_lo;
						const unsigned long long m = __ballot(f);
						if (f) ag[mid + run + __popcll(m & below)].x = (uint64_t)q;
						run += __popcll(m);
					}
					wave_global_fence();
					run = 0;
					for (int base = rb; base < mid; base += 64) {
						const int q = base + wl;
						const bool f = q < mid && lab[q] == d_hi;
						const unsigned long long m = __ballot(f);
						if (q < mid) ag[q].y = f ? ag[mid + run + __popcll(m & below)].x : (uint64_t)q;
						run += __popcll(m);
					}
					run = 0;
					for (int base = mid; base < re; base += 64) {
						const int q = base + wl;
						const bool f = q < re && lab[q] == d_lo;
						const unsigned long long m = __ballot(f);
						const int before = run + __popcll(m & below);                 // A elements in [mid, q)
						if (q < re) {
							uint64_t from = (uint64_t)q;
							if (before < n_cycles) from = (q == mid || lab[q - 1] == d_lo) ? ag[rb + before].x : (uint64_t)(q - 1);
							ag[q].y = from;
						}
						run += __popcll(m);
					}
				} else if (wl == 0) {
					for (int di = 0; di < n_digits; ++di) {                          // empty buckets have nothing to place
						const int d = (int)cnt[di];
						int hd = head[d];
						const int td = tail[d];
						while (hd != td) {
							int l = lab[hd];
							if (l != d) {
								int carry = hd;
								do {
									const int hp = head[l];
									ag[hp].y = (uint64_t)carry;
									carry = hp;
									head[l] = (uint16_t)(hp + 1);
									l = lab[hp];
								} while (l != d);
								ag[hd].y = (uint64_t)carry;
							} else ag[hd].y = (uint64_t)hd;
							++hd;
						}
					}
				}
				wave_global_fence();
				for (int q = rb + wl; q < re; q += 64) {
					const int p = (int)ag[q].y;
					ulonglong2 t; t.x = key[p]; t.y = idx[p];
					ag[q] = t;
				}
				wave_global_fence();
				for (int q = rb + wl; q < re; q += 64) { const ulonglong2 t = ag[q]; key[q] = t.x; idx[q] = (uint16_t)t.y; }
				wave_global_fence();
				if (sh) {                                                            // ksort.h:143-149: the buckets are the next round's work
					for (int d = 4 * wl; d < 4 * wl + 4; ++d) {
						const int sb = start[d], se = tail[d];
						if (se - sb > 64) nxt[atomicAdd(&ncnt[0], 1)] = SeedRange{(uint16_t)sb, (uint16_t)se, (uint16_t)next, 0};
						else if (se - sb > 1) {
							const int k = atomicAdd(&ncnt[1], 1);
							if (k < q_slots - q_big) nxt[q_big + k] = SeedRange{(uint16_t)sb, (uint16_t)se, (uint16_t)next, 0};
							else seed_isort(key, idx, sb, se);
						}
					}
				}
				wave_global_fence();
			}
			__syncthreads();
			if (lane == 0) { qn[2 * which] = 0; qn[2 * which + 1] = 0; }
			__syncthreads();
		}
		for (int i = lane; i < n; i += SEED_TPB) a[b + i] = src[b + idx[i]];
#ifdef SEED_STAMPS
		if (phase == 2 && lane == 0) printf("[seed stamp] phase 2 unit %lld of %lld: n=%d, %llu ticks\n", (long long)r0, (long long)n_units, n, (unsigned long long)(__builtin_amdgcn_s_memtime() - stp[0]));
#endif
	}
}

size_t seed_sort_lds_bytes(int max_n, int workers, int coop)
{
	(void)coop;
	return (((size_t)max_n * 10 + 7) & ~(size_t)7) + (size_t)seed_table_waves(workers) * (768 * 2 + 256 * 4)
	       + 2 * (size_t)(seed_big_slots(max_n) + seed_small_slots(workers)) * sizeof(SeedRange) + 16 + (((size_t)max_n + 7) & ~(size_t)7);
}

hipError_t launch_seed_collect(hipStream_t st, const SeedIndex &ix, int flag, int max_occ, int64_t n_reads, int64_t n_mini,
                               const int64_t *d_mini_off, const void *d_mini, const uint32_t *d_bid, SeedScratch sc,
                               int64_t *d_off, int64_t *d_mp_off, int32_t *d_rep_len)
{
	hipError_t e;
	if (n_mini > 0) {
		hipLaunchKernelGGL(k_seed_probe, dim3((unsigned)((n_mini + 255) / 256)), dim3(256), 0, st, ix, flag, max_occ, n_reads, n_mini, d_mini_off,
		                   (const ulonglong2*)d_mini, d_bid, sc.kept, sc.used, sc.src, sc.mstate);
	}
	if ((e = launch_scan_u64(st, n_mini, sc.kept, sc.tile_tmp, sc.totals)) != hipSuccess) return e;
	if ((e = launch_scan_u64(st, n_mini, sc.used, sc.tile_tmp, sc.totals + 1)) != hipSuccess) return e;
	hipLaunchKernelGGL(k_seed_reads, dim3((unsigned)((n_reads + 1 + 3) / 4)), dim3(256), 0, st, n_reads, n_mini, d_mini_off, (const ulonglong2*)d_mini,
	                   sc.kept, sc.used, sc.mstate, sc.totals, d_off, d_mp_off, d_rep_len);
	return hipGetLastError();
}

hipError_t launch_seed_expand_sort(hipStream_t st, const SeedIndex &ix, int flag, int64_t n_reads, int64_t n_mini,
                                   const int64_t *d_mini_off, const void *d_mini, const uint32_t *d_bid, const int32_t *d_qlen, SeedScratch sc,
                                   void *d_unsorted, void *d_a, const int64_t *d_off, unsigned long long *d_mini_pos, int max_n, int max_n2,
                                   int lab_cap, int64_t total)
{
	if (n_mini > 0) {
		hipLaunchKernelGGL(k_seed_expand, dim3((unsigned)((n_mini + 255) / 256)), dim3(256), 0, st, ix, flag, n_reads, n_mini, d_mini_off,
		                   (const ulonglong2*)d_mini, d_bid, d_qlen, sc.kept, sc.used, sc.src, sc.mstate, (ulonglong2*)d_unsorted, d_mini_pos);
	}
	if (n_reads > 0) {
		// the LDS sort takes reads of up to max_n / max_n2 anchors (what fits the device's LDS per workgroup); larger reads
		// have their top levels done by k_seed_sort_huge, which hands the buckets that fit back to the LDS sort as work items
		const int cap = max_n2 > max_n ? max_n2 : max_n;
		const int64_t huge_max64 = (int64_t)SEED_HUGE_STACK * cap;
		const int huge_max = cap <= 0 ? 0 : (int)(huge_max64 < 0x7fffffff ? huge_max64 : 0x7fffffff);
		SeedItem *items = (SeedItem*)sc.stacks;
		unsigned long long *n_items = sc.totals + 2;
		hipError_t e = hipMemsetAsync(n_items, 0, 16, st);                  // (and the count of tied units behind it)
		if (e != hipSuccess) return e;
		if (cap > 0 && total > cap) {
			const unsigned hgrid = (unsigned)(n_reads < 256 ? n_reads : 256);
			hipLaunchKernelGGL(k_seed_sort_huge, dim3(hgrid), dim3(256), (size_t)lab_cap + 4096 + 12 * SEED_HUGE_STACK + 16, st, n_reads, cap, huge_max, lab_cap,
			                   d_off, (ulonglong2*)d_unsorted, (ulonglong2*)d_a, items, n_items);
		}
		const int64_t units = n_reads + (total > cap && cap > 0 ? total / 65 : 0);
		static const int grid_cap = getenv("CHAINDP_SEED_GRID") ? atoi(getenv("CHAINDP_SEED_GRID")) : 256 * 8;   // (read once; tuning only)
		const unsigned grid = (unsigned)(units < grid_cap ? units : grid_cap);
		static const int try_network = getenv("CHAINDP_SEED_FORCE_EXACT") == nullptr;      // measurement switch: every read through the reference's procedure
		if (cap > 0) {
			size_t lds = seed_sort_lds_bytes(max_n, 32, 8);
			if (max_n2 > max_n && seed_sort_lds_bytes(max_n2, 4, 2) > lds) lds = seed_sort_lds_bytes(max_n2, 4, 2);
			if (try_network && sc.tied) {
				hipLaunchKernelGGL(k_seed_sort, dim3(grid), dim3(SEED_TPB), lds, st, n_reads, max_n, max_n2, 1, d_off, (const ulonglong2*)d_unsorted, (ulonglong2*)d_a, items, n_items,
				                   1, sc.tied, sc.totals + 3);
				hipLaunchKernelGGL(k_seed_sort, dim3(grid < 256 ? grid : 256), dim3(SEED_TPB), lds, st, n_reads, max_n, max_n2, 0, d_off, (const ulonglong2*)d_unsorted, (ulonglong2*)d_a, items, n_items,
				                   2, sc.tied, sc.totals + 3);
			} else
				hipLaunchKernelGGL(k_seed_sort, dim3(grid), dim3(SEED_TPB), lds, st, n_reads, max_n, max_n2, try_network, d_off, (const ulonglong2*)d_unsorted, (ulonglong2*)d_a, items, n_items,
				                   0, (uint32_t*)nullptr, sc.totals + 3);
		}
		// reads beyond SEED_HUGE_STACK x cap anchors: one thread each in global memory
		hipLaunchKernelGGL(k_seed_sort_big, dim3((unsigned)((n_reads + 63) / 64)), dim3(64), 0, st, n_reads, huge_max, d_off, (const ulonglong2*)d_unsorted,
		                   (ulonglong2*)d_a, (BtRange*)sc.stacks);
	}
	return hipGetLastError();
}

} // namespace chaindp
