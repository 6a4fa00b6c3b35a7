// chaindp_prepass.hip -- everything that runs before the chain DP kernel: the per-block read table, the
// anchor-parallel prepass (q_span sums, unit starts, singletons), the unit list in longest-first order, and the
// per-read gap-cost table.  See chaindp_kernels.hip for the overall scheme.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "chaindp_kernels.h"
#include "chaindp_wave.h"
#include "chaindp_lanes.h"

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
__global__ __launch_bounds__(256) void k_block_reads(int64_t n_reads, int64_t total, const int64_t *__restrict__ off, int2 *__restrict__ block_reads,
                                                    unsigned long long *__restrict__ sumq, unsigned long long *__restrict__ counters,
                                                    unsigned int *__restrict__ hist, unsigned long long *__restrict__ left_cnt)
{
	// the batch's accumulators start at zero: done here, by the first kernel of the step, instead of four memsets of a few bytes
	// to a few hundred kilobytes (~25 us each on the stream: 0.1 ms of a 4 ms step)
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_reads; i += (int64_t)gridDim.x * blockDim.x) sumq[i] = 0;
	if (blockIdx.x == 0) {
		if (threadIdx.x < 2) counters[threadIdx.x] = 0;
		if (threadIdx.x < 4 && left_cnt) left_cnt[threadIdx.x] = 0;
		for (int i = threadIdx.x; i < 2 * 128; i += blockDim.x) hist[i] = 0;      // (2 x UNIT_CLASSES: class counts, cursors)
	}
	const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t g0 = b * PRE_PER_BLOCK;
	if (g0 >= total) return;
	const int64_t g1 = g0 + PRE_PER_BLOCK < total ? g0 + PRE_PER_BLOCK : total;
	const int64_t rlo = read_of(off, 0, n_reads - 1, g0);
	block_reads[b] = make_int2((int)rlo, (int)read_of(off, rlo, n_reads - 1, g1 - 1));
}

// One wave per 1024-anchor block, sixteen tiles of 64 consecutive anchors in order: the wave streams 16 KB from one place, with
// PW_DEPTH tiles requested ahead of the one in hand (a workgroup of 256 threads taking one anchor per thread and pass had a single
// short burst in flight per wave, then its dependent loads of the read's bounds, a block-wide reduction and the end of the
// workgroup: 3.3 TB/s with neither the vector nor the scalar pipe busy).  What a tile needs from its neighbours comes out of
// registers -- the last x of the tile before, the first x of the tile after, already loaded -- and the read's bounds, the counts
// and the q_span sum are wave-uniform state carried from tile to tile, read boundaries inside a tile included (lane ranges, no
// per-lane search): no LDS, no barrier, one 128-byte store of the block's sixteen start masks, one atomic per block and read.
//
// The tile loads of a full block are issued and awaited by hand: loads return in order, so waiting until at most (tile loads issued
// after the one needed) operations are outstanding is safe whatever else -- the atomics of a read's q_span sum -- sits between them,
// whereas the compiler, which counts every conditional memory operation into its bookkeeping, ends up at s_waitcnt vmcnt(0) before
// every tile, i.e. waiting for the load it has just issued.  (Between four and ten tiles in flight the kernel's time does not move.)
#define PW_TILES (PRE_PER_BLOCK / 64)
#ifndef PW_DEPTH
#define PW_DEPTH 6                                                   // tiles requested ahead of the one in hand
#endif

typedef uint32_t pw_u32x4 __attribute__((ext_vector_type(4)));       // an anchor: x.lo, x.hi, y.lo, y.hi

// largest r in [lo, hi] with off[r] <= g for wave-uniform arguments; off[] through the scalar cache
__device__ __forceinline__ int64_t read_of_uniform(const int64_t *__restrict__ off, int64_t lo, int64_t hi, int64_t g)
{
	while (lo < hi) {
		const int64_t mid = (lo + hi + 1) >> 1;
		if (*TW_CONST(int64_t, off + mid) <= g) lo = mid; else hi = mid - 1;
	}
	return lo;
}

template <int N>
__device__ __forceinline__ void pw_wait(pw_u32x4 &q0, pw_u32x4 &q1)     // both tiles have arrived once at most N later operations are outstanding
{
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("s_waitcnt vmcnt(%2)" : "+v"(q0), "+v"(q1) : "n"(N));
#endif
}

template <bool FULL>                                                 // FULL: all 1024 anchors exist (every block but the batch's last)
__device__ __forceinline__ void prepass_block(const Params &par, const int64_t b, const int lane, const int64_t total,
                                              const int64_t *__restrict__ off, const ulonglong2 *__restrict__ a,
                                              unsigned long long *__restrict__ sumq, uint64_t *__restrict__ start_mask,
                                              uint64_t *__restrict__ single_mask, uint64_t *__restrict__ emit_mask,
                                              unsigned long long *__restrict__ block_cnt, const int2 *__restrict__ block_reads)
{
	const uint64_t maxx = (uint64_t)(int64_t)par.max_dist_x;
	const int64_t g0 = b * PRE_PER_BLOCK;
	const int64_t g1 = FULL ? g0 + PRE_PER_BLOCK : total;
	const int n_in = FULL ? PRE_PER_BLOCK : (int)(g1 - g0);          // anchors of the block
	const int nt = (n_in + 63) >> 6;                                 // tiles of the block
	const uint32_t voff = (uint32_t)lane * 16u;
	pw_u32x4 q[PW_TILES];
	auto request = [&](int t) {                                      // tile t of the block -> q[t]
		if (FULL) {
#if defined(__HIP_DEVICE_COMPILE__)
			const ulonglong2 *const base = a + g0 + (t & ~3) * 64;       // (the instruction's offset field reaches four tiles)
			if ((t & 3) == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(q[t]) : "v"(voff), "s"(base) : "memory");
			else if ((t & 3) == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(q[t]) : "v"(voff), "s"(base) : "memory");
			else if ((t & 3) == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(q[t]) : "v"(voff), "s"(base) : "memory");
			else asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(q[t]) : "v"(voff), "s"(base) : "memory");
#endif
		} else {                                                     // the batch's last block: plain loads, the compiler's own waits
			q[t] = (pw_u32x4)(0u);                                   // (an anchor that does not exist: x = 0, q_span = 0, no segment id)
			if (t * 64 + lane < n_in) q[t] = *(const pw_u32x4*)(a + g0 + t * 64 + lane);
		}
	};
#pragma unroll
	for (int k = 0; k < PW_DEPTH; ++k) request(k);
	const int2 rr = *TW_CONST(int2, block_reads + b);                // reads of the block's first and last anchor (k_block_reads)
	uint64_t x_carry = g0 > 0 ? *TW_CONST(uint64_t, &a[g0 - 1].x) : 0;   // x of the anchor before the tile in hand (wave-uniform)
	const uint64_t x_after = g1 < total ? *TW_CONST(uint64_t, &a[g1].x) : 0;   // x of the anchor after the block
	// the read the tile in hand lies in, its bounds relative to the block (32-bit: the scalar unit compares no 64-bit order):
	// rs_b < 0 when it starts before the block, re_b capped when it ends far behind it
	int64_t r_cur = rr.x;
	int rs_b, re_b;
	auto bounds = [&]() {
		const int64_t ds = *TW_CONST(int64_t, off + r_cur) - g0, de = *TW_CONST(int64_t, off + r_cur + 1) - g0;
		rs_b = ds < 0 ? -1 : (int)ds;
		re_b = ((uint64_t)de >> 20) ? 1 << 20 : (int)de;
	};
	bounds();
	unsigned int w_sum = 0, any_seg = 0;                             // per lane, of r_cur
	uint64_t span0_m = 0;                                            // lanes that saw a zero q_span in r_cur
	unsigned int n_units = 0, n_singles = 0;                         // wave-uniform
	uint32_t mk_lo = 0, mk_hi = 0, sk_lo = 0, sk_hi = 0, ek_lo = 0, ek_hi = 0;   // lane t: tile t's unit starts, singletons, emitted singletons
	auto flush = [&]() {                                             // r_cur's share of the block is complete
		unsigned int s = w_sum;
		for (int d = 32; d; d >>= 1) s += __shfl_xor(s, d, 64);
		if (lane == 0 && s) atomicAdd(&sumq[r_cur], (unsigned long long)s);
		if (__builtin_amdgcn_ballot_w64(any_seg != 0) && lane == 0) atomicOr(&sumq[r_cur], SUMQ_SEG_FLAG);     // rare: multi-segment reads only
		if (span0_m && lane == 0) atomicOr(&sumq[r_cur], SUMQ_SPAN0_FLAG);                                      // (never, in minimap2's own anchors)
		w_sum = 0; any_seg = 0; span0_m = 0;
	};
	auto tile = [&](auto tile_index) {                               // (a constant per call: the waits and lane selects below are immediates)
		constexpr int t = decltype(tile_index)::value;
		if (!FULL && t >= nt) return;
		const int tb = t * 64;                                       // the tile's place in the block
		const int in_tile = FULL ? 64 : (n_in - tb < 64 ? n_in - tb : 64);
		const uint64_t have_m = FULL || in_tile == 64 ? ~0ull : (1ull << in_tile) - 1;
		constexpr int LAST = PW_TILES - 1;
		constexpr int newest = t + PW_DEPTH - 1 < LAST ? t + PW_DEPTH - 1 : LAST, need = t < LAST ? t + 1 : LAST;
		if (FULL) pw_wait<newest - need>(q[t], q[need]);
		const pw_u32x4 an = q[t];
#ifdef PW_EXP_LOADONLY
		any_seg ^= an.x ^ an.w;
		if (t + PW_DEPTH < PW_TILES && (FULL || t + PW_DEPTH < nt)) request(t + PW_DEPTH);
		return;
#endif
		uint64_t x_follow = x_after;
		if (t + 1 < PW_TILES && (FULL || t + 1 < nt)) x_follow = readlane_u64((uint64_t)q[need].y << 32 | q[need].x, 0);
		const uint32_t plo = (uint32_t)dpp_or_old<DPP_WAVE_SHR1, 0xf>((int)(uint32_t)x_carry, (int)an.x);          // lane 0 keeps the carry
		const uint32_t phi = (uint32_t)dpp_or_old<DPP_WAVE_SHR1, 0xf>((int)(uint32_t)(x_carry >> 32), (int)an.y);
		const uint32_t nlo = (uint32_t)dpp_or_old<DPP_WAVE_SHL1, 0xf>((int)(uint32_t)x_follow, (int)an.x);         // lane 63 keeps what follows
		const uint32_t nhi = (uint32_t)dpp_or_old<DPP_WAVE_SHL1, 0xf>((int)(uint32_t)(x_follow >> 32), (int)an.y);
		const uint64_t x = (uint64_t)an.y << 32 | an.x, xprev = (uint64_t)phi << 32 | plo, xnext = (uint64_t)nhi << 32 | nlo;
		const int span = (int)(an.w & 0xffu);
		const uint32_t segbits = an.w & 0x00ff0000u;
		// lane masks, combined by the scalar unit
		const uint64_t far_prev = __builtin_amdgcn_uicmpl(x - xprev, maxx, 34), far_next = __builtin_amdgcn_uicmpl(xnext - x, maxx, 34);
		const uint64_t zero_m = __builtin_amdgcn_uicmp((unsigned)span, 0u, 32) & have_m;
		// the reads of the tile, one lane range after the other (usually one: the whole tile inside r_cur)
		uint64_t start_x = 0, last_x = 0;                            // starts / ends forced by the reads' bounds
		int seg_lo = 0;
		for (;;) {
			if (re_b <= tb + seg_lo) {                               // r_cur ends here: the range starts in a later read
				flush();
				r_cur = read_of_uniform(off, r_cur, rr.y, g0 + tb + seg_lo);
				bounds();
			}
			if (rs_b == tb + seg_lo) start_x |= 1ull << seg_lo;
			const int l_re = re_b - tb;                              // r_cur ends in front of this lane of the tile (> seg_lo)
			const int seg_hi = l_re < in_tile ? l_re : in_tile;
			if (l_re <= 64) last_x |= 1ull << (l_re - 1);
			if (seg_lo == 0 && seg_hi == 64) {
				w_sum += (unsigned int)span; any_seg |= segbits; span0_m |= zero_m;
			} else {                                                 // lanes [seg_lo, seg_hi) are r_cur's
				const uint64_t m = ((seg_hi == 64 ? 0ull : 1ull << seg_hi) - 1ull) & (~0ull << seg_lo);
				const bool mine = __builtin_amdgcn_inverse_ballot_w64(m);
				w_sum += mine ? (unsigned int)span : 0u; any_seg |= mine ? segbits : 0u; span0_m |= zero_m & m;
			}
			if (seg_hi >= in_tile) break;
			seg_lo = seg_hi;
		}
		const uint64_t start_m = (far_prev | start_x) & have_m;
		const uint64_t single_m = start_m & (far_next | last_x);
		// a singleton's results (chain.c:251,283-284 with an empty window: f = v = q_span, p = -1; emitted at its own step iff
		// v >= min_sc, chain.c:304) are not stored here: two masks say which anchors are singletons and which of those are emitted,
		// the compaction reads them beside flags[], and k_fill_singles writes f, p, v, flags for callers that look at those arrays
		// (104 M scattered partial stores on the 100k-read job cost the kernel a third of its time: 2.0 -> 2.9 ms)
		const uint64_t emit_m = single_m & __builtin_amdgcn_sicmp(span, par.min_sc, 39);
		const uint64_t em = start_m & ~single_m;
		n_units += (unsigned int)__builtin_popcountll(em);
		n_singles += (unsigned int)__builtin_popcountll(single_m);
#if defined(__HIP_DEVICE_COMPILE__)
		asm("v_writelane_b32 %0, %1, %2" : "+v"(mk_lo) : "s"((uint32_t)em), "n"(t));
		asm("v_writelane_b32 %0, %1, %2" : "+v"(mk_hi) : "s"((uint32_t)(em >> 32)), "n"(t));
		asm("v_writelane_b32 %0, %1, %2" : "+v"(sk_lo) : "s"((uint32_t)single_m), "n"(t));
		asm("v_writelane_b32 %0, %1, %2" : "+v"(sk_hi) : "s"((uint32_t)(single_m >> 32)), "n"(t));
		asm("v_writelane_b32 %0, %1, %2" : "+v"(ek_lo) : "s"((uint32_t)emit_m), "n"(t));
		asm("v_writelane_b32 %0, %1, %2" : "+v"(ek_hi) : "s"((uint32_t)(emit_m >> 32)), "n"(t));
#endif
		x_carry = readlane_u64(x, 63);                               // (a tile that is not full is the last one)
		if (t + PW_DEPTH < PW_TILES && (FULL || t + PW_DEPTH < nt)) request(t + PW_DEPTH);
		__builtin_amdgcn_sched_barrier(0);                           // tiles one after the other: interleaved, their masks run the scalar registers out
	};
	tile(std::integral_constant<int, 0>{}); tile(std::integral_constant<int, 1>{}); tile(std::integral_constant<int, 2>{}); tile(std::integral_constant<int, 3>{});
	tile(std::integral_constant<int, 4>{}); tile(std::integral_constant<int, 5>{}); tile(std::integral_constant<int, 6>{}); tile(std::integral_constant<int, 7>{});
	tile(std::integral_constant<int, 8>{}); tile(std::integral_constant<int, 9>{}); tile(std::integral_constant<int, 10>{}); tile(std::integral_constant<int, 11>{});
	tile(std::integral_constant<int, 12>{}); tile(std::integral_constant<int, 13>{}); tile(std::integral_constant<int, 14>{}); tile(std::integral_constant<int, 15>{});
	flush();
	if (lane < nt) {
		start_mask[(g0 >> 6) + lane] = (uint64_t)mk_hi << 32 | mk_lo;
		single_mask[(g0 >> 6) + lane] = (uint64_t)sk_hi << 32 | sk_lo;
		emit_mask[(g0 >> 6) + lane] = (uint64_t)ek_hi << 32 | ek_lo;
	}
	if (lane == 0) block_cnt[b] = (unsigned long long)n_singles << 32 | n_units;   // two counters, one scan
}

__global__ __launch_bounds__(PRE_BLOCK) void k_prepass(Params par, int64_t n_reads, int64_t total,
                                                       const int64_t *__restrict__ off, const ulonglong2 *__restrict__ a,
                                                       unsigned long long *__restrict__ sumq, uint64_t *__restrict__ start_mask,
                                                       uint64_t *__restrict__ single_mask, uint64_t *__restrict__ emit_mask,
                                                       unsigned long long *__restrict__ block_cnt, const int2 *__restrict__ block_reads)
{
	const int lane = threadIdx.x & 63;
	const int64_t b = (int64_t)blockIdx.x * (PRE_BLOCK / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int64_t g0 = b * PRE_PER_BLOCK;
	if (g0 >= total) return;
	if (g0 + PRE_PER_BLOCK <= total) prepass_block<true>(par, b, lane, total, off, a, sumq, start_mask, single_mask, emit_mask, block_cnt, block_reads);
	else prepass_block<false>(par, b, lane, total, off, a, sumq, start_mask, single_mask, emit_mask, block_cnt, block_reads);
}

// f, p, v and flags[] of the singletons, from the prepass' masks: for callers that read those arrays (chaindp_download, a run on the
// caller's own device arrays).  The compaction does not need it.
__global__ __launch_bounds__(256) void k_fill_singles(int64_t total, int min_sc, const ulonglong2 *__restrict__ a,
                                                      const uint64_t *__restrict__ single_mask,
                                                      int32_t *__restrict__ f, int32_t *__restrict__ p, int32_t *__restrict__ v,
                                                      uint8_t *__restrict__ flags)
{
	for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
		if (!((single_mask[g >> 6] >> (g & 63)) & 1u)) continue;
		const int span = span_of_hi((uint32_t)(a[g].y >> 32));
		f[g] = span; p[g] = -1; v[g] = span;
		flags[g] = (uint8_t)(span >= min_sc ? 2 | 8 : 0);
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
	// a workgroup takes many 256-word pieces and adds its class counts to the batch's once at the end: with a workgroup per piece
	// the adds to the few classes that hold most units queued up at their addresses (37 000 workgroups on the 100k-read job,
	// ~7.5 ns per same-address atomic: 0.3 of the kernel's 0.44 ms)
	// A wave takes 64 consecutive mask words (four blocks: a 16-lane row each).  What a word needs from its neighbours comes out of
	// the wave: the units in front of it inside its block from a row prefix of popcounts (it was up to fifteen loads), the start
	// behind its last unit from the next non-empty lane (it was a walk over the following words, a load each); only a word whose
	// next start lies beyond the wave's 64 still walks.
	const int lane = threadIdx.x & 63;
	for (int64_t wb = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~(int64_t)63; wb < n_words; wb += (int64_t)gridDim.x * blockDim.x) {
		const int64_t w = wb + lane;
		uint64_t m = w < n_words ? start_mask[w] : 0;
		const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
		uint32_t incl = cnt;                                         // units up to this word inside its block
		incl += (uint32_t)dpp_or_old<DPP_ROW_SHR(1), 0xf>(0, (int)incl);
		incl += (uint32_t)dpp_or_old<DPP_ROW_SHR(2), 0xf>(0, (int)incl);
		incl += (uint32_t)dpp_or_old<DPP_ROW_SHR(4), 0xf>(0, (int)incl);
		incl += (uint32_t)dpp_or_old<DPP_ROW_SHR(8), 0xf>(0, (int)incl);
		const uint64_t nz = __builtin_amdgcn_ballot_w64(m != 0);
		const uint64_t above = lane < 63 ? nz & (~0ull << (lane + 1)) : 0ull;
		const int nl = above ? __builtin_ctzll(above) : lane;        // the next lane that has a unit start
		const int first_there = __shfl(m ? __builtin_ctzll(m) : 0, nl, 64);
		const int64_t next_in_wave = above ? ((wb + nl) << 6) + first_there : -1;
		if (m) {
			const int64_t b = w / PRE_WORDS;
			uint64_t pos = (uint32_t)block_base[b] + (incl - cnt);   // low word of block_base: units before this block
			// the read of the word's first anchor: from the block's read range (k_block_reads) instead of a search over all reads
			// (fourteen dependent loads for 12 500 reads; a block usually lies in one read or straddles two)
			const int2 rr = block_reads[b];
			int64_t r = read_of(off, rr.x, rr.y, w << 6);
			int64_t re = off[r + 1];                                 // (kept across the word's units: a load per unit was a trip to L2 per unit, one after the other)
			while (m) {
				const int bit = __builtin_ctzll(m);
				m &= m - 1;
				const int64_t g = (w << 6) + bit;
				while (g >= re) re = off[++r + 1];                   // units of one word are in anchor order; reads only move forward
				// upper bound of the unit: the next unit's start or the end of the read (singletons in between are
				// not units, so this can overshoot the true end; the DP kernel finds the true end itself)
				int64_t next = -1;
				if (m) next = (w << 6) + __builtin_ctzll(m);
				else if (next_in_wave >= 0) next = next_in_wave;
				else for (int64_t k = wb + 64; k < n_words && (k << 6) < re; ++k) {
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
                                                      unsigned int *__restrict__ key_range, const int32_t *__restrict__ n_segs_pr)
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
				const bool segs = n_segs_pr ? n_segs_pr[u[k].read] > 1 : par.n_segs > 1;   // (collect_task_t::n_segs, fpga_chaindp.h:53: the read's own count)
				ax.flags = ((sq & (SUMQ_SEG_FLAG | SUMQ_SPAN0_FLAG)) || lut16 || segs) ? 1u : 0u; ax.pad = 0;
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
                          UnitAux *d_unit_aux, const int32_t *d_n_segs, unsigned long long *d_left_cnt)
{
	hipError_t e = hipSuccess;
	if (n_reads <= 0 || total <= 0) {                                // an empty batch: no units, no singletons, nothing handed over
		if ((e = hipMemsetAsync(d_counters, 0, 2 * sizeof(unsigned long long), st)) != hipSuccess) return e;
		return d_left_cnt ? hipMemsetAsync(d_left_cnt, 0, 4 * sizeof(unsigned long long), st) : hipSuccess;
	}
	const int64_t blocks = (total + PRE_PER_BLOCK - 1) / PRE_PER_BLOCK;
	const int64_t words = (total + 63) / 64;
	hipLaunchKernelGGL(k_block_reads, dim3((unsigned)((blocks + 255) / 256)), dim3(256), 0, st, n_reads, total, d_off, sc.block_reads,
	                   d_sumq, d_counters, sc.hist, d_left_cnt);
	hipLaunchKernelGGL(k_prepass, dim3((unsigned)((blocks + PRE_BLOCK / 64 - 1) / (PRE_BLOCK / 64))), dim3(PRE_BLOCK), 0, st, par, n_reads, total, d_off, (const ulonglong2*)d_a,
	                   d_sumq, sc.start_mask, sc.single_mask, sc.emit_mask, sc.block_cnt, sc.block_reads);
	if ((e = launch_scan_u64(st, blocks, sc.block_cnt, sc.tile_tmp, d_counters)) != hipSuccess) return e;
	hipLaunchKernelGGL(k_emit_units, dim3((unsigned)((words + 255) / 256 < 2048 ? (words + 255) / 256 : 2048)), dim3(256), 0, st, n_reads, words, d_off,
	                   sc.start_mask, sc.block_cnt, sc.units_tmp, sc.hist, sc.block_reads);
	hipLaunchKernelGGL(k_unit_bases, dim3(1), dim3(64), 0, st, sc.hist, sc.hist + UNIT_CLASSES, d_unit_aux ? sc.key_range : nullptr);
	hipLaunchKernelGGL(k_unit_scatter, dim3((unsigned)(blocks < 1024 ? (blocks > 0 ? blocks : 1) : 1024)), dim3(256), 0, st, d_counters, sc.units_tmp,
	                   sc.hist, sc.hist + UNIT_CLASSES, d_units, par, d_off, d_sumq, d_unit_aux, d_unit_aux ? sc.key_range : nullptr, d_n_segs);
	return hipGetLastError();
}

hipError_t launch_fill_singles(hipStream_t st, const Params &par, int64_t total, const void *d_a, PrepassScratch sc,
                               int32_t *d_f, int32_t *d_p, int32_t *d_v, uint8_t *d_flags)
{
	if (total <= 0) return hipSuccess;
	const int64_t blocks = (total + 255) / 256;
	hipLaunchKernelGGL(k_fill_singles, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, total, par.min_sc, (const ulonglong2*)d_a,
	                   sc.single_mask, d_f, d_p, d_v, d_flags);
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
