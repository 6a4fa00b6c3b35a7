// chaindp_kernels.hip -- hand-written CDNA4 (gfx950) kernels for minimap2's anchor-chaining DP.
//
// What is computed (reference chain.c:246-284, per read, anchors a[0..n) sorted by x):
//   for each i: scan predecessors j = i-1 .. st (window: a[i].x - a[j].x <= max_dist_x), score
//   each pair, keep the running max (strictly greater wins, so ties keep the larger j), stop
//   early once more than max_skip already-visited ("marked") predecessors failed to improve.
// How it is computed here (DESIGN.md "Wave formulation"; modelled on the CPU in oracle/wave_model.c):
//   * k_prepass  -- one thread per anchor: per-read sum of q_span (for avg_qspan, chain.c:240-241);
//                   split each read into UNITS where a[i].x-a[i-1].x > max_dist_x (independent DP
//                   problems); resolve single-anchor units on the spot; zero the global mark array.
//   * k_chain_units -- one wave64 per unit.  Anchors enter in coalesced 64-anchor tiles (16 B per
//                   lane) and live in an LDS ring (x,y,f,p,mark,v for the last RING anchors).  For
//                   anchor i, lane k evaluates predecessor j = i-1-64c-k of chunk c; the serial
//                   semantics of the scalar loop are recovered with wave primitives: DPP prefix-max
//                   for "is this a new running max", ballots + mbcnt (or a DPP prefix-min over the
//                   clamped walk) for n_skip and the break position, LDS scatter for the marks.
//                   Predecessors older than the ring are read back from HBM/L2 (the "deep" path).
//   Integer arithmetic only, except the reference's own (int)(dd * .01 * avg_qspan) in f64.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "chaindp_kernels.h"

namespace chaindp {

// ---------------------------------------------------------------- wave primitives (wave64, DPP)

// dpp_ctrl encodings (gfx9 family): row_shr:n = 0x110+n, wave_shr:1 = 0x138, row_bcast:15 = 0x142, row_bcast:31 = 0x143
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_WAVE_SHR1 0x138
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_old(int old, int src)
{
	return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}

// inclusive prefix max over the 64 lanes (lane 0 first)
__device__ __forceinline__ int wave_scan_max(int v)
{
	v = max(v, dpp_or_old<DPP_ROW_SHR(1), 0xf>(INT_MIN, v));
	v = max(v, dpp_or_old<DPP_ROW_SHR(2), 0xf>(INT_MIN, v));
	v = max(v, dpp_or_old<DPP_ROW_SHR(4), 0xf>(INT_MIN, v));
	v = max(v, dpp_or_old<DPP_ROW_SHR(8), 0xf>(INT_MIN, v));
	v = max(v, dpp_or_old<DPP_ROW_BCAST15, 0xa>(INT_MIN, v));
	v = max(v, dpp_or_old<DPP_ROW_BCAST31, 0xc>(INT_MIN, v));
	return v;
}

// inclusive prefix min
__device__ __forceinline__ int wave_scan_min(int v)
{
	v = min(v, dpp_or_old<DPP_ROW_SHR(1), 0xf>(INT_MAX, v));
	v = min(v, dpp_or_old<DPP_ROW_SHR(2), 0xf>(INT_MAX, v));
	v = min(v, dpp_or_old<DPP_ROW_SHR(4), 0xf>(INT_MAX, v));
	v = min(v, dpp_or_old<DPP_ROW_SHR(8), 0xf>(INT_MAX, v));
	v = min(v, dpp_or_old<DPP_ROW_BCAST15, 0xa>(INT_MAX, v));
	v = min(v, dpp_or_old<DPP_ROW_BCAST31, 0xc>(INT_MAX, v));
	return v;
}

// value of lane-1 (lane 0 receives `first`)
__device__ __forceinline__ int wave_shift_up1(int v, int first)
{
	return dpp_or_old<DPP_WAVE_SHR1, 0xf>(first, v);
}

// number of set bits of the wave-uniform mask m strictly below this lane
__device__ __forceinline__ int lanes_below(uint64_t m)
{
	return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l)
{
	uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
	uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
	return (uint64_t)hi << 32 | lo;
}

// Orders this wave's LDS/global accesses as seen by its own lanes.  Lanes of one wave hand data to
// each other through memory (lane 0 stores f[i], every lane reads it one step later); the hardware
// executes a wave's DS (and, per address space, VMEM) operations in issue order, so all that is needed
// is that the compiler keeps program order: a wavefront-scope fence emits no instruction.
__device__ __forceinline__ void wave_mem_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Same for the deep path, where the hand-off goes through global memory: the stores must have
// reached the CU's L1/L2 before the loads that follow are issued.  All traffic is from ONE wave on
// ONE CU, whose vector L1 is coherent for its own work-group, so work-group scope (s_waitcnt vmcnt(0))
// is sufficient; no agent-scope cache maintenance is involved.
__device__ __forceinline__ void wave_global_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---------------------------------------------------------------- field access (mmpriv.h:21-22, chain.c:250)

__device__ __forceinline__ int span_of_hi(uint32_t yhi) { return (int)(yhi & 0xffu); }        // (y>>32)&0xff
__device__ __forceinline__ int seg_of_hi(uint32_t yhi) { return (int)((yhi >> 16) & 0xffu); } // (y>>48)&0xff

// ---------------------------------------------------------------- K0: prepass (anchor-parallel, no hot atomics)
// k_prepass:    one thread per anchor, PRE_PER_BLOCK consecutive anchors of the batch per block.  The
//               block finds the reads its range touches by binary search in off[]; each thread derives
//               its unit-start flag and whether it is a singleton (resolved on the spot), zeroes its
//               global mark, and the block writes: a 64-bit unit-start mask per wave-tile, its unit and
//               singleton counts, and ONE integer atomic per (block, read) for the q_span sum
//               (order-independent, so deterministic).
// k_scan_blocks: exclusive scan of the per-block unit counts (single block) -> counters[0..1].
// k_emit_units: one thread per mask word; writes the Unit records in anchor order (deterministic).
// A single same-address atomic per wave would cap this stage at ~90 atomics/us (measured: 9 ms for
// 76 M anchors), hence count -> scan -> emit.

#define PRE_BLOCK 256
#define PRE_PER_BLOCK 1024
#define PRE_WORDS (PRE_PER_BLOCK / 64)

// largest r in [lo, hi] with off[r] <= g   (off is non-decreasing; empty reads are skipped over)
__device__ __forceinline__ int64_t read_of(const int64_t *__restrict__ off, int64_t lo, int64_t hi, int64_t g)
{
	while (lo < hi) {
		const int64_t mid = (lo + hi + 1) >> 1;
		if (off[mid] <= g) lo = mid; else hi = mid - 1;
	}
	return lo;
}

__global__ __launch_bounds__(PRE_BLOCK) void k_prepass(Params par, int64_t n_reads, int64_t total,
                                                       const int64_t *__restrict__ off, const ulonglong2 *__restrict__ a,
                                                       unsigned long long *__restrict__ sumq, uint64_t *__restrict__ start_mask,
                                                       uint32_t *__restrict__ block_units, uint32_t *__restrict__ block_singles,
                                                       int32_t *__restrict__ f, int32_t *__restrict__ p, int32_t *__restrict__ v,
                                                       int32_t *__restrict__ tg)
{
	__shared__ int64_t s_rlo, s_rhi;
	__shared__ unsigned int s_sum, s_units, s_singles;
	const int lane = threadIdx.x & 63;
	const uint64_t maxx = (uint64_t)(int64_t)par.max_dist_x;
	const int64_t g0 = (int64_t)blockIdx.x * PRE_PER_BLOCK;
	const int64_t g1 = g0 + PRE_PER_BLOCK < total ? g0 + PRE_PER_BLOCK : total;
	if (threadIdx.x == 0) {
		s_rlo = read_of(off, 0, n_reads - 1, g0);
		s_rhi = read_of(off, 0, n_reads - 1, g1 - 1);
		s_sum = 0; s_units = 0; s_singles = 0;
	}
	__syncthreads();
	const int64_t rlo = s_rlo, rhi = s_rhi;
	const bool one_read = rlo == rhi;
	unsigned int w_sum = 0, w_units = 0, w_singles = 0;
	for (int64_t gb = g0; gb < g1; gb += PRE_BLOCK) {
		const int64_t g = gb + threadIdx.x;
		const bool have = g < g1;
		bool start = false, single = false;
		int span = 0;
		int64_t r = rlo;
		if (have) {
			if (!one_read) r = read_of(off, rlo, rhi, g);
			const int64_t rs = off[r], re = off[r + 1];
			const ulonglong2 an = a[g];
			span = span_of_hi((uint32_t)(an.y >> 32));
			start = g == rs || an.x - a[g - 1].x > maxx;
			const bool next_starts = g + 1 >= re || a[g + 1].x - an.x > maxx;
			single = start && next_starts;
			tg[g] = 0;
			if (single) { f[g] = span; p[g] = -1; v[g] = span; }   // chain.c:251,283-284 with an empty window
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
		block_units[blockIdx.x] = s_units;
		block_singles[blockIdx.x] = s_singles;
	}
}

// in place: block_units[b] <- exclusive prefix sum; counters[0] <- total units, counters[1] <- total singletons
__global__ __launch_bounds__(1024) void k_scan_blocks(int64_t n_blocks, uint32_t *__restrict__ block_units,
                                                      const uint32_t *__restrict__ block_singles,
                                                      unsigned long long *__restrict__ counters)
{
	__shared__ unsigned long long part[1024], part_s[1024];
	const int tid = threadIdx.x;
	const int64_t per = (n_blocks + 1023) / 1024;
	const int64_t lo = (int64_t)tid * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
	unsigned long long s = 0, ss = 0;
	for (int64_t b = lo; b < hi; ++b) { s += block_units[b]; ss += block_singles[b]; }
	part[tid] = s; part_s[tid] = ss;
	__syncthreads();
	if (tid == 0) {
		unsigned long long acc = 0, acc_s = 0;
		for (int k = 0; k < 1024; ++k) { const unsigned long long t = part[k]; part[k] = acc; acc += t; acc_s += part_s[k]; }
		counters[0] = acc; counters[1] = acc_s;
	}
	__syncthreads();
	unsigned long long acc = part[tid];
	for (int64_t b = lo; b < hi; ++b) { const uint32_t t = block_units[b]; block_units[b] = (uint32_t)acc; acc += t; }
}

__global__ __launch_bounds__(256) void k_emit_units(int64_t n_reads, int64_t n_words, const int64_t *__restrict__ off,
                                                    const uint64_t *__restrict__ start_mask,
                                                    const uint32_t *__restrict__ block_base, Unit *__restrict__ units)
{
	const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= n_words) return;
	uint64_t m = start_mask[w];
	if (!m) return;
	const int64_t b = w / PRE_WORDS;
	uint64_t pos = block_base[b];
	for (int64_t k = b * PRE_WORDS; k < w; ++k) pos += (uint64_t)__builtin_popcountll(start_mask[k]);
	int64_t r = read_of(off, 0, n_reads - 1, w << 6);
	while (m) {
		const int bit = __builtin_ctzll(m);
		m &= m - 1;
		const int64_t g = (w << 6) + bit;
		while (g >= off[r + 1]) ++r;            // units of one word are in anchor order; reads only move forward
		Unit u;
		u.start = g; u.read = (int32_t)r; u.pad = 0;
		units[pos++] = u;
	}
}

// ---------------------------------------------------------------- K1: chain DP, one wave per unit

// chain.c:264-272 for one pair; sc0 = min(dq, dr, q_span)
__device__ __forceinline__ int pair_score(int sc0, int dd, int dr, int dq, bool same, int is_cdna, double avgd)
{
	const int lg = dd ? 31 - __builtin_clz((unsigned)dd) : 0;
	const int lin = (int)((double)dd * .01 * avgd);
	if (is_cdna || !same) {
		if (!same && dr == 0) return sc0 + 1;
		if (dr > dq || !same) return sc0 - (lin < lg ? lin : lg);
		return sc0 - (lin + (lg >> 1));
	}
	return sc0 - (lin + (lg >> 1));
}

template <int RING>
__global__ __launch_bounds__(64) void k_chain_units(Params par, const int64_t *__restrict__ off,
                                                    const ulonglong2 *__restrict__ a, const int32_t *__restrict__ n_segs_pr,
                                                    const unsigned long long *__restrict__ sumq, const Unit *__restrict__ units,
                                                    const unsigned long long *__restrict__ counters,
                                                    int32_t *f, int32_t *p, int32_t *v, int32_t *tg)
{
	constexpr int MASK = RING - 1;
	constexpr int DEPTH = RING - 64;        // tiles overwrite 64 slots at once, so only RING-64 predecessors are guaranteed resident
	static_assert((RING & MASK) == 0 && RING >= 128, "RING must be a power of two >= 128");
	__shared__ uint4 s_a[RING];             // x.lo, x.hi, qpos, y.hi
	__shared__ int2 s_fp[RING];             // f, p (unit-relative)
	__shared__ int s_t[RING];               // mark tag (chain.c's t[])
	__shared__ int s_v[RING];

	const int lane = threadIdx.x;
	for (int64_t ub = blockIdx.x; ub < (int64_t)counters[0]; ub += gridDim.x) {
		const Unit u = units[ub];
		const int64_t rs = off[u.read], re = off[u.read + 1];
		const int64_t base = u.start;
		const int64_t room = re - base;
		const int rel0 = (int)(base - rs);
		const double avgd = (double)((float)(uint64_t)sumq[u.read] / (float)(int64_t)(re - rs));   // chain.c:241: f32 divide of converted u64 and i64
		const int n_segs = n_segs_pr ? n_segs_pr[u.read] : par.n_segs;
		const uint64_t maxx = (uint64_t)(int64_t)par.max_dist_x;
		const int mdx = par.max_dist_x, mdy = par.max_dist_y, bw = par.bw, max_skip = par.max_skip, is_cdna = par.is_cdna;
		const bool seg_rule = n_segs > 1 && !is_cdna;        // chain.c:261

		wave_mem_fence();
		for (int k = lane; k < RING; k += 64) s_t[k] = 0;
		uint64_t x_carry = 0;

		for (int tile0 = 0;; tile0 += 64) {
			const int64_t gi = base + tile0 + lane;
			const bool have = tile0 + lane < room;
			ulonglong2 an = make_ulonglong2(0, 0);
			if (have) an = a[gi];
			// the unit ends at the first gap > max_dist_x (or at the end of the read)
			uint64_t xp;
			{
				const uint32_t lo = (uint32_t)wave_shift_up1((int)(uint32_t)an.x, (int)(uint32_t)x_carry);
				const uint32_t hi = (uint32_t)wave_shift_up1((int)(uint32_t)(an.x >> 32), (int)(uint32_t)(x_carry >> 32));
				xp = (uint64_t)hi << 32 | lo;
			}
			const bool stop = !have || ((tile0 + lane) > 0 && an.x - xp > maxx);
			const uint64_t stop_m = __builtin_amdgcn_ballot_w64(stop);
			const int cnt = stop_m ? __builtin_ctzll(stop_m) : 64;
			if (cnt == 0) break;
			x_carry = readlane_u64(an.x, 63);
			const uint32_t my_yhi = (uint32_t)(an.y >> 32);
			wave_mem_fence();
			if (lane < cnt)
				s_a[(tile0 + lane) & MASK] = make_uint4((uint32_t)an.x, (uint32_t)(an.x >> 32), (uint32_t)an.y, my_yhi);
			wave_mem_fence();

			int fo = 0, po = -1, vo = 0;
			for (int ii = 0; ii < cnt; ++ii) {
				const int i = tile0 + ii;                      // unit-relative index of the anchor being scored
				const uint64_t ri = readlane_u64(an.x, ii);
				const int qi = __builtin_amdgcn_readlane((int)(uint32_t)an.y, ii);
				const uint32_t yhi_i = (uint32_t)__builtin_amdgcn_readlane((int)my_yhi, ii);
				const int span = span_of_hi(yhi_i), sidi = seg_of_hi(yhi_i);
				const int tag = i + 1;
				int max_f = span, max_j = -1, n_skip = 0;

				for (int kb0 = 0; kb0 < i; kb0 += 64) {
					const bool deep = kb0 + 64 > DEPTH;        // wave-uniform
					const int k = kb0 + lane;
					const bool inr = k < i;
					const int j = i - 1 - k;
					const int slot = j & MASK;
					uint64_t xj = 0;
					int qj = 0, fj = 0, pj = -1;
					uint32_t yhj = 0;
					if (!deep) {
						const uint4 aj = s_a[slot];
						const int2 fp = s_fp[slot];
						xj = (uint64_t)aj.y << 32 | aj.x; qj = (int)aj.z; yhj = aj.w; fj = fp.x; pj = fp.y;
					} else {
						wave_global_fence();
						if (inr) {
							const ulonglong2 aj = a[base + j];
							xj = aj.x; qj = (int)(uint32_t)aj.y; yhj = (uint32_t)(aj.y >> 32);
							fj = f[base + j];
							pj = p[base + j];
							pj = pj < 0 ? -1 : pj - rel0;          // stored read-relative
						}
					}
					const uint64_t d64 = ri - xj;
					const bool live = inr && d64 <= maxx;          // chain.c:252 (window) per lane
					const int dr = (int)d64;
					const int dq = (int)((uint32_t)qi - (uint32_t)qj);
					const bool same = seg_of_hi(yhj) == sidi;
					const int dd = dr > dq ? dr - dq : dq - dr;
					bool ok = live;
					ok = ok && !((same && dr == 0) || dq <= 0);                    // chain.c:257
					ok = ok && !((same && dq > mdy) || dq > mdx);                  // chain.c:258
					ok = ok && !(same && dd > bw);                                 // chain.c:260
					ok = ok && !(seg_rule && same && dr > mdy);                    // chain.c:261
					int sc0 = dq < dr ? dq : dr;
					sc0 = sc0 > span ? span : sc0;                                 // chain.c:262-263
					int sc = pair_score(sc0, dd, dr, dq, same, is_cdna, avgd) + fj; // chain.c:264-273
					sc = ok ? sc : INT_MIN;

					// marks of every filter-passing lane first (chain.c:281), then each lane reads its own
					if (ok && pj >= 0) {
						if (i - pj <= DEPTH) s_t[pj & MASK] = tag;
						else tg[base + pj] = tag;
					}
					int tj = 0;
					if (!deep) {
						wave_mem_fence();
						tj = s_t[slot];
					} else {
						wave_global_fence();
						if (inr) tj = tg[base + j];
					}

					// new running max? strictly greater than everything before it (chain.c:274)
					const int incl = wave_scan_max(sc);
					int excl = wave_shift_up1(incl, max_f);
					excl = excl > max_f ? excl : max_f;
					const bool isA = ok && sc > excl;
					const bool isB = ok && !isA && tj == tag;                      // chain.c:277
					const uint64_t A = __builtin_amdgcn_ballot_w64(isA);
					const uint64_t B = __builtin_amdgcn_ballot_w64(isB);
					const bool all_live = __builtin_amdgcn_ballot_w64(live) == ~0ull;

					// n_skip walk (chain.c:276,278): A lanes x -> max(x-1,0), B lanes x -> x+1, break when > max_skip
					int kbrk = -1;
					if (B == 0) {
						n_skip -= __builtin_popcountll(A);
						n_skip = n_skip < 0 ? 0 : n_skip;
					} else if ((A & ~((B & (0 - B)) - 1)) == 0) {                  // every A lane precedes every B lane
						int x = n_skip - __builtin_popcountll(A);
						x = x < 0 ? 0 : x;
						int need = max_skip - x + 1;
						need = need < 1 ? 1 : need;
						const int cb = __builtin_popcountll(B);
						if (cb >= need) {
							const uint64_t m = __builtin_amdgcn_ballot_w64(isB && lanes_below(B) == need - 1);
							kbrk = __builtin_ctzll(m);
						} else n_skip = x + cb;
					} else {                                                        // general: clamped walk via prefix min
						const int S = n_skip + lanes_below(B) + (int)isB - lanes_below(A) - (int)isA;
						const int M = wave_scan_min(S);
						const int x = S - (M < 0 ? M : 0);
						const uint64_t m = __builtin_amdgcn_ballot_w64(isB && x > max_skip);
						if (m) kbrk = __builtin_ctzll(m);
						else n_skip = __builtin_amdgcn_readlane(x, 63);
					}
					// the last A lane before the break holds the final running max and its j
					const uint64_t Ap = kbrk >= 0 ? (A & ((1ull << kbrk) - 1)) : A;
					if (Ap) {
						const int ka = 63 - __builtin_clzll(Ap);
						max_f = __builtin_amdgcn_readlane(sc, ka);
						max_j = i - 1 - kb0 - ka;
					}
					if (kbrk >= 0 || !all_live) break;
				}

				// chain.c:283-284
				int vi = max_f;
				if (max_j >= 0) {
					int vj;
					if (i - max_j <= DEPTH) { wave_mem_fence(); vj = s_v[max_j & MASK]; }
					else { wave_global_fence(); vj = v[base + max_j]; }
					vi = vj > max_f ? vj : max_f;
				}
				if (lane == ii) { fo = max_f; po = max_j < 0 ? -1 : max_j + rel0; vo = vi; }
				wave_mem_fence();
				if (lane == 0) { s_fp[i & MASK] = make_int2(max_f, max_j); s_v[i & MASK] = vi; }
				wave_mem_fence();
			}
			if (lane < cnt) { f[gi] = fo; p[gi] = po; v[gi] = vo; }
			if (cnt < 64) break;
		}
	}
}

// ---------------------------------------------------------------- launchers

hipError_t launch_prepass(hipStream_t st, const Params &par, int64_t n_reads, int64_t total, const int64_t *d_off, const void *d_a,
                          unsigned long long *d_sumq, Unit *d_units, unsigned long long *d_counters, PrepassScratch sc,
                          int32_t *d_f, int32_t *d_p, int32_t *d_v, int32_t *d_tg)
{
	hipError_t e = hipMemsetAsync(d_counters, 0, 2 * sizeof(unsigned long long), st);
	if (e != hipSuccess || n_reads <= 0 || total <= 0) return e;
	if ((e = hipMemsetAsync(d_sumq, 0, (size_t)n_reads * sizeof(unsigned long long), st)) != hipSuccess) return e;
	const int64_t blocks = (total + PRE_PER_BLOCK - 1) / PRE_PER_BLOCK;
	const int64_t words = (total + 63) / 64;
	hipLaunchKernelGGL(k_prepass, dim3((unsigned)blocks), dim3(PRE_BLOCK), 0, st, par, n_reads, total, d_off, (const ulonglong2*)d_a,
	                   d_sumq, sc.start_mask, sc.block_units, sc.block_singles, d_f, d_p, d_v, d_tg);
	hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, blocks, sc.block_units, sc.block_singles, d_counters);
	hipLaunchKernelGGL(k_emit_units, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, n_reads, words, d_off,
	                   sc.start_mask, sc.block_units, d_units);
	return hipGetLastError();
}

size_t prepass_scratch_bytes(int64_t max_anchors, size_t *mask_bytes, size_t *blocks_bytes)
{
	const size_t words = (size_t)(max_anchors + 63) / 64, blocks = (size_t)(max_anchors + PRE_PER_BLOCK - 1) / PRE_PER_BLOCK;
	*mask_bytes = (words + 1) * 8;
	*blocks_bytes = (blocks + 1) * 4;
	return *mask_bytes + 2 * *blocks_bytes;
}

hipError_t launch_chain(hipStream_t st, int ring, const Params &par, int64_t max_units, const int64_t *d_off, const void *d_a,
                        const int32_t *d_n_segs, const unsigned long long *d_sumq, const Unit *d_units,
                        const unsigned long long *d_counters, int32_t *d_f, int32_t *d_p, int32_t *d_v, int32_t *d_tg)
{
	if (max_units <= 0) return hipSuccess;
	// The number of units is only known on the device (counters[0]); the grid is sized for the upper
	// bound and blocks beyond the count exit at once, so no host round trip sits between the kernels.
	int64_t blocks = max_units;
	const int64_t cap = 256LL * 32 * 16;
	if (blocks > cap) blocks = cap;
	const ulonglong2 *aa = (const ulonglong2*)d_a;
	switch (ring) {
	case 128: hipLaunchKernelGGL(k_chain_units<128>, dim3((unsigned)blocks), dim3(64), 0, st, par, d_off, aa, d_n_segs, d_sumq, d_units, d_counters, d_f, d_p, d_v, d_tg); break;
	case 512: hipLaunchKernelGGL(k_chain_units<512>, dim3((unsigned)blocks), dim3(64), 0, st, par, d_off, aa, d_n_segs, d_sumq, d_units, d_counters, d_f, d_p, d_v, d_tg); break;
	default:  hipLaunchKernelGGL(k_chain_units<256>, dim3((unsigned)blocks), dim3(64), 0, st, par, d_off, aa, d_n_segs, d_sumq, d_units, d_counters, d_f, d_p, d_v, d_tg); break;
	}
	return hipGetLastError();
}

} // namespace chaindp
