// chaindp_kernels.hip -- hand-written CDNA4 (gfx950) kernels for minimap2's anchor-chaining DP.
//
// What is computed (reference chain.c:246-284, per read, anchors a[0..n) sorted by x):
//   for each i: scan predecessors j = i-1 .. st (window: a[i].x - a[j].x <= max_dist_x), score
//   each pair, keep the running max (strictly greater wins, so ties keep the larger j), stop
//   early once more than max_skip already-visited ("marked") predecessors failed to improve.
// How it is computed here (DESIGN.md "Wave formulation"; modelled on the CPU in oracle/wave_model.c):
//   * k_prepass  -- one thread per anchor: per-read sum of q_span (for avg_qspan, chain.c:240-241);
//                   split each read into UNITS where a[i].x-a[i-1].x > max_dist_x (independent DP
//                   problems); resolve single-anchor units on the spot.
//                   k_emit_units / k_unit_scatter list the units, longest first; k_build_lut tabulates
//                   the read's gap cost (the reference's f32/f64 arithmetic, once per read and dd).
//   * k_chain_units -- one wave64 per unit.  Anchors enter in coalesced 64-anchor tiles (16 B per
//                   lane); the last RING scored anchors live in an LDS ring.  For anchor i, lane k
//                   evaluates predecessor j = i-1-64c-k of chunk c; the serial semantics of the scalar
//                   loop are recovered with wave primitives: a DPP prefix max for "is this a new running
//                   max", lane masks kept as 64-bit scalars (ballot / inverse_ballot), popcounts (or a DPP
//                   prefix min over the clamped walk) for n_skip and the break, an LDS scatter for the
//                   marks.  A table-driven 32-bit variant (run_unit_fast, written for VALU instruction
//                   count) serves ordinary reads, a 64-bit/f64 variant (run_unit) everything else;
//                   predecessors older than the ring are read back from HBM/L2 (the "deep" path).
//   Integer arithmetic only, except the reference's own f32 divide and (int)(dd * .01 * avg_qspan) in f64.
#include <hip/hip_runtime.h>
#include <mutex>
#include <utility>
#include <vector>
#include <stdint.h>
#include <limits.h>
#include "chaindp_kernels.h"
#include "chaindp_wave.h"
#include "chaindp_fast.h"

namespace chaindp {

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


// What one chunk of 64 predecessors (lane k <-> j = i-1-kb0-k) contributes before the serial semantics are applied.
struct Pairs {
	uint64_t ok;     // lanes that pass the filters of chain.c:252-261
	int sc;          // their score incl. f[j] (chain.c:262-273); INT_MIN on the other lanes
	int pj;          // p[j] (unit-relative) where ok
	bool cont;       // the chunk's last lane is still inside the window and the unit: another chunk may follow
	                 // (general/deep evaluation sets it; the fast evaluation leaves it to chunk_continues())
	uint32_t dr;     // fast evaluation only: x_i - x_j per lane
};

// General variant: the reference's formulas in 64-bit arithmetic.  DEEP = the predecessors are older than the
// ring and come from global memory.
template <int RING, bool DEEP>
__device__ __forceinline__ Pairs eval_general(const UnitCtx &c, const ulonglong2 &an, int ii, int qi, int span, int i, int kb0)
{
	constexpr int MASK = RING - 1;
	const int lane = c.lane;
	const uint64_t ri = readlane_u64(an.x, ii);
	const int sidi = seg_of_hi((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(an.y >> 32), ii));
	const bool inr = kb0 + lane < i;
	const int j = i - 1 - kb0 - lane;
	const int slot = j & MASK;
	uint64_t xj = 0;
	int qj = 0, fj = 0, pj = -1;
	uint32_t yhj = 0;
	if constexpr (!DEEP) {
		const uint4 e = *(const uint4*)(c.s_w + 4 * slot);
		xj = (uint64_t)c.s_xhi[slot] << 32 | e.x; qj = (int)e.y; fj = (int)e.z; pj = (int)e.w; yhj = c.s_yhi[slot];
	} else {
		wave_global_fence();
		if (inr) {
			const ulonglong2 aj = c.a[c.base + j];
			xj = aj.x; qj = (int)(uint32_t)aj.y; yhj = (uint32_t)(aj.y >> 32);
			fj = c.f[c.base + j];
			pj = c.p[c.base + j];
			pj = pj < 0 ? -1 : pj - c.rel0;                     // stored read-relative
		}
	}
	const uint64_t d64 = ri - xj;
	const bool live = inr && d64 <= c.maxx;                     // chain.c:252 (window) per lane
	const int dr = (int)d64;
	const int dq = (int)((uint32_t)qi - (uint32_t)qj);
	const bool same = seg_of_hi(yhj) == sidi;
	const int dd = dr > dq ? dr - dq : dq - dr;
	bool ok = live;
	ok = ok && !((same && dr == 0) || dq <= 0);                 // chain.c:257
	ok = ok && !((same && dq > c.mdy) || dq > c.mdx);           // chain.c:258
	ok = ok && !(same && dd > c.bw);                            // chain.c:260
	ok = ok && !(c.seg_rule && same && dr > c.mdy);             // chain.c:261
	int sc0 = dq < dr ? dq : dr;
	sc0 = sc0 > span ? span : sc0;
	const int sc = pair_score(sc0, dd, dr, dq, same, c.is_cdna, c.avgd) + fj;   // chain.c:264-273
	Pairs P;
	P.ok = __builtin_amdgcn_ballot_w64(ok);
	P.sc = ok ? sc : INT_MIN;
	P.pj = pj;
	P.cont = __builtin_amdgcn_ballot_w64(live) == ~0ull;
	P.dr = 0;
	return P;
}

// Applies the serial semantics of chain.c:274-281 to one evaluated chunk.  Marks whose target is still in the
// ring go to LDS.  Marks on older targets matter only if the scan later reaches a deep chunk; ring chunks do
// not write them (replay_far_marks does, on demand); deep chunks write all of theirs to the global array.
// Returns true when the scan for anchor i is complete (break taken, or window / unit exhausted).
template <int RING, bool DEEP>
__device__ __forceinline__ bool apply_chunk(const UnitCtx &c, const Pairs &P, int i, int kb0, int &max_f, int &max_j, int &n_skip)
{
	constexpr int MASK = RING - 1;
	const int lane = c.lane;
	const int tag = i;                                              // t[] starts at -1 (never a valid i)
	const int j = i - 1 - kb0 - lane;
	int tj = -1;
	if constexpr (!DEEP) {
		// lanes without a mark to make store into a dummy word instead of being masked off (no exec juggling)
		const int lo = i - RING > 0 ? i - RING : 0;
		const uint64_t near = P.ok & __builtin_amdgcn_ballot_w64(P.pj >= lo);
		int *dst = __builtin_amdgcn_inverse_ballot_w64(near) ? &c.s_t[P.pj & MASK] : c.s_dummy;
		*dst = tag;                                                                       // chain.c:281
		wave_mem_fence();
		tj = c.s_t[j & MASK];
	} else {
		if (__builtin_amdgcn_inverse_ballot_w64(P.ok) && P.pj >= 0) c.tg[c.base + P.pj] = c.tg_hi | (uint32_t)tag;
		wave_global_fence();
		if (kb0 + lane < i && c.tg[c.base + j] == (c.tg_hi | (uint32_t)tag)) tj = tag;
	}
	// new running max? strictly greater than everything before it (chain.c:274)
	const int incl = wave_scan_max(P.sc);
	int excl = wave_shift_up1(incl, max_f);
	excl = excl > max_f ? excl : max_f;
	const uint64_t A = P.ok & __builtin_amdgcn_ballot_w64(P.sc > excl);
	const uint64_t B = P.ok & ~A & __builtin_amdgcn_ballot_w64(tj == tag);                // chain.c:277
	// n_skip walk (chain.c:276,278): A lanes x -> max(x-1,0), B lanes x -> x+1, break when > max_skip
	const int hiA = highest_lane(A);                               // highest A lane (harmless -64 when A is empty)
	if ((B & low_mask64(hiA)) == 0) {                              // every A lane precedes every B lane (or one set is empty)
		if (A) {                                                   // the break, if any, is a B lane above every A lane: all A
			max_f = __builtin_amdgcn_readlane(P.sc, hiA);          // lanes count, and the last one holds the running max
			max_j = i - 1 - kb0 - hiA;
		}
		int x = n_skip - __builtin_popcountll(A);
		x = x < 0 ? 0 : x;
		const int cb = __builtin_popcountll(B);
		int need = c.max_skip - x + 1;
		need = need < 1 ? 1 : need;
		if (cb >= need) return true;                               // break taken (chain.c:278-279)
		n_skip = x + cb;
		return !P.cont;
	}
	// general: clamped walk via prefix min
	const bool isA = __builtin_amdgcn_inverse_ballot_w64(A), isB = __builtin_amdgcn_inverse_ballot_w64(B);
	const int S = n_skip + lanes_below(B) + (int)isB - lanes_below(A) - (int)isA;
	const int M = wave_scan_min(S);
	const int x = S - (M < 0 ? M : 0);
	const uint64_t m = B & __builtin_amdgcn_ballot_w64(x > c.max_skip);
	const uint64_t Ap = m ? (A & ((1ull << __builtin_ctzll(m)) - 1)) : A;   // A lanes before the break
	if (Ap) {
		const int ka = 63 - __builtin_clzll(Ap);
		max_f = __builtin_amdgcn_readlane(P.sc, ka);
		max_j = i - 1 - kb0 - ka;
	}
	if (m) return true;
	n_skip = __builtin_amdgcn_readlane(x, 63);
	return !P.cont;
}

// Before the first deep chunk of anchor i: write the marks of the ring chunks whose targets are older than the
// ring (skipped by apply_chunk<.., false>) into the global mark array.
template <int RING>
__device__ __forceinline__ void replay_far_marks(const UnitCtx &c, const ulonglong2 &an, int ii, int qi, int span, int i)
{
	for (int kb0 = 0; kb0 + 64 <= RING && kb0 < i; kb0 += 64) {
		const Pairs P = eval_general<RING, false>(c, an, ii, qi, span, i, kb0);
		if (__builtin_amdgcn_inverse_ballot_w64(P.ok) && P.pj >= 0 && P.pj < i - RING) c.tg[c.base + P.pj] = c.tg_hi | (uint32_t)i;
	}
}

// General variant of the unit loop: anything the fast variant (below) does not take.
template <int RING>
__device__ __forceinline__ void run_unit(const UnitCtx &c, int64_t room)
{
	constexpr int MASK = RING - 1;
	const int lane = c.lane;
	uint64_t x_carry = 0;
	for (int tile0 = 0;; tile0 += 64) {
		const int64_t gi = c.base + tile0 + lane;
		const bool have = tile0 + lane < room;
		ulonglong2 an = make_ulonglong2(0, 0);                 // (prefetching the next tile measured 1 % slower: A/B, tools/ab.sh)
		if (have) an = c.a[gi];
		// the unit ends at the first gap > max_dist_x (or at the end of the read)
		uint64_t xp;
		{
			const uint32_t lo = (uint32_t)wave_shift_up1((int)(uint32_t)an.x, (int)(uint32_t)x_carry);
			const uint32_t hi = (uint32_t)wave_shift_up1((int)(uint32_t)(an.x >> 32), (int)(uint32_t)(x_carry >> 32));
			xp = (uint64_t)hi << 32 | lo;
		}
		const bool stop = !have || ((tile0 + lane) > 0 && an.x - xp > c.maxx);
		const uint64_t stop_m = __builtin_amdgcn_ballot_w64(stop);
		const int cnt = stop_m ? __builtin_ctzll(stop_m) : 64;
		if (cnt == 0) break;
		x_carry = readlane_u64(an.x, 63);

		// v[i] = max(v[max_j], f[i]) (chain.c:284) is off the recurrence's critical path: the read of v[max_j]
		// is issued at the end of step i and consumed one step later.
		int pend_slot = -1, pend_mf = 0, pend_vj = INT_MIN;
		for (int ii = 0; ii < cnt; ++ii) {
			const int i = tile0 + ii;                          // unit-relative index of the anchor being scored
			const int qi = __builtin_amdgcn_readlane((int)(uint32_t)an.y, ii);
			const int span = span_of_hi((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(an.y >> 32), ii));
			int max_f = span, max_j = -1;
			{
				// chunk 0 (the 64 nearest predecessors) settles most anchors; further chunks are the exception.  It also
				// runs for the unit's first anchor: every lane then fails the window / range test and nothing happens.
				int n_skip = 0;
				Pairs P0;
				P0 = eval_general<RING, false>(c, an, ii, qi, span, i, 0);
				const bool done0 = apply_chunk<RING, false>(c, P0, i, 0, max_f, max_j, n_skip);
				if (__builtin_expect(!done0 && i > 64, 0)) {
					for (int kb0 = 64; kb0 < i; kb0 += 64) {
						bool done;
						if (kb0 + 64 <= RING) {
							Pairs P;
							P = eval_general<RING, false>(c, an, ii, qi, span, i, kb0);
							done = apply_chunk<RING, false>(c, P, i, kb0, max_f, max_j, n_skip);
						} else {
							if (kb0 == RING) replay_far_marks<RING>(c, an, ii, qi, span, i);
							const Pairs P = eval_general<RING, true>(c, an, ii, qi, span, i, kb0);
							done = apply_chunk<RING, true>(c, P, i, kb0, max_f, max_j, n_skip);
						}
						if (done) break;
					}
				}
			}
			// epilogue (chain.c:283-284): anchor i enters the ring; v of the previous anchor is completed
			const int vprev = pend_vj > pend_mf ? pend_vj : pend_mf;
			wave_mem_fence();
			if (lane == ii) {
				*(uint4*)(c.s_w + 4 * (i & MASK)) = make_uint4((uint32_t)an.x, (uint32_t)an.y, (uint32_t)max_f, (uint32_t)max_j);
				c.s_xhi[i & MASK] = (uint32_t)(an.x >> 32); c.s_yhi[i & MASK] = (uint32_t)(an.y >> 32);
				if (pend_slot >= 0) c.s_v[pend_slot] = vprev;
			}
			wave_mem_fence();
			pend_slot = i & MASK; pend_mf = max_f; pend_vj = INT_MIN;
			if (max_j >= 0) {
				if (i - max_j <= RING) pend_vj = c.s_v[max_j & MASK];
				else { wave_global_fence(); pend_vj = c.v[c.base + max_j]; }
			}
		}
		{
			const int vprev = pend_vj > pend_mf ? pend_vj : pend_mf;
			wave_mem_fence();
			if (lane == 0) c.s_v[pend_slot] = vprev;
			wave_mem_fence();
		}
		{
			bool not_self = false;
			if (lane < cnt) {
				const int my_slot = (tile0 + lane) & MASK;
				not_self = !(c.s_v[my_slot] >= c.min_sc || (int)c.s_w[4 * my_slot + 3] >= 0);
				if (not_self) c.first_child[gi] = NO_CHILD;                // see run_unit_fast
			}
			if (__builtin_amdgcn_ballot_w64(not_self)) wave_global_fence();
		}
		if (lane < cnt) {
			const int my_slot = (tile0 + lane) & MASK;
			const int2 fp = *(const int2*)(c.s_w + 4 * my_slot + 2);
			const int vi = c.s_v[my_slot];
			c.f[gi] = fp.x;
			c.p[gi] = fp.y < 0 ? -1 : fp.y + c.rel0;
			c.v[gi] = vi;
			// Compaction (chain.c:286-317) needs, for every anchor that is not emitted at its own step, its
			// first child; while f/p/v of the tile are at hand, record "emitted at own step" and feed that min.
			const int q = fp.y;
			int maybe_first = 0;
			if (q >= 0) {
				int vq, pq;
				if (tile0 + cnt - 1 - q < RING) { vq = c.s_v[q & MASK]; pq = (int)c.s_w[4 * (q & MASK) + 3]; }
				else { vq = c.v[c.base + q]; pq = c.p[c.base + q]; }
				if (!(vq >= c.min_sc || pq >= 0)) { atomicMin(&c.first_child[c.base + q], c.rel0 + tile0 + lane); maybe_first = 4; }
			}
			c.flags[gi] = (uint8_t)(((vi >= c.min_sc || q >= 0) ? 2 : 0) | maybe_first | (vi >= c.min_sc ? 8 : 0) | (fp.x < vi ? 16 : 0));
		}
		if (cnt < 64) break;
	}
}

// ---------------------------------------------------------------- K1, fast variant
// Ordinary reads (not cDNA, one segment, bw within the table) take this path.  It computes exactly what
// run_unit<RING, true> computes, with the per-anchor instruction count cut to the bone (the kernel is bound by
// VALU and SALU issue, not by memory):
//   * LDS is addressed with raw byte addresses (the kernel has no static LDS, so the dynamic segment starts at 0;
//     launch_chain checks that), which lets constant offsets fold into the DS instructions;
//   * the ring entry holds x.lo+1, qpos+1, f and 4*p: subtracting it from the raw x_i, q_i gives dr-1 and dq-1,
//     so the range tests "1 <= d <= max" become single unsigned compares, |dr-dq| is unchanged, and
//     min(dq,dr,span) + f - cost = min3(dq-1,dr-1,span-1) + f + (1-cost); 4*p is the byte offset of the mark to write;
//   * the per-anchor scalars x_i, q_i, span come through the scalar cache (s_load, one anchor ahead) instead of
//     v_readlane: the kernel is VALU-bound and the scalar side has room;
//   * the three filters of chain.c:252-260 are one compare: max3(dr-1, sat(dq-1 + (max_x - max_q)),
//     |dr-dq| + (max_x-1-bw)) < max_x (no wrap-around matters: when the first two are below max_x, so is |dr-dq|);
//   * the cost table holds 1-cost as int16, so the score is one three-operand add;
//   * prefix-max lanes without a source read 0 instead of INT_MIN: the running max is >= q_span >= 0, so a
//     floor of 0 changes nothing and saves the copy in front of the DPP chain;
//   * v[] (chain.c:284) is not part of the recurrence at all: v[i] = max(f[i], v[p[i]]) is computed per 64-anchor
//     tile at flush time by pointer doubling over the tile (6 rounds of ds_bpermute), not per anchor.

// One chunk of 64 ring predecessors of anchor i (lane k <-> j = jtop - k, S = 16 * jtop): scores, marks, and the
// lane masks A ("new running max", chain.c:274) and B ("marked and not better", chain.c:277).  Straight-line code.

template <int RING, bool SAMEGAP>
__device__ __forceinline__ FastMasks fast_masks(const FastK &k, uint32_t S, uint32_t xm1, uint32_t qm1, int spm1, int i, int kb0, int max_f)
{
	typedef FastLds<RING> L;
	const uint32_t addr = (S - k.L4) & (L::RB - 1u);
	const FastPairs P = fast_filters<RING, SAMEGAP>(k, addr, xm1, qm1);
	const int dqm1 = (int)P.e.y, drm1 = (int)P.drm1;
	int sc0 = dqm1 < drm1 ? dqm1 : drm1;
	sc0 = sc0 < spm1 ? sc0 : spm1;                                                      // chain.c:262-263, minus one
	const uint32_t di = P.dd < k.bw ? P.dd : k.bw;
	const int scu = sc0 + (int)P.e.z + lds_load_i16(L::LUT + 2u * di);                  // chain.c:272-273 via the table
	FastMasks m;
	const uint64_t okm = __builtin_amdgcn_ballot_w64(P.ok);
	m.sc = __builtin_amdgcn_inverse_ballot_w64(okm) ? scu : INT_MIN;    // (through the mask: two selects on one condition get
	                                                                    // turned into a divergent branch around the table lookup)
	m.drm1 = P.drm1;
	// marks (chain.c:281), kept by distance: the mark on p_j goes to word i-1-p_j, lane k of chunk kb0 owns word kb0+k.
	// No address arithmetic on the reading side, and "older than the ring" is a clamp to the dummy word, where
	// the lanes without a mark to make store as well (nothing is masked off).  p_j = -1 gives word i, which belongs
	// to the non-existent anchor -1.
	const uint32_t d4 = ((uint32_t)(i - 1) << 2) - P.e.w;
	const uint32_t dcl = d4 < 4u * RING ? d4 : 4u * RING;
	const uint32_t dst = P.ok ? dcl : k.far4;
	lds_store_b32(dst + L::T_OFF, i);
	wave_mem_fence();
	const int tj = lds_load_b32(k.trel + ((uint32_t)kb0 << 2) + L::T_OFF);
	int excl = wave_excl_max_floor0(m.sc);
	excl = excl > max_f ? excl : max_f;
	m.A = __builtin_amdgcn_ballot_w64(m.sc > excl);                                     // masked lanes hold INT_MIN
	m.B = okm & ~m.A & __builtin_amdgcn_ballot_w64(tj == i);
	return m;
}


// a further ring chunk (kb0 >= 64); returns true when the scan for anchor i is complete
template <int RING, bool SAMEGAP>
__device__ __forceinline__ bool fast_chunk(const FastK &k, uint32_t S, int jtop, uint32_t xm1, uint32_t qm1, int spm1,
                                           int i, int kb0, int &max_f, int &max_j, int &n_skip)
{
	const FastMasks m = fast_masks<RING, SAMEGAP>(k, S, xm1, qm1, spm1, i, kb0, max_f);
	if (fast_walk(k, m, jtop, max_f, max_j, n_skip)) return true;
	// x sorted => dr grows with the lane: another chunk can only matter if the last lane is inside the window
	return (uint32_t)__builtin_amdgcn_readlane((int)m.drm1, 63) + 1u > k.M;
}

// A deep chunk (predecessors older than the ring) of a table-driven unit: the same arithmetic as fast_masks with
// a[j], f[j], p[j] read back from HBM/L2 and marks in the global array.  Only the window test is done in 64 bits
// (x_i - x_j of a predecessor this old may exceed 32 bits; for a lane inside the window it does not, and every
// other difference is bounded by the window).  Returns true when the scan for anchor i is complete.
template <int RING, bool SAMEGAP>
__device__ __forceinline__ bool fast_deep_chunk(const UnitCtx &c, const FastK &k, uint64_t xi, uint32_t qi, int spm1, int i, int kb0,
                                                int &max_f, int &max_j, int &n_skip)
{
	typedef FastLds<RING> L;
	const int j = i - 1 - kb0 - c.lane;
	const bool inr = j >= 0;
	const int64_t gj = c.base + (inr ? j : 0);
	wave_global_fence();                                           // f/p of earlier tiles and the marks written so far
	const ulonglong2 aj = c.a[gj];
	const int fj = c.f[gj];
	const int pjr = c.p[gj];                                       // stored read-relative
	const bool live = inr && xi - aj.x <= c.maxx;                  // chain.c:252
	const uint32_t drm1 = (uint32_t)xi - (uint32_t)aj.x - 1u, dqm1 = qi - (uint32_t)aj.y - 1u;
	const uint32_t dd = absdiff_u32(drm1, dqm1);
	const uint32_t dqs = SAMEGAP ? dqm1 : __builtin_elementwise_add_sat(dqm1, k.dq_off);
	const uint32_t m2 = drm1 > dqs ? drm1 : dqs, t = dd + k.cbw;
	const bool ok = live && (m2 > t ? m2 : t) < k.M;               // chain.c:257-260 (same single compare as fast_filters)
	int sc0 = (int)dqm1 < (int)drm1 ? (int)dqm1 : (int)drm1;
	sc0 = sc0 < spm1 ? sc0 : spm1;
	const uint32_t di = dd < k.bw ? dd : k.bw;
	const int scu = sc0 + fj + lds_load_i16(L::LUT + 2u * di);
	FastMasks m;
	const uint64_t okm = __builtin_amdgcn_ballot_w64(ok);
	m.sc = __builtin_amdgcn_inverse_ballot_w64(okm) ? scu : INT_MIN;
	m.drm1 = drm1;
	const unsigned long long tag = c.tg_hi | (uint32_t)i;
	if (ok && pjr >= 0) c.tg[c.base + (pjr - c.rel0)] = tag;       // chain.c:281: a predecessor this old has an older one still
	wave_global_fence();
	const bool marked = inr && c.tg[gj] == tag;
	int excl = wave_excl_max_floor0(m.sc);
	excl = excl > max_f ? excl : max_f;
	m.A = __builtin_amdgcn_ballot_w64(m.sc > excl);
	m.B = okm & ~m.A & __builtin_amdgcn_ballot_w64(marked);
	if (fast_walk(k, m, i - 1 - kb0, max_f, max_j, n_skip)) return true;
	return __builtin_amdgcn_ballot_w64(live) != ~0ull;             // a lane outside the window (or the unit): nothing older can matter
}

// chunks beyond the first for anchor i: ring chunks, then the deep path (predecessors older than the ring, from HBM/L2)
template <int RING, bool SAMEGAP>
__device__ __forceinline__ void fast_more_chunks(const UnitCtx &c, const FastK &k, uint32_t xhi, int i, uint32_t xm1, uint32_t qm1,
                                                 int spm1, int &max_f, int &max_j, int &n_skip)
{
	typedef FastLds<RING> L;
	const int lo4 = max((i - RING) << 2, 0);           // 4 * (oldest anchor still in the ring)
	for (int kb0 = 64; kb0 < i; kb0 += 64) {
		bool done;
		if (kb0 + 64 <= RING) {
			done = fast_chunk<RING, SAMEGAP>(k, (uint32_t)(i - 1 - kb0) << 4, i - 1 - kb0, xm1, qm1, spm1, i, kb0, max_f, max_j, n_skip);
		} else {
			if (kb0 == RING) {
				++c.deep_n;
				// first deep chunk: marks of the ring chunks whose targets are older than the ring go to the global array now
				for (int kr = 0; kr < RING; kr += 64) {
					const FastPairs P = fast_filters<RING, SAMEGAP>(k, ((uint32_t)((i - 1 - kr) << 4) - k.L4) & (L::RB - 1u), xm1, qm1);
					const int pj4 = (int)P.e.w;
					if (P.ok && pj4 >= 0 && pj4 < lo4) c.tg[c.base + (pj4 >> 2)] = c.tg_hi | (uint32_t)i;
				}
			}
			done = fast_deep_chunk<RING, SAMEGAP>(c, k, (uint64_t)xhi << 32 | xm1, qm1, spm1, i, kb0, max_f, max_j, n_skip);
		}
		if (done) break;
	}
}

// One anchor of a tile: scan (chunk 0 inline, further chunks out of line), then the anchor enters the ring.
// a_cur holds this anchor's mm128_t as four scalar dwords; a_next receives the next anchor's (scalar load issued here).
template <int RING, bool SAMEGAP>
__device__ __forceinline__ void fast_anchor_step(const UnitCtx &c, const FastK &k, const ulonglong2 &an, int tile0, int ii, const u32x4_t &a_cur,
                                                 u32x4_t &a_next, const char *ap, uint32_t &off_next, uint32_t off_last, uint32_t waddr, u32x2_t xq)
{
	const int i = tile0 + ii;                          // unit-relative index of the anchor being scored
	const uint32_t xm1 = a_cur.x, qm1 = a_cur.z;       // ring entries hold x+1 and q+1
	asm volatile("" :: "s"(a_cur.y));                  // x.hi is not needed, but its register must stay reserved until the load
	                                                   // has landed (a reuse would force a wait right behind the s_load)
	const int spm1 = span_of_hi(a_cur.w) - 1;
	off_next = off_next + 16u < off_last ? off_next + 16u : off_last;
	int max_f = spm1 + 1, max_j = -1, n_skip = 0;
	// chunk 0 (the 64 nearest predecessors) settles most anchors.  It also runs for the unit's first anchors:
	// slots not written yet fail the window test, so nothing happens on those lanes.  Written out with
	// explicit exits so that the common path (break inside chunk 0) is straight-line scalar code.
	{
		const FastMasks m = fast_masks<RING, SAMEGAP>(k, (uint32_t)(i - 1) << 4, xm1, qm1, spm1, i, 0, max_f);
		// next anchor's scalars: issued only now (the address is made to depend on B), after this step's last LDS
		// wait, so that no LDS wait of this step also waits for the scalar load
		asm volatile("" : "+s"(off_next) : "s"(m.B));
		a_next = *(const u32x4_t*)(ap + off_next);
		const int hiA = highest_lane(m.A);
		// every A lane precedes every B lane, or B is empty (s_ff1 gives -1, the largest unsigned).  With A empty and B
		// not, hiA is -64 and the (equally correct) general walk runs.
		if ((uint32_t)lowest_lane(m.B) > (uint32_t)hiA) {
			if (hiA >= 0) {
				max_f = __builtin_amdgcn_readlane(m.sc, hiA);
				max_j = i - 1 - hiA;
			}
			n_skip = __builtin_popcountll(m.B);                // n_skip was 0: A lanes cannot lower it
			if (n_skip > k.ms0) goto anchor_done;              // break taken at a B lane (chain.c:278-279)
			if ((uint32_t)__builtin_amdgcn_readlane((int)m.drm1, 63) + 1u > k.M) goto anchor_done;   // window exhausted
			fast_more_chunks<RING, SAMEGAP>(c, k, a_cur.y, i, xm1, qm1, spm1, max_f, max_j, n_skip);
		} else {
			// A and B lanes interleave (rare): own copy of the tail, so that the common path above shares no
			// control flow (and no merged exit flags) with it
			if (fast_walk_general(k, m, i - 1, max_f, max_j, n_skip)) goto anchor_done;
			if ((uint32_t)__builtin_amdgcn_readlane((int)m.drm1, 63) + 1u > k.M) goto anchor_done;
			fast_more_chunks<RING, SAMEGAP>(c, k, a_cur.y, i, xm1, qm1, spm1, max_f, max_j, n_skip);
		}
	}
anchor_done:
	// anchor i enters the ring (chain.c:283): a single-lane store with the exec mask set by hand (no branch in the IR,
	// so the anchor loop has only wave-uniform control flow; exec is all ones here: 64-thread workgroups, uniform
	// branches only).  (One v_mov_b64 + ds_write2_b64 instead of two v_mov_b32 + ds_write_b128 measured 1 % slower.)
	{
		u32x4_t w4; w4.x = xq.x; w4.y = xq.y; w4.z = (uint32_t)max_f; w4.w = (uint32_t)(max_j << 2);
		wave_mem_fence();
		asm volatile("s_bfm_b64 exec, 1, %0\n\tds_write_b128 %1, %2\n\ts_mov_b64 exec, -1" :: "s"(ii), "v"(waddr), "v"(w4) : "memory");
		wave_mem_fence();
	}
}

// resume > 0 (a multiple of 64): the unit's anchors [0, resume) are done -- k_chain_twin scored and flushed them before it handed the
// unit over -- and this wave goes on from there: the ring gets their last RING entries back from a, f, p, v in HBM.  (Marks are per
// scan, so nothing else carries over.)
template <int RING, bool SAMEGAP>
__device__ __forceinline__ void run_unit_fast(const UnitCtx &c, int64_t room, int resume = 0)
{
	typedef FastLds<RING> L;
	constexpr int MASK = RING - 1;
	const int lane = c.lane;
	FastK k;
	k.L4 = (uint32_t)lane << 4;
	k.far4 = 4u * RING;
	k.trel = (uint32_t)lane << 2;
	asm volatile("" : "+v"(k.L4), "+v"(k.far4), "+v"(k.trel));   // opaque: keeps (S - 16*lane) & mask at two instructions and
	                                                             // the constant LDS offsets in the DS offset fields
	k.M = (uint32_t)c.maxx;
	k.bw = (uint32_t)c.bw;
	k.cbw = k.M - 1u > k.bw ? k.M - 1u - k.bw : 0u;
	k.dq_off = k.M - (uint32_t)c.mdq;
	k.max_skip = c.max_skip;
	k.ms0 = c.max_skip > 0 ? c.max_skip : 0;
	uint64_t x_carry = 0;
	if (resume > 0) {
		for (int j = resume - RING + lane; j < resume; j += 64) if (j >= 0) {
			const ulonglong2 aj = c.a[c.base + j];
			const int pj = c.p[c.base + j];                            // stored read-relative, -1 = none
			lds_store_b128(((uint32_t)(j & MASK) << 4), make_uint4((uint32_t)aj.x + 1u, (uint32_t)aj.y + 1u, (uint32_t)c.f[c.base + j],
			                                                       pj < 0 ? 0xfffffffcu : (uint32_t)(pj - c.rel0) << 2));
			lds_store_b32(L::V_OFF + ((uint32_t)(j & MASK) << 2), c.v[c.base + j]);
		}
		x_carry = c.a[c.base + resume - 1].x;                          // (the same address in every lane)
		wave_mem_fence();
	}
	for (int tile0 = resume;; tile0 += 64) {
		const int64_t gi = c.base + tile0 + lane;
		const bool have = tile0 + lane < room;
		ulonglong2 an = make_ulonglong2(0, 0);
		if (have) an = c.a[gi];
		uint64_t xp;
		{
			const uint32_t lo = (uint32_t)wave_shift_up1((int)(uint32_t)an.x, (int)(uint32_t)x_carry);
			const uint32_t hi = (uint32_t)wave_shift_up1((int)(uint32_t)(an.x >> 32), (int)(uint32_t)(x_carry >> 32));
			xp = (uint64_t)hi << 32 | lo;
		}
		const bool stop = !have || ((tile0 + lane) > 0 && an.x - xp > c.maxx);
		const uint64_t stop_m = __builtin_amdgcn_ballot_w64(stop);
		const int cnt = stop_m ? __builtin_ctzll(stop_m) : 64;
		if (cnt == 0) break;
		x_carry = readlane_u64(an.x, 63);

		// the per-anchor scalars (x_i-1, q_i-1, span-1) come through the scalar cache, one anchor ahead of their use:
		// SMEM + SALU work instead of three v_readlane (the kernel is VALU-bound)
		const char *ap = (const char*)(c.a + (c.base + tile0));
		const uint32_t off_last = (uint32_t)(cnt - 1) << 4;
		uint32_t off_next = 0;
		u32x4_t a_cur = *(const u32x4_t*)ap;
		const uint32_t waddr = (uint32_t)((tile0 + lane) & MASK) << 4;
		u32x2_t xq;                                            // first half of this lane's ring entry: x.lo + 1, qpos + 1
		xq.x = (uint32_t)an.x + 1u; xq.y = (uint32_t)an.y + 1u;
		// two steps per trip with the scalar registers swapped, so that the prefetched anchor is used in place (a copy at
		// the end of the step would need the load to have landed by then)
		{
			u32x4_t a_alt;
			int ii = 0;
			for (; ii + 2 <= cnt; ii += 2) {
				fast_anchor_step<RING, SAMEGAP>(c, k, an, tile0, ii, a_cur, a_alt, ap, off_next, off_last, waddr, xq);
				fast_anchor_step<RING, SAMEGAP>(c, k, an, tile0, ii + 1, a_alt, a_cur, ap, off_next, off_last, waddr, xq);
			}
			if (ii < cnt) fast_anchor_step<RING, SAMEGAP>(c, k, an, tile0, ii, a_cur, a_alt, ap, off_next, off_last, waddr, xq);
		}
		fast_flush_tile<RING>(c, tile0, cnt, waddr, gi);      // v (chain.c:284), f/p/v, the compaction helpers
		if (cnt < 64) break;
		// A unit whose scans keep reaching past the ring (dense repeats: the window holds hundreds of predecessors and few of
		// them are marked) spends its time in round trips to L2.  It is handed to k_chain_dense, which redoes it from scratch;
		// what this wave has stored so far is what that kernel stores again.
		if (c.deep_list && c.deep_n >= c.deep_min && (2 * c.deep_n >= tile0 + 64 || c.deep_left == 0) && room - tile0 >= c.deep_left && tile0 + 64 < room && room <= CHAINDP_DENSE_BITCAP &&
		    (unsigned int)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(c.deep_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < c.deep_cap) {
			if (lane == 0) { Unit un; un.start = c.base; un.read = c.read; un.len = (int32_t)room; c.deep_list[atomicAdd(c.deep_cnt, 1u)] = un; }
			return;
		}
	}
}

template <int RING>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_chain_units(Params par, const int64_t *__restrict__ off,
                                                    const ulonglong2 *__restrict__ a, const int32_t *__restrict__ n_segs_pr,
                                                    const unsigned long long *__restrict__ sumq,
                                                    const uint16_t *__restrict__ lut, int lut_stride,
                                                    const Unit *__restrict__ units,
                                                    const unsigned long long *__restrict__ counters,
                                                    int32_t *f, int32_t *p, int32_t *v, unsigned long long *tg, uint32_t epoch,
                                                    int32_t *first_child, uint8_t *flags,
                                                    const Unit *__restrict__ units_all, const unsigned long long *__restrict__ counters_all,
                                                    Unit *deep_list, unsigned int *deep_cnt, const unsigned int *__restrict__ long_units, int deep_eager, int deep_route)
{
	static_assert((RING & (RING - 1)) == 0 && RING >= 128, "RING must be a power of two >= 128");
	extern __shared__ uint4 smem[];
	UnitCtx c;
	c.a = a; c.f = f; c.p = p; c.v = v; c.tg = tg; c.tg_hi = (unsigned long long)epoch << 32; c.first_child = first_child; c.flags = flags; c.min_sc = par.min_sc;
	c.s_w = (uint32_t*)smem;
	c.s_t = (int*)(c.s_w + 4 * RING);
	c.s_v = c.s_t + RING;
	c.s_xhi = (uint32_t*)(c.s_v + RING);
	c.s_yhi = c.s_xhi + RING;
	c.s_dummy = (int*)(c.s_yhi + RING);                         // 16 B: one dummy word + padding
	uint16_t *s_lut = (uint16_t*)(c.s_yhi + RING + 4);
	c.s_lut = s_lut;
	c.lane = threadIdx.x;
	// k_chain_dense keeps 32-bit differences over a ring of CHAINDP_DENSE_RING anchors: exact under the same condition as x32_ok below
	c.deep_list = ((uint64_t)(int64_t)par.max_dist_x + 1) * (uint64_t)(CHAINDP_DENSE_RING + 1) < (1ull << 32) ? deep_list : nullptr;
	// Which kernel takes them depends on the batch: with a few long units it has a tail, and k_chain_dense puts eight waves on each
	// of the first CHAINDP_DENSE_UNITS handed over (units run longest first); with thousands of them (*long_units: units of 8192
	// anchors and more, from the prepass' length classes) every SIMD is busy to the end, the batch is bound by what a pair evaluation
	// costs, and k_chain_dense1 takes every unit that qualifies, one wave each
	c.deep_cap = dense_all(long_units, deep_route) ? 0xffffffffu : CHAINDP_DENSE_UNITS;
	c.deep_min = deep_eager ? 8 : CHAINDP_DEEP_HANDOVER; c.deep_left = deep_eager ? 0 : CHAINDP_DEEP_HANDOVER_LEFT;
	c.deep_cnt = deep_cnt;
	c.maxx = (uint64_t)(int64_t)par.max_dist_x;
	c.mdx = par.max_dist_x; c.mdy = par.max_dist_y; c.bw = par.bw; c.max_skip = par.max_skip; c.is_cdna = par.is_cdna;
	c.mdq = par.max_dist_x < par.max_dist_y ? par.max_dist_x : par.max_dist_y;   // dq > max_dist_y || dq > max_dist_x (same segment)
	// 32-bit differences are exact while (RING + 1) * max_dist_x + 2 < 2^32: consecutive anchors of a unit are at
	// most max_dist_x apart, the ring spans RING of them, and unwritten slots sit max_dist_x + 2 below the unit start
	const bool x32_ok = ((uint64_t)(int64_t)par.max_dist_x + 1) * (uint64_t)(RING + 1) < (1ull << 32);
	const int lane = threadIdx.x;

	int64_t n_units = (int64_t)(uint32_t)counters[0];             // low word: units, high word: singletons (prepass)
	// launched on the list of units k_chain_twin handed over: a count of all ones there means "every unit of the batch,
	// in the prepass' (longest first) order"
	if (units_all && (uint32_t)counters[0] == 0xffffffffu) { units = units_all; n_units = (int64_t)(uint32_t)counters_all[0]; }
	for (int64_t ub = blockIdx.x; ub < n_units; ub += gridDim.x) {
		Unit u = units[ub];
		// a unit k_chain_twin handed over after it had flushed some of its tiles carries that many anchors in the high word of its
		// start (anchor indices fit 31 bits): the table-driven variant goes on from there instead of starting over
		const int resume = (int)((uint64_t)u.start >> 32);
		u.start = (int64_t)((uint64_t)u.start & 0xffffffffull);
		const int64_t rs = off[u.read], re = off[u.read + 1];
		const unsigned long long sq = sumq[u.read];
		const int n_segs = n_segs_pr ? n_segs_pr[u.read] : par.n_segs;
		c.base = u.start; c.read = u.read; c.deep_n = 0;
		c.rel0 = (int)(u.start - rs);
		c.avgd = (double)((float)(uint64_t)(sq & ~SUMQ_FLAGS) / (float)(int64_t)(re - rs));   // chain.c:241: f32 divide of converted u64 and i64
		c.seg_rule = n_segs > 1 && !par.is_cdna;                   // chain.c:261
		const bool general = par.is_cdna || n_segs > 1 || (sq & SUMQ_SEG_FLAG) || lut == nullptr || !x32_ok
		                     || par.max_dist_x < 1 || par.max_dist_y < 0;

		wave_mem_fence();
		for (int k = lane; k < RING; k += 64) c.s_t[k] = -1;
		if (!general) {
			const uint4 *src = (const uint4*)(lut + (int64_t)u.read * lut_stride);   // lut_stride is a multiple of 8 entries (16 B)
			for (int k = lane; k * 8 < lut_stride; k += 64) ((uint4*)s_lut)[k] = src[k];
			const uint32_t x_none = (uint32_t)a[u.start].x - (uint32_t)c.maxx - 1u;  // "no anchor here yet" (x+1 encoding): fails the window test
			for (int k = lane; k < RING; k += 64) *(uint4*)(c.s_w + 4 * k) = make_uint4(x_none, 0u, 0u, 0xfffffffcu);
			for (int k = lane; k < RING; k += 64) ((int*)c.s_yhi)[k] = -1;          // the fast variant's mark words (FastLds::T_OFF)
		}
		wave_mem_fence();
		// u.len bounds the unit (next unit's start or the read's end); run_unit finds the true end at the first gap
		if (general) run_unit<RING>(c, (int64_t)u.len);
		else if (par.max_dist_y >= par.max_dist_x) run_unit_fast<RING, true>(c, (int64_t)u.len, resume);
		else run_unit_fast<RING, false>(c, (int64_t)u.len, resume);
	}
}

// ---------------------------------------------------------------- launchers

hipError_t check_no_static_lds(const void *fn)
{
	static std::mutex mu;
	static std::vector<std::pair<const void*, hipError_t>> seen;
	std::lock_guard<std::mutex> lk(mu);
	for (const auto &kv : seen) if (kv.first == fn) return kv.second;
	hipFuncAttributes fa;
	hipError_t e = hipFuncGetAttributes(&fa, fn);
	if (e != hipSuccess) return e;                                 // (not cached: a failed query is tried again)
	e = fa.sharedSizeBytes != 0 ? hipErrorInvalidConfiguration : hipSuccess;
	seen.emplace_back(fn, e);
	return e;
}

size_t chain_lds_bytes(int ring, int lut_stride)
{
	return (size_t)ring * 32 + 16 + (size_t)lut_stride * 2;
}
hipError_t launch_chain(hipStream_t st, int ring, const Params &par, int64_t max_units, const int64_t *d_off, const void *d_a,
                        const int32_t *d_n_segs, const unsigned long long *d_sumq, const uint16_t *d_lut, int lut_stride,
                        const Unit *d_units, const unsigned long long *d_counters,
                        int32_t *d_f, int32_t *d_p, int32_t *d_v, unsigned long long *d_tg, uint32_t epoch, int32_t *d_first_child, uint8_t *d_flags,
                        const Unit *d_units_all, const unsigned long long *d_counters_all, Unit *d_deep, unsigned int *d_deep_cnt,
                        const unsigned int *d_long_units, int deep_eager, int deep_route)
{
	if (max_units <= 0) return hipSuccess;
	// The number of units is only known on the device (counters[0]); the grid is sized for the upper
	// bound and blocks beyond the count exit at once, so no host round trip sits between the kernels.
	int64_t blocks = max_units;
	const int64_t cap = 256LL * 32 * 16;
	if (blocks > cap) blocks = cap;
	const ulonglong2 *aa = (const ulonglong2*)d_a;
	const size_t lds = chain_lds_bytes(ring, d_lut ? lut_stride : 0);
	{
		const void *fn = ring == 128 ? (const void*)k_chain_units<128> : ring == 512 ? (const void*)k_chain_units<512> : (const void*)k_chain_units<256>;
		const hipError_t e = check_no_static_lds(fn);        // LDS is addressed by raw byte offsets from 0
		if (e != hipSuccess) return e;
	}
	switch (ring) {
	case 128: hipLaunchKernelGGL(k_chain_units<128>, dim3((unsigned)blocks), dim3(64), lds, st, par, d_off, aa, d_n_segs, d_sumq, d_lut, lut_stride, d_units, d_counters, d_f, d_p, d_v, d_tg, epoch, d_first_child, d_flags, d_units_all, d_counters_all, d_deep, d_deep_cnt, d_long_units, deep_eager, deep_route); break;
	case 512: hipLaunchKernelGGL(k_chain_units<512>, dim3((unsigned)blocks), dim3(64), lds, st, par, d_off, aa, d_n_segs, d_sumq, d_lut, lut_stride, d_units, d_counters, d_f, d_p, d_v, d_tg, epoch, d_first_child, d_flags, d_units_all, d_counters_all, d_deep, d_deep_cnt, d_long_units, deep_eager, deep_route); break;
	default:  hipLaunchKernelGGL(k_chain_units<256>, dim3((unsigned)blocks), dim3(64), lds, st, par, d_off, aa, d_n_segs, d_sumq, d_lut, lut_stride, d_units, d_counters, d_f, d_p, d_v, d_tg, epoch, d_first_child, d_flags, d_units_all, d_counters_all, d_deep, d_deep_cnt, d_long_units, deep_eager, deep_route); break;
	}
	return hipGetLastError();
}

} // namespace chaindp
