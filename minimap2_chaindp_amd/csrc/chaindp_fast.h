// chaindp_fast.h -- device code shared by the table-driven chain DP kernels: k_chain_units (chaindp_kernels.hip, one wave per
// unit) and k_chain_dense (chaindp_dense.hip, the units whose scans run deep).  Per-unit context, LDS layout and raw LDS
// access, the pair filters of chain.c:252-260 as one compare, the n_skip walk of chain.c:274-279 on lane masks, and the tile
// flush (v[] by pointer doubling, f/p/v stores, compaction helpers).  gfx950 only.
#ifndef CHAINDP_FAST_H
#define CHAINDP_FAST_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "chaindp_kernels.h"
#include "chaindp_wave.h"

namespace chaindp {

// Per-unit constants and the LDS carve-up.  Ring entry k (16 B): x.lo, qpos, f, p (unit-relative);
// side arrays: mark tag t[], v[], and (general variant only) x.hi[], y.hi[]; then the read's cost table.
// During step i the ring holds anchors i-RING .. i-1 (entry i is written at the end of step i).
struct UnitCtx {
	const ulonglong2 *a;
	int32_t *f, *p, *v;
	unsigned long long *tg;   // global mark array (deep path): (run epoch << 32 | tag), so it is never re-initialised
	unsigned long long tg_hi; // run epoch << 32
	int32_t *first_child;   // compaction helper, see chaindp_compact.hip
	uint8_t *flags;
	int min_sc;
	uint32_t *s_w;          // ring entries, 4 dwords each
	int *s_t, *s_v;
	uint32_t *s_xhi, *s_yhi;
	const uint16_t *s_lut;
	int *s_dummy;           // sink for lanes that have no mark to write
	int64_t base;
	uint64_t maxx;
	double avgd;
	int rel0, lane, read;
	int mdx, mdy, mdq, bw, max_skip, is_cdna;
	bool seg_rule;
	// units whose scans keep reaching past the ring are handed to k_chain_dense (chaindp_dense.hip): list, count, and the
	// scans of the current unit that went past the ring so far (nullptr: this launch keeps every unit)
	Unit *deep_list;
	unsigned int *deep_cnt;
	unsigned int deep_cap;          // units handed over per batch at most
	int deep_min, deep_left;        // CHAINDP_DEEP_HANDOVER, CHAINDP_DEEP_HANDOVER_LEFT (tests: 8, 0 -- any unit with a few deep scans)
	mutable int deep_n;
};

// k_chain_units hands a table-driven unit to k_chain_dense once this many of its scans went past the ring, if that is at
// least every second anchor so far and the unit may have this many anchors left: the unit starts over there, and a short unit
// or one with the odd deep scan (ordinary ava-ont batches have a handful) would only become a tail of its own behind the launch
// true: the batch is dense all over (k_chain_dense1 takes the handed-over units), false: it has a tail (k_chain_dense does).
// route: 0 decides by the number of long units in the batch, 1 / 2 / 3 force k_chain_dense / k_chain_dense1 / k_chain_dense16 (tests)
__device__ __forceinline__ bool dense_all(const unsigned int *long_units, int route)
{
	return route == 2 || (route == 0 && long_units && *long_units > CHAINDP_DENSE_MAX_LONG);
}

// true: the handed-over units go to the sixteen-waves-per-unit build of k_chain_dense (rounds of 1024 predecessors): the batch's
// tail is at most one workgroup per CU of them, and 32-bit differences stay exact over a ring of 1024 anchors.  route 3 forces it.
// (Measured, tools/dense_probe.py: 200 units of 15-38 k anchors 107 -> 101 ms; 500 units 127 -> 179 ms: rounds are fewer, 1.3
// instead of 2 per anchor, but each costs sixteen waves' barriers, so it only pays while the chip is not full.)
#define CHAINDP_DENSE16_MAX_UNITS 256u
__device__ __forceinline__ bool dense_wide(const unsigned long long *count, const unsigned int *long_units, int route, int max_dist_x)
{
	if (route) return route == 3;
	return !dense_all(long_units, route) && (uint32_t)*count <= CHAINDP_DENSE16_MAX_UNITS &&
	       ((uint64_t)(int64_t)max_dist_x + 1) * 1025ull < (1ull << 32);
}

#ifndef CHAINDP_DEEP_HANDOVER
#define CHAINDP_DEEP_HANDOVER 64
#endif
#define CHAINDP_DEEP_HANDOVER_LEFT 2048

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
#define LDS_PTR(T, a) ((__attribute__((address_space(3))) T*)(a))
#else
#define LDS_PTR(T, a) ((T*)(uintptr_t)(a))          /* host pass of the single-source compile; never executed */
#endif
__device__ __forceinline__ uint4 lds_load_b128(uint32_t a) { const u32x4_t t = *LDS_PTR(const u32x4_t, a); return make_uint4(t.x, t.y, t.z, t.w); }
__device__ __forceinline__ int2 lds_load_b64(uint32_t a) { const i32x2_t t = *LDS_PTR(const i32x2_t, a); return make_int2(t.x, t.y); }
__device__ __forceinline__ int lds_load_b32(uint32_t a) { return *LDS_PTR(const int, a); }
__device__ __forceinline__ int lds_load_i16(uint32_t a) { return (int)*LDS_PTR(const short, a); }
__device__ __forceinline__ void lds_store_b32(uint32_t a, int v) { *LDS_PTR(int, a) = v; }
__device__ __forceinline__ void lds_store_b128(uint32_t a, uint4 v) { u32x4_t t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *LDS_PTR(u32x4_t, a) = t; }

// inclusive prefix max of max(v, 0)-floored values (see above), then the value of lane-1 (0 for lane 0)
__device__ __forceinline__ int wave_excl_max_floor0(int v)
{
	v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, true));
	v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(2), 0xf, 0xf, true));
	v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(4), 0xf, 0xf, true));
	v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(8), 0xf, 0xf, true));
	v = max(v, dpp_or_old<DPP_ROW_BCAST15, 0xa>(INT_MIN, v));
	v = max(v, dpp_or_old<DPP_ROW_BCAST31, 0xc>(INT_MIN, v));
	return __builtin_amdgcn_update_dpp(0, v, DPP_WAVE_SHR1, 0xf, 0xf, true);
}

template <int RING>
struct FastLds {
	static constexpr uint32_t RB = 16u * RING;            // ring entries
	static constexpr uint32_t T_OFF = 28u * RING;         // mark tags, indexed by distance: word d-1 belongs to anchor i-d (the general
	                                                      // variant's x.hi/y.hi space); word RING is the dummy word
	static constexpr uint32_t V_OFF = RB + 4u * RING;     // v
	static constexpr uint32_t DUMMY = 32u * RING;         // sink for lanes without a mark to write
	static constexpr uint32_t LUT = 32u * RING + 16u;     // table of 1 - cost (int16)
};

struct FastK {
	uint32_t L4;       // lane * 16
	uint32_t far4;     // 4 * RING: offset of the dummy word behind the mark array (kept in a VGPR for v_cndmask)
	uint32_t trel;     // 4 * lane: this lane's own mark word in chunk 0
	uint32_t M;        // max_dist_x
	uint32_t cbw;      // max(max_dist_x - 1 - bw, 0)
	uint32_t dq_off;   // max_dist_x - min(max_dist_x, max_dist_y)
	uint32_t bw;
	int max_skip;
	int ms0;           // max(max_skip, 0): with n_skip starting at 0 the break needs more than this many B lanes
};

struct FastPairs { uint4 e; uint32_t drm1, dd; bool ok; };

// filters of chain.c:252-260 for lane k <-> slot address (S - 16k) mod ring bytes
template <int RING, bool SAMEGAP>
__device__ __forceinline__ FastPairs fast_filters(const FastK &k, uint32_t addr, uint32_t xm1, uint32_t qm1)
{
	FastPairs P;
	P.e = lds_load_b128(addr);
	P.drm1 = xm1 - P.e.x;
	const uint32_t dqm1 = qm1 - P.e.y;
	P.dd = absdiff_u32(P.drm1, dqm1);
	const uint32_t dqs = SAMEGAP ? dqm1 : __builtin_elementwise_add_sat(dqm1, k.dq_off);
	const uint32_t m2 = P.drm1 > dqs ? P.drm1 : dqs, t = P.dd + k.cbw;
	P.ok = (m2 > t ? m2 : t) < k.M;
	P.e.y = dqm1;
	return P;
}

struct FastMasks { uint64_t A, B; int sc; uint32_t drm1; };

// n_skip walk when A and B lanes interleave (chain.c:276,278): A lanes x -> max(x-1,0), B lanes x -> x+1, break
// when x > max_skip; done with a prefix min over the unclamped walk.  Returns true when the break is taken.
__device__ __forceinline__ bool fast_walk_general(const FastK &k, const FastMasks &m, int jtop, int &max_f, int &max_j, int &n_skip)
{
	const bool isA = __builtin_amdgcn_inverse_ballot_w64(m.A), isB = __builtin_amdgcn_inverse_ballot_w64(m.B);
	const int Sk = n_skip + lanes_below(m.B) + (int)isB - lanes_below(m.A) - (int)isA;
	const int Mk = wave_scan_min(Sk);
	const int x = Sk - (Mk < 0 ? Mk : 0);
	const uint64_t brk = m.B & __builtin_amdgcn_ballot_w64(x > k.max_skip);
	const uint64_t Ap = brk ? (m.A & ((1ull << __builtin_ctzll(brk)) - 1)) : m.A;      // A lanes before the break
	if (Ap) {
		const int ka = 63 - __builtin_clzll(Ap);
		max_f = __builtin_amdgcn_readlane(m.sc, ka);
		max_j = jtop - ka;
	}
	n_skip = __builtin_amdgcn_readlane(x, 63);
	return brk != 0;
}

// n_skip walk over one evaluated chunk (any n_skip on entry); returns true when the break is taken
__device__ __forceinline__ bool fast_walk(const FastK &k, const FastMasks &m, int jtop, int &max_f, int &max_j, int &n_skip)
{
	const int hiA = highest_lane(m.A);
	if ((m.B & low_mask64(hiA)) == 0) {                            // every A lane precedes every B lane (or one set is empty)
		if (hiA >= 0) {
			max_f = __builtin_amdgcn_readlane(m.sc, hiA);
			max_j = jtop - hiA;
		}
		int x = n_skip - __builtin_popcountll(m.A);
		x = x < 0 ? 0 : x;
		const int cb = __builtin_popcountll(m.B);
		n_skip = x + cb;
		return cb > 0 && n_skip > k.max_skip;                      // break taken at a B lane (chain.c:278-279); n_skip is dead then
	}
	return fast_walk_general(k, m, jtop, max_f, max_j, n_skip);
}

// Tile flush of a table-driven unit: the tile's anchors tile0 .. tile0+cnt-1 (one per lane) have their f and 4*p in the ring
// (waddr = the lane's ring entry); v[] (chain.c:284) by pointer doubling over the tile, then f/p/v and the compaction helpers.
template <int RING>
__device__ __forceinline__ void fast_flush_tile(const UnitCtx &c, int tile0, int cnt, uint32_t waddr, int64_t gi)
{
	typedef FastLds<RING> L;
	constexpr int MASK = RING - 1;
	const int lane = c.lane;
	int fi = 0, pi = -1, val = 0, ptr = -1;
	if (lane < cnt) {
		const int2 zw = lds_load_b64(waddr + 8u);
		fi = zw.x; pi = zw.y >> 2;
		val = fi; ptr = pi;
	}
	const bool ext = ptr >= 0 && ptr < tile0;               // predecessor in an earlier tile: its v is final
	const bool ext_far = ext && tile0 - ptr > RING;          // ... and no longer in the LDS copy
	if (__builtin_amdgcn_ballot_w64(ext_far)) wave_global_fence();
	if (ext) {
		const int vext = ext_far ? c.v[c.base + ptr] : lds_load_b32(L::V_OFF + ((uint32_t)(ptr & MASK) << 2));
		val = vext > val ? vext : val;
		ptr = -1;
	}
	for (int r = 0; r < 6; ++r) {
		const int src = (ptr >= tile0 ? ptr - tile0 : lane) << 2;
		const int pv = __builtin_amdgcn_ds_bpermute(src, val);
		const int pp = __builtin_amdgcn_ds_bpermute(src, ptr);
		if (ptr >= tile0) { val = pv > val ? pv : val; ptr = pp; }
	}
	wave_mem_fence();
	if (lane < cnt) lds_store_b32(L::V_OFF + (waddr >> 2), val);
	wave_mem_fence();
	// first_child[] of an anchor that is not emitted at its own step starts at "none" here, before any child (this
	// tile or a later one, always this wave) lowers it: no batch-wide initialisation pass
	const bool self = val >= c.min_sc || pi >= 0;
	if (lane < cnt && !self) c.first_child[gi] = NO_CHILD;
	if (__builtin_amdgcn_ballot_w64(lane < cnt && !self)) wave_global_fence();
	if (lane < cnt) {
		c.f[gi] = fi;
		c.p[gi] = pi < 0 ? -1 : pi + c.rel0;
		c.v[gi] = val;
		// Compaction (chain.c:286-317) needs, for every anchor that is not emitted at its own step, its
		// first child; while f/p/v of the tile are at hand, record "emitted at own step" and feed that min.
		int maybe_first = 0;
		if (pi >= 0) {
			int vq, pq;
			if (tile0 + cnt - 1 - pi < RING) {
				vq = lds_load_b32(L::V_OFF + ((uint32_t)(pi & MASK) << 2));
				pq = lds_load_b32(((uint32_t)(pi & MASK) << 4) + 12u);
			} else { vq = c.v[c.base + pi]; pq = c.p[c.base + pi]; }
			if (!(vq >= c.min_sc || pq >= 0)) { atomicMin(&c.first_child[c.base + pi], c.rel0 + tile0 + lane); maybe_first = 4; }
		}
		c.flags[gi] = (uint8_t)((self ? 2 : 0) | maybe_first | (val >= c.min_sc ? 8 : 0) | (fi < val ? 16 : 0));   // bits 3,4: the record's flag bits (chain.c:313-314)
	}
}

} // namespace chaindp
#endif
