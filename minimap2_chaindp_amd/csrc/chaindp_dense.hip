// chaindp_dense.hip -- the chain DP kernel for units whose scans run deep (dense repeats): k_chain_dense, EIGHT waves per unit.
//
// In a dense repeat the window of chain.c:252 holds hundreds to thousands of predecessors and few of them get marked, so
// the scan of chain.c:253-282 does not end at the (max_skip + 1)-th marked predecessor after ~30 steps but walks on: a
// median of 270 predecessors, one scan in nine beyond 1000, one in a hundred across the whole unit (tens of thousands) on
// the generator's `dense` shape.  A unit is a serial chain of anchors, so one such unit on one wave -- ~1200 instructions
// per anchor, issued by a wave that has its SIMD to itself at ~10 cycles each -- is the batch's tail (a 38 k-anchor unit:
// 0.3 s, sixty times the whole 76 M-anchor benchmark batch).  k_chain_units hands a unit that keeps scanning past its
// LDS ring over to this kernel (CHAINDP_DEEP_HANDOVER), which computes exactly the same thing (run_unit_fast of
// chaindp_kernels.hip, reference chain.c:246-284 for one-segment, non-cDNA reads) in a different shape:
//   * a workgroup of eight waves per unit.  A scan proceeds in ROUNDS of eight 64-lane chunks (512 predecessors; the
//     first round is the LDS ring, the following ones come back from HBM/L2), a chunk per wave.  What a chunk contributes
//     before the serial semantics -- filters, scores, its own prefix max, marks -- does not depend on the chunks in front
//     (DESIGN.md section 4): the waves do that side by side and exchange the chunks' maxima (barrier).  With the maxima in
//     front of it a wave knows its chunk's new running maxima (chain.c:274) and marked non-improving lanes (chain.c:277),
//     hence its n_skip walk (chain.c:276-279) as a FUNCTION of the n_skip it starts from: x -> max(x + a, b), break iff
//     x >= thr -- two popcounts when every new maximum precedes every marked lane, a prefix min otherwise.  The eight
//     functions go to LDS (barrier), and every wave composes them with a three-step lane-parallel prefix ((a1, b1) then
//     (a2, b2) = (a1 + a2, max(b1 + a2, b2))), finds the chunk the scan ends in and the running maximum there.  Chunks
//     behind the break are evaluated for nothing; the marks they write are never read;
//   * marks by distance as ONE BIT each: bit d of an LDS bitmap = "the predecessor d + 1 behind the current anchor is
//     marked" (chain.c:281).  64 K distances are 8 KB, so marks never leave LDS whatever the depth
//     of the scan: a mark is one ds_or_b32, a chunk's "marked" lane mask is one 64-bit word, and the workgroup wipes the
//     bitmap after each anchor up to the anchor's index.  (Units of more than 64 K anchors stay with k_chain_units.)
// The LDS footprint is per unit (29 KB), so the eight waves also buy the occupancy one wave per unit cannot have (four units,
// 32 waves per CU).  Measured (tools/dense_probe.py, MI355X): 200 dense units of 15-38 k anchors 310 -> 108 ms, 2000 units
// 526 -> 277 ms; a batch of thousands of long units is bound by instruction count, not by its tail, and is not handed over.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include "chaindp_kernels.h"
#include "chaindp_wave.h"
#include "chaindp_fast.h"

// This file is compiled twice (csrc/Makefile): as it is -- k_chain_dense, eight waves per unit, rounds of 512 predecessors, the ring
// of CHAINDP_DENSE_RING anchors -- and with -DDN_VARIANT16 -- k_chain_dense16 in a namespace of its own, SIXTEEN waves per unit, rounds
// of 1024 predecessors and a ring as long: for a batch whose tail is at most a workgroup per CU of long units (256), where a unit's
// time is the number of rounds its scans take and most of the chip would otherwise idle (6 % on 200 units; slower from 500 on).  The device decides
// which of the two finds work (dense_wide(), chaindp_fast.h), as it does between them and k_chain_dense1.
namespace chaindp {
#ifdef DN_VARIANT16
namespace dense16 {
#define DN_RING 1024
#define DN_WAVES 16
#define DN_ROUND 16
#define DN_LAUNCH launch_chain_dense16
#define DN_WIDE 1
#else
#define DN_RING CHAINDP_DENSE_RING
#define DN_WAVES 8                      // waves per unit
#define DN_ROUND 8                      // chunks per round
#define DN_LAUNCH launch_chain_dense
#define DN_WIDE 0
#endif
#define DN_CPW (DN_ROUND / DN_WAVES)    // chunks a wave evaluates per round
static_assert(DN_CPW * DN_WAVES == DN_ROUND && DN_WAVES >= 2, "the waves must tile the round");
static_assert(64 * DN_ROUND == DN_RING, "the first round is the ring");
typedef FastLds<DN_RING> DnL;
// LDS (dynamic segment, raw byte offsets from 0): ring entries [0, 8 K) and v[] [10 K, 12 K) as FastLds<512> lays them out (the
// tile flush is shared with k_chain_units); the mark bitmap; a word per thread where a lane without a mark ORs its zero (LDS
// atomics of one instruction on ONE address take a turn each); what the waves tell each other about a round's chunks, twice (a
// round uses the half of its parity, so that a wave may start the next round while another still reads this one's): the
// chunk's scores per lane, its maximum, its summary; the read's table of 1 - cost (int16)
#define DN_BM (DnL::V_OFF + 4u * DN_RING)  // 12 288 for the ring of 512
#define DN_BM_BYTES (CHAINDP_DENSE_BITCAP / 8u)
#define DN_SINK (DN_BM + DN_BM_BYTES)
#define DN_SC (DN_SINK + 4u * 64u * DN_WAVES)
#define DN_SC_HALF (256u * DN_ROUND)
#define DN_M (DN_SC + 2u * DN_SC_HALF)
#define DN_M_HALF (4u * DN_ROUND)
#define DN_SUM (DN_M + 2u * DN_M_HALF)
#define DN_SUM_CHUNK 48u                // a, b, thr, flags | sc and lane of the chunk's last new maximum, pad | A, B lane masks
#define DN_SUM_HALF (DN_SUM_CHUNK * DN_ROUND)
#define DN_LUT ((DN_SUM + 2u * DN_SUM_HALF + 15u) & ~15u)
static_assert(DnL::V_OFF + 4u * DN_RING <= DN_BM, "v[] runs into the bitmap");
#define DN_NEG (-(1 << 28))             // "minus infinity" that survives a few additions

struct DenseArgs {
	Params par;
	const int64_t *off;
	const ulonglong2 *a;
	const uint16_t *lut;
	int lut_stride;
	const Unit *units;                  // the hand-over list
	const unsigned long long *count;    // its length (low 32 bits)
	int32_t *f, *p, *v;
	int32_t *first_child;
	uint8_t *flags;
	const unsigned int *long_units;     // see dense_all(), chaindp_fast.h
	unsigned int *queue;                // the next unit nobody has taken (zero when the step starts; the two builds of this kernel never both run)
	int route;
	unsigned long long *stamp;          // diagnostic build (-DCHAINDP_DENSE_STAMPS, run with CHAINDP_DENSE_STAMP=1): where wave 0's time goes
};

// in-kernel stamps: compiled in only with -DCHAINDP_DENSE_STAMPS (make -C csrc stamps)
#ifdef CHAINDP_DENSE_STAMPS
#define DN_STAMP(...) __VA_ARGS__
#else
#define DN_STAMP(...)
#endif
#define DN_NOW() __builtin_amdgcn_s_memtime()

__device__ __forceinline__ void dn_or_b32(uint32_t a, uint32_t v)
{
	(void)__hip_atomic_fetch_or(LDS_PTR(uint32_t, a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// chain.c:281 for one lane: the predecessor's predecessor p (4 * its unit-relative index, negative: none) is marked: bit
// i - 1 - p of the bitmap (i1x4 = 4 (i - 1)).  A lane without a mark ORs a zero into its own sink word: nothing is masked off.
__device__ __forceinline__ void dn_mark(uint32_t bm, uint32_t sink, uint32_t i1x4, bool ok, uint32_t p4)
{
	const uint32_t d4 = i1x4 - p4;
	const bool has = ok && (int)p4 >= 0;
	dn_or_b32(has ? bm + ((d4 >> 7) << 2) : sink, has ? 1u << ((d4 >> 2) & 31u) : 0u);
}

// the 64 mark bits of the chunk that starts kb predecessors back, as a lane mask (every lane reads the same word)
__device__ __forceinline__ uint64_t dn_marked(uint32_t bm, uint32_t kb)
{
	const int2 w = lds_load_b64(bm + (kb >> 3));
	return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(w.y) << 32 | (uint32_t)__builtin_amdgcn_readfirstlane(w.x);
}

// what a chunk contributes before the serial semantics are applied (lane <-> predecessor j = jtop - lane)
struct DenseChunk {
	int sc;                             // score incl. f[j] (chain.c:262-273); INT_MIN where the filters of chain.c:252-261 fail
	uint64_t okm;                       // lanes that pass them
	bool ends;                          // nothing older than this chunk can matter (window or unit exhausted)
};

// The ring chunk that starts kb predecessors back (first round): filters, score, mark.
template <bool SAMEGAP>
__device__ __forceinline__ DenseChunk dn_ring_chunk(const FastK &k, uint32_t bm, uint32_t sink, uint32_t xm1, uint32_t qm1, int spm1, int i, int kb)
{
	const uint32_t S = (uint32_t)(i - 1 - kb) << 4;
	const FastPairs P = fast_filters<DN_RING, SAMEGAP>(k, (S - k.L4) & (DnL::RB - 1u), xm1, qm1);
	const int dqm1 = (int)P.e.y, drm1 = (int)P.drm1;
	int sc0 = dqm1 < drm1 ? dqm1 : drm1;
	sc0 = sc0 < spm1 ? sc0 : spm1;                                                      // chain.c:262-263, minus one
	const uint32_t di = P.dd < k.bw ? P.dd : k.bw;
	const int scu = sc0 + (int)P.e.z + lds_load_i16(DN_LUT + 2u * di);                   // chain.c:272-273 via the table
	DenseChunk r;
	r.okm = __builtin_amdgcn_ballot_w64(P.ok);
	r.sc = __builtin_amdgcn_inverse_ballot_w64(r.okm) ? scu : INT_MIN;
	dn_mark(bm, sink, (uint32_t)(i - 1) << 2, P.ok, P.e.w);
	// the scan ends behind this chunk if its last lane is outside the window (x is sorted) or the unit starts here
	r.ends = (uint32_t)__builtin_amdgcn_readlane((int)P.drm1, 63) + 1u > k.M || kb + 64 >= i;
	return r;
}

// The deep chunk that starts kb predecessors back (a later round: older than the ring; a, f, p come back from HBM/L2).  Only the
// window test is done in 64 bits (x_i - x_j of a predecessor this old may exceed 32 bits; for a lane inside the window it does
// not, and every other difference is bounded by the window).
template <bool SAMEGAP>
__device__ __forceinline__ DenseChunk dn_deep_chunk(const UnitCtx &c, const FastK &k, uint32_t bm, uint32_t sink, uint64_t xi, uint32_t qi, int spm1, int i, int kb)
{
	const int j = i - 1 - kb - c.lane;
	const bool inr = j >= 0;
	const int64_t gj = c.base + (inr ? j : 0);
	const ulonglong2 aj = c.a[gj];
	const int fj = c.f[gj];
	const int pjr = c.p[gj];                                                            // stored read-relative
	const bool live = inr && xi - aj.x <= c.maxx;                                       // chain.c:252
	const uint32_t drm1 = (uint32_t)xi - (uint32_t)aj.x - 1u, dqm1 = qi - (uint32_t)aj.y - 1u;
	const uint32_t dd = absdiff_u32(drm1, dqm1);
	const uint32_t dqs = SAMEGAP ? dqm1 : __builtin_elementwise_add_sat(dqm1, k.dq_off);
	const uint32_t m2 = drm1 > dqs ? drm1 : dqs, t = dd + k.cbw;
	const bool ok = live && (m2 > t ? m2 : t) < k.M;                                    // chain.c:257-260 (one compare, as fast_filters)
	int sc0 = (int)dqm1 < (int)drm1 ? (int)dqm1 : (int)drm1;
	sc0 = sc0 < spm1 ? sc0 : spm1;
	const uint32_t di = dd < k.bw ? dd : k.bw;
	const int scu = sc0 + fj + lds_load_i16(DN_LUT + 2u * di);
	DenseChunk r;
	r.okm = __builtin_amdgcn_ballot_w64(ok);
	r.sc = __builtin_amdgcn_inverse_ballot_w64(r.okm) ? scu : INT_MIN;
	dn_mark(bm, sink, (uint32_t)(i - 1) << 2, ok, pjr >= 0 ? (uint32_t)(pjr - c.rel0) << 2 : 0xfffffffcu);
	r.ends = __builtin_amdgcn_ballot_w64(live) != ~0ull;            // a lane outside the window (or the unit): nothing older can matter
	return r;
}

// inclusive prefix over the first eight lanes (one per chunk of the round) with row_shr steps
__device__ __forceinline__ int dn_scan8_max(int v)
{
	v = max(v, dpp_or_old<DPP_ROW_SHR(1), 0xf>(INT_MIN, v));
	v = max(v, dpp_or_old<DPP_ROW_SHR(2), 0xf>(INT_MIN, v));
	v = max(v, dpp_or_old<DPP_ROW_SHR(4), 0xf>(INT_MIN, v));
	if (DN_ROUND > 8) v = max(v, dpp_or_old<DPP_ROW_SHR(8), 0xf>(INT_MIN, v));   // (sixteen chunks: still one row of lanes)
	return v;
}

// The n_skip walk of chain.c:276-279 over one chunk, as a function of the n_skip it starts from: lanes of A (new running
// maximum) take x -> max(x - 1, 0), lanes of B (marked, not better) x -> x + 1 and break when x > max_skip.  With S_k = #B - #A
// over the lanes up to k, the walk from x is S_k + max(x, -min_{m<=k} S_m): the chunk maps x to max(x + a, b), and it breaks
// iff x >= thr.  When every A lane precedes every B lane -- nearly always -- that is two popcounts.
struct DenseWalk { int a, b, thr; bool inter; };

__device__ __forceinline__ DenseWalk dn_chunk_walk(const FastK &k, uint64_t A, uint64_t B, int hiA)
{
	DenseWalk r;
	const int cA = __builtin_popcountll(A), cB = __builtin_popcountll(B);
	r.inter = (B & low_mask64(hiA)) != 0;                           // a B lane below an A lane (hiA = -64 for no A lane: empty mask)
	if (__builtin_expect(!r.inter, 1)) {
		r.a = cB - cA; r.b = cB;
		r.thr = cB == 0 ? INT_MAX : (cB > k.max_skip ? 0 : k.max_skip + 1 - cB + cA);
		return r;
	}
	const bool isA = __builtin_amdgcn_inverse_ballot_w64(A), isB = __builtin_amdgcn_inverse_ballot_w64(B);
	const int Sk = lanes_below(B) + (int)isB - lanes_below(A) - (int)isA;
	int mn = wave_scan_min(Sk);
	mn = mn < 0 ? mn : 0;
	const int need = !isB ? INT_MAX : (Sk - mn > k.max_skip ? 0 : k.max_skip + 1 - Sk);
	const int s63 = __builtin_amdgcn_readlane(Sk, 63);
	r.a = s63; r.b = s63 - __builtin_amdgcn_readlane(mn, 63);
	r.thr = __builtin_amdgcn_readlane(wave_scan_min(need), 63);
	return r;
}

template <bool SAMEGAP>
__device__ __forceinline__ void run_unit_dense(const UnitCtx &c, int64_t room, int w, int tid, unsigned long long *stamp_out)
{
	constexpr int MASK = DN_RING - 1;
	const int lane = c.lane;
	const uint32_t lane4 = (uint32_t)lane << 2;
	const uint32_t sink = DN_SINK + ((uint32_t)tid << 2);
	FastK k;
	k.L4 = (uint32_t)lane << 4;
	k.far4 = 0; k.trel = 0;
	k.M = (uint32_t)c.maxx;
	k.bw = (uint32_t)c.bw;
	k.cbw = k.M - 1u > k.bw ? k.M - 1u - k.bw : 0u;
	k.dq_off = k.M - (uint32_t)c.mdq;
	k.max_skip = c.max_skip;
	k.ms0 = c.max_skip > 0 ? c.max_skip : 0;
	DN_STAMP(unsigned long long st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long t_unit0 = DN_NOW();)
	uint64_t x_carry = 0;
	uint32_t rb = 0;                                                    // round parity: which half of the exchange area (consecutive rounds
	                                                                    // alternate, also across anchors: a wave may run one round ahead)
	for (int tile0 = 0;; tile0 += 64) {
		// every wave holds the tile's anchors (the same 1 KB, eight times from L2: nothing next to a scan)
		const int64_t gi = c.base + tile0 + lane;
		const bool have = tile0 + lane < room;
		ulonglong2 an = make_ulonglong2(0, 0);
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
		const uint32_t waddr = (uint32_t)((tile0 + lane) & MASK) << 4;
		for (int ii = 0; ii < cnt; ++ii) {
			const int i = tile0 + ii;                                   // unit-relative index of the anchor being scored
			const uint64_t xi = readlane_u64(an.x, ii);
			const uint32_t qi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)an.y, ii);
			const int span = span_of_hi((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(an.y >> 32), ii));
			// the scan's state between rounds (chain.c:251); every wave keeps the same copy
			int max_f = span, max_j = -1, n_skip = 0;
			const uint32_t bm = DN_BM;
			for (int kbr = 0;; kbr += 64 * DN_ROUND) {
				DN_STAMP(const unsigned long long ta0 = DN_NOW();)
				// ---- this wave's chunks of the round: the ring (the unit's first anchors find unwritten slots there: they fail the
				// window test), then deep rounds
				DenseChunk ch[DN_CPW];
				int excl[DN_CPW];
#pragma unroll
				for (int g = 0; g < DN_CPW; ++g) {
					const int kb = kbr + 64 * (DN_CPW * w + g);
					ch[g] = kbr == 0 ? dn_ring_chunk<SAMEGAP>(k, bm, sink, (uint32_t)xi, qi, span - 1, i, kb)
					                 : dn_deep_chunk<SAMEGAP>(c, k, bm, sink, xi, qi, span - 1, i, kb);
				}
#pragma unroll
				for (int g = 0; g < DN_CPW; ++g) {
					excl[g] = wave_excl_max_floor0(ch[g].sc);
					const int m63 = max(__builtin_amdgcn_readlane(excl[g], 63), __builtin_amdgcn_readlane(ch[g].sc, 63));
					lds_store_b32(DN_SC + rb * DN_SC_HALF + 256u * (uint32_t)(DN_CPW * w + g) + lane4, ch[g].sc);
					if (lane == 0) lds_store_b32(DN_M + rb * DN_M_HALF + 4u * (uint32_t)(DN_CPW * w + g), m63);
				}
				DN_STAMP(const unsigned long long tb0 = DN_NOW(); st[0] += tb0 - ta0;)
				__syncthreads();
				DN_STAMP(const unsigned long long tb1 = DN_NOW(); st[1] += tb1 - tb0;)
				// ---- with the maxima of the chunks in front: new running maxima (chain.c:274), marked and not better (chain.c:277),
				// and the chunk's n_skip walk as a function of the n_skip it starts from
				{
					const int mv = dn_scan8_max(lane < DN_ROUND ? lds_load_b32(DN_M + rb * DN_M_HALF + lane4) : INT_MIN);
#pragma unroll
					for (int g = 0; g < DN_CPW; ++g) {
						const int cs = DN_CPW * w + g;                                  // the chunk's slot in the round
						const int before = cs ? __builtin_amdgcn_readlane(mv, cs - 1) : INT_MIN;
						const int e = max(excl[g], max(before, max_f));
						const uint64_t A = __builtin_amdgcn_ballot_w64(ch[g].sc > e);   // masked lanes hold INT_MIN
						const uint64_t B = ch[g].okm & ~A & dn_marked(bm, (uint32_t)(kbr + 64 * cs));
						const int hiA = highest_lane(A);
						const DenseWalk wk = dn_chunk_walk(k, A, B, hiA);
						const int scA = A ? __builtin_amdgcn_readlane(ch[g].sc, hiA) : 0;
						if (lane == 0) {
							const uint32_t sa = DN_SUM + rb * DN_SUM_HALF + DN_SUM_CHUNK * (uint32_t)cs;
							lds_store_b128(sa, make_uint4((uint32_t)wk.a, (uint32_t)wk.b, (uint32_t)wk.thr, (A ? 1u : 0u) | (ch[g].ends ? 2u : 0u) | (wk.inter ? 4u : 0u)));
							lds_store_b128(sa + 16u, make_uint4((uint32_t)scA, (uint32_t)hiA, 0u, 0u));
							lds_store_b128(sa + 32u, make_uint4((uint32_t)A, (uint32_t)(A >> 32), (uint32_t)B, (uint32_t)(B >> 32)));
						}
					}
				}
				DN_STAMP(const unsigned long long tb2 = DN_NOW(); st[2] += tb2 - tb1;)
				__syncthreads();
				DN_STAMP(st[3] += DN_NOW() - tb2;)
				// ---- the round's eight walks composed, a lane per chunk (every wave does this for itself -- same numbers, no broadcast:
				// a third barrier per round costs more than seven redundant copies of these ~70 instructions)
				DN_STAMP(const unsigned long long tc0 = DN_NOW();)
				bool done;
				{
					uint4 s0 = make_uint4(0u, (uint32_t)DN_NEG, (uint32_t)INT_MAX, 0u);
					int2 s1 = make_int2(0, 0);
					if (lane < DN_ROUND) {
						const uint32_t sa = DN_SUM + rb * DN_SUM_HALF + DN_SUM_CHUNK * (uint32_t)lane;
						s0 = lds_load_b128(sa); s1 = lds_load_b64(sa + 16u);
					}
					// inclusive composition over the lanes: (a1, b1) then (a2, b2) is (a1 + a2, max(b1 + a2, b2))
					int ca = (int)s0.x, cb = (int)s0.y;
#pragma unroll
					for (int sh = 1; sh < DN_ROUND; sh <<= 1) {
						int la, lb;
						if (sh == 1) { la = dpp_or_old<DPP_ROW_SHR(1), 0xf>(0, ca); lb = dpp_or_old<DPP_ROW_SHR(1), 0xf>(DN_NEG, cb); }
						else if (sh == 2) { la = dpp_or_old<DPP_ROW_SHR(2), 0xf>(0, ca); lb = dpp_or_old<DPP_ROW_SHR(2), 0xf>(DN_NEG, cb); }
						else if (sh == 4) { la = dpp_or_old<DPP_ROW_SHR(4), 0xf>(0, ca); lb = dpp_or_old<DPP_ROW_SHR(4), 0xf>(DN_NEG, cb); }
						else { la = dpp_or_old<DPP_ROW_SHR(8), 0xf>(0, ca); lb = dpp_or_old<DPP_ROW_SHR(8), 0xf>(DN_NEG, cb); }
						cb = max(lb + ca, cb); ca = la + ca;
					}
					const int x_out = max(n_skip + ca, cb);                             // n_skip behind chunk `lane`
					const int x_in = dpp_or_old<DPP_ROW_SHR(1), 0xf>(n_skip, x_out);    // ... and in front of it
					const bool brk = lane < DN_ROUND && x_in >= (int)s0.z;
					const uint64_t brkm = __builtin_amdgcn_ballot_w64(brk);
					const uint64_t stopm = brkm | __builtin_amdgcn_ballot_w64(lane < DN_ROUND && (s0.w & 2u));
					done = stopm != 0;
					const int cstar = done ? __builtin_ctzll(stopm) : DN_ROUND - 1;      // the chunk the scan ends in (or the round's last)
					const uint64_t hasm = __builtin_amdgcn_ballot_w64(lane < DN_ROUND && (s0.w & 1u));
					const bool exact = done && (brkm >> cstar & 1ull) && (__builtin_amdgcn_readlane((int)s0.w, cstar) & 4);
					// the running maximum: the last chunk up to cstar with a new maximum.  If the break falls into a chunk whose A and B
					// lanes interleave, only the A lanes in front of the break count: that chunk is walked lane by lane below.
					const uint64_t cand = hasm & low_mask64(__builtin_amdgcn_readfirstlane(exact ? cstar : cstar + 1));
					if (cand) {
						const int cc = 63 - __builtin_clzll(cand);
						max_f = __builtin_amdgcn_readlane(s1.x, cc);
						max_j = i - 1 - kbr - 64 * cc - __builtin_amdgcn_readlane(s1.y, cc);
					}
					if (__builtin_expect(exact, 0)) {
						FastMasks m;
						const uint32_t sa = DN_SUM + rb * DN_SUM_HALF + DN_SUM_CHUNK * (uint32_t)cstar + 32u;
						const uint4 ab = lds_load_b128(sa);
						m.A = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)ab.y) << 32 | (uint32_t)__builtin_amdgcn_readfirstlane((int)ab.x);
						m.B = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)ab.w) << 32 | (uint32_t)__builtin_amdgcn_readfirstlane((int)ab.z);
						m.sc = lds_load_b32(DN_SC + rb * DN_SC_HALF + 256u * (uint32_t)cstar + lane4);
						m.drm1 = 0;
						int ns = __builtin_amdgcn_readlane(x_in, cstar);
						(void)fast_walk_general(k, m, i - 1 - kbr - 64 * cstar, max_f, max_j, ns);
					}
					n_skip = __builtin_amdgcn_readlane(x_out, DN_ROUND - 1);
				}
				DN_STAMP(st[4] += DN_NOW() - tc0; ++st[8]; if (kbr) ++st[9];)
				rb ^= 1u;
				if (done) break;
			}
			// anchor i enters the ring (chain.c:283); its marks are wiped: no distance beyond i - 1 can have been set.  (Measured
			// against this: a second bitmap wiped during the next anchor instead of this barrier, same time for one unit per CU and
			// 18 % more with several; one walking wave that tells the others, i.e. a third barrier per round, 10-20 % more.)
			if (w == 0 && lane == ii) lds_store_b128(waddr, make_uint4((uint32_t)an.x + 1u, (uint32_t)an.y + 1u, (uint32_t)max_f, (uint32_t)(max_j << 2)));
			{
				const uint32_t n_b = (uint32_t)(i + 31) >> 5 << 2;
				for (uint32_t q = (uint32_t)tid << 4; q < n_b; q += 1024u * DN_WAVES) lds_store_b128(DN_BM + q, make_uint4(0u, 0u, 0u, 0u));
			}
			DN_STAMP(const unsigned long long te0 = DN_NOW();)
			__syncthreads();
			DN_STAMP(st[5] += DN_NOW() - te0; ++st[7];)
		}
		// wave 0 flushes the tile (v of chain.c:284, f/p/v, the compaction helpers) while the others start on the next one
		DN_STAMP(const unsigned long long tf0 = DN_NOW();)
		if (w == 0) fast_flush_tile<DN_RING>(c, tile0, cnt, waddr, gi);
		DN_STAMP(st[6] += DN_NOW() - tf0;)
		if (cnt < 64) break;
	}
	DN_STAMP(if (stamp_out && w == 0 && lane == 0) { for (int q = 0; q < 10; ++q) atomicAdd(stamp_out + q, st[q]); atomicAdd(stamp_out + 10, DN_NOW() - t_unit0); })
}

template <bool SAMEGAP>
__global__ __launch_bounds__(64 * DN_WAVES) void k_chain_dense(DenseArgs g)
{
	if (dense_all(g.long_units, g.route)) return;                   // a batch that is dense all over: k_chain_dense1 has the units
	if (dense_wide(g.count, g.long_units, g.route, g.par.max_dist_x) != (DN_WIDE != 0)) return;   // a short tail: sixteen waves per unit (the other build of this file)
	UnitCtx c;
	c.a = g.a; c.f = g.f; c.p = g.p; c.v = g.v; c.tg = nullptr; c.tg_hi = 0; c.first_child = g.first_child; c.flags = g.flags; c.min_sc = g.par.min_sc;
	c.s_w = nullptr; c.s_t = nullptr; c.s_v = nullptr; c.s_xhi = nullptr; c.s_yhi = nullptr; c.s_lut = nullptr; c.s_dummy = nullptr;
	c.deep_list = nullptr; c.deep_cnt = nullptr; c.deep_cap = 0; c.deep_min = 0; c.deep_left = 0; c.deep_n = 0;
	c.lane = threadIdx.x & 63;
	c.maxx = (uint64_t)(int64_t)g.par.max_dist_x;
	c.mdx = g.par.max_dist_x; c.mdy = g.par.max_dist_y; c.bw = g.par.bw; c.max_skip = g.par.max_skip; c.is_cdna = 0;
	c.mdq = g.par.max_dist_x < g.par.max_dist_y ? g.par.max_dist_x : g.par.max_dist_y;
	c.avgd = 0; c.seg_rule = false;
	const int tid = threadIdx.x;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int64_t n_units = (int64_t)(uint32_t)g.count[0];
	// units are taken from a counter, not dealt by workgroup index: they differ 2.5x in length and arrive in no order, and a batch
	// with more units than the chip holds workgroups (2000 on 1024) otherwise ends when the unluckiest pair of them does
	for (;;) {
		__syncthreads();                                               // (the slot below, and the previous unit's LDS, are no longer in use)
		if (tid == 0) lds_store_b32(DN_M, (int)atomicAdd(g.queue, 1u));
		__syncthreads();
		const int64_t ub = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane(lds_load_b32(DN_M));
		if (ub >= n_units) break;
		const Unit u = g.units[ub];
		c.base = u.start; c.read = u.read;
		c.rel0 = (int)(u.start - g.off[u.read]);
		__syncthreads();                                               // the previous unit's LDS is no longer in use
		{
			const uint4 *src = (const uint4*)(g.lut + (int64_t)u.read * g.lut_stride);   // lut_stride is a multiple of 8 entries (16 B)
			for (int q = tid; q * 8 < g.lut_stride; q += 64 * DN_WAVES) lds_store_b128(DN_LUT + ((uint32_t)q << 4), src[q]);
			const uint32_t x_none = (uint32_t)g.a[u.start].x - (uint32_t)c.maxx - 1u;    // "no anchor here yet" (x+1 encoding): fails the window test
			for (int q = tid; q < DN_RING; q += 64 * DN_WAVES) lds_store_b128((uint32_t)q << 4, make_uint4(x_none, 0u, 0u, 0xfffffffcu));
			for (uint32_t o = (uint32_t)tid << 4; o < DN_BM_BYTES; o += 1024u * DN_WAVES) lds_store_b128(DN_BM + o, make_uint4(0u, 0u, 0u, 0u));
		}
		__syncthreads();
		run_unit_dense<SAMEGAP>(c, (int64_t)u.len, w, tid, g.stamp);
	}
}

static size_t dense_lds_bytes(int lut_stride) { return (size_t)DN_LUT + (size_t)lut_stride * 2; }

#ifdef DN_VARIANT16
}  // namespace dense16
using namespace dense16;
#endif

hipError_t DN_LAUNCH(hipStream_t st, const Params &par, int64_t max_units, const int64_t *d_off, const void *d_a,
                     const uint16_t *d_lut, int lut_stride, const Unit *d_deep, const unsigned long long *d_deep_cnt,
                     int32_t *d_f, int32_t *d_p, int32_t *d_v, int32_t *d_first_child, uint8_t *d_flags,
                     const unsigned int *d_long_units, int deep_route, unsigned int *d_queue)
{
	if (max_units <= 0 || !d_lut) return hipSuccess;
	if (!d_queue) return hipErrorInvalidValue;
	const size_t lds = dense_lds_bytes(lut_stride);
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	// the count is only known on the device: as many workgroups as the chip holds at this LDS size, each taking units in turn
	int64_t per_cu = (int64_t)(160 * 1024 / lds);
	if (per_cu > 8) per_cu = 8;
	if (per_cu < 1) per_cu = 1;
	int64_t blocks = (int64_t)cus * per_cu;
	if (DN_WIDE && blocks > (int64_t)CHAINDP_DENSE16_MAX_UNITS) blocks = CHAINDP_DENSE16_MAX_UNITS;   // (it only ever takes that many units: fewer workgroups to start and end for nothing)
	if (blocks > max_units) blocks = max_units;
	const void *fn = par.max_dist_y >= par.max_dist_x ? (const void*)k_chain_dense<true> : (const void*)k_chain_dense<false>;
	{
		const hipError_t e = check_no_static_lds(fn);        // LDS is addressed by raw byte offsets from 0
		if (e != hipSuccess) return e;
	}
	if (lds > 64 * 1024) {
		const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return e;
	}
	DenseArgs g;
	g.par = par; g.off = d_off; g.a = (const ulonglong2*)d_a; g.lut = d_lut; g.lut_stride = lut_stride;
	g.units = d_deep; g.count = d_deep_cnt; g.f = d_f; g.p = d_p; g.v = d_v; g.first_child = d_first_child; g.flags = d_flags;
	g.long_units = d_long_units; g.route = deep_route; g.queue = d_queue;
	g.stamp = nullptr;
#ifdef CHAINDP_DENSE_STAMPS
	static unsigned long long *d_stamp = nullptr;
	static const bool stamp = getenv("CHAINDP_DENSE_STAMP") != nullptr;
	if (stamp && !d_stamp && hipMalloc((void**)&d_stamp, 16 * 8) != hipSuccess) d_stamp = nullptr;
	if (stamp && d_stamp) { (void)hipMemsetAsync(d_stamp, 0, 16 * 8, st); g.stamp = d_stamp; }
#endif
	if (par.max_dist_y >= par.max_dist_x) hipLaunchKernelGGL(k_chain_dense<true>, dim3((unsigned)blocks), dim3(64 * DN_WAVES), lds, st, g);
	else hipLaunchKernelGGL(k_chain_dense<false>, dim3((unsigned)blocks), dim3(64 * DN_WAVES), lds, st, g);
#ifdef CHAINDP_DENSE_STAMPS
	if (g.stamp) {
		unsigned long long h[16];
		if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h, g.stamp, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[7]) {
			const double na = (double)h[7], nr = (double)h[8];
			fprintf(stderr, "[dense stamp] wave 0, shader-clock ticks per ANCHOR: total %.0f | evaluate %.0f, barrier %.0f, summarise %.0f, barrier %.0f, compose %.0f, "
			                "wipe + barrier %.0f, flush %.0f | rounds per anchor %.2f (deep %.2f)\n", (double)h[10] / na, h[0] / na, h[1] / na, h[2] / na, h[3] / na, h[4] / na,
			        h[5] / na, h[6] / na, nr / na, (double)h[9] / na);
		}
	}
#endif
	return hipGetLastError();
}

} // namespace chaindp
