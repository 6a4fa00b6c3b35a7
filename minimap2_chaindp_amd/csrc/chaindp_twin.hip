// chaindp_twin.hip -- the chain DP kernel for ordinary long-read units: TWO units per wave64, one per 32-lane half.
//
// Why (measured on MI355X, tools/issue_calib.hip -> profiles/r02_issue_calib.json): the one-unit-per-wave kernel
// (chaindp_kernels.hip) spends ~29 scalar and ~31 vector instructions per anchor.  The scalar unit is shared by the
// CU's four SIMDs and issues ONE instruction per cycle per CU, so 29 of them cost each SIMD ~125 of its 140 cycles per
// anchor; vector instructions that read an SGPR, have three register sources or use DPP issue at half rate (4.2
// cycles instead of 2.4).  And the scan of an ava-ont anchor ends (max_skip + 1 marked predecessors, chain.c:277-279)
// within 32 predecessors 99.99 % of the time (map-ont: 71 %, within 48: 99.9 %), so half of a 64-lane pass is wasted.
// Hence: chunks of 32 predecessors, two independent units side by side in one wave, and everything that is uniform
// per unit kept in VECTOR registers (replicated over the half's lanes) and updated by plain two-operand VALU
// instructions; the scalar unit only combines 64-bit lane masks (both halves at once) and branches.
//
// What it computes is exactly run_unit_fast of chaindp_kernels.hip (reference chain.c:246-284 for reads with one
// segment, not cDNA, bw <= 511, every q_span > 0), with the same four derivations on 32-lane chunks.  Units it
// does not take -- general-variant reads, and any unit whose scan needs a predecessor older than its LDS ring (64
// anchors) -- are appended to a leftover list and run by k_chain_units afterwards, from scratch.
//
// Round 3: a half's anchors enter and leave in tiles of 64, and the per-tile service (next tile in, finished tile's f/p/v out,
// unit switch) is done by ALL 64 lanes of the wave for one half at a time -- it was 32 anchors by the half's own 32 lanes with the
// other half idle, a third of every wave's time and 30 % of its vector instructions.  Scores are kept minus one inside the kernel
// (the PF ring holds f - 1, the current anchor's floor is q_span - 1; chain.c:251,274 compare the same way when both sides are
// shifted), so that a pass needs q_span - 1 only; reads with a zero q_span go to k_chain_units.
//
// LDS per wave (dynamic segment, starts at byte 0; h = half):
//   XY  [128 slots][2 halves] 8 B   x.lo+1, qpos+1 of anchor (slot = i & 127), written a whole tile at a time
//   PF  [ 64 slots][2 halves] 8 B   4*p (unit-relative, -4 = none), f - 1 of anchor (slot = i & 63)
//   V   [ 64 slots][2 halves] 4 B   v | "emitted at its own step" << 31
//   XQ  [2 halves][64] 8 B          the current tile's anchors as a pass wants them: x.lo, qpos
//   LUT [2 halves][512] int8        the read's table of 1 - cost (reads whose costs do not fit a byte go to k_chain_units)
//   ST  [2 halves] 56 B             the half's cold state (TwinCold) and what a scan carries into its second chunk
//   MK  [2 halves][65] 4 B          marks by distance: word d-1 holds the scan tag of the anchor d behind; word 64 = sink
//   SP  [2 halves][64] 1 B          q_span - 1 of the current tile's anchors (entry n of XQ at byte a has its SP byte at a / 8 + const)
//   KEY [2 halves] 4 B              which read's table the half's LUT holds (reads with the same avg_qspan share it)
// 6400 bytes.  LDS is handed out in pieces of 1280 bytes on this chip (tools/lds_occupancy_probe.hip measures how many workgroups
// a CU holds; the occupancy API does not know): 6400 bytes are the most that leave 24 waves per CU, i.e. six per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "chaindp_kernels.h"
#include "chaindp_wave.h"
#include "chaindp_lanes.h"

namespace chaindp {

#define TW_XY 0u
#define TW_PF 2048u
#define TW_V 3072u
#define TW_XQ 3584u
#define TW_XQ_HALF 512u
#define TW_LUT 4608u
#define TW_LUT_HALF 512u
#define TW_ST 5632u
#define TW_ST_HALF 56u
#define TW_MK 5744u
#define TW_MK_HALF 260u
#define TW_SP 6264u
#define TW_SP_HALF 64u
#define TW_SP_OF_XQ (TW_SP - TW_XQ / 8u)  // SP address = (XQ address >> 3) + this
#define TW_KEY 6392u                    // [2 halves] 4 B: key of the cost table the half's LUT holds (UnitAux::lutkey)
#define TW_LDS_BYTES 6400u
#define TW_TILE 64                      // anchors a half takes in / flushes at a time
#define TW_RING 64                      // predecessors a scan can reach in this kernel (two chunks of 32)
#define TW_QCH 8                        // units a half takes from the queue at a time

// exclusive prefix max over the 32 lanes of each half, floor 0 (scores are >= 0 where they matter): inclusive scan
// inside the 16-lane rows (4 DPP steps), one-lane shift inside the rows, and for the upper row of each half the lower
// row's total (row_bcast:15 into rows 1 and 3; harmless for the lanes that already hold a larger prefix)
__device__ __forceinline__ int tw_excl_max32(int v)
{
	v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, true));
	v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(2), 0xf, 0xf, true));
	v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(4), 0xf, 0xf, true));
	v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(8), 0xf, 0xf, true));
	int e = __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, true);
#if defined(__HIP_DEVICE_COMPILE__)
	// rows 1 and 3: e = max(e, lane 15 of the row below); rows 0 and 2 keep e (one DPP instruction; the two s_nop cover the
	// VALU-write -> DPP-read hazard on v and e whatever the scheduler puts in front)
	asm("s_nop 1\n\tv_max_i32_dpp %0, %1, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(e) : "v"(v));
#endif
	return e;
}
// inclusive prefix min over the 32 lanes of each half (general walk)
__device__ __forceinline__ int tw_incl_min32(int v)
{
	v = min(v, dpp_or_old<DPP_ROW_SHR(1), 0xf>(INT_MAX, v));
	v = min(v, dpp_or_old<DPP_ROW_SHR(2), 0xf>(INT_MAX, v));
	v = min(v, dpp_or_old<DPP_ROW_SHR(4), 0xf>(INT_MAX, v));
	v = min(v, dpp_or_old<DPP_ROW_SHR(8), 0xf>(INT_MAX, v));
	v = min(v, dpp_or_old<DPP_ROW_BCAST15, 0xa>(INT_MAX, v));
	return v;
}

// bits of the 64-bit lane mask m below this lane, counted inside the lane's own half
__device__ __forceinline__ int tw_below_in_half(uint64_t m, bool hi_half)
{
	const int lo = (int)__builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u);
	const int hi = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), 0u);
	return hi_half ? hi : lo;
}

// per half: mask of all 32 lanes if any bit of m is set in that half (four scalar instructions; written out because the
// compiler turns the C form into 64-bit vector compares)
__device__ __forceinline__ uint64_t tw_smear_halves(uint64_t m)
{
	uint32_t lo = 0, hi = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	asm("s_cmp_lg_u32 %2, 0\n\ts_cselect_b32 %0, -1, 0\n\ts_cmp_lg_u32 %3, 0\n\ts_cselect_b32 %1, -1, 0"
	    : "=&s"(lo), "=&s"(hi) : "s"(TW_UNI((uint32_t)m)), "s"(TW_UNI((uint32_t)(m >> 32))) : "scc");
#endif
	return (uint64_t)hi << 32 | lo;
}

// per half: the highest set bit of m alone, or the half's lane 0 if m has none there (s_flbit gives -1 for 0, and a
// shift only uses the low five bits of its count: 0x80000000 >> 31 = lane 0)
__device__ __forceinline__ uint64_t tw_last_or_lane0(uint64_t m)
{
	uint32_t lo = 0, hi = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	asm("s_flbit_i32_b32 %0, %2\n\ts_flbit_i32_b32 %1, %3\n\ts_lshr_b32 %0, 0x80000000, %0\n\ts_lshr_b32 %1, 0x80000000, %1"
	    : "=&s"(lo), "=&s"(hi) : "s"(TW_UNI((uint32_t)m)), "s"(TW_UNI((uint32_t)(m >> 32))));
#endif
	return (uint64_t)hi << 32 | lo;
}

// nonzero iff m has a set bit in BOTH halves (three scalar instructions with the compare the caller adds)
__device__ __forceinline__ uint32_t tw_both_halves(uint64_t m)
{
	uint32_t t = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	asm("s_cmp_lg_u32 %1, 0\n\ts_cselect_b32 %0, %2, 0" : "=s"(t) : "s"(TW_UNI((uint32_t)m)), "s"(TW_UNI((uint32_t)(m >> 32))) : "scc");
#endif
	return t;
}

// per half: bits of m below the half's lowest set bit of b (all of m where b has none)
__device__ __forceinline__ uint64_t tw_below_first(uint64_t m, uint64_t b)
{
	const uint32_t blo = (uint32_t)b, bhi = (uint32_t)(b >> 32);
	const uint32_t klo = blo ? (blo & (0u - blo)) - 1u : 0xffffffffu, khi = bhi ? (bhi & (0u - bhi)) - 1u : 0xffffffffu;
	return m & ((uint64_t)khi << 32 | klo);
}

#define TW_HI31 0x8000000080000000ull

struct TwinArgs {
	Params par;
	const int64_t *off;
	const ulonglong2 *a;
	const unsigned long long *sumq;
	const uint16_t *lut;
	int lut_stride;
	const Unit *units;
	const UnitAux *aux;               // beside units[]: rel0, table key, "general" flag
	const unsigned long long *counters;
	int32_t *f, *p, *v;
	int32_t *first_child;
	uint8_t *flags;
	Unit *left;                       // leftover list for k_chain_units
	unsigned int *left_cnt;
	unsigned int *queue;              // eight grab counters, 64 words apart (the halves' first grabs are dealt statically: they start behind them)
	const unsigned int *route;        // *route != 0: k_chain_quad (launched in front of this kernel) has taken the batch
	int force_left;                   // test switch: 1 hand every unit over untouched, 2 hand every unit over after its first tile (resumed there)
	int64_t total;                    // anchors of the batch
	unsigned long long *stamp;        // diagnostic run (CHAINDP_TWIN_STAMP): per block 8 counters; nullptr otherwise
};

// the state of a half that only the service path needs lives in LDS (TW_ST + 64 h), so that the pass loop carries
// nothing but what a pass reads
struct TwinCold {                     // 40 bytes at TW_ST + 56 h (8-byte aligned: read and written as 64-bit words)
	int64_t next;                     // next unit of this half (grid-stride over pairs)
	int64_t base;                     // global index of the unit's first anchor
	uint64_t x_carry;                 // x of the previous tile's last anchor
	int32_t rel0, room, read, tile0;  // unit start relative to its read; anchors the unit may have; its read; current tile's first anchor
};

// what a pass reads and writes, per half, replicated over the half's lanes
struct TwinHot {
	uint32_t S;                       // 16 * jtop + 8h, jtop = i - 1 - 32c: ring offset of lane 0's predecessor
	uint32_t m4;                      // 4 * (i - 1) + mark base: mark distance base and the scan's tag
	uint32_t pc, pend;                // XQ entry of the current anchor; end of the tile's entries
};
// What a scan carries into its second chunk (one scan in thirty) is not worth registers: it sits behind the half's cold state
// (TW_ST + 56 h + 40): 4 * max_j (-4: none), the running max of the scan (chain.c:274) minus one, n_skip, and the second chunks
// the unit has needed so far.
#define TW_CARRY 40u

// in-kernel stamps (where a wave's time goes): compiled in only with -DCHAINDP_TWIN_STAMPS, because even switched off they cost
// registers the kernel does not have to spare (make -C csrc stamps; then run with CHAINDP_TWIN_STAMP=1)
#ifdef CHAINDP_TWIN_STAMPS
#define TW_STAMP(...) __VA_ARGS__
#else
#define TW_STAMP(...)
#endif
// two sets of stamps inside service(), one per build (-DCHAINDP_TWIN_STAMPS=1: flush and unit switch; =2: the rest): both do not fit
#if defined(CHAINDP_TWIN_STAMPS) && CHAINDP_TWIN_STAMPS == 2
#define TW_STAMP_A(...)
#define TW_STAMP_B(...) __VA_ARGS__
#elif defined(CHAINDP_TWIN_STAMPS) && CHAINDP_TWIN_STAMPS == 1
#define TW_STAMP_A(...) __VA_ARGS__
#define TW_STAMP_B(...)
#else
#define TW_STAMP_A(...)
#define TW_STAMP_B(...)
#endif
// build 3: inside the unit switch (head: until the unit's record is there; take: until its first tile is there; tail: table, ring
// initialisation and the tile taken; n_unit counts the first tiles that were requested ahead)
#if defined(CHAINDP_TWIN_STAMPS) && CHAINDP_TWIN_STAMPS == 3
#define TW_STAMP_C(...) __VA_ARGS__
#else
#define TW_STAMP_C(...)
#endif
#define TW_NOW() __builtin_amdgcn_s_memtime()

template <bool SAMEGAP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_chain_twin(TwinArgs g)
{
	const int lane = threadIdx.x;
	const bool hi_half = lane >= 32;
	const int h = lane >> 5, hl = lane & 31;

	// ---- per-lane constants (vector registers on purpose, see TW_VREG)
	uint32_t L16 = (uint32_t)hl << 4;
	const uint32_t mkbase = TW_MK + TW_MK_HALF * (uint32_t)h;      // this half's mark words
	const uint32_t curbase = TW_XQ + TW_XQ_HALF * (uint32_t)h;     // this half's XQ entries
	uint32_t c_mkbase = mkbase;
	uint32_t c_far = mkbase + 256u;                                // its sink word
	uint32_t c_own = mkbase + ((uint32_t)hl << 2);                 // lane's own mark word in chunk 0
	uint32_t c_lut = TW_LUT + TW_LUT_HALF * (uint32_t)h;
	uint32_t c_8h = (uint32_t)h << 3;
	uint32_t c_M = (uint32_t)g.par.max_dist_x;
	uint32_t c_bw = (uint32_t)g.par.bw;
	uint32_t c_cbw = c_M - 1u > c_bw ? c_M - 1u - c_bw : 0u;
	const uint32_t mdq = (uint32_t)(g.par.max_dist_x < g.par.max_dist_y ? g.par.max_dist_x : g.par.max_dist_y);
	uint32_t c_dqoff = c_M - mdq;
	int c_ms = g.par.max_skip;
	int c_min = INT_MIN;
	int c_Mout = hl == 31 ? g.par.max_dist_x : INT_MAX;            // window test that only the half's last lane can fail
	uint32_t c_bwl = c_bw + c_lut;                                 // table address of the last entry
	uint32_t c_cbwl = c_cbw - c_lut;                               // (dd + c_lut) + this = dd + c_cbw
	TW_VREG(L16); TW_VREG(c_far); TW_VREG(c_own); TW_VREG(c_lut); TW_VREG(c_M);
	if (!SAMEGAP) TW_VREG(c_dqoff);
	TW_VREG(c_ms); TW_VREG(c_min); TW_VREG(c_Mout); TW_VREG(c_bwl); TW_VREG(c_cbwl);

	const uint64_t maxx = (uint64_t)(int64_t)g.par.max_dist_x;
	if (g.route && *g.route) return;                               // (uniform) the batch went four units per wave
	const int64_t n_units = (int64_t)(uint32_t)g.counters[0];
	// the kernel's 32-bit differences (and the signed window test) are exact while 129 * (max_dist_x + 1) < 2^31
	// Units of a few thousand anchors (map-ont shape) are few and each is a long serial chain: two of them side by side gain
	// nothing and their scans need the second chunk for one anchor in three.  Such batches go to k_chain_units as a whole.
	const int64_t n_single = (int64_t)(g.counters[0] >> 32);
	const bool short_units = (g.total - n_single) <= 512 * n_units;
	if (!short_units) {                                            // (uniform: every block leaves; one of them says so)
		if (blockIdx.x == 0 && lane == 0) *g.left_cnt = 0xffffffffu;
		return;
	}
	const bool params_ok = g.lut != nullptr && !g.par.is_cdna && g.par.max_dist_x >= 1 && g.par.max_dist_y >= 0 &&
	                       ((uint64_t)(int64_t)g.par.max_dist_x + 1) * 129ull < (1ull << 31) && g.par.bw + 1 <= (int)TW_LUT_HALF && g.force_left != 1;

	TwinHot u;
	u.S = 0; u.m4 = 0; u.pc = curbase; u.pend = curbase;
	uint64_t live_m = ~0ull;                                       // halves that still have (or may get) work
	uint64_t contm = 0;                                            // halves that are in their second (= last) chunk
	const uint32_t st_addr = TW_ST + TW_ST_HALF * (uint32_t)h;
#define TW_COLD (*TW_LDS(TwinCold, st_addr))
	// The unit queue.  Eight counters, a cache line apart (one same-address atomic takes ~7.5 ns: 133 M a second for the whole chip
	// on ONE counter), workgroup b on counter b mod 8.  A half's p-th grab of its counter k is the chunk of TW_QCH units number
	// 8 p + k while the list's long front lasts (units [0, U1)), and ONE unit -- number 8 (p - G1k) + k of [U1, n_units) -- for the
	// last sixteen units per half: the list is longest first, every half works on units of the same length at any time, and the
	// halves run out of work within one grab of each other -- eight units of 140 anchors were 0.4 ms, 0.2 ms of idle tail on average
	// behind a 3 ms kernel.  (Smaller grabs all along cost more than they save: same-address atomics.)
	const uint32_t xcd = blockIdx.x & 7u;
	const uint32_t n_halves = 2u * gridDim.x;
#ifndef TW_END_UNITS
#define TW_END_UNITS 16u                // units per half that are dealt in small pieces at the end of the list
#define TW_PIECE 1u                     // ... this many at a time (measured on the 76 M-anchor shard: 16 / 1: 2.86 ms, 16 / 2: 2.88, 24 / 1: 2.88, 32 / 2: 2.90, 8 / 2: 2.95; one counter, 8 at a time all along: 3.04)
#endif
	const uint32_t U1 = n_units > (long long)TW_END_UNITS * n_halves ? ((uint32_t)n_units - TW_END_UNITS * n_halves) & ~(8u * TW_QCH - 1u) : 0u;
	const uint32_t G1k = U1 / (8u * TW_QCH);                             // grabs of whole chunks per counter
	auto grab = [&](uint32_t p, uint32_t &nx, uint32_t &ne) {
		if (p < G1k) { nx = (8u * p + xcd) * TW_QCH; ne = nx + TW_QCH; }
		else { nx = U1 + TW_PIECE * (8u * (p - G1k) + xcd); ne = nx + TW_PIECE; }
	};
	if (hl == 0) {
		uint32_t nx, ne;
		grab(((blockIdx.x >> 3) << 1) | (uint32_t)h, nx, ne);                            // the half's first grab: dealt statically
		tw_st64(st_addr, nx, ne); tw_st64(st_addr + 8u, 0u, 0u);                         // TwinCold: next (low word: next unit, high word: end of the chunk), base
		tw_st64(st_addr + 16u, 0u, 0u); tw_st64(st_addr + 24u, 0u, 0u);                 // x_carry, rel0, room
		tw_st64(st_addr + 32u, 0u, (uint32_t)-TW_TILE);                                 // read, tile0
		tw_st64(st_addr + TW_CARRY, 0xfffffffcu, 0u); tw_st64(st_addr + TW_CARRY + 8u, 0u, 0u);   // carry, second chunks so far
		tw_st32(TW_KEY + 4u * (uint32_t)h, -1);                                         // no table yet (an avg_qspan is never a NaN)
	}
	wave_mem_fence();

	TW_STAMP(unsigned long long st_t0 = 0; unsigned int st_service = 0, st_n_service = 0, st_n_fast = 0, st_flush = 0, st_unit = 0, st_n_unit = 0, st_head = 0, st_take = 0, st_tail = 0;)   // (32-bit sums: a wave's ticks fit, and the build has no registers to spare)
	uint64_t nx0_x = 0, nx0_y = 0, nx1_x = 0, nx1_y = 0;           // each half's NEXT tile, requested a tile ahead (one anchor per lane; zeros where the unit has none)
	int32_t pfu0 = -1, pfu1 = -1;                                  // ... or, when the unit ends with the current tile, the first tile of the half's NEXT
	                                                               // unit: its first anchor's global index (wave-uniform), -1 = the registers hold no such tile

	// One service round for the halves in `svc`, whose tile is exhausted (or which have no unit yet).  One half at a time, by ALL 64
	// lanes of the wave (a tile is 64 anchors, one per lane): everything that is per half (the cold state, the unit being picked,
	// loop conditions) is wave-uniform and lives in scalar registers, loads by scalar address go through the scalar cache, and the
	// loops are scalar branches; what goes back into the half's hot state is selected into its 32 lanes (TW_SEL by `hm`).  Order:
	// the unit's NEXT tile (requested ahead) is taken first -- before the finished tile's f/p/v are stored, so that nothing waits
	// for those stores --, then the finished tile is flushed; a half whose unit is over picks its next unit and loads that one's
	// first tile.  Once per 64 anchors and half.
	auto service = [&](uint64_t svc) {
#if defined(__HIP_DEVICE_COMPILE__)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (whatever is outstanding was issued a tile of passes ago: no wait in practice)
#endif
		wave_mem_fence();
		uint64_t retired = 0;
		for (int hs = 0; hs < 2; ++hs) {
			if (((svc >> (32 * hs)) & 1ull) == 0) continue;
			TW_STAMP_B(const unsigned long long th0 = g.stamp ? TW_NOW() : 0;)
			const uint64_t hm = hs ? 0xffffffff00000000ull : 0x00000000ffffffffull;   // the lanes that carry this half's hot state
			const uint32_t sa = TW_ST + TW_ST_HALF * (uint32_t)hs;
			const uint32_t curb = TW_XQ + TW_XQ_HALF * (uint32_t)hs, spb = TW_SP + TW_SP_HALF * (uint32_t)hs;   // (SP: a byte per anchor)
			const uint32_t mkb = TW_MK + TW_MK_HALF * (uint32_t)hs, lutb = TW_LUT + TW_LUT_HALF * (uint32_t)hs;
			const tw_u32x2 cw0 = tw_ld64(sa), cw1 = tw_ld64(sa + 8u), cw2 = tw_ld64(sa + 16u), cw3 = tw_ld64(sa + 24u), cw4 = tw_ld64(sa + 32u);
			int64_t c_next = (int64_t)((uint64_t)TW_UNI(cw0.y) << 32 | TW_UNI(cw0.x));
			int64_t c_base = (int64_t)((uint64_t)TW_UNI(cw1.y) << 32 | TW_UNI(cw1.x));
			uint64_t c_xcarry = (uint64_t)TW_UNI(cw2.y) << 32 | TW_UNI(cw2.x);
			int c_rel0 = (int)TW_UNI(cw3.x), c_room = (int)TW_UNI(cw3.y), c_read = (int)TW_UNI(cw4.x), c_tile0 = (int)TW_UNI(cw4.y);
			const int cnt_prev = (int)(((uint32_t)__builtin_amdgcn_readlane((int)u.pend, 32 * hs) - curb) >> 3);   // anchors of the tile that has just been scored
			const int slow_h = (int)TW_UNI((uint32_t)tw_ld32(sa + TW_CARRY + 12u));
			const uint32_t cur_key = TW_UNI((uint32_t)tw_ld32(TW_KEY + 4u * (uint32_t)hs));
			const int tile_prev = c_tile0, rel0_prev = c_rel0;
			const int64_t base_prev = c_base;
			bool live = true;
			// The half's next unit (and the one after it), when this call will need them -- the unit ends here, or with the tile taken
			// now (its successor's first tile is then requested a tile ahead, like any other tile): records and UnitAux through the
			// scalar cache, issued before the work below and read after it.
			const uint32_t nx0 = (uint32_t)c_next, ne0 = (uint32_t)((uint64_t)c_next >> 32);
			tw_u32x4 rec0 = {0u, 0u, 0u, 0u}, aux0 = {0u, 0u, 0u, 0u}, rec1 = {0u, 0u, 0u, 0u};
			const bool rec0_ok = nx0 < ne0 && (int64_t)nx0 < n_units;       // (whenever they are known: a unit can end before its bound says so)
			const bool rec1_ok = rec0_ok && nx0 + 1u < ne0 && (int64_t)nx0 + 1 < n_units;
			if (rec0_ok) { rec0 = *TW_CONST(tw_u32x4, g.units + nx0); aux0 = *TW_CONST(tw_u32x4, g.aux + nx0); }
			if (rec1_ok) rec1 = *TW_CONST(tw_u32x4, g.units + nx0 + 1u);

			// takes a tile's anchors (one per lane, raw mm128_t) into the half's LDS: where the unit ends (first gap > max_dist_x,
			// chain.c:252), XY ring (the anchors as predecessors), XQ / SP (as the current anchor).  Returns the anchors the tile holds
			// (0: the unit ended exactly at its start).
			auto take_tile = [&](const uint64_t an_x, const uint64_t an_y) -> int {
				const int i_lane = c_tile0 + lane;
				const bool have = i_lane < c_room;
				uint64_t xp;
				{
					uint32_t lo = (uint32_t)wave_shift_up1((int)(uint32_t)an_x, 0), hi = (uint32_t)wave_shift_up1((int)(uint32_t)(an_x >> 32), 0);
					if (lane == 0) { lo = (uint32_t)c_xcarry; hi = (uint32_t)(c_xcarry >> 32); }
					xp = (uint64_t)hi << 32 | lo;
				}
				const bool stop = !have || (i_lane > 0 && an_x - xp > maxx);
				const uint64_t stop_m = __builtin_amdgcn_ballot_w64(stop);
				const int cnt = stop_m ? __builtin_ctzll(stop_m) : TW_TILE;
				c_xcarry = readlane_u64(an_x, 63);
				u.pc = TW_SEL(hm, curb, u.pc); u.pend = TW_SEL(hm, curb + ((uint32_t)cnt << 3), u.pend);
				if (cnt == 0) return 0;
				wave_mem_fence();
				if (lane < cnt) {
					const int sp = span_of_hi((uint32_t)(an_y >> 32));
					tw_st64((((uint32_t)i_lane & 127u) << 4 | (uint32_t)hs << 3) + TW_XY, (uint32_t)an_x + 1u, (uint32_t)an_y + 1u);
					tw_st64(curb + ((uint32_t)lane << 3), (uint32_t)an_x, (uint32_t)an_y);
					tw_st8(spb + (uint32_t)lane, sp - 1);
					// first_child[] starts at "none" for every anchor the kernel takes in: stored here, a whole tile of passes before the
					// tile's flush (or any later one) lowers it with atomics -- and service() waits for the wave's outstanding memory
					// operations when it starts, so those atomics come after this store in memory as well.  No batch-wide memset.
					g.first_child[c_base + i_lane] = NO_CHILD;
				}
				wave_mem_fence();
				// the tile after this one: the load is issued now and read at the half's next service, 64 anchors of work later
				if (cnt == TW_TILE && c_tile0 + TW_TILE < c_room) {            // (else: the unit ends with this tile; the registers are for its successor)
					uint64_t rx = 0, ry = 0;
					if (i_lane + TW_TILE < c_room) { const ulonglong2 t = g.a[c_base + i_lane + TW_TILE]; rx = t.x; ry = t.y; }
					if (hs) { nx1_x = rx; nx1_y = ry; pfu1 = -1; } else { nx0_x = rx; nx0_y = ry; pfu0 = -1; }
				}
				return cnt;
			};

			// ---- the unit goes on?
			bool goes_on = cnt_prev == TW_TILE && c_tile0 + TW_TILE < c_room;
			if (goes_on && (slow_h * 8 > c_tile0 + TW_TILE || g.force_left == 2)) {
				// a unit that keeps needing second chunks (more than one anchor in eight) is cheaper in k_chain_units: hand the rest of
				// it over.  The tiles up to the one flushed below are done: k_chain_units goes on behind them (the count rides in the
				// high word of the start; force_left == 2 is the tests' way to send every unit down this road)
				if (lane == 0) {
					Unit un; un.start = (int64_t)((uint64_t)c_base | (uint64_t)(uint32_t)(c_tile0 + TW_TILE) << 32); un.read = c_read; un.len = c_room;
					g.left[atomicAdd(g.left_cnt, 1u)] = un;
				}
				goes_on = false;
			}
			if (goes_on) {
				c_tile0 += TW_TILE;
				if (take_tile(hs ? nx1_x : nx0_x, hs ? nx1_y : nx0_y) == 0) goes_on = false;   // it ended exactly on the boundary
			}
			TW_STAMP_B(if (g.stamp) st_take += (unsigned int)(TW_NOW() - th0);)
			// ---- flush the finished tile
			TW_STAMP_A(const unsigned long long tf0 = g.stamp ? TW_NOW() : 0;)
			if (cnt_prev > 0) {
				const int i_lane = tile_prev + lane;                         // this lane's anchor of the finished tile
				const bool have = lane < cnt_prev;
				const int64_t gi = base_prev + i_lane;
				int fi = 0, p4 = -4;
				if (have) {
					const tw_u32x2 pf = tw_ld64((((uint32_t)i_lane & 63u) << 4 | (uint32_t)hs << 3) + TW_PF);
					p4 = (int)pf.x; fi = (int)pf.y + 1;                      // (the ring holds f - 1)
				}
				const int pi = p4 >> 2;                                      // unit-relative predecessor, -1 = none
				int val = fi, ptr = have ? pi : -1;
				const bool ext = ptr >= 0 && ptr < tile_prev;                // predecessor in an earlier tile: its v is final, in the V ring
				int vext = 0;
				if (ext) {
					vext = tw_ld32((((uint32_t)ptr & 63u) << 3 | (uint32_t)hs << 2) + TW_V);
					val = max(val, vext & 0x7fffffff);
					ptr = -1;
				}
				const bool ext_self = ext && vext < 0;                       // (bit 31 of a V entry: the anchor was emitted at its own step)
				for (int r = 0; r < 6; ++r) {                                // v[i] = max(f[i], v[p[i]]) (chain.c:284) by pointer doubling over the tile
					if (__builtin_amdgcn_ballot_w64(ptr >= tile_prev) == 0) break;
					const int src = (ptr >= tile_prev ? ptr - tile_prev : lane) << 2;
					const int pv = __builtin_amdgcn_ds_bpermute(src, val);
					const int pp = __builtin_amdgcn_ds_bpermute(src, ptr);
					if (ptr >= tile_prev) { val = max(val, pv); ptr = pp; }
				}
				const bool self = val >= g.par.min_sc || pi >= 0;            // emitted at its own step (chain.c:304)
				// is the predecessor emitted at its own step?  in-tile predecessors: ask their lane
				const int srcp = (pi >= tile_prev ? pi - tile_prev : lane) << 2;
				const int pself_in = __builtin_amdgcn_ds_bpermute(srcp, self ? 1 : 0);
				const bool pred_self = ext ? ext_self : pself_in != 0;
				wave_mem_fence();
				if (have) tw_st32((((uint32_t)i_lane & 63u) << 3 | (uint32_t)hs << 2) + TW_V, val | (self ? INT_MIN : 0));
				wave_mem_fence();
				// (first_child[] of the tile's anchors is NO_CHILD since the tile was taken in)
				if (have) {
					g.f[gi] = fi;
					g.p[gi] = pi < 0 ? -1 : pi + rel0_prev;
					g.v[gi] = val;
					int maybe_first = 0;
					if (pi >= 0 && !pred_self) { atomicMin(&g.first_child[base_prev + pi], rel0_prev + i_lane); maybe_first = 4; }
					g.flags[gi] = (uint8_t)((self ? 2 : 0) | maybe_first | (val >= g.par.min_sc ? 8 : 0) | (fi < val ? 16 : 0));
				}
			}
			TW_STAMP_A(if (g.stamp) st_flush += (unsigned int)(TW_NOW() - tf0);)
			TW_STAMP_A(const unsigned long long tu0 = g.stamp ? TW_NOW() : 0;)
			TW_STAMP_A(if (g.stamp && !goes_on) ++st_n_unit;)
			// ---- the unit is over: the half's next unit (units that are not for this kernel are handed over), its LDS, its first tile
			bool have_rec = rec0_ok;                                       // rec0 / aux0 are the records of unit c_next
			const bool rec0_used = !goes_on;                               // the loop below takes unit nx0 (or hands it over)
			uint32_t cur_key_now = cur_key;
			while (!goes_on && live) {
				uint32_t nx = (uint32_t)c_next, ne = (uint32_t)((uint64_t)c_next >> 32);
				if (nx >= ne) {
					// the chunk is used up: the counter's next grab (the list is longest first, so the two halves of a wave, and all
					// waves, work on units of similar length at any time and run out of work together)
					uint32_t q0 = 0;
					if (lane == 0) q0 = atomicAdd(g.queue + 64u * xcd, 1u);
					grab(TW_UNI(q0), nx, ne);
					have_rec = false;
				}
				if ((int64_t)nx >= n_units) {
					c_next = (int64_t)((uint64_t)ne << 32 | nx); live = false;
					u.pc = TW_SEL(hm, curb, u.pc); u.pend = TW_SEL(hm, curb, u.pend);
					break;
				}
				TW_STAMP_C(const unsigned long long tc0 = g.stamp ? TW_NOW() : 0;)
				if (!have_rec) { rec0 = *TW_CONST(tw_u32x4, g.units + nx); aux0 = *TW_CONST(tw_u32x4, g.aux + nx); }
				have_rec = false;
				TW_STAMP_C(if (g.stamp) { asm volatile("" :: "s"(rec0.x), "s"(aux0.x)); st_head += (unsigned int)(TW_NOW() - tc0); })
				c_next = (int64_t)((uint64_t)ne << 32 | (nx + 1u));
				Unit un;
				un.start = (int64_t)((uint64_t)rec0.y << 32 | rec0.x); un.read = (int32_t)rec0.z; un.len = (int32_t)rec0.w;
				if (!params_ok || (aux0.z & 1u) || g.par.n_segs > 1) {         // not for this kernel: hand the unit over
					if (lane == 0) g.left[atomicAdd(g.left_cnt, 1u)] = un;
					continue;
				}
				c_base = un.start; c_rel0 = (int)aux0.x; c_room = un.len; c_read = un.read; c_tile0 = 0;
				// the unit's first tile: requested a tile ago if the unit before it ended as foreseen
				uint64_t tl_x, tl_y;
				TW_STAMP_C(const unsigned long long tc1 = g.stamp ? TW_NOW() : 0;)
				TW_STAMP_C(if (g.stamp && (hs ? pfu1 : pfu0) == (int32_t)c_base) ++st_n_unit;)
				if ((hs ? pfu1 : pfu0) == (int32_t)c_base) { tl_x = hs ? nx1_x : nx0_x; tl_y = hs ? nx1_y : nx0_y; }
				else {
					tl_x = 0; tl_y = 0;
					if (lane < c_room) { const ulonglong2 t = g.a[c_base + lane]; tl_x = t.x; tl_y = t.y; }
				}
				if (hs) pfu1 = -1; else pfu0 = -1;
				TW_STAMP_C(if (g.stamp) { asm volatile("" :: "v"(tl_x), "v"(tl_y)); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st_take += (unsigned int)(TW_NOW() - tc1); })
				TW_STAMP_C(const unsigned long long tc2 = g.stamp ? TW_NOW() : 0;)
				// LDS of the half for a new unit: marks never match, every XY slot fails the window test, the read's table (as bytes) unless
				// the table in place is that of a read with the same avg_qspan
				wave_mem_fence();
				if (aux0.y != cur_key_now) {
					const uint2 *src = (const uint2*)(g.lut + (int64_t)c_read * g.lut_stride);
					for (int k = lane; k * 4 <= g.par.bw; k += 64) {           // lut_stride is a multiple of 8 entries: whole uint2 loads
						const uint2 t = src[k];
						const uint32_t w = (t.x & 0xffu) | (t.x >> 8 & 0xff00u) | (t.y << 16 & 0xff0000u) | (t.y << 8 & 0xff000000u);
						tw_st32(lutb + ((uint32_t)k << 2), (int)w);
					}
					cur_key_now = aux0.y;
					if (lane == 0) tw_st32(TW_KEY + 4u * (uint32_t)hs, (int)cur_key_now);
				}
				const uint32_t x_none = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tl_x, 0) - (uint32_t)maxx - 1u;   // "no anchor here" (x+1 encoding)
				for (int k = lane; k < 128; k += 64) tw_st64(((uint32_t)k << 4 | (uint32_t)hs << 3) + TW_XY, x_none, 0u);
				for (int k = lane; k < 65; k += 64) tw_st32(mkb + ((uint32_t)k << 2), -1);
				wave_mem_fence();
				c_xcarry = 0;
				if (lane == 0) tw_st32(sa + TW_CARRY + 12u, 0);
				if (take_tile(tl_x, tl_y) > 0) goes_on = true;                 // (a unit has at least two anchors: always)
				TW_STAMP_C(if (g.stamp) { st_tail += (unsigned int)(TW_NOW() - tc2); ++st_unit; })
			}
			// ---- the unit ends with the tile just taken: request its successor's first tile now, if the successor is known
			if (live) {
				const uint32_t cnt_now = ((uint32_t)__builtin_amdgcn_readlane((int)u.pend, 32 * hs) - curb) >> 3;
				const uint32_t nxn = (uint32_t)c_next, nen = (uint32_t)((uint64_t)c_next >> 32);
				if (!(cnt_now == TW_TILE && c_tile0 + TW_TILE < c_room) && (hs ? pfu1 : pfu0) < 0 && nxn < nen && (int64_t)nxn < n_units &&
				    ((nxn == nx0 && rec0_ok && !rec0_used) || (nxn == nx0 + 1u && rec1_ok))) {
					const tw_u32x4 rn = nxn == nx0 ? rec0 : rec1;                // (nxn == nx0: the loop above did not run, rec0 is untouched)
					const int32_t st = (int32_t)rn.x, ln = (int32_t)rn.w;
					uint64_t rx = 0, ry = 0;
					if (lane < ln) { const ulonglong2 t = g.a[(int64_t)st + lane]; rx = t.x; ry = t.y; }
					if (hs) { nx1_x = rx; nx1_y = ry; pfu1 = st; } else { nx0_x = rx; nx0_y = ry; pfu0 = st; }
				}
			}
			TW_STAMP_A(if (g.stamp) st_unit += (unsigned int)(TW_NOW() - tu0);)
			// ---- the tile's first anchor becomes current
			if (live) {
				const uint32_t i = (uint32_t)c_tile0;
				u.S = TW_SEL(hm, (i - 1u) << 4 | (uint32_t)hs << 3, u.S); u.m4 = TW_SEL(hm, ((i - 1u) << 2) + mkb, u.m4);
			} else retired |= hm;
			if (lane == 0) {
				tw_st64(sa, (uint32_t)c_next, (uint32_t)((uint64_t)c_next >> 32)); tw_st64(sa + 8u, (uint32_t)c_base, (uint32_t)((uint64_t)c_base >> 32));
				tw_st64(sa + 16u, (uint32_t)c_xcarry, (uint32_t)(c_xcarry >> 32)); tw_st64(sa + 24u, (uint32_t)c_rel0, (uint32_t)c_room);
				tw_st64(sa + 32u, (uint32_t)c_read, (uint32_t)c_tile0);
			}
		}
		wave_mem_fence();
		live_m &= ~retired;
		contm &= ~svc;
	};

	// The tail of a pass in which not both halves finish their scan in their first chunk (or a half is idle): per half either
	// the next anchor becomes current, or the second chunk follows, or -- still undecided after the second chunk -- the unit is
	// handed over.  Returns the halves to service.
	auto slow_tail = [&](uint64_t D, uint32_t a_cur, int nskip_after) -> uint64_t {
		// There are two chunks.  Lane 31 of the second (j = i - 64) is not evaluated (its PF slot is anchor i's own), so a half
		// that is still undecided after it is handed over to k_chain_units.
		const uint64_t giveup = ~D & contm;
		const int vlast = __builtin_amdgcn_ds_bpermute(((h << 5) + 31) << 2, nskip_after);
		wave_mem_fence();
		const tw_u32x2 cur = tw_ld64(a_cur);                                 // the running max just written
		if (__builtin_amdgcn_inverse_ballot_w64(D)) {
			u.m4 += 4u;
			u.S = ((u.m4 - c_mkbase) << 2) | c_8h;
			u.pc += 8u;
		} else {
			if (hl == 0) {                                                   // what the second chunk starts from
				tw_st64(st_addr + TW_CARRY, cur.x, cur.y);
				tw_st32(st_addr + TW_CARRY + 8u, vlast);
				tw_st32(st_addr + TW_CARRY + 12u, tw_ld32(st_addr + TW_CARRY + 12u) + 1);
			}
			u.S -= 512u;
		}
		contm = ~D & ~giveup & live_m;
		if (__builtin_expect(giveup != 0, 0)) {
			if (__builtin_amdgcn_inverse_ballot_w64(giveup)) {
				wave_mem_fence();
				if (hl == 0) {
					const TwinCold c = TW_COLD;
					Unit un; un.start = (int64_t)((uint64_t)c.base | (uint64_t)(uint32_t)c.tile0 << 32); un.read = c.read; un.len = c.room;
					g.left[atomicAdd(g.left_cnt, 1u)] = un;                  // k_chain_units goes on from the tile this scan is in (the tiles
					                                                         // before it are flushed).  The unit is over for this kernel: an empty tile ...
				}
				u.pc = curbase; u.pend = curbase;                            // ... has nothing to flush and cannot go on: service() picks the half's next unit
			}
			return giveup;
		}
		return 0;
	};

	TW_STAMP(if (g.stamp) st_t0 = TW_NOW();)
	service(~0ull);
	bool force_general = false;
	// ======================================================================================== main loop: one chunk pass per trip
	while (live_m != 0) {
		wave_mem_fence();                                                    // PF[i-1] of the previous pass, rings written by service()
		uint64_t svc = 0;
		if (__builtin_expect(contm == 0 && live_m == ~0ull && !force_general, 1)) {
			// ------------------------------------------------------------ both halves in their first chunk (n_skip = 0, max_j = none).
			// A loop of its own: while both halves finish every scan in the first chunk nothing but the pass below runs, and its
			// state is updated in place.
			uint64_t B, X, tile;
			int cB;
			uint32_t a_cur;
			for (;;) {
				const uint32_t t0 = u.S - L16;                               // lane k <-> predecessor j = jtop - k of its half's anchor
				const tw_u32x2 xy = tw_ld64((t0 & 0x7f8u) + TW_XY);
				const tw_u32x2 pf = tw_ld64((t0 & 0x3f8u) + TW_PF);
				const tw_u32x2 cur = tw_ld64(u.pc);                          // the anchor itself: x, q
				const int spm1 = tw_ld_u8((u.pc >> 3) + TW_SP_OF_XQ);        // ... and q_span - 1
				const uint32_t drm1 = cur.x - xy.x, dqm1 = cur.y - xy.y;     // the ring holds x + 1, q + 1: differences minus one
				const uint32_t ddl = tw_sad(drm1, dqm1, c_lut);              // |dr - dq| + the half's table base
				const uint32_t dqs = SAMEGAP ? dqm1 : __builtin_elementwise_add_sat(dqm1, c_dqoff);
				const uint32_t m3 = max(max(drm1, dqs), ddl + c_cbwl);
				const uint64_t okm = TW_ULT(m3, c_M);                        // chain.c:252-260 as one compare
				// the mark round trip (chain.c:281: store by distance, the others to the sink; then the lane's own word) and the table
				// lookup are issued back to back, before anything waits for either
				const uint32_t dst = TW_SEL(okm, min(u.m4 - pf.x, c_far), c_far);
				tw_st32(dst, (int)u.m4);
				wave_mem_fence();
				const int tj = tw_ld32(c_own);
				const int lutv = tw_ld_i8(min(ddl, c_bwl));
#if defined(__HIP_DEVICE_COMPILE__)
				__builtin_amdgcn_sched_barrier(0);
#endif
				const int sc0 = min(min((int)dqm1, (int)drm1), spm1);        // chain.c:262-263, minus one
				const int sc = TW_SEL(okm, sc0 + (int)pf.y + lutv, c_min);   // chain.c:272-273 via the table, minus one (the ring holds f - 1)
				const int excl = max(tw_excl_max32(sc), spm1);               // (q_span - 1 >= 0: the scan's zero fill stays below it)
				const uint64_t A = TW_SGT(sc, excl);                         // new running max (chain.c:274); masked lanes hold INT_MIN
				B = TW_EQ(tj, u.m4) & okm & ~A;                              // marked and not better (chain.c:277)
				const uint64_t OUT = TW_SGE(drm1, c_Mout);                   // the half's last lane is outside the window (or no anchor there yet)
				cB = tw_below_in_half(B, hi_half);
				// n_skip walk (chain.c:276,278) from n_skip = 0.  When every A lane of a half precedes every B lane of it, n_skip at a
				// B lane is the number of B lanes up to it: the break is the (max_skip + 1)-th of them.  An A lane above a B lane
				// (interleaved, rare) leaves this loop for the general pass, which redoes the anchor.
				tile = 0; X = 0; a_cur = 0;
				if (__builtin_expect((TW_SGT(cB, 0) & A) != 0, 0)) { force_general = true; break; }
				// the running max goes to PF[i]: the half's last A lane writes its own score and predecessor, or (none) the half's
				// lane 0 writes "no predecessor, q_span" (minus one)
				const uint32_t S1 = u.S + 16u;
				a_cur = S1 & 0x3f8u;                                         // PF slot of anchor i (S = 16 (i - 1) + 8h in the first chunk)
				{
					const uint32_t wp = TW_SEL(A, u.m4 - c_own, 0xfffffffcu);      // 4 j of the lane's predecessor: 4 (i - 1 - k)
					const int wf = TW_SEL(A, sc, spm1);
					if (__builtin_amdgcn_inverse_ballot_w64(tw_last_or_lane0(A))) tw_st64(a_cur + TW_PF, wp, (uint32_t)wf);
				}
				// scan complete: break taken, or the half's last lane is outside the window (x is sorted: nothing older can matter)
				X = (TW_SGE(cB, c_ms) & B) | OUT;
				if (__builtin_expect(tw_both_halves(X) == 0, 0)) break;
				u.m4 += 4u; u.S = S1;
				u.pc += 8u;
				tile = TW_SGE(u.pc, u.pend);
				TW_STAMP(if (g.stamp) ++st_n_fast;)
				if (__builtin_expect(tile != 0, 0)) break;
				wave_mem_fence();
			}
			if (!force_general && tile == 0) {
				// a half wants its second chunk.  n_skip after the first: #B, as no A lane follows a B lane
				svc = slow_tail(tw_smear_halves(X), a_cur + TW_PF, cB + (int)__builtin_amdgcn_inverse_ballot_w64(B));
			}
		} else {
			// ------------------------------------------------------------ general pass: second chunks, idle halves, interleaved walks
			force_general = false;
			const uint32_t t0 = u.S - L16;
			const tw_u32x2 xy = tw_ld64((t0 & 0x7f8u) + TW_XY);
			const tw_u32x2 pf = tw_ld64((t0 & 0x3f8u) + TW_PF);
			const tw_u32x2 cur = tw_ld64(u.pc);
			const int spm1 = tw_ld_u8((u.pc >> 3) + TW_SP_OF_XQ);
			const uint64_t first = ~contm;                                       // halves in their first chunk: running max = q_span, nothing carried
			const tw_u32x2 carry = tw_ld64(st_addr + TW_CARRY);
			const int maxf = TW_SEL(first, spm1, (int)carry.y);
			const uint32_t maxj4 = TW_SEL(first, 0xfffffffcu, carry.x);
			const int nskip0 = TW_SEL(first, 0, tw_ld32(st_addr + TW_CARRY + 8u));
			const uint32_t kb4 = TW_SEL(first, 0u, 128u);                        // 128 * c
			const uint32_t drm1 = cur.x - xy.x, dqm1 = cur.y - xy.y;
			const uint32_t ddl = tw_sad(drm1, dqm1, c_lut);
			const uint32_t dqs = SAMEGAP ? dqm1 : __builtin_elementwise_add_sat(dqm1, c_dqoff);
			const uint32_t m3 = max(max(drm1, dqs), ddl + c_cbwl);
			// not evaluated: lane 31 of a second chunk (j = i - 64 shares its PF slot with anchor i itself) and idle halves
			const uint64_t okm = TW_ULT(m3, c_M) & ~(contm & TW_HI31) & live_m;
			const int sc0 = min(min((int)dqm1, (int)drm1), spm1);
			const int lutv = tw_ld_i8(min(ddl, c_bwl));
			const uint32_t dst = TW_SEL(okm, min(u.m4 - pf.x, c_far), c_far);
			tw_st32(dst, (int)u.m4);
			wave_mem_fence();
			const int tj = tw_ld32(c_own + kb4);
			const int sc = TW_SEL(okm, sc0 + (int)pf.y + lutv, c_min);
			const int excl = max(tw_excl_max32(sc), maxf);
			const uint64_t A = TW_SGT(sc, excl);
			const uint64_t B = TW_EQ(tj, u.m4) & okm & ~A;
			const uint64_t OUT = TW_SGE(drm1, c_M);
			const int cB = tw_below_in_half(B, hi_half), cA = tw_below_in_half(A, hi_half);
			const int isA = (int)__builtin_amdgcn_inverse_ballot_w64(A), isB = (int)__builtin_amdgcn_inverse_ballot_w64(B);
			const uint64_t inter = TW_SGT(cB, 0) & A;
			uint64_t brk, Ap;
			int nskip_after;
			if (inter == 0) {
				// at a B lane all A lanes of the half are below it: n_skip = max(n0 - #A, 0) + #B up to and including it
				nskip_after = max(nskip0 - cA - isA, 0) + cB + isB;
				brk = TW_SGT(nskip_after, c_ms) & B;
				Ap = A;
			} else {
				const int Sk = nskip0 + cB + isB - cA - isA;
				nskip_after = Sk - min(tw_incl_min32(Sk), 0);
				brk = TW_SGT(nskip_after, c_ms) & B;
				Ap = tw_below_first(A, brk);
			}
			const uint32_t a_cur = (((((u.m4 - c_mkbase) + 4u) << 2) & 0x3f8u) | c_8h) + TW_PF;   // PF slot of anchor i (m4 = 4 (i - 1) + mark base)
			{
				const uint32_t wp = TW_SEL(Ap, u.m4 - c_own - kb4, maxj4);          // 4 j = 4 (i - 1 - 32 c - k)
				const int wf = TW_SEL(Ap, sc, maxf);
				if (__builtin_amdgcn_inverse_ballot_w64(tw_last_or_lane0(Ap))) tw_st64(a_cur, wp, (uint32_t)wf);
			}
			const uint64_t D = tw_smear_halves(brk | (OUT & TW_HI31) | ~live_m);   // idle halves count as done
			svc = slow_tail(D, a_cur, nskip_after);
		}
		// ---------------------------------------------------------------- tile exhausted (or unit handed over): flush, next tile / unit
		svc |= tw_smear_halves(TW_SGE(u.pc, u.pend) & live_m & ~contm);
		if (__builtin_expect(svc != 0, 0)) {
			TW_STAMP(const unsigned long long ts = g.stamp ? TW_NOW() : 0;)
			service(svc & live_m);
			TW_STAMP(if (g.stamp) { st_service += (unsigned int)(TW_NOW() - ts); ++st_n_service; })
		}
	}
	TW_STAMP(if (g.stamp && lane == 0) {
		unsigned long long *o = g.stamp + 12 * (size_t)blockIdx.x;
		o[0] = TW_NOW() - st_t0; o[1] = st_service; o[2] = st_flush; o[3] = st_unit; o[4] = st_n_service; o[5] = st_n_fast; o[6] = st_n_unit; o[7] = 1; o[8] = st_head; o[9] = st_take; o[10] = st_tail;
	})
#undef TW_COLD
}

size_t twin_lds_bytes() { return TW_LDS_BYTES; }

hipError_t launch_chain_twin(hipStream_t st, const Params &par, int64_t max_units, const int64_t *d_off, const void *d_a,
                             const unsigned long long *d_sumq, const uint16_t *d_lut, int lut_stride, const Unit *d_units,
                             const unsigned long long *d_counters, int32_t *d_f, int32_t *d_p, int32_t *d_v,
                             int32_t *d_first_child, uint8_t *d_flags, Unit *d_left, unsigned int *d_left_cnt, int force_left, int64_t total,
                             const UnitAux *d_unit_aux, const unsigned int *d_route, unsigned int *d_queue)
{
	if (max_units <= 0) return hipSuccess;
	if (!d_unit_aux) return hipErrorInvalidValue;
	// persistent waves: as many as the chip holds at the kernel's occupancy (6 per SIMD), each half taking units from a queue
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	int64_t blocks = (max_units + 2 * TW_QCH - 1) / (2 * TW_QCH);
	static const int wg_per_cu = getenv("CHAINDP_TWIN_WG_PER_CU") ? atoi(getenv("CHAINDP_TWIN_WG_PER_CU")) : 24;   // (tuning: read once)
	const int64_t cap = (int64_t)cus * (wg_per_cu >= 1 && wg_per_cu <= 24 ? wg_per_cu : 24);
	if (blocks > cap) blocks = cap;
	if (blocks < 1) blocks = 1;
	blocks = (blocks + 7) & ~(int64_t)7;                           // eight grab counters, workgroup b on counter b mod 8: the same number of halves on each
	if (!d_queue) return hipErrorInvalidValue;
	{
		const hipError_t e = hipMemsetD32Async((hipDeviceptr_t)d_queue, (int)(2 * blocks / 8), 8 * 64, st);   // every counter starts behind its halves' first grabs
		if (e != hipSuccess) return e;
	}
	TwinArgs g;
	g.par = par; g.off = d_off; g.a = (const ulonglong2*)d_a; g.sumq = d_sumq; g.lut = d_lut; g.lut_stride = lut_stride;
	g.units = d_units; g.aux = d_unit_aux; g.counters = d_counters; g.f = d_f; g.p = d_p; g.v = d_v; g.first_child = d_first_child; g.flags = d_flags;
	g.left = d_left; g.left_cnt = d_left_cnt; g.queue = d_queue; g.route = d_route; g.force_left = force_left; g.total = total;
	// diagnostic: CHAINDP_TWIN_STAMP=1 makes the kernel stamp where its waves' time goes (s_memtime: shader-clock ticks) and this
	// function print the averages -- it synchronises, so never set it in a timed run
	static unsigned long long *d_stamp = nullptr;
	const bool stamp = getenv("CHAINDP_TWIN_STAMP") != nullptr;
	if (stamp && !d_stamp && hipMalloc((void**)&d_stamp, (size_t)cap * 96) != hipSuccess) d_stamp = nullptr;
	g.stamp = stamp ? d_stamp : nullptr;
	if (g.stamp) (void)hipMemsetAsync(d_stamp, 0, (size_t)blocks * 96, st);
	{
		hipFuncAttributes fa;                                        // LDS is addressed by raw byte offsets from 0: no static LDS may sit in front
		const void *fn = par.max_dist_y >= par.max_dist_x ? (const void*)k_chain_twin<true> : (const void*)k_chain_twin<false>;
		const hipError_t e = hipFuncGetAttributes(&fa, fn);
		if (e != hipSuccess) return e;
		if (fa.sharedSizeBytes != 0) return hipErrorInvalidConfiguration;
	}
	if (par.max_dist_y >= par.max_dist_x) hipLaunchKernelGGL(k_chain_twin<true>, dim3((unsigned)blocks), dim3(64), TW_LDS_BYTES, st, g);
	else hipLaunchKernelGGL(k_chain_twin<false>, dim3((unsigned)blocks), dim3(64), TW_LDS_BYTES, st, g);
	if (g.stamp) {
		std::vector<unsigned long long> hb((size_t)blocks * 12);
		if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(hb.data(), d_stamp, hb.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
			double tot = 0, svc = 0, flush = 0, unit = 0, nsvc = 0, nfast = 0, nunit = 0, nb = 0, head = 0, take = 0, tail = 0;
			for (int64_t b = 0; b < blocks; ++b) if (hb[(size_t)b * 12 + 7]) {
				const unsigned long long *o = &hb[(size_t)b * 12];
				tot += o[0]; svc += o[1]; flush += o[2]; unit += o[3]; nsvc += o[4]; nfast += o[5]; nunit += o[6]; head += o[8]; take += o[9]; tail += o[10]; ++nb;
			}
			fprintf(stderr, "[twin stamp raw] sums over waves: head %.0f take %.0f tail %.0f unit %.0f n_unit %.0f (build 3: ticks until the record / the first tile / the rest of a unit switch; switches; first tiles requested ahead)\n", head, take, tail, unit, nunit);
			fprintf(stderr, "[twin stamp] %.0f waves, %.0f ticks each: service %.1f%% (%.0f calls, %.0f ticks each: cold state and decisions %.0f, next tile %.0f, flush %.0f, "
			                "unit switch %.0f (%.0f switches, %.0f ticks each), first anchor and cold state back %.0f), passes %.0f (%.0f ticks each, everything else included)\n",
			        nb, tot / nb, 100.0 * svc / tot, nsvc, svc / (nsvc > 0 ? nsvc : 1), head / (nsvc > 0 ? nsvc : 1), take / (nsvc > 0 ? nsvc : 1), flush / (nsvc > 0 ? nsvc : 1),
			        unit / (nsvc > 0 ? nsvc : 1), nunit, unit / (nunit > 0 ? nunit : 1), tail / (nsvc > 0 ? nsvc : 1), nfast, (tot - svc) / (nfast > 0 ? nfast : 1));
		}
	}
	return hipGetLastError();
}

} // namespace chaindp
