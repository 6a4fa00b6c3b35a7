// chaindp_quad.hip -- the chain DP kernel for ordinary long-read units when every read of the batch has the same gap-cost table:
// FOUR units per wave64, a 16-lane DPP row each, every lane evaluating TWO predecessors per pass.
//
// Why (round 3, measured on k_chain_twin): with two units per wave a pass costs 42 vector instructions for
// two anchors, and 26 of them are not per-pair arithmetic but bookkeeping ACROSS the lanes of a scan -- prefix max, counts of
// marked lanes, who writes the running max -- paid once per 32 predecessors; the scalar unit runs at two thirds of its rate on
// the lane masks, LDS takes eight instructions per pass.  Here a lane holds the ring's slots 2P and 2P+1 (one 16-byte read each
// from the x/q and the p/f ring), the 32 predecessors of a scan are the two elements of 16 lanes, the prefix operations are four
// DPP steps inside a row (no step across rows), and a pass does FOUR anchors: ~65 vector, ~30 scalar and 11 LDS instructions
// per pass, i.e. 16 + 8 + 3 per anchor against 21 + 12 + 4.
//
// What it computes is exactly k_chain_twin's fast variant (reference chain.c:246-284 for reads with one segment, not cDNA,
// bw <= 511, every q_span > 0, table of 1 - cost in int8), with the same derivations.  Scan order inside a
// chunk: lanes in ascending order, in a lane the element of the higher slot first.  Because slots come in aligned pairs, the
// chunk of anchor i starts at predecessor i - 1 when i - 1 is odd and one slot EARLIER -- at anchor i itself, which fails the
// window test against itself -- when it is even; a second chunk continues where the first ended.
//
// One table per wave: LDS per unit is what bounds the units in flight (3200 bytes each in k_chain_twin); with the table shared
// (512 bytes per wave instead of per unit) and v[] of the previous tile kept in registers a unit needs 2432 bytes, a wave 10240,
// and a SIMD holds four waves = sixteen units instead of twelve.  The kernel therefore takes a batch only if all its units' reads
// have the same avg_qspan (UnitAux::lutkey; minimap2's own minimizers without homopolymer compression: always); otherwise it
// leaves the batch to k_chain_twin (`route`).
//
// LDS per wave (dynamic segment from byte 0; q = unit of the wave, 0..3):
//   2048 q + 0     XY  [128 slots] 8 B   x.lo+1, qpos+1 (slot = i & 127), written a tile at a time
//   2048 q + 1024  PF  [ 64 slots] 8 B   4*p (unit-relative, -4 = none), f - 1 (slot = i & 63)
//   2048 q + 1536  XQ  [64] 8 B          the current tile's anchors: x.lo, qpos
//   8192 + 384 q   MK  [1 + 64 + 1] 4 B  marks by distance: word -1 (never a real predecessor), words 0..63, sink
//          + 264   SP  [64] 1 B          q_span - 1 of the current tile's anchors
//          + 328   ST  56 B              cold state (QuadCold), what a scan carries into its second chunk
//   9728           LUT [512] int8        the batch's table of 1 - cost
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include "chaindp_kernels.h"
#include "chaindp_wave.h"
#include "chaindp_lanes.h"

namespace chaindp {

#define QD_PF 1024u
#define QD_XQ 1536u
#define QD_AUX 8192u
#define QD_AUX_UNIT 384u
#define QD_MK0 4u                       // word 0 of the marks inside a unit's AUX block (word -1 in front of it)
#define QD_SINK 260u
#define QD_SP 264u
#define QD_ST 328u
#define QD_CARRY 40u                    // inside ST: 4 * max_j, running max - 1, n_skip, second chunks so far
#define QD_LUT 9728u
#define QD_LDS_BYTES 10240u
#define QD_TILE 64
#define QD_QCH 8
#define QD_XYMASK 0x1bf0u               // unit bits (11-12) | ring offset of an aligned slot pair (bit 3 = parity of the top slot, dropped)
#define QD_PFMASK 0x19f0u
#define QD_PFSLOT 0x19f8u
#define QD_G15 0x8000800080008000ull    // lane 15 of every group
#define QD_G0 0x0001000100010001ull     // lane 0 of every group

typedef uint32_t qd_u32x2a4 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ tw_u32x2 qd_ld2x32(uint32_t a)               // two consecutive words at a 4-byte aligned address (ds_read2_b32)
{
	const qd_u32x2a4 t = *TW_LDS(const qd_u32x2a4, a);
	tw_u32x2 r; r.x = t.x; r.y = t.y; return r;
}

// exclusive prefix max over the 16 lanes of each row, floor 0 (scores are >= 0 where they matter)
__device__ __forceinline__ int qd_excl_max16(int v)
{
	v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, true));
	v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(2), 0xf, 0xf, true));
	v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(4), 0xf, 0xf, true));
	v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(8), 0xf, 0xf, true));
	return __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, true);
}
// exclusive prefix sum over the 16 lanes of each row
__device__ __forceinline__ int qd_excl_sum16(int v)
{
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, true);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(2), 0xf, 0xf, true);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(4), 0xf, 0xf, true);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(8), 0xf, 0xf, true);
	return __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, true);
}

// m ? a : b per lane as ONE v_cndmask (the compiler turns a select of a select into divergent branches)
__device__ __forceinline__ uint32_t qd_sel(uint64_t m, uint32_t a, uint32_t b)
{
	uint32_t r = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	asm("v_cndmask_b32_e64 %0, %2, %1, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
#endif
	return r;
}

// per group: all 16 lanes if the group's lane 15 is set in m
__device__ __forceinline__ uint64_t qd_smear15(uint64_t m)
{
	const uint32_t lo = ((TW_UNI((uint32_t)m) >> 15) & 0x00010001u) * 0xffffu, hi = ((TW_UNI((uint32_t)(m >> 32)) >> 15) & 0x00010001u) * 0xffffu;
	return (uint64_t)hi << 32 | lo;
}
// per group: all 16 lanes if any lane of the group is set in m
__device__ __forceinline__ uint64_t qd_smear_any(uint64_t m)
{
	const uint32_t lo = TW_UNI((uint32_t)m), hi = TW_UNI((uint32_t)(m >> 32));
	const uint32_t rl = ((lo & 0xffffu) ? 0xffffu : 0u) | ((lo >> 16) ? 0xffff0000u : 0u);
	const uint32_t rh = ((hi & 0xffffu) ? 0xffffu : 0u) | ((hi >> 16) ? 0xffff0000u : 0u);
	return (uint64_t)rh << 32 | rl;
}
// per group: the highest set lane of m alone, or the group's lane 0 if m has none there (s_flbit gives -1 for 0, and a shift
// only uses the low five bits of its count: 0x80000000 >> 31 = bit 0)
__device__ __forceinline__ uint32_t qd_last_or_lane0_word(uint32_t w)
{
	uint32_t a = 0, b = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	asm("s_and_b32 %0, %2, 0xffff\n\ts_lshr_b32 %1, %2, 16\n\ts_flbit_i32_b32 %0, %0\n\ts_flbit_i32_b32 %1, %1\n\t"
	    "s_lshr_b32 %0, 0x80000000, %0\n\ts_lshr_b32 %1, 0x80000000, %1\n\ts_lshl_b32 %1, %1, 16\n\ts_or_b32 %0, %0, %1"
	    : "=&s"(a), "=&s"(b) : "s"(w));
#endif
	return a;
}
__device__ __forceinline__ uint64_t qd_last_or_lane0(uint64_t m)
{
	return (uint64_t)qd_last_or_lane0_word(TW_UNI((uint32_t)(m >> 32))) << 32 | qd_last_or_lane0_word(TW_UNI((uint32_t)m));
}

struct QuadArgs {
	Params par;
	const ulonglong2 *a;
	const uint16_t *lut;
	int lut_stride;
	const Unit *units;
	const UnitAux *aux;
	const unsigned long long *counters;
	const unsigned int *key_range;      // [0] min, [1] max of the units' table keys (prepass)
	int32_t *f, *p, *v;
	int32_t *first_child;
	uint8_t *flags;
	Unit *left;                         // leftover list for k_chain_units
	unsigned int *left_cnt;
	unsigned int *queue;                // next unit nobody has taken yet
	unsigned int *route;                // set to 1 when this kernel takes the batch (k_chain_twin then leaves it alone)
	int force_left;                     // test switch, as in k_chain_twin
	int64_t total;
};

struct QuadCold {                       // 40 bytes at the unit's ST (8-byte aligned)
	int64_t next;                       // low word: next unit of this group; high word: end of its chunk of the queue
	int64_t base;                       // global index of the unit's first anchor
	uint64_t x_carry;                   // x of the previous tile's last anchor
	int32_t rel0, room, read, tile0;    // unit start relative to its read; anchors the unit may have; its read; current tile's first anchor
};

struct QuadHot {                        // per group, replicated over its 16 lanes
	uint32_t S;                         // unit << 11 | c, c = 8 * (i - 1) mod 1024 kept in [256, 1792): ring offset of the top slot
	uint32_t m4;                        // 4 * (i - 1) + mark base: mark distance base and the scan's tag
	uint32_t pc, pend;                  // XQ entry of the current anchor; end of the tile's entries
	uint32_t ps;                        // SP byte of the current anchor
	uint32_t mka;                       // address of this lane's first mark word in chunk 0 (toggles with the parity of i - 1)
};

template <bool SAMEGAP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_chain_quad(QuadArgs g)
{
	const int lane = threadIdx.x;
	const int q = lane >> 4, l = lane & 15;
	const uint32_t blk = 2048u * (uint32_t)q, aux = QD_AUX + QD_AUX_UNIT * (uint32_t)q;

	// ---- which batches: short units (as k_chain_twin), one table key, parameters inside the 32-bit / int8 forms
	const int64_t n_units = (int64_t)(uint32_t)g.counters[0];
	const int64_t n_single = (int64_t)(g.counters[0] >> 32);
	const bool short_units = (g.total - n_single) <= 512 * n_units;
	const bool params_ok = g.lut != nullptr && !g.par.is_cdna && g.par.n_segs <= 1 && g.par.max_dist_x >= 1 && g.par.max_dist_y >= 0 &&
	                       ((uint64_t)(int64_t)g.par.max_dist_x + 1) * 129ull < (1ull << 31) && g.par.bw + 1 <= 512 && g.force_left != 1;
	const uint32_t wave_key = g.key_range[0];
	if (!short_units || !params_ok || n_units <= 0 || wave_key != g.key_range[1]) return;   // (uniform: every block leaves; k_chain_twin takes over)
	if (blockIdx.x == 0 && lane == 0) *g.route = 1u;

	// ---- per-lane constants (vector registers on purpose: a scalar operand halves a VALU instruction's issue rate)
	uint32_t L16 = (uint32_t)l << 4;
	const uint32_t mkbase = aux + QD_MK0;
	uint32_t c_far = aux + QD_SINK;
	uint32_t c_mka0 = mkbase - 4u + ((uint32_t)l << 3);            // word 2 l - 1: this lane's first mark word when i - 1 is even
	uint32_t c_M = (uint32_t)g.par.max_dist_x;
	const uint32_t c_bw = (uint32_t)g.par.bw;
	const uint32_t c_cbw = c_M - 1u > c_bw ? c_M - 1u - c_bw : 0u;
	const uint32_t mdq = (uint32_t)(g.par.max_dist_x < g.par.max_dist_y ? g.par.max_dist_x : g.par.max_dist_y);
	uint32_t c_dqoff = c_M - mdq;
	int c_ms = g.par.max_skip;
	int c_ms0 = g.par.max_skip > 0 ? g.par.max_skip : 0;
	int c_min = INT_MIN;
	int c_Mout = l == 15 ? g.par.max_dist_x : INT_MAX;             // window test that only the group's last lane can fail
	uint32_t c_lut = QD_LUT;
	uint32_t c_bwl = c_bw + QD_LUT;
	uint32_t c_cbwl = c_cbw - QD_LUT;
	TW_VREG(L16); TW_VREG(c_far); TW_VREG(c_M); TW_VREG(c_ms0); TW_VREG(c_min); TW_VREG(c_Mout); TW_VREG(c_bwl); TW_VREG(c_cbwl); TW_VREG(c_lut);
	if (!SAMEGAP) TW_VREG(c_dqoff);

	const uint64_t maxx = (uint64_t)(int64_t)g.par.max_dist_x;
	QuadHot u;
	u.S = blk | 256u; u.m4 = mkbase; u.pc = blk + QD_XQ; u.pend = u.pc; u.ps = aux + QD_SP; u.mka = c_mka0;
	uint64_t live_m = ~0ull;                                       // groups that still have (or may get) work
	uint64_t contm = 0;                                            // groups that are in their second (= last) chunk
	const uint32_t st_addr = aux + QD_ST;
	if (l == 0) {
		const uint32_t nx = QD_QCH * (4u * blockIdx.x + (uint32_t)q);                   // the group's first chunk of units: dealt statically
		tw_st64(st_addr, nx, nx + QD_QCH); tw_st64(st_addr + 8u, 0u, 0u);
		tw_st64(st_addr + 16u, 0u, 0u); tw_st64(st_addr + 24u, 0u, 0u);
		tw_st64(st_addr + 32u, 0u, (uint32_t)-QD_TILE);
		tw_st64(st_addr + QD_CARRY, 0xfffffffcu, 0u); tw_st64(st_addr + QD_CARRY + 8u, 0u, 0u);
	}
	bool lut_loaded = false;
	// each group's NEXT tile, requested a tile ahead (one anchor per lane), or the first tile of its next unit (pfu = its global start)
	uint64_t nxx0 = 0, nxy0 = 0, nxx1 = 0, nxy1 = 0, nxx2 = 0, nxy2 = 0, nxx3 = 0, nxy3 = 0;
	int32_t pfu0 = -1, pfu1 = -1, pfu2 = -1, pfu3 = -1;
	int vp0 = 0, vp1 = 0, vp2 = 0, vp3 = 0;                        // v | "emitted at its own step" << 31 of each group's PREVIOUS tile (lane = anchor)
	wave_mem_fence();
#define QD_GET4(qs, a0, a1, a2, a3) ((qs) == 0 ? (a0) : (qs) == 1 ? (a1) : (qs) == 2 ? (a2) : (a3))
#define QD_SET4(qs, a0, a1, a2, a3, val) do { if ((qs) == 0) a0 = (val); else if ((qs) == 1) a1 = (val); else if ((qs) == 2) a2 = (val); else a3 = (val); } while (0)

	// One service round for the groups in `svc` (tile exhausted, or no unit yet): one group at a time, by all 64 lanes of the wave,
	// exactly as in k_chain_twin (see there), with v[] of the previous tile in a register instead of an LDS ring.
	auto service = [&](uint64_t svc) {
#if defined(__HIP_DEVICE_COMPILE__)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // first_child stores of earlier tiles before the atomics below (no wait in practice)
#endif
		wave_mem_fence();
		uint64_t retired = 0;
#pragma unroll
		for (int qs = 0; qs < 4; ++qs) {                                    // (unrolled: the per-group registers below are picked by a constant)
			if (((svc >> (16 * qs)) & 1ull) == 0) continue;
			const uint64_t hm = 0xffffull << (16 * qs);
			const uint32_t sblk = 2048u * (uint32_t)qs, saux = QD_AUX + QD_AUX_UNIT * (uint32_t)qs;
			const uint32_t sa = saux + QD_ST, curb = sblk + QD_XQ, spb = saux + QD_SP, mkb = saux + QD_MK0;
			const tw_u32x2 cw0 = tw_ld64(sa), cw1 = tw_ld64(sa + 8u), cw2 = tw_ld64(sa + 16u), cw3 = tw_ld64(sa + 24u), cw4 = tw_ld64(sa + 32u);
			int64_t c_next = (int64_t)((uint64_t)TW_UNI(cw0.y) << 32 | TW_UNI(cw0.x));
			int64_t c_base = (int64_t)((uint64_t)TW_UNI(cw1.y) << 32 | TW_UNI(cw1.x));
			uint64_t c_xcarry = (uint64_t)TW_UNI(cw2.y) << 32 | TW_UNI(cw2.x);
			int c_rel0 = (int)TW_UNI(cw3.x), c_room = (int)TW_UNI(cw3.y), c_read = (int)TW_UNI(cw4.x), c_tile0 = (int)TW_UNI(cw4.y);
			const int cnt_prev = (int)(((uint32_t)__builtin_amdgcn_readlane((int)u.pend, 16 * qs) - curb) >> 3);
			const int slow_h = (int)TW_UNI((uint32_t)tw_ld32(sa + QD_CARRY + 12u));
			const int tile_prev = c_tile0, rel0_prev = c_rel0;
			const int64_t base_prev = c_base;
			bool live = true;
			const uint32_t nx0 = (uint32_t)c_next, ne0 = (uint32_t)((uint64_t)c_next >> 32);
			tw_u32x4 rec0 = {0u, 0u, 0u, 0u}, aux0 = {0u, 0u, 0u, 0u}, rec1 = {0u, 0u, 0u, 0u};
			const bool rec0_ok = nx0 < ne0 && (int64_t)nx0 < n_units;
			const bool rec1_ok = rec0_ok && nx0 + 1u < ne0 && (int64_t)nx0 + 1 < n_units;
			if (rec0_ok) { rec0 = *TW_CONST(tw_u32x4, g.units + nx0); aux0 = *TW_CONST(tw_u32x4, g.aux + nx0); }
			if (rec1_ok) rec1 = *TW_CONST(tw_u32x4, g.units + nx0 + 1u);

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
				const int cnt = stop_m ? __builtin_ctzll(stop_m) : QD_TILE;
				c_xcarry = readlane_u64(an_x, 63);
				u.pc = TW_SEL(hm, curb, u.pc); u.pend = TW_SEL(hm, curb + ((uint32_t)cnt << 3), u.pend); u.ps = TW_SEL(hm, spb, u.ps);
				if (cnt == 0) return 0;
				wave_mem_fence();
				if (lane < cnt) {
					const int sp = span_of_hi((uint32_t)(an_y >> 32));
					tw_st64(sblk + (((uint32_t)i_lane & 127u) << 3), (uint32_t)an_x + 1u, (uint32_t)an_y + 1u);
					tw_st64(curb + ((uint32_t)lane << 3), (uint32_t)an_x, (uint32_t)an_y);
					tw_st8(spb + (uint32_t)lane, sp - 1);
					g.first_child[c_base + i_lane] = NO_CHILD;                // "none" a tile of passes before any atomic lowers it
				}
				wave_mem_fence();
				if (cnt == QD_TILE && c_tile0 + QD_TILE < c_room) {            // the tile after this one (else the registers are for the unit's successor)
					uint64_t rx = 0, ry = 0;
					if (i_lane + QD_TILE < c_room) { const ulonglong2 t = g.a[c_base + i_lane + QD_TILE]; rx = t.x; ry = t.y; }
					QD_SET4(qs, nxx0, nxx1, nxx2, nxx3, rx); QD_SET4(qs, nxy0, nxy1, nxy2, nxy3, ry); QD_SET4(qs, pfu0, pfu1, pfu2, pfu3, -1);
				}
				return cnt;
			};

			// ---- the unit goes on?
			bool goes_on = cnt_prev == QD_TILE && c_tile0 + QD_TILE < c_room;
			if (goes_on && (slow_h * 8 > c_tile0 + QD_TILE || g.force_left == 2)) {
				// a unit that keeps needing second chunks goes to k_chain_units, which resumes behind the tiles flushed here
				if (lane == 0) {
					Unit un; un.start = (int64_t)((uint64_t)c_base | (uint64_t)(uint32_t)(c_tile0 + QD_TILE) << 32); un.read = c_read; un.len = c_room;
					g.left[atomicAdd(g.left_cnt, 1u)] = un;
				}
				goes_on = false;
			}
			if (goes_on) {
				c_tile0 += QD_TILE;
				if (take_tile(QD_GET4(qs, nxx0, nxx1, nxx2, nxx3), QD_GET4(qs, nxy0, nxy1, nxy2, nxy3)) == 0) goes_on = false;
			}
			// ---- flush the finished tile
			if (cnt_prev > 0) {
				const int i_lane = tile_prev + lane;
				const bool have = lane < cnt_prev;
				const int64_t gi = base_prev + i_lane;
				int fi = 0, p4 = -4;
				if (have) {
					const tw_u32x2 pf = tw_ld64(sblk + QD_PF + (((uint32_t)i_lane & 63u) << 3));
					p4 = (int)pf.x; fi = (int)pf.y + 1;                      // (the ring holds f - 1)
				}
				const int pi = p4 >> 2;
				int val = fi, ptr = have ? pi : -1;
				const bool ext = ptr >= 0 && ptr < tile_prev;                // predecessor in the previous tile: its v is final, in that tile's register
				const int vprev = QD_GET4(qs, vp0, vp1, vp2, vp3);
				const int vext = __builtin_amdgcn_ds_bpermute((ext ? ptr - (tile_prev - QD_TILE) : lane) << 2, vprev);
				if (ext) { val = max(val, vext & 0x7fffffff); ptr = -1; }
				const bool ext_self = ext && vext < 0;
				for (int r = 0; r < 6; ++r) {                                // v[i] = max(f[i], v[p[i]]) (chain.c:284) by pointer doubling over the tile
					if (__builtin_amdgcn_ballot_w64(ptr >= tile_prev) == 0) break;
					const int src = (ptr >= tile_prev ? ptr - tile_prev : lane) << 2;
					const int pv = __builtin_amdgcn_ds_bpermute(src, val);
					const int pp = __builtin_amdgcn_ds_bpermute(src, ptr);
					if (ptr >= tile_prev) { val = max(val, pv); ptr = pp; }
				}
				const bool self = val >= g.par.min_sc || pi >= 0;            // emitted at its own step (chain.c:304)
				const int srcp = (pi >= tile_prev ? pi - tile_prev : lane) << 2;
				const int pself_in = __builtin_amdgcn_ds_bpermute(srcp, self ? 1 : 0);
				const bool pred_self = ext ? ext_self : pself_in != 0;
				QD_SET4(qs, vp0, vp1, vp2, vp3, have ? (val | (self ? INT_MIN : 0)) : 0);
				if (have) {
					g.f[gi] = fi;
					g.p[gi] = pi < 0 ? -1 : pi + rel0_prev;
					g.v[gi] = val;
					int maybe_first = 0;
					if (pi >= 0 && !pred_self) { atomicMin(&g.first_child[base_prev + pi], rel0_prev + i_lane); maybe_first = 4; }
					g.flags[gi] = (uint8_t)((self ? 2 : 0) | maybe_first | (val >= g.par.min_sc ? 8 : 0) | (fi < val ? 16 : 0));
				}
			}
			// ---- the unit is over: the group's next unit, its LDS, its first tile
			bool have_rec = rec0_ok;
			const bool rec0_used = !goes_on;
			while (!goes_on && live) {
				uint32_t nx = (uint32_t)c_next, ne = (uint32_t)((uint64_t)c_next >> 32);
				if (nx >= ne) {
					uint32_t q0 = 0;
					if (lane == 0) q0 = atomicAdd(g.queue, (unsigned int)QD_QCH);
					nx = TW_UNI(q0); ne = nx + QD_QCH;
					have_rec = false;
				}
				if ((int64_t)nx >= n_units) {
					c_next = (int64_t)((uint64_t)ne << 32 | nx); live = false;
					u.pc = TW_SEL(hm, curb, u.pc); u.pend = TW_SEL(hm, curb, u.pend);
					break;
				}
				if (!have_rec) { rec0 = *TW_CONST(tw_u32x4, g.units + nx); aux0 = *TW_CONST(tw_u32x4, g.aux + nx); }
				have_rec = false;
				c_next = (int64_t)((uint64_t)ne << 32 | (nx + 1u));
				Unit un;
				un.start = (int64_t)((uint64_t)rec0.y << 32 | rec0.x); un.read = (int32_t)rec0.z; un.len = (int32_t)rec0.w;
				if ((aux0.z & 1u) || aux0.y != wave_key) {                    // not for this kernel (the key test cannot fail: key_range)
					if (lane == 0) g.left[atomicAdd(g.left_cnt, 1u)] = un;
					continue;
				}
				c_base = un.start; c_rel0 = (int)aux0.x; c_room = un.len; c_read = un.read; c_tile0 = 0;
				uint64_t tl_x, tl_y;
				if (QD_GET4(qs, pfu0, pfu1, pfu2, pfu3) == (int32_t)c_base) { tl_x = QD_GET4(qs, nxx0, nxx1, nxx2, nxx3); tl_y = QD_GET4(qs, nxy0, nxy1, nxy2, nxy3); }
				else {
					tl_x = 0; tl_y = 0;
					if (lane < c_room) { const ulonglong2 t = g.a[c_base + lane]; tl_x = t.x; tl_y = t.y; }
				}
				QD_SET4(qs, pfu0, pfu1, pfu2, pfu3, -1);
				wave_mem_fence();
				if (!lut_loaded) {                                            // the batch's table (as bytes), once per wave
					const uint2 *src = (const uint2*)(g.lut + (int64_t)c_read * g.lut_stride);
					for (int k = lane; k * 4 <= g.par.bw; k += 64) {
						const uint2 t = src[k];
						const uint32_t w = (t.x & 0xffu) | (t.x >> 8 & 0xff00u) | (t.y << 16 & 0xff0000u) | (t.y << 8 & 0xff000000u);
						tw_st32(QD_LUT + ((uint32_t)k << 2), (int)w);
					}
					lut_loaded = true;
				}
				const uint32_t x_none = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tl_x, 0) - (uint32_t)maxx - 1u;   // "no anchor here" (x+1 encoding)
				for (int k = lane; k < 128; k += 64) tw_st64(sblk + ((uint32_t)k << 3), x_none, 0u);
				for (int k = lane; k < 66; k += 64) tw_st32(saux + ((uint32_t)k << 2), -1);
				wave_mem_fence();
				c_xcarry = 0;
				if (lane == 0) tw_st32(sa + QD_CARRY + 12u, 0);
				QD_SET4(qs, vp0, vp1, vp2, vp3, 0);
				if (take_tile(tl_x, tl_y) > 0) goes_on = true;
			}
			// ---- the unit ends with the tile just taken: request its successor's first tile now, if the successor is known
			if (live) {
				const uint32_t cnt_now = ((uint32_t)__builtin_amdgcn_readlane((int)u.pend, 16 * qs) - curb) >> 3;
				const uint32_t nxn = (uint32_t)c_next, nen = (uint32_t)((uint64_t)c_next >> 32);
				if (!(cnt_now == QD_TILE && c_tile0 + QD_TILE < c_room) && QD_GET4(qs, pfu0, pfu1, pfu2, pfu3) < 0 && nxn < nen && (int64_t)nxn < n_units &&
				    ((nxn == nx0 && rec0_ok && !rec0_used) || (nxn == nx0 + 1u && rec1_ok))) {
					const tw_u32x4 rn = nxn == nx0 ? rec0 : rec1;
					const int32_t st = (int32_t)rn.x, ln = (int32_t)rn.w;
					uint64_t rx = 0, ry = 0;
					if (lane < ln) { const ulonglong2 t = g.a[(int64_t)st + lane]; rx = t.x; ry = t.y; }
					QD_SET4(qs, nxx0, nxx1, nxx2, nxx3, rx); QD_SET4(qs, nxy0, nxy1, nxy2, nxy3, ry); QD_SET4(qs, pfu0, pfu1, pfu2, pfu3, st);
				}
			}
			// ---- the tile's first anchor becomes current
			if (live) {
				const uint32_t jt = (uint32_t)c_tile0 - 1u;                    // i - 1
				uint32_t c = (jt << 3) & 0x3ffu;
				if (c < 256u) c += 1024u;
				u.S = TW_SEL(hm, sblk | c, u.S); u.m4 = TW_SEL(hm, (jt << 2) + mkb, u.m4);
				u.mka = TW_SEL(hm, c_mka0 | ((jt & 1u) << 2), u.mka);
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

	// What follows a pass in which not every group finished its scan in its first chunk: per group either the next anchor becomes
	// current (D), or the second chunk follows, or -- still undecided after the second chunk -- the unit is handed over.  `nsk`:
	// n_skip behind the group's last element, valid in its lane 15.  Returns the groups to service.
	auto slow_tail = [&](uint64_t D, uint32_t a_cur, int nsk) -> uint64_t {
		const uint64_t giveup = ~D & contm;
		wave_mem_fence();
		const tw_u32x2 curw = tw_ld64(a_cur);                                // the running max just written
		if (__builtin_amdgcn_inverse_ballot_w64(D & live_m)) {               // (an idle group's state stays put: its S carries the unit's LDS block)
			u.m4 += 4u; u.S += 8u; u.pc += 8u; u.ps += 1u; u.mka ^= 4u;
		} else if (l == 15 && __builtin_amdgcn_inverse_ballot_w64(~D)) {     // what the second chunk starts from
			tw_st64(st_addr + QD_CARRY, curw.x, curw.y);
			tw_st32(st_addr + QD_CARRY + 8u, nsk);
			tw_st32(st_addr + QD_CARRY + 12u, tw_ld32(st_addr + QD_CARRY + 12u) + 1);
		}
		contm = ~D & ~giveup & live_m;
		if (__builtin_expect(giveup != 0, 0)) {
			if (__builtin_amdgcn_inverse_ballot_w64(giveup)) {
				wave_mem_fence();
				if (l == 0) {
					const QuadCold c = *TW_LDS(QuadCold, st_addr);
					Unit un; un.start = (int64_t)((uint64_t)c.base | (uint64_t)(uint32_t)c.tile0 << 32); un.read = c.read; un.len = c.room;
					g.left[atomicAdd(g.left_cnt, 1u)] = un;                  // k_chain_units goes on from the tile this scan is in
				}
				u.pend = u.pc = blk + QD_XQ;                                 // an empty tile: nothing to flush, cannot go on: service() picks the next unit
			}
			return giveup;
		}
		return 0;
	};

	bool force_general = false;
	uint64_t svc = ~0ull;                                              // every group starts by picking its first unit
	// ======================================================================================== main loop: one chunk pass per trip
	for (;;) {
		// tile exhausted (or unit handed over, or no unit yet): flush, next tile / unit.  (The one call site of service(): its body is
		// unrolled over the four groups.)
		if (__builtin_expect(svc != 0, 0)) service(svc & live_m);
		if (live_m == 0) break;
		wave_mem_fence();
		svc = 0;
		if (__builtin_expect(contm == 0 && live_m == ~0ull && !force_general, 1)) {
			// ------------------------------------------------------------ every group in its first chunk (n_skip = 0, max_j = none)
			uint64_t X, tile;
			int cnt_incl;
			uint32_t a_cur;
			for (;;) {
				const uint32_t t0 = u.S - L16;                               // lane l <-> slots 2P, 2P+1 with P = ((i - 1) >> 1) - l
				const tw_u32x4 xy = tw_ld128(t0 & QD_XYMASK);                // .xy: slot 2P (scanned second), .zw: slot 2P+1 (scanned first)
				const tw_u32x4 pf = tw_ld128((t0 & QD_PFMASK) + QD_PF);
				const tw_u32x2 cur = tw_ld64(u.pc);                          // the anchor itself: x, q
				const int spm1 = tw_ld_u8(u.ps);                             // ... and q_span - 1
				const uint32_t drh = cur.x - xy.z, dqh = cur.y - xy.w, drl = cur.x - xy.x, dql = cur.y - xy.y;   // differences minus one
				const uint32_t ddh = tw_sad(drh, dqh, c_lut), ddl = tw_sad(drl, dql, c_lut);
				const uint32_t dsh = SAMEGAP ? dqh : __builtin_elementwise_add_sat(dqh, c_dqoff), dsl = SAMEGAP ? dql : __builtin_elementwise_add_sat(dql, c_dqoff);
				const uint64_t okh = TW_ULT(max(max(drh, dsh), ddh + c_cbwl), c_M);   // chain.c:252-260 as one compare
				const uint64_t okl = TW_ULT(max(max(drl, dsl), ddl + c_cbwl), c_M);
				tw_st32(TW_SEL(okh, min(u.m4 - pf.z, c_far), c_far), (int)u.m4);      // chain.c:281: marks by distance, the others to the sink
				tw_st32(TW_SEL(okl, min(u.m4 - pf.x, c_far), c_far), (int)u.m4);
				wave_mem_fence();
				const tw_u32x2 tj = qd_ld2x32(u.mka);                        // this lane's own two mark words
				const int luth = tw_ld_i8(min(ddh, c_bwl)), lutl = tw_ld_i8(min(ddl, c_bwl));
#if defined(__HIP_DEVICE_COMPILE__)
				__builtin_amdgcn_sched_barrier(0);
#endif
				// marked elements (chain.c:277 without its "not better" half) and how many of them lie in the lanes below: known before
				// the scores are, so this DPP chain runs beside the one of the running max instead of behind it
				const uint64_t Mh = TW_EQ(tj.x, u.m4) & okh, Ml = TW_EQ(tj.y, u.m4) & okl;
				const int isMh = (int)__builtin_amdgcn_inverse_ballot_w64(Mh), isMl = (int)__builtin_amdgcn_inverse_ballot_w64(Ml);
				const int cMx = qd_excl_sum16(isMh + isMl);
				cnt_incl = cMx + isMh + isMl;
				const int sch = TW_SEL(okh, min(min((int)dqh, (int)drh), spm1) + (int)pf.w + luth, c_min);   // chain.c:262-273, minus one
				const int scl = TW_SEL(okl, min(min((int)dql, (int)drl), spm1) + (int)pf.y + lutl, c_min);
				const int eh = max(qd_excl_max16(max(sch, scl)), spm1);      // running max in front of the lane's first element
				const uint64_t Ah = TW_SGT(sch, eh);                         // new running max (chain.c:274)
				const uint64_t Al = TW_SGT(scl, max(eh, sch));
				tile = 0; X = 0; a_cur = 0;
				// the general pass redoes the anchor when a marked element is a new maximum (then "marked" is not "marked and not
				// better") or a new maximum comes behind a marked element of its group (the n_skip walk is not a count then): rare
				if (__builtin_expect(((Ah & Mh) | (Al & Ml) | (TW_SGT(cMx, 0) & Ah) | (TW_SGT(cMx + isMh, 0) & Al)) != 0, 0)) { force_general = true; break; }
				// the running max goes to PF[i]: the group's last A element, or (none) its lane 0 writes "no predecessor, q_span - 1"
				a_cur = (u.S + 8u) & QD_PFSLOT;
				{
					const uint32_t wph = u.m4 - u.mka;                       // 4 j of the lane's first element
					uint32_t c_none = 0xfffffffcu;
					TW_VREG(c_none);
					const uint32_t wp = qd_sel(Al, wph - 4u, qd_sel(Ah, wph, c_none));
					const int wf = (int)qd_sel(Al, (uint32_t)scl, qd_sel(Ah, (uint32_t)sch, (uint32_t)spm1));
					// who writes: the group's last lane with a new maximum -- lane 0 when there is none, and (19 scans in 20) when the only
					// new maximum is the nearest predecessor, which sits in lane 0 as well: the sixteen scalar instructions that find the
					// last lane per group run only when some other lane has one
					uint64_t writer = QD_G0;
					if (__builtin_expect(((Ah | Al) & ~QD_G0) != 0, 0)) writer = qd_last_or_lane0(Ah | Al);
					if (__builtin_amdgcn_inverse_ballot_w64(writer)) tw_st64(a_cur + QD_PF, wp, (uint32_t)wf);
				}
				// scan complete: the (max_skip + 1)-th marked element exists, or the group's last element is outside the window
				X = (TW_SGT(cnt_incl, c_ms0) | TW_SGE(drl, c_Mout)) & QD_G15;
				if (__builtin_expect(X != QD_G15, 0)) break;
				u.m4 += 4u; u.S += 8u; u.pc += 8u; u.ps += 1u; u.mka ^= 4u;
				tile = TW_SGE(u.pc, u.pend);
				if (__builtin_expect(tile != 0, 0)) break;
				wave_mem_fence();
			}
			if (!force_general && tile == 0) svc = slow_tail(qd_smear15(X), a_cur + QD_PF, cnt_incl);
		} else {
			// ------------------------------------------------------------ general pass: second chunks, idle groups, interleaved walks
			force_general = false;
			const uint64_t first = ~contm;                                   // groups in their first chunk: running max = q_span - 1, nothing carried
			const uint32_t cg = (u.S & 0x1bffu) | 0x400u;                    // the counter in [1024, 2048): room for the second chunk's offset
			const uint32_t t0 = cg - TW_SEL(first, 0u, 256u) - L16;
			const uint32_t kb4 = TW_SEL(first, 0u, 128u);                    // 4 * 32 c
			const tw_u32x4 xy = tw_ld128(t0 & QD_XYMASK);
			const tw_u32x4 pf = tw_ld128((t0 & QD_PFMASK) + QD_PF);
			const tw_u32x2 cur = tw_ld64(u.pc);
			const int spm1 = tw_ld_u8(u.ps);
			const tw_u32x2 carry = tw_ld64(st_addr + QD_CARRY);
			const int maxf0 = TW_SEL(first, spm1, (int)carry.y);
			const uint32_t maxj4 = TW_SEL(first, 0xfffffffcu, carry.x);
			const int n0 = TW_SEL(first, 0, tw_ld32(st_addr + QD_CARRY + 8u));
			const uint32_t drh = cur.x - xy.z, dqh = cur.y - xy.w, drl = cur.x - xy.x, dql = cur.y - xy.y;
			const uint32_t ddh = tw_sad(drh, dqh, c_lut), ddl = tw_sad(drl, dql, c_lut);
			const uint32_t dsh = SAMEGAP ? dqh : __builtin_elementwise_add_sat(dqh, c_dqoff), dsl = SAMEGAP ? dql : __builtin_elementwise_add_sat(dql, c_dqoff);
			// not evaluated: idle groups, and the last element of a second chunk when i - 1 is odd (j = i - 64 shares its PF slot with anchor i)
			const uint64_t odd = TW_EQ(u.mka & 4u, 4u);
			const uint64_t okh = TW_ULT(max(max(drh, dsh), ddh + c_cbwl), c_M) & live_m;
			const uint64_t okl = TW_ULT(max(max(drl, dsl), ddl + c_cbwl), c_M) & live_m & ~(contm & QD_G15 & odd);
			const int luth = tw_ld_i8(min(ddh, c_bwl)), lutl = tw_ld_i8(min(ddl, c_bwl));
			tw_st32(TW_SEL(okh, min(u.m4 - pf.z, c_far), c_far), (int)u.m4);
			tw_st32(TW_SEL(okl, min(u.m4 - pf.x, c_far), c_far), (int)u.m4);
			wave_mem_fence();
			const tw_u32x2 tj = qd_ld2x32(u.mka + kb4);
			const int sch = TW_SEL(okh, min(min((int)dqh, (int)drh), spm1) + (int)pf.w + luth, c_min);
			const int scl = TW_SEL(okl, min(min((int)dql, (int)drl), spm1) + (int)pf.y + lutl, c_min);
			const int eh = max(qd_excl_max16(max(sch, scl)), maxf0);
			uint64_t Ah = TW_SGT(sch, eh), Al = TW_SGT(scl, max(eh, sch));
			uint64_t Bh = TW_EQ(tj.x, u.m4) & okh & ~Ah, Bl = TW_EQ(tj.y, u.m4) & okl & ~Al;
			const int isAh = (int)__builtin_amdgcn_inverse_ballot_w64(Ah), isAl = (int)__builtin_amdgcn_inverse_ballot_w64(Al);
			const int isBh = (int)__builtin_amdgcn_inverse_ballot_w64(Bh), isBl = (int)__builtin_amdgcn_inverse_ballot_w64(Bl);
			const int cAx = qd_excl_sum16(isAh + isAl), cBx = qd_excl_sum16(isBh + isBl);
			// n_skip behind each element when every A element of the group precedes every B element (chain.c:276,278)
			const int nAh = cAx + isAh, nBh = cBx + isBh, nAl = nAh + isAl, nBl = nBh + isBl;
			int nh = max(n0 - nAh, 0) + nBh, nl = max(n0 - nAl, 0) + nBl;
			uint64_t brk = (TW_SGT(nh, c_ms) & Bh) | (TW_SGT(nl, c_ms) & Bl);
			int nsk = nl;
			const uint64_t inter_g = qd_smear_any((TW_SGT(cBx, 0) & Ah) | (TW_SGT(nBh, 0) & Al));
			if (__builtin_expect(inter_g != 0, 0)) {
				// A and B elements interleave in a group: the walk element by element, on the scalar side (one scan in thousands).  Of the
				// group's A elements only the last one in front of the break stays, of its B elements only the one that breaks.
				for (int gq = 0; gq < 4; ++gq) {
					if (((inter_g >> (16 * gq)) & 1ull) == 0) continue;
					const uint32_t ah = (uint32_t)(Ah >> (16 * gq)) & 0xffffu, al = (uint32_t)(Al >> (16 * gq)) & 0xffffu;
					const uint32_t bh = (uint32_t)(Bh >> (16 * gq)) & 0xffffu, bl = (uint32_t)(Bl >> (16 * gq)) & 0xffffu;
					int n = __builtin_amdgcn_readlane(n0, 16 * gq), last = -1, broke = -1;
					for (int e = 0; e < 32 && broke < 0; ++e) {
						const uint32_t bit = 1u << (e >> 1);
						if (((e & 1) ? al : ah) & bit) { last = e; n = n > 0 ? n - 1 : 0; }
						else if (((e & 1) ? bl : bh) & bit) { if (++n > g.par.max_skip) broke = e; }
					}
					const uint64_t gm = 0xffffull << (16 * gq);
					Ah &= ~gm; Al &= ~gm; brk &= ~gm;
					if (last >= 0) { const uint64_t b = 1ull << (16 * gq + (last >> 1)); if (last & 1) Al |= b; else Ah |= b; }
					if (broke >= 0) brk |= 1ull << (16 * gq + (broke >> 1));
					nsk = TW_SEL(gm, n, nsk);
				}
			}
			const uint32_t a_cur = ((u.S + 8u) & QD_PFSLOT) + QD_PF;         // PF slot of anchor i
			{
				const uint32_t wph = u.m4 - u.mka - kb4;                     // 4 j of the lane's first element
				const uint32_t wp = qd_sel(Al, wph - 4u, qd_sel(Ah, wph, maxj4));
				const int wf = (int)qd_sel(Al, (uint32_t)scl, qd_sel(Ah, (uint32_t)sch, (uint32_t)maxf0));
				if (__builtin_amdgcn_inverse_ballot_w64(qd_last_or_lane0(Ah | Al) & live_m)) tw_st64(a_cur, wp, (uint32_t)wf);
			}
			const uint64_t D = qd_smear_any(brk) | qd_smear15(TW_SGE(drl, c_Mout) & QD_G15) | ~live_m;   // idle groups count as done
			svc = slow_tail(D, a_cur, nsk);
		}
		svc |= qd_smear_any(TW_SGE(u.pc, u.pend) & live_m & ~contm);
	}
}

size_t quad_lds_bytes() { return QD_LDS_BYTES; }

hipError_t launch_chain_quad(hipStream_t st, const Params &par, int64_t max_units, const void *d_a, const uint16_t *d_lut, int lut_stride,
                             const Unit *d_units, const UnitAux *d_unit_aux, const unsigned long long *d_counters, const unsigned int *d_key_range,
                             int32_t *d_f, int32_t *d_p, int32_t *d_v, int32_t *d_first_child, uint8_t *d_flags, Unit *d_left,
                             unsigned int *d_left_cnt, unsigned int *d_queue, unsigned int *d_route, int force_left, int64_t total)
{
	if (max_units <= 0 || !d_unit_aux || !d_key_range) return hipSuccess;
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	// persistent waves: sixteen per CU (10240 bytes of LDS each), every group of sixteen lanes taking units from the queue
	int64_t blocks = (max_units + 4 * QD_QCH - 1) / (4 * QD_QCH);
	const int64_t cap = (int64_t)cus * 16;
	if (blocks > cap) blocks = cap;
	if (blocks < 1) blocks = 1;
	{
		const hipError_t e = hipMemsetD32Async((hipDeviceptr_t)d_queue, (int)(4 * QD_QCH * blocks), 1, st);   // the queue starts behind the first chunks
		if (e != hipSuccess) return e;
	}
	QuadArgs g;
	g.par = par; g.a = (const ulonglong2*)d_a; g.lut = d_lut; g.lut_stride = lut_stride; g.units = d_units; g.aux = d_unit_aux;
	g.counters = d_counters; g.key_range = d_key_range; g.f = d_f; g.p = d_p; g.v = d_v; g.first_child = d_first_child; g.flags = d_flags;
	g.left = d_left; g.left_cnt = d_left_cnt; g.queue = d_queue; g.route = d_route; g.force_left = force_left; g.total = total;
	const void *fn = par.max_dist_y >= par.max_dist_x ? (const void*)k_chain_quad<true> : (const void*)k_chain_quad<false>;
	{
		const hipError_t e = check_no_static_lds(fn);                   // LDS is addressed by raw byte offsets from 0
		if (e != hipSuccess) return e;
	}
	if (par.max_dist_y >= par.max_dist_x) hipLaunchKernelGGL(k_chain_quad<true>, dim3((unsigned)blocks), dim3(64), QD_LDS_BYTES, st, g);
	else hipLaunchKernelGGL(k_chain_quad<false>, dim3((unsigned)blocks), dim3(64), QD_LDS_BYTES, st, g);
	return hipGetLastError();
}

} // namespace chaindp
