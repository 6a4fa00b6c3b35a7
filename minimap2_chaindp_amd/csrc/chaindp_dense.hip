// chaindp_dense.hip -- the chain DP kernel for units whose scans run deep (dense repeats): k_chain_dense.
//
// In a dense repeat the window of chain.c:252 holds hundreds to thousands of predecessors and few of them get marked, so
// the scan of chain.c:253-282 does not end at the (max_skip + 1)-th marked predecessor after ~30 steps but walks on: a
// median of 270 predecessors, one scan in nine beyond 1000, one in a hundred across the whole unit (tens of thousands) on
// the generator's `dense` shape.  k_chain_units serves predecessors older than its LDS ring from HBM/L2 one 64-lane
// chunk at a time, two dependent round trips per chunk (a/f/p, then the marks it keeps in a global array), and 64
// scattered mark stores per chunk that load the L2 as much as the reads do.  It hands a unit that keeps doing that over
// to this kernel (CHAINDP_DEEP_HANDOVER), which computes exactly the same thing (run_unit_fast of chaindp_kernels.hip,
// reference chain.c:246-284 for one-segment, non-cDNA reads) with three differences:
//   * marks by distance as ONE BIT each: bit d of an LDS bitmap = "the predecessor d+1 behind the current anchor is
//     marked" (chain.c:281).  64 K distances are 8 KB, so marks never leave LDS, whatever the
//     depth of the scan: a chunk's "marked" lane mask (chain.c:277) is one 64-bit LDS word, a mark is one ds_or_b32, and the
//     bitmap is cleared after each anchor up to the anchor's index (i / 2048 wave-wide stores).  Only units longer than
//     the bitmap keep marks on predecessors beyond it in the global array;
//   * chunks in groups: what a chunk contributes before the serial semantics (filters, scores, marks, its own prefix
//     max) does not depend on the chunks in front of it, so 4 ring chunks or 8 deep chunks are evaluated side by side --
//     all their loads in flight together, one trip to LDS or L2 per group -- and only the walks over their lane masks
//     (fast_walk: a handful of scalar instructions each) run one after the other.  Chunks behind the break are evaluated
//     for nothing; the marks they write are never read;
//   * a ring of 512 anchors and no per-anchor scalar loads: the occupancy (7 waves per CU at 21 KB) is set by LDS, and
//     a wave has its SIMD nearly to itself, so the groups' instruction-level parallelism is what keeps it busy.
// One wave per unit; as many waves as the chip holds at that LDS size take the units from the hand-over list.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "chaindp_kernels.h"
#include "chaindp_wave.h"
#include "chaindp_fast.h"

namespace chaindp {

#define DN_RING CHAINDP_DENSE_RING
typedef FastLds<DN_RING> DnL;
// LDS (dynamic segment, raw byte offsets from 0): ring entries [0, 8 K), v[] [10 K, 12 K) as FastLds<512> lays them out (the
// tile flush is shared with k_chain_units), then the mark bitmap and the read's table of 1 - cost (int16)
#define DN_BM 12288u
#define DN_SINK (DN_BM + CHAINDP_DENSE_BITCAP / 8u)   // 64 words: where a lane without a mark ORs its zero (a word of its own:
                                                  // LDS atomics of one instruction on ONE address take a turn each)
#define DN_LUT (DN_SINK + 256u)
static_assert(DnL::V_OFF + 4u * DN_RING <= DN_BM, "v[] runs into the bitmap");
#define DN_RING_GROUP 4                 // ring chunks evaluated per trip to LDS
#define DN_DEEP_GROUP 8                 // deep chunks evaluated per trip to L2
static_assert(DN_RING % (64 * DN_RING_GROUP) == 0 && (64 * DN_DEEP_GROUP) % 512 == 0, "groups must tile the ring and the bitmap");

struct DenseArgs {
	Params par;
	const int64_t *off;
	const ulonglong2 *a;
	const unsigned long long *sumq;
	const uint16_t *lut;
	int lut_stride;
	const Unit *units;                  // the hand-over list
	const unsigned long long *count;    // its length (low 32 bits)
	int32_t *f, *p, *v;
	unsigned long long *tg;
	uint32_t epoch;
	int32_t *first_child;
	uint8_t *flags;
	int bitcap;                         // distances the bitmap covers (multiple of 512, <= CHAINDP_DENSE_BITCAP)
};

// what a scan needs besides the unit context: the mark bookkeeping of the current anchor
struct DenseScan {
	uint32_t i1x4;                      // 4 * (i - 1)
	uint32_t cap4;                      // 4 * bitcap
	uint32_t lane4;                     // 4 * lane
	int bitcap;
	bool far_possible;                  // i > bitcap: a mark can fall behind the bitmap
	unsigned long long tag;             // global mark of this anchor (only behind the bitmap)
};

__device__ __forceinline__ void dn_or_b32(uint32_t a, uint32_t v)
{
	(void)__hip_atomic_fetch_or(LDS_PTR(uint32_t, a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// chain.c:281 for one lane: the predecessor's predecessor p (4 * its unit-relative index, negative: none) is marked.  Near
// targets set their bit (idle lanes OR a zero into a word of their own: nothing is masked off); targets behind the
// bitmap, which only exist for anchors beyond it, go to the global array.
__device__ __forceinline__ void dn_mark(const UnitCtx &c, const DenseScan &s, bool ok, uint32_t p4)
{
	const uint32_t d4 = s.i1x4 - p4;
	const bool has = ok && (int)p4 >= 0;
	const bool near = has && d4 < s.cap4;
	dn_or_b32(near ? DN_BM + ((d4 >> 7) << 2) : DN_SINK + (s.lane4), near ? 1u << ((d4 >> 2) & 31u) : 0u);
	if (s.far_possible && has && !near) c.tg[c.base + (p4 >> 2)] = s.tag;
}

// the 64 mark bits of the chunk that starts kb predecessors back, as a lane mask (every lane reads the same word)
__device__ __forceinline__ uint64_t dn_marked(uint32_t kb)
{
	const int2 w = lds_load_b64(DN_BM + (kb >> 3));
	return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(w.y) << 32 | (uint32_t)__builtin_amdgcn_readfirstlane(w.x);
}

// G consecutive ring chunks (kb0, kb0 + 64, ...) of anchor i.  Returns true when the scan is complete.
template <bool SAMEGAP, int G>
__device__ __forceinline__ bool dn_ring_group(const UnitCtx &c, const FastK &k, const DenseScan &s, uint32_t xm1, uint32_t qm1, int spm1,
                                              int i, int kb0, int &max_f, int &max_j, int &n_skip)
{
	int sc[G], excl[G];
	uint32_t drl[G];
	uint64_t okm[G], mk[G];
#pragma unroll
	for (int g = 0; g < G; ++g) {
		const uint32_t S = (uint32_t)(i - 1 - kb0 - 64 * g) << 4;
		const FastPairs P = fast_filters<DN_RING, SAMEGAP>(k, (S - k.L4) & (DnL::RB - 1u), xm1, qm1);
		const int dqm1 = (int)P.e.y, drm1 = (int)P.drm1;
		int sc0 = dqm1 < drm1 ? dqm1 : drm1;
		sc0 = sc0 < spm1 ? sc0 : spm1;                                                  // chain.c:262-263, minus one
		const uint32_t di = P.dd < k.bw ? P.dd : k.bw;
		const int scu = sc0 + (int)P.e.z + lds_load_i16(DN_LUT + 2u * di);               // chain.c:272-273 via the table
		okm[g] = __builtin_amdgcn_ballot_w64(P.ok);
		sc[g] = __builtin_amdgcn_inverse_ballot_w64(okm[g]) ? scu : INT_MIN;
		drl[g] = P.drm1;
		dn_mark(c, s, P.ok, P.e.w);
	}
	wave_mem_fence();
#pragma unroll
	for (int g = 0; g < G; ++g) {
		mk[g] = dn_marked((uint32_t)(kb0 + 64 * g));
		excl[g] = wave_excl_max_floor0(sc[g]);
	}
#pragma unroll
	for (int g = 0; g < G; ++g) {
		FastMasks m;
		const int e = excl[g] > max_f ? excl[g] : max_f;
		m.sc = sc[g]; m.drm1 = drl[g];
		m.A = __builtin_amdgcn_ballot_w64(sc[g] > e);                                   // new running max (chain.c:274)
		m.B = okm[g] & ~m.A & mk[g];                                                    // marked and not better (chain.c:277)
		if (fast_walk(k, m, i - 1 - kb0 - 64 * g, max_f, max_j, n_skip)) return true;
		if ((uint32_t)__builtin_amdgcn_readlane((int)drl[g], 63) + 1u > k.M) return true;   // window exhausted (x is sorted)
		if (kb0 + 64 * (g + 1) >= i) return true;                                       // the unit starts here
	}
	return false;
}

// G consecutive deep chunks (predecessors older than the ring: a, f, p come back from HBM/L2) of anchor i.  Only the window
// test is done in 64 bits (x_i - x_j of a predecessor this old may exceed 32 bits; for a lane inside the window it does
// not, and every other difference is bounded by the window).  Returns true when the scan is complete.
template <bool SAMEGAP, int G>
__device__ __forceinline__ bool dn_deep_group(const UnitCtx &c, const FastK &k, const DenseScan &s, uint64_t xi, uint32_t qi, int spm1,
                                              int i, int kb0, int &max_f, int &max_j, int &n_skip)
{
	ulonglong2 aj[G];
	int fj[G], pjr[G], sc[G], excl[G];
	uint64_t okm[G], livem[G], mk[G];
	wave_global_fence();                                                                // f/p of earlier tiles
#pragma unroll
	for (int g = 0; g < G; ++g) {
		const int j = i - 1 - kb0 - 64 * g - c.lane;
		const int64_t gj = c.base + (j >= 0 ? j : 0);
		aj[g] = c.a[gj]; fj[g] = c.f[gj]; pjr[g] = c.p[gj];                             // (p is stored read-relative)
	}
#pragma unroll
	for (int g = 0; g < G; ++g) {
		const bool inr = i - 1 - kb0 - 64 * g - c.lane >= 0;
		const bool live = inr && xi - aj[g].x <= c.maxx;                                // chain.c:252
		const uint32_t drm1 = (uint32_t)xi - (uint32_t)aj[g].x - 1u, dqm1 = qi - (uint32_t)aj[g].y - 1u;
		const uint32_t dd = absdiff_u32(drm1, dqm1);
		const uint32_t dqs = SAMEGAP ? dqm1 : __builtin_elementwise_add_sat(dqm1, k.dq_off);
		const uint32_t m2 = drm1 > dqs ? drm1 : dqs, t = dd + k.cbw;
		const bool ok = live && (m2 > t ? m2 : t) < k.M;                                // chain.c:257-260 (one compare, as fast_filters)
		int sc0 = (int)dqm1 < (int)drm1 ? (int)dqm1 : (int)drm1;
		sc0 = sc0 < spm1 ? sc0 : spm1;
		const uint32_t di = dd < k.bw ? dd : k.bw;
		const int scu = sc0 + fj[g] + lds_load_i16(DN_LUT + 2u * di);
		okm[g] = __builtin_amdgcn_ballot_w64(ok);
		livem[g] = __builtin_amdgcn_ballot_w64(live);
		sc[g] = __builtin_amdgcn_inverse_ballot_w64(okm[g]) ? scu : INT_MIN;
		dn_mark(c, s, ok, pjr[g] >= 0 ? (uint32_t)(pjr[g] - c.rel0) << 2 : 0xfffffffcu);
	}
	if (kb0 >= s.bitcap) {
		// behind the bitmap (a unit longer than it): the group's marks are in the global array -- a second trip to L2
		wave_global_fence();
		unsigned long long tgv[G];
#pragma unroll
		for (int g = 0; g < G; ++g) {
			const int j = i - 1 - kb0 - 64 * g - c.lane;
			tgv[g] = c.tg[c.base + (j >= 0 ? j : 0)];
		}
#pragma unroll
		for (int g = 0; g < G; ++g) mk[g] = __builtin_amdgcn_ballot_w64(tgv[g] == s.tag);
	} else {
		wave_mem_fence();
#pragma unroll
		for (int g = 0; g < G; ++g) mk[g] = dn_marked((uint32_t)(kb0 + 64 * g));
	}
#pragma unroll
	for (int g = 0; g < G; ++g) excl[g] = wave_excl_max_floor0(sc[g]);
#pragma unroll
	for (int g = 0; g < G; ++g) {
		FastMasks m;
		const int e = excl[g] > max_f ? excl[g] : max_f;
		m.sc = sc[g]; m.drm1 = 0;
		m.A = __builtin_amdgcn_ballot_w64(sc[g] > e);
		m.B = okm[g] & ~m.A & mk[g];                                                    // (okm implies j >= 0)
		if (fast_walk(k, m, i - 1 - kb0 - 64 * g, max_f, max_j, n_skip)) return true;
		if (livem[g] != ~0ull) return true;                         // a lane outside the window (or the unit): nothing older can matter
	}
	return false;
}

template <bool SAMEGAP>
__device__ __forceinline__ void run_unit_dense(const UnitCtx &c, int64_t room, int bitcap, unsigned long long tag_hi)
{
	constexpr int MASK = DN_RING - 1;
	const int lane = c.lane;
	FastK k;
	k.L4 = (uint32_t)lane << 4;
	k.far4 = 0; k.trel = 0;
	k.M = (uint32_t)c.maxx;
	k.bw = (uint32_t)c.bw;
	k.cbw = k.M - 1u > k.bw ? k.M - 1u - k.bw : 0u;
	k.dq_off = k.M - (uint32_t)c.mdq;
	k.max_skip = c.max_skip;
	k.ms0 = c.max_skip > 0 ? c.max_skip : 0;
	DenseScan s;
	s.cap4 = (uint32_t)bitcap << 2; s.bitcap = bitcap; s.lane4 = (uint32_t)lane << 2;
	uint64_t x_carry = 0;
	for (int tile0 = 0;; tile0 += 64) {
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
			int max_f = span, max_j = -1, n_skip = 0;
			s.i1x4 = (uint32_t)(i - 1) << 2;
			s.far_possible = i > bitcap;
			s.tag = tag_hi | (uint32_t)i;
			// ring chunks (the unit's first anchors find unwritten slots there: they fail the window test), then deep chunks
			bool done = false;
			for (int kb0 = 0; kb0 < DN_RING && !done; kb0 += 64 * DN_RING_GROUP)
				done = dn_ring_group<SAMEGAP, DN_RING_GROUP>(c, k, s, (uint32_t)xi, qi, span - 1, i, kb0, max_f, max_j, n_skip);
			for (int kb0 = DN_RING; !done; kb0 += 64 * DN_DEEP_GROUP)
				done = dn_deep_group<SAMEGAP, DN_DEEP_GROUP>(c, k, s, xi, qi, span - 1, i, kb0, max_f, max_j, n_skip);
			// anchor i enters the ring (chain.c:283); its marks are wiped: no distance beyond i - 1 can have been set
			wave_mem_fence();
			if (lane == ii) lds_store_b128(waddr, make_uint4((uint32_t)an.x + 1u, (uint32_t)an.y + 1u, (uint32_t)max_f, (uint32_t)(max_j << 2)));
			const uint32_t n_dw = (uint32_t)((i < bitcap ? i : bitcap) + 31) >> 5;
			for (uint32_t o = (uint32_t)lane; o < n_dw; o += 64u) lds_store_b32(DN_BM + (o << 2), 0);
			wave_mem_fence();
		}
		fast_flush_tile<DN_RING>(c, tile0, cnt, waddr, gi);             // v (chain.c:284), f/p/v, the compaction helpers
		if (cnt < 64) break;
	}
}

template <bool SAMEGAP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 4))) void k_chain_dense(DenseArgs g)
{
	UnitCtx c;
	c.a = g.a; c.f = g.f; c.p = g.p; c.v = g.v; c.tg = g.tg; c.tg_hi = 0; c.first_child = g.first_child; c.flags = g.flags; c.min_sc = g.par.min_sc;
	c.s_w = nullptr; c.s_t = nullptr; c.s_v = nullptr; c.s_xhi = nullptr; c.s_yhi = nullptr; c.s_lut = nullptr; c.s_dummy = nullptr;
	c.deep_list = nullptr; c.deep_cnt = nullptr; c.deep_n = 0;
	c.lane = threadIdx.x;
	c.maxx = (uint64_t)(int64_t)g.par.max_dist_x;
	c.mdx = g.par.max_dist_x; c.mdy = g.par.max_dist_y; c.bw = g.par.bw; c.max_skip = g.par.max_skip; c.is_cdna = 0;
	c.mdq = g.par.max_dist_x < g.par.max_dist_y ? g.par.max_dist_x : g.par.max_dist_y;
	c.avgd = 0; c.seg_rule = false;
	// global marks of this kernel carry bit 31: the launch that handed the unit over left marks of the same run epoch behind
	const unsigned long long tag_hi = (unsigned long long)g.epoch << 32 | 0x80000000ull;
	const int lane = threadIdx.x;
	const int64_t n_units = (int64_t)(uint32_t)g.count[0];
	for (int64_t ub = blockIdx.x; ub < n_units; ub += gridDim.x) {
		const Unit u = g.units[ub];
		c.base = u.start; c.read = u.read;
		c.rel0 = (int)(u.start - g.off[u.read]);
		wave_mem_fence();
		{
			const uint4 *src = (const uint4*)(g.lut + (int64_t)u.read * g.lut_stride);   // lut_stride is a multiple of 8 entries (16 B)
			for (int q = lane; q * 8 < g.lut_stride; q += 64) lds_store_b128(DN_LUT + ((uint32_t)q << 4), src[q]);
			const uint32_t x_none = (uint32_t)g.a[u.start].x - (uint32_t)c.maxx - 1u;    // "no anchor here yet" (x+1 encoding): fails the window test
			for (int q = lane; q < DN_RING; q += 64) lds_store_b128((uint32_t)q << 4, make_uint4(x_none, 0u, 0u, 0xfffffffcu));
			for (uint32_t o = (uint32_t)lane << 4; o < CHAINDP_DENSE_BITCAP / 8u; o += 1024u) lds_store_b128(DN_BM + o, make_uint4(0u, 0u, 0u, 0u));
		}
		wave_mem_fence();
		run_unit_dense<SAMEGAP>(c, (int64_t)u.len, g.bitcap, tag_hi);
	}
}

size_t dense_lds_bytes(int lut_stride) { return (size_t)DN_LUT + (size_t)lut_stride * 2; }

hipError_t launch_chain_dense(hipStream_t st, const Params &par, int64_t max_units, const int64_t *d_off, const void *d_a,
                              const unsigned long long *d_sumq, const uint16_t *d_lut, int lut_stride,
                              const Unit *d_deep, const unsigned long long *d_deep_cnt,
                              int32_t *d_f, int32_t *d_p, int32_t *d_v, unsigned long long *d_tg, uint32_t epoch, int32_t *d_first_child, uint8_t *d_flags,
                              int bitcap)
{
	if (max_units <= 0 || !d_lut) return hipSuccess;
	if (bitcap < 512 || bitcap > CHAINDP_DENSE_BITCAP || bitcap % 512) return hipErrorInvalidValue;
	const size_t lds = dense_lds_bytes(lut_stride);
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	// the count is only known on the device: as many waves as the chip holds at this LDS size, each taking units in turn
	int64_t per_cu = (int64_t)(160 * 1024 / lds);
	if (per_cu > 16) per_cu = 16;
	if (per_cu < 1) per_cu = 1;
	int64_t blocks = (int64_t)cus * per_cu;
	if (blocks > max_units) blocks = max_units;
	const void *fn = par.max_dist_y >= par.max_dist_x ? (const void*)k_chain_dense<true> : (const void*)k_chain_dense<false>;
	{
		hipFuncAttributes fa;                                        // LDS is addressed by raw byte offsets from 0: no static LDS may sit in front
		const hipError_t e = hipFuncGetAttributes(&fa, fn);
		if (e != hipSuccess) return e;
		if (fa.sharedSizeBytes != 0) return hipErrorInvalidConfiguration;
	}
	if (lds > 64 * 1024) {
		const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return e;
	}
	DenseArgs g;
	g.par = par; g.off = d_off; g.a = (const ulonglong2*)d_a; g.sumq = d_sumq; g.lut = d_lut; g.lut_stride = lut_stride;
	g.units = d_deep; g.count = d_deep_cnt; g.f = d_f; g.p = d_p; g.v = d_v; g.tg = d_tg; g.epoch = epoch;
	g.first_child = d_first_child; g.flags = d_flags; g.bitcap = bitcap;
	if (par.max_dist_y >= par.max_dist_x) hipLaunchKernelGGL(k_chain_dense<true>, dim3((unsigned)blocks), dim3(64), lds, st, g);
	else hipLaunchKernelGGL(k_chain_dense<false>, dim3((unsigned)blocks), dim3(64), lds, st, g);
	return hipGetLastError();
}

} // namespace chaindp
