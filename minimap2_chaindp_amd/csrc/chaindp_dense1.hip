// chaindp_dense1.hip -- k_chain_dense1: dense-repeat units when the WHOLE batch is dense -- one wave per unit, many waves per CU.
//
// k_chain_dense (chaindp_dense.hip) puts eight waves on a unit: right for a batch's tail, where a handful of long units would
// otherwise each crawl on a SIMD of their own.  A batch with thousands of such units has no tail -- every SIMD is busy to the
// end -- and is bound by what a pair evaluation costs.  There k_chain_units loses to its own memory traffic: predecessors
// older than its LDS ring come back from HBM/L2 64 at a time with two dependent round trips each, and every chunk scatters
// up to 64 eight-byte marks into a global array, which loads the L2's request ports as much as all its reads together
// (~10 cycles per instruction and SIMD at eight waves per SIMD).  This kernel keeps one wave per unit, as k_chain_units
// does, and takes from k_chain_dense what removes that traffic:
//   * marks by distance as one bit each in LDS (derivation 11), sized for the units of the launch --
//     a 32 K-anchor unit needs 4 KB -- so that a wave's LDS footprint stays small (ring of 128 anchors: 8.5 KB in all,
//     18 waves per CU) and no mark ever goes to memory;
//   * chunks in groups: the two ring chunks, then the deep chunks two at a time, are evaluated side by side -- loads in
//     flight together, one trip to L2 per group instead of two per chunk -- and only the walks over their lane masks
//     (fast_walk) run one after the other.
// It computes exactly what run_unit_fast of chaindp_kernels.hip computes (reference chain.c:246-284).  Which of the two dense
// kernels takes the handed-over units is decided on the device from the number of long units in the batch (the prepass'
// length classes): both are launched, one of them finds nothing to do.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "chaindp_kernels.h"
#include "chaindp_wave.h"
#include "chaindp_fast.h"

namespace chaindp {

#define D1_RING 128
typedef FastLds<D1_RING> D1L;
// LDS per wave (dynamic segment, raw byte offsets from 0): ring entries [0, 2 K) and v[] [2.5 K, 3 K) as FastLds<128> lays them
// out (the tile flush is shared with k_chain_units), the mark bitmap (bm_bytes, a launch parameter), a word per lane where a
// lane without a mark ORs its zero, the read's table of 1 - cost (int16)
#define D1_BM 3072u
#define D1_RING_GROUP 2                 // = the ring
#ifndef D1_WAVES_MAX
#define D1_WAVES_MAX 6                  // waves per SIMD the register budget is cut for (LDS allows 4.5)
#endif
#ifndef D1_DEEP_GROUP
#define D1_DEEP_GROUP 2                 // deep chunks evaluated per trip to L2 (8000 dense units: 1 -> 716 ms, 2 -> 551, 3 -> 602, 4 -> 600, 8 -> 679;
                                        // with the next group's loads requested a group ahead 590: the wave waits for LDS, not for L2)
#endif
static_assert(D1L::V_OFF + 4u * D1_RING <= D1_BM && 64 * D1_RING_GROUP == D1_RING, "LDS layout");

struct Dense1Args {
	Params par;
	const int64_t *off;
	const ulonglong2 *a;
	const uint16_t *lut;
	int lut_stride;
	const Unit *units;                  // the hand-over list
	const unsigned long long *count;    // its length (low 32 bits)
	const unsigned int *long_units;     // see dense_all(), chaindp_fast.h
	int route;
	int32_t *f, *p, *v;
	int32_t *first_child;
	uint8_t *flags;
	unsigned int *queue;                // next list entry to look at (zero when the launch starts)
	int min_len, max_len;               // this launch takes the units with min_len < len <= max_len (max_len = 8 * bm_bytes)
	uint32_t bm_bytes;
};

struct Dense1Lds { uint32_t sink, lut; };     // this lane's sink word; the cost table

__device__ __forceinline__ void d1_or_b32(uint32_t a, uint32_t v)
{
	(void)__hip_atomic_fetch_or(LDS_PTR(uint32_t, a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// chain.c:281 for one lane: the predecessor's predecessor p (4 * its unit-relative index, negative: none) is marked: bit
// i - 1 - p of the bitmap (i1x4 = 4 (i - 1)).  A lane without a mark ORs a zero into its own sink word.
__device__ __forceinline__ void d1_mark(uint32_t sink, uint32_t i1x4, bool ok, uint32_t p4)
{
	const uint32_t d4 = i1x4 - p4;
	const bool has = ok && (int)p4 >= 0;
	d1_or_b32(has ? D1_BM + ((d4 >> 7) << 2) : sink, has ? 1u << ((d4 >> 2) & 31u) : 0u);
}

// the 64 mark bits of the chunk that starts kb predecessors back, as a lane mask (every lane reads the same word)
__device__ __forceinline__ uint64_t d1_marked(uint32_t kb)
{
	const int2 w = lds_load_b64(D1_BM + (kb >> 3));
	return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(w.y) << 32 | (uint32_t)__builtin_amdgcn_readfirstlane(w.x);
}

// The ring (chunks 0 and 1) of anchor i.  Returns true when the scan is complete.
template <bool SAMEGAP>
__device__ __forceinline__ bool d1_ring_group(const FastK &k, const Dense1Lds &l, uint32_t xm1, uint32_t qm1, int spm1, int i,
                                              int &max_f, int &max_j, int &n_skip)
{
	constexpr int G = D1_RING_GROUP;
	int sc[G], excl[G];
	uint32_t drl[G];
	uint64_t okm[G], mk[G];
#pragma unroll
	for (int g = 0; g < G; ++g) {
		const uint32_t S = (uint32_t)(i - 1 - 64 * g) << 4;
		const FastPairs P = fast_filters<D1_RING, SAMEGAP>(k, (S - k.L4) & (D1L::RB - 1u), xm1, qm1);
		const int dqm1 = (int)P.e.y, drm1 = (int)P.drm1;
		int sc0 = dqm1 < drm1 ? dqm1 : drm1;
		sc0 = sc0 < spm1 ? sc0 : spm1;                                                  // chain.c:262-263, minus one
		const uint32_t di = P.dd < k.bw ? P.dd : k.bw;
		const int scu = sc0 + (int)P.e.z + lds_load_i16(l.lut + 2u * di);                // chain.c:272-273 via the table
		okm[g] = __builtin_amdgcn_ballot_w64(P.ok);
		sc[g] = __builtin_amdgcn_inverse_ballot_w64(okm[g]) ? scu : INT_MIN;
		drl[g] = P.drm1;
		d1_mark(l.sink, (uint32_t)(i - 1) << 2, P.ok, P.e.w);
	}
	wave_mem_fence();
#pragma unroll
	for (int g = 0; g < G; ++g) {
		mk[g] = d1_marked((uint32_t)(64 * g));
		excl[g] = wave_excl_max_floor0(sc[g]);
	}
#pragma unroll
	for (int g = 0; g < G; ++g) {
		FastMasks m;
		const int e = excl[g] > max_f ? excl[g] : max_f;
		m.sc = sc[g]; m.drm1 = drl[g];
		m.A = __builtin_amdgcn_ballot_w64(sc[g] > e);                                   // new running max (chain.c:274)
		m.B = okm[g] & ~m.A & mk[g];                                                    // marked and not better (chain.c:277)
		if (fast_walk(k, m, i - 1 - 64 * g, max_f, max_j, n_skip)) return true;
		if ((uint32_t)__builtin_amdgcn_readlane((int)drl[g], 63) + 1u > k.M) return true;   // window exhausted (x is sorted)
		if (64 * (g + 1) >= i) return true;                                             // the unit starts here
	}
	return false;
}

// D1_DEEP_GROUP consecutive deep chunks (predecessors older than the ring: a, f, p come back from HBM/L2) of anchor i.  Only the window
// test is done in 64 bits (x_i - x_j of a predecessor this old may exceed 32 bits; for a lane inside the window it does not,
// and every other difference is bounded by the window).  Returns true when the scan is complete.
template <bool SAMEGAP>
__device__ __forceinline__ bool d1_deep_group(const UnitCtx &c, const FastK &k, const Dense1Lds &l, uint64_t xi, uint32_t qi, int spm1,
                                              int i, int kb0, int &max_f, int &max_j, int &n_skip)
{
	constexpr int G = D1_DEEP_GROUP;
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
		const int scu = sc0 + fj[g] + lds_load_i16(l.lut + 2u * di);
		okm[g] = __builtin_amdgcn_ballot_w64(ok);
		livem[g] = __builtin_amdgcn_ballot_w64(live);
		sc[g] = __builtin_amdgcn_inverse_ballot_w64(okm[g]) ? scu : INT_MIN;
		d1_mark(l.sink, (uint32_t)(i - 1) << 2, ok, pjr[g] >= 0 ? (uint32_t)(pjr[g] - c.rel0) << 2 : 0xfffffffcu);
	}
	wave_mem_fence();
#pragma unroll
	for (int g = 0; g < G; ++g) {
		mk[g] = d1_marked((uint32_t)(kb0 + 64 * g));
		excl[g] = wave_excl_max_floor0(sc[g]);
	}
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
__device__ __forceinline__ void run_unit_dense1(const UnitCtx &c, const Dense1Lds &l, int64_t room)
{
	constexpr int MASK = D1_RING - 1;
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
			// the ring (the unit's first anchors find unwritten slots there: they fail the window test), then deep chunks
			bool done = d1_ring_group<SAMEGAP>(k, l, (uint32_t)xi, qi, span - 1, i, max_f, max_j, n_skip);
			for (int kb0 = D1_RING; !done; kb0 += 64 * D1_DEEP_GROUP)
				done = d1_deep_group<SAMEGAP>(c, k, l, xi, qi, span - 1, i, kb0, max_f, max_j, n_skip);
			// anchor i enters the ring (chain.c:283); its marks are wiped: no distance beyond i - 1 can have been set
			wave_mem_fence();
			if (lane == ii) lds_store_b128(waddr, make_uint4((uint32_t)an.x + 1u, (uint32_t)an.y + 1u, (uint32_t)max_f, (uint32_t)(max_j << 2)));
			const uint32_t n_b = (uint32_t)(i + 31) >> 5 << 2;
			for (uint32_t o = (uint32_t)lane << 4; o < n_b; o += 1024u) lds_store_b128(D1_BM + o, make_uint4(0u, 0u, 0u, 0u));
			wave_mem_fence();
		}
		fast_flush_tile<D1_RING>(c, tile0, cnt, waddr, gi);             // v (chain.c:284), f/p/v, the compaction helpers
		if (cnt < 64) break;
	}
}

template <bool SAMEGAP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, D1_WAVES_MAX))) void k_chain_dense1(Dense1Args g)
{
	if (!dense_all(g.long_units, g.route)) return;                  // a batch with a tail: k_chain_dense has the units
	UnitCtx c;
	c.a = g.a; c.f = g.f; c.p = g.p; c.v = g.v; c.tg = nullptr; c.tg_hi = 0; c.first_child = g.first_child; c.flags = g.flags; c.min_sc = g.par.min_sc;
	c.s_w = nullptr; c.s_t = nullptr; c.s_v = nullptr; c.s_xhi = nullptr; c.s_yhi = nullptr; c.s_lut = nullptr; c.s_dummy = nullptr;
	c.deep_list = nullptr; c.deep_cnt = nullptr; c.deep_cap = 0; c.deep_min = 0; c.deep_left = 0; c.deep_n = 0;
	c.lane = threadIdx.x;
	c.maxx = (uint64_t)(int64_t)g.par.max_dist_x;
	c.mdx = g.par.max_dist_x; c.mdy = g.par.max_dist_y; c.bw = g.par.bw; c.max_skip = g.par.max_skip; c.is_cdna = 0;
	c.mdq = g.par.max_dist_x < g.par.max_dist_y ? g.par.max_dist_x : g.par.max_dist_y;
	c.avgd = 0; c.seg_rule = false;
	const int lane = threadIdx.x;
	Dense1Lds l;
	l.sink = D1_BM + g.bm_bytes + ((uint32_t)lane << 2);
	l.lut = D1_BM + g.bm_bytes + 256u;
	const int64_t n_units = (int64_t)(uint32_t)g.count[0];
	for (;;) {
		// the next entry of the list nobody has taken (a unit is 15-40 k serial anchors: dealt in turns, a wave with two long ones
		// would be the launch's tail)
		unsigned int nx = 0;
		if (lane == 0) nx = atomicAdd(g.queue, 1u);
		const int64_t ub = (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)nx);
		if (ub >= n_units) break;
		const Unit u = g.units[ub];
		if (u.len <= g.min_len || u.len > g.max_len) continue;      // another launch's (uniform: one unit per wave)
		c.base = u.start; c.read = u.read;
		c.rel0 = (int)(u.start - g.off[u.read]);
		wave_mem_fence();
		{
			const uint4 *src = (const uint4*)(g.lut + (int64_t)u.read * g.lut_stride);   // lut_stride is a multiple of 8 entries (16 B)
			for (int q = lane; q * 8 < g.lut_stride; q += 64) lds_store_b128(l.lut + ((uint32_t)q << 4), src[q]);
			const uint32_t x_none = (uint32_t)g.a[u.start].x - (uint32_t)c.maxx - 1u;    // "no anchor here yet" (x+1 encoding): fails the window test
			for (int q = lane; q < D1_RING; q += 64) lds_store_b128((uint32_t)q << 4, make_uint4(x_none, 0u, 0u, 0xfffffffcu));
			for (uint32_t o = (uint32_t)lane << 4; o < g.bm_bytes; o += 1024u) lds_store_b128(D1_BM + o, make_uint4(0u, 0u, 0u, 0u));
		}
		wave_mem_fence();
		run_unit_dense1<SAMEGAP>(c, l, (int64_t)u.len);
	}
}

hipError_t launch_chain_dense1(hipStream_t st, const Params &par, int64_t max_units, const int64_t *d_off, const void *d_a,
                               const uint16_t *d_lut, int lut_stride, const Unit *d_deep, const unsigned long long *d_deep_cnt,
                               const unsigned int *d_long_units, int deep_route, unsigned int *d_queues,
                               int32_t *d_f, int32_t *d_p, int32_t *d_v, int32_t *d_first_child, uint8_t *d_flags)
{
	if (max_units <= 0 || !d_lut) return hipSuccess;
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	const void *fn = par.max_dist_y >= par.max_dist_x ? (const void*)k_chain_dense1<true> : (const void*)k_chain_dense1<false>;
	{
		const hipError_t e = check_no_static_lds(fn);        // LDS is addressed by raw byte offsets from 0
		if (e != hipSuccess) return e;
	}
	Dense1Args g;
	g.par = par; g.off = d_off; g.a = (const ulonglong2*)d_a; g.lut = d_lut; g.lut_stride = lut_stride;
	g.units = d_deep; g.count = d_deep_cnt; g.long_units = d_long_units; g.route = deep_route; g.f = d_f; g.p = d_p; g.v = d_v; g.first_child = d_first_child; g.flags = d_flags;
	// two launches: units of up to 32 K anchors with a 4 KB bitmap (most waves per CU), the longer ones with 8 KB; the count is
	// only known on the device, so each launch has as many waves as the chip holds at its LDS size, taking units in turn
	const int caps[2] = {CHAINDP_DENSE_BITCAP / 2, CHAINDP_DENSE_BITCAP};
	for (int q = 0; q < 2; ++q) {
		g.min_len = q ? caps[q - 1] : 0; g.max_len = caps[q]; g.bm_bytes = (uint32_t)caps[q] / 8u; g.queue = d_queues + q;
		const size_t lds = (size_t)D1_BM + g.bm_bytes + 256u + (size_t)lut_stride * 2;
		int64_t per_cu = (int64_t)(160 * 1024 / lds);
		if (per_cu > 24) per_cu = 24;
		if (per_cu < 1) per_cu = 1;
		int64_t blocks = (int64_t)cus * per_cu;
		if (blocks > max_units) blocks = max_units;
		if (par.max_dist_y >= par.max_dist_x) hipLaunchKernelGGL(k_chain_dense1<true>, dim3((unsigned)blocks), dim3(64), lds, st, g);
		else hipLaunchKernelGGL(k_chain_dense1<false>, dim3((unsigned)blocks), dim3(64), lds, st, g);
	}
	return hipGetLastError();
}

} // namespace chaindp
