// chaindp_wave.h -- wave64 primitives (DPP scans, lane-mask helpers, wave-scope fences) and the anchor field
// accessors shared by the prepass and the chain DP kernels.  gfx950 only.
#ifndef CHAINDP_WAVE_H
#define CHAINDP_WAVE_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>

namespace chaindp {

#define NO_CHILD 0x7f7f7f7f   // first_child[]: larger than any read-relative index in use

// ---------------------------------------------------------------- wave primitives (wave64, DPP)

// dpp_ctrl encodings (gfx9 family): row_shr:n = 0x110+n, wave_shl:1 = 0x130, wave_shr:1 = 0x138, row_bcast:15 = 0x142, row_bcast:31 = 0x143
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_WAVE_SHR1 0x138
#define DPP_WAVE_SHL1 0x130
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

// sumq[r] holds the read's q_span sum (< 2^40); its top bit records "some anchor of the read carries a
// non-zero segment id", which sends the read's units to the general variant of the DP kernel.
#define SUMQ_SEG_FLAG (1ull << 63)
#define SUMQ_LUT16_FLAG (1ull << 62)   // the read's table of 1 - cost does not fit int8 (set by k_build_lut; k_chain_twin hands such reads over)
#define SUMQ_SPAN0_FLAG (1ull << 61)   // some anchor of the read has q_span 0 (k_chain_twin keeps scores minus one with a floor of 0: it hands such reads over)
#define SUMQ_FLAGS (SUMQ_SEG_FLAG | SUMQ_LUT16_FLAG | SUMQ_SPAN0_FLAG)

__device__ __forceinline__ int span_of_hi(uint32_t yhi) { return (int)(yhi & 0xffu); }        // (y>>32)&0xff
__device__ __forceinline__ int seg_of_hi(uint32_t yhi) { return (int)((yhi >> 16) & 0xffu); } // (y>>48)&0xff

// ---------------------------------------------------------------- scalar bit tricks on lane masks

// |x - y| of two unsigned values
__device__ __forceinline__ uint32_t absdiff_u32(uint32_t x, uint32_t y)
{
	uint32_t d;
	asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(x), "v"(y));
	return d;
}

// (1 << n) - 1 for n in [0, 63] as one scalar instruction
__device__ __forceinline__ uint64_t low_mask64(int n)
{
	uint64_t m;
	asm("s_bfm_b64 %0, %1, 0" : "=s"(m) : "s"(n));
	return m;
}

// highest set bit of a 64-bit lane mask; -64 for an empty mask (s_flbit_i32_b64 returns -1), which still gives an
// empty s_bfm_b64 mask because only the low 6 bits of the width are used
__device__ __forceinline__ int highest_lane(uint64_t m)
{
	int r;
	asm("s_flbit_i32_b64 %0, %1" : "=s"(r) : "s"(m));
	return r ^ 63;
}

// lowest set bit of a 64-bit lane mask; -1 for an empty mask
__device__ __forceinline__ int lowest_lane(uint64_t m)
{
	int r;
	asm("s_ff1_i32_b64 %0, %1" : "=s"(r) : "s"(m));
	return r;
}


} // namespace chaindp
#endif
