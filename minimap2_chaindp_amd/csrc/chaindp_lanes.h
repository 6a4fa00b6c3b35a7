// chaindp_lanes.h -- what the part-wave DP kernels (chaindp_twin.hip: two units per wave; chaindp_quad.hip: four) share: raw LDS
// access by byte address, loads through the scalar cache, lane masks straight from vector compares.  gfx950 only.
#ifndef CHAINDP_LANES_H
#define CHAINDP_LANES_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace chaindp {

#if defined(__HIP_DEVICE_COMPILE__)
#define TW_LDS(T, a) ((__attribute__((address_space(3))) T*)(a))
#else
#define TW_LDS(T, a) ((T*)(uintptr_t)(a))          /* host pass of the single-source compile; never executed */
#endif
// unit records and their UnitAux are read through the scalar cache: a load from the constant address space with a wave-uniform
// address is an s_load (the arrays are written by the prepass, never by this kernel)
#if defined(__HIP_DEVICE_COMPILE__)
#define TW_CONST(T, p) ((const __attribute__((address_space(4))) T*)(uintptr_t)(p))
#else
#define TW_CONST(T, p) ((const T*)(p))
#endif
typedef uint32_t tw_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t tw_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ tw_u32x2 tw_ld64(uint32_t a) { return *TW_LDS(const tw_u32x2, a); }
__device__ __forceinline__ int tw_ld32(uint32_t a) { return *TW_LDS(const int, a); }
__device__ __forceinline__ int tw_ld_i8(uint32_t a) { return (int)*TW_LDS(const signed char, a); }
__device__ __forceinline__ int tw_ld_u8(uint32_t a) { return (int)*TW_LDS(const unsigned char, a); }
__device__ __forceinline__ tw_u32x4 tw_ld128(uint32_t a) { return *TW_LDS(const tw_u32x4, a); }
__device__ __forceinline__ void tw_st64(uint32_t a, uint32_t x, uint32_t y) { tw_u32x2 t; t.x = x; t.y = y; *TW_LDS(tw_u32x2, a) = t; }
__device__ __forceinline__ void tw_st32(uint32_t a, int v) { *TW_LDS(int, a) = v; }
__device__ __forceinline__ void tw_st8(uint32_t a, int v) { *TW_LDS(signed char, a) = (signed char)v; }
__device__ __forceinline__ void tw_st128(uint32_t a, uint32_t x, uint32_t y, uint32_t z, uint32_t w) { tw_u32x4 t; t.x = x; t.y = y; t.z = z; t.w = w; *TW_LDS(tw_u32x4, a) = t; }

// keeps a value in a vector register: the compiler would otherwise hold wave-uniform values in SGPRs and feed them
// to VALU instructions as scalar operands, which halves their issue rate
#if defined(__HIP_DEVICE_COMPILE__)
#define TW_VREG(x) asm volatile("" : "+v"(x))
#else
#define TW_VREG(x) ((void)(x))
#endif

// a lane mask is wave-uniform by construction; where the compiler's divergence analysis loses track of that (values merged
// behind loops) this keeps it in scalar registers (folds away when it already is)
#define TW_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))

// |a - b| + c in one instruction
__device__ __forceinline__ uint32_t tw_sad(uint32_t a, uint32_t b, uint32_t c)
{
	uint32_t d = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
#endif
	return d;
}

// lane masks straight from a vector compare (v_cmp_*_e64 into an SGPR pair, no bool in between)
#define TW_ULT(a, b) __builtin_amdgcn_uicmp((unsigned)(a), (unsigned)(b), 36)
#define TW_EQ(a, b)  __builtin_amdgcn_uicmp((unsigned)(a), (unsigned)(b), 32)
#define TW_SGT(a, b) __builtin_amdgcn_sicmp((int)(a), (int)(b), 38)
#define TW_SGE(a, b) __builtin_amdgcn_sicmp((int)(a), (int)(b), 39)
#define TW_SEL(m, a, b) (__builtin_amdgcn_inverse_ballot_w64(m) ? (a) : (b))

} // namespace chaindp
#endif
