// issue_calib.hip -- measures the instruction-issue ceilings that bound k_chain_units on gfx950:
// wave64 integer VALU (plain, three-operand, DPP), SALU, LDS round trips, and mixes of them, at the
// occupancy the DP kernel runs at (8 waves per SIMD, 64-thread workgroups) and with every CU busy.
// Prints one JSON object; bench.py uses the figures as the "issue" ceilings next to the HBM roofline.
//   hipcc -O3 --offload-arch=gfx950 tools/issue_calib.hip -o tools/issue_calib && tools/issue_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// Each body executes, per loop trip, REP copies of a block of independent instructions.
#define R4(x) x x x x
#define R16(x) R4(R4(x))

enum { K_VADD, K_VMIN3, K_VDPP, K_VDPP_DEP, K_SALU, K_SALU64, K_MIX11, K_MIX_DP, K_LDS_RT, K_VCMP_BALLOT, K_READLANE, K_VADD_HALF, K_VADD_ONE, K_MIX21, K_MIX31, K_SNOP_MIX, K_VADD_SGPR, K_E64ADD, K_INLADD, K_LITADD, K_CNDVCC, K_CMPVCC, K_CMPE64, K_MOVS, K_MBCNT, K_BPERM, K_SDWA, K_ADDC, K_LDSRD128, K_PK, K_CND2, K_CND64, K_CMPONLY, K_BFI, K_SAD, K_LSHLADD, K_BCNT, K_RDLANE, K_DSMAX, K_DSW128, K_DSR128, K_DSR32, K_DSW32, K_CMPCND, K_CMPCND64, K_CNDINIT, K_VOP2MIX, K_SHMOV, K_N };
static const char *kname[K_N] = {"v_add_u32", "v_min3_i32", "v_max_i32_dpp_row_shr(indep)", "v_max_i32_dpp(dependent chain)", "s_add_u32",
                                 "s_and_b64", "mix 1 VALU : 1 SALU", "mix 31 VALU : 29 SALU : 5 LDS (DP shape)", "LDS write->read round trip (dependent)",
                                 "v_cmp + s_and_b64 vcc", "v_readlane_b32 + s_add", "v_add_u32 exec=low32", "v_add_u32 exec=1 lane",
                                 "mix 2 VALU : 1 SALU", "mix 3 VALU : 1 SALU", "mix 1 VALU : 1 s_nop 1", "v_add_u32 v, s, v",
	"v_add_u32_e64 v,v,v", "v_add_u32 v, 7(inline), v", "v_add_u32 v, 0x12345(literal), v", "v_cndmask_b32_e32 (vcc)", "v_cmp_gt_i32_e32 vcc + v_add", "v_cmp_gt_i32_e64 s[] + v_add", "v_mov_b32 v, s", "v_mbcnt_lo/hi", "ds_bpermute_b32 (x4, then wait)", "v_add_u32_sdwa", "v_addc_co_u32 (vcc carry-in)", "ds_read_b128 broadcast (x4, then wait)", "v_pk_add_u16 / v_pk_max_i16 (VOP3P)",
	"v_cndmask_b32_e32 vcc, distinct dst", "v_cndmask_b32_e64 s[20:21]", "v_cmp_gt_i32_e32 vcc only", "v_bfi_b32 v, v, v, v", "v_sad_u32 v, v, v, 0", "v_lshl_add_u32 v, v, 1, v", "v_bcnt_u32_b32 v, v, v", "v_readlane_b32 only", "ds_max_i64 exec=2 lanes (x4, wait)", "ds_write_b128 exec=2 lanes (x4, wait)", "ds_read_b128 aligned per-lane (x4, wait)", "ds_read_b32 per-lane (x4, wait)", "ds_write_b32 per-lane (x4, wait)",
	"v_cmp vcc + v_cndmask vcc pairs", "v_cmp_e64 s + v_cndmask_e64 s pairs", "v_cndmask vcc (vcc=0x5555.. set once per block)", "v_sub/v_min_u32/v_max_i32/v_and", "v_lshlrev/v_lshrrev/v_mov/v_or"};
// wave-instructions of each class per loop trip
static const int n_valu[K_N] = {64, 64, 64, 64, 0, 0, 32, 31, 0, 32, 32, 64, 64, 64, 96, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 0, 64, 64, 0, 64, 64, 64, 64, 64, 64, 64, 64, 64, 0, 0, 0, 0, 0, 64, 64, 64, 64, 64};
static const int n_salu[K_N] = {0, 0, 0, 0, 64, 64, 32, 29, 0, 32, 32, 0, 0, 32, 32, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 32, 0, 0};
static const int n_lds[K_N]  = {0, 0, 0, 0, 0, 0, 0, 5, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 64, 0, 0, 64, 0, 0, 0, 0, 0, 0, 0, 0, 0, 64, 64, 64, 64, 64, 0, 0, 0, 0, 0};

template <int KIND>
__global__ __launch_bounds__(64) void k_issue(int trips, uint32_t *out, unsigned long long *cyc)
{
	__shared__ __attribute__((aligned(16))) uint32_t lds[512];
	uint32_t v0 = threadIdx.x, v1 = v0 * 3, v2 = v0 ^ 5, v3 = v0 + 7, v4 = 1, v5 = 2, v6 = 3, v7 = 4;
	uint32_t s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
	uint64_t m0 = 0x5555, m1 = 0xff00ff;
	lds[threadIdx.x] = v0;
	const uint32_t la = threadIdx.x * 4, la16 = threadIdx.x * 16;
	(void)la16;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int t = 0; t < trips; ++t) {
		if constexpr (KIND == K_VADD) {
			R16(asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));)
		} else if constexpr (KIND == K_VMIN3) {
			R16(asm volatile("v_min3_i32 %0, %0, %4, %5\n\tv_min3_i32 %1, %1, %4, %5\n\tv_min3_i32 %2, %2, %4, %5\n\tv_min3_i32 %3, %3, %4, %5"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "v"(v5));)
		} else if constexpr (KIND == K_VDPP) {
			R16(asm volatile("v_max_i32_dpp %0, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
			                 "v_max_i32_dpp %1, %5, %5 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
			                 "v_max_i32_dpp %2, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
			                 "v_max_i32_dpp %3, %7, %7 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "v"(v5), "v"(v6), "v"(v7));)
		} else if constexpr (KIND == K_VDPP_DEP) {
			R16(asm volatile("v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
			                 "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
			                 "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
			                 "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0"
			                 : "+v"(v0));)
		} else if constexpr (KIND == K_SALU) {
			R16(asm volatile("s_add_u32 %0, %0, %4\n\ts_add_u32 %1, %1, %4\n\ts_add_u32 %2, %2, %4\n\ts_add_u32 %3, %3, %4"
			                 : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(trips) : "scc");)
		} else if constexpr (KIND == K_SALU64) {
			R16(asm volatile("s_and_b64 %0, %0, %1\n\ts_or_b64 %1, %1, %0\n\ts_and_b64 %0, %0, %1\n\ts_or_b64 %1, %1, %0"
			                 : "+s"(m0), "+s"(m1) :: "scc");)
		} else if constexpr (KIND == K_MIX11) {
			R4(R4(R4(asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, %3" : "+v"(v0), "+s"(s0) : "v"(v4), "s"(trips) : "scc");))
			   asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, %3" : "+v"(v1), "+s"(s1) : "v"(v4), "s"(trips) : "scc");)
			// the above is 4 * (16 + 1) = 68 pairs; trimmed below to the declared 32 by running half the trips (see host)
		} else if constexpr (KIND == K_MIX_DP) {
			// 31 VALU, 29 SALU, 5 LDS: roughly the DP kernel's per-anchor mix, with its two dependent LDS round trips
			asm volatile(
				"ds_read_b128 v[20:23], %4\n\t"
				"s_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t"
				"s_waitcnt lgkmcnt(0)\n\t"
				"v_sub_u32 %0, %0, v20\n\tv_sub_u32 %1, %1, v21\n\tv_sad_u32 v24, %0, %1, 0\n\tv_add_u32 v25, v24, %5\n\t"
				"v_max3_u32 v25, %0, %1, v25\n\tv_min3_i32 v26, %0, %1, %5\n\tv_min_u32 v24, v24, %5\n\tv_lshlrev_b32 v24, 1, v24\n\t"
				"s_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t"
				"ds_read_u16 v27, %6\n\t"
				"v_cmp_lt_u32 vcc, v25, %5\n\tv_sub_u32 v28, %5, v23\n\tv_min_u32 v28, v28, %5\n\tv_cndmask_b32 v28, %5, v28, vcc\n\t"
				"ds_write_b32 %6, %0\n\t"
				"ds_read_b32 v29, %6\n\t"
				"s_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\ts_add_u32 %2, %2, 1\n\t"
				"s_waitcnt lgkmcnt(0)\n\t"
				"v_add3_u32 v26, v26, v22, v27\n\tv_cndmask_b32 v26, %5, v26, vcc\n\t"
				"v_max_i32_dpp v30, v26, v26 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
				"v_max_i32_dpp v30, v30, v30 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
				"v_max_i32_dpp v30, v30, v30 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
				"v_max_i32_dpp v30, v30, v30 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
				"v_max_i32_dpp v30, v30, v30 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
				"v_max_i32_dpp v30, v30, v30 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
				"v_mov_b32_dpp v31, v30 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
				"v_max_i32 v31, v31, %5\n\t"
				"v_cmp_gt_i32 vcc, v26, v31\n\t"
				"s_and_b64 s[20:21], vcc, exec\n\t"
				"v_cmp_eq_u32 vcc, v29, %5\n\t"
				"s_andn2_b64 s[22:23], vcc, s[20:21]\n\t"
				"s_flbit_i32_b64 s24, s[20:21]\n\ts_ff1_i32_b64 s25, s[22:23]\n\ts_xor_b32 s24, s24, 63\n\ts_cmp_gt_u32 s25, s24\n\t"
				"s_bcnt1_i32_b64 s26, s[22:23]\n\ts_cmp_gt_u32 s26, 25\n\ts_sub_u32 s27, %2, s24\n\ts_lshl_b32 s27, s27, 2\n\t"
				"v_readlane_b32 s28, v26, 3\n\tv_readlane_b32 s29, %0, 63\n\t"
				"s_add_u32 s29, s29, 1\n\ts_cmp_gt_u32 s29, %2\n\ts_add_u32 %3, %3, s28\n\ts_add_u32 %2, %2, 16\n\ts_min_u32 %2, %2, %3\n\t"
				"v_mov_b32 v32, s28\n\tv_mov_b32 v33, s27\n\t"
				"ds_write_b128 %4, v[30:33]\n\t"
				"s_add_u32 %3, %3, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t"
				: "+v"(v0), "+v"(v1), "+s"(s0), "+s"(s1) : "v"(la * 4), "v"(v4), "v"(la)
				: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33",
				  "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "vcc", "scc", "memory");
		} else if constexpr (KIND == K_LDS_RT) {
			asm volatile("ds_write_b32 %1, %0\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(v0) : "v"(la) : "memory");
		} else if constexpr (KIND == K_VCMP_BALLOT) {
			R4(R4(asm volatile("v_cmp_gt_u32 vcc, %0, %2\n\ts_and_b64 %1, vcc, %1\n\tv_cmp_lt_u32 vcc, %0, %2\n\ts_or_b64 %1, vcc, %1"
			                   : "+v"(v0), "+s"(m0) : "v"(v4) : "vcc", "scc");))
		} else if constexpr (KIND == K_READLANE) {
			R4(R4(asm volatile("v_readlane_b32 %1, %0, 5\n\ts_add_u32 %2, %2, %1\n\tv_readlane_b32 %1, %0, 9\n\ts_add_u32 %2, %2, %1"
			                   : "+v"(v0), "+s"(s2), "+s"(s0) :: "scc");))
		} else if constexpr (KIND == K_VADD_HALF || KIND == K_VADD_ONE) {
			if constexpr (KIND == K_VADD_HALF) asm volatile("s_mov_b64 exec, 0xffffffff" ::: "memory");
			else asm volatile("s_mov_b64 exec, 1" ::: "memory");
			R16(asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));)
			asm volatile("s_mov_b64 exec, -1" ::: "memory");
		} else if constexpr (KIND == K_MIX21) {
			R16(asm volatile("v_add_u32 %0, %0, %3\n\tv_add_u32 %1, %1, %3\n\ts_add_u32 %2, %2, %4\n\tv_add_u32 %0, %0, %3\n\tv_add_u32 %1, %1, %3\n\ts_add_u32 %2, %2, %4"
			                 : "+v"(v0), "+v"(v1), "+s"(s0) : "v"(v4), "s"(trips) : "scc");)
		} else if constexpr (KIND == K_MIX31) {
			R16(asm volatile("v_add_u32 %0, %0, %3\n\tv_add_u32 %1, %1, %3\n\tv_add_u32 %0, %0, %3\n\ts_add_u32 %2, %2, %4\n\tv_add_u32 %1, %1, %3\n\tv_add_u32 %0, %0, %3\n\tv_add_u32 %1, %1, %3\n\ts_add_u32 %2, %2, %4"
			                 : "+v"(v0), "+v"(v1), "+s"(s0) : "v"(v4), "s"(trips) : "scc");)
		} else if constexpr (KIND == K_SNOP_MIX) {
			R16(asm volatile("v_add_u32 %0, %0, %2\n\ts_nop 1\n\tv_add_u32 %1, %1, %2\n\ts_nop 1\n\tv_add_u32 %0, %0, %2\n\ts_nop 1\n\tv_add_u32 %1, %1, %2\n\ts_nop 1"
			                 : "+v"(v0), "+v"(v1) : "v"(v4));)
		} else if constexpr (KIND == K_VADD_SGPR) {
			R16(asm volatile("v_add_u32 %0, %4, %0\n\tv_add_u32 %1, %4, %1\n\tv_add_u32 %2, %4, %2\n\tv_add_u32 %3, %4, %3"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "s"(trips));)
		} else if constexpr (KIND == K_E64ADD) {
			R16(asm volatile("v_add_u32_e64 %0, %0, %4\n\tv_add_u32_e64 %1, %1, %4\n\tv_add_u32_e64 %2, %2, %4\n\tv_add_u32_e64 %3, %3, %4"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_INLADD) {
			R16(asm volatile("v_add_u32 %0, 7, %0\n\tv_add_u32 %1, 7, %1\n\tv_add_u32 %2, 7, %2\n\tv_add_u32 %3, 7, %3"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_LITADD) {
			R16(asm volatile("v_add_u32 %0, 0x12345, %0\n\tv_add_u32 %1, 0x12345, %1\n\tv_add_u32 %2, 0x12345, %2\n\tv_add_u32 %3, 0x12345, %3"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_CNDVCC) {
			R16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_CMPVCC) {
			R16(asm volatile("v_cmp_gt_i32 vcc, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_cmp_gt_i32 vcc, %2, %4\n\tv_add_u32 %3, %3, %4"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_CMPE64) {
			R16(asm volatile("v_cmp_gt_i32_e64 s[20:21], %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_cmp_gt_i32_e64 s[22:23], %2, %4\n\tv_add_u32 %3, %3, %4"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_MOVS) {
			R16(asm volatile("v_mov_b32 %0, s20\n\tv_mov_b32 %1, s21\n\tv_mov_b32 %2, s22\n\tv_mov_b32 %3, s23"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_MBCNT) {
			R16(asm volatile("v_mbcnt_lo_u32_b32 %0, s20, 0\n\tv_mbcnt_hi_u32_b32 %1, s21, %0\n\tv_mbcnt_lo_u32_b32 %2, s22, 0\n\tv_mbcnt_hi_u32_b32 %3, s23, %2"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_BPERM) {
			R16(asm volatile("ds_bpermute_b32 %0, %4, %0\n\tds_bpermute_b32 %1, %4, %1\n\tds_bpermute_b32 %2, %4, %2\n\tds_bpermute_b32 %3, %4, %3\n\ts_waitcnt lgkmcnt(0)"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(la) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_SDWA) {
			R16(asm volatile("v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\tv_add_u32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\tv_add_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\tv_add_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_ADDC) {
			R16(asm volatile("v_addc_co_u32 %0, vcc, %0, %4, vcc\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32 %2, vcc, %2, %4, vcc\n\tv_addc_co_u32 %3, vcc, %3, %4, vcc"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_LDSRD128) {
			R16(asm volatile("ds_read_b128 v[20:23], %4\n\tds_read_b128 v[24:27], %4\n\tds_read_b128 v[28:31], %4\n\tds_read_b128 v[32:35], %4\n\ts_waitcnt lgkmcnt(0)"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(la) : "vcc", "s20", "s21", "s22", "s23", "memory", "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");)
		} else if constexpr (KIND == K_PK) {
			R16(asm volatile("v_pk_add_u16 %0, %0, %4\n\tv_pk_max_i16 %1, %1, %4\n\tv_pk_add_u16 %2, %2, %4\n\tv_pk_max_i16 %3, %3, %4"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_CND2) {
			R16(asm volatile("v_cndmask_b32 %0, %4, %1, vcc\n\tv_cndmask_b32 %1, %4, %2, vcc\n\tv_cndmask_b32 %2, %4, %3, vcc\n\tv_cndmask_b32 %3, %4, %0, vcc"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_CND64) {
			R16(asm volatile("v_cndmask_b32_e64 %0, %4, %1, s[20:21]\n\tv_cndmask_b32_e64 %1, %4, %2, s[20:21]\n\tv_cndmask_b32_e64 %2, %4, %3, s[20:21]\n\tv_cndmask_b32_e64 %3, %4, %0, s[20:21]"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_CMPONLY) {
			R16(asm volatile("v_cmp_gt_i32 vcc, %0, %4\n\tv_cmp_gt_i32 vcc, %1, %4\n\tv_cmp_gt_i32 vcc, %2, %4\n\tv_cmp_gt_i32 vcc, %3, %4"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_BFI) {
			R16(asm volatile("v_bfi_b32 %0, %4, %0, %4\n\tv_bfi_b32 %1, %4, %1, %4\n\tv_bfi_b32 %2, %4, %2, %4\n\tv_bfi_b32 %3, %4, %3, %4"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_SAD) {
			R16(asm volatile("v_sad_u32 %0, %0, %4, 0\n\tv_sad_u32 %1, %1, %4, 0\n\tv_sad_u32 %2, %2, %4, 0\n\tv_sad_u32 %3, %3, %4, 0"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_LSHLADD) {
			R16(asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n\tv_lshl_add_u32 %1, %1, 1, %4\n\tv_lshl_add_u32 %2, %2, 1, %4\n\tv_lshl_add_u32 %3, %3, 1, %4"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_BCNT) {
			R16(asm volatile("v_bcnt_u32_b32 %0, %0, %4\n\tv_bcnt_u32_b32 %1, %1, %4\n\tv_bcnt_u32_b32 %2, %2, %4\n\tv_bcnt_u32_b32 %3, %3, %4"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_RDLANE) {
			R16(asm volatile("v_readlane_b32 s20, %0, 3\n\tv_readlane_b32 s21, %1, 5\n\tv_readlane_b32 s22, %2, 7\n\tv_readlane_b32 s23, %3, 9"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_DSMAX) {
			R16(asm volatile("s_mov_b32 exec_lo, 1\n\ts_mov_b32 exec_hi, 1\n\tds_max_i64 %4, v[20:21]\n\tds_max_i64 %4, v[20:21]\n\tds_max_i64 %4, v[20:21]\n\tds_max_i64 %4, v[20:21]\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(la16) : "vcc", "s20", "s21", "s22", "s23", "memory", "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");)
		} else if constexpr (KIND == K_DSW128) {
			R16(asm volatile("s_mov_b32 exec_lo, 1\n\ts_mov_b32 exec_hi, 1\n\tds_write_b128 %4, v[20:23]\n\tds_write_b128 %4, v[20:23]\n\tds_write_b128 %4, v[20:23]\n\tds_write_b128 %4, v[20:23]\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(la16) : "vcc", "s20", "s21", "s22", "s23", "memory", "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");)
		} else if constexpr (KIND == K_DSR128) {
			R16(asm volatile("ds_read_b128 v[20:23], %4\n\tds_read_b128 v[24:27], %4\n\tds_read_b128 v[28:31], %4\n\tds_read_b128 v[32:35], %4\n\ts_waitcnt lgkmcnt(0)"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(la16) : "vcc", "s20", "s21", "s22", "s23", "memory", "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");)
		} else if constexpr (KIND == K_DSR32) {
			R16(asm volatile("ds_read_b32 v20, %4\n\tds_read_b32 v21, %4\n\tds_read_b32 v22, %4\n\tds_read_b32 v23, %4\n\ts_waitcnt lgkmcnt(0)"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(la) : "vcc", "s20", "s21", "s22", "s23", "memory", "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");)
		} else if constexpr (KIND == K_DSW32) {
			R16(asm volatile("ds_write_b32 %4, v20\n\tds_write_b32 %4, v21\n\tds_write_b32 %4, v22\n\tds_write_b32 %4, v23\n\ts_waitcnt lgkmcnt(0)"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(la) : "vcc", "s20", "s21", "s22", "s23", "memory", "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");)
		} else if constexpr (KIND == K_CMPCND) {
			R16(asm volatile("v_cmp_gt_i32 vcc, %0, %4\n\tv_cndmask_b32 %1, %4, %1, vcc\n\tv_cmp_gt_i32 vcc, %2, %4\n\tv_cndmask_b32 %3, %4, %3, vcc"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_CMPCND64) {
			R16(asm volatile("v_cmp_gt_i32_e64 s[20:21], %0, %4\n\tv_cndmask_b32_e64 %1, %4, %1, s[20:21]\n\tv_cmp_gt_i32_e64 s[22:23], %2, %4\n\tv_cndmask_b32_e64 %3, %4, %3, s[22:23]"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_CNDINIT) {
			R16(asm volatile("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555\n\tv_cndmask_b32 %0, %4, %1, vcc\n\tv_cndmask_b32 %1, %4, %2, vcc\n\tv_cndmask_b32 %2, %4, %3, vcc\n\tv_cndmask_b32 %3, %4, %0, vcc"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_VOP2MIX) {
			R16(asm volatile("v_sub_u32 %0, %0, %4\n\tv_min_u32 %1, %1, %4\n\tv_max_i32 %2, %2, %4\n\tv_and_b32 %3, %3, %4"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		} else if constexpr (KIND == K_SHMOV) {
			R16(asm volatile("v_lshlrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_mov_b32 %2, %4\n\tv_or_b32 %3, %3, %4"
			                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc", "s20", "s21", "s22", "s23", "memory");)
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	out[blockIdx.x * 64 + threadIdx.x] = v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3 + (uint32_t)m0 + (uint32_t)m1 + lds[(threadIdx.x + 1) & 63];
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*kfn)(int, uint32_t*, unsigned long long*);

int main(int argc, char **argv)
{
	int dev = 0;
	CK(hipSetDevice(dev));
	hipDeviceProp_t pr;
	CK(hipGetDeviceProperties(&pr, dev));
	const int cus = pr.multiProcessorCount;
	const int waves_per_simd_list[3] = {1, 4, 8};
	uint32_t *d_out; unsigned long long *d_cyc;
	const int max_blocks = cus * 32;
	CK(hipMalloc(&d_out, (size_t)max_blocks * 64 * 4));
	CK(hipMalloc(&d_cyc, (size_t)max_blocks * 8));
	kfn fns[K_N] = {k_issue<K_VADD>, k_issue<K_VMIN3>, k_issue<K_VDPP>, k_issue<K_VDPP_DEP>, k_issue<K_SALU>, k_issue<K_SALU64>,
	                k_issue<K_MIX11>, k_issue<K_MIX_DP>, k_issue<K_LDS_RT>, k_issue<K_VCMP_BALLOT>, k_issue<K_READLANE>,
	                k_issue<K_VADD_HALF>, k_issue<K_VADD_ONE>, k_issue<K_MIX21>, k_issue<K_MIX31>, k_issue<K_SNOP_MIX>, k_issue<K_VADD_SGPR>,
	                k_issue<K_E64ADD>, k_issue<K_INLADD>, k_issue<K_LITADD>, k_issue<K_CNDVCC>, k_issue<K_CMPVCC>, k_issue<K_CMPE64>, k_issue<K_MOVS>, k_issue<K_MBCNT>, k_issue<K_BPERM>, k_issue<K_SDWA>, k_issue<K_ADDC>, k_issue<K_LDSRD128>, k_issue<K_PK>,
	                k_issue<K_CND2>, k_issue<K_CND64>, k_issue<K_CMPONLY>, k_issue<K_BFI>, k_issue<K_SAD>, k_issue<K_LSHLADD>, k_issue<K_BCNT>, k_issue<K_RDLANE>, k_issue<K_DSMAX>, k_issue<K_DSW128>, k_issue<K_DSR128>, k_issue<K_DSR32>, k_issue<K_DSW32>,
	                k_issue<K_CMPCND>, k_issue<K_CMPCND64>, k_issue<K_CNDINIT>, k_issue<K_VOP2MIX>, k_issue<K_SHMOV>};
	int nv[K_N], ns[K_N], nl[K_N];
	memcpy(nv, n_valu, sizeof nv); memcpy(ns, n_salu, sizeof ns); memcpy(nl, n_lds, sizeof nl);
	nv[K_MIX11] = 68; ns[K_MIX11] = 68;
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"results\": [\n", pr.gcnArchName, cus, pr.clockRate / 1000);
	bool first = true;
	for (int k = argc > 1 ? atoi(argv[1]) : 0; k < K_N; ++k) {
		for (int wi = 0; wi < 3; ++wi) {
			const int wps = waves_per_simd_list[wi];
			const int blocks = cus * 4 * wps;         // one resident round: every SIMD holds wps waves
			const int trips = 20000;
			hipLaunchKernelGGL(fns[k], dim3(blocks), dim3(64), 0, 0, 200, d_out, d_cyc);
			CK(hipDeviceSynchronize());
			CK(hipEventRecord(e0, 0));
			hipLaunchKernelGGL(fns[k], dim3(blocks), dim3(64), 0, 0, trips, d_out, d_cyc);
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms = 0;
			CK(hipEventElapsedTime(&ms, e0, e1));
			std::vector<unsigned long long> cyc(blocks);
			CK(hipMemcpy(cyc.data(), d_cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost));
			double csum = 0; for (auto c : cyc) csum += (double)c;
			const double cyc_wave = csum / blocks;                  // s_memtime ticks (100 MHz constant clock on gfx9? reported as is)
			const double sec = ms * 1e-3;
			const double inst_total = (double)blocks * trips;
			const double valu_rate = inst_total * nv[k] / sec / 1e9, salu_rate = inst_total * ns[k] / sec / 1e9, lds_rate = inst_total * nl[k] / sec / 1e9;
			// cycles per instruction per SIMD (VALU) / per CU (SALU), assuming the nominal clock
			const double clk = pr.clockRate * 1e3;
			const double cyc_per_valu_simd = nv[k] ? sec * clk / ((double)wps * trips * nv[k]) : 0;
			const double cyc_per_salu_cu = ns[k] ? sec * clk / ((double)wps * 4 * trips * ns[k]) : 0;
			printf("%s  {\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"memtime_ticks_per_wave\": %.0f, \"valu_Ginst_s\": %.1f, \"salu_Ginst_s\": %.1f, \"lds_Ginst_s\": %.1f, "
			       "\"nominal_cycles_per_valu_per_simd\": %.3f, \"nominal_cycles_per_salu_per_cu\": %.3f, \"ns_per_trip_per_wave\": %.2f}",
			       first ? "" : ",\n", kname[k], wps, ms, cyc_wave, valu_rate, salu_rate, lds_rate, cyc_per_valu_simd, cyc_per_salu_cu, sec * 1e9 / trips);
			first = false;
		}
	}
	printf("\n]}\n");
	return 0;
}
