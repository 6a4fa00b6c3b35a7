// chaindp_compact.hip -- f/p/v -> new_seed[] (reference chain.c:286-317) as anchor-parallel kernels, plus the
// small multi-block scan both this stage and the prepass use.
//
// The reference compacts while it runs the recurrence, in anchor order:
//   at step k: if p[k] >= 0 and p[k] has not been emitted yet, emit p[k] first ("late" emission,
//   chain.c:287-303); then emit k itself iff v[k] >= min_sc || p[k] >= 0 (chain.c:304-316).
// An anchor i is emitted at its own step iff self(i) = v[i] >= min_sc || p[i] >= 0 (flags bit1); otherwise it is
// emitted late, just before the FIRST later k with p[k] == i (if any).  Hence, per read:
//   first_child[i] = min{k : p[k] == i}                      (only needed where !self(i))
//   late(k)  = p[k] >= 0 && !self(p[k]) && first_child[p[k]] == k
//   count(k) = late(k) + self(k)   in {0,1,2};   pos = exclusive prefix sum of count
//   id[p[k]] = pos[k] if late(k);  id[k] = pos[k] + late(k) if self(k)
//   record of anchor i = { a[i], (id[p[i]] << 2 or -4) | (v[i] >= min_sc) | (f[i] < v[i]) << 1, f[i] }
// (a late-emitted anchor has p < 0, so its record carries -4 | flags).  Because a read's records are
// contiguous and reads follow each other, the prefix sum is taken over the whole batch and a read's
// offset is the value at its first anchor: seeds_off[r] = pos[off[r]]; ids are made read-relative when
// the records are written.  oracle/chain_oracle.c:co_compact is the sequential statement of the same thing;
// tests compare the bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "chaindp_kernels.h"

namespace chaindp {

struct SeedRec { uint64_t x, y; int32_t p, f; };   // == struct new_seed (minimap.h:51-55)

#define CMP_BLOCK 256
#define CMP_PER_BLOCK CHAINDP_BLOCK_ANCHORS

__device__ __forceinline__ bool self_emit(int32_t vi, int32_t pi, int min_sc) { return vi >= min_sc || pi >= 0; }

// largest r in [lo, hi] with off[r] <= g
__device__ __forceinline__ int64_t read_of_c(const int64_t *__restrict__ off, int64_t lo, int64_t hi, int64_t g)
{
	if (hi - lo == 1) return g >= off[hi] ? hi : lo;               // a block that straddles one read boundary: the usual case
	while (lo < hi) {
		const int64_t mid = (lo + hi + 1) >> 1;
		if (off[mid] <= g) lo = mid; else hi = mid - 1;
	}
	return lo;
}

// reads touched by a block's anchor range: tabulated by the prepass (k_block_reads), same 1024-anchor blocks
__device__ __forceinline__ void block_read_range(const int2 *__restrict__ block_reads, int64_t &rlo, int64_t &rhi)
{
	const int2 rr = block_reads[blockIdx.x];
	rlo = rr.x; rhi = rr.y;
}

// ---------------------------------------------------------------- exclusive scan of uint64 items (3 small kernels)
// Items are per-1024-anchor block counts (two 32-bit counters packed in one word), so even a 1 G anchor batch
// is only 1 M items; level 1 scans tiles of 1024 items, level 2 the tile totals in a single block, level 3 adds.

__global__ __launch_bounds__(256) void k_scan_l1(int64_t n, unsigned long long *__restrict__ data, unsigned long long *__restrict__ tile_tot)
{
	__shared__ unsigned long long s_w[4];
	const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
	unsigned long long v[4], sum = 0;
	for (int k = 0; k < 4; ++k) { v[k] = base + k < n ? data[base + k] : 0; sum += v[k]; }
	unsigned long long incl = sum;
	for (int d = 1; d < 64; d <<= 1) { const unsigned long long t = __shfl_up(incl, d, 64); if ((threadIdx.x & 63) >= d) incl += t; }
	if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
	__syncthreads();
	unsigned long long woff = 0;
	for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += s_w[w];
	unsigned long long ex = woff + incl - sum;
	for (int k = 0; k < 4; ++k) { if (base + k < n) data[base + k] = ex; ex += v[k]; }
	if (threadIdx.x == 255) tile_tot[blockIdx.x] = woff + incl;
}

__global__ __launch_bounds__(1024) void k_scan_l2(int64_t n_tiles, unsigned long long *__restrict__ tile_tot, unsigned long long *__restrict__ total_out)
{
	__shared__ unsigned long long part[1024];
	const int tid = threadIdx.x;
	const int64_t per = (n_tiles + 1023) / 1024;
	const int64_t lo = (int64_t)tid * per, hi = lo + per < n_tiles ? lo + per : n_tiles;
	unsigned long long s = 0;
	for (int64_t k = lo; k < hi; ++k) s += tile_tot[k];
	// exclusive scan of the 1024 partial sums: inside each wave by shuffles, the sixteen wave totals by every thread (one thread
	// walking all 1024 was 12 us of a 4 ms step, twice per step)
	unsigned long long incl = s;
	for (int d = 1; d < 64; d <<= 1) { const unsigned long long t = __shfl_up(incl, d, 64); if ((tid & 63) >= d) incl += t; }
	if ((tid & 63) == 63) part[tid >> 6] = incl;
	__syncthreads();
	unsigned long long woff = 0, all = 0;
	for (int w = 0; w < 16; ++w) { const unsigned long long t = part[w]; if (w < (tid >> 6)) woff += t; all += t; }
	if (tid == 0) *total_out = all;
	unsigned long long acc = woff + incl - s;
	for (int64_t k = lo; k < hi; ++k) { const unsigned long long t = tile_tot[k]; tile_tot[k] = acc; acc += t; }
}

__global__ __launch_bounds__(256) void k_scan_l3(int64_t n, unsigned long long *__restrict__ data, const unsigned long long *__restrict__ tile_off)
{
	const unsigned long long add = tile_off[blockIdx.x];
	const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
	for (int k = 0; k < 4; ++k) if (base + k < n) data[base + k] += add;
}

hipError_t launch_scan_u64(hipStream_t st, int64_t n, unsigned long long *d_data, unsigned long long *d_tile_tmp,
                           unsigned long long *d_total)
{
	if (n <= 0) return hipMemsetAsync(d_total, 0, sizeof(unsigned long long), st);
	const int64_t tiles = (n + 1023) / 1024;
	hipLaunchKernelGGL(k_scan_l1, dim3((unsigned)tiles), dim3(256), 0, st, n, d_data, d_tile_tmp);
	hipLaunchKernelGGL(k_scan_l2, dim3(1), dim3(1024), 0, st, tiles, d_tile_tmp, d_total);
	if (tiles > 1) hipLaunchKernelGGL(k_scan_l3, dim3((unsigned)tiles), dim3(256), 0, st, n, d_data, d_tile_tmp);
	return hipGetLastError();
}

// ---------------------------------------------------------------- compaction


// C1 happens inside the DP kernel's tile flush (k_chain_units) and the prepass (singletons): they set bit1 of
// flags[] ("emitted at its own step"), bit2 ("my predecessor is not emitted at its own step, so I may be its first
// child"), bits 3-4 (v >= min_sc, f < v: the record's own flag bits) and feed first_child[] with atomicMin for exactly
// those predecessors.
// C2: late bit and per-block record counts.  Only anchors with bit2 look at p[] and first_child[]; for everything
// else this pass reads one byte per anchor.  One wave per 1024-anchor block, 16 consecutive anchors per lane (one
// 16-byte load in flight per lane; a 256-thread block per 1024 anchors was bound by its own start-up latency).
// Flag bytes of four consecutive anchors with the singletons among them set right (s4 / e4: the singleton / emitted-singleton bits
// of the four, from the prepass' masks -- nobody stores a singleton's flag byte, what the array holds there is an earlier batch's).
// A singleton: bit5; an emitted one also "own step" (bit1) and v >= min_sc (bit3); never a first-child candidate, never late.
__device__ __forceinline__ uint32_t flags_with_singles(uint32_t w, uint32_t s4, uint32_t e4)
{
	const uint32_t sb = (s4 * 0x00204081u) & 0x01010101u, eb = (e4 * 0x00204081u) & 0x01010101u;   // bit i -> bit 0 of byte i
	return (w & ~(sb * 0xffu)) | sb * 0x20u | eb * 0x0au;
}
// the mask bits of the n <= 16 anchors from g on (g a multiple of 4: they sit in one word)
__device__ __forceinline__ uint32_t mask_bits16(const uint64_t *__restrict__ m, int64_t g) { return (uint32_t)(m[g >> 6] >> (g & 63)) & 0xffffu; }

__device__ __forceinline__ unsigned int flag_word(const uint4 &v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }

#define CNT_BLOCKS_PER_WAVE 4          // 1024-anchor blocks a wave of k_count takes: their flag loads are all in flight before the first is used
__device__ __forceinline__ void flag_or(uint4 &v, int e, unsigned int bits)
{
	const unsigned int m = bits << (8 * (e & 3));
	if ((e >> 2) == 0) v.x |= m; else if ((e >> 2) == 1) v.y |= m; else if ((e >> 2) == 2) v.z |= m; else v.w |= m;
}
__global__ __launch_bounds__(CMP_BLOCK) void k_count(int64_t n_reads, int64_t total,
                                                     const int64_t *__restrict__ off, const int32_t *__restrict__ p,
                                                     const int32_t *__restrict__ first_child,
                                                     const uint8_t *__restrict__ flags, unsigned long long *__restrict__ block_cnt,
                                                     const int2 *__restrict__ block_reads, uint32_t *__restrict__ sub,
                                                     const uint64_t *__restrict__ single_mask, const uint64_t *__restrict__ emit_mask)
{
	const int lane = threadIdx.x & 63;
	const int64_t blk0 = ((int64_t)blockIdx.x * (CMP_BLOCK / 64) + (threadIdx.x >> 6)) * CNT_BLOCKS_PER_WAVE;
	if (blk0 * CMP_PER_BLOCK >= total) return;
	// A wave per 1024-anchor block spent its life waiting: for one 16-byte load per lane, then -- walking its 16 anchors in step
	// with the other lanes -- for two dependent loads (p, first_child) at every position where ANY lane had a candidate (110 us for
	// 76 M anchors: 0.7 TB/s of a byte per anchor).  Now: four blocks per wave, their flag loads in flight together, and every
	// lane takes its own candidates one after the other whatever their positions, so a wave waits as often as its busiest lane
	// has candidates (two or three times), not sixteen times.
	uint4 vv[CNT_BLOCKS_PER_WAVE];
	int2 rr[CNT_BLOCKS_PER_WAVE];
	uint32_t sm[CNT_BLOCKS_PER_WAVE];                               // the lane's singleton bits (low half) and emitted-singleton bits
	unsigned long long cand = 0;                                    // bit 16 b + e: anchor e of this lane's 16 in block b may be a first child
#pragma unroll
	for (int b = 0; b < CNT_BLOCKS_PER_WAVE; ++b) {
		const int64_t g = (blk0 + b) * CMP_PER_BLOCK + 16 * lane;
		const int64_t g1 = (blk0 + b + 1) * CMP_PER_BLOCK < total ? (blk0 + b + 1) * CMP_PER_BLOCK : total;
		vv[b] = make_uint4(0u, 0u, 0u, 0u);
		rr[b] = make_int2(0, 0); sm[b] = 0;
		if (g < g1) { vv[b] = *(const uint4*)(flags + g); rr[b] = block_reads[blk0 + b]; sm[b] = mask_bits16(single_mask, g) | mask_bits16(emit_mask, g) << 16; }   // 16 flag bytes (the array is padded to 16 B)
	}
#pragma unroll
	for (int b = 0; b < CNT_BLOCKS_PER_WAVE; ++b) {                 // the singletons' bytes
		vv[b].x = flags_with_singles(vv[b].x, sm[b] & 0xfu, (sm[b] >> 16) & 0xfu);
		vv[b].y = flags_with_singles(vv[b].y, (sm[b] >> 4) & 0xfu, (sm[b] >> 20) & 0xfu);
		vv[b].z = flags_with_singles(vv[b].z, (sm[b] >> 8) & 0xfu, (sm[b] >> 24) & 0xfu);
		vv[b].w = flags_with_singles(vv[b].w, (sm[b] >> 12) & 0xfu, (sm[b] >> 28) & 0xfu);
	}
#pragma unroll
	for (int b = 0; b < CNT_BLOCKS_PER_WAVE; ++b) {
		const int64_t g = (blk0 + b) * CMP_PER_BLOCK + 16 * lane;
		const int64_t g1 = (blk0 + b + 1) * CMP_PER_BLOCK < total ? (blk0 + b + 1) * CMP_PER_BLOCK : total;
		const int n = g >= g1 ? 0 : g1 - g < 16 ? (int)(g1 - g) : 16;
		for (int k = 0; k < 4; ++k) {
			const unsigned int w = flag_word(vv[b], k) & 0x04040404u;
			// bit 2 of byte j of word k -> bit 4 k + j
			const unsigned int c4 = ((w >> 2) & 1u) | ((w >> 9) & 2u) | ((w >> 16) & 4u) | ((w >> 23) & 8u);
			cand |= (unsigned long long)c4 << (16 * b + 4 * k);
		}
		if (n < 16) cand &= ~(((1ull << (16 - n)) - 1ull) << (16 * b + n));
	}
	// two candidates per lane and trip: their loads (the read's start and p, then first_child) are in flight side by side
	while (__builtin_amdgcn_ballot_w64(cand != 0)) {
		int bit[2];
		int64_t ge[2] = {0, 0}, rs[2] = {0, 0};
		int32_t pe[2] = {0, 0}, fc[2] = {0, 0};
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			bit[c] = -1;
			if (cand) {
				bit[c] = __builtin_ctzll(cand);
				cand &= cand - 1;
				const int b = bit[c] >> 4;
				ge[c] = (blk0 + b) * CMP_PER_BLOCK + 16 * lane + (bit[c] & 15);
				const int2 r2 = b == 0 ? rr[0] : b == 1 ? rr[1] : b == 2 ? rr[2] : rr[3];
				rs[c] = off[r2.x == r2.y ? r2.x : read_of_c(off, r2.x, r2.y, ge[c])];
				pe[c] = p[ge[c]];
			}
		}
#pragma unroll
		for (int c = 0; c < 2; ++c) if (bit[c] >= 0) fc[c] = first_child[rs[c] + pe[c]];
#pragma unroll
		for (int c = 0; c < 2; ++c)
			if (bit[c] >= 0 && fc[c] == (int32_t)(ge[c] - rs[c])) {
				const int b = bit[c] >> 4, e = bit[c] & 15;
				if (b == 0) flag_or(vv[0], e, 1u); else if (b == 1) flag_or(vv[1], e, 1u); else if (b == 2) flag_or(vv[2], e, 1u); else flag_or(vv[3], e, 1u);
			}
	}
#pragma unroll
	for (int b = 0; b < CNT_BLOCKS_PER_WAVE; ++b) {
		const int64_t blk = blk0 + b;
		const int64_t g0 = blk * CMP_PER_BLOCK;
		if (g0 >= total) break;
		const int64_t g1 = g0 + CMP_PER_BLOCK < total ? g0 + CMP_PER_BLOCK : total;
		const int64_t g = g0 + 16 * lane;
		unsigned int mine = 0;
		if (g < g1) {
			const int n = g1 - g < 16 ? (int)(g1 - g) : 16;
			for (int k = 0; k < 4; ++k) {
				unsigned int w = flag_word(vv[b], k);
				const int left = n - 4 * k;
				if (left <= 0) w = 0; else if (left < 4) w &= (1u << (8 * left)) - 1u;
				mine += (unsigned int)__builtin_popcount(w & 0x03030303u);    // late + self per anchor
			}
		}
		// records in front of each lane's 16 anchors inside the block (low half): lets k_emit_seeds find the position of any anchor
		// of an earlier block from block_base[] + this + at most 15 flag bytes.  High half: the late bits of the lane's 16 anchors --
		// flags[] itself is left as the DP kernels wrote it (a byte store per late anchor was a read-modify-write of a whole
		// memory burst each: 1.1 GB of writes for 18 M late anchors on the 100k-read job, most of this kernel's time)
		unsigned int late = 0;
		for (int k = 0; k < 4; ++k) {
			const unsigned int w = flag_word(vv[b], k) & 0x01010101u;
			late |= ((w & 1u) | ((w >> 7) & 2u) | ((w >> 14) & 4u) | ((w >> 21) & 8u)) << (4 * k);
		}
		unsigned int incl = mine;
		for (int d = 1; d < 64; d <<= 1) { const unsigned int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
		sub[blk * 64 + lane] = (incl - mine) | late << 16;
		if (lane == 63) block_cnt[blk] = incl;
	}
}

// C3 + C4 in one pass (no id[] array in between): positions of a 1024-anchor block's records from the scanned block counts and
// the block's flag bytes, kept in LDS; records written from there.  What a record needs from OUTSIDE the block -- the position of
// a predecessor (or of its first child) in an earlier block, the offset of a read that started in an earlier block -- is
// recomputed from block_base[] and the flag bytes in front of it (a short loop, needed by about one anchor in a hundred).
__device__ __forceinline__ uint32_t pos_before(const uint8_t *__restrict__ flags, const unsigned long long *__restrict__ block_base,
                                               const uint32_t *__restrict__ sub, const uint64_t *__restrict__ single_mask,
                                               const uint64_t *__restrict__ emit_mask, int64_t x)
{
	const int64_t bx = x / CMP_PER_BLOCK, b0 = bx * CMP_PER_BLOCK;
	const int l16 = (int)((x - b0) >> 4);                            // k_count's lane: 16 anchors each
	const uint32_t sl = sub[bx * 64 + l16];
	const int64_t s0 = b0 + 16 * l16;
	uint32_t c = (uint32_t)block_base[bx] + (sl & 0xffffu) + (uint32_t)__builtin_popcount((sl >> 16) & ((1u << (int)(x - s0)) - 1u));   // late records
	const uint32_t *w = (const uint32_t*)(flags + s0);               // the array is 16-byte aligned and padded
	const uint32_t s16 = mask_bits16(single_mask, s0), e16 = mask_bits16(emit_mask, s0);
	const int nfull = (int)((x - s0) >> 2), tail = (int)((x - s0) & 3);
	for (int k = 0; k < nfull; ++k)                                  // own-step records
		c += (uint32_t)__builtin_popcount(flags_with_singles(w[k], (s16 >> (4 * k)) & 0xfu, (e16 >> (4 * k)) & 0xfu) & 0x02020202u);
	if (tail) c += (uint32_t)__builtin_popcount(flags_with_singles(w[nfull], (s16 >> (4 * nfull)) & 0xfu, (e16 >> (4 * nfull)) & 0xfu) & 0x02020202u & ((1u << (8 * tail)) - 1u));
	return c;
}
// anchor x's flag byte as the compaction sees it: a singleton's from the masks, the late bit from sub[] (k_count)
__device__ __forceinline__ uint32_t flag_with_late(const uint8_t *__restrict__ flags, const uint32_t *__restrict__ sub,
                                                   const uint64_t *__restrict__ single_mask, const uint64_t *__restrict__ emit_mask, int64_t x)
{
	if ((single_mask[x >> 6] >> (x & 63)) & 1u) return 0x20u | (((emit_mask[x >> 6] >> (x & 63)) & 1u) ? 0x0au : 0u);
	return ((uint32_t)flags[x] & ~1u) | ((sub[x >> 4] >> (16 + (int)(x & 15))) & 1u);     // (a block is 64 runs of 16: sub[] is indexed by x / 16)
}

__global__ __launch_bounds__(CMP_BLOCK) void k_emit_seeds(Params par, int64_t n_reads, int64_t total,
                                                          const int64_t *__restrict__ off, const ulonglong2 *__restrict__ a,
                                                          const int32_t *__restrict__ f, const int32_t *__restrict__ p,
                                                          const int32_t *__restrict__ v, const uint8_t *__restrict__ flags,
                                                          const int32_t *__restrict__ first_child,
                                                          const unsigned long long *__restrict__ block_base,
                                                          int64_t *__restrict__ seeds_off, SeedRec *__restrict__ seeds,
                                                          const int2 *__restrict__ block_reads, const uint32_t *__restrict__ sub,
                                                          const uint64_t *__restrict__ single_mask, const uint64_t *__restrict__ emit_mask)
{
	__shared__ uint32_t s_pos[CMP_PER_BLOCK];                        // position of the first record an anchor emits (were it to emit any)
	__shared__ uint8_t s_fl[CMP_PER_BLOCK];
	__shared__ uint32_t s_wsum[CMP_BLOCK / 64];
	__shared__ uint32_t s_so_lo;                                     // seeds_off of the block's first read
	const int64_t g0 = (int64_t)blockIdx.x * CMP_PER_BLOCK;
	const int64_t g1 = g0 + CMP_PER_BLOCK < total ? g0 + CMP_PER_BLOCK : total;
	int64_t rlo, rhi;
	block_read_range(block_reads, rlo, rhi);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int min_sc = par.min_sc;
	const int64_t g = g0 + 4 * (int64_t)threadIdx.x;                 // four consecutive anchors per thread
	uint32_t w = 0;
	int n = 0;
	if (g < g1) {
		const uint32_t l4 = (sub[g >> 4] >> (16 + (int)(g & 15))) & 0xfu;           // the late bits of the four (k_count)
		w = (*(const uint32_t*)(flags + g) & ~0x01010101u) | (l4 & 1u) | (l4 & 2u) << 7 | (l4 & 4u) << 14 | (l4 & 8u) << 21;
		w = flags_with_singles(w, mask_bits16(single_mask, g) & 0xfu, mask_bits16(emit_mask, g) & 0xfu);
		n = g1 - g < 4 ? (int)(g1 - g) : 4;
		if (n < 4) w &= (1u << (8 * n)) - 1u;
	}
	const uint32_t mine = (uint32_t)__builtin_popcount(w & 0x03030303u);
	uint32_t incl = mine;
	for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
	if (lane == 63) s_wsum[wave] = incl;
	if (threadIdx.x == 0) {
		const int64_t rs = off[rlo];
		s_so_lo = rs >= g0 ? 0xffffffffu : pos_before(flags, block_base, sub, single_mask, emit_mask, rs);    // (a read that starts in this block: from s_pos below)
	}
	__syncthreads();
	uint32_t pos = (uint32_t)block_base[blockIdx.x] + incl - mine;
	for (int k = 0; k < wave; ++k) pos += s_wsum[k];
	for (int e = 0; e < n; ++e) {
		const uint32_t fl = (w >> (8 * e)) & 0xffu;
		s_pos[4 * threadIdx.x + e] = pos; s_fl[4 * threadIdx.x + e] = (uint8_t)fl;
		pos += (fl & 1) + ((fl >> 1) & 1);
	}
	__syncthreads();
	// position of the first record of anchor x (batch-global index; x <= the block's last anchor)
	auto pos_of = [&](int64_t x) -> uint32_t { return x >= g0 ? s_pos[x - g0] : pos_before(flags, block_base, sub, single_mask, emit_mask, x); };
	auto flag_of = [&](int64_t x) -> uint32_t { return x >= g0 ? (uint32_t)s_fl[x - g0] : flag_with_late(flags, sub, single_mask, emit_mask, x); };
	// records: a thread per anchor, consecutive lanes on consecutive anchors (their records are consecutive too: the stores of a
	// wave are contiguous)
	for (int64_t ge = g0 + threadIdx.x; ge < g1; ge += CMP_BLOCK) {
		const uint32_t fl = s_fl[ge - g0];
		const uint32_t mypos = s_pos[ge - g0];
		const int64_t r = rlo == rhi ? rlo : read_of_c(off, rlo, rhi, ge);
		const int64_t rs = off[r];
		if (ge == rs) for (int64_t q = r; q >= 0 && off[q] == rs; --q) seeds_off[q] = (int64_t)mypos;   // (empty reads in front share the value)
		if (!(fl & 2)) continue;                                       // not emitted at its own step
		const uint32_t so = rs >= g0 ? s_pos[rs - g0] : (r == rlo ? s_so_lo : pos_before(flags, block_base, sub, single_mask, emit_mask, rs));
		const ulonglong2 ak = a[ge];
		int32_t q = p[ge], fk = f[ge];
		if (fl & 0x20u) { q = -1; fk = (int32_t)((uint32_t)(ak.y >> 32) & 0xffu); }   // a singleton (nobody stored its p and f): chain.c:251,283 with an empty window
		const uint32_t idk = mypos + (fl & 1);
		int32_t pfield = (int32_t)(0xfffffffcu);                       // (-1)<<2
		if (q >= 0) {
			const int64_t qg = rs + q;
			if (fl & 1) {                                              // late emission of q, chain.c:292-302
				const int32_t vq = v[qg], fq = f[qg];
				const ulonglong2 aq = a[qg];
				SeedRec rec;
				rec.x = aq.x; rec.y = aq.y; rec.f = fq;
				rec.p = (int32_t)(0xfffffffcu | (uint32_t)(vq >= min_sc) | ((uint32_t)(fq < vq) << 1));
				seeds[idk - 1] = rec;
			}
			// new index of q: its own record if it is emitted at its own step (behind the late record it may have triggered), else the
			// slot in front of its first child
			const uint32_t flq = flag_of(qg);
			const uint32_t idq = (fl & 1) ? idk - 1 : (flq & 2) ? pos_of(qg) + (flq & 1) : pos_of(rs + first_child[qg]);
			pfield = (int32_t)((idq - so) << 2);                       // chain.c:310, read-relative index
		}
		SeedRec rec;
		rec.x = ak.x; rec.y = ak.y; rec.f = fk;
		rec.p = pfield | ((fl >> 3) & 3);                              // chain.c:313-314: (v >= min_sc) | (f < v) << 1, from the DP kernel
		seeds[idk] = rec;
	}
}

// seeds_off of trailing empty reads and the end marker
__global__ void k_finish_offsets(int64_t n_reads, int64_t total, const int64_t *__restrict__ off,
                                 const unsigned long long *__restrict__ n_seeds, int64_t *__restrict__ seeds_off)
{
	const int64_t m = (int64_t)(uint32_t)*n_seeds;
	seeds_off[n_reads] = m;
	for (int64_t r = n_reads - 1; r >= 0 && off[r] == total; --r) seeds_off[r] = m;
}

size_t compact_scratch_bytes(int64_t max_anchors, size_t *flags_bytes, size_t *blocks_bytes)
{
	const size_t blocks = (size_t)(max_anchors + CMP_PER_BLOCK - 1) / CMP_PER_BLOCK;
	*flags_bytes = ((size_t)max_anchors + 15) & ~(size_t)15;
	*blocks_bytes = (blocks + 1) * 8;
	return *flags_bytes + 2 * *blocks_bytes + 8;
}

hipError_t launch_compact(hipStream_t st, const Params &par, int64_t n_reads, int64_t total, const int64_t *d_off,
                          const void *d_a, const int32_t *d_f, const int32_t *d_p, const int32_t *d_v,
                          int32_t *d_first_child, int64_t *d_seeds_off, void *d_seeds, CompactScratch sc)
{
	hipError_t e;
	if (n_reads <= 0) return hipMemsetAsync(d_seeds_off, 0, sizeof(int64_t), st);
	if (total <= 0) return hipMemsetAsync(d_seeds_off, 0, (size_t)(n_reads + 1) * sizeof(int64_t), st);
	const int64_t blocks = (total + CMP_PER_BLOCK - 1) / CMP_PER_BLOCK;
	const dim3 g((unsigned)blocks), b(CMP_BLOCK);
	const int64_t cw = (blocks + CNT_BLOCKS_PER_WAVE - 1) / CNT_BLOCKS_PER_WAVE;    // waves of k_count
	const dim3 gc((unsigned)((cw + CMP_BLOCK / 64 - 1) / (CMP_BLOCK / 64)));
	hipLaunchKernelGGL(k_count, gc, b, 0, st, n_reads, total, d_off, d_p, d_first_child, sc.flags, sc.block_cnt, sc.block_reads, sc.sub, sc.single_mask, sc.emit_mask);
	if ((e = launch_scan_u64(st, blocks, sc.block_cnt, sc.tile_tmp, sc.n_seeds)) != hipSuccess) return e;
	hipLaunchKernelGGL(k_emit_seeds, g, b, 0, st, par, n_reads, total, d_off, (const ulonglong2*)d_a, d_f, d_p, d_v, sc.flags, d_first_child,
	                   sc.block_cnt, d_seeds_off, (SeedRec*)d_seeds, sc.block_reads, sc.sub, sc.single_mask, sc.emit_mask);
	hipLaunchKernelGGL(k_finish_offsets, dim3(1), dim3(1), 0, st, n_reads, total, d_off, sc.n_seeds, d_seeds_off);
	return hipGetLastError();
}

} // namespace chaindp
