// seed_oracle.h -- CPU restatement of the reference's seed collection (collect_seed_hits, map.c:112-236, with
// mm_idx_get, index.c:221-238, over the FPGA index image of index.c:603-720, and radix_sort_128x, ksort.h:101-151).
//
// TEST INFRASTRUCTURE ONLY, like the rest of oracle/: tests/ load libseedoracle.so as the checker of the GPU's seed
// collection (minimap2_chaindp_amd/csrc/chaindp_seed.hip); the product never links or calls it.
// Parity status: PINNED against the reference's own collect_seed_hits results (tests/golden/seeds/*.npz, produced by
// oracle/mt_dump.c from the unmodified reference): tests/test_seed_collect.py.
#ifndef SEED_ORACLE_H
#define SEED_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include <vector>

namespace seedoracle {

struct U128 { uint64_t x, y; };   // mm128_t (minimap.h:48)

// The four blobs of the image, as written by index.c:603-720:
//   B: per hash bucket 16 bytes: w0 = (p_off & 0xff) << 56 | n_buckets << 24;  w1 = h_off << 28 | p_off >> 8
//      (h_off in hash slots, rounded up to 8 per bucket; p_off in entries of P; an empty bucket is all zero)
//   H: per 8 hash slots 64 bytes: 4 B khash flag word (2 bits per slot, 16 slots), 8 x 6 B keys (low 48 bits), 12 B pad
//   V: per hash slot 8 bytes: the khash value (a position if the key's bit 0 is set, else p index << 32 | count)
//   P: 8 bytes per position
class IndexImage {
public:
	void append(int type, const void *data, size_t bytes);   // a blob or a chunk of one, type 4..7 = B, H, V, P (fpga.h:20-23)
	bool complete() const { return !B_.empty() && !H_.empty() && !V_.empty(); }
	void clear();
	// mm_idx_get (index.c:221-238) over the image: positions of minimizer `minier`, *n of them (0 if absent)
	const uint64_t *get(uint64_t minier, int *n) const;
	void prefetch(uint64_t minier, int level) const;   // level 0: bucket entry; 1: first hash group and value
	const std::vector<uint8_t> &blob(int k) const { return k == 0 ? B_ : k == 1 ? H_ : k == 2 ? V_ : P_; }   // 0..3 = B, H, V, P
private:
	void seal();
	std::vector<uint8_t> B_, H_, V_, P_;
	int b_bits_ = -1;
};

// collect_seed_hits (map.c:187-236, with collect_matches map.c:112-146 and skip_seed map.c:148-185): the sorted
// anchors of one read, its repetitive length and the positions of the minimizers that were used.
void collect_seed_hits(const IndexImage &idx, int flag, int max_occ, const U128 *mv, size_t mv_n, uint32_t bid, int qlen,
                       std::vector<U128> &a, int *rep_len, std::vector<uint64_t> &mini_pos);

// radix_sort_128x (ksort.h:101-151 instantiated at misc.c:136): in-place MSD radix sort on x, 8 bits a level,
// insertion sort below 65 elements.  Unstable; the order of equal keys is part of the result the DP sees, so the
// procedure is followed step by step.
void radix_sort_128x(U128 *beg, U128 *end);

} // namespace seedoracle
#endif
