// chaindp_rsort.h -- the reference's radix_sort_128x (ksort.h:101-151) for one sequential GPU thread.
// Unstable; the order it gives equal keys is part of the reference's output (chain order in mm_chain_dp_bottom,
// anchor order in collect_seed_hits), so the procedure is followed step by step.
#ifndef CHAINDP_RSORT_H
#define CHAINDP_RSORT_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace chaindp {

__device__ inline void bt_insertion(ulonglong2 *beg, ulonglong2 *end)
{
	for (ulonglong2 *i = beg + 1; i < end; ++i) {
		if (i->x < (i - 1)->x) {
			const ulonglong2 tmp = *i;
			ulonglong2 *j = i;
			while (j > beg && tmp.x < (j - 1)->x) { *j = *(j - 1); --j; }
			*j = tmp;
		}
	}
}

struct BtRange { int32_t beg, end, shift; };

// stack: room for n / 65 + 2 pending ranges
__device__ inline void bt_radix_128x(ulonglong2 *a, int32_t n, BtRange *stack)
{
	if (n <= 64) { bt_insertion(a, a + n); return; }
	int32_t head[256], tail[256];
	int sp = 0;
	stack[sp++] = BtRange{0, n, 56};
	while (sp > 0) {
		const BtRange rg = stack[--sp];
		for (int d = 0; d < 256; ++d) tail[d] = 0;
		for (int32_t q = rg.beg; q < rg.end; ++q) ++tail[a[q].x >> rg.shift & 0xff];
		int32_t acc = rg.beg;
		for (int d = 0; d < 256; ++d) { head[d] = acc; acc += tail[d]; tail[d] = acc; }
		for (int d = 0; d < 256;) {                                 // cycle-leader permutation, buckets in ascending order
			if (head[d] != tail[d]) {
				int l = (int)(a[head[d]].x >> rg.shift & 0xff);
				if (l != d) {
					ulonglong2 carry = a[head[d]], swap;
					do {
						swap = carry; carry = a[head[l]]; a[head[l]++] = swap;
						l = (int)(carry.x >> rg.shift & 0xff);
					} while (l != d);
					a[head[d]++] = carry;
				} else ++head[d];
			} else ++d;
		}
		if (rg.shift) {
			const int32_t next = rg.shift > 8 ? rg.shift - 8 : 0;
			int32_t b = rg.beg;
			for (int d = 0; d < 256; ++d) {
				const int32_t e = tail[d];
				if (e - b > 64) stack[sp++] = BtRange{b, e, next};      // disjoint ranges: the order they are sorted in is immaterial
				else if (e - b > 1) bt_insertion(a + b, a + e);
				b = e;
			}
		}
	}
}

} // namespace chaindp
#endif
