// seed_oracle.cpp -- see seed_oracle.h (TEST INFRASTRUCTURE ONLY).  Every routine cites the reference lines it restates.
#include "seed_oracle.h"

#include <string.h>

namespace seedoracle {

namespace {

// reference bit layout of a position word (map.c:208-214, the fork's own packing):
//   63..43 reference id | 42..22 reference position | 21 strand | 20..0 rank id
const uint64_t P_STRAND = 1ull << 21;                 // mmpriv.h:20
const uint64_t SEED_TANDEM = 1ull << 42, SEED_SELF = 1ull << 43;   // mmpriv.h:18-19
const int SEED_SEG_SHIFT = 48;                        // mmpriv.h:22
const int F_NO_DIAG = 0x001, F_NO_DUAL = 0x002, F_FOR_ONLY = 0x100000, F_REV_ONLY = 0x200000;   // minimap.h:8-9,28-29

inline uint64_t load_u64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint32_t load_u32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t load_u48(const uint8_t *p) { uint64_t v = 0; memcpy(&v, p, 6); return v; }

struct Match {                 // mm_match_t (map.c:104-109)
	uint32_t n, q_pos, q_span, seg_id;
	bool is_tandem;
	const uint64_t *cr;
};

} // namespace

void IndexImage::append(int type, const void *data, size_t bytes)
{
	std::vector<uint8_t> *dst = type == 4 ? &B_ : type == 5 ? &H_ : type == 6 ? &V_ : type == 7 ? &P_ : nullptr;
	if (!dst || !data || bytes == 0) return;
	// a B chunk after the rest of an image has arrived starts the image of the next index part (main.c:201-204
	// sends B, H, V, P in this order for every part)
	if (type == 4 && (!H_.empty() || !V_.empty() || !P_.empty())) clear();
	const uint8_t *p = (const uint8_t*)data;
	dst->insert(dst->end(), p, p + bytes);
	if (type == 4) seal();
}

void IndexImage::clear()
{
	B_.clear(); H_.clear(); V_.clear(); P_.clear();
	b_bits_ = -1;
}

// The B blob has one 16-byte entry per bucket and 2^b buckets; its last chunk is padded to 64 bytes
// (index.c:113), which cannot add a power of two, so b is the floor of log2(entries).
void IndexImage::seal()
{
	size_t entries = B_.size() / 16;
	int b = 0;
	while ((size_t)2 << b <= entries) ++b;
	b_bits_ = entries ? b : 0;
}

const uint64_t *IndexImage::get(uint64_t minier, int *n) const
{
	*n = 0;
	if (B_.empty() || b_bits_ < 0) return nullptr;
	const uint64_t mask = (1ull << b_bits_) - 1;
	const uint8_t *be = B_.data() + (minier & mask) * 16;
	const uint64_t w0 = load_u64(be), w1 = load_u64(be + 8);
	const uint32_t n_buckets = (uint32_t)(w0 >> 24);
	if (n_buckets == 0) return nullptr;                               // index.c:227: bucket without a hash table
	const uint64_t h_off = w1 >> 28, p_off = (w1 & ((1ull << 28) - 1)) << 8 | w0 >> 56;
	const uint64_t key = minier >> b_bits_ << 1;                       // index.c:229
	// kh_get (khash.h:218-231) with idx_hash(a) = a >> 1, idx_eq(a, b) = (a >> 1 == b >> 1) (index.c:23-25)
	const uint32_t m = n_buckets - 1;
	uint32_t i = (uint32_t)(key >> 1) & m, step = 0;
	const uint32_t last = i;
	for (;;) {
		const uint64_t slot = h_off + i;
		const uint8_t *grp = H_.data() + (slot >> 3) * 64;
		if ((size_t)(grp - H_.data()) + 64 > H_.size()) return nullptr;   // truncated image
		const uint32_t fl = (load_u32(grp) >> ((i & 0xfu) << 1)) & 3u;    // bit1 empty, bit0 deleted (khash.h:166-168)
		const uint64_t k48 = load_u48(grp + 4 + (slot & 7) * 6);
		if (fl & 2u) return nullptr;                                      // empty slot: not present
		if (!(fl & 1u) && (k48 >> 1) == ((key & 0xffffffffffffull) >> 1)) {
			if ((slot + 1) * 8 > V_.size()) return nullptr;
			const uint64_t *val = (const uint64_t*)(V_.data() + slot * 8);
			if (k48 & 1) { *n = 1; return val; }                          // index.c:231-233: a single position lives in the value
			const uint64_t v = load_u64((const uint8_t*)val);
			const uint64_t first = p_off + (v >> 32);
			const uint32_t cnt = (uint32_t)v;
			if ((first + cnt) * 8 > P_.size()) return nullptr;
			*n = (int)cnt;
			return (const uint64_t*)(P_.data() + first * 8);             // index.c:235-236
		}
		i = (i + (++step)) & m;
		if (i == last) return nullptr;
	}
}

// Lookups are four dependent cache misses (B entry, hash group, value, positions) into an image far larger than
// the caches; the caller walks a read's minimizers in order, so the first two levels are prefetched a few
// minimizers ahead.
void IndexImage::prefetch(uint64_t minier, int level) const
{
	if (B_.empty() || b_bits_ < 0) return;
	const uint8_t *be = B_.data() + (minier & ((1ull << b_bits_) - 1)) * 16;
	if (level == 0) { __builtin_prefetch(be); return; }
	const uint64_t w0 = load_u64(be), w1 = load_u64(be + 8);
	const uint32_t n_buckets = (uint32_t)(w0 >> 24);
	if (n_buckets == 0) return;
	const uint64_t slot = (w1 >> 28) + ((uint32_t)(minier >> b_bits_) & (n_buckets - 1));
	if ((slot >> 3) * 64 + 64 <= H_.size()) __builtin_prefetch(H_.data() + (slot >> 3) * 64);
	if ((slot + 1) * 8 <= V_.size()) __builtin_prefetch(V_.data() + slot * 8);
}

// ---- radix_sort_128x, ksort.h:101-151 with rskey = x, sizeof_key = 8, RS_MIN_SIZE 64, RS_MAX_BITS 8

static void insertion_sort_x(U128 *beg, U128 *end)                   // ksort.h:107-117
{
	for (U128 *i = beg + 1; i < end; ++i) {
		if (i->x < (i - 1)->x) {
			U128 tmp = *i, *j;
			for (j = i; j > beg && tmp.x < (j - 1)->x; --j) *j = *(j - 1);
			*j = tmp;
		}
	}
}

static void flag_sort_x(U128 *beg, U128 *end, int shift)             // ksort.h:118-145, n_bits = 8
{
	struct Bucket { U128 *b, *e; } bk[256];
	for (int k = 0; k < 256; ++k) bk[k].b = bk[k].e = beg;
	for (U128 *i = beg; i != end; ++i) ++bk[i->x >> shift & 255].e;       // counts, kept as end pointers
	for (int k = 1; k < 256; ++k) { bk[k].e += bk[k - 1].e - beg; bk[k].b = bk[k - 1].e; }
	for (int k = 0; k < 256;) {                                           // cycle-leader permutation
		if (bk[k].b != bk[k].e) {
			int l = (int)(bk[k].b->x >> shift & 255);
			if (l != k) {
				U128 tmp = *bk[k].b, swap;
				do {
					swap = tmp; tmp = *bk[l].b; *bk[l].b++ = swap;
					l = (int)(tmp.x >> shift & 255);
				} while (l != k);
				*bk[k].b++ = tmp;
			} else ++bk[k].b;
		} else ++k;
	}
	bk[0].b = beg;
	for (int k = 1; k < 256; ++k) bk[k].b = bk[k - 1].e;
	if (shift) {
		const int next = shift > 8 ? shift - 8 : 0;
		for (int k = 0; k < 256; ++k) {
			if (bk[k].e - bk[k].b > 64) flag_sort_x(bk[k].b, bk[k].e, next);
			else if (bk[k].e - bk[k].b > 1) insertion_sort_x(bk[k].b, bk[k].e);
		}
	}
}

void radix_sort_128x(U128 *beg, U128 *end)                            // ksort.h:146-150
{
	if (end - beg <= 64) insertion_sort_x(beg, end);
	else flag_sort_x(beg, end, 56);
}

// ---- collect_seed_hits

// skip_seed, map.c:148-185.  Note the test at map.c:152, `1 & flag & (NO_DIAG|NO_DUAL)`: only bit 0 (NO_DIAG)
// opens the block, as written.
static inline bool skip_seed(int flag, uint64_t r, const Match &q, uint32_t bid, bool *is_self)
{
	*is_self = false;
	if (1 & flag & (F_NO_DIAG | F_NO_DUAL)) {
		const uint32_t rank_id = (uint32_t)r & 0x1FFFFFu;
		const int flg = (int)((bid & 0x80000000u) >> 31);
		const uint32_t val = bid & 0x7fffffffu;
		int cmp;
		if (val > rank_id) cmp = 1;
		else if (val < rank_id) cmp = -1;
		else cmp = flg ? 0 : -1;
		if ((flag & F_NO_DIAG) && cmp == 0) {
			if (((r >> 22) & 0x1fffff) == (q.q_pos >> 1)) return true;           // the diagonal itself
			if (((r & P_STRAND) >> 21) == (q.q_pos & 1)) *is_self = true;
		}
		if ((flag & F_NO_DUAL) && cmp > 0) return true;                          // all-vs-all: map once
	}
	if (flag & (F_FOR_ONLY | F_REV_ONLY)) {
		if (((r & P_STRAND) >> 21) == (q.q_pos & 1)) { if (flag & F_REV_ONLY) return true; }
		else { if (flag & F_FOR_ONLY) return true; }
	}
	return false;
}

void collect_seed_hits(const IndexImage &idx, int flag, int max_occ, const U128 *mv, size_t mv_n, uint32_t bid, int qlen,
                       std::vector<U128> &a, int *rep_len, std::vector<uint64_t> &mini_pos)
{
	// collect_matches, map.c:112-146
	std::vector<Match> m;
	m.reserve(mv_n);
	mini_pos.clear();
	int rep_st = 0, rep_en = 0;
	size_t n_a = 0;
	*rep_len = 0;
	for (size_t i = 0; i < mv_n; ++i) {
		const U128 &p = mv[i];
		const uint32_t q_pos = (uint32_t)p.y, q_span = (uint32_t)(p.x & 0xff);
		int t;
		if (i + 16 < mv_n) idx.prefetch(mv[i + 16].x >> 8, 0);
		if (i + 8 < mv_n) idx.prefetch(mv[i + 8].x >> 8, 1);
		const uint64_t *cr = idx.get(p.x >> 8, &t);
		if (t >= max_occ) {                                            // too frequent: only its span counts, as repetitive
			const int en = (int)(q_pos >> 1) + 1, st = en - (int)q_span;
			if (st > rep_en) { *rep_len += rep_en - rep_st; rep_st = st; rep_en = en; }
			else rep_en = en;
		} else {
			Match q;
			q.q_pos = q_pos; q.q_span = q_span; q.cr = cr; q.n = (uint32_t)t;
			q.seg_id = (uint32_t)(p.y >> 32) & 0x7fffffffu;
			q.is_tandem = (i > 0 && p.x >> 8 == mv[i - 1].x >> 8) || (i + 1 < mv_n && p.x >> 8 == mv[i + 1].x >> 8);
			n_a += q.n;
			mini_pos.push_back((uint64_t)q_span << 32 | q_pos >> 1);
			m.push_back(q);
		}
	}
	*rep_len += rep_en - rep_st;
	// map.c:197-231
	a.clear();
	a.reserve(n_a);
	for (const Match &q : m) {
		for (uint32_t k = 0; k < q.n; ++k) {
			const uint64_t r = q.cr[k];
			const uint64_t rpos = (r >> 22) & 0x1fffff;
			bool is_self;
			if (skip_seed(flag, r, q, bid, &is_self)) continue;
			U128 s;
			if (((r & P_STRAND) >> 21) == (q.q_pos & 1)) {             // forward strand
				s.x = ((r & 0xfffff80000000000ull) >> 11) | rpos;
				s.y = (uint64_t)q.q_span << 32 | q.q_pos >> 1;
			} else {                                                   // reverse strand; the query coordinate is 32-bit unsigned arithmetic
				s.x = 1ull << 63 | ((r & 0xfffff80000000000ull) >> 11) | rpos;
				s.y = (uint64_t)q.q_span << 32 | (uint32_t)((uint32_t)qlen - ((q.q_pos >> 1) + 1 - q.q_span) - 1);
			}
			s.y |= (uint64_t)q.seg_id << SEED_SEG_SHIFT;
			if (q.is_tandem) s.y |= SEED_TANDEM;
			if (is_self) s.y |= SEED_SELF;
			a.push_back(s);
		}
	}
	radix_sort_128x(a.data(), a.data() + a.size());                    // map.c:233
}

} // namespace seedoracle

// ---- C entry points for tests/oracle_lib.py -------------------------------------------------------------------

extern "C" void *so_index_create(const void *B, size_t nB, const void *H, size_t nH, const void *V, size_t nV, const void *P, size_t nP)
{
	seedoracle::IndexImage *ix = new seedoracle::IndexImage();
	ix->append(4, B, nB); ix->append(5, H, nH); ix->append(6, V, nV); ix->append(7, P, nP);
	if (!ix->complete()) { delete ix; return nullptr; }
	return ix;
}

extern "C" void so_index_destroy(void *ix) { delete (seedoracle::IndexImage*)ix; }

// collect_seed_hits for one read.  Returns 0, or -2 when cap_anchors is too small (*n_anchors holds the need).
extern "C" int so_collect_seed_hits(const void *ix, int flag, int max_occ, uint32_t bid, int qlen, const void *mini, int64_t n_mini,
                                    void *anchors, int64_t cap_anchors, int64_t *n_anchors, int *rep_len, uint64_t *mini_pos, int *n_mini_pos)
{
	std::vector<seedoracle::U128> a;
	std::vector<uint64_t> mp;
	int rl = 0;
	seedoracle::collect_seed_hits(*(const seedoracle::IndexImage*)ix, flag, max_occ, (const seedoracle::U128*)mini, (size_t)(n_mini > 0 ? n_mini : 0),
	                              bid, qlen, a, &rl, mp);
	if (n_anchors) *n_anchors = (int64_t)a.size();
	if (rep_len) *rep_len = rl;
	if (n_mini_pos) *n_mini_pos = (int)mp.size();
	if (mini_pos && !mp.empty()) memcpy(mini_pos, mp.data(), mp.size() * sizeof(uint64_t));
	if ((int64_t)a.size() > cap_anchors) return -2;
	if (anchors && !a.empty()) memcpy(anchors, a.data(), a.size() * 16);
	return 0;
}
