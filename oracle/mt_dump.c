/*
 * mt_dump.c -- anchor dumper driving the UNMODIFIED reference front half.
 * TEST INFRASTRUCTURE ONLY; build-container only (needs /root/reference),
 * built by `make -C oracle ref-dump` into oracle/_ref/mt_dump.
 *
 * The reference names a dump branch (README.md:10) that is not in the tree, so
 * this harness plays its role for BASELINE.json config 1: it runs the
 * reference's own  sketch -> index -> collect_seed_hits -> radix_sort_128x
 * (map.c:87-99, map.c:187-236, index.c:541) on a query/target FASTA pair and
 * writes, per query read, the sorted mm128_t anchor array that
 * mm_chain_dp_fpga (chain.c:218) would receive, plus the per-read DP arguments.
 *
 * collect_seed_hits/collect_minimizers are `static` in map.c, so map.c is
 * #included from where it lies.  The eleven fpga_* driver symbols map.c and
 * index.c reference (libfpgadrv.a is absent) are never CALLED on this path; the
 * link leaves them unresolved (-Wl,--unresolved-symbols=ignore-all) rather than
 * providing stand-ins.
 *
 * Output (little endian): magic "ANCHDMP1", int32 n_reads, then per read:
 *   int32 max_dist_x (gap_ref), int32 max_dist_y (gap_qry), int32 bw, int32 max_skip,
 *   int32 min_sc, int32 is_cdna, int32 n_segs, int32 min_cnt, int64 n, n * mm128_t.
 *
 * With a fifth argument it also writes what the reference hands to its device for the seed-collection step
 * (SURVEY row N2), so that a restatement of collect_seed_hits over the device's index image can be pinned:
 *   magic "SEEDDMP1"; int32 flag, int32 mid_occ, int32 n_reads; the index image the reference serialises for the
 *   FPGA (index.c:603-720; what main.c:201-204 sends through fpga_load_index as types 4..7), as four blobs
 *   (int64 bytes, data) in the order B, H, V, P; then per read: uint32 bid, int32 qlen, int64 n_mini, n_mini * mm128_t
 *   minimizers (map.c:352), int64 n_a, n_a * mm128_t sorted anchors (collect_seed_hits' result), int32 rep_len,
 *   int32 n_mini_pos, n_mini_pos * uint64 mini_pos.
 */
#include "map.c"

int main(int argc, char **argv)
{
	mm_idxopt_t io;
	mm_mapopt_t mo;
	mm_idx_reader_t *rd;
	mm_idx_t *mi;
	mm_bseq_file_t *fp;
	mm_bseq1_t *seqs;
	FILE *out, *sd = 0;
	int n_seq = 0, i, n_written = 0;
	long pos_n;

	if (argc < 5) {
		fprintf(stderr, "usage: mt_dump <preset> <target.fa> <query.fa> <out.bin>\n");
		return 2;
	}
	mm_verbose = 0;
	mm_set_opt(0, &io, &mo);
	if (mm_set_opt(argv[1], &io, &mo) < 0) { fprintf(stderr, "unknown preset\n"); return 2; }
	rd = mm_idx_reader_open(argv[2], &io, 0);
	if (!rd) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
	mi = mm_idx_reader_read(rd, 1);
	mm_mapopt_update(&mo, mi);
	g_B = mi->B, g_b = mi->b;                       /* map.c:700-701 */

	fp = mm_bseq_open(argv[3]);
	seqs = mm_bseq_read(fp, 1 << 30, 0, &n_seq);
	out = fopen(argv[4], "wb");
	fwrite("ANCHDMP1", 1, 8, out);
	pos_n = ftell(out);
	fwrite(&n_written, 4, 1, out);
	if (argc > 5) {
		idx_buf_t *img[4] = { mi->b_idx, mi->h_idx, mi->v_idx, mi->p_idx };
		int32_t h3[3] = { mo.flag, mo.mid_occ, 0 };
		int k;
		sd = fopen(argv[5], "wb");
		fwrite("SEEDDMP1", 1, 8, sd);
		fwrite(h3, 4, 3, sd);
		for (k = 0; k < 4; ++k) {
			int64_t nb = (int64_t)img[k]->pos;
			fwrite(&nb, 8, 1, sd);
			fwrite(img[k]->buf, 1, nb, sd);
		}
	}

	for (i = 0; i < n_seq; ++i) {
		const char *s = seqs[i].seq;
		int qlen = seqs[i].l_seq, rep_len = 0, n_mini_pos = 0;
		int32_t hdr[8];
		int64_t n_a = 0;
		uint64_t *mini_pos = 0;
		mm128_v mv = {0, 0, 0};
		mm128_t *a;
		unsigned int bid = dichotomy_sort(seqs[i].name, mi->rname_rid, mi->n_seq);   /* map.c:350 */
		int is_sr = !!(mo.flag & MM_F_SR), gap_qry, gap_ref;
		collect_minimizers(NULL, &mo, mi, 1, &qlen, &s, &mv);                          /* map.c:352 */
		if (mv.n == 0) continue;
		/* map.c:358-366 */
		if (is_sr) gap_qry = qlen > mo.max_gap ? qlen : mo.max_gap; else gap_qry = mo.max_gap;
		if (mo.max_gap_ref > 0) gap_ref = mo.max_gap_ref;
		else if (mo.max_frag_len > 0) { gap_ref = mo.max_frag_len - qlen; if (gap_ref < mo.max_gap) gap_ref = mo.max_gap; }
		else gap_ref = mo.max_gap;
		a = collect_seed_hits(mo.flag, mo.mid_occ, mv.a, mv.n, bid, qlen, &n_a, &rep_len, &n_mini_pos, &mini_pos); /* map.c:523 */
		hdr[0] = gap_ref, hdr[1] = gap_qry, hdr[2] = mo.bw, hdr[3] = mo.max_chain_skip;
		hdr[4] = mo.min_chain_score, hdr[5] = !!(mo.flag & MM_F_SPLICE), hdr[6] = 1, hdr[7] = mo.min_cnt;
		fwrite(hdr, 4, 8, out);
		fwrite(&n_a, 8, 1, out);
		fwrite(a, sizeof(mm128_t), n_a, out);
		if (sd) {
			int64_t nm = mv.n;
			int32_t t2[2] = { rep_len, n_mini_pos };
			fwrite(&bid, 4, 1, sd); fwrite(&qlen, 4, 1, sd);
			fwrite(&nm, 8, 1, sd); fwrite(mv.a, sizeof(mm128_t), mv.n, sd);
			fwrite(&n_a, 8, 1, sd); fwrite(a, sizeof(mm128_t), n_a, sd);
			fwrite(t2, 4, 2, sd); fwrite(mini_pos, 8, n_mini_pos, sd);
		}
		++n_written;
		free(a); free(mini_pos); kfree(0, mv.a);
	}
	fseek(out, pos_n, SEEK_SET);
	fwrite(&n_written, 4, 1, out);
	fclose(out);
	if (sd) { fseek(sd, 16, SEEK_SET); fwrite(&n_written, 4, 1, sd); fclose(sd); }
	fprintf(stderr, "mt_dump: %d reads written\n", n_written);
	return 0;
}
