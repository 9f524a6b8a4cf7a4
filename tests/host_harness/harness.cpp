// tests/host_harness/harness.cpp — TEST INFRASTRUCTURE ONLY.
// Drives the product's HOST mapper (winnowmap_amd/csrc/host/*.cpp: index, MCAS fibers, hit/align glue) with a
// DeviceOps implementation backed by the ORACLE (oracle/wm_oracle.c), so that the host logic can be compared with
// the real reference end-to-end on a machine without a GPU. The product never links this file.
#include "../../winnowmap_amd/csrc/host/wm_core.cpp"
#include "../../winnowmap_amd/csrc/host/wm_index.cpp"
#include "../../winnowmap_amd/csrc/host/wm_seqio.cpp"
#include "../../winnowmap_amd/csrc/host/wm_hit.cpp"
#include "../../winnowmap_amd/csrc/host/wm_align.cpp"
#include "../../winnowmap_amd/csrc/host/wm_mapper.cpp"
#include "../../winnowmap_amd/csrc/host/wm_chain.cpp"
#include "../../winnowmap_amd/csrc/host/wm_ops.cpp"
#include "../../winnowmap_amd/csrc/host/wm_format.cpp"
#include "../../winnowmap_amd/csrc/host/wm_pipeline.cpp"
#include "../../oracle/wm_oracle.h"
#include "../../winnowmap_amd/csrc/reads2bit.h"
#include <fstream>

using namespace wm;

struct OracleOps : DeviceOps {
	const Index *idx; wmo_bloom_t *bloom; const MapOpt *opt;
	int max_inflight() const override { return 64; }             // stateless CPU calls: any number may run at once
	// "resident" data like the device implementation keeps it: the batch's read codes (and the index's packed reference). Requests are
	// served from their host views; the resident positions they carry are decoded as well and must give the same bytes.
	// (kept the way the device keeps them: 2 bits per base + an ambiguity bitmap, packed by the product's wm_pack_codes — csrc/reads2bit.h)
	// One slab per slot, like the device's (two mapping calls may be in flight on one ops object: wm_map_file's lanes); a call's offsets start at slot * SLAB.
	static constexpr int64_t SLAB = (int64_t)1 << 40;
	std::vector<uint64_t> rd_pk[4], rd_nm[4]; size_t rd_n[4] = { 0, 0, 0, 0 };
	std::atomic<long> n_pos_checked{0}, n_pos_bad{0};
	bool load_reads(const uint8_t *codes, size_t n, int slot, int64_t *base) override
	{
		if (slot < 0 || slot >= 4) { *base = 0; return false; }
		rd_pk[slot].assign(wm_pk_words(n), ~0ULL); rd_nm[slot].assign(wm_nm_words(n), ~0ULL);
		wm_pack_codes(codes, n, rd_pk[slot].data(), rd_nm[slot].data());
		rd_n[slot] = n; *base = (int64_t)slot * SLAB;
		return true;
	}
	bool rd_in(int64_t p, int64_t len) const { const int64_t sl = p / SLAB, o = p % SLAB; return p >= 0 && sl < 4 && (size_t)(o + len) <= rd_n[sl]; }
	uint8_t rd_at(int64_t p) const { const int64_t sl = p / SLAB; return (uint8_t)wm_rd_code(rd_pk[sl].data(), rd_nm[sl].data(), (uint64_t)(p % SLAB)); }
	uint8_t two_strand(const KswReq &r, int64_t p) const
	{   // KswReq: [0,L) forward strand, [L,2L) reverse complement, negative = N padding
		const int64_t L = r.qwin_len;
		if (p < 0 || p >= 2 * L) return 4;
		const uint8_t c = p < L ? rd_at(r.qwin_off + p) : rd_at(r.qwin_off + (2 * L - 1 - p));
		return p < L ? c : (c < 4 ? 3 - c : 4);
	}
	void check_positions(const KswReq &r, const std::vector<uint8_t> &q, const std::vector<uint8_t> &t)
	{
		if (!r.resident()) return;
		bool bad = false, any_n = false;
		for (int i = 0; i < r.ql; ++i) { bad |= two_strand(r, (int64_t)r.q_pos + (int64_t)i * r.step) != q[i]; any_n |= q[i] >= 4; }
		std::vector<uint8_t> tt(r.tl > 0 ? r.tl : 0);
		if (r.tl > 0) {
			const int lo = r.step > 0 ? r.t_pos : r.t_pos - (r.tl - 1);
			idx->getseq(r.rid, lo, lo + r.tl, tt.data());
			for (int i = 0; i < r.tl; ++i) { bad |= tt[r.step > 0 ? i : r.tl - 1 - i] != t[i]; any_n |= t[i] >= 4; }
		}
		if (any_n && !r.has_n) bad = true;                        // has_n may be conservative, never optimistic
		++n_pos_checked; if (bad) ++n_pos_bad;
	}
	void sketch_batch(int w, int k, std::vector<SketchReq*> &reqs) override
	{
		for (SketchReq *r : reqs) {
			if (r->dev_off >= 0) { ++n_pos_checked; bool bad = !rd_in(r->dev_off, r->len); for (int64_t t = 0; !bad && t < r->len; ++t) bad = rd_at(r->dev_off + t) != r->seq[t]; if (bad) ++n_pos_bad; }
			std::vector<uint64_t> x(r->len + 8), y(r->len + 8);
			int64_t n = (idx->flag & 1 ? wmo_sketch_hpc : wmo_sketch)((const char*)r->seq, r->len, w, k, 0, bloom, x.data(), y.data(), r->len + 8);
			r->mini.resize(n);
			for (int64_t i = 0; i < n; ++i) r->mini[i].x = x[i], r->mini[i].y = y[i];
		}
	}
	void seed_batch(std::vector<SeedReq*> &reqs) override
	{   // collect_matches + collect_seed_hits restated for the test (src/map.c:97-130, 222-254)
		for (SeedReq *r : reqs) {
			int rep_st = 0, rep_en = 0, rep_len = 0;
			std::vector<wmo128_t> a;
			for (int i = 0; i < r->n_mini; ++i) {
				const m128 &p = r->mini[i];
				const uint32_t q_pos = (uint32_t)p.y, q_span = p.x & 0xff;
				int t;
				const uint64_t *cr = idx->get(p.x >> 8, &t);
				if (t >= r->max_occ) {
					int en = (q_pos >> 1) + 1, st = en - q_span;
					if (st > rep_en) { rep_len += rep_en - rep_st; rep_st = st, rep_en = en; } else rep_en = en;
					continue;
				}
				bool tandem = false;
				if (i > 0 && p.x >> 8 == r->mini[i - 1].x >> 8) tandem = true;
				if (i < r->n_mini - 1 && p.x >> 8 == r->mini[i + 1].x >> 8) tandem = true;
				for (int k = 0; k < t; ++k) {
					const uint64_t rr = cr[k];
					const int32_t rpos = (uint32_t)rr >> 1;
					const bool fwd = (rr & 1) == (q_pos & 1);
					if ((r->flag & F_FOR_ONLY) && !fwd) continue;
					if ((r->flag & F_REV_ONLY) && fwd) continue;
					wmo128_t e;
					if (fwd) { e.x = (rr & 0xffffffff00000000ULL) | (uint64_t)rpos; e.y = (uint64_t)q_span << 32 | q_pos >> 1; }
					else { e.x = 1ULL << 63 | (rr & 0xffffffff00000000ULL) | (uint64_t)rpos; e.y = (uint64_t)q_span << 32 | (uint32_t)(r->qlen - ((q_pos >> 1) + 1 - q_span) - 1); }
					if (tandem) e.y |= SEED_TANDEM;
					a.push_back(e);
				}
			}
			rep_len += rep_en - rep_st;
			wmo_radix_sort_128x(a.data(), a.data() + a.size());
			r->a.resize(a.size());
			for (size_t i = 0; i < a.size(); ++i) r->a[i].x = a[i].x, r->a[i].y = a[i].y;
			r->rep_len = rep_len;
		}
	}
	void chain_batch(std::vector<ChainReq*> &reqs) override
	{
		for (ChainReq *r : reqs) {
			const int64_t n = (int64_t)r->a.size();
			std::vector<wmo128_t> b(n ? n : 1);
			std::vector<uint64_t> u(n ? n : 1);
			int n_u = 0;
			if (const char *dp = getenv("WM_CHAIN_DUMP")) {    // probes: append the job (n, 8 ints + gap_scale, anchors) to a file
				FILE *fp = fopen(dp, "ab");
				int64_t hdr[2] = { n, 0 };
				int32_t par[10] = { r->max_dist_x, r->min_dist_x, r->max_dist_y, r->bw, r->max_skip, r->max_iter, r->min_cnt, r->min_sc, 0, 0 };
				memcpy(&par[8], &r->gap_scale, 4);
				fwrite(hdr, 8, 2, fp); fwrite(par, 4, 10, fp); fwrite(r->a.data(), 16, n, fp); fclose(fp);
			}
			extern int64_t wmo_chain_stat[4];
			wmo_chain_stat[0] = wmo_chain_stat[1] = wmo_chain_stat[2] = wmo_chain_stat[3] = 0;
			wmo_chain_set_cdna(r->is_cdna ? 1 : 0);
			int64_t n_v = wmo_chain_dp(r->max_dist_x, r->min_dist_x, r->max_dist_y, r->bw, r->max_skip, r->max_iter, r->min_cnt, r->min_sc, r->gap_scale,
			                           n, (const wmo128_t*)r->a.data(), &n_u, u.data(), b.data());
			if (getenv("WM_CHAIN_STATS")) fprintf(stderr, "CHAINJOB %lld %lld %lld %lld %lld\n", (long long)n, (long long)wmo_chain_stat[0], (long long)wmo_chain_stat[1], (long long)wmo_chain_stat[2], (long long)wmo_chain_stat[3]);
			r->u.assign(u.begin(), u.begin() + n_u);
			r->a.resize(n_v);
			for (int64_t i = 0; i < n_v; ++i) r->a[i].x = b[i].x, r->a[i].y = b[i].y;
		}
	}
	void ksw_batch(const wm_ksw_score_t &sc, std::vector<KswReq*> &reqs) override
	{
		int8_t mat[25];
		for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) mat[i * 5 + j] = (i == 4 || j == 4) ? sc.sc_ambi : i == j ? sc.match : sc.mismatch;
		wm::parallel_for(4, reqs.size(), [&](size_t ri) {
			KswReq *r = reqs[ri];
			wmo_ez_t ez;
			if (getenv("WM_KSW_STATS")) fprintf(stderr, "KSWJOB %d %d %d %d %d\n", r->ql, r->tl, r->w, r->zdrop, r->flag);
			std::vector<uint8_t> q(r->ql > 0 ? r->ql : 0), t(r->tl > 0 ? r->tl : 0);
			r->copy_query(q.data()); r->copy_target(t.data());
			check_positions(*r, q, t);
			if (const char *dp = getenv("WM_DUMP_KSW")) {        // real alignment jobs of a mapping run, for replays through the kernel emulator
				static std::mutex mu;
				std::lock_guard<std::mutex> lk(mu);
				if (FILE *f = fopen(dp, "ab")) {
					const int32_t hdr[10] = { (int32_t)q.size(), (int32_t)t.size(), r->w, r->zdrop, r->end_bonus, r->flag, sc.match, sc.mismatch, sc.q | sc.e << 8 | sc.q2 << 16 | sc.e2 << 24, sc.sc_ambi };
					fwrite(hdr, 4, 10, f); fwrite(q.data(), 1, q.size(), f); fwrite(t.data(), 1, t.size(), f);
					fclose(f);
				}
			}
			std::vector<uint32_t> cig(q.size() + t.size() + 4);
			wmo_ksw_extd2((int)q.size(), q.data(), (int)t.size(), t.data(), 5, mat, sc.q, sc.e, sc.q2, sc.e2, r->w, r->zdrop, r->end_bonus, r->flag, &ez, cig.data(), 0);
			r->ez.max = ez.max; r->ez.zdropped = ez.zdropped; r->ez.max_q = ez.max_q; r->ez.max_t = ez.max_t; r->ez.mqe = ez.mqe; r->ez.mqe_t = ez.mqe_t;
			r->ez.mte = ez.mte; r->ez.mte_q = ez.mte_q; r->ez.score = ez.score; r->ez.reach_end = ez.reach_end; r->ez.n_cigar = ez.n_cigar; r->ez.cig_off = 0;
			r->cigar.assign(cig.begin(), cig.begin() + ez.n_cigar);
		});
	}
	void exts2_batch(const wm_ksw_score_t &sc, int noncan, int junc_bonus, std::vector<KswReq*> &reqs) override
	{   // splice mode: ksw_exts2_sse per request (src/align.c:326-327)
		int8_t mat[25];
		for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) mat[i * 5 + j] = (i == 4 || j == 4) ? sc.sc_ambi : i == j ? sc.match : sc.mismatch;
		wm::parallel_for(4, reqs.size(), [&](size_t ri) {
			KswReq *r = reqs[ri];
			wmo_ez_t ez;
			std::vector<uint8_t> q(r->ql > 0 ? r->ql : 0), t(r->tl > 0 ? r->tl : 0);
			r->copy_query(q.data()); r->copy_target(t.data());
			check_positions(*r, q, t);
			std::vector<uint32_t> cig(q.size() + t.size() + 4);
			wmo_ksw_exts2((int)q.size(), q.data(), (int)t.size(), t.data(), 5, mat, sc.q, sc.e, sc.q2, noncan, r->zdrop, junc_bonus, r->flag, r->junc.empty() ? 0 : r->junc.data(), &ez, cig.data());
			r->ez.max = ez.max; r->ez.zdropped = ez.zdropped; r->ez.max_q = ez.max_q; r->ez.max_t = ez.max_t; r->ez.mqe = ez.mqe; r->ez.mqe_t = ez.mqe_t;
			r->ez.mte = ez.mte; r->ez.mte_q = ez.mte_q; r->ez.score = ez.score; r->ez.reach_end = ez.reach_end; r->ez.n_cigar = ez.n_cigar; r->ez.cig_off = 0;
			r->cigar.assign(cig.begin(), cig.begin() + ez.n_cigar);
		});
	}
};

struct Harness { Index idx; wmo_bloom_t *bloom; };
static std::atomic<long> g_pos_checked{0}, g_pos_bad{0};       // resident-position checks of all OracleOps so far (h_pos_check)
static void note_pos(const OracleOps &o) { g_pos_checked += o.n_pos_checked.load(); g_pos_bad += o.n_pos_bad.load(); }

extern "C" {

void *h_index_build_flag(const char *fasta, const char *kmer_file, int k, int w, int idx_flag, int n_threads);
void *h_index_build(const char *fasta, const char *kmer_file, int k, int w, int n_threads) { return h_index_build_flag(fasta, kmer_file, k, w, 0, n_threads); }
void *h_index_build_flag(const char *fasta, const char *kmer_file, int k, int w, int idx_flag, int n_threads)      // idx_flag 1 = MM_I_HPC (-H)
{
	Harness *h = new Harness();
	IdxOpt io; io.k = k; io.w = w; io.flag = idx_flag;
	std::string err;
	if (index_build_from_fasta(io, fasta, kmer_file ? kmer_file : "", n_threads, h->idx, err) < 0) { fprintf(stderr, "h_index_build: %s\n", err.c_str()); delete h; return 0; }
	std::vector<uint64_t> kms;
	if (kmer_file && kmer_file[0]) { std::ifstream in(kmer_file); std::string km; uint64_t f; while (in >> km >> f) kms.push_back(wmo_encode_kmer(km.c_str(), (int)km.size())); }
	h->bloom = wmo_bloom_new(kms.size());
	for (uint64_t x : kms) wmo_bloom_insert(h->bloom, x);
	return h;
}
int h_index_read_bed(void *hv, const char *path) { std::string err; return index_read_bed(((Harness*)hv)->idx, path, true, err); }      // --junc-bed
int h_bed_junc(void *hv, int ctg, int st, int en, uint8_t *out) { return ((Harness*)hv)->idx.bed_junc(ctg, st, en, out); }
int h_getseq(void *hv, uint32_t rid, uint32_t st, uint32_t en, uint8_t *out) { return ((Harness*)hv)->idx.getseq(rid, st, en, out); }
uint64_t h_index_n_minimizers(void *hv) { return ((Harness*)hv)->idx.n_minimizers; }
int h_index_get(void *hv, uint64_t minier, uint64_t *out, int cap)
{
	int n; const uint64_t *p = ((Harness*)hv)->idx.get(minier, &n);
	for (int i = 0; i < n && i < cap; ++i) out[i] = p[i];
	return n;
}
int64_t h_sketch(void *hv, const char *seq, int len, int w, int k, uint32_t rid, uint64_t *ox, uint64_t *oy, int64_t cap)
{
	std::vector<m128> v;
	sketch(seq, len, w, k, rid, &((Harness*)hv)->idx.bloom, v, (((Harness*)hv)->idx.flag & 1) != 0);
	for (size_t i = 0; i < v.size() && (int64_t)i < cap; ++i) ox[i] = v[i].x, oy[i] = v[i].y;
	return (int64_t)v.size();
}
void h_radix_sort_128x(uint64_t *x, uint64_t *y, int64_t n)
{
	std::vector<m128> a(n);
	for (int64_t i = 0; i < n; ++i) a[i].x = x[i], a[i].y = y[i];
	radix_sort_128x(a.data(), a.data() + n);
	for (int64_t i = 0; i < n; ++i) x[i] = a[i].x, y[i] = a[i].y;
}
void h_radix_sort_128x_parallel(uint64_t *x, uint64_t *y, int64_t n, int n_threads)
{
	std::vector<m128> a(n);
	for (int64_t i = 0; i < n; ++i) a[i].x = x[i], a[i].y = y[i];
	radix_sort_128x_parallel(a.data(), a.data() + n, n_threads);
	for (int64_t i = 0; i < n; ++i) x[i] = a[i].x, y[i] = a[i].y;
}
// parallel_tasks (wm_core.h): every index exactly once, on the plain path and through a ParHook with the coarse-task chunk hint
int h_parallel_tasks_selftest(int n_threads, int n)
{
	std::vector<std::atomic<int>> hit(n);
	for (auto &h : hit) h.store(0);
	parallel_tasks(n_threads, (size_t)n, [&](size_t i) { hit[i].fetch_add(1); });
	for (auto &h : hit) if (h.load() != 1) return -1;
	struct Hook : ParHook {
		size_t chunk_seen = 0;
		void run(size_t m, const std::function<void(size_t)> &fn) override { chunk_seen = tl_par_chunk(); tl_par_chunk() = 0; for (size_t i = 0; i < m; ++i) fn(i); }
	} hook;
	tl_par_hook() = &hook;
	parallel_tasks(n_threads, (size_t)n, [&](size_t i) { hit[i].fetch_add(1); });
	tl_par_hook() = 0;
	for (auto &h : hit) if (h.load() != 2) return -2;
	if (n >= 2 && hook.chunk_seen != 1) return -3;          // coarse tasks ask the hook for chunks of one
	return tl_par_chunk() == 0 ? 0 : -4;
}
// the vectorised host routines against their portable specifications (wm_align.cpp): ksw_ll_i16 and the scan of mm_update_extra
int h_extra_walk_both(const uint8_t *q, const uint8_t *t, const uint32_t *cigar, int n_cigar, int match, int mismatch, int ambi, int gq, int ge, int32_t *fast6, int32_t *generic6)
{
	wm::extra_walk_both(q, t, cigar, n_cigar, match, mismatch, ambi, gq, ge, fast6, generic6);
	return 0;
}
int h_ll_i16_portable(int qlen, const uint8_t *q, int tlen, const uint8_t *t, const int8_t *mat, int gapo, int gape, int *qe, int *te)
{
#if defined(__SSE2__)
	return ll_i16_portable(qlen, q, tlen, t, mat, gapo, gape, qe, te);
#else
	return ll_i16(qlen, q, tlen, t, mat, gapo, gape, qe, te);
#endif
}
int h_ll_i16(int qlen, const uint8_t *q, int tlen, const uint8_t *t, const int8_t *mat, int gapo, int gape, int *qe, int *te) { return ll_i16(qlen, q, tlen, t, mat, gapo, gape, qe, te); }

int64_t h_chain_extract(int64_t n, const uint64_t *ax, const uint64_t *ay, const int32_t *f, const int32_t *p, const int32_t *v, int min_cnt, int min_sc,
                        int *n_u, uint64_t *u_out, uint64_t *bx, uint64_t *by)
{
	std::vector<m128> a(n);
	for (int64_t i = 0; i < n; ++i) a[i].x = ax[i], a[i].y = ay[i];
	std::vector<uint64_t> u; std::vector<m128> b;
	chain_extract(n, a.data(), f, p, v, min_cnt, min_sc, u, b);
	*n_u = (int)u.size();
	for (size_t i = 0; i < u.size(); ++i) u_out[i] = u[i];
	for (size_t i = 0; i < b.size(); ++i) bx[i] = b[i].x, by[i] = b[i].y;
	return (int64_t)b.size();
}
float h_avg_qspan(int64_t n, const uint64_t *ay) { std::vector<m128> a(n); for (int64_t i = 0; i < n; ++i) a[i].y = ay[i]; return chain_avg_qspan(n, a.data()); }
void h_index_view(void *hv, const uint64_t **hkey, const uint64_t **hval, const uint64_t **P, int *hbits, uint64_t *nslots, uint64_t *np)
{
	Index &ix = ((Harness*)hv)->idx;
	*hkey = ix.hkey.data(); *hval = ix.hval.data(); *P = ix.P.data(); *hbits = ix.hbits; *nslots = ix.hkey.size(); *np = ix.P.size();
}
void h_bloom_view(void *hv, uint32_t *table_bits, uint32_t *salts, const uint8_t **bits)
{
	Index &ix = ((Harness*)hv)->idx;
	*table_bits = (uint32_t)ix.bloom.table_bits; salts[0] = ix.bloom.salt[0]; salts[1] = ix.bloom.salt[1]; *bits = ix.bloom.bits.data();
}

// index files in the reference's format
int h_index_save_mmi(void *hv, const char *path) { std::string err; return index_save_mmi(((Harness*)hv)->idx, path, err); }
void *h_index_load_mmi(const char *path, const char *kmer_file)
{
	Harness *h = new Harness();
	std::string err;
	if (index_load_mmi(path, kmer_file ? kmer_file : "", h->idx, err) < 0) { fprintf(stderr, "%s\n", err.c_str()); delete h; return 0; }
	std::vector<uint64_t> kms;                                        // the oracle-backed sketch op uses the oracle's own filter
	if (kmer_file && kmer_file[0]) { std::ifstream in(kmer_file); std::string km; uint64_t f; while (in >> km >> f) kms.push_back(wmo_encode_kmer(km.c_str(), (int)km.size())); }
	h->bloom = wmo_bloom_new(kms.size());
	for (uint64_t x : kms) wmo_bloom_insert(h->bloom, x);
	return h;
}

// same output layout as refshim_map (oracle/ref_shim.cpp)
static int64_t g_max_sw_mat = 0, g_flag_clear = 0;
static thread_local int g_last_rl_defined = 1;
int h_last_rep_len_defined() { return g_last_rl_defined; }        // of the last h_map call on this thread: did the mapper ASSIGN rep_len where the reference does (src/map.c:808-813, 859-861)
void h_set_flag_clear(int64_t bits) { g_flag_clear = bits; }     // option bits the h_map* calls that follow clear after the preset (e.g. -uf clears MM_F_SPLICE_REV)
int h_map(void *hv, const char *preset, int64_t flag_extra, const char *seq, int len, const char *name,
          int32_t *hit_out, int hit_cap, uint32_t *cig_out, int64_t cig_cap, int64_t *n_cig_total, uint64_t *stats_out)
{
	Harness *h = (Harness*)hv;
	IdxOpt io; MapOpt mo;
	set_preset(0, io, mo);
	if (preset && preset[0] && set_preset(preset, io, mo) < 0) return -1;
	mo.flag |= flag_extra; mo.flag &= ~g_flag_clear;
	mapopt_update(mo, h->idx);
	OracleOps ops; ops.idx = &h->idx; ops.bloom = h->bloom; ops.opt = &mo;
	std::vector<ReadIn> reads(1);
	reads[0].name = name; reads[0].seq.assign(seq, len);
	std::vector<ReadOut> out;
	MapStats st;
	map_batch(h->idx, mo, &ops, reads, out, &st);
	note_pos(ops);
	if (!st.internal_error.empty()) { fprintf(stderr, "[harness] %s\n", st.internal_error.c_str()); return -5; }      // a violated invariant of the host mapper (it used to be dropped here)
	if (ops.n_pos_bad.load()) return -3;
	if (stats_out) { stats_out[0] = st.n_flush; stats_out[1] = st.n_ksw; stats_out[2] = st.n_chain; stats_out[3] = st.n_sketch; }
	int64_t nc = 0;
	const std::vector<Reg> &regs = out[0].regs;
	g_last_rl_defined = out[0].rep_len_defined ? 1 : 0;
	for (size_t i = 0; i < regs.size() && (int)i < hit_cap; ++i) {
		const Reg &r = regs[i];
		int32_t *o = hit_out + 16 * i;
		o[0] = r.rid; o[1] = r.rs; o[2] = r.re; o[3] = r.qs; o[4] = r.qe; o[5] = r.rev; o[6] = r.mapq; o[7] = r.has_p ? (int)r.cigar.size() : 0;
		o[8] = r.score; o[9] = r.cnt; o[10] = r.mlen; o[11] = r.blen; o[12] = r.dp_score; o[13] = r.dp_max; o[14] = r.dp_max2;
		o[15] = (r.parent == r.id) | r.inv << 1 | r.sam_pri << 2 | r.split << 3 | (r.has_p ? r.trans_strand << 5 : 0);
		for (uint32_t c : r.cigar) { if (nc < cig_cap) cig_out[nc] = c; ++nc; }
	}
	*n_cig_total = nc;
	return (int)regs.size();
}

// how many requests carried resident positions (checked against their host views) and how many of them disagreed
void h_pos_check(long *checked, long *bad) { *checked = g_pos_checked.load(); *bad = g_pos_bad.load(); }

void h_set_max_sw_mat(int64_t v) { g_max_sw_mat = v; }          // mm_mapopt_t::max_sw_mat for the h_map_many calls that follow (src/align.c:323-325)

// several reads at once on a team of `n_threads` schedulers; hits of read i start at hit_first[i] (16 ints each)
int h_map_many(void *hv, const char *preset, int64_t flag_extra, int n, const char *const *seqs, const int *lens, int n_threads,
               int32_t *hit_out, int hit_cap, int64_t *hit_first, uint32_t *cig_out, int64_t cig_cap, int64_t *n_cig_total)
{
	Harness *h = (Harness*)hv;
	IdxOpt io; MapOpt mo;
	set_preset(0, io, mo);
	if (preset && preset[0] && set_preset(preset, io, mo) < 0) return -1;
	mo.flag |= flag_extra; mo.flag &= ~g_flag_clear;
	mo.max_sw_mat = g_max_sw_mat;
	mapopt_update(mo, h->idx);
	OracleOps ops; ops.idx = &h->idx; ops.bloom = h->bloom; ops.opt = &mo;
	std::vector<ReadIn> reads(n);
	for (int i = 0; i < n; ++i) { reads[i].name = "read" + std::to_string(i); reads[i].seq.assign(seqs[i], lens[i]); }
	std::vector<ReadOut> out;
	map_batch(h->idx, mo, &ops, reads, out, 0, n_threads);
	note_pos(ops);
	if (ops.n_pos_bad.load()) return -3;
	if (prof_on()) prof_report(stderr);
	int64_t nc = 0; int nh = 0;
	for (int k = 0; k < n; ++k) {
		hit_first[k] = nh;
		for (const Reg &r : out[k].regs) {
			if (nh >= hit_cap) return -2;
			int32_t *o = hit_out + 16 * (size_t)nh++;
			o[0] = r.rid; o[1] = r.rs; o[2] = r.re; o[3] = r.qs; o[4] = r.qe; o[5] = r.rev; o[6] = r.mapq; o[7] = r.has_p ? (int)r.cigar.size() : 0;
			o[8] = r.score; o[9] = r.cnt; o[10] = r.mlen; o[11] = r.blen; o[12] = r.dp_score; o[13] = r.dp_max; o[14] = r.dp_max2;
			o[15] = (r.parent == r.id) | r.inv << 1 | r.sam_pri << 2 | r.split << 3 | (r.has_p ? r.trans_strand << 5 : 0);
			for (uint32_t c : r.cigar) { if (nc < cig_cap) cig_out[nc] = c; ++nc; }
		}
	}
	hit_first[n] = nh;
	*n_cig_total = nc;
	return nh;
}

// formatted records (PAF, or SAM with MM_F_OUT_SAM in flag_extra) of several reads; returns the text length (or -needed)
int64_t h_map_text(void *hv, const char *preset, int64_t flag_extra, int n, const char *const *names, const char *const *seqs, const int *lens, int n_threads,
                   char *out, int64_t cap)
{
	Harness *h = (Harness*)hv;
	IdxOpt io; MapOpt mo;
	set_preset(0, io, mo);
	if (preset && preset[0] && set_preset(preset, io, mo) < 0) return -1;
	mo.flag |= flag_extra; mo.flag &= ~g_flag_clear;
	mapopt_update(mo, h->idx);
	OracleOps ops; ops.idx = &h->idx; ops.bloom = h->bloom; ops.opt = &mo;
	std::vector<ReadIn> reads(n);
	for (int i = 0; i < n; ++i) { reads[i].name = names[i]; reads[i].seq.assign(seqs[i], lens[i]); }
	std::vector<ReadOut> outv;
	map_batch(h->idx, mo, &ops, reads, outv, 0, n_threads);
	note_pos(ops);
	if (ops.n_pos_bad.load()) return -3;
	std::string text;
	for (int i = 0; i < n; ++i) write_read(text, h->idx, reads[i], outv[i], mo.flag);
	if ((int64_t)text.size() > cap) return -(int64_t)text.size();
	memcpy(out, text.data(), text.size());
	return (int64_t)text.size();
}

// the file-level pipeline (host/wm_pipeline.cpp) over oracle-backed ops: FASTA/FASTQ(.gz) -> PAF/SAM file
int h_map_file(void *hv, const char *preset, int64_t flag_extra, const char *reads_path, const char *out_path, int64_t mini_batch_bases, int n_threads, double *stats)
{
	Harness *h = (Harness*)hv;
	IdxOpt io; MapOpt mo;
	set_preset(0, io, mo);
	if (preset && preset[0] && set_preset(preset, io, mo) < 0) return -1;
	mo.flag |= flag_extra; mo.flag &= ~g_flag_clear;
	mapopt_update(mo, h->idx);
	OracleOps ops; ops.idx = &h->idx; ops.bloom = h->bloom; ops.opt = &mo;
	FILE *out = fopen(out_path, "wb");
	if (!out) return -2;
	std::string err;
	FileStats fs;
	const int rc = map_file(reads_path, mini_batch_bases, (mo.flag & 0x8) != 0, [&](std::vector<ReadIn> &batch, std::string &text, int lane) {
		std::vector<ReadOut> outv;
		map_batch(h->idx, mo, &ops, batch, outv, 0, n_threads, 0, lane);
		for (size_t i = 0; i < batch.size(); ++i) write_read(text, h->idx, batch[i], outv[i], mo.flag);
		return 0;
	}, out, &fs, err);
	fclose(out);
	if (stats) { stats[0] = (double)fs.n_reads; stats[1] = (double)fs.n_bases; stats[2] = (double)fs.n_batches; }
	return rc;
}

// a reference indexed in parts + the merge (host/wm_pipeline.cpp: map_file_split) over oracle-backed ops; returns the number of parts or < 0
int h_map_file_split(const char *fasta, const char *kmer_file, int k, int w, int64_t batch_bases, const char *preset, int64_t flag_extra,
                     const char *reads_path, const char *out_path, int64_t mini_batch_bases, int n_threads)
{
	IdxOpt io; MapOpt mo;
	set_preset(0, io, mo);
	if (preset && preset[0] && set_preset(preset, io, mo) < 0) return -1;
	io.k = k; io.w = w;
	mo.flag |= flag_extra; mo.flag &= ~g_flag_clear;
	std::string err;
	std::vector<uint64_t> kms;
	if (kmer_file && kmer_file[0]) { std::ifstream in(kmer_file); std::string km; uint64_t f; while (in >> km >> f) kms.push_back(wmo_encode_kmer(km.c_str(), (int)km.size())); }
	wmo_bloom_t *bloom = wmo_bloom_new(kms.size());
	for (uint64_t x : kms) wmo_bloom_insert(bloom, x);
	// one part at a time (src/main.c:398-429): the next part is read and built when its turn comes, mapped, spilled and destroyed
	IndexPartReader rd;
	if (rd.open(fasta, err) < 0) { fprintf(stderr, "h_map_file_split: %s\n", err.c_str()); return -2; }
	SplitRun run(reads_path, mini_batch_bases, mo, k, w);
	std::vector<std::string> names, seqs;
	std::vector<size_t> part_seqs;
	int n_parts = 0;
	while (rd.next((uint64_t)batch_bases, names, seqs) > 0) {
		Index part;
		if (index_build(io, names, seqs, kmer_file ? kmer_file : "", n_threads, part, err) < 0) { fprintf(stderr, "h_map_file_split: %s\n", err.c_str()); return -2; }
		std::vector<std::string>().swap(seqs);
		MapOpt cur = mo;
		mapopt_update(cur, part);
		OracleOps ops; ops.idx = &part; ops.bloom = bloom; ops.opt = &cur;
		const int rc = run.add_part(part.seq, [&](std::vector<ReadIn> &batch, std::vector<ReadOut> &o, int lane) -> int { map_batch(part, cur, &ops, batch, o, 0, n_threads, 0, lane); return 0; }, err);
		if (rc) { fprintf(stderr, "h_map_file_split: %s\n", err.c_str()); return -4; }
		part_seqs.push_back(part.seq.size());
		++n_parts;
	}
	{   // the parts formed one at a time are the parts the all-at-once builder forms
		std::vector<Index> all;
		const int n2 = index_build_parts_from_fasta(io, fasta, kmer_file ? kmer_file : "", n_threads, (uint64_t)batch_bases, all, err);
		if (n2 != n_parts) { fprintf(stderr, "h_map_file_split: %d parts one at a time, %d at once\n", n_parts, n2); return -5; }
		for (int j = 0; j < n2; ++j) if (all[j].seq.size() != part_seqs[j]) { fprintf(stderr, "h_map_file_split: part %d differs\n", j); return -5; }
	}
	FILE *out = fopen(out_path, "wb");
	if (!out) return -3;
	if (mo.flag & 0x8) { std::string hdr; write_sam_header(hdr, run.dict(), 0, 0); fwrite(hdr.data(), 1, hdr.size(), out); }
	FileStats fs;
	const int rc = run.finish(out, &fs, err);
	fclose(out);
	if (rc) { fprintf(stderr, "h_map_file_split: %s\n", err.c_str()); return -4; }
	return n_parts;
}

// the mapper's own mm_test_zdrop (scan + verdict incl. the inversion test) and mm_update_extra (mm_fix_cigar + the vector scan), for
// tests/test_walks_vs_ref.py: compared with the reference's static functions (oracle/ref_align_shim.cpp)
int h_test_zdrop(int64_t flag, int zdrop, int zdrop_inv, int q, int e, int max_gap, int min_chain_score, int a, int min_dp_max,
                 const uint8_t *qseq, const uint8_t *tseq, const uint32_t *cigar, int n_cigar, const int8_t *mat)
{
	MapOpt o;
	o.flag = flag; o.zdrop = zdrop; o.zdrop_inv = zdrop_inv; o.q = q; o.e = e; o.max_gap = max_gap; o.min_chain_score = min_chain_score; o.a = a; o.min_dp_max = min_dp_max;
	std::vector<uint32_t> cg(cigar, cigar + n_cigar);
	return test_zdrop(o, qseq, tseq, cg, mat);
}
int h_update_extra(int rev, int qs, int qe, int rs, int re, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int q, int e,
                   const uint32_t *cigar, int n_cigar, int32_t *out6, uint32_t *cigar_out, int cap, int *n_out)
{
	Reg r;
	r.rev = rev; r.qs = qs; r.qe = qe; r.rs = rs; r.re = re; r.has_p = true;
	r.cigar.assign(cigar, cigar + n_cigar);
	update_extra(r, qseq, tseq, mat, q, e);
	out6[0] = r.blen; out6[1] = r.mlen; out6[2] = (int32_t)r.n_ambi; out6[3] = r.dp_max; out6[4] = rev ? r.qe : r.qs; out6[5] = r.rs;
	*n_out = (int)r.cigar.size();
	for (size_t i = 0; i < r.cigar.size() && (int)i < cap; ++i) cigar_out[i] = r.cigar[i];
	std::string ie;
	return take_internal_error(ie) ? -5 : 0;
}

int h_usable_cores(void) { return usable_cores(); }
// the product's host packer of the resident reads (csrc/reads2bit.h)
void h_pack_codes(const uint8_t *codes, size_t n, uint64_t *pk, uint64_t *nm) { wm_pack_codes(codes, n, pk, nm); }
size_t h_pk_words(size_t n) { return wm_pk_words(n); }
size_t h_nm_words(size_t n) { return wm_nm_words(n); }

// the host's compile of the CIGAR walks the device also runs (winnowmap_amd/csrc/cigar_walk.h)
void h_zdrop_walk(const uint8_t *q, const uint8_t *t, const uint32_t *cigar, int n_cigar, int match, int mismatch, int ambi, int gq, int ge, int32_t *out5)
{
	wm_zd_t z;
	wm_zdrop_walk(q, t, cigar, n_cigar, match, mismatch, ambi, gq, ge, &z);
	out5[0] = z.max_zdrop; out5[1] = z.t0; out5[2] = z.t1; out5[3] = z.q0; out5[4] = z.q1;
}
} // extern "C"
