// wm_hit.cpp — see wm_hit.h. Host glue between the chaining kernel and the alignment kernels.
#include "wm_hit.h"
#include <math.h>
#include <algorithm>

namespace wm {

static inline int32_t a_span(const m128 &a) { return (int32_t)(a.y >> 32 & 0xff); }
static inline int32_t a_rpos(const m128 &a) { return (int32_t)a.x; }
static inline int32_t a_qpos(const m128 &a) { return (int32_t)a.y; }

void reg_set_coor(Reg &r, int32_t qlen, const m128 *a)
{
	const m128 &first = a[r.as], &last = a[r.as + r.cnt - 1];
	const int32_t span0 = a_span(first);
	r.rev = (uint32_t)(first.x >> 63);
	r.rid = (int32_t)(first.x << 1 >> 33);
	r.rs = a_rpos(first) + 1 > span0 ? a_rpos(first) + 1 - span0 : 0;
	r.re = a_rpos(last) + 1;
	if (!r.rev) {
		r.qs = a_qpos(first) + 1 - span0;
		r.qe = a_qpos(last) + 1;
	} else {
		r.qs = qlen - (a_qpos(last) + 1);
		r.qe = qlen - (a_qpos(first) + 1 - span0);
	}
	// approximate matched / block lengths along the chain (mm_cal_fuzzy_len, src/hit.c:8-21)
	r.mlen = r.blen = 0;
	if (r.cnt <= 0) return;
	r.mlen = r.blen = span0;
	for (int i = r.as + 1; i < r.as + r.cnt; ++i) {
		const int sp = a_span(a[i]);
		const int dt = a_rpos(a[i]) - a_rpos(a[i - 1]), dq = a_qpos(a[i]) - a_qpos(a[i - 1]);
		r.blen += dt > dq ? dt : dq;
		r.mlen += (dt > sp && dq > sp) ? sp : (dt < dq ? dt : dq);
	}
}

std::vector<Reg> gen_regs(uint32_t hash, int qlen, int n_u, const uint64_t *u, const m128 *a)
{
	WM_PROF("hit.gen_regs");
	std::vector<Reg> regs;
	if (n_u == 0) return regs;
	// order chains by (score, per-read pseudo-random tie-break); the sort is the unstable reference one
	std::vector<m128> z(n_u);
	for (int i = 0, k = 0; i < n_u; ++i) {
		const uint32_t h = (uint32_t)hash64_full((hash64_full(a[k].x) + hash64_full(a[k].y)) ^ hash);
		z[i].x = u[i] ^ h;
		z[i].y = (uint64_t)k << 32 | (uint32_t)(int32_t)u[i];
		k += (int32_t)u[i];
	}
	{ WM_PROF("hit.gen_regs.sort"); radix_sort_128x(z.data(), z.data() + n_u); }
	std::reverse(z.begin(), z.end());
	{ WM_PROF("hit.gen_regs.resize"); regs.resize(n_u); }
	for (int i = 0; i < n_u; ++i) {
		Reg &r = regs[i];
		r.id = i;
		r.parent = PARENT_UNSET;
		r.score = r.score0 = (int32_t)(z[i].x >> 32);
		r.hash = (uint32_t)z[i].x;
		r.cnt = (int32_t)z[i].y;
		r.as = (int32_t)(z[i].y >> 32);
		r.div = -1.0f;
		reg_set_coor(r, qlen, a);
	}
	return regs;
}

void split_reg(Reg &r, Reg &r2, int n, int qlen, const m128 *a)
{
	if (n <= 0 || n >= r.cnt) return;
	r2 = r;
	r2.id = -1;
	r2.sam_pri = 0;
	r2.drop_p();
	r2.split_inv = 0;
	r2.cnt = r.cnt - n;
	r2.score = (int32_t)(r.score * ((float)r2.cnt / r.cnt) + .499);
	r2.as = r.as + n;
	if (r.parent == r.id) r2.parent = PARENT_TMP_PRI;
	reg_set_coor(r2, qlen, a);
	r.cnt -= r2.cnt;
	r.score -= r2.score;
	reg_set_coor(r, qlen, a);
	r.split |= 1; r2.split |= 2;
}

// The chain-level bookkeeping below (set_parent, select_sub, sync_regs) runs on every chain the chaining kernel returns — thousands per read in
// repeats, nearly all of them dropped by select_sub — so it is written once over the element type: Reg, or the slim ChainRec that
// gen_regs_select uses before any Reg is built.
namespace {
struct ChainRec {                       // the fields of a Reg that mm_set_parent / mm_select_sub / mm_sync_regs read or write, for a chain without an alignment
	int32_t id, cnt, rid, score, qs, qe, rs, re, parent, subsc, as, n_sub;
	uint32_t hash, rev, inv, sam_pri;
	static constexpr bool has_p = false;
	static constexpr int32_t dp_max = 0;
	int32_t dp_max2;
};

template <class R> void set_parent_impl(float mask_level, int mask_len, std::vector<R> &r, int sub_diff, int hard_mask_level)
{
	const int n = (int)r.size();
	if (n <= 0) return;
	for (int i = 0; i < n; ++i) r[i].id = i;
	std::vector<uint64_t> cov(n);
	std::vector<int> pri;                                   // indices of the primary hits seen so far
	pri.push_back(0);
	r[0].parent = 0;
	for (int i = 1; i < n; ++i) {
		R &ri = r[i];
		const int si = ri.qs, ei = ri.qe;
		int uncov = 0;
		bool secondary = false;
		bool overlaps = hard_mask_level != 0;
		if (!hard_mask_level) {                            // query bases of i not covered by any primary hit
			int nc = 0;
			for (int p : pri) {
				int sj = r[p].qs, ej = r[p].qe;
				if (ej <= si || sj >= ei) continue;
				if (sj < si) sj = si;
				if (ej > ei) ej = ei;
				cov[nc++] = (uint64_t)sj << 32 | (uint32_t)ej;
			}
			if (nc > 0) {
				overlaps = true;
				radix_sort_64(cov.data(), cov.data() + nc);
				int x = si;
				for (int j = 0; j < nc; ++j) {
					if ((int)(cov[j] >> 32) > x) uncov += (int)(cov[j] >> 32) - x;
					x = (int32_t)cov[j] > x ? (int32_t)cov[j] : x;
				}
				if (ei > x) uncov += ei - x;
			}
		}
		if (overlaps) {
			for (int p : pri) {
				R &rp = r[p];
				const int sj = rp.qs, ej = rp.qe;
				if (ej <= si || sj >= ei) continue;
				const int mn = ej - sj < ei - si ? ej - sj : ei - si;
				const int mx = ej - sj > ei - si ? ej - sj : ei - si;
				const int ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
				if ((float)ol / mn - (float)uncov / mx > mask_level && uncov <= mask_len) {
					int cnt_sub = 0, sci = ri.score;
					ri.parent = rp.parent;
					rp.subsc = rp.subsc > sci ? rp.subsc : sci;
					if (ri.cnt >= rp.cnt) cnt_sub = 1;
					if (rp.has_p && ri.has_p && (rp.rid != ri.rid || rp.rs != ri.rs || rp.re != ri.re || ol != mn)) {
						sci = ri.dp_max;
						rp.dp_max2 = rp.dp_max2 > sci ? rp.dp_max2 : sci;
						if (rp.dp_max - ri.dp_max <= sub_diff) cnt_sub = 1;
					}
					if (cnt_sub) ++rp.n_sub;
					secondary = true;
					break;
				}
			}
		}
		if (!secondary) { pri.push_back(i); ri.parent = i; ri.n_sub = 0; }
	}
}

template <class R> int set_sam_pri_impl(std::vector<R> &r)
{
	int n_pri = 0;
	for (R &x : r) {
		if (x.id == x.parent) { ++n_pri; x.sam_pri = (n_pri == 1); }
		else x.sam_pri = 0;
	}
	return n_pri;
}

template <class R> void sync_regs_impl(std::vector<R> &r)
{
	const int n = (int)r.size();
	if (n <= 0) return;
	int max_id = -1;
	for (const R &x : r) max_id = max_id > x.id ? max_id : x.id;
	std::vector<int> where(max_id + 1 > 0 ? max_id + 1 : 0, -1);
	for (int i = 0; i < n; ++i) if (r[i].id >= 0) where[r[i].id] = i;
	for (int i = 0; i < n; ++i) {
		R &x = r[i];
		x.id = i;
		if (x.parent == PARENT_TMP_PRI) x.parent = i;
		else if (x.parent >= 0 && where[x.parent] >= 0) x.parent = where[x.parent];
		else x.parent = PARENT_UNSET;
	}
	set_sam_pri_impl(r);
}

template <class R> void select_sub_impl(float pri_ratio, int min_diff, int best_n, std::vector<R> &r)
{
	if (!(pri_ratio > 0.0f) || r.empty()) return;
	const int n = (int)r.size();
	int k = 0, n_2nd = 0;
	for (int i = 0; i < n; ++i) {
		const int p = r[i].parent;
		bool keep = false;
		if (p == i || r[i].inv) keep = true;
		else if ((r[i].score >= r[p].score * pri_ratio || r[i].score + min_diff >= r[p].score) && n_2nd < best_n) {
			// NB: r[p] may already have been overwritten by compaction in the reference too (p < i, r[k++]=r[i])
			if (!(r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].rid == r[p].rid && r[i].rs == r[p].rs && r[i].re == r[p].re)) { keep = true; ++n_2nd; }
		}
		if (keep) { if (k != i) r[k] = r[i]; ++k; }
	}
	if (k != n) { r.resize(k); sync_regs_impl(r); }
}
} // namespace

void set_parent(float mask_level, int mask_len, std::vector<Reg> &r, int sub_diff, int hard_mask_level)
{
	WM_PROF("hit.set_parent");
	set_parent_impl(mask_level, mask_len, r, sub_diff, hard_mask_level);
}

void hit_sort(std::vector<Reg> &r)
{
	WM_PROF("hit.hit_sort");
	const int n = (int)r.size();
	if (n <= 1) return;
	std::vector<m128> aux;
	aux.reserve(n);
	for (int i = 0; i < n; ++i) {
		if (r[i].inv || r[i].cnt > 0) {
			const int score = r[i].has_p ? r[i].dp_max : r[i].score;
			m128 t = { (uint64_t)score << 32 | r[i].hash, (uint64_t)i };
			aux.push_back(t);
		} else r[i].drop_p();
	}
	radix_sort_128x(aux.data(), aux.data() + aux.size());
	std::vector<Reg> t;
	t.reserve(aux.size());
	for (int i = (int)aux.size() - 1; i >= 0; --i) t.push_back(std::move(r[aux[i].y]));
	r.swap(t);
}

int set_sam_pri(std::vector<Reg> &r) { return set_sam_pri_impl(r); }

void sync_regs(std::vector<Reg> &r) { sync_regs_impl(r); }

void select_sub(float pri_ratio, int min_diff, int best_n, std::vector<Reg> &r)
{
	WM_PROF("hit.select_sub");
	select_sub_impl(pri_ratio, min_diff, best_n, r);
}

// gen_regs + set_parent + select_sub (src/map.c:256-262 after mm_gen_regs, :375) with the same result, for the usual case that most chains are dropped:
// the three steps run on ChainRec (64 bytes, coordinates only) and a Reg — with its fuzzy lengths, a walk over the chain's anchors — is built for
// the survivors alone. A stage-1 window in a repeat returns thousands of chains of which select_sub keeps a handful: building and ordering full
// Regs for all of them was a third of the host glue's CPU time (tests/host_harness/prof.sh).
std::vector<Reg> gen_regs_select(uint32_t hash, int qlen, int n_u, const uint64_t *u, const m128 *a, float mask_level, int mask_len, int sub_diff, int hard_mask_level,
                                 float pri_ratio, int min_diff, int best_n)
{
	WM_PROF("hit.gen_regs_select");
	std::vector<Reg> regs;
	if (n_u == 0) return regs;
	std::vector<m128> z(n_u);
	for (int i = 0, k = 0; i < n_u; ++i) {
		const uint32_t h = (uint32_t)hash64_full((hash64_full(a[k].x) + hash64_full(a[k].y)) ^ hash);
		z[i].x = u[i] ^ h;
		z[i].y = (uint64_t)k << 32 | (uint32_t)(int32_t)u[i];
		k += (int32_t)u[i];
	}
	radix_sort_128x(z.data(), z.data() + n_u);
	std::vector<ChainRec> c;
	c.reserve(n_u);
	for (int i = 0; i < n_u; ++i) {                         // descending score (the reference reverses the sorted array, src/hit.c:75-77)
		const m128 &zi = z[n_u - 1 - i];
		ChainRec x;
		x.id = i; x.parent = PARENT_UNSET; x.score = (int32_t)(zi.x >> 32); x.hash = (uint32_t)zi.x; x.cnt = (int32_t)zi.y; x.as = (int32_t)(zi.y >> 32);
		x.subsc = 0; x.n_sub = 0; x.inv = 0; x.sam_pri = 0; x.dp_max2 = 0;
		const m128 &first = a[x.as], &last = a[x.as + x.cnt - 1];            // mm_reg_set_coor without the lengths (reg_set_coor above)
		const int32_t span0 = a_span(first);
		x.rev = (uint32_t)(first.x >> 63);
		x.rid = (int32_t)(first.x << 1 >> 33);
		x.rs = a_rpos(first) + 1 > span0 ? a_rpos(first) + 1 - span0 : 0;
		x.re = a_rpos(last) + 1;
		if (!x.rev) { x.qs = a_qpos(first) + 1 - span0; x.qe = a_qpos(last) + 1; }
		else { x.qs = qlen - (a_qpos(last) + 1); x.qe = qlen - (a_qpos(first) + 1 - span0); }
		c.push_back(x);
	}
	set_parent_impl(mask_level, mask_len, c, sub_diff, hard_mask_level);
	select_sub_impl(pri_ratio, min_diff, best_n, c);
	regs.resize(c.size());
	for (size_t i = 0; i < c.size(); ++i) {
		const ChainRec &x = c[i];
		Reg &r = regs[i];
		r.id = x.id; r.parent = x.parent; r.score = r.score0 = x.score; r.hash = x.hash; r.cnt = x.cnt; r.as = x.as;
		r.subsc = x.subsc; r.n_sub = x.n_sub; r.sam_pri = x.sam_pri; r.div = -1.0f;
		reg_set_coor(r, qlen, a);
	}
	return regs;
}

void filter_regs(const MapOpt &opt, int qlen, std::vector<Reg> &r)
{
	size_t k = 0;
	for (size_t i = 0; i < r.size(); ++i) {
		Reg &x = r[i];
		bool flt = false;
		if (!x.inv && x.cnt < opt.min_cnt) flt = true;
		if (x.has_p) {
			if (x.mlen < opt.min_chain_score) flt = true;
			else if (x.dp_max < opt.min_dp_max) flt = true;
			else if (x.qs > qlen * opt.max_clip_ratio && qlen - x.qe > qlen * opt.max_clip_ratio) flt = true;
		}
		if (!flt) { if (k != i) r[k] = std::move(r[i]); ++k; }
	}
	r.resize(k);
}

int squeeze_a(std::vector<Reg> &r, m128 *a)
{
	const int n = (int)r.size();
	std::vector<uint64_t> aux(n);
	for (int i = 0; i < n; ++i) aux[i] = (uint64_t)r[i].as << 32 | (uint32_t)i;
	radix_sort_64(aux.data(), aux.data() + n);
	int as = 0;
	for (int i = 0; i < n; ++i) {
		Reg &x = r[(int32_t)aux[i]];
		if (x.as != as) {
			memmove(&a[as], &a[x.as], (size_t)x.cnt * sizeof(m128));
			x.as = as;
		}
		as += x.cnt;
	}
	return as;
}

void join_long(const MapOpt &opt, int qlen, std::vector<Reg> &r, m128 *a)
{
	WM_PROF("hit.join_long");
	const int n = (int)r.size();
	if (n < 2) return;
	squeeze_a(r, a);
	std::vector<uint64_t> aux;
	for (int i = 0; i < n; ++i)
		if (r[i].parent == i || r[i].parent < 0) aux.push_back((uint64_t)r[i].as << 32 | (uint32_t)i);
	radix_sort_64(aux.data(), aux.data() + aux.size());
	int n_drop = 0;
	for (int i = (int)aux.size() - 1; i >= 1; --i) {
		Reg &r0 = r[(int32_t)aux[i - 1]], &r1 = r[(int32_t)aux[i]];
		if (r0.as + r0.cnt != r1.as) continue;
		if (r0.rid != r1.rid || r0.rev != r1.rev) continue;
		const m128 &a0e = a[r0.as + r0.cnt - 1], &a1s = a[r1.as];
		if (a1s.x <= a0e.x || (int32_t)a1s.y <= (int32_t)a0e.y) continue;
		int max_gap, min_gap;
		max_gap = min_gap = (int32_t)a1s.y - (int32_t)a0e.y;
		max_gap = a0e.x + max_gap > a1s.x ? max_gap : (int)(a1s.x - a0e.x);
		min_gap = a0e.x + min_gap < a1s.x ? min_gap : (int)(a1s.x - a0e.x);
		if (max_gap > opt.max_join_long || min_gap > opt.max_join_short) continue;
		const int sc_thres = (int)((float)opt.min_join_flank_sc / opt.max_join_long * max_gap + .499);
		if (r0.score < sc_thres || r1.score < sc_thres) continue;
		const int min_flank = (int)(max_gap * opt.min_join_flank_ratio);
		if (r0.re - r0.rs < min_flank || r0.qe - r0.qs < min_flank) continue;
		if (r1.re - r1.rs < min_flank || r1.qe - r1.qs < min_flank) continue;
		a[r1.as].y |= SEED_LONG_JOIN;
		r0.cnt += r1.cnt; r0.score += r1.score;
		reg_set_coor(r0, qlen, a);
		r1.cnt = 0;
		r1.parent = r0.id;
		++n_drop;
	}
	if (n_drop > 0) {
		for (int i = 0; i < n; ++i) {
			Reg &x = r[i];
			if (x.parent >= 0 && x.id != x.parent)
				if (r[x.parent].parent >= 0 && r[x.parent].parent != x.parent) x.parent = r[x.parent].parent;
		}
		filter_regs(opt, qlen, r);
		sync_regs(r);
	}
}

static void set_inv_mapq(std::vector<Reg> &r)
{   // mm_set_inv_mapq, src/hit.c:435-461
	const int n = (int)r.size();
	if (n < 3) return;
	bool any = false;
	for (const Reg &x : r) any |= x.inv != 0;
	if (!any) return;
	std::vector<m128> aux;
	for (int i = 0; i < n; ++i)
		if (r[i].parent == i || r[i].parent < 0) { m128 t = { (uint64_t)r[i].rid << 32 | (uint32_t)r[i].rs, (uint64_t)i }; aux.push_back(t); }
	radix_sort_128x(aux.data(), aux.data() + aux.size());
	for (int i = 1; i < (int)aux.size() - 1; ++i) {
		Reg &v = r[aux[i].y];
		if (v.inv) {
			const Reg &l = r[aux[i - 1].y], &g = r[aux[i + 1].y];
			v.mapq = l.mapq < g.mapq ? l.mapq : g.mapq;
		}
	}
}

void set_mapq(std::vector<Reg> &r, int min_chain_sc, int match_sc, int rep_len, int is_sr)
{
	WM_PROF("hit.set_mapq");
	static const float q_coef = 40.0f;
	if (r.empty()) return;
	int64_t sum_sc = 0;
	for (const Reg &x : r) if (x.parent == x.id) sum_sc += x.score;
	const float uniq_ratio = (float)sum_sc / (sum_sc + rep_len);
	for (Reg &x : r) {
		if (x.inv) { x.mapq = 0; continue; }
		if (x.parent != x.id) { x.mapq = 0; continue; }
		int mapq;
		const float pen_s1 = (x.score > 100 ? 1.0f : 0.01f * x.score) * uniq_ratio;
		float pen_cm = x.cnt > 10 ? 1.0f : 0.1f * x.cnt;
		pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
		const int subsc = x.subsc > min_chain_sc ? x.subsc : min_chain_sc;
		if (x.has_p && x.dp_max2 > 0 && x.dp_max > 0) {
			const float identity = (float)x.mlen / x.blen;
			const float t = (float)x.dp_max2 * subsc / x.dp_max / x.score0;
			mapq = (int)(identity * pen_cm * q_coef * (1.0f - t * t) * logf((float)x.dp_max / match_sc));
			if (!is_sr) {
				const int mapq_alt = (int)(6.02f * identity * identity * (x.dp_max - x.dp_max2) / match_sc + .499f);
				mapq = mapq < mapq_alt ? mapq : mapq_alt;
			}
		} else {
			const float t = (float)subsc / x.score0;
			if (x.has_p) {
				const float identity = (float)x.mlen / x.blen;
				mapq = (int)(identity * pen_cm * q_coef * (1.0f - t) * logf((float)x.dp_max / match_sc));
			} else mapq = (int)(pen_cm * q_coef * (1.0f - t) * logf(x.score));
		}
		mapq -= (int)(4.343f * logf(x.n_sub + 1) + .499f);
		mapq = mapq > 0 ? mapq : 0;
		x.mapq = mapq < 60 ? mapq : 60;
		if (x.has_p && x.dp_max > x.dp_max2 && x.mapq == 0) x.mapq = 1;
	}
	set_inv_mapq(r);
}

} // namespace wm
