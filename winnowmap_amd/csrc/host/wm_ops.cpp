// wm_ops.cpp — DeviceOps::window_batch composed from the per-operation batches (see wm_ops.h).
#include "wm_ops.h"

namespace wm {

void DeviceOps::window_batch(int w, int k, std::vector<WindowReq*> &reqs)
{
	const size_t n = reqs.size();
	std::vector<SketchReq> sk(n);
	std::vector<SketchReq*> skp;
	for (size_t i = 0; i < n; ++i)
		if (reqs[i]->len > 0) { sk[i].seq = reqs[i]->seq; sk[i].len = reqs[i]->len; sk[i].dev_off = reqs[i]->dev_off; skp.push_back(&sk[i]); }
	if (!skp.empty()) sketch_batch(w, k, skp);
	// collect_seed_hits takes one (max_occ, flag) per batch: group the requests (one group in practice)
	std::vector<SeedReq> sd(n);
	std::vector<char> done(n, 0);
	for (size_t i = 0; i < n; ++i) {
		if (done[i] || reqs[i]->len <= 0 || sk[i].mini.empty()) continue;
		std::vector<SeedReq*> grp;
		for (size_t j = i; j < n; ++j)
			if (!done[j] && reqs[j]->len > 0 && !sk[j].mini.empty() && reqs[j]->max_occ == reqs[i]->max_occ && reqs[j]->flag == reqs[i]->flag) {
				sd[j].mini = sk[j].mini.data(); sd[j].n_mini = (int)sk[j].mini.size(); sd[j].qlen = reqs[j]->len; sd[j].max_occ = reqs[j]->max_occ; sd[j].flag = reqs[j]->flag;
				grp.push_back(&sd[j]); done[j] = 1;
			}
		seed_batch(grp);
	}
	std::vector<ChainReq> ch(n);
	std::vector<ChainReq*> chp;
	for (size_t i = 0; i < n; ++i) {
		WindowReq &r = *reqs[i];
		ChainReq &c = ch[i];
		c.a = std::move(r.pre);
		r.rep_len = sd[i].rep_len;
		if (r.len > 0) {
			const bool both = !c.a.empty();
			c.a.insert(c.a.end(), sd[i].a.begin(), sd[i].a.end());
			if (both) radix_sort_128x(c.a.data(), c.a.data() + c.a.size());       // src/map.c:833
		}
		r.n_anchors = (int)c.a.size();
		c.max_dist_x = r.max_dist_x; c.min_dist_x = r.min_dist_x; c.max_dist_y = r.max_dist_y; c.bw = r.bw; c.max_skip = r.max_skip; c.max_iter = r.max_iter;
		c.min_cnt = r.min_cnt; c.min_sc = r.min_sc; c.gap_scale = r.gap_scale; c.is_cdna = r.is_cdna;
		if (!c.a.empty()) chp.push_back(&c);
	}
	if (!chp.empty()) chain_batch(chp);
	for (size_t i = 0; i < n; ++i) {
		WindowReq &r = *reqs[i];
		if (r.n_anchors > 0) { r.a = std::move(ch[i].a); r.u = std::move(ch[i].u); } else { r.a.clear(); r.u.clear(); }
	}
}

} // namespace wm
