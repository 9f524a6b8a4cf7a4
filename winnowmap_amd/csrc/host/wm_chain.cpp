#include "wm_chain.h"
#include <algorithm>
namespace wm {

float chain_avg_qspan(int64_t n, const m128 *a)
{
	uint64_t sum = 0;
	for (int64_t i = 0; i < n; ++i) sum += a[i].y >> 32 & 0xff;
	return (float)sum / n;
}

void chain_extract(int64_t n, const m128 *a, const int32_t *f, const int32_t *p, const int32_t *v, int min_cnt, int min_sc,
                   std::vector<uint64_t> &u, std::vector<m128> &b)
{
	u.clear(); b.clear();
	if (n == 0) return;
	std::vector<int32_t> mark(n, 0), order;
	// a chain ends at an anchor nobody continues from; its score is the peak f on the way back (src/chain.c:93-110)
	for (int64_t i = 0; i < n; ++i) if (p[i] >= 0) mark[p[i]] = 1;
	for (int64_t i = 0; i < n; ++i) {
		if (mark[i] != 0 || v[i] < min_sc) continue;
		int64_t j = i;
		while (j >= 0 && f[j] < v[j]) j = p[j];
		if (j < 0) j = i;
		u.push_back((uint64_t)f[j] << 32 | (uint64_t)j);
	}
	if (u.empty()) return;
	radix_sort_64(u.data(), u.data() + u.size());
	std::reverse(u.begin(), u.end());                  // best chain first
	// backtrack; an anchor belongs to the best chain that reaches it (:119-135)
	std::fill(mark.begin(), mark.end(), 0);
	size_t k = 0;
	for (size_t i = 0; i < u.size(); ++i) {
		const size_t n0 = order.size(), k0 = k;
		int64_t j = (int32_t)u[i];
		do { order.push_back((int32_t)j); mark[j] = 1; j = p[j]; } while (j >= 0 && mark[j] == 0);
		const int64_t cnt = (int64_t)(order.size() - n0);
		if (j < 0) { if (cnt >= min_cnt) u[k++] = u[i] >> 32 << 32 | (uint64_t)cnt; }
		else if ((int32_t)(u[i] >> 32) - f[j] >= min_sc) { if (cnt >= min_cnt) u[k++] = ((u[i] >> 32) - (uint64_t)f[j]) << 32 | (uint64_t)cnt; }
		if (k0 == k) order.resize(n0);
	}
	u.resize(k);
	// anchors of each chain in ascending order, chains ordered by their first anchor's x (:141-165)
	std::vector<m128> tmp(order.size()), w(k);
	size_t pos = 0;
	for (size_t i = 0; i < k; ++i) {
		const int32_t ni = (int32_t)u[i];
		for (int32_t j = 0; j < ni; ++j) tmp[pos + j] = a[order[pos + (ni - j - 1)]];
		w[i].x = tmp[pos].x; w[i].y = (uint64_t)pos << 32 | (uint64_t)i;
		pos += ni;
	}
	radix_sort_128x(w.data(), w.data() + k);
	std::vector<uint64_t> u2(k);
	b.resize(order.size());
	pos = 0;
	for (size_t i = 0; i < k; ++i) {
		const int32_t src = (int32_t)w[i].y, cnt = (int32_t)u[src];
		u2[i] = u[src];
		std::copy(tmp.begin() + (w[i].y >> 32), tmp.begin() + (w[i].y >> 32) + cnt, b.begin() + pos);
		pos += cnt;
	}
	u.swap(u2);
}

} // namespace wm
