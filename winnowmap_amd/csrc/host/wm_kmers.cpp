// wm_kmers.cpp — the `-W` list: canonical k-mers whose count exceeds the smallest threshold c such that at least
// `distinct` of the distinct k-mers have count <= c (meryl `print greater-than distinct=0.9998`, README.md:29-30,
// ext/meryl/src/meryl/merylOp-nextMer.C:103-115), written as "kmer<TAB>count" lines (the format src/index.c:397 reads).
// meryl itself cannot be built offline; this is SURVEY §8f-3, host version.
#include "wm_core.h"
#include <stdio.h>
#include <algorithm>
namespace wm {

int write_repetitive_kmers(const std::vector<std::string> &seqs, int k, double distinct, const std::string &out_path, uint64_t *n_out, std::string &err)
{
	if (k < 1 || k > 28) { err = "k out of range"; return -1; }
	const uint64_t mask = (1ULL << 2 * k) - 1;
	std::vector<uint64_t> kmers;           // sort-based counting (k > 16) or direct table (k <= 16)
	std::vector<uint32_t> table;
	const bool direct = k <= 16;
	if (direct) table.assign((size_t)1 << 2 * k, 0);
	for (const std::string &s : seqs) {
		uint64_t fw = 0, rc = 0;
		int run = 0;
		for (size_t i = 0; i < s.size(); ++i) {
			const int c = nt4_table[(uint8_t)s[i]];
			if (c > 3) { run = 0; continue; }
			fw = (fw << 2 | (uint64_t)c) & mask;
			rc = rc >> 2 | (3ULL ^ (uint64_t)c) << (2 * (k - 1));
			if (++run < k) continue;
			const uint64_t km = fw < rc ? fw : rc;
			if (direct) { if (table[km] != 0xffffffffu) ++table[km]; }
			else kmers.push_back(km);
		}
	}
	std::vector<std::pair<uint64_t, uint32_t>> uniq;    // only used for the sort-based path
	std::vector<uint64_t> hist;
	uint64_t n_distinct = 0;
	auto bump = [&](uint32_t c) { if (c >= hist.size()) hist.resize((size_t)c + 1, 0); ++hist[c]; ++n_distinct; };
	if (direct) { for (uint32_t c : table) if (c) bump(c); }
	else {
		std::sort(kmers.begin(), kmers.end());
		for (size_t i = 0; i < kmers.size();) { size_t j = i; while (j < kmers.size() && kmers[j] == kmers[i]) ++j; uniq.push_back(std::make_pair(kmers[i], (uint32_t)(j - i))); bump((uint32_t)(j - i)); i = j; }
	}
	// threshold exactly as merylOp-nextMer.C:103-115: the target is TRUNCATED to an integer and only count values that occur are visited
	uint64_t cum = 0, thr = 0;
	const uint64_t target = (uint64_t)(distinct * (double)n_distinct);
	for (size_t c = 1; c < hist.size(); ++c) {
		if (hist[c] == 0) continue;
		cum += hist[c];
		if (cum >= target) { thr = c; break; }
	}
	FILE *fp = fopen(out_path.c_str(), "w");
	if (!fp) { err = "cannot write " + out_path; return -1; }
	uint64_t n = 0;
	char buf[40];
	auto emit = [&](uint64_t km, uint32_t c) {
		for (int i = 0; i < k; ++i) buf[i] = "ACGT"[km >> (2 * (k - 1 - i)) & 3];
		buf[k] = 0;
		fprintf(fp, "%s\t%u\n", buf, c); ++n;
	};
	if (direct) { for (uint64_t km = 0; km < table.size(); ++km) if (table[km] > thr) emit(km, table[km]); }
	else for (auto &p : uniq) if (p.second > thr) emit(p.first, p.second);
	fclose(fp);
	if (n_out) *n_out = n;
	return 0;
}

} // namespace wm
