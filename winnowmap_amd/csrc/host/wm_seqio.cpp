// wm_seqio.cpp — FASTA/FASTQ (optionally gzip) reader with the record semantics of the reference's kseq/bseq
// layer (src/bseq.c:66-78, src/kseq.h): name = header up to the first blank, the rest is the comment,
// multi-line sequences are concatenated, 'U'/'u' become 'T'/'t'. Plain host I/O.
#include <zlib.h>
#include <string>
#include <vector>
#include <string.h>

namespace wm {

namespace {
struct GzLines {
	gzFile fp;
	std::vector<char> buf;
	size_t pos = 0, end = 0;
	bool eof = false;
	explicit GzLines(gzFile f) : fp(f), buf(1 << 20) {}
	bool getline(std::string &ln)
	{
		ln.clear();
		for (;;) {
			if (pos == end) {
				if (eof) return !ln.empty();
				int n = gzread(fp, buf.data(), (unsigned)buf.size());
				if (n <= 0) { eof = true; return !ln.empty(); }
				pos = 0; end = (size_t)n;
			}
			const char *s = buf.data() + pos;
			const char *nl = (const char*)memchr(s, '\n', end - pos);
			if (nl) { ln.append(s, nl - s); pos += (size_t)(nl - s) + 1; if (!ln.empty() && ln.back() == '\r') ln.pop_back(); return true; }
			ln.append(s, end - pos);
			pos = end;
		}
	}
};
}

} // namespace wm
#include "wm_mapper.h"
#include "wm_pipeline.h"
namespace wm {

// streaming reader: same record semantics, one mini-batch at a time (mm_bseq_read3, src/bseq.c:80-118: records are taken
// until their summed length reaches the chunk size)
struct FastxReader::Impl { gzFile fp; GzLines in; std::string ln; bool have; Impl(gzFile f) : fp(f), in(f), have(false) {} };

FastxReader::FastxReader() : p_(0) {}
FastxReader::~FastxReader() { close(); }
void FastxReader::close() { if (p_) { gzclose(p_->fp); delete p_; p_ = 0; } }
int FastxReader::open(const std::string &fn, std::string &err)
{
	close();
	gzFile fp = fn == "-" ? gzdopen(0, "r") : gzopen(fn.c_str(), "r");
	if (!fp) { err = "failed to open file '" + fn + "'"; return -1; }
	p_ = new Impl(fp);
	p_->have = p_->in.getline(p_->ln);
	return 0;
}
// appends up to ~max_bases of records to `out`; returns the number of records read (0 = end of file)
int FastxReader::next_batch(int64_t max_bases, bool with_qual, std::vector<ReadIn> &out)
{
	if (!p_) return 0;
	GzLines &in = p_->in;
	std::string &ln = p_->ln;
	bool &have = p_->have;
	int n = 0;
	int64_t bases = 0;
	while (have) {
		if (ln.empty() || (ln[0] != '>' && ln[0] != '@')) { have = in.getline(ln); continue; }
		const bool fq = ln[0] == '@';
		ReadIn r;
		size_t sp = ln.find_first_of(" \t");
		r.name = ln.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
		r.comment = sp == std::string::npos ? std::string() : ln.substr(sp + 1);
		have = in.getline(ln);
		while (have && !(ln.size() && (ln[0] == '>' || ln[0] == '+' || (ln[0] == '@' && !fq)))) {
			if (fq && ln.size() && ln[0] == '@' && !r.seq.empty()) break;
			for (char c : ln) if (c > ' ') r.seq.push_back(c);
			have = in.getline(ln);
		}
		if (have && ln.size() && ln[0] == '+') {
			std::string qual;
			have = in.getline(ln);
			while (have && qual.size() < r.seq.size()) { qual += ln; have = in.getline(ln); }
			if (with_qual) r.qual = std::move(qual);
		}
		for (char &c : r.seq) if (c == 'u' || c == 'U') --c;
		bases += (int64_t)r.seq.size();
		out.push_back(std::move(r));
		++n;
		if (bases >= max_bases) break;
	}
	return n;
}

int read_fastx(const std::string &fn, std::vector<std::string> &names, std::vector<std::string> &seqs,
               std::vector<std::string> *quals, std::vector<std::string> *comments, std::string &err)
{
	gzFile fp = fn == "-" ? gzdopen(0, "r") : gzopen(fn.c_str(), "r");
	if (!fp) { err = "failed to open file '" + fn + "'"; return -1; }
	GzLines in(fp);
	std::string ln, pending;
	bool have = in.getline(ln);
	while (have) {
		if (ln.empty() || (ln[0] != '>' && ln[0] != '@')) { have = in.getline(ln); continue; }
		const bool fq = ln[0] == '@';
		size_t sp = ln.find_first_of(" \t");
		names.push_back(ln.substr(1, sp == std::string::npos ? std::string::npos : sp - 1));
		if (comments) comments->push_back(sp == std::string::npos ? std::string() : ln.substr(sp + 1));
		std::string seq, qual;
		have = in.getline(ln);
		while (have && !(ln.size() && (ln[0] == '>' || ln[0] == '+' || (ln[0] == '@' && !fq)))) {
			if (fq && ln.size() && ln[0] == '@' && !seq.empty()) break;
			for (char c : ln) if (c > ' ') seq.push_back(c);
			have = in.getline(ln);
		}
		if (have && ln.size() && ln[0] == '+') {                 // quality block: as many characters as bases
			have = in.getline(ln);
			while (have && qual.size() < seq.size()) { qual += ln; have = in.getline(ln); }
		}
		for (char &c : seq) if (c == 'u' || c == 'U') --c;
		seqs.push_back(std::move(seq));
		if (quals) quals->push_back(std::move(qual));
	}
	gzclose(fp);
	return (int)seqs.size();
}

} // namespace wm
