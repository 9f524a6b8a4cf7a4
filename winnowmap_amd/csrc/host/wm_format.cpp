#include "wm_format.h"
#include <stdio.h>
namespace wm {

// decimal digits without printf: a record of a 15-kb read carries ~5 000 integers (one per CIGAR operation), and snprintf("%lld") for each of them was
// a tenth of the host's CPU samples in a bench run (gpurun_out/r05a/sprof_report.txt: libc's vfprintf internals)
static inline void put_int(std::string &s, long long v)
{
	char b[24];
	int n = 24;
	unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
	do { b[--n] = (char)('0' + u % 10); u /= 10; } while (u);
	if (v < 0) b[--n] = '-';
	s.append(b + n, (size_t)(24 - n));
}

void write_sam_header(std::string &s, const Index &idx, int argc, const char *const *argv)
{
	for (const RefSeq &r : idx.seq) { s += "@SQ\tSN:"; s += r.name; s += "\tLN:"; put_int(s, r.len); s += '\n'; }
	s += "@PG\tID:Winnowmap\tPN:Winnowmap\tVN:2.03";
	if (argc > 1) { s += "\tCL:winnowmap"; for (int i = 1; i < argc; ++i) { s += ' '; s += argv[i]; } }
	s += '\n';
}

static double event_identity(const Reg &r)
{   // mm_event_identity, src/format.c:268-278
	int n_gapo = 0, n_gap = 0;
	for (uint32_t c : r.cigar) { const int op = c & 0xf, len = c >> 4; if (op == 1 || op == 2) ++n_gapo, n_gap += len; }
	return (double)r.mlen / (r.blen + (int)r.n_ambi - n_gap + n_gapo);
}

static void write_tags(std::string &s, const Reg &r)
{   // src/format.c:280-306
	const char type = r.id == r.parent ? (r.inv ? 'I' : 'P') : (r.inv ? 'i' : 'S');
	if (r.has_p) {
		s += "\tNM:i:"; put_int(s, r.blen - r.mlen + (int)r.n_ambi);
		s += "\tms:i:"; put_int(s, r.dp_max);
		s += "\tAS:i:"; put_int(s, r.dp_score);
		s += "\tnn:i:"; put_int(s, r.n_ambi);
		if (r.trans_strand == 1 || r.trans_strand == 2) { s += "\tts:A:"; s += "?+-?"[r.trans_strand]; }
	}
	s += "\ttp:A:"; s += type;
	s += "\tcm:i:"; put_int(s, r.cnt);
	s += "\ts1:i:"; put_int(s, r.score);
	if (r.parent == r.id) { s += "\ts2:i:"; put_int(s, r.subsc); }
	if (r.has_p) {
		char buf[16];
		const double div = 1.0 - event_identity(r);
		if (div == 0.0) buf[0] = '0', buf[1] = 0;
		else snprintf(buf, 16, "%.4f", 1.0 - event_identity(r));
		s += "\tde:f:"; s += buf;
	} else if (r.div >= 0.0f && r.div <= 1.0f) {
		char buf[16];
		if (r.div == 0.0f) buf[0] = '0', buf[1] = 0;
		else snprintf(buf, 16, "%.4f", r.div);
		s += "\tdv:f:"; s += buf;
	}
	if (r.split) { s += "\tzd:i:"; put_int(s, r.split); }
}

// cs:Z / MD:Z tags (write_cs_or_MD, src/format.c:140-231): walk the CIGAR over the aligned target / query codes
static void write_cs_or_md(std::string &s, const Index &idx, const ReadIn &t, const Reg &r, int64_t flag)
{
	if (!r.has_p) return;
	const int ql = r.qe - r.qs, tl = r.re - r.rs;
	std::vector<uint8_t> tseq(tl > 0 ? tl : 1), qseq(ql > 0 ? ql : 1);
	if (tl > 0) idx.getseq(r.rid, r.rs, r.re, tseq.data());
	for (int i = r.qs; i < r.qe; ++i) {
		const uint8_t c = nt4_table[(uint8_t)t.seq[i]];
		if (!r.rev) qseq[i - r.qs] = c;
		else qseq[r.qe - i - 1] = c >= 4 ? 4 : 3 - c;
	}
	static const char *lower = "acgtn", *upper = "ACGTN";
	int q_off = 0, t_off = 0;
	if (flag & F_OUT_MD) {
		s += "\tMD:Z:";
		int run = 0;
		for (uint32_t c : r.cigar) {
			const int op = c & 0xf, len = (int)(c >> 4);
			if (op == 0 || op == 7 || op == 8) {
				for (int j = 0; j < len; ++j) {
					if (qseq[q_off + j] != tseq[t_off + j]) { put_int(s, run); s += upper[tseq[t_off + j]]; run = 0; }
					else ++run;
				}
				q_off += len; t_off += len;
			} else if (op == 1) q_off += len;
			else if (op == 2) {
				put_int(s, run); s += '^';
				for (int j = 0; j < len; ++j) s += upper[tseq[t_off + j]];
				run = 0; t_off += len;
			} else if (op == 3) t_off += len;
		}
		if (run > 0) put_int(s, run);
		return;
	}
	const bool long_form = (flag & F_OUT_CS_LONG) != 0;
	s += "\tcs:Z:";
	for (uint32_t c : r.cigar) {
		const int op = c & 0xf, len = (int)(c >> 4);
		if (op == 0 || op == 7 || op == 8) {
			int run = 0;
			auto flush = [&](int end) {                                    // identical stretch ending before position `end`
				if (run == 0) return;
				if (long_form) { s += '='; for (int j = end - run; j < end; ++j) s += upper[qseq[q_off + j]]; }
				else { s += ':'; put_int(s, run); }
				run = 0;
			};
			for (int j = 0; j < len; ++j) {
				if (qseq[q_off + j] != tseq[t_off + j]) { flush(j); s += '*'; s += lower[tseq[t_off + j]]; s += lower[qseq[q_off + j]]; }
				else ++run;
			}
			flush(len);
			q_off += len; t_off += len;
		} else if (op == 1) { s += '+'; for (int j = 0; j < len; ++j) s += lower[qseq[q_off + j]]; q_off += len; }
		else if (op == 2) { s += '-'; for (int j = 0; j < len; ++j) s += lower[tseq[t_off + j]]; t_off += len; }
		else {                                                             // intron (splice mode only)
			s += '~'; s += lower[tseq[t_off]]; s += lower[tseq[t_off + 1]]; put_int(s, len); s += lower[tseq[t_off + len - 2]]; s += lower[tseq[t_off + len - 1]];
			t_off += len;
		}
	}
}

static void write_paf(std::string &s, const Index &idx, const ReadIn &t, const Reg *r, int64_t flag, int rep_len)
{   // mm_write_paf3, src/format.c:308-334
	const int l_seq = (int)t.seq.size();
	if (r == 0) {
		s += t.name; s += '\t'; put_int(s, l_seq); s += "\t0\t0\t*\t*\t0\t0\t0\t0\t0\t0";
		if (rep_len >= 0) { s += "\trl:i:"; put_int(s, rep_len); }
		return;
	}
	s += t.name; s += '\t'; put_int(s, l_seq); s += '\t'; put_int(s, r->qs); s += '\t'; put_int(s, r->qe); s += '\t'; s += "+-"[r->rev]; s += '\t';
	s += idx.seq[r->rid].name;
	s += '\t'; put_int(s, idx.seq[r->rid].len); s += '\t'; put_int(s, r->rs); s += '\t'; put_int(s, r->re);
	s += '\t'; put_int(s, r->mlen); s += '\t'; put_int(s, r->blen); s += '\t'; put_int(s, r->mapq);
	write_tags(s, *r);
	if (rep_len >= 0) { s += "\trl:i:"; put_int(s, rep_len); }
	if (r->has_p && (flag & F_OUT_CG)) {
		s += "\tcg:Z:";
		s.reserve(s.size() + r->cigar.size() * 4 + 64);
		for (uint32_t c : r->cigar) { put_int(s, c >> 4); s += "MIDNSHP=XB"[c & 0xf]; }
	}
	if (r->has_p && (flag & (F_OUT_CS | F_OUT_MD))) write_cs_or_md(s, idx, t, *r, flag);
	if ((flag & F_COPY_COMMENT) && !t.comment.empty()) { s += '\t'; s += t.comment; }
}

static void put_seq(std::string &s, const char *seq, int l, bool rev, bool comp)
{   // sam_write_sq, src/format.c:341-353
	if (!rev) { s.append(seq, l); return; }
	for (int i = 0; i < l; ++i) {
		char c = seq[l - 1 - i];
		if (comp) {
			switch (c) {   // seq_comp_table (src/bseq.c): IUPAC complement, case preserved
			case 'A': c = 'T'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break; case 'T': c = 'A'; break; case 'U': c = 'A'; break;
			case 'a': c = 't'; break; case 'c': c = 'g'; break; case 'g': c = 'c'; break; case 't': c = 'a'; break; case 'u': c = 'a'; break;
			case 'M': c = 'K'; break; case 'K': c = 'M'; break; case 'R': c = 'Y'; break; case 'Y': c = 'R'; break;
			case 'V': c = 'B'; break; case 'B': c = 'V'; break; case 'H': c = 'D'; break; case 'D': c = 'H'; break;
			case 'm': c = 'k'; break; case 'k': c = 'm'; break; case 'r': c = 'y'; break; case 'y': c = 'r'; break;
			case 'v': c = 'b'; break; case 'b': c = 'v'; break; case 'h': c = 'd'; break; case 'd': c = 'h'; break;
			default: break;
			}
		}
		s += c;
	}
}

static void write_sam(std::string &s, const Index &idx, const ReadIn &t, const std::vector<Reg> &regs, int reg_idx, int64_t flag_opt, int rep_len)
{   // mm_write_sam3 for one segment, src/format.c:391-548
	const int l_seq = (int)t.seq.size(), n_regs = (int)regs.size();
	const Reg *r = reg_idx >= 0 && reg_idx < n_regs ? &regs[reg_idx] : 0;
	int flag = 0;
	s += t.name;
	if (r == 0) flag |= 0x4;
	else { if (r->rev) flag |= 0x10; if (r->parent != r->id) flag |= 0x100; else if (!r->sam_pri) flag |= 0x800; }
	s += '\t'; put_int(s, flag);
	bool cigar_in_tag = false;
	const int clip_char_tag = 0;
	(void)clip_char_tag;
	if (r == 0) s += "\t*\t0\t0\t*";
	else {
		s += '\t'; s += idx.seq[r->rid].name; s += '\t'; put_int(s, r->rs + 1); s += '\t'; put_int(s, r->mapq); s += '\t';
		if ((flag_opt & F_LONG_CIGAR) && r->has_p && r->cigar.size() > 65535 - 2) {
			int n_cigar = (int)r->cigar.size();
			if (r->qs != 0) ++n_cigar;
			if (r->qe != l_seq) ++n_cigar;
			if (n_cigar > 65535) cigar_in_tag = true;
		}
		if (cigar_in_tag) {
			int slen;
			if ((flag & 0x900) == 0 || (flag_opt & F_SOFTCLIP)) slen = l_seq;
			else if (flag & 0x100) slen = 0;
			else slen = r->qe - r->qs;
			put_int(s, slen); s += 'S'; put_int(s, r->re - r->rs); s += 'N';
		} else if (!r->has_p) s += '*';
		else {
			const uint32_t clip0 = r->rev ? l_seq - r->qe : r->qs, clip1 = r->rev ? r->qs : l_seq - r->qe;
			const char cc = (flag & 0x800) && !(flag_opt & F_SOFTCLIP) ? 'H' : 'S';
			if (clip0) { put_int(s, clip0); s += cc; }
			for (uint32_t c : r->cigar) { put_int(s, c >> 4); s += "MIDNSHP=XB"[c & 0xf]; }
			if (clip1) { put_int(s, clip1); s += cc; }
		}
	}
	s += "\t*\t0\t0\t";
	const bool has_qual = !t.qual.empty() && !(flag_opt & F_NO_QUAL);
	if (r == 0) {
		s += t.seq; s += '\t';
		if (has_qual) s += t.qual; else s += '*';
	} else if ((flag & 0x900) == 0 || (flag_opt & F_SOFTCLIP)) {
		put_seq(s, t.seq.data(), l_seq, r->rev, r->rev); s += '\t';
		if (has_qual) put_seq(s, t.qual.data(), l_seq, r->rev, false); else s += '*';
	} else if (flag & 0x100) s += "*\t*";
	else {
		put_seq(s, t.seq.data() + r->qs, r->qe - r->qs, r->rev, r->rev); s += '\t';
		if (has_qual) put_seq(s, t.qual.data() + r->qs, r->qe - r->qs, r->rev, false); else s += '*';
	}
	if (r) {
		write_tags(s, *r);
		if (r->parent == r->id && r->has_p && n_regs > 1) {               // SA: the other primary-chain alignments
			int n_sa = 0;
			for (int i = 0; i < n_regs; ++i) if (i != reg_idx && regs[i].parent == regs[i].id && regs[i].has_p) ++n_sa;
			if (n_sa > 0) {
				s += "\tSA:Z:";
				for (int i = 0; i < n_regs; ++i) {
					const Reg *q = &regs[i];
					if (r == q || q->parent != q->id || !q->has_p) continue;
					int l_M, l_I = 0, l_D = 0;
					if (q->qe - q->qs < q->re - q->rs) l_M = q->qe - q->qs, l_D = (q->re - q->rs) - l_M;
					else l_M = q->re - q->rs, l_I = (q->qe - q->qs) - l_M;
					const int clip5 = q->rev ? l_seq - q->qe : q->qs, clip3 = q->rev ? q->qs : l_seq - q->qe;
					s += idx.seq[q->rid].name; s += ','; put_int(s, q->rs + 1); s += ','; s += "+-"[q->rev]; s += ',';
					if (clip5) { put_int(s, clip5); s += 'S'; }
					if (l_M) { put_int(s, l_M); s += 'M'; }
					if (l_I) { put_int(s, l_I); s += 'I'; }
					if (l_D) { put_int(s, l_D); s += 'D'; }
					if (clip3) { put_int(s, clip3); s += 'S'; }
					s += ','; put_int(s, q->mapq); s += ','; put_int(s, q->blen - q->mlen + (int)q->n_ambi); s += ';';
				}
			}
		}
		if (r->has_p && (flag_opt & (F_OUT_CS | F_OUT_MD))) write_cs_or_md(s, idx, t, *r, flag_opt);
		if (cigar_in_tag) {
			const uint32_t clip0 = r->rev ? l_seq - r->qe : r->qs, clip1 = r->rev ? r->qs : l_seq - r->qe;
			const int cc = (flag & 0x800) && !(flag_opt & F_SOFTCLIP) ? 5 : 4;
			s += "\tCG:B:I";
			if (clip0) { s += ','; put_int(s, clip0 << 4 | cc); }
			for (uint32_t c : r->cigar) { s += ','; put_int(s, c); }
			if (clip1) { s += ','; put_int(s, clip1 << 4 | cc); }
		}
	}
	if (rep_len >= 0) { s += "\trl:i:"; put_int(s, rep_len); }
	if ((flag_opt & F_COPY_COMMENT) && !t.comment.empty()) { s += '\t'; s += t.comment; }
}

void write_read(std::string &s, const Index &idx, const ReadIn &rd, const ReadOut &out, int64_t flag)
{   // output step of worker_pipeline, src/map.c:1189-1207
	const int n = (int)out.regs.size();
	if (n > 0) {
		for (int j = 0; j < n; ++j) {
			const Reg &r = out.regs[j];
			if ((flag & F_NO_PRINT_2ND) && r.id != r.parent) continue;
			if (flag & F_OUT_SAM) write_sam(s, idx, rd, out.regs, j, flag, out.rep_len);
			else write_paf(s, idx, rd, &r, flag, out.rep_len);
			s += '\n';
		}
	} else if ((flag & F_PAF_NO_HIT) || ((flag & F_OUT_SAM) && !(flag & F_SAM_HIT_ONLY))) {
		if (flag & F_OUT_SAM) write_sam(s, idx, rd, out.regs, -1, flag, out.rep_len);
		else write_paf(s, idx, rd, 0, flag, out.rep_len);
		s += '\n';
	}
}

} // namespace wm
