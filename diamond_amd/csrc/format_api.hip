// format_api.hip -- output formats of the hot path's records (host code only; part of libdiamond_hip.so so that the CLI, the
// Python mirror and a reference-side binding print through the same code).
// Mirrors: TabularFormat (src/output/blast_tab_format.cpp:46-620), PairwiseFormat (src/output/blast_pairwise_format.cpp:24-85),
// print_cigar (src/output/sam_format.cpp:67-84), HspContext::Iterator (src/basic/match.h:300-370), TextBuffer number printing
// (src/util/text_buffer.h:224-254), OutputFormat::print_title (src/output/output_format.cpp:150-168).
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/diamond_hip.h"
#include "ctx.h"

using namespace dmnd;

namespace {

const char* const AA = "ARNDCQEGHILKMFPSTWYVBJZX*_";      // amino_acid_traits.alphabet (basic/value.cpp:25)
const char* const NT = "ACGTN";
const char* const FIELD_NAMES[DMND_F_COUNT] = { "qseqid", "qlen", "sseqid", "sallseqid", "slen", "qstart", "qend", "sstart", "send", "qseq", "sseq",
	"evalue", "bitscore", "score", "length", "pident", "nident", "mismatch", "positive", "gapopen", "gaps", "ppos", "qframe", "btop", "stitle",
	"salltitles", "qcovhsp", "qtitle", "full_sseq", "qnum", "snum", "scovhsp", "full_qseq", "qseq_gapped", "sseq_gapped", "qstrand", "cigar",
	"qseq_translated", "hspnum" };
// fields of the reference that this build does not print
const char* const UNAVAILABLE[] = { "staxids", "sscinames", "sskingdoms", "skingdoms", "sphylums", "slineages", "qqual", "full_qqual", "full_qseq_mate",
	"normalized_bitscore", "normalized_bitscore_query", "normalized_nident", "approx_pident", "corrected_bitscore" };

enum { OP_MATCH = 0, OP_INSERTION = 1, OP_DELETION = 2, OP_SUBSTITUTION = 3 };

std::atomic<uint32_t> g_format_flags(0);        // dmnd_set_format_flags: the reference reads these from its global config

struct Out {
	std::string s;
	Out& operator<<(const char* x) { s += x; return *this; }
	Out& operator<<(char x) { s += x; return *this; }
	Out& operator<<(long long x) { s += std::to_string(x); return *this; }
	Out& operator<<(int x) { return *this << (long long)x; }
	Out& operator<<(unsigned x) { return *this << (long long)x; }
	// Util::String::format_double (util/string/string.h:87-92)
	Out& operator<<(double x)
	{
		char b[48];
		if (x >= 100.0) std::snprintf(b, sizeof b, "%lli", (long long)std::floor(x));
		else { const long long i = std::llround(x * 10.0); std::snprintf(b, sizeof b, "%lli.%lli", i / 10, i % 10); }
		s += b;
		return *this;
	}
	void print_e(double x)
	{
		char b[48];
		if (x == 0.0) std::snprintf(b, sizeof b, "0.0"); else std::snprintf(b, sizeof b, "%.2e", x);
		s += b;
	}
	// TextBuffer::print(unsigned, width): right-aligned, and only `width` characters are kept
	void print_width(unsigned i, unsigned width)
	{
		char b[32];
		std::snprintf(b, 16, "%*u", (int)width, i);
		s.append(b, std::min<size_t>(width, std::strlen(b)));
	}
	void until(const char* t, const char* delims) { s.append(t, std::strcspn(t, delims)); }
};

const char* const ID_DELIMITERS = " \a\b\f\n\r\t\v\1";   // Util::Seq::id_delimiters

// OutputFormat::print_title: the titles of a record are separated by \1 or " >"
void print_title(Out& o, const char* id, bool full_titles, bool all_titles, const char* separator)
{
	const char* p = id;
	int n = 0;
	for (;;) {
		const char* a = std::strchr(p, '\1');
		const char* b = std::strstr(p, " >");
		const char* e = a && b ? std::min(a, b) : a ? a : b;
		const size_t len = e ? (size_t)(e - p) : std::strlen(p);
		if (n++ > 0) o << separator;
		const std::string tok(p, len);
		if (full_titles) o.s += tok; else o.until(tok.c_str(), ID_DELIMITERS);
		if (!e || !all_titles) break;
		p = e + (*e == '\1' ? 1 : 2);
	}
}

// HspContext::Iterator: one alignment column at a time
// (frameshift alignments, blastx -F: a substitution whose letter is 26 / 27 is a frame shift back / forward, packed_transcript.h:44-58.
// The cursor then leaves its frame -- TranslatedPosition::shift_forward / shift_back, translated_position.h:97-113 -- and reads the
// query letters from the view's three frames of the strand)
enum { FS_REVERSE = 26, FS_FORWARD = 27 };
struct Walk {
	const dmnd_hsp_view& v;
	const uint8_t* t;
	int left, count = 0, op = 0, letter = 0;
	int qpos, spos, foff;
	bool ok = false;
	explicit Walk(const dmnd_hsp_view& view) : v(view), t(view.transcript), left(view.match->hsp.transcript_len), qpos(view.match->hsp.q_begin), spos(view.match->hsp.s_begin),
		foff(view.match->frame % 3) { fetch(); }
	int shift() const { return op == OP_SUBSTITUTION ? (letter == FS_FORWARD ? 1 : letter == FS_REVERSE ? -1 : 0) : 0; }
	int in_strand() const { return v.source_seq ? foff + 3 * qpos : qpos; }      // TranslatedPosition::in_strand of the cursor's frame
	void fetch()
	{
		ok = false;
		while (left > 0) {
			const uint8_t b = *t;
			op = b >> 6;
			if (op == OP_MATCH || op == OP_INSERTION) { count = b & 63; letter = 0; }
			else { count = 1; letter = b & 63; }
			++t; --left;
			if (count > 0) { ok = true; return; }
		}
	}
	bool good() const { return ok; }
	void next()
	{
		const int s = shift();
		if (s > 0) { if (++foff == 3) { foff = 0; ++qpos; } }
		else if (s < 0) { if (--foff < 0) { foff = 2; --qpos; } }
		else {
			if (op != OP_DELETION) ++qpos;
			if (op != OP_INSERTION) ++spos;
		}
		if (--count == 0) fetch();
	}
	int query() const { return (v.qframes[foff] ? v.qframes[foff][qpos] : v.qseq[qpos]) & 31; }
	int subject() const { return (op == OP_MATCH || op == OP_INSERTION) ? query() : letter; }
	char query_char() const { const int s = shift(); return s > 0 ? '\\' : s < 0 ? '/' : op == OP_DELETION ? '-' : AA[query()]; }
	char subject_char() const { return op == OP_INSERTION || shift() ? '-' : AA[subject()]; }
};

struct Frame {
	bool translated; int offset; bool forward; int dna_len;
	explicit Frame(const dmnd_hsp_view& v) : translated(v.source_seq != nullptr), offset(v.match->frame % 3), forward(v.match->frame < 3), dna_len(v.source_seq ? v.source_len : v.qlen) {}
	int in_strand(int pos) const { return translated ? offset + 3 * pos : pos; }                 // TranslatedPosition::in_strand
	int oriented(int p) const { return forward ? p : dna_len - 1 - p; }                           // oriented_position
	int absolute(int pos) const { return oriented(in_strand(pos)); }
	int blast_frame() const { return translated ? (forward ? offset + 1 : -(offset + 1)) : 0; }   // Hsp::blast_query_frame
};

// Hsp::query_source_range (basic/match.h, TranslatedPosition::absolute_interval): the DNA interval of the aligned codons
void source_range(const dmnd_hsp_view& v, int& b, int& e)
{
	const Frame f(v);
	const dmnd_hsp& h = v.match->hsp;
	if (!f.translated) { b = h.q_begin; e = h.q_end; return; }
	if (v.match->read_end > v.match->read_begin) { b = v.match->read_begin; e = v.match->read_end; return; }      // frameshift alignment: the range is part of the record
	if (f.forward) { b = f.offset + 3 * h.q_begin; e = f.offset + 3 * h.q_end; }
	else { e = f.dna_len - f.offset - 3 * h.q_begin; b = f.dna_len - f.offset - 3 * h.q_end; }
}

int need_transcript(int id)
{
	switch (id) {
	case DMND_F_SSEQ: case DMND_F_BTOP: case DMND_F_QSEQ_GAPPED: case DMND_F_SSEQ_GAPPED: case DMND_F_CIGAR: return 1;
	default: return 0;
	}
}

// print_cigar (sam_format.cpp:67-84): runs of M (match + substitution), I, D, \ (frame shift forward), / (back)
void print_cigar(Out& o, const dmnd_hsp_view& v)
{
	static const int map[6] = { 0, 1, 2, 0, 3, 4 };
	static const char letter[5] = { 'M', 'I', 'D', '\\', '/' };
	unsigned n = 0;
	int op = 0;
	const uint8_t* t = v.transcript;
	for (int i = 0; i < v.match->hsp.transcript_len; ++i) {
		int o2 = t[i] >> 6;
		const int cnt = (o2 == OP_MATCH || o2 == OP_INSERTION) ? (t[i] & 63) : 1;
		if (o2 == OP_SUBSTITUTION && (t[i] & 63) == FS_FORWARD) o2 = 4;
		else if (o2 == OP_SUBSTITUTION && (t[i] & 63) == FS_REVERSE) o2 = 5;
		if (map[o2] == op) n += (unsigned)cnt;
		else { if (n > 0) { o << n << letter[op]; } n = (unsigned)cnt; op = map[o2]; }
	}
	if (n > 0) o << n << letter[op];
}

int print_field(Out& o, const dmnd_hsp_view& v, int id)
{
	const dmnd_match& m = *v.match;
	const dmnd_hsp& h = m.hsp;
	const Frame f(v);
	if (need_transcript(id) && !v.transcript) return fail(DMND_E_ARG, std::string("dmnd_format_fields: field ") + FIELD_NAMES[id] + " needs the transcript");
	int sb, se;
	source_range(v, sb, se);
	switch (id) {
	case DMND_F_QSEQID: o.until(v.qtitle, ID_DELIMITERS); break;
	case DMND_F_QLEN: o << (f.translated ? v.source_len : v.qlen); break;
	case DMND_F_SSEQID: print_title(o, v.stitle, false, false, ""); break;
	case DMND_F_SALLSEQID: print_title(o, v.stitle, false, true, ";"); break;
	case DMND_F_SLEN: o << v.slen; break;
	// oriented_query_range: reverse frames print qstart > qend
	case DMND_F_QSTART: o << (f.translated ? (f.forward ? sb + 1 : se) : h.q_begin + 1); break;
	case DMND_F_QEND: o << (f.translated ? (f.forward ? se : sb + 1) : h.q_end); break;
	case DMND_F_SSTART: o << h.s_begin + 1; break;
	case DMND_F_SEND: o << h.s_end; break;
	case DMND_F_QSEQ:                                       // the source sequence over query_source_range: DNA for a translated search
		if (f.translated) for (int i = sb; i < se; ++i) o << NT[v.source_seq[i] & 7];
		else for (int i = h.q_begin; i < h.q_end; ++i) o << AA[v.qseq[i] & 31];
		break;
	case DMND_F_SSEQ:
		// (a frame-shift column is not skipped: the reference's iterator answers its subject() with the query letter it stands on,
		// blast_tab_format.cpp:269-276 with match.h:325-335)
		for (Walk w(v); w.good(); w.next()) if (w.op != OP_INSERTION) o << AA[w.shift() ? w.query() : w.subject()];
		break;
	case DMND_F_EVALUE: o.print_e(m.evalue); break;
	case DMND_F_BITSCORE: o << m.bit_score; break;
	case DMND_F_SCORE: o << h.score; break;
	case DMND_F_LENGTH: o << h.length; break;
	case DMND_F_PIDENT: o << (double)h.identities * 100.0 / (double)h.length; break;
	case DMND_F_NIDENT: o << h.identities; break;
	case DMND_F_MISMATCH: o << h.mismatches; break;
	case DMND_F_POSITIVE: o << h.positives; break;
	case DMND_F_GAPOPEN: o << h.gap_openings; break;
	case DMND_F_GAPS: o << h.gaps; break;
	case DMND_F_PPOS: o << (double)h.positives * 100.0 / h.length; break;
	case DMND_F_QFRAME: o << f.blast_frame(); break;
	case DMND_F_BTOP: {
		unsigned n_matches = 0;
		for (Walk w(v); w.good(); w.next()) {
			if (w.op == OP_MATCH) { ++n_matches; continue; }
			if (n_matches > 0) { o << n_matches; n_matches = 0; }
			if (w.op == OP_SUBSTITUTION) o << w.query_char() << w.subject_char();      // (a frame shift prints as \\- or /-: blast_tab_format.cpp:372-380)
			else if (w.op == OP_INSERTION) o << w.query_char() << '-';
			else o << '-' << w.subject_char();
		}
		if (n_matches > 0) o << n_matches;
		break;
	}
	case DMND_F_STITLE: print_title(o, v.stitle, true, false, "<>"); break;
	case DMND_F_SALLTITLES: print_title(o, v.stitle, true, true, "<>"); break;
	case DMND_F_QCOVHSP: o << (double)(se - sb) * 100.0 / (f.translated ? v.source_len : v.qlen); break;
	case DMND_F_QTITLE: o << v.qtitle; break;
	case DMND_F_FULL_SSEQ:
		if (!v.full_sseq) return fail(DMND_E_ARG, "dmnd_format_fields: full_sseq needs the target letters");
		for (int i = 0; i < v.slen; ++i) o << AA[v.full_sseq[i] & 31];
		break;
	case DMND_F_QNUM: o << (long long)v.qnum; break;
	case DMND_F_SNUM: o << (long long)v.snum; break;
	case DMND_F_SCOVHSP: o << (double)(h.s_end - h.s_begin) * 100.0 / v.slen; break;
	case DMND_F_FULL_QSEQ:
		if (f.translated) for (int i = 0; i < v.source_len; ++i) o << NT[v.source_seq[i] & 7];
		else for (int i = 0; i < v.qlen; ++i) o << AA[v.qseq[i] & 31];
		break;
	case DMND_F_QSEQ_GAPPED: for (Walk w(v); w.good(); w.next()) o << w.query_char(); break;
	case DMND_F_SSEQ_GAPPED: for (Walk w(v); w.good(); w.next()) o << w.subject_char(); break;
	case DMND_F_QSTRAND: o << (f.translated ? (f.blast_frame() > 0 ? '+' : '-') : '+'); break;
	case DMND_F_CIGAR: print_cigar(o, v); break;
	case DMND_F_QSEQ_TRANSLATED:
		// with a frameshift penalty set the reference reads the translated query along the alignment (blast_tab_format.cpp:565-574)
		if ((g_format_flags.load() & DMND_FMT_FRAMESHIFT) && v.transcript) { for (Walk w(v); w.good(); w.next()) if (w.op != OP_DELETION && !w.shift()) o << AA[w.query()]; }
		else for (int i = h.q_begin; i < h.q_end; ++i) o << AA[v.qseq[i] & 31];
		break;
	case DMND_F_HSPNUM: o << 0; break;                      // max_hsps = 1
	default: return fail(DMND_E_ARG, "dmnd_format_fields: unknown field id");
	}
	return DMND_OK;
}

int64_t emit(const Out& o, char* buf, int64_t cap, const char* who)
{
	if ((int64_t)o.s.size() >= cap) return fail(DMND_E_CAP, std::string(who) + ": buffer too small");
	std::memcpy(buf, o.s.data(), o.s.size());
	buf[o.s.size()] = 0;
	return (int64_t)o.s.size();
}

bool view_ok(const dmnd_hsp_view* v)
{
	return v && v->match && v->qtitle && v->stitle && v->qseq && v->match->frame >= 0 && v->match->frame <= 5 && (!v->source_seq || v->source_len > 0)
		&& v->match->hsp.q_begin >= 0 && (v->match->hsp.q_end <= v->qlen || v->match->read_end > v->match->read_begin /* frameshift alignment: q_end is a position of another frame */) && v->match->hsp.s_begin >= 0 && v->match->hsp.s_end <= v->slen && v->match->hsp.length > 0;
}

}  // namespace

extern "C" int dmnd_output_fields(const char* const* names, int n, int32_t* ids, int* needs_transcript)
{
	if (!names || n < 1 || !ids) return fail(DMND_E_ARG, "dmnd_output_fields: bad argument");
	if (needs_transcript) *needs_transcript = 0;
	for (int i = 0; i < n; ++i) {
		if (!names[i]) return fail(DMND_E_ARG, "dmnd_output_fields: NULL field name");
		int id = -1;
		for (int k = 0; k < DMND_F_COUNT; ++k) if (std::strcmp(names[i], FIELD_NAMES[k]) == 0) id = k;
		if (id < 0) {
			for (const char* u : UNAVAILABLE)
				if (std::strcmp(names[i], u) == 0) return fail(DMND_E_ARG, std::string("Output field not available in this build: ") + names[i]);
			return fail(DMND_E_ARG, std::string("Invalid output field: ") + names[i]);       // blast_tab_format.cpp:661
		}
		ids[i] = id;
		if (needs_transcript && need_transcript(id)) *needs_transcript = 1;
	}
	return DMND_OK;
}

extern "C" int64_t dmnd_format_fields(const dmnd_hsp_view* v, const int32_t* ids, int n, char* buf, int64_t cap)
{
	if (!view_ok(v) || !ids || n < 1 || !buf) return fail(DMND_E_ARG, "dmnd_format_fields: bad argument");
	Out o;
	for (int i = 0; i < n; ++i) {
		if (ids[i] < 0 || ids[i] >= DMND_F_COUNT) return fail(DMND_E_ARG, "dmnd_format_fields: unknown field id");
		if (i) o << '\t';
		if (int rc = print_field(o, *v, ids[i])) return rc;
	}
	o << '\n';
	return emit(o, buf, cap, "dmnd_format_fields");
}

extern "C" int64_t dmnd_format_pairwise_intro(const char* qtitle, int32_t qlen, int unaligned, char* buf, int64_t cap)
{
	if (!qtitle || !buf) return fail(DMND_E_ARG, "dmnd_format_pairwise_intro: NULL argument");
	Out o;
	o << "Query= " << qtitle << "\n\nLength=" << qlen << "\n\n";
	if (unaligned) o << "\n***** No hits found *****\n\n\n";
	return emit(o, buf, cap, "dmnd_format_pairwise_intro");
}

extern "C" int64_t dmnd_format_pairwise(const dmnd_hsp_view* v, const int8_t* matrix8, char* buf, int64_t cap)
{
	if (!view_ok(v) || !matrix8 || !buf) return fail(DMND_E_ARG, "dmnd_format_pairwise: bad argument");
	if (!v->transcript) return fail(DMND_E_ARG, "dmnd_format_pairwise: the pairwise format needs the transcript");
	const dmnd_match& m = *v->match;
	const dmnd_hsp& h = m.hsp;
	const Frame f(*v);
	const unsigned width = 60;
	Out o;
	o << '>';
	print_title(o, v->stitle, true, true, " ");
	o << "\nLength=" << v->slen << "\n\n";
	o << " Score = " << m.bit_score << " bits (" << h.score << "),  Expect = ";
	o.print_e(m.evalue);
	o << '\n';
	const unsigned len = (unsigned)h.length;
	o << " Identities = " << h.identities << '/' << h.length << " (" << (unsigned)h.identities * 100u / len << "%), Positives = " << h.positives << '/' << h.length
		<< " (" << (unsigned)h.positives * 100u / len << "%), Gaps = " << h.gaps << '/' << h.length << " (" << (unsigned)h.gaps * 100u / len << "%)\n";
	if (f.translated) o << " Frame = " << f.blast_frame() << '\n';
	o << '\n';
	int sb, se;
	source_range(*v, sb, se);
	const unsigned digits = (unsigned)std::max(std::ceil(std::log10((double)h.s_end)), std::ceil(std::log10((double)se)));
	Walk qi(*v), mi(*v), si(*v);
	while (qi.good()) {
		o << "Query  ";
		o.print_width((unsigned)(f.oriented(qi.in_strand()) + 1), digits);
		o << "  ";
		for (unsigned i = 0; i < width && qi.good(); ++i, qi.next()) o << qi.query_char();
		o << " " << f.oriented(qi.in_strand() - 1) + 1 << '\n';
		for (unsigned i = 0; i < digits + 9; ++i) o << ' ';
		for (unsigned i = 0; i < width && mi.good(); ++i, mi.next())
			o << (mi.op == OP_MATCH ? AA[mi.query()] : mi.op == OP_SUBSTITUTION && !mi.shift() ? (matrix8[mi.query() * 32 + mi.subject()] > 0 ? '+' : ' ') : ' ');
		o << '\n';
		o << "Sbjct  ";
		o.print_width((unsigned)(si.spos + 1), digits);
		o << "  ";
		for (unsigned i = 0; i < width && si.good(); ++i, si.next()) o << si.subject_char();
		o << " " << si.spos << "\n\n";
	}
	return emit(o, buf, cap, "dmnd_format_pairwise");
}

extern "C" int64_t dmnd_format_paf(const dmnd_hsp_view* v, const char* unaligned_qtitle, char* buf, int64_t cap)
{
	if (!buf || (!v && !unaligned_qtitle)) return fail(DMND_E_ARG, "dmnd_format_paf: bad argument");
	Out o;
	if (!v) {
		o.until(unaligned_qtitle, ID_DELIMITERS);
		o << "\t4\t*\t0\t255\t*\t*\t0\t0\t*\t*\n";
		return emit(o, buf, cap, "dmnd_format_paf");
	}
	if (!view_ok(v)) return fail(DMND_E_ARG, "dmnd_format_paf: bad argument");
	const dmnd_match& m = *v->match;
	const dmnd_hsp& h = m.hsp;
	const Frame f(*v);
	int sb, se;
	source_range(*v, sb, se);
	o.until(v->qtitle, ID_DELIMITERS);
	o << '\t' << (f.translated ? v->source_len : v->qlen) << '\t' << sb << '\t' << se - 1 << '\t' << (f.forward ? '+' : '-') << '\t';
	print_title(o, v->stitle, false, false, "<>");
	// AS = (uint32_t)ScoreMatrix::bitscore(score): the record's bit score, truncated
	o << '\t' << v->slen << '\t' << h.s_begin << '\t' << h.s_end - 1 << '\t' << h.identities << '\t' << h.length << '\t' << "255" << '\t'
		<< "AS:i:" << (long long)(uint32_t)m.bit_score << '\t' << "ZR:i:" << h.score << '\t' << "ZE:f:";
	o.print_e(m.evalue);
	o << '\n';
	return emit(o, buf, cap, "dmnd_format_paf");
}

extern "C" int64_t dmnd_format_sam(const dmnd_hsp_view* v, const char* unaligned_qtitle, char* buf, int64_t cap)
{
	if (!buf || (!v && !unaligned_qtitle)) return fail(DMND_E_ARG, "dmnd_format_sam: bad argument");
	Out o;
	if (!v) {
		o.until(unaligned_qtitle, ID_DELIMITERS);
		o << "\t4\t*\t0\t255\t*\t*\t0\t0\t*\t*\n";
		return emit(o, buf, cap, "dmnd_format_sam");
	}
	if (!view_ok(v)) return fail(DMND_E_ARG, "dmnd_format_sam: bad argument");
	if (!v->transcript) return fail(DMND_E_ARG, "dmnd_format_sam: the SAM format needs the transcript");
	const dmnd_match& m = *v->match;
	const dmnd_hsp& h = m.hsp;
	const Frame f(*v);
	int sb, se;
	source_range(*v, sb, se);
	o.until(v->qtitle, ID_DELIMITERS);
	o << '\t' << '0' << '\t';
	print_title(o, v->stitle, false, false, "<>");
	o << '\t' << h.s_begin + 1 << '\t' << "255" << '\t';
	print_cigar(o, *v);
	o << '\t' << '*' << '\t' << '0' << '\t' << '0' << '\t';
	for (int i = h.q_begin; i < h.q_end; ++i) o << AA[v->qseq[i] & 31];
	o << '\t' << '*' << '\t' << "AS:i:" << (long long)(uint32_t)m.bit_score << '\t' << "NM:i:" << h.length - h.identities << '\t' << "ZL:i:" << v->slen << '\t'
		<< "ZR:i:" << h.score << '\t' << "ZE:f:";
	o.print_e(m.evalue);
	// ZF = Frame::signed_frame (1 for untranslated queries), ZS = oriented_query_range().begin_ + 1
	o << '\t' << "ZI:i:" << h.identities * 100 / h.length << '\t' << "ZF:i:" << (f.forward ? f.offset + 1 : -(f.offset + 1)) << '\t'
		<< "ZS:i:" << (f.translated ? (f.forward ? sb + 1 : se) : h.q_begin + 1) << '\t' << "MD:Z:";
	// print_md (sam_format.cpp:30-65)
	unsigned matches = 0, del = 0;
	for (int i = 0; i < h.transcript_len; ++i) {
		const int op = v->transcript[i] >> 6, arg = v->transcript[i] & 63;
		if (op == OP_MATCH) { del = 0; matches += (unsigned)arg; }
		else if (op == OP_SUBSTITUTION && (arg == FS_FORWARD || arg == FS_REVERSE)) continue;      // a frame shift: no case of print_md's switch
		else if (op == OP_SUBSTITUTION) {
			if (matches > 0) { o << matches; matches = 0; }
			else if (del > 0) { o << '0'; del = 0; }
			o << AA[arg & 31];
		}
		else if (op == OP_DELETION) {
			if (matches > 0) { o << matches; matches = 0; }
			if (del == 0) o << '^';
			o << AA[arg & 31];
			++del;
		}
	}
	if (matches > 0) o << matches;
	if (g_format_flags.load() & DMND_FMT_SAM_QUERY_LEN) o << '\t' << "ZQ:i:" << (f.translated ? v->source_len : v->qlen);      // --sam-query-len, sam_format.cpp:129-130
	o << '\n';
	return emit(o, buf, cap, "dmnd_format_sam");
}

// ---- BLAST XML (-f 5 / xml): src/output/xml_format.cpp ------------------------------------------------------------------------
namespace {

// EscapeSequences::XML (util/util.cpp:80-88)
void xml_escaped(Out& o, const char* p, size_t n)
{
	for (size_t i = 0; i < n; ++i) {
		switch (p[i]) {
		case '"': o.s += "&quot;"; break;
		case '\'': o.s += "&apos;"; break;
		case '<': o.s += "&lt;"; break;
		case '>': o.s += "&gt;"; break;
		case '&': o.s += "&amp;"; break;
		default: o.s += p[i];
		}
	}
}

// OutputFormat::print_title(buf, id, true, all_titles, separator, &EscapeSequences::XML): the titles of a record, escaped
void xml_title(Out& o, const char* id, bool all_titles, const char* separator)
{
	const char* p = id;
	int n = 0;
	for (;;) {
		const char* a = std::strchr(p, '\1');
		const char* b = std::strstr(p, " >");
		const char* e = a && b ? std::min(a, b) : a ? a : b;
		const size_t len = e ? (size_t)(e - p) : std::strlen(p);
		if (n++ > 0) o << separator;
		xml_escaped(o, p, len);
		if (!e || !all_titles) break;
		p = e + (*e == '\1' ? 1 : 2);
	}
}

// Util::Seq::get_accession (util/sequence/sequence.cpp:76-103): UniRef prefix, gi|N|db|acc|, db|acc|name and the version suffix
std::string accession_of(std::string t)
{
	size_t i;
	if (t.compare(0, 6, "UniRef") == 0) t.erase(0, t.find('_', 0) + 1);
	else if ((i = t.find_first_of('|', 0)) != std::string::npos) {
		if (t.compare(0, 3, "gi|") == 0) {
			t.erase(0, t.find_first_of('|', i + 1) + 1);
			i = t.find_first_of('|', 0);
		}
		t.erase(0, i + 1);
		i = t.find_first_of('|', 0);
		if (i != std::string::npos) t.erase(i);
	}
	i = t.find_last_of('.');
	if (i != std::string::npos) t.erase(i);
	return t;
}

void print_lf(Out& o, double x) { char b[48]; std::snprintf(b, sizeof b, "%lf", x); o.s += b; }      // TextBuffer::print_d

}  // namespace

extern "C" int dmnd_set_format_flags(uint32_t flags)
{
	if (flags & ~(uint32_t)(DMND_FMT_XML_BLORD | DMND_FMT_NO_PARSE_SEQIDS | DMND_FMT_SAM_QUERY_LEN | DMND_FMT_FRAMESHIFT)) return fail(DMND_E_ARG, "dmnd_set_format_flags: unknown flag");
	g_format_flags.store(flags);
	return DMND_OK;
}

extern "C" int64_t dmnd_format_xml_header(const char* program, const char* version, const char* database, const char* first_qtitle, int32_t first_qlen,
	const char* matrix, int gap_open, int gap_extend, double max_evalue, char* buf, int64_t cap)
{
	if (!program || !version || !database || !first_qtitle || !matrix || !buf) return fail(DMND_E_ARG, "dmnd_format_xml_header: NULL argument");
	Out o;
	o << "<?xml version=\"1.0\"?>\n<!DOCTYPE BlastOutput PUBLIC \"-//NCBI//NCBI BlastOutput/EN\" \"http://www.ncbi.nlm.nih.gov/dtd/NCBI_BlastOutput.dtd\">\n<BlastOutput>\n"
		<< "  <BlastOutput_program>" << program << "</BlastOutput_program>\n  <BlastOutput_version>" << version << "</BlastOutput_version>\n"
		<< "  <BlastOutput_reference>Benjamin Buchfink, Xie Chao, and Daniel Huson (2015), &quot;Fast and sensitive protein alignment using DIAMOND&quot;, Nature Methods 12:59-60.</BlastOutput_reference>\n"
		<< "  <BlastOutput_db>" << database << "</BlastOutput_db>\n  <BlastOutput_query-ID>Query_1</BlastOutput_query-ID>\n  <BlastOutput_query-def>";
	// the escaped title, cut where the unescaped one has its first \1 (xml_format.cpp:120-123)
	Out esc;
	xml_escaped(esc, first_qtitle, std::strlen(first_qtitle));
	const char* sep = std::strchr(first_qtitle, '\1');
	o.s += sep ? esc.s.substr(0, (size_t)(sep - first_qtitle)) : esc.s;
	char ev[48];
	std::snprintf(ev, sizeof ev, "%g", max_evalue);                 // stringstream << double
	o << "</BlastOutput_query-def>\n  <BlastOutput_query-len>" << first_qlen << "</BlastOutput_query-len>\n  <BlastOutput_param>\n    <Parameters>\n"
		<< "      <Parameters_matrix>" << matrix << "</Parameters_matrix>\n      <Parameters_expect>" << ev << "</Parameters_expect>\n"
		<< "      <Parameters_gap-open>" << gap_open << "</Parameters_gap-open>\n      <Parameters_gap-extend>" << gap_extend << "</Parameters_gap-extend>\n"
		<< "      <Parameters_filter>F</Parameters_filter>\n    </Parameters>\n  </BlastOutput_param>\n<BlastOutput_iterations>\n";
	return emit(o, buf, cap, "dmnd_format_xml_header");
}

extern "C" int64_t dmnd_format_xml_query_intro(const char* qtitle, int64_t qnum, int32_t qlen, char* buf, int64_t cap)
{
	if (!qtitle || !buf) return fail(DMND_E_ARG, "dmnd_format_xml_query_intro: NULL argument");
	Out o;
	o << "<Iteration>\n  <Iteration_iter-num>" << (long long)(qnum + 1) << "</Iteration_iter-num>\n  <Iteration_query-ID>Query_" << (long long)(qnum + 1)
		<< "</Iteration_query-ID>\n  <Iteration_query-def>";
	xml_title(o, qtitle, false, "");
	o << "</Iteration_query-def>\n  <Iteration_query-len>" << qlen << "</Iteration_query-len>\n<Iteration_hits>\n";
	return emit(o, buf, cap, "dmnd_format_xml_query_intro");
}

extern "C" int64_t dmnd_format_xml(const dmnd_hsp_view* v, int32_t hit_num, int32_t hsp_num, const int8_t* matrix8, char* buf, int64_t cap)
{
	if (!view_ok(v) || !matrix8 || !buf || hit_num < 0 || hsp_num < 0) return fail(DMND_E_ARG, "dmnd_format_xml: bad argument");
	if (!v->transcript) return fail(DMND_E_ARG, "dmnd_format_xml: the XML format needs the transcript");
	const dmnd_match& m = *v->match;
	const dmnd_hsp& h = m.hsp;
	Out o;
	if (hsp_num == 0) {
		if (hit_num > 0) o << "  </Hit_hsps>\n</Hit>\n";
		o << "<Hit>\n  <Hit_num>" << hit_num + 1 << "</Hit_num>\n";
		// Util::Seq::get_title_def: the id up to the first delimiter, the rest is the definition
		const size_t cut = std::strcspn(v->stitle, ID_DELIMITERS), total = std::strlen(v->stitle);
		const std::string id(v->stitle, cut), def = cut >= total ? std::string() : std::string(v->stitle + cut + 1);
		const uint32_t flags = g_format_flags.load();
		if (flags & DMND_FMT_XML_BLORD) {                      // --xml-blord-format, xml_format.cpp:43-48
			o << "  <Hit_id>gnl|BL_ORD_ID|" << (long long)v->snum << "</Hit_id>\n  <Hit_def>";
			xml_title(o, v->stitle, true, " &gt;");
		}
		else {
			o << "  <Hit_id>";
			xml_escaped(o, id.data(), id.size());
			o << "</Hit_id>\n  <Hit_def>";
			xml_title(o, def.c_str(), true, " &gt;");
		}
		o << "</Hit_def>\n  <Hit_accession>";
		const std::string acc = (flags & DMND_FMT_NO_PARSE_SEQIDS) ? id : accession_of(id);
		xml_escaped(o, acc.data(), acc.size());
		o << "</Hit_accession>\n  <Hit_len>" << v->slen << "</Hit_len>\n  <Hit_hsps>\n";
	}
	int sb, se;
	source_range(*v, sb, se);
	const Frame f(*v);
	o << "    <Hsp>\n      <Hsp_num>" << hsp_num + 1 << "</Hsp_num>\n      <Hsp_bit-score>" << m.bit_score << "</Hsp_bit-score>\n      <Hsp_score>" << h.score
		<< "</Hsp_score>\n      <Hsp_evalue>";
	o.print_e(m.evalue);
	o << "</Hsp_evalue>\n      <Hsp_query-from>" << sb + 1 << "</Hsp_query-from>\n      <Hsp_query-to>" << se << "</Hsp_query-to>\n      <Hsp_hit-from>" << h.s_begin + 1
		<< "</Hsp_hit-from>\n      <Hsp_hit-to>" << h.s_end << "</Hsp_hit-to>\n      <Hsp_query-frame>" << f.blast_frame() << "</Hsp_query-frame>\n"
		<< "      <Hsp_hit-frame>0</Hsp_hit-frame>\n      <Hsp_identity>" << h.identities << "</Hsp_identity>\n      <Hsp_positive>" << h.positives
		<< "</Hsp_positive>\n      <Hsp_gaps>" << h.gaps << "</Hsp_gaps>\n      <Hsp_align-len>" << h.length << "</Hsp_align-len>\n         <Hsp_qseq>";
	for (Walk w(*v); w.good(); w.next()) o << w.query_char();
	o << "</Hsp_qseq>\n         <Hsp_hseq>";
	for (Walk w(*v); w.good(); w.next()) o << w.subject_char();
	o << "</Hsp_hseq>\n      <Hsp_midline>";
	for (Walk w(*v); w.good(); w.next())
		o << (w.op == OP_MATCH ? AA[w.query()] : w.op == OP_SUBSTITUTION && !w.shift() ? (matrix8[w.query() * 32 + w.subject()] > 0 ? '+' : ' ') : ' ');
	o << "</Hsp_midline>\n    </Hsp>\n";
	return emit(o, buf, cap, "dmnd_format_xml");
}

extern "C" int64_t dmnd_format_xml_query_epilog(int unaligned, int64_t db_seqs, int64_t db_letters, double K, double lambda, char* buf, int64_t cap)
{
	if (!buf) return fail(DMND_E_ARG, "dmnd_format_xml_query_epilog: NULL argument");
	Out o;
	if (!unaligned) o << "  </Hit_hsps>\n</Hit>\n";
	o << "</Iteration_hits>\n  <Iteration_stat>\n    <Statistics>\n";
	if (db_seqs >= 0) o << "      <Statistics_db-num>" << (long long)db_seqs << "</Statistics_db-num>\n";
	if (db_letters >= 0) o << "      <Statistics_db-len>" << (long long)db_letters << "</Statistics_db-len>\n";
	o << "      <Statistics_hsp-len>0</Statistics_hsp-len>\n      <Statistics_eff-space>0</Statistics_eff-space>\n      <Statistics_kappa>";
	print_lf(o, K);
	o << "</Statistics_kappa>\n      <Statistics_lambda>";
	print_lf(o, lambda);
	o << "</Statistics_lambda>\n      <Statistics_entropy>0</Statistics_entropy>\n    </Statistics>\n  </Iteration_stat>\n</Iteration>\n";
	return emit(o, buf, cap, "dmnd_format_xml_query_epilog");
}

// ---- DAA (-f 100 / daa): src/legacy/daa/daa_write.cpp, daa_file.h ------------------------------------------------------------------
namespace {

struct Bin {
	std::string s;
	template<typename T> void put(T x) { s.append((const char*)&x, sizeof x); }
	void packed(unsigned x)            // TextBuffer::write_packed: the narrowest of 1 / 2 / 4 bytes
	{
		if (x <= 0xffu) put((uint8_t)x); else if (x <= 0xffffu) put((uint16_t)x); else put((uint32_t)x);
	}
};

unsigned length_flag(unsigned x) { return x <= 0xffu ? 0 : x <= 0xffffu ? 1 : 2; }       // get_length_flag, output/output.h:33-40

int64_t emit_bin(const Bin& b, char* buf, int64_t cap, const char* who)
{
	if ((int64_t)b.s.size() > cap) return fail(DMND_E_CAP, std::string(who) + ": output buffer too small");
	std::memcpy(buf, b.s.data(), b.s.size());
	return (int64_t)b.s.size();
}

}  // namespace

extern "C" int64_t dmnd_format_daa_header(const dmnd_daa_header* h, char* buf, int64_t cap)
{
	if (!h || !buf || !h->matrix || std::strlen(h->matrix) > 15 || (h->mode != 2 && h->mode != 3)) return fail(DMND_E_ARG, "dmnd_format_daa_header: bad argument");
	Bin b;
	b.put((uint64_t)0x3c0e53476d3ee36bULL); b.put((uint64_t)1);                         // DAA_header1: magic number, version
	b.put((uint64_t)h->build); b.put((uint64_t)h->db_seqs); b.put((uint64_t)h->db_seqs_used); b.put((uint64_t)h->db_letters);
	b.put((uint64_t)0); b.put((uint64_t)h->query_records);                               // flags, query_records
	b.put((int32_t)h->mode); b.put((int32_t)h->gap_open); b.put((int32_t)h->gap_extend);
	b.put((int32_t)0); b.put((int32_t)0); b.put((int32_t)0); b.put((int32_t)0); b.put((int32_t)0);    // reward, penalty, reserved1-3
	b.put(h->K); b.put(h->lambda); b.put(h->max_evalue); b.put((double)0);
	char name[16] = { 0 };
	for (size_t i = 0; h->matrix[i]; ++i) name[i] = (char)std::tolower((unsigned char)h->matrix[i]);
	b.s.append(name, 16);
	uint64_t block_size[256] = { 0 };
	char block_type[256] = { 0 };
	if (h->finished) {
		block_size[0] = (uint64_t)h->alignment_bytes; block_size[1] = (uint64_t)h->ref_name_bytes; block_size[2] = (uint64_t)h->db_seqs_used * 4;
		block_type[0] = 1; block_type[1] = 2; block_type[2] = 3;                          // alignments, ref_names, ref_lengths
	}
	b.s.append((const char*)block_size, sizeof block_size);
	b.s.append(block_type, sizeof block_type);
	return emit_bin(b, buf, cap, "dmnd_format_daa_header");
}

extern "C" int64_t dmnd_format_daa_query(const char* qtitle, const int8_t* seq, int32_t len, int dna, char* buf, int64_t cap)
{
	if (!qtitle || !seq || len < 0 || !buf) return fail(DMND_E_ARG, "dmnd_format_daa_query: bad argument");
	Bin b;
	b.put((uint32_t)0);                                       // record size: the caller fills it in when the query's matches are written
	b.put((uint32_t)len);
	b.s.append(qtitle, std::strcspn(qtitle, ID_DELIMITERS));
	b.s += '\0';
	// PackedSequence (basic/packed_sequence.h:34-105): 5 bits per amino acid; DNA 2 bits per base, 3 when the read has an N
	bool has_n = false;
	if (dna) for (int32_t i = 0; i < len; ++i) has_n |= seq[i] == 4;
	b.put((uint8_t)(has_n ? 1 : 0));
	const unsigned bits = dna ? (has_n ? 3 : 2) : 5;
	unsigned x = 0, n = 0;
	for (int32_t i = 0; i < len; ++i) {
		x |= (unsigned)(seq[i] & 31) << n;
		n += bits;
		if (n >= 8) { b.s += (char)(x & 0xff); n -= 8; x >>= 8; }
	}
	if (n > 0) b.s += (char)(x & 0xff);
	return emit_bin(b, buf, cap, "dmnd_format_daa_query");
}

extern "C" int64_t dmnd_format_daa_match(const dmnd_hsp_view* v, uint32_t dict_id, char* buf, int64_t cap)
{
	if (!view_ok(v) || !buf) return fail(DMND_E_ARG, "dmnd_format_daa_match: bad argument");
	if (!v->transcript) return fail(DMND_E_ARG, "dmnd_format_daa_match: the DAA format needs the transcript");
	const dmnd_hsp& h = v->match->hsp;
	int sb, se;
	source_range(*v, sb, se);
	const bool rev = v->match->frame > 2;
	const unsigned qb = (unsigned)(rev ? se - 1 : sb);                   // Hsp::oriented_range().begin_, basic/match.h:168-174
	Bin b;
	b.put((uint32_t)dict_id);
	b.put((uint8_t)(length_flag((unsigned)h.score) | (length_flag(qb) << 2) | (length_flag((unsigned)h.s_begin) << 4) | ((rev ? 1u : 0u) << 6)));
	b.packed((unsigned)h.score);
	b.packed(qb);
	b.packed((unsigned)h.s_begin);
	b.s.append((const char*)v->transcript, (size_t)h.transcript_len);
	b.s += '\0';                                                        // PackedOperation::terminator
	return emit_bin(b, buf, cap, "dmnd_format_daa_match");
}

// ---- reading a DAA record back (the `view` command): DAA_query_record::Match::read, legacy/daa/daa_record.cpp:52-83 -------------
extern "C" int dmnd_daa_match_read(const uint8_t* p, int64_t avail, int translated, int32_t source_len, uint32_t* dict_id, dmnd_match* m, int64_t* transcript_off,
	int64_t* used)
{
	if (!p || avail < 0 || !dict_id || !m || !transcript_off || !used) return fail(DMND_E_ARG, "dmnd_daa_match_read: NULL argument");
	int64_t o = 0;
	auto need = [&](int64_t n) { return o + n <= avail; };
	auto packed = [&](unsigned flag, uint32_t& x) -> bool {           // read_packed: 0 = one byte, 1 = two, 2 = four
		const int n = flag == 0 ? 1 : flag == 1 ? 2 : 4;
		if (flag > 2 || !need(n)) return false;
		x = 0;
		std::memcpy(&x, p + o, (size_t)n);
		o += n;
		return true;
	};
	if (!need(5)) return fail(DMND_E_ARG, "dmnd_daa_match_read: truncated record");
	m->read_begin = m->read_end = 0;
	std::memcpy(dict_id, p, 4);
	const uint8_t flag = p[4];
	o = 5;
	uint32_t score = 0, qb = 0, sb = 0;
	if (!packed(flag & 3, score) || !packed((flag >> 2) & 3, qb) || !packed((flag >> 4) & 3, sb)) return fail(DMND_E_ARG, "dmnd_daa_match_read: truncated record");
	*transcript_off = o;
	while (o < avail && p[o] != 0) ++o;                                 // PackedOperation::terminator
	if (o >= avail) return fail(DMND_E_ARG, "dmnd_daa_match_read: transcript without terminator");
	std::memset(m, 0, sizeof *m);
	m->hsp.transcript_len = (int32_t)(o - *transcript_off);
	m->hsp.transcript_off = *transcript_off;
	m->hsp.score = (int32_t)score;
	m->hsp.s_begin = (int32_t)sb;
	if (translated) {
		const bool rev = (flag & (1 << 6)) != 0;
		m->frame = rev ? 3 + (int32_t)(((uint32_t)source_len - 1 - qb) % 3) : (int32_t)(qb % 3);
		// Hsp::set_translated_query_begin, basic/match.h:176-183
		m->hsp.q_begin = rev ? (int32_t)(((uint32_t)source_len - 1 - (uint32_t)(m->frame - 3) - qb) / 3) : (int32_t)((qb - (uint32_t)m->frame) / 3);
	}
	else { m->frame = 0; m->hsp.q_begin = (int32_t)qb; }
	*used = o + 1;
	return DMND_OK;
}

// HspContext::parse (basic/hssp.cpp:48-105): ends, length, identities, mismatches, positives, gaps and gap openings (a run of
// insertions and deletions is one opening) from the transcript; e-value and bit score from the score.
static int hsp_from_transcript(const dmnd_params* params, const int8_t* const* qframes, const int32_t* qframe_len, int32_t source_len, int32_t evalue_qlen, int32_t slen,
	const uint8_t* transcript, dmnd_match* m)
{
	dmnd_hsp& h = m->hsp;
	h.length = h.identities = h.mismatches = h.positives = h.gap_openings = h.gaps = 0;
	const int off0 = qframes[1] ? m->frame % 3 : 0;         // one frame given: it is the alignment's
	int qpos = h.q_begin, spos = h.s_begin, run = 0, foff = off0;
	bool shifted = false;
	for (int32_t k = 0; k < h.transcript_len; ++k) {
		const uint8_t b = transcript[k];
		const int op = b >> 6;
		const int count = (op == OP_MATCH || op == OP_INSERTION) ? (b & 63) : 1;
		const int shift = op == OP_SUBSTITUTION ? ((b & 63) == FS_FORWARD ? 1 : (b & 63) == FS_REVERSE ? -1 : 0) : 0;
		if (shift && !qframes[1]) return fail(DMND_E_ARG, "dmnd_hsp_from_transcript: a frameshift alignment needs the three frames of its strand");
		for (int c = 0; c < count; ++c) {
			if (op != OP_DELETION && (qpos < 0 || qpos >= qframe_len[foff])) return fail(DMND_E_ARG, "Query sequence index out of bounds.");
			++h.length;
			if (shift) {                                          // counted as a column, nothing else (HspContext::parse, hssp.cpp:67-91)
				shifted = true;
				if (shift > 0) { if (++foff == 3) { foff = 0; ++qpos; } }
				else if (--foff < 0) { foff = 2; --qpos; }
				continue;
			}
			if (op == OP_MATCH) { ++h.identities; ++h.positives; run = 0; }
			else if (op == OP_SUBSTITUTION) {
				++h.mismatches;
				if (params->matrix8[(qframes[foff][qpos] & 31) * 32 + (b & 31)] > 0) ++h.positives;
				run = 0;
			}
			else { if (run == 0) ++h.gap_openings; ++run; ++h.gaps; }
			if (op != OP_DELETION) ++qpos;
			if (op != OP_INSERTION) ++spos;
		}
	}
	h.q_end = qpos; h.s_end = spos;
	if (shifted) {
		// TranslatedPosition::absolute_interval over the walk's first and last position (translated_position.h:129-135)
		const int b_in = off0 + 3 * h.q_begin, e_in = foff + 3 * qpos;
		if (m->frame < 3) { m->read_begin = b_in; m->read_end = e_in; }
		else { m->read_begin = source_len - e_in; m->read_end = source_len - b_in; }
	}
	m->evalue = dmnd_evalue_p(params, h.score, evalue_qlen, slen);
	m->bit_score = dmnd_bitscore_p(params, (double)h.score);
	return DMND_OK;
}

extern "C" int dmnd_hsp_from_transcript(const dmnd_params* params, const int8_t* qseq, int32_t qlen, int32_t evalue_qlen, int32_t slen, const uint8_t* transcript, dmnd_match* m)
{
	if (!params || !qseq || !transcript || !m || qlen < 0) return fail(DMND_E_ARG, "dmnd_hsp_from_transcript: bad argument");
	const int8_t* frames[3] = { qseq, nullptr, nullptr };
	const int32_t lens[3] = { qlen, 0, 0 };
	return hsp_from_transcript(params, frames, lens, 0, evalue_qlen, slen, transcript, m);
}

extern "C" int dmnd_hsp_from_transcript_frames(const dmnd_params* params, const int8_t* const qframes[3], const int32_t qframe_len[3], int32_t source_len, int32_t evalue_qlen,
	int32_t slen, const uint8_t* transcript, dmnd_match* m)
{
	if (!params || !qframes || !qframe_len || !qframes[0] || !qframes[1] || !qframes[2] || !transcript || !m || source_len < 0) return fail(DMND_E_ARG, "dmnd_hsp_from_transcript_frames: bad argument");
	return hsp_from_transcript(params, qframes, qframe_len, source_len, evalue_qlen, slen, transcript, m);
}

extern "C" int64_t dmnd_format_fields_unaligned(const char* qtitle, const int8_t* qseq, int32_t qlen, const int8_t* source_seq, int32_t source_len,
	const int32_t* ids, int n, char* buf, int64_t cap)
{
	if (!qtitle || !qseq || !ids || n < 1 || !buf) return fail(DMND_E_ARG, "dmnd_format_fields_unaligned: bad argument");
	Out o;
	for (int i = 0; i < n; ++i) {
		if (i) o << '\t';
		switch (ids[i]) {
		case DMND_F_QSEQID: o.until(qtitle, ID_DELIMITERS); break;
		case DMND_F_QLEN: o << (source_seq ? source_len : qlen); break;
		case DMND_F_QTITLE: o << qtitle; break;
		case DMND_F_FULL_QSEQ:
			if (source_seq) for (int k = 0; k < source_len; ++k) o << NT[source_seq[k] & 7];
			else for (int k = 0; k < qlen; ++k) o << AA[qseq[k] & 31];
			break;
		case DMND_F_QFRAME: o << '0'; break;
		case DMND_F_SSEQID: case DMND_F_SALLSEQID: case DMND_F_QSEQ: case DMND_F_SSEQ: case DMND_F_BTOP: case DMND_F_STITLE: case DMND_F_SALLTITLES:
		case DMND_F_FULL_SSEQ: case DMND_F_QSEQ_GAPPED: case DMND_F_SSEQ_GAPPED: case DMND_F_QSTRAND: case DMND_F_CIGAR: case DMND_F_QSEQ_TRANSLATED:
			o << '*'; break;
		case DMND_F_HSPNUM: return fail(DMND_E_ARG, "Invalid output field: hspnum");       // no handler for unaligned queries in the reference either
		default:
			if (ids[i] < 0 || ids[i] >= DMND_F_COUNT) return fail(DMND_E_ARG, "dmnd_format_fields_unaligned: unknown field id");
			o << "-1";
		}
	}
	o << '\n';
	return emit(o, buf, cap, "dmnd_format_fields_unaligned");
}

extern "C" int64_t dmnd_format_fields_header(const int32_t* ids, int n, char* buf, int64_t cap)
{
	if (!ids || n < 1 || !buf) return fail(DMND_E_ARG, "dmnd_format_fields_header: bad argument");
	Out o;
	for (int i = 0; i < n; ++i) {
		if (ids[i] < 0 || ids[i] >= DMND_F_COUNT) return fail(DMND_E_ARG, "dmnd_format_fields_header: unknown field id");
		if (i) o << '\t';
		o << FIELD_NAMES[ids[i]];
	}
	o << '\n';
	return emit(o, buf, cap, "dmnd_format_fields_header");
}
