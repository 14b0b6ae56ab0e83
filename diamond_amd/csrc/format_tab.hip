// format_tab.hip -- one BLAST tabular line of a match record (/root/reference/src/output/blast_tab_format.cpp; util/text_buffer.h:238-260).
// Split out of extend_host.hip in round 6; the other output formats are in format_api.hip.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include "ctx.h"

using namespace dmnd;

// BLAST tabular line of one match (qseqid sseqid pident length mismatch gapopen qstart qend sstart send evalue bitscore),
// formatted as the reference prints it (src/output/blast_tab_format.cpp; util/text_buffer.h:238-260).
namespace {

int format_tab_impl(const dmnd_match* m, const char* qseqid, const char* sseqid, int qstart, int qend, char* buf, int64_t cap)
{
	const dmnd_hsp& h = m->hsp;
	// Util::String::format_double (util/string/string.h:87-92): >= 100 -> floor, else one rounded decimal
	auto fd = [](double x, char* p, size_t n) {
		if (x >= 100.0) std::snprintf(p, n, "%lli", (long long)std::floor(x));
		else { const long long i = std::llround(x * 10.0); std::snprintf(p, n, "%lli.%lli", i / 10, i % 10); }
	};
	char pid[64], ev[64], bs[64];
	fd((double)h.identities * 100.0 / (double)h.length, pid, sizeof pid);     // Hsp::id_percent
	if (m->evalue == 0.0) std::snprintf(ev, sizeof ev, "0.0");                 // TextBuffer::print_e
	else std::snprintf(ev, sizeof ev, "%.2e", m->evalue);
	fd(m->bit_score, bs, sizeof bs);
	const int w = std::snprintf(buf, (size_t)cap, "%s\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%s\t%s\n", qseqid, sseqid, pid, h.length,
		h.mismatches, h.gap_openings, qstart, qend, h.s_begin + 1, h.s_end, ev, bs);
	return w < cap ? w : DMND_E_CAP;
}

}

extern "C" int dmnd_format_tab(const dmnd_match* m, const char* qseqid, const char* sseqid, char* buf, int64_t cap)
{
	if (!m || !qseqid || !sseqid || !buf) return fail(DMND_E_ARG, "dmnd_format_tab: NULL argument");
	return format_tab_impl(m, qseqid, sseqid, m->hsp.q_begin + 1, m->hsp.q_end, buf, cap);
}

// Hsp::oriented_query_range over query_source_range (basic/match.h:168-174; TranslatedPosition::absolute_interval,
// basic/translated_position.h:131-137): forward frame f reads DNA [f + 3 b, f + 3 e), reverse frames count from the 3' end
extern "C" int dmnd_format_tab_translated(const dmnd_match* m, const char* qseqid, const char* sseqid, int32_t source_len, char* buf, int64_t cap)
{
	if (!m || !qseqid || !sseqid || !buf) return fail(DMND_E_ARG, "dmnd_format_tab_translated: NULL argument");
	if (m->frame < 0 || m->frame > 5) return fail(DMND_E_ARG, "dmnd_format_tab_translated: frame out of range");
	const int b = m->hsp.q_begin, e = m->hsp.q_end;
	int qstart, qend;
	if (m->read_end > m->read_begin) {                         // frameshift alignment: Hsp::query_source_range, oriented by the strand
		if (m->frame < 3) { qstart = m->read_begin + 1; qend = m->read_end; }
		else { qstart = m->read_end; qend = m->read_begin + 1; }
	}
	else if (m->frame < 3) { qstart = m->frame + 3 * b + 1; qend = m->frame + 3 * e; }
	else { const int off = m->frame - 3; qstart = source_len - off - 3 * b; qend = source_len - off - 3 * e + 1; }
	return format_tab_impl(m, qseqid, sseqid, qstart, qend, buf, cap);
}

