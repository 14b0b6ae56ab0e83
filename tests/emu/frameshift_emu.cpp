// tests/emu/frameshift_emu.cpp -- TEST INFRASTRUCTURE ONLY. Runs the per-item code of the three-frame sweep kernels
// (diamond_amd/csrc/frameshift_core.h: what one lane of frameshift_kernels.hip executes) on the CPU, one item per call.
#include <cstring>
#include <vector>
#include "../../diamond_amd/csrc/frameshift_core.h"

using namespace dmnd;

static F3Item make_item(const int8_t* const* frames, const int32_t* lens, const int8_t* target, int tlen, int i0, int i1, int pos0)
{
	F3Item it;
	for (int f = 0; f < 3; ++f) { it.frame[f] = frames[f]; it.len[f] = lens[f]; }
	it.target = target; it.tlen = tlen; it.i0 = i0; it.i1 = i1; it.pos0 = pos0;
	return it;
}

extern "C" int emu_3frame_score(const int8_t* const* frames, const int32_t* lens, const int8_t* target, int tlen, int i0, int i1, int pos0,
	const int8_t* M, int gap_open, int gap_extend, int shift, int stride, int* max_col)
{
	const F3Item it = make_item(frames, lens, target, tlen, i0, i1, pos0);
	const int B = (i1 - i0 + 1) * 3;
	// stride > 1: the lane's entries lie `stride` apart as on the device; the gaps in between must stay untouched
	std::vector<int32_t> s((size_t)(B + 2) * stride, 0), g((size_t)(B + 3) * stride, 0);
	for (int k = 0; k < (int)s.size(); ++k) if (k % stride) s[k] = 0x5a5a5a5a;
	for (int k = 0; k < (int)g.size(); ++k) if (k % stride) g[k] = 0x5a5a5a5a;
	const int best = f3_sweep_score(it, F3Column{ s.data(), stride }, F3Column{ g.data(), stride }, M, F3Penalties{ gap_open + gap_extend, gap_extend, shift }, *max_col);
	for (int k = 0; k < (int)s.size(); ++k) if (k % stride && s[k] != 0x5a5a5a5a) return -1000;
	for (int k = 0; k < (int)g.size(); ++k) if (k % stride && g[k] != 0x5a5a5a5a) return -1000;
	return best;
}

// out[16]: score frame q_begin q_end s_begin s_end read_begin read_end length identities mismatches positives gap_openings gaps transcript_len status
extern "C" int emu_3frame_traceback(const int8_t* const* frames, const int32_t* lens, int strand, int dna_len, const int8_t* target, int tlen,
	int d_begin, int d_end, const int8_t* M, int gap_open, int gap_extend, int shift, int32_t* out, uint8_t* transcript, int cap)
{
	F3Item it = make_item(frames, lens, target, tlen, 0, 0, 0);
	f3_own_geometry(it, d_begin, d_end);
	const int B = (it.i1 - it.i0 + 1) * 3, cols = f3_trace_cols(it);
	std::vector<int32_t> T((size_t)(cols + 2) * (B + 1), 0), g((size_t)B + 3, 0);
	int max_col = 0;
	const int best = f3_sweep_trace(it, T.data(), F3Column{ g.data(), 1 }, M, F3Penalties{ gap_open + gap_extend, gap_extend, shift }, max_col);
	std::memset(out, 0, 16 * sizeof(int32_t));
	out[0] = best;
	if (best <= 0) return 0;
	const F3Walk w = f3_walk(it, T.data(), M, gap_open, gap_extend, shift, best, max_col, transcript, cap, dna_len);
	int rb = 0, re = 0;
	f3_read_range(w, strand, dna_len, rb, re);
	const int32_t v[16] = { best, strand * 3 + w.frame, w.q_begin, w.q_end, w.s_begin, w.s_end, rb, re, w.length, w.identities, w.mismatches, w.positives,
		w.gap_openings, w.gaps, w.transcript_len, w.status };
	std::memcpy(out, v, sizeof v);
	if (w.status == 0) std::memmove(transcript, transcript + cap - w.transcript_len, (size_t)w.transcript_len);
	return w.status;
}

extern "C" void emu_3frame_score_range(int strand, int dna_len, int qlen, int band, int i0, int pos0, int max_col, int32_t* out)
{
	f3_score_range(strand, dna_len, qlen, band, i0, pos0, max_col, out[0], out[1], out[2], out[3]);
}
