// frameshift_kernels.h -- launch interface between frameshift_api.hip and frameshift_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/diamond_hip.h"

namespace dmnd {

// one lane's work item, resolved by the host: block offsets, lengths, the geometry of its sweep (frameshift_core.h, F3Item)
struct F3DevItem {
	int64_t frame_off[3];
	int64_t target_off;
	int32_t len[3];
	int32_t tlen;
	int32_t i0, i1, pos0;
	int32_t strand, dna_len;
	int32_t out;                 // index of the caller's item this result belongs to
	int64_t trace_off;           // traceback: offset (in int32 entries) of the item's kept columns inside the trace arena
	int64_t transcript_off;      // traceback: offset of its transcript slot
	int32_t transcript_cap, pad;
};

struct F3Result {
	int32_t score, max_col;
	int32_t frame, q_begin, q_end, s_begin, s_end, read_begin, read_end;
	int32_t length, identities, mismatches, positives, gap_openings, gaps, transcript_len, status, pad;
};

struct F3Args {
	const int8_t* qblock;
	const int8_t* tblock;
	const int8_t* matrix;        // 32 x 32 int8
	const F3DevItem* items;      // launch order: 64 consecutive items share a wavefront
	const int64_t* wave_off;     // per wavefront: offset (int32 entries) of its interleaved state columns inside `state`
	const int32_t* wave_rows;    // per wavefront: 3 * (its widest band)
	int32_t* state;              // zero-filled
	int32_t* trace;              // traceback: zero-filled arena of kept columns
	uint8_t* transcript;
	F3Result* results;           // by launch position
	int64_t n;
	int32_t gap_open, gap_extend, frame_shift;
};

hipError_t launch_frameshift_sweep(bool traceback, const F3Args& a, hipStream_t stream);

}  // namespace dmnd
