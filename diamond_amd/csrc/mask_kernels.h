// mask_kernels.h -- launch interface of the tantan masking kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mask_core.h"

namespace dmnd {

struct TantanArgs {
	TantanParams p;
	int8_t* data;                 // block letters (HBM), masked in place
	const int64_t* limits;        // sequence i = data[limits[i], limits[i+1] - 1)
	int64_t n_seqs;
	const float* lr;              // 32x32 likelihood ratios (HBM)
	float* pb;                    // scratch: one float per block letter (indexed like data)
	float* scale;                 // scratch: limits[i] / 16 + i is the first slot of sequence i
	unsigned long long* n_masked; // out: number of letters masked
	const int32_t* ids;           // optional: only these sequences (block sequence ids), n_seqs of them; scratch_off[k] = where sequence
	const int64_t* scratch_off;   //   ids[k] keeps its floats in pb / scale (cumulative letters + 1 of the sequences before it)
	uint32_t* masked_pos;         // out (optional): block offsets of the letters that were overwritten, in no particular order
	unsigned long long* n_pos;    //   their number (may exceed pos_cap: then the list is incomplete)
	unsigned long long pos_cap;
};

hipError_t launch_tantan(const TantanArgs& a, hipStream_t st);

// Lane-per-sequence variant (round 4; mask_kernels.hip, tantan_lanes_kernel): the sequences are ordered by length on the device, a
// wavefront takes 64 neighbours of that order and every lane runs the whole recurrence of its own sequence -- the 50 repeat-offset
// states in registers, the sums in the lane -- so that no step crosses lanes. Work buffers are the caller's.
struct TantanLanesArgs {
	TantanArgs t;                 // p, data, limits, n_seqs, lr, n_masked, ids, masked_pos, n_pos, pos_cap as above; pb / scale / scratch_off unused
	uint32_t* keys[2];            // n_seqs entries each: sequence lengths, unsorted and sorted (descending)
	uint32_t* order;              // n_seqs entries: work index (into ids, or the sequence id itself) by descending length
	int64_t* wave_off;            // 2 (n_waves + 1) + 1 entries: where a wavefront keeps its floats inside `scratch`; behind them the wavefronts' sizes, then a counter
	float* scratch;               // scratch_floats entries
	int64_t scratch_floats;
	int64_t long_len;             // sequences longer than this are left out (their length key is 0): the caller gives them to launch_tantan
	void** sort_tmp;              // rocPRIM scratch of the context (grown here when too small)
	size_t* sort_tmp_bytes;
};
// floats of scratch that any order of the sequences needs: total_len = sum of the lengths, max_len = the longest
inline int64_t tantan_lanes_scratch(int64_t n_seqs, int64_t total_len, int64_t max_len) { return (total_len + 64 * max_len) / 16 * 17 + 128 * ((n_seqs + 63) / 64) + 4096; }
hipError_t prepare_tantan_lanes(const TantanLanesArgs& a, hipStream_t st);   // sizes rocPRIM's work space (launch_ does it too; this one runs no kernel)
hipError_t launch_tantan_lanes(const TantanLanesArgs& a, hipStream_t st);

// motif soft masking (mask_core.h): hit[p] = the 8-mer at block position p is in the sorted table; then per sequence the
// qualifying covered stretches of `soft` (a copy of the block) are overwritten with the mask letter
struct MotifArgs {
	const int8_t* data;           // block letters
	int8_t* soft;                 // copy of the block that receives the masks
	uint8_t* hit;                 // scratch, indexed like data
	const int64_t* limits;
	int64_t n_seqs, begin, end;   // letter range [limits[0], limits[n_seqs])
	const uint64_t* table;        // sorted 8-mer codes (HBM)
	int n_table;
	int max_range;
	unsigned long long* n_covered;
};
hipError_t launch_motif_mask(const MotifArgs& a, hipStream_t st);

}  // namespace dmnd
