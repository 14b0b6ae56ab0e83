// swipe_kernels.h -- launch interface between the host API (api.hip) and swipe_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/diamond_hip.h"

namespace dmnd {

enum { WAVES_PER_BLOCK = 4 };

struct SwipeEnd {            // per item: best score, its end cell (row i, column j) and the carried statistics;
	                             // pad[0] = 1: the packed 16-bit sweep saturated, the item must be re-run in 32 bits
	int32_t score, end_i, end_j, stat_a, stat_b, pad[3];
};

// dmnd_dp_target::cbs_off <= -2: the item's own matrix and whether it is biased as well (diamond_hip.h)
__host__ __device__ inline int64_t own_matrix_number(int64_t cbs_off) { return (-2 - cbs_off) & (DMND_CBS_MATRIX_WITH_BIAS - 1); }
__host__ __device__ inline bool own_matrix_biased(int64_t cbs_off) { return cbs_off <= -2 && ((-2 - cbs_off) & DMND_CBS_MATRIX_WITH_BIAS) != 0; }

// kernel variants (launch_banded_swipe's kmode)
enum { K_SCORE = 0, K_COORDS = 1, K_TRACE = 2, K_STATS_FWD = 3, K_STATS_BWD_REV = 4 };

struct SwipeArgs {
	const int8_t* qblock;        // DMND_QUERY block letters (HBM)
	const int8_t* tblock;        // DMND_TARGET block letters (HBM)
	const int8_t* cbs;           // concatenated composition-bias vectors or nullptr
	const int8_t* matrix;        // 32x32 int8 (HBM; staged to LDS per wavefront)
	const int8_t* matrices;      // composition-adjusted matrices, 32x32 int8 each (HBM): an item with cbs_off <= -2 is scored with
	                             // number -2 - cbs_off instead of `matrix`, and without bias (dmnd_upload_matrices)
	const dmnd_dp_target* items; // all items of the call (HBM)
	const int32_t* order;        // slot -> item index, this launch's items (one P class)
	const int64_t* trace_off;    // slot -> byte offset into trace (TRACEBACK only)
	uint8_t* trace;
	SwipeEnd* ends;              // indexed by item
	int64_t n;                   // slots in this launch
	int32_t gap_open, gap_extend;
};

struct TracebackArgs {
	const int8_t* qblock;
	const int8_t* tblock;
	const int8_t* cbs;
	const int8_t* matrix;
	const int8_t* matrices;      // as in SwipeArgs
	const dmnd_dp_target* items;
	const int32_t* order;        // slot -> item index (all TRACEBACK slots of the chunk)
	const int32_t* p_of_slot;    // slot -> band class P of the item (its trace layout, swipe_core.h trace_byte_index)
	const int64_t* trace_off;    // slot -> trace byte offset
	const int64_t* transcript_off; // slot -> transcript byte offset, n+1 entries
	const uint8_t* trace;
	uint8_t* transcript;
	const SwipeEnd* ends;
	dmnd_hsp* hsps;              // indexed by item
	int32_t* status;             // min over items of the walk status (0 = ok)
	int64_t n;
	int32_t gap_open, gap_extend;
};

// packed-int16 sweep, two items per wavefront (swipe16_kernels.hip; eight for the row classes P = 3, 5): band classes P <= 5, at most 65535 pair-steps per item,
// items scored with the context's standard matrix only (one LDS table per workgroup; the host sends items with an adjusted matrix
// of their own to the 32-bit kernels)
struct Swipe16Args {
	const int8_t* qblock;
	const int8_t* tblock;
	const int8_t* cbs;
	const int8_t* matrix;
	const dmnd_dp_target* items;
	const int32_t* pairs;        // class_items_per_wave16(P) = 2 (8 for a row class) item indices per wavefront of this launch (one P class); -1: no item (only the first is always there)
	const int64_t* trace_off;    // item index -> byte offset into trace (traceback mode)
	uint8_t* trace;
	SwipeEnd* ends;              // indexed by item; score == 32767: saturated, to be re-run in the 32-bit kernel
	int64_t n_pairs;             // wavefronts
	int32_t gap_open, gap_extend;
	bool score_only = false;     // (without trace) only ends[].score and the saturation flag are wanted: no end cells
};

hipError_t launch_banded_swipe(int P, int mode, const SwipeArgs& a, hipStream_t stream);
hipError_t launch_banded_swipe16(int P, bool trace, const Swipe16Args& a, hipStream_t stream);
hipError_t launch_traceback(const TracebackArgs& a, hipStream_t stream);

}  // namespace dmnd
