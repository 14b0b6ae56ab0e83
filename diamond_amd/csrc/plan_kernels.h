// plan_kernels.h -- launch interface of the device planner of the extension stage (round 6): what load_hits + ungapped_stage +
// Chaining::run + add_dp_targets do per query on the reference's host threads
// (/root/reference/src/align/load_hits.h:44-127, align/ungapped.cpp:62-126, chaining/greedy_align.cpp:482-497,
// align/gapped_score.cpp:107-180), done for ALL seed hits of a block pair in a handful of launches: the hits are grouped by
// (query, target), every group's hits become diagonal segments (from the x-drop kernel's results), the segments are chained
// (chain_graph.h, the same source the host compiles) and the chains' bands merged into the DpTargets of round 1.
// The host keeps the ranking logic (which groups of a query are extended when) and only looks bands up.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/diamond_hip.h"
#include "xdrop_core.h"

namespace dmnd {

enum {
	PLAN_ON_HOST = 255,        // PlanGroup::n_bands: the group is planned by the host (too many hits / segments for a lane's arrays)
	PLAN_NEED_CHAIN = 254,     // between the two planning kernels: segments written, chaining still to do
	PLAN_MAX_HITS = 32,
	PLAN_MAX_SEGS = 16
};

struct PlanGroup {             // the seed hits of one (query, target) pair = SeedHitList entry of load_hits
	uint32_t target;           // block sequence id
	uint32_t hit_begin;        // first hit of the group in the call's hit list
	uint32_t n_hits;
	uint32_t band_begin;       // first of its n_bands entries in the band list
	uint16_t score;            // best stage-1 score of its hits (TargetScore::score; WorkTarget::ungapped_score for one context)
	uint8_t pass;              // some hit passed the gapped filter (1 everywhere when the filter is off)
	uint8_t n_bands;           // DpTargets of round 1, or PLAN_ON_HOST
};
struct PlanBand { int32_t d_begin, d_end; };
struct PlanQuery { uint32_t query, group_begin, hit_begin; };      // one per query that has hits, in hit order; one sentinel entry behind the last
struct PlanCounters { uint32_t n_groups, n_queries, n_bands, unsorted, n_on_host, n_chain, n_chain_big, pad; };      // n_chain / n_chain_big: groups with two to PLAN_SMALL_SEGS / more segments (plan_chain_list_kernel)

struct PlanArgs {
	const int8_t* qblock; const int8_t* tblock;
	const int64_t* qlimits; const int64_t* tlimits;
	int64_t n_targets;
	const int8_t* matrix;          // 32 x 32 int8
	const dmnd_seed_hit* hits; int64_t n_hits;
	const uint8_t* gf_flags;       // gapped filter flag per hit, or NULL (filter off)
	const XdropSeg* xd;            // x-drop extension of every hit (xdrop_seg_kernel)
	int gap_open, gap_extend, band_fast;
	int small_segs;                // groups with at most this many segments are chained with the small workspace (0: none; plan_kernels.hip)
	// work and output arrays, all in HBM, sized for n_hits entries (+ 1 where a sentinel follows)
	uint32_t* tgt;                 // target of every hit
	uint64_t* heads;               // group head flag | query head flag << 32, then their inclusive scan
	uint64_t* head_scan;
	PlanGroup* groups;
	PlanQuery* queries;
	int32_t* segs;                 // 4 ints per hit slot: the sorted segments of a multi-segment group, from its first hit slot on
	PlanBand* band_slots;          // bands of a group, from its first hit slot on
	uint32_t* chain_list; uint32_t chain_cap;      // the groups that need chaining (at most one per two hits): small ones from the front, the others from the back
	uint32_t* band_count;          // per group, then its exclusive scan in band_off
	uint32_t* band_off;
	PlanBand* bands;               // dense
	PlanCounters* counters;
	void** scan_tmp; size_t* scan_tmp_bytes;
};

hipError_t launch_plan(const PlanArgs& a, hipStream_t st);

}  // namespace dmnd
