// extend_device.h -- the device half of dmnd_extend as extend_host.hip calls it (extend_device.hip; round 6): the planner over the
// call's seed hits (plan_kernels.hip) and the extension of the planned queries in HBM (extend_kernels.hip).
#pragma once
#include <stdint.h>
#include <vector>
#include "ctx.h"
#include "plan_kernels.h"

namespace dmnd {

// what the two halves need of the call's configuration (extend_host.hip HostCfg)
struct DeviceCfg {
	int gap_open = 11, gap_extend = 1;
	int band_mode_fast = 1;              // Extension::Mode::BANDED_FAST
	int max_target_seqs = 25;
	int64_t ranking_chunk = 128;         // ranking_chunk_size (extend.cpp:79-92)
	int64_t max_swipe_dp = 1000000;
	bool use_cbs = true;                 // Hauser bias
	double max_evalue = 0.001;
};

// What the device planner hands over (page-locked host copies of its lists; valid until the context's next dmnd_extend)
struct DevPlan {
	const PlanGroup* groups = nullptr;
	const PlanQuery* queries = nullptr;       // n_queries + 1 entries
	const PlanBand* bands = nullptr;
	uint32_t n_groups = 0, n_queries = 0, n_bands = 0, n_on_host = 0;
	PlanArgs dev;                             // the same lists (and the planner's inputs) where they lie in HBM
};

// Runs the planner over the call's hits (in c->xd_hits, with their x-drop extensions in c->xd_out and -- gf_on -- their gapped
// filter flags in c->gf_flags), waits for it and leaves its lists in HBM. planned = false: the hits are not in
// (query, location, seed offset) order, the host has to plan.
int plan_on_device(dmnd_ctx* c, const DeviceCfg& h, int64_t n_hits, bool gf_on, DevPlan& plan, bool& planned);
// The planner's lists on the host (page-locked copies, valid until the context's next dmnd_extend): only the host path reads them
int plan_fetch_lists(dmnd_ctx* c, DevPlan& plan);
// The extension of the planned queries in HBM, from the planner's bands to the match records (extend_kernels.h). records: those
// queries' matches in output order with the HOST's e-value and bit score; qstate[k] (k = index into plan.queries): EXT_Q_DEVICE =
// done here, anything else = the host path has to extend the query. done = false: nothing was done here, every query goes to the host path.
int extend_on_device(dmnd_ctx* c, const DeviceCfg& h, const DevPlan& plan, int threads, std::vector<dmnd_match>& records, std::vector<uint8_t>& qstate, bool& done);

}  // namespace dmnd
