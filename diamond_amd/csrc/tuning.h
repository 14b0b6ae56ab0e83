// tuning.h -- the library's tuning values in ONE place (round 6; the round-5 review counted 51 getenv calls scattered over csrc/).
// None of them is part of the ABI; every one has a measured default and is read ONCE per process from the environment variable
// named beside it (the sweeps that chose the defaults: DESIGN.md 6.4, 6.6, 4.17; tools/stream_sweep.py, tools/seed_modes.py).
// Not here, on purpose: diagnostics (DMND_TRACE*, DMND_SEED_PHASES, DMND_CLI_TIMELINE), the tests' hooks that force a rare path
// per call (buffer caps, DMND_SEED_TILED / _FUSED / _CLASSES*, DMND_TRACE_ARENA_MB, DMND_SWIPE32 ...) and the A/B switches that keep
// the previous form of a stage alive as the second implementation parity tests compare with (DMND_EXTEND_DEVICE, _PLAN_GPU,
// _XDROP_GPU, _KEEP_TRACE): DESIGN.md 9 lists them all.
#pragma once
#include <algorithm>
#include <cstdlib>

namespace dmnd {

struct Tuning {
	// ---- host waits (api.hip sync_stream / wait_event)
	bool spin_sync = false;            // DMND_SPIN_SYNC=1: the runtime's spinning waits (lowest latency, a core per waiting thread)
	int sync_spin_us = 150;            // DMND_SYNC_SPIN_US: busy poll before the sleeping poll (a short kernel's count is back by then)
	// ---- host worker pool (host_pool.h)
	int pool_spins = 600;              // DMND_POOL_SPINS: pause iterations a worker spins for the next loop before it sleeps
	// ---- thread layout of the extension's HOST path (extend_host.hip; the device half has no host threads)
	int extend_team = 8;               // DMND_EXTEND_TEAM: fixed host slices per call
	int extend_split = 1;              // DMND_EXTEND_SPLIT: sub-batches with their own streams (the layout of rounds 2-3), 1 = one range
	int extend_runners = 1;            // DMND_EXTEND_RUNNERS
	int extend_sub_threads = 0;        // DMND_EXTEND_SUB_THREADS: 0 = by the hits per runner
	// ---- sweeps (api.hip sweep_rows_min_items)
	int sweep_rows_min_items = 1 << 17; // DMND_SWEEP_ROWS_MIN: items of a launch set from which on bands of <= 96 / 160 diagonals take the row classes
	int extend_resweep_below_pct = 40; // DMND_EXTEND_RESWEEP_BELOW_PCT: device half: an iteration of which at most this share of the targets can survive the culling is swept for scores only, its survivors again with traceback (0 = always keep trace rows)
	// ---- streams
	bool no_stream_priority = false;   // DMND_NO_STREAM_PRIORITY: every stream at the default priority
	// ---- seed stage geometry (seed_api.hip seed_sizes; 0 / -1 = the size-dependent default chosen there)
	int seed_slots_x8 = 0;             // DMND_SEED_SLOTS_X8: table slots per query position x 8 (8 .. 64; default 32 fused / 16)
	int seed_bitmap1_log2 = 0;         // DMND_SEED_BITMAP1_LOG2: level-1 filter bits (15 .. 27; default 24, 25 for short seeds by class)
	int seed_bm1_kb = 0;               // DMND_SEED_BM1_KB: level-1 filter size in KB, overrides the above
	int seed_bm1_k = 0;                // DMND_SEED_BM1_K: 3 = three bits per key in one word, else two
	int seed_stream_nt = -1;           // DMND_SEED_STREAM_NT: non-temporal loads of the streamed letters
	int seed_probe_policy = -1;        // DMND_SEED_PROBE_POLICY: cache-policy bits of the filter probes (stream_sweep.py)
	int seed_need_fold_log2 = 0;       // DMND_SEED_NEED_FOLD_LOG2: words of the LDS-folded need map of the deferred pass (11 .. 15; default 13)
};

inline const Tuning& tuning()
{
	static const Tuning t = [] {
		Tuning x;
		auto num = [](const char* name, int fallback) { const char* e = std::getenv(name); return e ? std::atoi(e) : fallback; };
		{ const char* e = std::getenv("DMND_SPIN_SYNC"); x.spin_sync = e && e[0] == '1'; }
		x.sync_spin_us = std::max(0, num("DMND_SYNC_SPIN_US", x.sync_spin_us));
		x.pool_spins = std::max(0, num("DMND_POOL_SPINS", x.pool_spins));
		x.extend_team = std::max(1, num("DMND_EXTEND_TEAM", x.extend_team));
		x.extend_split = std::max(1, std::min(64, num("DMND_EXTEND_SPLIT", x.extend_split)));
		x.extend_runners = std::max(1, num("DMND_EXTEND_RUNNERS", x.extend_runners));
		x.extend_sub_threads = std::max(0, num("DMND_EXTEND_SUB_THREADS", x.extend_sub_threads));
		x.sweep_rows_min_items = std::max(0, num("DMND_SWEEP_ROWS_MIN", x.sweep_rows_min_items));
		x.extend_resweep_below_pct = std::max(0, std::min(100, num("DMND_EXTEND_RESWEEP_BELOW_PCT", x.extend_resweep_below_pct)));
		x.no_stream_priority = std::getenv("DMND_NO_STREAM_PRIORITY") != nullptr;
		if (std::getenv("DMND_SEED_SLOTS_X8")) x.seed_slots_x8 = std::min(64, std::max(8, num("DMND_SEED_SLOTS_X8", 0)));
		if (std::getenv("DMND_SEED_BITMAP1_LOG2")) x.seed_bitmap1_log2 = std::min(27, std::max(15, num("DMND_SEED_BITMAP1_LOG2", 0)));
		if (std::getenv("DMND_SEED_BM1_KB")) x.seed_bm1_kb = std::min(65536, std::max(4, num("DMND_SEED_BM1_KB", 0)));
		x.seed_bm1_k = num("DMND_SEED_BM1_K", 0);
		x.seed_stream_nt = num("DMND_SEED_STREAM_NT", -1);
		x.seed_probe_policy = num("DMND_SEED_PROBE_POLICY", -1);
		x.seed_need_fold_log2 = num("DMND_SEED_NEED_FOLD_LOG2", 0);
		return x;
	}();
	return t;
}

}  // namespace dmnd
