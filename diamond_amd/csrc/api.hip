// api.hip -- the C ABI of libdiamond_hip.so (include/diamond_hip.h): context, HBM residency of the
// sequence blocks, batching of DpTargets into wavefront launches, result collection.
//
// Host-side restatement of the binning/ordering work of the reference's swipe wrapper
// (/root/reference/src/dp/swipe/swipe_wrapper.cpp:317-362 swipe_bin, :446-470 swipe) re-thought for
// the GPU: instead of 8/16/32-bit score bins with overflow escalation, items are classed by band
// width (P = diagonals-per-lane class) and ordered longest-first so the 256 CUs drain evenly.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <time.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <numeric>
#include <unordered_map>
#include <string>
#include <vector>
#include "../../include/diamond_hip.h"
#include "swipe_core.h"
#include "swipe16_core.h"
#include "host_pool.h"
#include "swipe_kernels.h"
#include "ctx.h"
#include "score_matrices.h"

using namespace dmnd;

static thread_local std::string g_last_error;

int dmnd::fail(int code, const std::string& msg)
{
	g_last_error = msg;
	return code;
}

namespace {
std::mutex g_sync_mutex;
std::unordered_map<hipStream_t, hipEvent_t> g_sync_events;
}

static int sync_spin_us()
{
	// Poll for up to DMND_SYNC_SPIN_US microseconds (default 150) before the interrupt-driven wait: a short kernel's count is back
	// before a sleeping thread would have been woken; the CPU time this can burn is bounded per wait, unlike DMND_SPIN_SYNC
	return tuning().sync_spin_us;
}

hipError_t dmnd::wait_event(hipEvent_t ev)
{
	if (spin_sync()) return hipEventSynchronize(ev);
	// Poll, then sleep between polls: 0.7 CPU-ms per 50 ms of kernel (tools/probes/wait_probe.hip), as cheap as the runtime's
	// interrupt-driven wait (hipDeviceScheduleBlockingSync: 0.5 ms) and without its rare long stalls -- with the blocking wait one
	// step in ~100 of the C2 bench took 10 - 50 ms (a wait that slept through its completion until a timeout), with the sleeping
	// poll none. The sleeps grow from 20 to 200 us: a short kernel is picked up within microseconds, a long one costs a poll per 0.2 ms.
	const int spin_us = sync_spin_us();
	const auto t0 = std::chrono::steady_clock::now();
	long sleep_ns = 20000;
	for (;;) {
		const hipError_t q = hipEventQuery(ev);
		if (q == hipSuccess) return hipSuccess;
		if (q != hipErrorNotReady) return q;
		if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() < spin_us) continue;
		timespec ts{ 0, sleep_ns };
		nanosleep(&ts, nullptr);
		if (sleep_ns < 200000) sleep_ns += sleep_ns / 2;
	}
}

hipError_t dmnd::sync_stream(hipStream_t s)
{
	if (spin_sync()) return hipStreamSynchronize(s);
	hipEvent_t ev = nullptr;
	{
		// (looked up and created under the lock: the reference block's upload lane lets a second host thread wait on a context's
		// main stream, so "one host thread per stream" does not hold for the first wait)
		std::lock_guard<std::mutex> g(g_sync_mutex);
		auto it = g_sync_events.find(s);
		if (it != g_sync_events.end()) ev = it->second;
		else {
			const hipError_t e = hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming);
			if (e != hipSuccess) return e;
			g_sync_events[s] = ev;
		}
	}
	const hipError_t e = hipEventRecord(ev, s);
	if (e != hipSuccess) return e;
	return wait_event(ev);
}

void dmnd::forget_stream(hipStream_t s)
{
	hipEvent_t ev = nullptr;
	{
		std::lock_guard<std::mutex> g(g_sync_mutex);
		auto it = g_sync_events.find(s);
		if (it == g_sync_events.end()) return;
		ev = it->second;
		g_sync_events.erase(it);
	}
	(void)hipEventDestroy(ev);
}

int dmnd::PinBuf::ensure(size_t bytes)
{
	if (bytes <= cap) return DMND_OK;
	release();
	const size_t want = bytes + bytes / 2 + 4096;
	if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return fail(DMND_E_NOMEM, "hipHostMalloc of " + std::to_string(want) + " bytes failed"); }
	cap = want;
	return DMND_OK;
}

int dmnd::DevBuf::ensure(size_t bytes)
{
	if (bytes <= cap && own)
		return DMND_OK;
	release();                                         // an alias of another context's buffer (own == false) is dropped, not freed
	const size_t want = bytes + bytes / 4 + 256;
	static const bool trace = std::getenv("DMND_TRACE_ALLOC") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	if (hipMalloc(&p, want) != hipSuccess) {
		p = nullptr;
		return fail(DMND_E_NOMEM, "hipMalloc of " + std::to_string(want) + " bytes failed");
	}
	if (trace) std::fprintf(stderr, "hipMalloc %zu bytes: %.3f ms\n", want, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
	cap = want;
	return DMND_OK;
}

extern "C" int dmnd_abi_version(void) { return DMND_ABI_VERSION; }

extern "C" const char* dmnd_last_error(void) { return g_last_error.c_str(); }

extern "C" int dmnd_matrix_params(const char* name, int gap_open, int gap_extend, dmnd_params* p)
{
	if (!p || !name) return fail(DMND_E_ARG, "dmnd_matrix_params: NULL argument");
	std::string n(name);
	for (char& ch : n) ch = (char)std::tolower((unsigned char)ch);
	const StandardMatrixTable* m = nullptr;
	for (int i = 0; i < N_STANDARD_MATRICES; ++i)
		if (n == STANDARD_MATRICES[i].name) m = &STANDARD_MATRICES[i];
	if (!m) return fail(DMND_E_ARG, std::string("Unknown scoring matrix: ") + name);
	if (gap_open == -1) gap_open = m->default_gap_open;
	if (gap_extend == -1) gap_extend = m->default_gap_extend;
	const GumbelRow* g = nullptr;
	for (int i = 1; i < m->n_rows; ++i)
		if (m->rows[i].gap_open == gap_open && m->rows[i].gap_extend == gap_extend) g = &m->rows[i];
	if (!g) return fail(DMND_E_ARG, "Gap penalty settings are outside the supported range for this scoring matrix.");
	// Scores<int8_t>: the 26 letters of the table, everything else (delimiter, padding) the lowest score (score_matrix.h:35-50)
	for (int i = 0; i < 32; ++i)
		for (int j = 0; j < 32; ++j) {
			int8_t v = -128;
			if (i < 26 && j < 26) { const char ch = m->scores[i * 26 + j]; v = (int8_t)(ch >= 'a' ? ch - 'a' : -(ch - 'A' + 1)); }
			p->matrix8[i * 32 + j] = v;
		}
	p->gap_open = gap_open; p->gap_extend = gap_extend;
	p->lambda = g->lambda; p->K = g->K; p->alpha = g->alpha; p->alpha_v = g->alpha_v; p->sigma = g->sigma;
	p->u_alpha = m->rows[0].alpha; p->u_alpha_v = m->rows[0].alpha_v;
	return DMND_OK;
}

extern "C" int dmnd_default_params(dmnd_params* p)
{
	// BLOSUM62, gap open 11 / extend 1 (the reference's defaults, src/stats/score_matrix.cpp:51-52)
	if (!p) return fail(DMND_E_ARG, "params is NULL");
	p->db_letters = 0.0;
	p->max_evalue = 0.001;      // config.max_evalue default, src/basic/config.cpp
	return dmnd_matrix_params("blosum62", -1, -1, p);
}

extern "C" int dmnd_device_count(void)
{
	int count = 0, usable = 0;
	if (hipGetDeviceCount(&count) != hipSuccess) return 0;
	for (int d = 0; d < count; ++d) {
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++usable;
	}
	return usable;
}

namespace { __global__ void init_marker_kernel(int* p) { if (p) *p = 1; } }

// streams made ahead of their use by dmnd_init (a stream of the right priority is taken from here before a new one is created)
namespace {
struct PooledStream { int device; int priority; hipStream_t s; };
std::mutex g_stream_pool_mutex;
std::vector<PooledStream> g_stream_pool;

hipError_t take_stream(hipStream_t* s, int device, int priority)
{
	{
		std::lock_guard<std::mutex> g(g_stream_pool_mutex);
		for (size_t i = 0; i < g_stream_pool.size(); ++i)
			if (g_stream_pool[i].device == device && g_stream_pool[i].priority == priority) {
				*s = g_stream_pool[i].s;
				g_stream_pool.erase(g_stream_pool.begin() + (ptrdiff_t)i);
				return hipSuccess;
			}
	}
	return hipStreamCreateWithPriority(s, hipStreamNonBlocking, priority);
}
}

extern "C" hipError_t dmnd_touch_bias(hipStream_t), dmnd_touch_gapped(hipStream_t), dmnd_touch_mask(hipStream_t), dmnd_touch_seed(hipStream_t),
	dmnd_touch_swipe16(hipStream_t), dmnd_touch_swipe(hipStream_t), dmnd_touch_frameshift(hipStream_t), dmnd_touch_plan(hipStream_t), dmnd_touch_extend(hipStream_t);

// Host waits must not spin: measured on ROCm 7.2 (tools/probes/wait_probe.hip, round 6) hipEventSynchronize and hipStreamSynchronize
// burn 50.0 CPU-ms per 50 ms of kernel whatever the event's flags (hipEventBlockingSync alone changes nothing) -- the seed stage's
// thread spent 142 of C3's 239 host CPU-ms per step that way. Only the DEVICE flag hipDeviceScheduleBlockingSync makes the runtime
// sleep (0.5 CPU-ms per 50 ms), for every wait of the process; the library's own waits (sync_stream / wait_event) poll and sleep
// instead and do not depend on it. The flag is still set once per device, for the few runtime waits left (hipMemcpy of a parameter
// table, stream destruction) and for a host program's own synchronisations. DMND_SPIN_SYNC=1: spinning waits, no flag.
static void blocking_waits(int device)
{
	static std::mutex m;
	static std::vector<int> done;
	std::lock_guard<std::mutex> g(m);
	if (spin_sync() || std::find(done.begin(), done.end(), device) != done.end()) return;
	(void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
	(void)hipGetLastError();
	done.push_back(device);
}

extern "C" int dmnd_init(int device)
{
	const auto t0 = std::chrono::steady_clock::now();
	const bool trace = getenv("DMND_TRACE") != nullptr;
	auto lap = [&](const char* what) { if (trace) std::fprintf(stderr, "dmnd_init %8.2f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what); };
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return fail(DMND_E_DEVICE, "dmnd_init: no HIP device visible (this library has no CPU fallback)");
	lap("runtime up, devices counted");
	if (device < 0) device = 0;
	if (device >= count) return fail(DMND_E_ARG, "dmnd_init: device index out of range");
	HIP_TRY(hipSetDevice(device));
	blocking_waits(device);
	// The first launch of a kernel of a translation unit loads that unit's code object onto the device: all of them now, so that
	// no stage of the search pays for it later. The runtime loads two code objects from two threads at the same time (measured:
	// 18 + 5 ms one after the other, 16 ms together), and a stream costs 7.6 ms to create on these boxes -- so the two large units
	// (seed stage 18 ms, masking 8 ms) and the streams of the first context each get a thread of their own beside this one.
	hipError_t side_rc[5] = { hipSuccess, hipSuccess, hipSuccess, hipSuccess, hipSuccess };
	std::thread side[5];
	// (round 6) the planner's and the device extension's units carry rocPRIM's sort and scan kernels: 3.7 and 6.3 ms one after the other
	side[3] = std::thread([&] { side_rc[3] = hipSetDevice(device); if (side_rc[3] == hipSuccess) side_rc[3] = dmnd_touch_plan(nullptr); lap("plan"); });
	side[4] = std::thread([&] { side_rc[4] = hipSetDevice(device); if (side_rc[4] == hipSuccess) side_rc[4] = dmnd_touch_extend(nullptr); lap("extend"); });
	side[0] = std::thread([&] { side_rc[0] = hipSetDevice(device); if (side_rc[0] == hipSuccess) side_rc[0] = dmnd_touch_seed(nullptr); lap("seed"); });
	side[1] = std::thread([&] { side_rc[1] = hipSetDevice(device); if (side_rc[1] == hipSuccess) side_rc[1] = dmnd_touch_mask(nullptr); lap("mask"); });
	side[2] = std::thread([&] {
		side_rc[2] = hipSetDevice(device);
		int least = 0, greatest = 0;
		(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
		if (tuning().no_stream_priority) least = 0;
		for (int prio : { least, 0 }) {                   // dmnd_create's stream, and the reference block's upload lane
			hipStream_t s = nullptr;
			if (side_rc[2] == hipSuccess) side_rc[2] = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio);
			if (side_rc[2] == hipSuccess) { std::lock_guard<std::mutex> g(g_stream_pool_mutex); g_stream_pool.push_back(PooledStream{ device, prio, s }); }
		}
		lap("streams");
	});
	// (the first launch is not worth taking alone first: it then returns after the 20 ms of a one-kernel process instead of 42, but the
	// stream creation and the code objects behind it take 35 ms instead of overlapping with it -- 110 against 100 ms, tools/gpu_r06j.sh)
	hipLaunchKernelGGL(init_marker_kernel, dim3(1), dim3(1), 0, nullptr, (int*)nullptr);
	hipError_t rc = hipGetLastError();
	lap("first kernel (api)");
	struct { const char* name; hipError_t (*fn)(hipStream_t); } units[] = { { "bias", dmnd_touch_bias },
		{ "swipe16", dmnd_touch_swipe16 }, { "swipe", dmnd_touch_swipe }, { "gapped", dmnd_touch_gapped }, { "frameshift", dmnd_touch_frameshift } };
	for (auto& u : units) {
		if (rc == hipSuccess) rc = u.fn(nullptr);
		lap(u.name);
	}
	for (std::thread& t : side) t.join();
	for (hipError_t e : side_rc) if (rc == hipSuccess) rc = e;
	HIP_TRY(rc);
	HIP_TRY(hipDeviceSynchronize());
	lap("all code objects loaded");
	return DMND_OK;
}

extern "C" dmnd_ctx* dmnd_create(int device, const dmnd_params* params)
{
	if (!params) { fail(DMND_E_ARG, "dmnd_create: params is NULL"); return nullptr; }
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
		fail(DMND_E_DEVICE, "dmnd_create: no HIP device visible (this library has no CPU fallback)");
		return nullptr;
	}
	if (device < 0) {
		if (hipGetDevice(&device) != hipSuccess) { fail(DMND_E_DEVICE, "hipGetDevice failed"); return nullptr; }
	}
	if (device >= count) { fail(DMND_E_ARG, "dmnd_create: device index out of range"); return nullptr; }
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) { fail(DMND_E_DEVICE, "hipGetDeviceProperties failed"); return nullptr; }
	if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
		fail(DMND_E_DEVICE, std::string("dmnd_create: device is ") + prop.gcnArchName + ", kernels are built for gfx950 (MI355X) only");
		return nullptr;
	}
	if (hipSetDevice(device) != hipSuccess) { fail(DMND_E_DEVICE, "hipSetDevice failed"); return nullptr; }
	blocking_waits(device);
	dmnd_ctx* c = new dmnd_ctx();
	c->device = device;
	// host worker pool of the context's extension calls: contexts driven by different host threads (one per GPU) do not share one
	static std::atomic<int> n_created(0);
	c->pool_id = MAX_POOLS / 2 + n_created.fetch_add(1) % (MAX_POOLS / 2);
	c->params = *params;
	c->evaluer.init(*params);
	if (const char* mb = getenv("DMND_TRACE_ARENA_MB"))
		c->trace_arena_max = (size_t)std::max(8L, atol(mb)) << 20;
	// The context's own stream (seed stage, masking, uploads) runs at the lowest priority and the streams of the extension
	// stage's runners (aux_context) at the highest: when a driver overlaps the seed stage of the next batch with the extension
	// of the current one, the swipe kernels -- which sit on the extension's critical path between host phases -- get the CUs
	// first and the seed kernels fill the gaps.
	int prio_least = 0, prio_greatest = 0;
	(void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
	if (tuning().no_stream_priority) prio_least = prio_greatest = 0;
	if (take_stream(&c->stream, device, prio_least) != hipSuccess
		|| hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->ev2) != hipSuccess
		|| c->matrix.ensure(32 * 32) != DMND_OK
		|| hipMemcpy(c->matrix.p, params->matrix8, 32 * 32, hipMemcpyHostToDevice) != hipSuccess) {
		fail(DMND_E_DEVICE, "dmnd_create: stream/event/matrix setup failed");
		dmnd_destroy(c);
		return nullptr;
	}
	return c;
}

extern "C" void dmnd_destroy(dmnd_ctx* c)
{
	if (!c) return;
	if (c->sort_tmp) { (void)hipFree(c->sort_tmp); c->sort_tmp = nullptr; c->sort_tmp_bytes = 0; }
	if (c->pinned_cbs) (void)hipHostFree(c->pinned_cbs);
	for (DevBuf& kb : c->keep_trace) kb.release();
	c->stage_h.release(); c->ends_h.release(); c->stage_d.release();
	c->xd_hits.release(); c->xd_out.release(); c->xd_host.release();
	c->plan_dev.release(); c->plan_host.release();
	c->ext_dev.release(); c->ext_trace.release(); c->ext_host.release(); c->ext_ev.release();
	for (DevBuf& b : c->ext_trace_more) b.release();
	if (c->plan_tmp) { (void)hipFree(c->plan_tmp); c->plan_tmp = nullptr; c->plan_tmp_bytes = 0; }
	for (int i = 0; i < 2; ++i) { c->up_stage[i].release(); if (c->up_ev[i]) (void)hipEventDestroy(c->up_ev[i]); c->up_ev[i] = nullptr; }
	for (int i = 0; i < 2; ++i) { c->t_stage[i].release(); if (c->t_ev[i]) (void)hipEventDestroy(c->t_ev[i]); c->t_ev[i] = nullptr; }
	if (c->t_stream) { (void)hipStreamSynchronize(c->t_stream); forget_stream(c->t_stream); (void)hipStreamDestroy(c->t_stream); c->t_stream = nullptr; }
	delete c->kts; c->kts = nullptr;
	for (dmnd_ctx* a : c->aux) dmnd_destroy(a);
	c->aux.clear();
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	for (DevBuf* b : { &c->block[0], &c->block[1], &c->cbs, &c->matrix, &c->bias_ids, &c->items, &c->order, &c->p_of_slot, &c->trace_off,
		&c->transcript_off, &c->ends, &c->hsps, &c->trace, &c->transcript, &c->status, &c->pairs, &c->trace_off_item, &c->host_q, &c->host_t, &c->host_cbs,
		&c->d_limits[0], &c->d_limits[1], &c->qid_of, &c->mask_time, &c->seed_keys, &c->seed_next, &c->seed_qlist, &c->seed_qkeys, &c->seed_slot2, &c->seed_loc2, &c->seed_survivors, &c->seed_scored, &c->seed_need, &c->seed_qfold, &c->seed_tfold, &c->seed_tcodes, &c->seed_tflags, &c->seed_tplanes, &c->seed_tclass,
		&c->matched_slot, &c->matched_loc, &c->counters, &c->seed_hits, &c->seed_bitmap, &c->seed_deferred, &c->seed_eslot, &c->seed_eloc, &c->seed_hits_sorted, &c->sort_keys[0], &c->sort_keys[1], &c->sort_idx[0], &c->sort_idx[1], &c->gf_tables, &c->gf_hits, &c->gf_flags, &c->gf_scores, &c->gf_units, &c->alt_targets, &c->mask_lr, &c->mask_pb, &c->mask_scale, &c->mask_pos, &c->mask_ids, &c->mask_soff, &c->mask_long_ids, &c->mask_long_soff, &c->mask_long_pb, &c->mask_long_scale, &c->soft[0], &c->soft[1], &c->motif_hit, &c->motif_table, &c->adj_matrices, &c->join_keep, &c->join_pos, &c->join_in, &c->join_out, &c->join_recv })
		b->release();
	if (c->ev0) (void)hipEventDestroy(c->ev0);
	if (c->ev1) (void)hipEventDestroy(c->ev1);
	if (c->ev2) (void)hipEventDestroy(c->ev2);
	if (c->stream) { forget_stream(c->stream); (void)hipStreamDestroy(c->stream); }
	delete c;
}

extern "C" int dmnd_set_db_letters(dmnd_ctx* c, double db_letters)
{
	if (!c) return fail(DMND_E_ARG, "ctx is NULL");
	c->params.db_letters = db_letters;
	c->evaluer.db_letters = db_letters;
	return DMND_OK;
}

extern "C" void* dmnd_host_alloc(size_t bytes)
{
	void* p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); fail(DMND_E_NOMEM, "dmnd_host_alloc: hipHostMalloc of " + std::to_string(bytes) + " bytes failed"); return nullptr; }
	return p;
}

extern "C" void dmnd_host_free(void* p) { if (p) (void)hipHostFree(p); }

// Host -> HBM on the context's stream (asynchronous; the caller synchronises). A copy from pageable memory goes through the
// runtime's own small staging buffer and reached 1.8 GB/s here (163 ms for the 301 MB reference block of C2, BENCH_r02): the
// bytes are staged instead through TWO page-locked buffers of the context -- while the DMA engine moves one chunk the host
// fills the other -- so the transfer runs at the host's memcpy rate. A source that is page-locked already (dmnd_host_alloc,
// hipHostMalloc, hipHostRegister) is handed to the DMA engine as it is.
static int upload_bytes(dmnd_ctx* c, void* dst, const void* src, size_t bytes, bool target_lane = false)
{
	constexpr size_t CHUNK = (size_t)8 << 20;
	hipStream_t stream = target_lane ? c->t_stream : c->stream;
	dmnd::PinBuf* stage = target_lane ? c->t_stage : c->up_stage;
	hipEvent_t* ev = target_lane ? c->t_ev : c->up_ev;
	bool* busy = target_lane ? c->t_busy : c->up_busy;
	hipPointerAttribute_t attr;
	const bool pinned = hipPointerGetAttributes(&attr, src) == hipSuccess && (attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged);
	if (!pinned) (void)hipGetLastError();              // an unregistered pointer is an "invalid value", not a failure
	if (pinned || bytes <= ((size_t)256 << 10)) {
		HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, stream));
		return DMND_OK;
	}
	// A large pageable source is page-locked where it lies for the duration of the copy: registering 300 MB takes 5 ms on these
	// boxes (hipHostMalloc of as much: 46 ms), after which the DMA engine reads it at the PCIe rate -- 33 ms through the staging
	// chunks below (one host thread's memcpy rate) against 12 ms this way. The registration ends with the copy (synchronous here).
	if (bytes >= ((size_t)32 << 20)) {
		if (hipHostRegister(const_cast<void*>(src), bytes, hipHostRegisterDefault) == hipSuccess) {
			const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
			const hipError_t e2 = e == hipSuccess ? sync_stream(stream) : e;
			(void)hipHostUnregister(const_cast<void*>(src));
			HIP_TRY(e2);
			return DMND_OK;
		}
		(void)hipGetLastError();                         // e.g. a read-only mapping: staged below
	}
	for (int i = 0; i < 2; ++i) {
		if (int rc = stage[i].ensure(CHUNK)) return rc;
		if (!ev[i]) HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
	}
	size_t done = 0;
	for (int i = 0; done < bytes; i ^= 1) {
		const size_t n = std::min(CHUNK, bytes - done);
		if (busy[i]) HIP_TRY(wait_event(ev[i]));
		std::memcpy(stage[i].p, static_cast<const char*>(src) + done, n);
		HIP_TRY(hipMemcpyAsync(static_cast<char*>(dst) + done, stage[i].p, n, hipMemcpyHostToDevice, stream));
		HIP_TRY(hipEventRecord(ev[i], stream));
		busy[i] = true;
		done += n;
	}
	return DMND_OK;
}

// HBM -> host, synchronous, through the same page-locked chunks: the runtime's own path for a pageable destination took 10-30 ms
// for half a megabyte of seed hits right after large allocations or frees (round 3 timeline), a DMA into page-locked memory
// plus a memcpy does not
int dmnd::download_bytes(dmnd_ctx* c, void* dst, const void* src, size_t bytes)
{
	constexpr size_t CHUNK = (size_t)8 << 20;
	if (bytes == 0) return DMND_OK;
	for (int i = 0; i < 2; ++i) {
		if (int rc = c->up_stage[i].ensure(CHUNK)) return rc;
		if (!c->up_ev[i]) HIP_TRY(hipEventCreateWithFlags(&c->up_ev[i], hipEventDisableTiming));
		if (c->up_busy[i]) { HIP_TRY(wait_event(c->up_ev[i])); c->up_busy[i] = false; }
	}
	size_t issued = 0, copied = 0;
	int i = 0;
	// chunk k + 1 is in flight while chunk k is copied out of its staging buffer
	HIP_TRY(hipMemcpyAsync(c->up_stage[0].p, src, std::min(CHUNK, bytes), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipEventRecord(c->up_ev[0], c->stream));
	issued = std::min(CHUNK, bytes);
	while (copied < bytes) {
		const size_t n = std::min(CHUNK, bytes - copied);
		if (issued < bytes) {
			const size_t m = std::min(CHUNK, bytes - issued);
			HIP_TRY(hipMemcpyAsync(c->up_stage[i ^ 1].p, static_cast<const char*>(src) + issued, m, hipMemcpyDeviceToHost, c->stream));
			HIP_TRY(hipEventRecord(c->up_ev[i ^ 1], c->stream));
			issued += m;
		}
		HIP_TRY(wait_event(c->up_ev[i]));
		std::memcpy(static_cast<char*>(dst) + copied, c->up_stage[i].p, n);
		copied += n;
		i ^= 1;
	}
	return DMND_OK;
}

extern "C" int dmnd_share_block(dmnd_ctx* c, int which, const dmnd_ctx* src)
{
	if (!c || !src || c == src || (which != DMND_QUERY && which != DMND_TARGET) || !src->block[which].p || c->device != src->device)
		return fail(DMND_E_ARG, "dmnd_share_block: bad argument (the source context must hold the block, on the same device)");
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(sync_stream(c->stream));                   // nothing of this context still reads the block that is replaced
	for (DevBuf* b : { &c->block[which], &c->d_limits[which] }) b->release();
	c->block[which].p = src->block[which].p; c->block[which].cap = src->block[which].cap; c->block[which].own = false;
	c->d_limits[which].p = src->d_limits[which].p; c->d_limits[which].cap = src->d_limits[which].cap; c->d_limits[which].own = false;
	c->block_len[which] = src->block_len[which];
	c->limits[which] = src->limits[which];
	c->coarse[which] = src->coarse[which];
	c->soft_valid[which] = false;
	if (which == DMND_QUERY) { c->source_lens = src->source_lens; ++c->query_generation; }
	return DMND_OK;
}

extern "C" int dmnd_copy_block(dmnd_ctx* c, int which, const dmnd_ctx* src)
{
	if (!c || !src || c == src || (which != DMND_QUERY && which != DMND_TARGET) || !src->block[which].p || !c->block[which].p || c->device != src->device)
		return fail(DMND_E_ARG, "dmnd_copy_block: bad argument (both contexts must hold the block, on the same device)");
	if (!c->block[which].own) return fail(DMND_E_ARG, "dmnd_copy_block: the block is an alias (dmnd_share_block) and read-only");
	if (c->block_len[which] != src->block_len[which] || c->limits[which] != src->limits[which])
		return fail(DMND_E_ARG, "dmnd_copy_block: the two blocks differ in shape");
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(sync_stream(src->stream));                 // whatever wrote the source block has finished
	HIP_TRY(hipMemcpyAsync(c->block[which].p, src->block[which].p, (size_t)c->block_len[which], hipMemcpyDeviceToDevice, c->stream));
	c->soft_valid[which] = false;
	if (which == DMND_QUERY) ++c->query_generation;
	return DMND_OK;
}

extern "C" int dmnd_upload_block(dmnd_ctx* c, int which, const int8_t* data, int64_t data_len, const int64_t* limits, int64_t n_seqs)
{
	if (!c || (which != DMND_QUERY && which != DMND_TARGET) || !data || data_len <= 0 || n_seqs < 0)
		return fail(DMND_E_ARG, "dmnd_upload_block: bad argument");
	// the limits are validated before any state of the context changes: the device kernels and load_hits index with them
	if (limits) {
		if (n_seqs < 1) return fail(DMND_E_ARG, "dmnd_upload_block: limits given for an empty block");
		if (limits[0] < 0 || limits[n_seqs] > data_len) return fail(DMND_E_ARG, "dmnd_upload_block: limits outside [0, data_len]");
		for (int64_t i = 0; i < n_seqs; ++i)
			if (limits[i + 1] <= limits[i]) return fail(DMND_E_ARG, "dmnd_upload_block: limits must be strictly increasing (sequence " + std::to_string(i) + ")");
	}
	HIP_TRY(hipSetDevice(c->device));
	if (int rc = c->block[which].ensure((size_t)data_len + 64)) return rc;
	if (limits) if (int rc = c->d_limits[which].ensure((size_t)(n_seqs + 1) * sizeof(int64_t))) return rc;
	// the reference block travels on its own lane: nothing this call touches is shared with calls on the QUERY block of the same
	// context, so a driver may run it on a helper thread beside them (diamond-hip does, for the first block of a run)
	const bool lane = which == DMND_TARGET;
	if (lane && !c->t_stream) HIP_TRY(take_stream(&c->t_stream, c->device, 0));
	if (lane && c->block_len[which] > 0) HIP_TRY(sync_stream(c->stream));      // whatever still reads the block that is replaced
	if (int rc = upload_bytes(c, c->block[which].p, data, (size_t)data_len, lane)) return rc;
	if (limits) if (int rc = upload_bytes(c, c->d_limits[which].p, limits, (size_t)(n_seqs + 1) * sizeof(int64_t), lane)) return rc;
	HIP_TRY(sync_stream(lane ? c->t_stream : c->stream));
	c->block_len[which] = data_len;
	c->soft_valid[which] = false;
	if (which == DMND_QUERY) { c->source_lens.clear(); ++c->query_generation; }
	c->limits[which].clear();
	c->coarse[which].clear();
	if (limits) {
		c->limits[which].assign(limits, limits + n_seqs + 1);
		std::vector<uint32_t>& co = c->coarse[which];
		co.assign((size_t)(limits[n_seqs] >> dmnd_ctx::COARSE_SHIFT) + 2, 0);
		int64_t sidx = 0;
		for (size_t b = 0; b < co.size(); ++b) {
			const int64_t pos = (int64_t)b << dmnd_ctx::COARSE_SHIFT;
			while (sidx + 1 <= n_seqs && limits[sidx + 1] <= pos) ++sidx;
			co[b] = (uint32_t)std::min<int64_t>(sidx, n_seqs - 1);
		}
	}
	return DMND_OK;
}

extern "C" int dmnd_upload_cbs(dmnd_ctx* c, const int8_t* cbs, int64_t len)
{
	if (!c || len < 0 || (len > 0 && !cbs))
		return fail(DMND_E_ARG, "dmnd_upload_cbs: bad argument");
	HIP_TRY(hipSetDevice(c->device));
	c->cbs_len = len;
	if (len == 0)
		return DMND_OK;
	c->cbs_generation = ~(uint64_t)0;                  // the caller's own bias: not dmnd_extend's cached one
	if (int rc = c->cbs.ensure((size_t)len + 256)) return rc;      // slack: the gapped filter reads up to 130 bytes past a query (values unused)
	HIP_TRY(hipMemcpyAsync(c->cbs.p, cbs, (size_t)len, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(sync_stream(c->stream));
	return DMND_OK;
}

extern "C" int32_t dmnd_banded_cols(int32_t qlen, int32_t tlen, int32_t d_begin, int32_t d_end)
{
	// DpTarget::banded_cols, src/dp/dp.h:47-52
	const int32_t pos = std::max(d_end - 1, 0) - (d_end - 1);
	const int32_t j1 = std::min(qlen - 1 - d_begin, tlen - 1) + 1;
	return j1 - pos;
}

extern "C" double dmnd_evalue(const dmnd_ctx* c, int32_t raw_score, uint32_t query_len, uint32_t subject_len)
{
	return c ? c->evaluer.evalue(raw_score, query_len, subject_len) : 0.0;
}

extern "C" double dmnd_evalue_p(const dmnd_params* p, int32_t raw_score, uint32_t query_len, uint32_t subject_len)
{
	if (!p) return 0.0;
	Evaluer e;
	e.init(*p);
	return e.evalue(raw_score, query_len, subject_len);
}

extern "C" int dmnd_evalue_batch(const dmnd_params* p, const int32_t* raw_score, const int32_t* query_len, const int32_t* subject_len,
	int64_t n, double* out)
{
	if (!p || !raw_score || !query_len || !subject_len || !out || n < 0) return fail(DMND_E_ARG, "dmnd_evalue_batch: bad argument");
	Evaluer e;
	e.init(*p);
	for (int64_t i = 0; i < n; ++i)
		out[i] = e.evalue(raw_score[i], (unsigned)query_len[i], (unsigned)subject_len[i]);
	return DMND_OK;
}

extern "C" double dmnd_bitscore_p(const dmnd_params* p, double raw_score)
{
	if (!p) return 0.0;
	Evaluer e;
	e.init(*p);
	return e.bitscore(raw_score);
}

extern "C" double dmnd_bitscore(const dmnd_ctx* c, double raw_score)
{
	return c ? c->evaluer.bitscore(raw_score) : 0.0;
}

extern "C" int dmnd_last_kernel_ms(const dmnd_ctx* c, double* swipe_ms, double* traceback_ms)
{
	if (!c) return fail(DMND_E_ARG, "ctx is NULL");
	if (swipe_ms) *swipe_ms = c->swipe_ms;
	if (traceback_ms) *traceback_ms = c->traceback_ms;
	return DMND_OK;
}

namespace {

struct Bases {
	const int8_t* q; int64_t q_len;
	const int8_t* t; int64_t t_len;
	const int8_t* cbs; int64_t cbs_len;
	const int8_t* matrices = nullptr; int64_t n_matrices = 0;      // adjusted matrices of the work context (dmnd_upload_matrices)
};

// P: band class, + ADJ_CLASS for an item that is scored with an adjusted matrix of its own (a launch class of its own: those
// items go to the 32-bit kernels, which stage one matrix per wavefront)
enum { ADJ_CLASS = 1 << 20 };
struct Slot { int32_t item; int32_t P; int64_t steps; };
inline int band_p(const Slot& s) { return s.P & (ADJ_CLASS - 1); }
inline bool own_matrix(const Slot& s) { return (s.P & ADJ_CLASS) != 0; }

// DMND_SWIPE32=1: every sweep in the 32-bit kernels (A/B runs; the packed 16-bit kernels are the default for band classes
// up to SW16_MAX_P)
bool force_swipe32()
{
	static const bool v = [] { const char* e = std::getenv("DMND_SWIPE32"); return e && e[0] == '1'; }();
	return v;
}

}  // namespace

// The row classes pay when a launch fills the chip: a wavefront of eight items issues three to five times the instructions of a
// wavefront of two per anti-diagonal step, so a launch of a few thousand items (C2: 15 600 per batch, under one wavefront per SIMD
// in the row form; C5: 38 600 per call, measured 10 % slower in the row classes) is done sooner as many thin wavefronts, and a launch
// of 10^5 and more (C2skew, C3) sooner with 1.4 x fewer instructions. Items of one call (host path) / one ranking iteration (device half) from which on the row classes are used.
// DMND_SWEEP_ROWS (read per call: an A/B switch of the tests): 0 = never, 1 = always.
int64_t dmnd::sweep_rows_min_items()
{
	if (force_swipe32()) return INT64_MAX;
	const char* e = std::getenv("DMND_SWEEP_ROWS");
	if (e && e[0] == '0') return INT64_MAX;
	if (e && e[0] == '1') return 0;
	return tuning().sweep_rows_min_items;
}

namespace {

// launch class of an item: its band class, or -- for what the packed 16-bit kernels take (score / coordinates / traceback sweeps
// on the context's matrix, at most 65535 pair-steps) -- the row class its band fits (swipe_core.h band_class_rows)
inline int sweep_class(int band, int64_t steps, bool own, int kmode, bool rows)
{
	if (rows && !own && kmode <= K_TRACE && steps <= 2 * (int64_t)SW16_MAX_PAIRS) return band_class_rows(band);
	return band_class(band);
}

struct SweepLaunch { int64_t s0, s1; bool k16; int64_t pair_off; };      // pair_off: first entry of the launch in `pairs`

// One launch per band class of `slots` (grouped by class P ascending, longest first inside a class): the packed-int16 kernel
// with two items per wavefront -- eight for a row class -- (neighbours in launch order = similar lengths) when the class is eligible,
// else the 32-bit kernel with one item per wavefront. kmode: K_SCORE / K_COORDS / K_TRACE. pairs receives the item pairs of the 16-bit launches.
void plan_sweeps(const std::vector<Slot>& slots, int kmode, bool force32, std::vector<SweepLaunch>& launches, std::vector<int32_t>& pairs)
{
	const int64_t n = (int64_t)slots.size();
	launches.clear(); pairs.clear();
	for (int64_t s0 = 0; s0 < n;) {
		int64_t s1 = s0, max_steps = 0;
		while (s1 < n && slots[(size_t)s1].P == slots[(size_t)s0].P) { max_steps = std::max(max_steps, slots[(size_t)s1].steps); ++s1; }
		const bool k16 = !force32 && !force_swipe32() && kmode <= K_TRACE && !own_matrix(slots[(size_t)s0]) && sw16_class(slots[(size_t)s0].P) && max_steps <= 2 * (int64_t)SW16_MAX_PAIRS;
		launches.push_back(SweepLaunch{ s0, s1, k16, (int64_t)pairs.size() });
		if (k16) {
			const int64_t per = class_items_per_wave16(slots[(size_t)s0].P);
			for (int64_t s = s0; s < (s1 - s0 + per - 1) / per * per + s0; ++s)
				pairs.push_back(s < s1 ? slots[(size_t)s].item : -1);
		}
		s0 = s1;
	}
}

// order_dev / trace_off_slot_dev: slot-indexed (32-bit kernel); pairs_dev / trace_off_item_dev: for the 16-bit kernel
int issue_sweeps(dmnd_ctx* work, const Bases& b, const dmnd_dp_target* d_items, const std::vector<Slot>& slots, const std::vector<SweepLaunch>& launches,
	const int32_t* order_dev, const int64_t* trace_off_slot_dev, const int32_t* pairs_dev, const int64_t* trace_off_item_dev, uint8_t* trace_dev, int kmode)
{
	const bool trace = kmode == K_TRACE;
	for (const SweepLaunch& l : launches) {
		const int P = band_p(slots[(size_t)l.s0]);
		if (l.k16) {
			Swipe16Args a;
			a.qblock = b.q; a.tblock = b.t; a.cbs = b.cbs; a.matrix = work->matrix.as<int8_t>();
			a.items = d_items;
			a.pairs = pairs_dev + l.pair_off;
			a.trace_off = trace ? trace_off_item_dev : nullptr;
			a.trace = trace ? trace_dev : nullptr;
			a.ends = work->ends.as<SwipeEnd>();
			a.n_pairs = (l.s1 - l.s0 + class_items_per_wave16(P) - 1) / class_items_per_wave16(P);
			a.gap_open = work->params.gap_open; a.gap_extend = work->params.gap_extend;
			a.score_only = kmode == K_SCORE;
			HIP_TRY(launch_banded_swipe16(P, trace, a, work->stream));
		}
		else {
			SwipeArgs a;
			a.qblock = b.q; a.tblock = b.t; a.cbs = b.cbs; a.matrix = work->matrix.as<int8_t>(); a.matrices = b.matrices;
			a.items = d_items;
			a.order = order_dev + l.s0;
			a.trace_off = trace ? trace_off_slot_dev + l.s0 : nullptr;
			a.trace = trace ? trace_dev : nullptr;
			a.ends = work->ends.as<SwipeEnd>();
			a.n = l.s1 - l.s0;
			a.gap_open = work->params.gap_open; a.gap_extend = work->params.gap_extend;
			HIP_TRY(launch_banded_swipe(P, kmode, a, work->stream));
		}
	}
	return DMND_OK;
}

// plan + upload of the 16-bit kernel's arrays + issue (the general path; dmnd_swipe_keep packs its own single upload)
int launch_sweeps(dmnd_ctx* work, const Bases& b, const dmnd_dp_target* d_items, int64_t n_items_total, const std::vector<Slot>& slots,
	const int32_t* order_dev, const int64_t* trace_off_slot_dev, const std::vector<int64_t>* trace_off_slot, uint8_t* trace_dev, int kmode, bool force32)
{
	const int64_t n = (int64_t)slots.size();
	const bool trace = kmode == K_TRACE;
	std::vector<SweepLaunch> launches;
	std::vector<int32_t>& pairs = work->h_pairs;
	plan_sweeps(slots, kmode, force32, launches, pairs);
	if (!pairs.empty()) {
		if (int rc = work->pairs.ensure(pairs.size() * sizeof(int32_t))) return rc;
		HIP_TRY(hipMemcpyAsync(work->pairs.p, pairs.data(), pairs.size() * sizeof(int32_t), hipMemcpyHostToDevice, work->stream));
		if (trace) {
			// the 16-bit kernel addresses trace rows by item index
			std::vector<int64_t>& by_item = work->h_trace_off_item;
			by_item.assign((size_t)n_items_total, 0);
			for (int64_t s = 0; s < n; ++s) by_item[(size_t)slots[(size_t)s].item] = (*trace_off_slot)[(size_t)s];
			if (int rc = work->trace_off_item.ensure(by_item.size() * sizeof(int64_t))) return rc;
			HIP_TRY(hipMemcpyAsync(work->trace_off_item.p, by_item.data(), by_item.size() * sizeof(int64_t), hipMemcpyHostToDevice, work->stream));
		}
	}
	return issue_sweeps(work, b, d_items, slots, launches, order_dev, trace_off_slot_dev, work->pairs.as<int32_t>(), work->trace_off_item.as<int64_t>(), trace_dev, kmode);
}

// classes ascending, longest items first inside a class (load balance): a bucket sort on (P, steps / 16) is O(n) and enough --
// the exact order inside a bucket does not matter (results are written by item index)
void order_slots(std::vector<Slot>& v, std::vector<Slot>& tmp, std::vector<uint32_t>& count)
{
	if (v.size() < 2048) {
		std::sort(v.begin(), v.end(), [](const Slot& x, const Slot& y) { return x.P < y.P || (x.P == y.P && x.steps > y.steps); });
		return;
	}
	const int NB = 1024;
	count.assign((size_t)32 * NB + 1, 0);
	auto cls = [](const Slot& x) { return class_index(band_p(x)) + (own_matrix(x) ? 16 : 0); };      // P = 1, 2, 4, ... 512, the rows; then the same with own matrices
	auto bucket = [&](const Slot& x) { return (size_t)cls(x) * NB + (size_t)(NB - 1 - std::min<int64_t>(x.steps >> 4, NB - 1)); };
	for (const Slot& x : v) ++count[bucket(x) + 1];
	for (size_t i = 1; i < count.size(); ++i) count[i] += count[i - 1];
	tmp.resize(v.size());
	for (const Slot& x : v) tmp[count[bucket(x)]++] = x;
	v.swap(tmp);
}

// Runs one chunk of items [begin, end) (indices into `items`), all modes.
int run_chunk(dmnd_ctx* c, const Bases& b, const dmnd_dp_target* items, int64_t n_items_total, const dmnd_dp_target* d_items, const std::vector<Slot>& slots,
	int kmode, dmnd_hsp* out, std::vector<uint8_t>* chunk_transcripts, std::vector<int64_t>* chunk_tr_off, bool force32 = false)
{
	const int64_t n = (int64_t)slots.size();
	if (n == 0) return DMND_OK;
	const bool trace = kmode == K_TRACE;
	std::vector<int32_t> order(n), p_of(n);
	std::vector<int64_t> trace_off(n + 1, 0), tr_off(n + 1, 0);
	for (int64_t s = 0; s < n; ++s) {
		order[s] = slots[s].item;
		p_of[s] = band_p(slots[s]);
		if (trace) {
			const dmnd_dp_target& it = items[slots[s].item];
			const Geom g = make_geom(it.query_len, it.target_len, it.d_begin, it.d_end);
			trace_off[s + 1] = trace_off[s] + trace_bytes(g, band_p(slots[s]));
			tr_off[s + 1] = tr_off[s] + (int64_t)it.query_len + it.target_len + 2;
		}
	}
	if (int rc = c->order.ensure(n * sizeof(int32_t))) return rc;
	HIP_TRY(hipMemcpyAsync(c->order.p, order.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
	if (trace) {
		if (int rc = c->p_of_slot.ensure(n * sizeof(int32_t))) return rc;
		if (int rc = c->trace_off.ensure((n + 1) * sizeof(int64_t))) return rc;
		if (int rc = c->transcript_off.ensure((n + 1) * sizeof(int64_t))) return rc;
		if (int rc = c->trace.ensure((size_t)trace_off[n] + 64)) return rc;
		if (chunk_transcripts) if (int rc = c->transcript.ensure((size_t)tr_off[n] + 64)) return rc;
		if (int rc = c->status.ensure(sizeof(int32_t))) return rc;
		HIP_TRY(hipMemcpyAsync(c->p_of_slot.p, p_of.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(c->trace_off.p, trace_off.data(), (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(c->transcript_off.p, tr_off.data(), (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemsetAsync(c->status.p, 0, sizeof(int32_t), c->stream));
	}

	HIP_TRY(hipEventRecord(c->ev0, c->stream));
	// slots are grouped by P (ascending) by the caller: one launch per class
	if (int rc = launch_sweeps(c, b, d_items, n_items_total, slots, c->order.as<int32_t>(), trace ? c->trace_off.as<int64_t>() : nullptr, &trace_off,
		trace ? c->trace.as<uint8_t>() : nullptr, kmode, force32)) return rc;
	HIP_TRY(hipEventRecord(c->ev1, c->stream));
	if (trace) {
		TracebackArgs t;
		t.qblock = b.q; t.tblock = b.t; t.cbs = b.cbs; t.matrix = c->matrix.as<int8_t>(); t.matrices = b.matrices;
		t.items = d_items; t.order = c->order.as<int32_t>(); t.p_of_slot = c->p_of_slot.as<int32_t>();
		t.trace_off = c->trace_off.as<int64_t>(); t.transcript_off = c->transcript_off.as<int64_t>();
		t.trace = c->trace.as<uint8_t>(); t.transcript = chunk_transcripts ? c->transcript.as<uint8_t>() : nullptr;
		t.ends = c->ends.as<SwipeEnd>(); t.hsps = c->hsps.as<dmnd_hsp>(); t.status = c->status.as<int32_t>();
		t.n = n; t.gap_open = c->params.gap_open; t.gap_extend = c->params.gap_extend;
		HIP_TRY(launch_traceback(t, c->stream));
		HIP_TRY(hipEventRecord(c->ev2, c->stream));
		if (chunk_transcripts) {
			chunk_transcripts->resize((size_t)tr_off[n]);
			HIP_TRY(hipMemcpyAsync(chunk_transcripts->data(), c->transcript.p, (size_t)tr_off[n], hipMemcpyDeviceToHost, c->stream));
			*chunk_tr_off = tr_off;
		}
	}
	HIP_TRY(sync_stream(c->stream));
	float ms = 0.f;
	HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
	c->swipe_ms += ms;
	if (trace) {
		HIP_TRY(hipEventElapsedTime(&ms, c->ev1, c->ev2));
		c->traceback_ms += ms;
		int32_t st = 0;
		HIP_TRY(copy_now(c->stream, &st, c->status.p, sizeof(st), hipMemcpyDeviceToHost));
		if (st != 0)
			return fail(st, st == DMND_E_TRACEBACK ? "Traceback error." : "transcript slot too small");
	}
	(void)out;
	return DMND_OK;
}

int swipe_impl(dmnd_ctx* c, const Bases& b, const dmnd_dp_target* items, int64_t n, int mode, uint32_t hsp_values,
	dmnd_hsp* out, uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used)
{
	(void)hsp_values;
	if (transcript_used) *transcript_used = 0;
	if (n == 0) return DMND_OK;
	if (!items || !out || n < 0) return fail(DMND_E_ARG, "dmnd_banded_swipe: NULL items/out");
	if (n > 0x7fffffff) return fail(DMND_E_ARG, "dmnd_banded_swipe: more than 2^31-1 items in one call");
	int kmode;
	switch (mode) {
	case DMND_SWIPE_SCORE: kmode = K_SCORE; break;
	case DMND_SWIPE_COORDS: kmode = K_COORDS; break;
	case DMND_SWIPE_TRACEBACK: kmode = K_TRACE; break;
	case DMND_SWIPE_STATS: kmode = K_STATS_FWD; break;
	default: return fail(DMND_E_ARG, "dmnd_banded_swipe: unknown mode");
	}
	// TRACEBACK with transcript == NULL: coordinates and statistics from the traceback walk, transcripts not returned
	if (!b.q || !b.t) return fail(DMND_E_ARG, "dmnd_banded_swipe: sequence blocks not uploaded");
	HIP_TRY(hipSetDevice(c->device));
	auto wall = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double t_mark = wall();
	auto lap = [&](int slot) { const double t = wall(); c->host_ms[slot] += t - t_mark; t_mark = t; };

	std::vector<Slot> slots((size_t)n);
	const bool rows = n >= sweep_rows_min_items();
	// a saturated item goes to the 32-bit kernels in its power-of-two class
	auto for_32_bits = [&](Slot x) { const dmnd_dp_target& it = items[x.item]; x.P = band_class(it.d_end - it.d_begin) | (x.P & (int)ADJ_CLASS); return x; };
	std::atomic<int64_t> bad_item(-1), bad_band(-1);
	const int host_threads = n >= 4096 ? 8 : 1;
	const int64_t chunk = 2048, n_chunks = (n + chunk - 1) / chunk;
	parallel_for((size_t)n_chunks, host_threads, [&](size_t ci, int) {
		for (int64_t i = (int64_t)ci * chunk; i < std::min(n, ((int64_t)ci + 1) * chunk); ++i) {
			const dmnd_dp_target& it = items[i];
			const int band = it.d_end - it.d_begin;
			if (band <= 0 || it.query_len <= 0 || it.target_len <= 0 || it.query_off < 0 || it.target_off < 0
				|| it.query_off + it.query_len > b.q_len || it.target_off + it.target_len > b.t_len
				|| (it.cbs_off >= 0 && (!b.cbs || it.cbs_off + it.query_len > b.cbs_len))
				|| (it.cbs_off <= -2 && (own_matrix_number(it.cbs_off) >= b.n_matrices || (own_matrix_biased(it.cbs_off) && (!b.cbs || it.query_off + it.query_len > b.cbs_len))))) { bad_item.store(i); slots[i] = Slot{ (int32_t)i, 1, 0 }; continue; }
			// up to 32 (16 with statistics) one wavefront sweeps the item; wider bands take up to 16 wavefronts (swipe_kernels.hip)
			if (band_class(band) > 32 * 16 || (kmode == K_STATS_FWD && band_class(band) > 16 * 16)) { bad_band.store(i); slots[i] = Slot{ (int32_t)i, 1, 0 }; continue; }
			const Geom g = make_geom(it.query_len, it.target_len, it.d_begin, it.d_end);
			const int P = sweep_class(band, n_steps(g), it.cbs_off <= -2, kmode, rows);
			slots[i] = Slot{ (int32_t)i, P | (it.cbs_off <= -2 ? (int)ADJ_CLASS : 0), n_steps(g) };
		}
	});
	if (bad_item.load() >= 0) return fail(DMND_E_ARG, "dmnd_banded_swipe: item " + std::to_string(bad_item.load()) + " out of range");
	if (bad_band.load() >= 0)
		return fail(DMND_E_BAND, "Band size exceeds the supported maximum (" + std::to_string(kmode == K_STATS_FWD ? DMND_MAX_BAND / 2 : DMND_MAX_BAND) + ")");
	if (int rc = c->items.ensure(n * sizeof(dmnd_dp_target))) return rc;
	if (int rc = c->ends.ensure(n * sizeof(SwipeEnd))) return rc;
	if (kmode == K_TRACE) { if (int rc = c->hsps.ensure(n * sizeof(dmnd_hsp))) return rc; }
	HIP_TRY(hipMemcpyAsync(c->items.p, items, n * sizeof(dmnd_dp_target), hipMemcpyHostToDevice, c->stream));
	c->swipe_ms = c->traceback_ms = 0.0;

	std::vector<SwipeEnd> ends((size_t)n);
	std::vector<dmnd_hsp> hsps;
	auto by_class = [](const Slot& x, const Slot& y) { return x.P < y.P || (x.P == y.P && x.steps > y.steps); };
	std::vector<Slot> sort_tmp;
	std::vector<uint32_t> sort_count;
	auto order_slots = [&](std::vector<Slot>& v) { ::order_slots(v, sort_tmp, sort_count); };
	if (kmode != K_TRACE) {
		order_slots(slots);
		lap(0);
		if (int rc = run_chunk(c, b, items, n, c->items.as<dmnd_dp_target>(), slots, kmode, out, nullptr, nullptr)) return rc;
		lap(1);
		HIP_TRY(copy_now(c->stream, ends.data(), c->ends.p, n * sizeof(SwipeEnd), hipMemcpyDeviceToHost));
		{
			// items that saturated the 16-bit sweep (score >= 32767): once more in the 32-bit kernels, as the reference escalates
			// its score vectors (swipe_wrapper.cpp:317-360)
			std::vector<Slot> again;
			for (const Slot& x : slots) if (ends[(size_t)x.item].pad[0]) again.push_back(for_32_bits(x));
			if (!again.empty()) {
				std::sort(again.begin(), again.end(), by_class);
				if (int rc = run_chunk(c, b, items, n, c->items.as<dmnd_dp_target>(), again, kmode, out, nullptr, nullptr, true)) return rc;
				HIP_TRY(copy_now(c->stream, ends.data(), c->ends.p, n * sizeof(SwipeEnd), hipMemcpyDeviceToHost));
			}
		}
		for (int64_t i = 0; i < n; ++i) {
			dmnd_hsp h;
			std::memset(&h, 0, sizeof(h));
			h.score = ends[i].score;
			if (kmode != K_SCORE && h.score > 0) { h.q_end = ends[i].end_i + 1; h.s_end = ends[i].end_j + 1; }
			if (kmode == K_STATS_FWD && h.score > 0) { h.identities = ends[i].stat_a; h.length = ends[i].stat_b; }
			out[i] = h;
		}
		lap(2);
		if (kmode != K_STATS_FWD)
			return DMND_OK;
		// statistics without traceback: second, reversed pass over the target prefix [0, s_end)
		// (recompute_reversed, swipe_wrapper.cpp:364-444); only if a start coordinate or a backward statistic is wanted
		const uint32_t need_rev = DMND_HSP_QUERY_START | DMND_HSP_TARGET_START | DMND_HSP_MISMATCHES | DMND_HSP_GAP_OPENINGS;
		if (hsp_values != 0 && !(hsp_values & need_rev))
			return DMND_OK;
		std::vector<dmnd_dp_target> rev;
		std::vector<int64_t> src;
		for (int64_t i = 0; i < n; ++i) {
			if (out[i].score <= 0) continue;
			dmnd_dp_target r = items[i];
			int rt, rd0, rd1;
			reversed_band(r.query_len, out[i].s_end, r.d_begin, r.d_end, rt, rd0, rd1);
			r.target_len = rt; r.d_begin = rd0; r.d_end = rd1;
			rev.push_back(r);
			src.push_back(i);
		}
		const int64_t m = (int64_t)rev.size();
		if (m == 0) return DMND_OK;
		std::vector<Slot> rslots((size_t)m);
		for (int64_t k = 0; k < m; ++k) {
			const Geom g = make_geom(rev[k].query_len, rev[k].target_len, rev[k].d_begin, rev[k].d_end);
			rslots[k] = Slot{ (int32_t)k, band_class(rev[k].d_end - rev[k].d_begin) | (rev[k].cbs_off <= -2 ? (int)ADJ_CLASS : 0), n_steps(g) };
		}
		std::sort(rslots.begin(), rslots.end(), by_class);
		HIP_TRY(hipMemcpyAsync(c->items.p, rev.data(), m * sizeof(dmnd_dp_target), hipMemcpyHostToDevice, c->stream));
		const double ms_fwd = c->swipe_ms;
		if (int rc = run_chunk(c, b, rev.data(), m, c->items.as<dmnd_dp_target>(), rslots, K_STATS_BWD_REV, out, nullptr, nullptr)) return rc;
		(void)ms_fwd;
		HIP_TRY(copy_now(c->stream, ends.data(), c->ends.p, m * sizeof(SwipeEnd), hipMemcpyDeviceToHost));
		for (int64_t k = 0; k < m; ++k) {
			dmnd_hsp& h = out[src[k]];
			h.score = ends[k].score;                                          // the reversed pass' score is what the reference reports
			h.q_begin = rev[k].query_len - (ends[k].end_i + 1);              // banded_swipe.h:114-115
			h.s_begin = rev[k].target_len - (ends[k].end_j + 1);
			h.mismatches = ends[k].stat_a;
			h.gap_openings = ends[k].stat_b;
			h.gaps = h.length - h.identities - h.mismatches;                  // assign_stats, stat_cell.h:216-220
		}
		return DMND_OK;
	}

	// TRACEBACK: chunk by trace arena size, keep input order across chunks
	hsps.resize((size_t)n);
	int64_t used = 0;
	for (int64_t c0 = 0; c0 < n;) {
		int64_t c1 = c0;
		size_t bytes = 0;
		while (c1 < n) {
			const dmnd_dp_target& it = items[c1];
			const Geom g = make_geom(it.query_len, it.target_len, it.d_begin, it.d_end);
			const size_t need = (size_t)trace_bytes(g, band_p(slots[c1]));
			if (c1 > c0 && bytes + need > c->trace_arena_max) break;
			bytes += need;
			++c1;
		}
		std::vector<Slot> chunk(slots.begin() + c0, slots.begin() + c1);
		order_slots(chunk);
		std::vector<uint8_t> tr;
		std::vector<int64_t> tr_off;
		lap(0);
		if (int rc = run_chunk(c, b, items, n, c->items.as<dmnd_dp_target>(), chunk, K_TRACE, out, transcript ? &tr : nullptr, transcript ? &tr_off : nullptr)) return rc;
		lap(1);
		HIP_TRY(copy_now(c->stream, hsps.data() + c0, c->hsps.as<dmnd_hsp>() + c0, (size_t)(c1 - c0) * sizeof(dmnd_hsp), hipMemcpyDeviceToHost));
		// items that saturated the 16-bit sweep (the walk skipped them: transcript_len < 0): sweep + walk again in 32 bits
		std::vector<Slot> again;
		std::vector<uint8_t> tr2;
		std::vector<int64_t> tr_off2;
		for (const Slot& x : chunk) if (hsps[(size_t)x.item].transcript_len < 0) again.push_back(for_32_bits(x));
		if (!again.empty()) {
			std::sort(again.begin(), again.end(), by_class);
			if (int rc = run_chunk(c, b, items, n, c->items.as<dmnd_dp_target>(), again, K_TRACE, out, transcript ? &tr2 : nullptr, transcript ? &tr_off2 : nullptr, true)) return rc;
			for (const Slot& x : again)
				HIP_TRY(copy_now(c->stream, &hsps[(size_t)x.item], c->hsps.as<dmnd_hsp>() + x.item, sizeof(dmnd_hsp), hipMemcpyDeviceToHost));
		}
		if (!transcript) {
			for (int64_t i = c0; i < c1; ++i) { out[i] = hsps[i]; out[i].transcript_off = -1; }
			c0 = c1;
			continue;
		}
		// pack the transcripts tightly into the caller's arena, in input order
		std::vector<int64_t> slot_of((size_t)(c1 - c0)), slot2_of((size_t)(c1 - c0), -1);
		for (size_t s = 0; s < chunk.size(); ++s) slot_of[(size_t)(chunk[s].item - c0)] = (int64_t)s;
		for (size_t s = 0; s < again.size(); ++s) slot2_of[(size_t)(again[s].item - c0)] = (int64_t)s;
		for (int64_t i = c0; i < c1; ++i) {
			dmnd_hsp h = hsps[i];
			const int64_t s2 = slot2_of[(size_t)(i - c0)], s = slot_of[(size_t)(i - c0)];
			const uint8_t* src = s2 >= 0 ? tr2.data() + tr_off2[(size_t)s2] : tr.data() + tr_off[(size_t)s];
			const int64_t len = h.transcript_len + 1;
			if (used + len > transcript_cap)
				return fail(DMND_E_CAP, "dmnd_banded_swipe: transcript arena too small");
			std::memcpy(transcript + used, src, (size_t)len);
			h.transcript_off = used;
			used += len;
			out[i] = h;
		}
		c0 = c1;
	}
	if (transcript_used) *transcript_used = used;
	lap(2);
	return DMND_OK;
}

}  // namespace

// The sweeps (traceback mode, or scores only with trace_dev == NULL) of items that were prepared ON THE DEVICE (extend_kernels.hip): the launch order, trace offsets and
// item pairs are in HBM already, the host only knows how many items every launch class has (class c: P = class_of_index(c)) and the
// longest one's step count. One launch per class, as plan_sweeps / issue_sweeps do for a host-prepared list; the `pairs` entries of
// class c (class_items_per_wave16 per wavefront, the last wavefront filled up with -1) follow those of the classes before it.
int dmnd_sweep_classes(dmnd_ctx* work, const dmnd_ctx* c, const dmnd_dp_target* d_items, const uint32_t* class_count, const uint32_t* class_max_steps, int n_classes,
	const int32_t* order_dev, const int64_t* off_slot_dev, const int32_t* pairs_dev, const int64_t* off_item_dev, uint8_t* trace_dev, SwipeEnd* ends_dev)
{
	if (!c || !work) return fail(DMND_E_ARG, "ctx is NULL");
	const int8_t* cbs = c->cbs_len > 0 ? c->cbs.as<int8_t>() : nullptr;
	int64_t s0 = 0, pair0 = 0;
	for (int k = 0; k < n_classes; ++k) {
		const int64_t count = class_count[k];
		if (count == 0) continue;
		const int P = class_of_index(k);
		const int64_t per = class_items_per_wave16(P), waves = (count + per - 1) / per;
		const bool k16 = !force_swipe32() && sw16_class(P) && (int64_t)class_max_steps[k] <= 2 * (int64_t)SW16_MAX_PAIRS;
		if (row_class(P) && !k16) return fail(DMND_E_ARG, "dmnd_sweep_classes: a row class outside the 16-bit kernels");
		if (k16) {
			Swipe16Args a;
			a.qblock = c->block[DMND_QUERY].as<int8_t>(); a.tblock = c->block[DMND_TARGET].as<int8_t>(); a.cbs = cbs; a.matrix = c->matrix.as<int8_t>();
			a.items = d_items; a.pairs = pairs_dev + pair0; a.trace_off = trace_dev ? off_item_dev : nullptr; a.trace = trace_dev; a.ends = ends_dev;
			a.n_pairs = waves;
			a.gap_open = c->params.gap_open; a.gap_extend = c->params.gap_extend;
			a.score_only = trace_dev == nullptr;          // the device half reads end cells only behind traceback-mode sweeps
			HIP_TRY(launch_banded_swipe16(P, trace_dev != nullptr, a, work->stream));
		}
		else {
			SwipeArgs a;
			a.qblock = c->block[DMND_QUERY].as<int8_t>(); a.tblock = c->block[DMND_TARGET].as<int8_t>(); a.cbs = cbs; a.matrix = c->matrix.as<int8_t>(); a.matrices = nullptr;
			a.items = d_items; a.order = order_dev + s0; a.trace_off = trace_dev ? off_slot_dev + s0 : nullptr; a.trace = trace_dev; a.ends = ends_dev;
			a.n = count;
			a.gap_open = c->params.gap_open; a.gap_extend = c->params.gap_extend;
			HIP_TRY(launch_banded_swipe(P, trace_dev ? K_TRACE : K_SCORE, a, work->stream));
		}
		s0 += count; pair0 += waves * per;
	}
	return DMND_OK;
}

int dmnd_swipe_shared(dmnd_ctx* work, const dmnd_ctx* c, const dmnd_dp_target* items, int64_t n, int mode, uint32_t hsp_values,
	dmnd_hsp* out, uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used)
{
	if (!c || !work) return fail(DMND_E_ARG, "ctx is NULL");
	const Bases b{ c->block[DMND_QUERY].as<int8_t>(), c->block_len[DMND_QUERY], c->block[DMND_TARGET].as<int8_t>(), c->block_len[DMND_TARGET],
		c->cbs_len > 0 ? c->cbs.as<int8_t>() : nullptr, c->cbs_len, work->adj_matrices.as<int8_t>(), work->n_adj_matrices };
	if (work != c) {                                       // scoring state of the owner, by reference
		work->matrix.p = c->matrix.p; work->matrix.cap = c->matrix.cap; work->matrix.own = false;
		work->params = c->params; work->evaluer = c->evaluer;
	}
	return swipe_impl(work, b, items, n, mode, hsp_values, out, transcript, transcript_cap, transcript_used);
}

// The same with the target letters in a device buffer of the caller instead of the resident target block (items[].target_off
// counts from t): the masked target copies of the alternative-HSP rounds (--max-hsps, extend_host.hip)
int dmnd_swipe_targets(dmnd_ctx* work, const dmnd_ctx* c, const int8_t* t, int64_t t_len, const dmnd_dp_target* items, int64_t n, int mode, uint32_t hsp_values,
	dmnd_hsp* out, uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used)
{
	if (!c || !work || !t) return fail(DMND_E_ARG, "ctx is NULL");
	const Bases b{ c->block[DMND_QUERY].as<int8_t>(), c->block_len[DMND_QUERY], t, t_len, c->cbs_len > 0 ? c->cbs.as<int8_t>() : nullptr, c->cbs_len,
		work->adj_matrices.as<int8_t>(), work->n_adj_matrices };
	if (work != c) {
		work->matrix.p = c->matrix.p; work->matrix.cap = c->matrix.cap; work->matrix.own = false;
		work->params = c->params; work->evaluer = c->evaluer;
	}
	return swipe_impl(work, b, items, n, mode, hsp_values, out, transcript, transcript_cap, transcript_used);
}

int dmnd_swipe_keep(dmnd_ctx* work, const dmnd_ctx* c, const dmnd_dp_target* items, int64_t n, int arena, dmnd_hsp* out, KeptTrace& kt)
{
	kt.arena = arena; kt.kept = false;          // the vectors keep their capacity from call to call
	if (!c || !work) return fail(DMND_E_ARG, "ctx is NULL");
	if (n == 0) return DMND_OK;
	const Bases b{ c->block[DMND_QUERY].as<int8_t>(), c->block_len[DMND_QUERY], c->block[DMND_TARGET].as<int8_t>(), c->block_len[DMND_TARGET],
		c->cbs_len > 0 ? c->cbs.as<int8_t>() : nullptr, c->cbs_len, work->adj_matrices.as<int8_t>(), work->n_adj_matrices };
	auto wall = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t0 = wall();
	// One pass over the items: geometry, class, trace rows, range checks. Past the trace budget (or with an item the traceback
	// path cannot take) the plain score-only call does the job and reports the errors.
	int64_t total = 0;
	bool usable = arena >= 0 && arena < 16 && n <= 0x7fffffff, in_range = true;
	const bool rows = n >= sweep_rows_min_items();
	static thread_local std::vector<Slot> slots, sort_tmp;
	static thread_local std::vector<uint32_t> sort_count;
	static thread_local std::vector<int64_t> rows_of;
	slots.resize((size_t)n); rows_of.resize((size_t)n);
	for (int64_t i = 0; i < n && usable; ++i) {
		const dmnd_dp_target& it = items[i];
		const int band = it.d_end - it.d_begin;
		if (band <= 0 || it.query_len <= 0 || it.target_len <= 0 || band_class(band) > 32) { usable = false; break; }
		in_range &= !(it.query_off < 0 || it.target_off < 0 || it.query_off + it.query_len > b.q_len || it.target_off + it.target_len > b.t_len
			|| (it.cbs_off >= 0 && (!b.cbs || it.cbs_off + it.query_len > b.cbs_len)) || (it.cbs_off <= -2 && (own_matrix_number(it.cbs_off) >= b.n_matrices || (own_matrix_biased(it.cbs_off) && (!b.cbs || it.query_off + it.query_len > b.cbs_len)))));
		const Geom g = make_geom(it.query_len, it.target_len, it.d_begin, it.d_end);
		const int P = sweep_class(band, n_steps(g), it.cbs_off <= -2, K_TRACE, rows);
		slots[(size_t)i] = Slot{ (int32_t)i, P | (it.cbs_off <= -2 ? (int)ADJ_CLASS : 0), n_steps(g) };
		rows_of[(size_t)i] = trace_bytes(g, P);
		total += rows_of[(size_t)i];
	}
	if (!usable || (size_t)total > work->trace_arena_max)
		return dmnd_swipe_shared(work, c, items, n, DMND_SWIPE_SCORE, 0, out, nullptr, 0, nullptr);
	if (!in_range) return fail(DMND_E_ARG, "dmnd_swipe_keep: an item lies outside the uploaded blocks");
	if (work != c) {
		work->matrix.p = c->matrix.p; work->matrix.cap = c->matrix.cap; work->matrix.own = false;
		work->params = c->params; work->evaluer = c->evaluer;
	}
	HIP_TRY(hipSetDevice(work->device));
	order_slots(slots, sort_tmp, sort_count);
	std::vector<SweepLaunch> launches;
	plan_sweeps(slots, K_TRACE, false, launches, work->h_pairs);
	// every array the launches read, in ONE page-locked buffer and one copy:
	// [items n][order n][trace offset by slot n + 1][trace offset by item n][pairs]
	const size_t o_items = 0, o_order = o_items + (size_t)n * sizeof(dmnd_dp_target), o_off_slot = (o_order + (size_t)n * sizeof(int32_t) + 7) & ~(size_t)7,
		o_off_item = o_off_slot + ((size_t)n + 1) * sizeof(int64_t), o_pairs = o_off_item + (size_t)n * sizeof(int64_t),
		bytes = o_pairs + work->h_pairs.size() * sizeof(int32_t);
	if (int rc = work->stage_h.ensure(bytes)) return rc;
	if (int rc = work->stage_d.ensure(bytes)) return rc;
	char* hs = work->stage_h.as<char>();
	std::memcpy(hs + o_items, items, (size_t)n * sizeof(dmnd_dp_target));
	int32_t* order = reinterpret_cast<int32_t*>(hs + o_order);
	int64_t* off_slot = reinterpret_cast<int64_t*>(hs + o_off_slot);
	int64_t* off_item = reinterpret_cast<int64_t*>(hs + o_off_item);
	kt.trace_off.resize((size_t)n); kt.P.resize((size_t)n);
	off_slot[0] = 0;
	for (int64_t s = 0; s < n; ++s) {
		const int32_t item = slots[(size_t)s].item;
		order[s] = item;
		off_slot[s + 1] = off_slot[s] + rows_of[(size_t)item];
		off_item[item] = off_slot[s];
		kt.trace_off[(size_t)item] = off_slot[s];
		kt.P[(size_t)item] = band_p(slots[(size_t)s]);
	}
	if (!work->h_pairs.empty()) std::memcpy(hs + o_pairs, work->h_pairs.data(), work->h_pairs.size() * sizeof(int32_t));
	if ((int)work->keep_trace.size() <= arena) work->keep_trace.resize((size_t)arena + 1);
	DevBuf& tr = work->keep_trace[(size_t)arena];
	if (int rc = tr.ensure((size_t)total + 64)) return rc;
	if (int rc = work->ends.ensure(n * sizeof(SwipeEnd))) return rc;
	if (int rc = work->ends_h.ensure(n * sizeof(SwipeEnd))) return rc;
	HIP_TRY(hipMemcpyAsync(work->stage_d.p, hs, bytes, hipMemcpyHostToDevice, work->stream));
	const char* ds = work->stage_d.as<char>();
	const dmnd_dp_target* d_items = reinterpret_cast<const dmnd_dp_target*>(ds + o_items);
	work->host_ms[0] += wall() - t0;
	const double t1 = wall();
	HIP_TRY(hipEventRecord(work->ev0, work->stream));
	if (int rc = issue_sweeps(work, b, d_items, slots, launches, reinterpret_cast<const int32_t*>(ds + o_order), reinterpret_cast<const int64_t*>(ds + o_off_slot),
		reinterpret_cast<const int32_t*>(ds + o_pairs), reinterpret_cast<const int64_t*>(ds + o_off_item), tr.as<uint8_t>(), K_TRACE)) return rc;
	HIP_TRY(hipEventRecord(work->ev1, work->stream));
	SwipeEnd* ends = work->ends_h.as<SwipeEnd>();
	HIP_TRY(copy_now(work->stream, ends, work->ends.p, n * sizeof(SwipeEnd), hipMemcpyDeviceToHost));
	{
		// items that saturated the 16-bit sweep: once more in the 32-bit kernel, into the same trace rows
		std::vector<Slot> again;
		for (const Slot& x : slots) if (ends[x.item].pad[0]) again.push_back(x);
		// (an item of a row class has trace rows of that class, which the 32-bit kernels do not write: scores only then -- the
		// caller's round 2 sweeps its survivors again)
		for (const Slot& x : again)
			if (row_class(band_p(x))) return dmnd_swipe_shared(work, c, items, n, DMND_SWIPE_SCORE, 0, out, nullptr, 0, nullptr);
		if (!again.empty()) {
			std::vector<int32_t> order2(again.size());
			std::vector<int64_t> off2(again.size());
			for (size_t k = 0; k < again.size(); ++k) { order2[k] = again[k].item; off2[k] = kt.trace_off[(size_t)again[k].item]; }
			if (int rc = work->order.ensure(order2.size() * sizeof(int32_t))) return rc;
			if (int rc = work->trace_off.ensure(off2.size() * sizeof(int64_t))) return rc;
			HIP_TRY(hipMemcpyAsync(work->order.p, order2.data(), order2.size() * sizeof(int32_t), hipMemcpyHostToDevice, work->stream));
			HIP_TRY(hipMemcpyAsync(work->trace_off.p, off2.data(), off2.size() * sizeof(int64_t), hipMemcpyHostToDevice, work->stream));
			if (int rc = launch_sweeps(work, b, d_items, n, again, work->order.as<int32_t>(), work->trace_off.as<int64_t>(), &off2,
				tr.as<uint8_t>(), K_TRACE, true)) return rc;
			HIP_TRY(copy_now(work->stream, ends, work->ends.p, n * sizeof(SwipeEnd), hipMemcpyDeviceToHost));
		}
	}
	float ms = 0.f;
	HIP_TRY(hipEventElapsedTime(&ms, work->ev0, work->ev1));
	work->swipe_ms = ms; work->traceback_ms = 0.0;
	work->host_ms[1] += wall() - t1;
	kt.score.resize((size_t)n); kt.end_i.resize((size_t)n); kt.end_j.resize((size_t)n);
	for (int64_t i = 0; i < n; ++i) {
		dmnd_hsp h;
		std::memset(&h, 0, sizeof(h));
		h.score = ends[i].score;
		if (h.score > 0) { h.q_end = ends[i].end_i + 1; h.s_end = ends[i].end_j + 1; }
		h.transcript_off = -1;
		out[i] = h;
		kt.score[(size_t)i] = ends[i].score; kt.end_i[(size_t)i] = ends[i].end_i; kt.end_j[(size_t)i] = ends[i].end_j;
	}
	kt.kept = true;
	return DMND_OK;
}

int dmnd_traceback_kept(dmnd_ctx* work, const dmnd_ctx* c, const dmnd_dp_target* items, const KeptTrace& kt, const int64_t* src, int64_t n, dmnd_hsp* out)
{
	if (!c || !work || !kt.kept || kt.arena < 0 || kt.arena >= (int)work->keep_trace.size()) return fail(DMND_E_ARG, "dmnd_traceback_kept: no kept trace");
	if (n == 0) return DMND_OK;
	const Bases b{ c->block[DMND_QUERY].as<int8_t>(), c->block_len[DMND_QUERY], c->block[DMND_TARGET].as<int8_t>(), c->block_len[DMND_TARGET],
		c->cbs_len > 0 ? c->cbs.as<int8_t>() : nullptr, c->cbs_len, work->adj_matrices.as<int8_t>(), work->n_adj_matrices };
	HIP_TRY(hipSetDevice(work->device));
	// every array the walk reads, in one page-locked buffer and one copy:
	// [items n][end cells n][trace offsets n][transcript offsets n + 1 (all 0: no transcripts)][order n][band class n][status 1]
	const size_t o_items = 0, o_ends = o_items + (size_t)n * sizeof(dmnd_dp_target), o_off = o_ends + (size_t)n * sizeof(SwipeEnd),
		o_tr = o_off + (size_t)n * sizeof(int64_t), o_order = o_tr + ((size_t)n + 1) * sizeof(int64_t), o_p = o_order + (size_t)n * sizeof(int32_t),
		o_status = o_p + (size_t)n * sizeof(int32_t), bytes = o_status + sizeof(int32_t);
	if (int rc = work->stage_h.ensure(bytes)) return rc;
	if (int rc = work->stage_d.ensure(bytes)) return rc;
	if (int rc = work->hsps.ensure(n * sizeof(dmnd_hsp))) return rc;
	char* hs = work->stage_h.as<char>();
	std::memcpy(hs + o_items, items, (size_t)n * sizeof(dmnd_dp_target));
	SwipeEnd* ends = reinterpret_cast<SwipeEnd*>(hs + o_ends);
	int64_t* trace_off = reinterpret_cast<int64_t*>(hs + o_off);
	int32_t* order = reinterpret_cast<int32_t*>(hs + o_order);
	int32_t* p_of = reinterpret_cast<int32_t*>(hs + o_p);
	std::memset(hs + o_tr, 0, ((size_t)n + 1) * sizeof(int64_t));
	*reinterpret_cast<int32_t*>(hs + o_status) = 0;
	for (int64_t k = 0; k < n; ++k) {
		const int64_t x = src[k];
		if (x < 0 || x >= (int64_t)kt.P.size()) return fail(DMND_E_ARG, "dmnd_traceback_kept: item index out of range");
		order[k] = (int32_t)k; p_of[k] = kt.P[(size_t)x]; trace_off[k] = kt.trace_off[(size_t)x];
		SwipeEnd e;
		std::memset(&e, 0, sizeof(e));
		e.score = kt.score[(size_t)x]; e.end_i = kt.end_i[(size_t)x]; e.end_j = kt.end_j[(size_t)x];
		ends[k] = e;
	}
	HIP_TRY(hipMemcpyAsync(work->stage_d.p, hs, bytes, hipMemcpyHostToDevice, work->stream));
	char* ds = work->stage_d.as<char>();
	HIP_TRY(hipEventRecord(work->ev1, work->stream));
	TracebackArgs t;
	t.qblock = b.q; t.tblock = b.t; t.cbs = b.cbs; t.matrix = work->matrix.as<int8_t>(); t.matrices = b.matrices;
	t.items = reinterpret_cast<const dmnd_dp_target*>(ds + o_items); t.order = reinterpret_cast<const int32_t*>(ds + o_order);
	t.p_of_slot = reinterpret_cast<const int32_t*>(ds + o_p);
	t.trace_off = reinterpret_cast<const int64_t*>(ds + o_off); t.transcript_off = reinterpret_cast<const int64_t*>(ds + o_tr);
	t.trace = work->keep_trace[(size_t)kt.arena].as<uint8_t>(); t.transcript = nullptr;
	t.ends = reinterpret_cast<const SwipeEnd*>(ds + o_ends); t.hsps = work->hsps.as<dmnd_hsp>(); t.status = reinterpret_cast<int32_t*>(ds + o_status);
	t.n = n; t.gap_open = work->params.gap_open; t.gap_extend = work->params.gap_extend;
	HIP_TRY(launch_traceback(t, work->stream));
	HIP_TRY(hipEventRecord(work->ev2, work->stream));
	// results and status come back through the page-locked buffer, one wait for both
	if (int rc = work->ends_h.ensure((size_t)n * sizeof(dmnd_hsp) + sizeof(int32_t))) return rc;
	HIP_TRY(hipMemcpyAsync(work->ends_h.p, work->hsps.p, (size_t)n * sizeof(dmnd_hsp), hipMemcpyDeviceToHost, work->stream));
	HIP_TRY(hipMemcpyAsync(work->ends_h.as<char>() + (size_t)n * sizeof(dmnd_hsp), ds + o_status, sizeof(int32_t), hipMemcpyDeviceToHost, work->stream));
	HIP_TRY(sync_stream(work->stream));
	std::memcpy(out, work->ends_h.p, (size_t)n * sizeof(dmnd_hsp));
	float ms = 0.f;
	HIP_TRY(hipEventElapsedTime(&ms, work->ev1, work->ev2));
	work->swipe_ms = 0.0; work->traceback_ms = ms;
	const int32_t st = *reinterpret_cast<const int32_t*>(work->ends_h.as<char>() + (size_t)n * sizeof(dmnd_hsp));
	if (st != 0) return fail(st, st == DMND_E_TRACEBACK ? "Traceback error." : "transcript slot too small");
	for (int64_t k = 0; k < n; ++k) out[k].transcript_off = -1;
	return DMND_OK;
}

__global__ void touch_kernel() {}

extern "C" int dmnd_touch_streams(dmnd_ctx* c)
{
	if (!c) return fail(DMND_E_ARG, "ctx is NULL");
	HIP_TRY(hipSetDevice(c->device));
	// a marker kernel plus one small pageable copy in each direction: the runtime's staging buffers for pageable copies
	// are among the things it lets go of at a device-wide synchronize
	std::vector<dmnd_ctx*> all(1, c);
	all.insert(all.end(), c->aux.begin(), c->aux.end());
	std::vector<char> host((size_t)256 << 10, 0);
	for (dmnd_ctx* a : all) {
		if (int rc = a->items.ensure(host.size())) return rc;
		hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, a->stream);
		HIP_TRY(hipMemcpyAsync(a->items.p, host.data(), host.size(), hipMemcpyHostToDevice, a->stream));
		HIP_TRY(hipMemcpyAsync(host.data(), a->items.p, host.size(), hipMemcpyDeviceToHost, a->stream));
	}
	HIP_TRY(hipGetLastError());
	for (dmnd_ctx* a : all) HIP_TRY(sync_stream(a->stream));
	return DMND_OK;
}

static int aux_priority()
{
	int least = 0, greatest = 0;
	(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
	return tuning().no_stream_priority ? 0 : greatest;
}

dmnd_ctx* aux_context(dmnd_ctx* c, int k, int split)
{
	if (!c || k < 0) return nullptr;
	while ((int)c->aux.size() <= k) {
		dmnd_ctx* a = new dmnd_ctx();
		a->device = c->device;
		if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithPriority(&a->stream, hipStreamNonBlocking, aux_priority()) != hipSuccess || hipEventCreate(&a->ev0) != hipSuccess
			|| hipEventCreate(&a->ev1) != hipSuccess || hipEventCreate(&a->ev2) != hipSuccess) { delete a; return nullptr; }
		c->aux.push_back(a);
	}
	dmnd_ctx* a = c->aux[(size_t)k];
	a->trace_arena_max = std::max<size_t>(c->trace_arena_max / (size_t)std::max(split, 1), (size_t)256 << 20);
	return a;
}

// Replaces (n_keep == 0) or extends the context's adjusted matrices: the first n_keep stay, `n` are appended behind them.
int dmnd_append_matrices(dmnd_ctx* c, int64_t n_keep, const int8_t* matrices, int64_t n)
{
	if (n == 0 && n_keep == 0 && c) { c->n_adj_matrices = 0; return DMND_OK; }
	if (!c || n_keep < 0 || n < 0 || n_keep > c->n_adj_matrices || (n > 0 && !matrices)) return fail(DMND_E_ARG, "dmnd_upload_matrices: bad argument");
	HIP_TRY(hipSetDevice(c->device));
	const size_t M = 32 * 32, need = (size_t)(n_keep + n) * M;
	if (need > c->adj_matrices.cap) {                     // grow: the kept matrices move to the new buffer on the device
		dmnd::DevBuf bigger;
		if (int rc = bigger.ensure(std::max(need * 2, (size_t)64 << 10))) return rc;
		if (n_keep > 0) HIP_TRY(hipMemcpyAsync(bigger.p, c->adj_matrices.p, (size_t)n_keep * M, hipMemcpyDeviceToDevice, c->stream));
		HIP_TRY(sync_stream(c->stream));
		c->adj_matrices.release();
		c->adj_matrices = bigger;
	}
	if (n > 0) HIP_TRY(copy_now(c->stream, c->adj_matrices.as<int8_t>() + (size_t)n_keep * M, matrices, (size_t)n * M, hipMemcpyHostToDevice));
	c->n_adj_matrices = n_keep + n;
	return DMND_OK;
}

extern "C" int dmnd_upload_matrices(dmnd_ctx* c, const int8_t* matrices, int64_t n)
{
	if (!c) return fail(DMND_E_ARG, "ctx is NULL");
	return dmnd_append_matrices(c, 0, matrices, n);
}

extern "C" int dmnd_banded_swipe(dmnd_ctx* c, const dmnd_dp_target* items, int64_t n, int mode, uint32_t hsp_values,
	dmnd_hsp* out, uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used)
{
	return dmnd_swipe_shared(c, c, items, n, mode, hsp_values, out, transcript, transcript_cap, transcript_used);
}

extern "C" int dmnd_banded_swipe_host(dmnd_ctx* c, const int8_t* query, int32_t query_len, const int8_t* cbs,
	const dmnd_host_target* targets, int64_t n, int mode, uint32_t hsp_values,
	dmnd_hsp* out, uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used)
{
	if (!c) return fail(DMND_E_ARG, "ctx is NULL");
	if (transcript_used) *transcript_used = 0;
	if (n == 0) return DMND_OK;
	if (!query || query_len <= 0 || !targets || n < 0) return fail(DMND_E_ARG, "dmnd_banded_swipe_host: bad argument");
	HIP_TRY(hipSetDevice(c->device));
	std::vector<dmnd_dp_target> items((size_t)n);
	std::vector<int8_t> mats;
	int64_t total = 0;
	for (int64_t i = 0; i < n; ++i) {
		if (!targets[i].seq || targets[i].len <= 0) return fail(DMND_E_ARG, "dmnd_banded_swipe_host: empty target");
		items[i] = dmnd_dp_target{ 0, total, cbs ? 0 : -1, query_len, targets[i].len, targets[i].d_begin, targets[i].d_end };
		if (targets[i].matrix) {                          // the target's own matrix: 26 letter rows as the reference holds them, the rest never scores
			items[i].cbs_off = -2 - (int64_t)(mats.size() / (32 * 32));
			mats.resize(mats.size() + 32 * 32, (int8_t)-128);
			std::memcpy(mats.data() + mats.size() - 32 * 32, targets[i].matrix, 26 * 32);
		}
		total += targets[i].len;
	}
	if (int rc = dmnd_append_matrices(c, 0, mats.data(), (int64_t)(mats.size() / (32 * 32)))) return rc;
	std::vector<int8_t> tbuf((size_t)total);
	for (int64_t i = 0; i < n; ++i)
		std::memcpy(tbuf.data() + items[i].target_off, targets[i].seq, (size_t)targets[i].len);
	if (int rc = c->host_q.ensure((size_t)query_len + 64)) return rc;
	if (int rc = c->host_t.ensure((size_t)total + 64)) return rc;
	HIP_TRY(hipMemcpyAsync(c->host_q.p, query, (size_t)query_len, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipMemcpyAsync(c->host_t.p, tbuf.data(), (size_t)total, hipMemcpyHostToDevice, c->stream));
	if (cbs) {
		if (int rc = c->host_cbs.ensure((size_t)query_len + 64)) return rc;
		HIP_TRY(hipMemcpyAsync(c->host_cbs.p, cbs, (size_t)query_len, hipMemcpyHostToDevice, c->stream));
	}
	HIP_TRY(sync_stream(c->stream));
	const Bases b{ c->host_q.as<int8_t>(), query_len, c->host_t.as<int8_t>(), total, cbs ? c->host_cbs.as<int8_t>() : nullptr, cbs ? query_len : 0, c->adj_matrices.as<int8_t>(), c->n_adj_matrices };
	return swipe_impl(c, b, items.data(), n, mode, hsp_values, out, transcript, transcript_cap, transcript_used);
}
