// rank_join.hip -- the final top-k merge of a search that ran on several GPUs of one node, over RCCL (round 5; SURVEY.md 8(e):
// "query batches shard embarrassingly across the GPUs of one node with RCCL over xGMI only for the final top-k merge").
//
// One process drives N contexts, one per GPU (diamond-hip --gpus N; the multi-process form is diamond_amd/multigpu.py over
// torch.distributed). Every GPU has searched all queries against ITS reference blocks and holds match records in host memory, where
// dmnd_extend leaves them. dmnd_join_ranks turns them into what the reference's block join would print
// (/root/reference/src/output/join_blocks.cpp:129-256): GPU j becomes the owner of the query range [j Q/N, (j + 1) Q/N);
//   1. every GPU orders its records by owner (a counting sort on the host) and uploads them once;
//   2. ONE grouped exchange -- ncclGroupStart, an ncclSend / ncclRecv pair per (source, owner) that has records, ncclGroupEnd -- moves
//      each record to its owner over xGMI, device memory to device memory. The byte counts need no collective: one process knows them;
//   3. every owner merges its range on its own device (dmnd_join_blocks_device: radix sorts of a permutation, top-k per query);
//   4. the survivors come back to the host, owner after owner: that concatenation is in query order.
// RCCL is loaded at run time (dlopen "librccl.so"): a single-GPU run of the library never pays for it, and the library has no link-time
// dependency on it. Contexts that share ONE device (the test hook of a 1-GPU box: diamond-hip's DMND_CLI_SHARE_GPU, RCCL refuses two
// ranks on a device) exchange with device-to-device copies instead -- same partition, same merge; a group of one context still goes
// through RCCL (send and receive to itself inside the group), which is how the RCCL calls are exercised on a 1-GPU box.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
// The few RCCL declarations this file uses, stated here: the library is dlopen'ed (a ROCm install without RCCL still builds and
// runs -- dmnd_join_ranks then reports that the library is missing and the caller joins on the host). Values as in rccl.h.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint8 = 1 } ncclDataType_t;
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
const char* ncclGetErrorString(ncclResult_t result);
}
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "ctx.h"

using namespace dmnd;

extern "C" int dmnd_join_blocks_device(dmnd_ctx* c, const dmnd_match* records_dev, int64_t n, int max_target_seqs, double top_percent, uint32_t max_query,
	dmnd_match* out_dev, int64_t* n_out);

namespace {

struct Rccl {
	void* lib = nullptr;
	decltype(&ncclCommInitAll) comm_init_all = nullptr;
	decltype(&ncclCommDestroy) comm_destroy = nullptr;
	decltype(&ncclGroupStart) group_start = nullptr;
	decltype(&ncclGroupEnd) group_end = nullptr;
	decltype(&ncclSend) send = nullptr;
	decltype(&ncclRecv) recv = nullptr;
	decltype(&ncclGetErrorString) error_string = nullptr;
	std::string why;
};

Rccl& rccl()
{
	static Rccl r = [] {
		Rccl x;
		for (const char* name : { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" }) {
			x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
			if (x.lib) break;
		}
		if (!x.lib) { x.why = std::string("librccl.so cannot be loaded: ") + (dlerror() ? dlerror() : "?"); return x; }
		auto sym = [&](const char* s) { void* p = dlsym(x.lib, s); if (!p && x.why.empty()) x.why = std::string("librccl.so lacks ") + s; return p; };
		x.comm_init_all = reinterpret_cast<decltype(x.comm_init_all)>(sym("ncclCommInitAll"));
		x.comm_destroy = reinterpret_cast<decltype(x.comm_destroy)>(sym("ncclCommDestroy"));
		x.group_start = reinterpret_cast<decltype(x.group_start)>(sym("ncclGroupStart"));
		x.group_end = reinterpret_cast<decltype(x.group_end)>(sym("ncclGroupEnd"));
		x.send = reinterpret_cast<decltype(x.send)>(sym("ncclSend"));
		x.recv = reinterpret_cast<decltype(x.recv)>(sym("ncclRecv"));
		x.error_string = reinterpret_cast<decltype(x.error_string)>(sym("ncclGetErrorString"));
		if (!x.why.empty()) { dlclose(x.lib); x.lib = nullptr; }
		return x;
	}();
	return r;
}

// the communicators of a device list, made once per process (ncclCommInitAll costs tens of milliseconds)
struct CommSet { std::vector<int> devices; std::vector<ncclComm_t> comms; };
std::mutex g_comm_mutex;
std::vector<CommSet> g_comm_sets;

int comms_for(const std::vector<int>& devices, std::vector<ncclComm_t>& out)
{
	std::lock_guard<std::mutex> lock(g_comm_mutex);
	for (const CommSet& s : g_comm_sets) if (s.devices == devices) { out = s.comms; return DMND_OK; }
	Rccl& r = rccl();
	if (!r.lib) return fail(DMND_E_DEVICE, "dmnd_join_ranks: " + r.why);
	CommSet s;
	s.devices = devices;
	s.comms.resize(devices.size());
	const ncclResult_t e = r.comm_init_all(s.comms.data(), (int)devices.size(), devices.data());
	if (e != ncclSuccess) return fail(DMND_E_DEVICE, std::string("dmnd_join_ranks: ncclCommInitAll: ") + r.error_string(e));
	g_comm_sets.push_back(s);
	out = s.comms;
	return DMND_OK;
}

// owner of query q among n ranks over Q queries: rank g owns [g Q/n + min(g, Q%n) ...), the split of multigpu.shard_range
int owner_of(int64_t q, int64_t Q, int n)
{
	const int64_t base = Q / n, rem = Q % n, big = rem * (base + 1);
	if (q < big) return (int)(q / (base + 1));
	return base > 0 ? (int)(rem + (q - big) / base) : n - 1;
}

// The exchange of dmnd_join_ranks as plain arithmetic (no device; dmnd_join_ranks_plan exposes it to the CPU tests): source g holds
// counts[g] records; cnt[g][j] of them belong to owner j and are sent from offset send_off(g, j) = sum_{j' < j} cnt[g][j'] of g's
// owner-ordered copy to offset recv_off(j, g) = sum_{g' < g} cnt[g'][j] of owner j's receive buffer, which holds n_recv[j] records.
struct ExchangePlan {
	int n = 0;
	std::vector<int64_t> cnt, n_recv;                   // cnt[g * n + j]
	int64_t count(int g, int j) const { return cnt[(size_t)g * (size_t)n + (size_t)j]; }
	int64_t send_off(int g, int j) const { int64_t o = 0; for (int x = 0; x < j; ++x) o += count(g, x); return o; }
	int64_t recv_off(int j, int g) const { int64_t o = 0; for (int x = 0; x < g; ++x) o += count(x, j); return o; }
};

// query_of(g, i) = query of record i of source g; place(g, i, k): record i of source g is entry k of g's owner-ordered copy (stable:
// block and query order inside an owner's share survive). Returns the first record whose query lies outside [0, n_queries), or -1.
template<typename QueryOf, typename Place>
int64_t plan_exchange(int n, const int64_t* counts, int64_t n_queries, QueryOf query_of, Place place, ExchangePlan& p)
{
	p.n = n;
	p.cnt.assign((size_t)n * (size_t)n, 0);
	p.n_recv.assign((size_t)n, 0);
	for (int g = 0; g < n; ++g) {
		for (int64_t i = 0; i < counts[g]; ++i) {
			const int64_t q = query_of(g, i);
			if (q < 0 || q >= n_queries) return i;
			++p.cnt[(size_t)g * (size_t)n + (size_t)owner_of(q, n_queries, n)];
		}
		std::vector<int64_t> at((size_t)n, 0);
		for (int j = 1; j < n; ++j) at[(size_t)j] = at[(size_t)j - 1] + p.count(g, j - 1);
		for (int64_t i = 0; i < counts[g]; ++i) place(g, i, at[(size_t)owner_of(query_of(g, i), n_queries, n)]++);
	}
	for (int g = 0; g < n; ++g) for (int j = 0; j < n; ++j) p.n_recv[(size_t)j] += p.count(g, j);
	return -1;
}

}  // namespace

// transport_used (may be NULL): 1 = RCCL (ncclSend / ncclRecv), 2 = device-to-device copies (contexts sharing a device)
extern "C" int dmnd_join_ranks(dmnd_ctx* const* ctx, int n_ctx, const dmnd_match* const* records, const int64_t* counts, int64_t n_queries, int max_target_seqs,
	double top_percent, dmnd_match* out, int64_t cap, int64_t* n_out, int* transport_used)
{
	if (!ctx || n_ctx < 1 || n_ctx > 64 || !records || !counts || n_queries < 1 || max_target_seqs < 1 || top_percent > 100.0 || !n_out || cap < 0 || (cap > 0 && !out))
		return fail(DMND_E_ARG, "dmnd_join_ranks: bad argument");
	*n_out = 0;
	const int N = n_ctx;
	std::vector<int> devices((size_t)N);
	bool distinct = true;
	for (int g = 0; g < N; ++g) {
		if (!ctx[g] || counts[g] < 0 || (counts[g] > 0 && !records[g])) return fail(DMND_E_ARG, "dmnd_join_ranks: bad argument");
		devices[(size_t)g] = ctx[g]->device;
		for (int j = 0; j < g; ++j) distinct = distinct && devices[(size_t)j] != devices[(size_t)g] && ctx[j] != ctx[g];
	}
	for (int g = 0; g < N; ++g) for (int j = 0; j < g; ++j) if (ctx[j] == ctx[g]) return fail(DMND_E_ARG, "dmnd_join_ranks: the same context twice (its join buffers would be shared by two sources)");
	const bool use_rccl = distinct;                       // (one context: RCCL with itself)
	if (!use_rccl) for (int g = 1; g < N; ++g) if (devices[(size_t)g] != devices[0]) return fail(DMND_E_ARG, "dmnd_join_ranks: contexts either on distinct devices (RCCL) or all on one (copies)");
	if (transport_used) *transport_used = use_rccl ? 1 : 2;
	const size_t item = sizeof(dmnd_match);
	// 1. per source: records ordered by owner (stable: block and query order inside an owner's share survive), counts per owner
	ExchangePlan plan;
	std::vector<std::vector<dmnd_match>> sorted((size_t)N);
	for (int g = 0; g < N; ++g) sorted[(size_t)g].resize((size_t)counts[g]);
	if (plan_exchange(N, counts, n_queries, [&](int g, int64_t i) { return (int64_t)records[g][i].query; },
		[&](int g, int64_t i, int64_t k) { sorted[(size_t)g][(size_t)k] = records[g][i]; }, plan) >= 0)
		return fail(DMND_E_ARG, "dmnd_join_ranks: a record's query lies outside [0, n_queries)");
	const std::vector<int64_t>& n_recv = plan.n_recv;
	auto cnt_of = [&](int g, int j) { return plan.count(g, j); };
	// the page-locked-less `sorted` copies are the sources of asynchronous uploads: no exit before every stream has drained
	struct Drain { dmnd_ctx* const* ctx; int n; ~Drain() { for (int g = 0; g < n; ++g) if (ctx[g] && ctx[g]->stream) { (void)hipSetDevice(ctx[g]->device); (void)sync_stream(ctx[g]->stream); } } } drain{ ctx, N };
	// 2. upload, buffers
	for (int g = 0; g < N; ++g) {
		dmnd_ctx* c = ctx[g];
		HIP_TRY(hipSetDevice(c->device));
		if (int rc = c->join_in.ensure(std::max<size_t>((size_t)counts[g], 1) * item)) return rc;
		if (int rc = c->join_recv.ensure(std::max<size_t>((size_t)n_recv[(size_t)g], 1) * item)) return rc;
		if (int rc = c->join_out.ensure(std::max<size_t>((size_t)n_recv[(size_t)g], 1) * item)) return rc;
		if (counts[g] > 0) HIP_TRY(hipMemcpyAsync(c->join_in.p, sorted[(size_t)g].data(), (size_t)counts[g] * item, hipMemcpyHostToDevice, c->stream));
	}
	// 3. the exchange: source g's share for owner j lies at send offset S(g, j) = sum_{j' < j} cnt[g][j'], and lands at receive offset
	// R(j, g) = sum_{g' < g} cnt[g'][j] of owner j
	auto send_off = [&](int g, int j) { return plan.send_off(g, j); };
	auto recv_off = [&](int j, int g) { return plan.recv_off(j, g); };
	if (use_rccl) {
		std::vector<ncclComm_t> comms;
		if (int rc = comms_for(devices, comms)) return rc;
		Rccl& r = rccl();
		ncclResult_t e = r.group_start();
		for (int g = 0; g < N && e == ncclSuccess; ++g) {
			dmnd_ctx* c = ctx[g];
			for (int j = 0; j < N && e == ncclSuccess; ++j) {
				if (cnt_of(g, j) > 0)
					e = r.send(c->join_in.as<char>() + (size_t)send_off(g, j) * item, (size_t)cnt_of(g, j) * item, ncclUint8, j, comms[(size_t)g], c->stream);
				if (e == ncclSuccess && cnt_of(j, g) > 0)
					e = r.recv(c->join_recv.as<char>() + (size_t)recv_off(g, j) * item, (size_t)cnt_of(j, g) * item, ncclUint8, j, comms[(size_t)g], c->stream);
			}
		}
		const ncclResult_t e2 = r.group_end();
		if (e != ncclSuccess || e2 != ncclSuccess) return fail(DMND_E_DEVICE, std::string("dmnd_join_ranks: RCCL exchange: ") + r.error_string(e != ncclSuccess ? e : e2));
	}
	else {
		for (int g = 0; g < N; ++g) {
			dmnd_ctx* c = ctx[g];
			HIP_TRY(hipSetDevice(c->device));
			for (int j = 0; j < N; ++j)
				if (cnt_of(g, j) > 0)
					HIP_TRY(hipMemcpyAsync(ctx[j]->join_recv.as<char>() + (size_t)recv_off(j, g) * item, c->join_in.as<char>() + (size_t)send_off(g, j) * item,
						(size_t)cnt_of(g, j) * item, hipMemcpyDeviceToDevice, c->stream));
		}
		for (int g = 0; g < N; ++g) HIP_TRY(sync_stream(ctx[g]->stream));      // an owner's merge reads what every source's stream wrote
	}
	// 4. every owner merges its query range on its device (concurrently: one host thread per owner), 5. survivors to the host in owner order
	std::vector<int64_t> kept((size_t)N, 0);
	std::vector<int> rcs((size_t)N, DMND_OK);
	std::vector<std::string> errs((size_t)N);
	auto merge = [&](int g) {
		dmnd_ctx* c = ctx[g];
		if (n_recv[(size_t)g] == 0) return;
		rcs[(size_t)g] = dmnd_join_blocks_device(c, c->join_recv.as<dmnd_match>(), n_recv[(size_t)g], max_target_seqs, top_percent, (uint32_t)std::min<int64_t>(n_queries - 1, 0xffffffffLL),
			c->join_out.as<dmnd_match>(), &kept[(size_t)g]);
		if (rcs[(size_t)g] != DMND_OK) errs[(size_t)g] = dmnd_last_error();
	};
	if (N == 1) merge(0);
	else {
		std::vector<std::thread> th;
		for (int g = 0; g < N; ++g) th.emplace_back(merge, g);
		for (std::thread& t : th) t.join();
	}
	int64_t total = 0;
	for (int g = 0; g < N; ++g) {
		if (rcs[(size_t)g] != DMND_OK) return fail(rcs[(size_t)g], errs[(size_t)g]);
		total += kept[(size_t)g];
	}
	*n_out = total;
	if (total > cap) return fail(DMND_E_CAP, "dmnd_join_ranks: record buffer too small");
	int64_t at = 0;
	for (int g = 0; g < N; ++g) {
		if (kept[(size_t)g] == 0) continue;
		HIP_TRY(hipSetDevice(ctx[g]->device));
		if (int rc = download_bytes(ctx[g], out + at, ctx[g]->join_out.p, (size_t)kept[(size_t)g] * item)) return rc;
		at += kept[(size_t)g];
	}
	return DMND_OK;
}

// The exchange plan of dmnd_join_ranks without a device (tests/test_multigpu_gloo.py feeds it uneven counts for 2 - 8 ranks and moves
// bytes by it): queries[g][i] = query of record i of source g. Outputs, all n x n row-major unless noted: cnt[g][j] records source g
// sends to owner j, send_off[g][j], recv_off[j][g], n_recv[j] (n entries), place[g][i] (counts[g] entries per source, may be NULL) =
// position of record i in g's owner-ordered copy.
extern "C" int dmnd_join_ranks_plan(int n, const int64_t* counts, const uint32_t* const* queries, int64_t n_queries, int64_t* cnt, int64_t* send_off, int64_t* recv_off,
	int64_t* n_recv, int64_t* const* place)
{
	if (n < 1 || n > 64 || !counts || !queries || n_queries < 1 || !cnt || !send_off || !recv_off || !n_recv) return fail(DMND_E_ARG, "dmnd_join_ranks_plan: bad argument");
	for (int g = 0; g < n; ++g) if (counts[g] < 0 || (counts[g] > 0 && !queries[g])) return fail(DMND_E_ARG, "dmnd_join_ranks_plan: bad argument");
	ExchangePlan p;
	if (plan_exchange(n, counts, n_queries, [&](int g, int64_t i) { return (int64_t)queries[g][i]; },
		[&](int g, int64_t i, int64_t k) { if (place && place[g]) place[g][i] = k; }, p) >= 0)
		return fail(DMND_E_ARG, "dmnd_join_ranks_plan: a query lies outside [0, n_queries)");
	for (int g = 0; g < n; ++g)
		for (int j = 0; j < n; ++j) {
			cnt[(size_t)g * n + j] = p.count(g, j);
			send_off[(size_t)g * n + j] = p.send_off(g, j);
			recv_off[(size_t)j * n + g] = p.recv_off(j, g);
		}
	for (int j = 0; j < n; ++j) n_recv[j] = p.n_recv[(size_t)j];
	return DMND_OK;
}
