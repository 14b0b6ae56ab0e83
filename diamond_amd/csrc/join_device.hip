// join_device.hip -- the block join on the device (round 5; SURVEY.md 8(e): "the final top-k merge"; row f3).
//
// What it replaces: join_query's heap merge of the per-block record lists of a query and the GlobalCulling behind it
// (/root/reference/src/output/join_blocks.cpp:129-199, output/target_culling.h:37-105), for match records that are ALREADY in HBM:
// the records a rank has received from the other ranks over RCCL (multigpu.query_range_join, diamond-hip --gpus N), or the records of
// the reference blocks one GPU has searched one after the other. The host form (dmnd_join_blocks / dmnd_join_blocks_top,
// extend_host.hip) stays for --max-hsps != 1 (a target's HSP records travel as a group) and range culling.
//
// The merged order of a query is a total order -- (e-value ascending, score descending, target ordinal ascending) =
// JoinRecord::cmp_evalue, or (score descending, target ascending) = cmp_score with --top -- so the heap merge of sorted lists is
// a sort of their union: three stable LSD radix sorts of a 4-byte permutation (rocPRIM) by (score, target), e-value bits and query,
// never of the 104-byte records; then one kernel decides per record whether it is among the query's first k (or inside the top
// per cent of the query's best bit score), a scan numbers the survivors and one gather writes them out. 104 B read and written
// per record once each, 12-20 B per record and sort pass: HBM-bound, microseconds for the 1e5-1e6 records of a block pair.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "ctx.h"

using namespace dmnd;

namespace {

// pass 0: key = ~score << 32 | target, identity permutation; pass 1: key = bits of the e-value (non-negative doubles order like
// their bit patterns); pass 2: key = query
__global__ void join_keys_kernel(const dmnd_match* __restrict__ r, const uint32_t* __restrict__ perm, int64_t n, int pass, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx_out)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t src = perm ? perm[i] : (uint32_t)i;
	const dmnd_match& m = r[src];
	uint64_t k;
	if (pass == 0) k = ((uint64_t)(~(uint32_t)m.hsp.score) << 32) | (uint64_t)m.target;
	else if (pass == 1) { double e = m.evalue; __builtin_memcpy(&k, &e, 8); }
	else k = (uint64_t)m.query;
	keys[i] = k;
	if (idx_out) idx_out[i] = src;
}

// keep[j] = 1 iff the j-th record of the merged order survives its query's culling: its rank inside the query below k, or
// (1 - bit score / the query's best bit score) * 100 <= top (GlobalCulling::cull, target_culling.h:56-64)
__global__ void join_keep_kernel(const dmnd_match* __restrict__ r, const uint64_t* __restrict__ query_sorted, const uint32_t* __restrict__ perm, int64_t n,
	int k, double top, uint32_t* __restrict__ keep)
{
	const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	const uint64_t q = query_sorted[j];
	int64_t lo = 0, hi = j;                                   // first record of the query: the merged order is sorted by query
	while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (query_sorted[mid] < q) lo = mid + 1; else hi = mid; }
	if (top < 0.0) { keep[j] = j - lo < (int64_t)k ? 1u : 0u; return; }
	const double best = r[perm[lo]].bit_score, mine = r[perm[j]].bit_score;
	keep[j] = (1.0 - mine / best) * 100.0 <= top ? 1u : 0u;
}

__global__ void join_gather_kernel(const dmnd_match* __restrict__ r, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pos,
	int64_t n, dmnd_match* __restrict__ out)
{
	// 104-byte records as 26 dwords: the lanes of a wavefront move consecutive dwords of consecutive output records
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t j = t / 26;
	const int w = (int)(t - j * 26);
	if (j >= n || !keep[j]) return;
	reinterpret_cast<uint32_t*>(out + pos[j])[w] = reinterpret_cast<const uint32_t*>(r + perm[j])[w];
}

hipError_t sort_u64(const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, int64_t n, int bits, void** tmp, size_t* tmp_bytes, hipStream_t st)
{
	size_t need = 0;
	hipError_t e = rocprim::radix_sort_pairs(nullptr, need, kin, kout, vin, vout, (size_t)n, 0, bits, st);
	if (e != hipSuccess) return e;
	if (need > *tmp_bytes) {
		if (*tmp) (void)hipFree(*tmp);
		*tmp = nullptr; *tmp_bytes = 0;
		e = hipMalloc(tmp, need);
		if (e != hipSuccess) return e;
		*tmp_bytes = need;
	}
	return rocprim::radix_sort_pairs(*tmp, need, kin, kout, vin, vout, (size_t)n, 0, bits, st);
}

}  // namespace

static_assert(sizeof(dmnd_match) == 104, "join_gather_kernel moves records as 26 dwords");

// records_dev / out_dev: HBM of ctx's device; they must not overlap. The call is enqueued on the context's stream and returns when
// the result (and *n_out) are final. max_query: the largest query id among the records (the width of the last sort's key), or 0 = unknown.
extern "C" int dmnd_join_blocks_device(dmnd_ctx* c, const dmnd_match* records_dev, int64_t n, int max_target_seqs, double top_percent, uint32_t max_query,
	dmnd_match* out_dev, int64_t* n_out)
{
	if (!c || n < 0 || (n > 0 && (!records_dev || !out_dev)) || max_target_seqs < 1 || top_percent > 100.0 || !n_out) return fail(DMND_E_ARG, "dmnd_join_blocks_device: bad argument");
	*n_out = 0;
	if (n == 0) return DMND_OK;
	if (n > 0xffffffffLL) return fail(DMND_E_CAP, "dmnd_join_blocks_device: more than 2^32 records");
	HIP_TRY(hipSetDevice(c->device));
	hipStream_t st = c->stream;
	for (int i = 0; i < 2; ++i) {
		if (int rc = c->sort_keys[i].ensure((size_t)n * sizeof(uint64_t))) return rc;
		if (int rc = c->sort_idx[i].ensure((size_t)n * sizeof(uint32_t))) return rc;
	}
	if (int rc = c->join_keep.ensure((size_t)n * sizeof(uint32_t))) return rc;
	if (int rc = c->join_pos.ensure((size_t)n * sizeof(uint32_t))) return rc;
	uint64_t* keys[2] = { c->sort_keys[0].as<uint64_t>(), c->sort_keys[1].as<uint64_t>() };
	uint32_t* idx[2] = { c->sort_idx[0].as<uint32_t>(), c->sort_idx[1].as<uint32_t>() };
	const dim3 grid((unsigned)((n + 255) / 256)), block(256);
	// least significant criterion first; every pass is stable
	hipLaunchKernelGGL(join_keys_kernel, grid, block, 0, st, records_dev, (const uint32_t*)nullptr, n, 0, keys[0], idx[0]);
	HIP_TRY(sort_u64(keys[0], keys[1], idx[0], idx[1], n, 64, &c->sort_tmp, &c->sort_tmp_bytes, st));
	int cur = 1;
	if (top_percent < 0.0) {
		hipLaunchKernelGGL(join_keys_kernel, grid, block, 0, st, records_dev, (const uint32_t*)idx[1], n, 1, keys[0], (uint32_t*)nullptr);
		HIP_TRY(sort_u64(keys[0], keys[1], idx[1], idx[0], n, 64, &c->sort_tmp, &c->sort_tmp_bytes, st));
		cur = 0;
	}
	int query_bits = 32;
	if (max_query > 0) { query_bits = 1; while (query_bits < 32 && (max_query >> query_bits) != 0) ++query_bits; }
	hipLaunchKernelGGL(join_keys_kernel, grid, block, 0, st, records_dev, (const uint32_t*)idx[cur], n, 2, keys[0], (uint32_t*)nullptr);
	HIP_TRY(sort_u64(keys[0], keys[1], idx[cur], idx[cur ^ 1], n, query_bits, &c->sort_tmp, &c->sort_tmp_bytes, st));
	const uint32_t* perm = idx[cur ^ 1];
	uint32_t* keep = c->join_keep.as<uint32_t>();
	uint32_t* pos = c->join_pos.as<uint32_t>();
	hipLaunchKernelGGL(join_keep_kernel, grid, block, 0, st, records_dev, (const uint64_t*)keys[1], perm, n, max_target_seqs, top_percent, keep);
	size_t need = 0;
	HIP_TRY(rocprim::exclusive_scan(nullptr, need, keep, pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
	if (need > c->sort_tmp_bytes) {
		if (c->sort_tmp) (void)hipFree(c->sort_tmp);
		c->sort_tmp = nullptr; c->sort_tmp_bytes = 0;
		HIP_TRY(hipMalloc(&c->sort_tmp, need));
		c->sort_tmp_bytes = need;
	}
	HIP_TRY(rocprim::exclusive_scan(c->sort_tmp, need, keep, pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
	hipLaunchKernelGGL(join_gather_kernel, dim3((unsigned)((n * 26 + 255) / 256)), block, 0, st, records_dev, perm, (const uint32_t*)keep, (const uint32_t*)pos, n, out_dev);
	HIP_TRY(hipGetLastError());
	uint32_t last[2] = { 0, 0 };                                // survivors = pos[n - 1] + keep[n - 1]
	if (int rc = download_bytes(c, &last[0], pos + (n - 1), sizeof(uint32_t))) return rc;
	if (int rc = download_bytes(c, &last[1], keep + (n - 1), sizeof(uint32_t))) return rc;
	*n_out = (int64_t)last[0] + (int64_t)last[1];
	return DMND_OK;
}

// The same for records in host memory (one upload, the join, one download of the survivors): what a single GPU that has searched
// several reference blocks calls instead of dmnd_join_blocks -- the sort runs on the device beside the next block pair's host phases.
extern "C" int dmnd_join_blocks_device_host(dmnd_ctx* c, dmnd_match* records, int64_t n, int max_target_seqs, double top_percent, int64_t* n_out)
{
	if (!c || n < 0 || (n > 0 && !records) || !n_out) return fail(DMND_E_ARG, "dmnd_join_blocks_device_host: bad argument");
	*n_out = 0;
	if (n == 0) return DMND_OK;
	HIP_TRY(hipSetDevice(c->device));
	if (int rc = c->join_in.ensure((size_t)n * sizeof(dmnd_match))) return rc;
	if (int rc = c->join_out.ensure((size_t)n * sizeof(dmnd_match))) return rc;
	uint32_t max_query = 0;
	for (int64_t i = 0; i < n; ++i) max_query = records[i].query > max_query ? records[i].query : max_query;
	HIP_TRY(copy_now(c->stream, c->join_in.p, records, (size_t)n * sizeof(dmnd_match), hipMemcpyHostToDevice));
	if (int rc = dmnd_join_blocks_device(c, c->join_in.as<dmnd_match>(), n, max_target_seqs, top_percent, max_query, c->join_out.as<dmnd_match>(), n_out)) return rc;
	if (*n_out > 0) if (int rc = download_bytes(c, records, c->join_out.p, (size_t)*n_out * sizeof(dmnd_match))) return rc;
	return DMND_OK;
}
