// seed_sj_kernels.hip -- round 5 experiment, off by default (DMND_SEED_SJ=1): the short-seed join of the reference stream as a
// SCATTER + JOIN over 64 key partitions instead of the fused by-class stream kernel (seed_kernels.hip).
//
// Why: the fused kernel's time is random fabric reads -- ~1 miss per joined window whatever the probe
// structure, because a key class's query side (table eighth, lists, folded windows: 9-13 MB) does not fit an XCD's 4 MB L2. A join
// whose random accesses DO fit needs partitions eight times smaller, i.e. the windows of a partition brought together first:
//   K1 seed_sj_scatter_kernel: the reference block is streamed as the long-seed stream does it (16 window starts per thread, keys
//      from nibble windows, level-1 filter probes in batches); a level-1 positive -- 27 % real joins + the filter's false
//      positives -- becomes a 32-byte entry { compact key, position, the window's 48 folded letters } in the slab of its
//      (workgroup, partition). A slab is private to its workgroup: ranks come from an LDS counter per partition, entries of a slab
//      are written back to back, NO global cursor and no global atomic (round 3's partitioned join spent 6 ms per shape on 1e8
//      returning fabric atomics and partial-line writes). A full slab spills to a global overflow list (rare).
//   K2 seed_sj_join_kernel: partition p = (key class c, three top bits of the hash) owns a contiguous 1/64 of the table
//      (SeedArgs::home with parts = 64); workgroups of class c run on XCD c (blockIdx mod 8, the same affinity the by-class kernel
//      uses) and walk the sub-partitions in order, so that at any time an XCD probes ~1/64 of the query side: 2 MB of slots + the
//      1.5 MB of folded query windows. An entry carries everything the Hamming pre-filter needs from the reference side.
//   seed_sj_overflow_kernel: the spilled windows, one thread each.
// Output = the fused kernel's: JOINED marks, the joined-position list (slot, position) for the deferred pass, the survivor list.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "seed_core.h"
#include "seed_kernels.h"

namespace dmnd {

namespace {

__device__ __forceinline__ int sj_window_identity(const uint32_t* a, const uint32_t* b)
{
	int n = 0;
#pragma unroll
	for (int w = 0; w < 12; ++w) {
		const uint32_t d = (a[w] ^ b[w]) & 0x1f1f1f1fu;
		n += 4 - __builtin_popcount((d + 0x7f7f7f7fu) & 0x80808080u);
	}
	return n;
}

__device__ __forceinline__ uint32_t sj_reduce4(uint32_t letter, uint64_t map_lo, uint64_t map_hi)
{
	const uint64_t m = (letter & 16) ? map_hi : map_lo;
	return (uint32_t)(m >> ((letter & 15) * 4)) & 15u;
}

// the 48 folded letters [pos - 16, pos + 32) of a block's folded copy (4 bits per letter, two per byte)
__device__ __forceinline__ void sj_fold_window(const uint8_t* fold, int64_t pos, uint32_t (&tf)[6])
{
	const int64_t t0 = pos - 16;
	uint32_t raw[8];
	__builtin_memcpy(raw, fold + (t0 >> 1), 32);
	const uint32_t sh = (uint32_t)(t0 & 1) * 4;
#pragma unroll
	for (int w = 0; w < 6; ++w) tf[w] = __builtin_amdgcn_alignbit(raw[w + 1], raw[w], sh);
}

// key <-> its 4 bits per care position, packed (weight <= 8)
__device__ __forceinline__ uint32_t sj_compact(const SeedParams& c, int sid, uint64_t key)
{
	uint32_t k = 0;
	for (int j = 0; j < c.shape_weight[sid]; ++j) k |= ((uint32_t)(key >> (4 * c.shape_pos[sid][j])) & 15u) << (4 * j);
	return k;
}
__device__ __forceinline__ uint64_t sj_expand(const SeedParams& c, int sid, uint32_t k)
{
	uint64_t key = 0;
	for (int j = 0; j < c.shape_weight[sid]; ++j) key |= (uint64_t)((k >> (4 * j)) & 15u) << (4 * c.shape_pos[sid][j]);
	return key;
}

__device__ __forceinline__ uint32_t sj_part(uint64_t key, uint32_t h) { return (seed_class(key) << 3) | (h >> 29); }

// one (reference window, query position) pair: the pre-filter on folded letters, the exact identity count for the few that pass
__device__ __forceinline__ bool sj_pair_passes(const SeedArgs& a, const uint32_t (&tf)[6], uint32_t x, int64_t pos)
{
	const int64_t x0 = a.q_begin + (int64_t)x - 16;
	uint32_t raw[8];
	__builtin_memcpy(raw, a.qfold + (x0 >> 1), 32);
	const uint32_t sh = (uint32_t)(x0 & 1) * 4;
	int mism = 0;
#pragma unroll
	for (int w = 0; w < 6; ++w) {
		const uint32_t d = tf[w] ^ __builtin_amdgcn_alignbit(raw[w + 1], raw[w], sh);
		mism += __builtin_popcount((((d & 0x77777777u) + 0x77777777u) | d) & 0x88888888u);
	}
	if (48 - mism < a.params.hamming_filter_id) return false;
	uint32_t qw[12], tw[12];
	__builtin_memcpy(qw, a.qdata + x0, 48);
	__builtin_memcpy(tw, a.tdata + pos - 16, 48);
	return sj_window_identity(tw, qw) >= a.params.hamming_filter_id;
}

// Everything behind a level-1 positive whose key, position and folded window are at hand: table probe, JOINED mark, and for the
// seed's query positions the Hamming pre-filter on folded letters, the exact count for the few that pass.
// on_join(slot, head, count) -> true: the caller filters the list itself (heavy lists); on_survivor(slot, x).
template<typename OnJoin, typename OnSurvivor>
__device__ __forceinline__ void sj_join_window(const SeedArgs& a, uint64_t key, uint32_t h, int64_t pos, const uint32_t (&tf)[6], OnJoin&& on_join, OnSurvivor&& on_survivor)
{
	uint64_t slot = a.home((uint64_t)h, key);
	SeedSlot sl = a.slot(slot);
	for (;;) {
		if (sl.key == SEED_EMPTY) return;
		if (sl.key == key) break;
		slot = (slot + 1) & a.slot_mask;
		sl = a.slot(slot);
	}
	if (!(sl.flags & SLOT_JOINED)) a.slot(slot).flags = sl.flags | SLOT_JOINED;      // benign race: every writer stores the same value
	if (sl.flags & SLOT_LOWC) return;
	const uint32_t count = sl.flags >> 8;
	if (on_join((uint32_t)slot, sl.head, count)) return;
	for (uint32_t i = 0; i < count; ++i) {
		const uint32_t x = count == 1 ? sl.head : a.qlist[sl.head + i];
		const int64_t x0 = a.q_begin + (int64_t)x - 16;
		uint32_t raw[8];
		__builtin_memcpy(raw, a.qfold + (x0 >> 1), 32);
		const uint32_t sh = (uint32_t)(x0 & 1) * 4;
		int mism = 0;
#pragma unroll
		for (int w = 0; w < 6; ++w) {
			const uint32_t d = tf[w] ^ __builtin_amdgcn_alignbit(raw[w + 1], raw[w], sh);
			mism += __builtin_popcount((((d & 0x77777777u) + 0x77777777u) | d) & 0x88888888u);
		}
		if (48 - mism < a.params.hamming_filter_id) continue;
		uint32_t qw[12], tw[12];
		__builtin_memcpy(qw, a.qdata + x0, 48);
		__builtin_memcpy(tw, a.tdata + pos - 16, 48);
		if (sj_window_identity(tw, qw) >= a.params.hamming_filter_id) on_survivor((uint32_t)slot, x);
	}
}

}  // namespace

// ---- K1 ---------------------------------------------------------------------------------------------------------------
// Per tile of 4096 window starts: (A) the probes -- a thread's 16 windows in two batches of eight, positives only QUEUED in LDS
// (key, offset, partition); (B) all threads take queue entries in turn: rank in the partition's slab from an LDS counter, the folded
// window, one 32-byte entry. (First version: the entry was written where the positive was found -- a lane's positives one after the
// other, each a chain LDS atomic -> two loads -> two stores, with a third of the lanes busy: 3.5 ms per shape.)
__global__ __launch_bounds__(256) void seed_sj_scatter_kernel(SeedArgs a, SeedSjArgs j, int sid, uint64_t map_lo, uint64_t map_hi, int64_t base, uint64_t care64)
{
	constexpr unsigned QCAP = 2560;                            // positives of a tile: 1270 expected (31 %)
	__shared__ unsigned cnt[SEED_SJ_PARTS];
	__shared__ uint64_t q_key[QCAP];
	__shared__ uint32_t q_meta[QCAP];                          // offset in the tile (12 bits) | partition << 12
	__shared__ unsigned q_n;
	if (threadIdx.x < SEED_SJ_PARTS) cnt[threadIdx.x] = 0;
	if (threadIdx.x == 0) q_n = 0;
	__syncthreads();
	SeedSjEntry* const my_slabs = j.slabs + (size_t)blockIdx.x * SEED_SJ_PARTS * SEED_SJ_SLAB;
	const int len = a.params.shape_len[sid];
	const uint32_t care = a.params.shape_mask[sid], span = (1u << len) - 1;      // len <= 16
	auto emit = [&](uint64_t key, uint32_t part, int64_t pos) {
		const unsigned r = atomicAdd(&cnt[part], 1u);
		if (r >= j.slab_limit) {                                 // slab full: the window goes to the overflow list
			const unsigned long long o = atomicAdd(j.overflow_count, 1ull);
			if (o < (unsigned long long)j.overflow_cap) { j.overflow[2 * o] = key; j.overflow[2 * o + 1] = (uint64_t)pos; }
			return;
		}
		uint32_t tf[6];
		sj_fold_window(a.tfold, pos, tf);
		uint4* e = reinterpret_cast<uint4*>(my_slabs + (size_t)part * SEED_SJ_SLAB + r);
		e[0] = make_uint4(sj_compact(a.params, sid, key), (uint32_t)(pos - base), tf[0], tf[1]);
		e[1] = make_uint4(tf[2], tf[3], tf[4], tf[5]);
	};
#pragma unroll 1
	for (int sub = 0; sub < SEED_SJ_TILES; ++sub) {
		const int64_t tile0 = base + ((int64_t)blockIdx.x * SEED_SJ_TILES + sub) * 4096;
		const int64_t p0 = tile0 + (int64_t)threadIdx.x * 16;
		if (p0 < a.t_end) {
			typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
			const u32x4 v0 = *reinterpret_cast<const u32x4*>(a.tseed + p0), v1 = *reinterpret_cast<const u32x4*>(a.tseed + p0 + 16);
			const uint32_t w[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
			uint64_t codes[2] = { 0, 0 };
			uint32_t delim = 0, bad = 0;
#pragma unroll
			for (int k = 0; k < 32; ++k) {
				const uint32_t l = (w[k >> 2] >> ((k & 3) * 8)) & LETTER_MASK;
				const uint32_t c = sj_reduce4(l, map_lo, map_hi);
				codes[k >> 4] |= (uint64_t)c << ((k & 15) * 4);
				delim |= (l == L_DELIM ? 1u : 0u) << k;
				bad |= (c == 15u ? 1u : 0u) << k;
			}
			const int64_t first = a.t_begin - p0, last = a.t_end - p0;                 // valid window starts: first <= i < last
#pragma unroll 1
			for (int half = 0; half < 2; ++half) {
				uint64_t key[8];
				uint32_t hash[8], pos_mask = 0;
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					const int sh = (8 * half + i) * 4;
					key[i] = (sh == 0 ? codes[0] : (codes[0] >> sh) | (codes[1] << (64 - sh))) & care64;
				}
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					const int w0 = 8 * half + i;
					const bool ok = w0 >= first && w0 < last && ((delim >> w0) & span) == 0 && ((bad >> w0) & care) == 0;
					hash[i] = seed_hash_a(key[i]);
					const uint32_t bw = ok ? a.bitmap1[a.bm1_index(hash[i], key[i])] : 0u;
					const uint32_t need = bm1_bits(hash[i], a.bitmap1_k3);
					pos_mask |= ((bw & need) == need ? 1u : 0u) << i;
				}
				if (pos_mask) {
					unsigned at = atomicAdd(&q_n, (unsigned)__builtin_popcount(pos_mask));
#pragma unroll
					for (int i = 0; i < 8; ++i)
						if ((pos_mask >> i) & 1u) {
							const uint32_t part = sj_part(key[i], hash[i]);
							if (at < QCAP) { q_key[at] = key[i]; q_meta[at] = (uint32_t)(threadIdx.x * 16 + 8 * half + i) | (part << 12); }
							else emit(key[i], part, p0 + 8 * half + i);      // queue full (a tile of nothing but positives): on the spot
							++at;
						}
				}
			}
		}
		__syncthreads();
		const unsigned n = q_n < QCAP ? q_n : QCAP;
		for (unsigned e = threadIdx.x; e < n; e += 256) emit(q_key[e], q_meta[e] >> 12, tile0 + (int64_t)(q_meta[e] & 4095u));
		__syncthreads();
		if (threadIdx.x == 0) q_n = 0;
		__syncthreads();
	}
	if (threadIdx.x < SEED_SJ_PARTS) j.counts[(size_t)blockIdx.x * SEED_SJ_PARTS + threadIdx.x] = cnt[threadIdx.x] < j.slab_limit ? cnt[threadIdx.x] : j.slab_limit;
}

// ---- K2 ---------------------------------------------------------------------------------------------------------------
// A workgroup takes the slabs of G consecutive scatter workgroups for one partition. Phase 1: one entry per thread and turn -- table
// probe, JOINED mark, the join staged in LDS (slot, entry, list start and size). Phase 2: the (join, list element) pairs of ALL staged
// joins, one per thread and turn: short lists from a pair list, long ones (> LIGHT) through a prefix over their sizes -- never a thread
// walking a list (the wavefront waits for its longest: the by-class kernel's lesson) and never a list at a time (the
// first version filtered the long lists one after the other with the whole workgroup: ~2 us each, 16 per workgroup, most of the
// kernel's 4.7 ms). A pair reads its reference window from the entry again (L2: the workgroup has just read it).
__global__ __launch_bounds__(256, 8) void seed_sj_join_kernel(SeedArgs a, SeedSjArgs j, int sid, int64_t base, int units_per_part)
{
	// (the kernel's time is its workgroups' chains of memory round trips over the workgroups a CU holds: small LDS -- the staging area
	// is sized to the EXPECTED joins of a unit, 600, with a slow path behind it -- and several independent loads per thread in flight)
	constexpr unsigned G = SEED_SJ_GROUP, STAGE = 768, SURV = 192, HEAVY = 96, LIGHT = 8, PAIRS = 1536;
	__shared__ unsigned pre[G + 1];
	__shared__ uint32_t st_slot[STAGE], st_pos[STAGE], st_head[STAGE];
	__shared__ uint16_t st_ent[STAGE], st_count[STAGE];      // entry number in the unit; list size, saturated (a list that long is read back from its slot)
	__shared__ uint16_t pr[PAIRS];                             // join << 3 | element
	__shared__ uint16_t hv_k[HEAVY];
	__shared__ unsigned hv_pre[HEAVY + 1];
	__shared__ uint32_t sv_slot[SURV], sv_x[SURV], sv_pos[SURV];
	__shared__ unsigned st_n, sv_n, hv_n, pr_n;
	__shared__ unsigned long long st_base;
	static_assert(STAGE <= 8192, "a pair carries its join in 13 bits");
	// DMND_SEED_PHASES=1 (SeedArgs::phase_ticks): thread 0 adds the 100 MHz ticks between the workgroup's phase boundaries
	uint64_t phase_t = a.phase_ticks ? wall_clock64() : 0;
	auto mark = [&](int i) { if (a.phase_ticks && threadIdx.x == 0) { const uint64_t now = wall_clock64(); atomicAdd(&a.phase_ticks[i], (unsigned long long)(now - phase_t)); phase_t = now; } };
	const uint32_t cls = blockIdx.x & 7u;
	const int64_t jj = (int64_t)(blockIdx.x >> 3);
	const uint32_t sub = (uint32_t)(jj / units_per_part);
	const int64_t u = jj % units_per_part;
	if (sub >= 8) return;
	const uint32_t part = (cls << 3) | sub;
	const int64_t wg0 = u * G;
	if (threadIdx.x == 0) { st_n = 0; sv_n = 0; hv_n = 0; pr_n = 0; pre[0] = 0; }
	if (threadIdx.x < G) {
		const int64_t wg = wg0 + threadIdx.x;
		pre[threadIdx.x + 1] = wg < j.n_wg ? j.counts[(size_t)wg * SEED_SJ_PARTS + part] : 0u;
	}
	__syncthreads();
	if (threadIdx.x == 0) for (unsigned g = 0; g < G; ++g) pre[g + 1] += pre[g];
	__syncthreads();
	const unsigned E = pre[G];
	mark(0);
	auto entry_at = [&](unsigned e) {
		unsigned g = 0;
#pragma unroll
		for (unsigned step = G / 2; step > 0; step >>= 1) if (pre[g + step] <= e) g += step;
		return j.slabs + ((size_t)(wg0 + g) * SEED_SJ_PARTS + part) * SEED_SJ_SLAB + (e - pre[g]);
	};
	auto survive = [&](uint32_t slot, uint32_t x, uint32_t rel) {
		const unsigned k = atomicAdd(&sv_n, 1u);
		if (k < SURV) { sv_slot[k] = slot; sv_x[k] = x; sv_pos[k] = rel; }
		else {
			const unsigned long long idx = atomicAdd(a.survivor_count, 1ull);
			if (idx < (unsigned long long)a.survivor_cap) a.survivors[idx] = SeedSurvivor{ slot, x, base + (int64_t)rel };
		}
	};
	// phase 1: four entries per thread in flight (keys, then their home slots)
	auto slow_join = [&](uint32_t slot, uint32_t head, uint32_t cnt, unsigned e, uint32_t rel) {      // no room in the staging area: everything here
		const unsigned long long idx = atomicAdd(a.matched_count, 1ull);
		if (idx < (unsigned long long)a.matched_cap) { a.matched_slot[idx] = slot; a.matched_loc[idx] = base + (int64_t)rel; }
		const SeedSjEntry* src = entry_at(e);
		const uint4 e0 = reinterpret_cast<const uint4*>(src)[0], e1 = reinterpret_cast<const uint4*>(src)[1];
		const uint32_t tf[6] = { e0.z, e0.w, e1.x, e1.y, e1.z, e1.w };
		for (uint32_t i = 0; i < cnt; ++i) {
			const uint32_t x = cnt == 1 ? head : a.qlist[head + i];
			if (sj_pair_passes(a, tf, x, base + (int64_t)rel)) survive(slot, x, rel);
		}
	};
	constexpr int B1 = 2;
	for (unsigned e0 = threadIdx.x; e0 < E; e0 += B1 * 256) {
		uint2 kp[B1];
		uint64_t key[B1], slot[B1];
		SeedSlot sl[B1];
#pragma unroll
		for (int b = 0; b < B1; ++b) { const unsigned e = e0 + (unsigned)b * 256; kp[b] = e < E ? *reinterpret_cast<const uint2*>(entry_at(e)) : make_uint2(0u, 0u); }
#pragma unroll
		for (int b = 0; b < B1; ++b) {
			key[b] = sj_expand(a.params, sid, kp[b].x);
			slot[b] = a.home((uint64_t)seed_hash_a(key[b]), key[b]);
			if (e0 + (unsigned)b * 256 < E) sl[b] = a.slot(slot[b]);
		}
#pragma unroll
		for (int b = 0; b < B1; ++b) {
			const unsigned e = e0 + (unsigned)b * 256;
			if (e >= E) continue;
			bool found = false;
			for (;;) {
				if (sl[b].key == SEED_EMPTY) break;
				if (sl[b].key == key[b]) { found = true; break; }
				slot[b] = (slot[b] + 1) & a.slot_mask;
				sl[b] = a.slot(slot[b]);
			}
			if (!found) continue;
			if (!(sl[b].flags & SLOT_JOINED)) a.slot(slot[b]).flags = sl[b].flags | SLOT_JOINED;      // benign race: every writer stores the same value
			if (sl[b].flags & SLOT_LOWC) continue;
			const unsigned k = atomicAdd(&st_n, 1u);
			if (k >= STAGE) { slow_join((uint32_t)slot[b], sl[b].head, sl[b].flags >> 8, e, kp[b].y); continue; }
			st_slot[k] = (uint32_t)slot[b]; st_pos[k] = kp[b].y; st_head[k] = sl[b].head; st_ent[k] = (uint16_t)e;
			st_count[k] = (uint16_t)((sl[b].flags >> 8) < 0xffffu ? (sl[b].flags >> 8) : 0xffffu);
		}
	}
	__syncthreads();
	mark(1);
	// phase 2: the pairs
	const unsigned n_joined = st_n < STAGE ? st_n : STAGE;
	for (unsigned k = threadIdx.x; k < n_joined; k += 256) {
		const uint32_t count = st_count[k];
		if (count > LIGHT) {
			const unsigned hk = atomicAdd(&hv_n, 1u);
			if (hk < HEAVY) { hv_k[hk] = (uint16_t)k; continue; }
		}
		else {
			const unsigned at = atomicAdd(&pr_n, count);
			if (at + count <= PAIRS) { for (uint32_t i = 0; i < count; ++i) pr[at + i] = (uint16_t)((k << 3) | i); continue; }
			for (unsigned i = at; i < PAIRS; ++i) pr[i] = 0xffffu;       // no room: the slots of the reservation that exist are voided
		}
		// no room in the lists: this thread walks the join's list itself
		uint32_t cnt = count == 0xffffu ? a.slot(st_slot[k]).flags >> 8 : count;
		const SeedSjEntry* src = entry_at(st_ent[k]);
		const uint4 e0 = reinterpret_cast<const uint4*>(src)[0], e1 = reinterpret_cast<const uint4*>(src)[1];
		const uint32_t tf[6] = { e0.z, e0.w, e1.x, e1.y, e1.z, e1.w };
		for (uint32_t i = 0; i < cnt; ++i) {
			const uint32_t x = cnt == 1 ? st_head[k] : a.qlist[st_head[k] + i];
			if (sj_pair_passes(a, tf, x, base + (int64_t)st_pos[k])) survive(st_slot[k], x, st_pos[k]);
		}
	}
	__syncthreads();
	mark(2);
	const unsigned n_heavy = hv_n < HEAVY ? hv_n : HEAVY;
	if (threadIdx.x == 0) {
		unsigned acc = 0;
		for (unsigned h = 0; h < n_heavy; ++h) {
			hv_pre[h] = acc;
			const unsigned k = hv_k[h];
			acc += st_count[k] == 0xffffu ? a.slot(st_slot[k]).flags >> 8 : st_count[k];
		}
		hv_pre[n_heavy] = acc;
	}
	__syncthreads();
	mark(3);
	const unsigned n_light = pr_n < PAIRS ? pr_n : PAIRS, n_all = n_light + hv_pre[n_heavy];
	auto pair_of = [&](unsigned p, unsigned& k, unsigned& i) {
		if (p < n_light) {
			if (pr[p] == 0xffffu) return false;
			k = pr[p] >> 3; i = pr[p] & 7u;
			return true;
		}
		const unsigned q = p - n_light;
		unsigned lo = 0, hi = n_heavy;                          // the heavy list q falls into: hv_pre[lo] <= q < hv_pre[lo + 1]
		while (hi - lo > 1) { const unsigned mid = (lo + hi) >> 1; if (hv_pre[mid] <= q) lo = mid; else hi = mid; }
		k = hv_k[lo]; i = q - hv_pre[lo];
		return true;
	};
	constexpr int B3 = 1;
	for (unsigned p0 = threadIdx.x; p0 < n_all; p0 += B3 * 256) {
		unsigned k[B3], i[B3];
		bool on[B3];
		uint4 f0[B3], f1[B3];
		uint32_t x[B3];
#pragma unroll
		for (int b = 0; b < B3; ++b) {
			const unsigned p = p0 + (unsigned)b * 256;
			on[b] = p < n_all && pair_of(p, k[b], i[b]);
			if (on[b]) {
				const SeedSjEntry* src = entry_at(st_ent[k[b]]);
				f0[b] = reinterpret_cast<const uint4*>(src)[0]; f1[b] = reinterpret_cast<const uint4*>(src)[1];
				x[b] = st_count[k[b]] == 1 ? st_head[k[b]] : a.qlist[st_head[k[b]] + i[b]];
			}
		}
#pragma unroll
		for (int b = 0; b < B3; ++b) {
			if (!on[b]) continue;
			const uint32_t tf[6] = { f0[b].z, f0[b].w, f1[b].x, f1[b].y, f1[b].z, f1[b].w };
			if (sj_pair_passes(a, tf, x[b], base + (int64_t)st_pos[k[b]])) survive(st_slot[k[b]], x[b], st_pos[k[b]]);
		}
	}
	__syncthreads();
	__syncthreads();
	mark(4);
	const unsigned n_sv = sv_n < SURV ? sv_n : SURV;
	__shared__ unsigned long long sv_base;
	if (threadIdx.x == 0 && n_joined) st_base = atomicAdd(a.matched_count, (unsigned long long)n_joined);      // (the two reservations in flight together)
	if (threadIdx.x == 64 && n_sv) sv_base = atomicAdd(a.survivor_count, (unsigned long long)n_sv);
	__syncthreads();
	for (unsigned k = threadIdx.x; k < n_sv; k += 256) {
		const unsigned long long idx = sv_base + k;
		if (idx < (unsigned long long)a.survivor_cap) a.survivors[idx] = SeedSurvivor{ sv_slot[k], sv_x[k], base + (int64_t)sv_pos[k] };
	}
	for (unsigned k = threadIdx.x; k < n_joined; k += 256) {
		const unsigned long long idx = st_base + k;
		if (idx < (unsigned long long)a.matched_cap) { a.matched_slot[idx] = st_slot[k]; a.matched_loc[idx] = base + (int64_t)st_pos[k]; }
	}
	mark(5);
}

// the windows whose slab was full: one thread each, lists walked whole, appends straight to the global lists
__global__ __launch_bounds__(256) void seed_sj_overflow_kernel(SeedArgs a, SeedSjArgs j, int sid, int64_t n)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t key = j.overflow[2 * i];
	const int64_t pos = (int64_t)j.overflow[2 * i + 1];
	uint32_t tf[6];
	sj_fold_window(a.tfold, pos, tf);
	sj_join_window(a, key, seed_hash_a(key), pos, tf,
		[&](uint32_t slot, uint32_t, uint32_t) {
			const unsigned long long idx = atomicAdd(a.matched_count, 1ull);
			if (idx < (unsigned long long)a.matched_cap) { a.matched_slot[idx] = slot; a.matched_loc[idx] = pos; }
			return false;
		},
		[&](uint32_t slot, uint32_t x) {
			const unsigned long long idx = atomicAdd(a.survivor_count, 1ull);
			if (idx < (unsigned long long)a.survivor_cap) a.survivors[idx] = SeedSurvivor{ slot, x, pos };
		});
}

bool seed_sj_supported(const SeedParams& c)
{
	if (c.seed_encoding != SEED_SPACED) return false;
	for (int sid = 0; sid < c.n_shapes; ++sid)
		if (!seed_nibble_mode(c, sid) || c.shape_weight[sid] > 8) return false;
	return true;
}

int64_t seed_sj_workgroups(int64_t t_begin, int64_t t_end)
{
	const int64_t base = t_begin & ~(int64_t)15, threads = (t_end - base + 15) / 16;
	return (threads + 256 * SEED_SJ_TILES - 1) / (256 * SEED_SJ_TILES);
}

hipError_t launch_seed_sj_scatter(const SeedArgs& a, const SeedSjArgs& j, int sid, hipStream_t st)
{
	const SeedParams& c = a.params;
	uint64_t lo = 0, hi = 0;
	for (int l = 0; l < 32; ++l) {
		const uint64_t code = c.reduction[l] == L_MASK ? 15u : (uint64_t)c.reduction[l];
		(l < 16 ? lo : hi) |= code << ((l & 15) * 4);
	}
	uint64_t care64 = 0;
	for (int k = 0; k < c.shape_weight[sid]; ++k) care64 |= (uint64_t)15 << (4 * c.shape_pos[sid][k]);
	const int64_t base = a.t_begin & ~(int64_t)15;
	hipLaunchKernelGGL(seed_sj_scatter_kernel, dim3((unsigned)j.n_wg), dim3(256), 0, st, a, j, sid, lo, hi, base, care64);
	return hipGetLastError();
}

hipError_t launch_seed_sj_join(const SeedArgs& a, const SeedSjArgs& j, int sid, hipStream_t st)
{
	const int64_t base = a.t_begin & ~(int64_t)15;
	const int units = (int)((j.n_wg + SEED_SJ_GROUP - 1) / SEED_SJ_GROUP);
	hipLaunchKernelGGL(seed_sj_join_kernel, dim3((unsigned)(8 * 8 * units)), dim3(256), 0, st, a, j, sid, base, units);
	return hipGetLastError();
}

hipError_t launch_seed_sj_overflow(const SeedArgs& a, const SeedSjArgs& j, int sid, int64_t n, hipStream_t st)
{
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(seed_sj_overflow_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, j, sid, n);
	return hipGetLastError();
}

}  // namespace dmnd
