// tests/emu/seed_emu.cpp -- TEST INFRASTRUCTURE ONLY.
// CPU emulation of the GPU seed-stage data flow (diamond_amd/csrc/seed_kernels.hip) built from the SAME per-thread
// code (diamond_amd/csrc/seed_core.h): index the query seeds, stream the reference once, mark joined seeds, derive the
// per-letter mask times, then filter every (query, reference) pair independently. Checks on the CPU that the
// order-free formulation reproduces the reference's sequential index-chunk semantics. Never used by the product.
#include <algorithm>
#include <climits>
#include <cstdint>
#include <unordered_map>
#include <vector>
#include "../../diamond_amd/csrc/seed_core.h"

using namespace dmnd;

struct EmuHit { uint32_t query; int32_t seed_offset; int64_t subject; int32_t score; int32_t pad; };

// qseed / tseed (may be NULL): the motif-soft-masked views of the blocks that seeds are generated from (dmnd_soft_mask_block)
extern "C" int64_t emu_seed_search_soft(const SeedParams* cp, const int8_t* matrix, const int8_t* qdata, const int64_t* qlimits, int64_t nq,
	const int8_t* tdata, const int64_t* tlimits, int64_t nt, const int8_t* qseed, const int8_t* tseed, EmuHit* hits, int64_t cap);

extern "C" int64_t emu_seed_search(const SeedParams* cp, const int8_t* matrix, const int8_t* qdata, const int64_t* qlimits, int64_t nq,
	const int8_t* tdata, const int64_t* tlimits, int64_t nt, EmuHit* hits, int64_t cap)
{
	return emu_seed_search_soft(cp, matrix, qdata, qlimits, nq, tdata, tlimits, nt, nullptr, nullptr, hits, cap);
}

extern "C" int64_t emu_seed_search_soft(const SeedParams* cp, const int8_t* matrix, const int8_t* qdata, const int64_t* qlimits, int64_t nq,
	const int8_t* tdata, const int64_t* tlimits, int64_t nt, const int8_t* qseed, const int8_t* tseed, EmuHit* hits, int64_t cap)
{
	const SeedParams& c = *cp;
	const int64_t qraw = qlimits[nq], traw = tlimits[nt];
	std::vector<uint8_t> mask_time((size_t)qraw + 256, SEED_NEVER);
	if (!tseed || c.seed_encoding == SEED_HASHED) tseed = tdata;
	if (qseed) {
		// seed_soft_time_kernel
		for (int64_t p = qlimits[0]; p < qraw; ++p) {
			if ((qdata[p] & LETTER_MASK) == L_DELIM) continue;
			int first_soft = 1 << 30;
			for (int r = 0; r < 32 && (qdata[p + r] & LETTER_MASK) != L_DELIM; ++r)
				if (qseed[p + r] != qdata[p + r]) { first_soft = r; break; }
			for (int sid = 0; sid < c.n_shapes; ++sid)
				if (first_soft < c.shape_len[sid]) { mask_time[(size_t)p] = (uint8_t)(sid * c.index_chunks); break; }
		}
	}
	else qseed = qdata;
	std::vector<uint32_t> qid_of((size_t)qraw, 0);
	for (int64_t i = 0; i < nq; ++i)
		for (int64_t p = qlimits[i]; p < qlimits[i + 1]; ++p) qid_of[(size_t)p] = (uint32_t)i;
	struct Group { std::vector<int64_t> q; bool present = false, erased = false, need = false; };
	std::vector<std::unordered_map<uint64_t, Group>> tables(c.n_shapes);
	std::vector<std::vector<std::pair<uint64_t, int64_t>>> matched(c.n_shapes);
	// phase 1: index queries, stream the reference, complexity masks -- for every shape
	const bool hashed = c.seed_encoding == SEED_HASHED;         // query-indexed algorithm: seed_index_kernel filters + masks, no seed_mask_kernel
	for (int sid = 0; sid < c.n_shapes; ++sid) {
		auto& tab = tables[sid];
		for (int64_t p = qlimits[0]; p < qraw; ++p) {
			uint64_t s;
			if (!(hashed ? seed_key_hashed(c, sid, qseed + p, s) : seed_at(c, sid, qseed + p, s))) continue;
			if (hashed && !seed_is_complex(c, sid, qseed + p)) {
				mask_time[(size_t)p] = (uint8_t)std::min<int>(mask_time[(size_t)p], sid * c.index_chunks);
				continue;
			}
			tab[s].q.push_back(p);
		}
		for (int64_t p = tlimits[0]; p < traw; ++p) {
			uint64_t s;
			if (!(hashed ? seed_key_hashed(c, sid, tseed + p, s) : seed_at(c, sid, tseed + p, s))) continue;
			auto it = tab.find(s);
			if (it == tab.end()) continue;
			it->second.present = true;
			matched[sid].push_back({ s, p });
		}
		for (auto& kv : tab) {
			Group& g = kv.second;
			if (!g.present || hashed) continue;
			const int64_t first = *std::min_element(g.q.begin(), g.q.end());
			if (!seed_is_complex(c, sid, qdata + first)) {
				g.erased = true;
				const int t = sid * c.index_chunks + seed_chunk(c, kv.first);
				for (int64_t p : g.q) mask_time[(size_t)p] = (uint8_t)std::min<int>(mask_time[(size_t)p], t);
			}
		}
	}
	// phase 2: every joined pair independently; pairs scoring above 255 are deferred (seed_pair_kernel)
	struct Deferred { size_t m; int64_t qp; int score; };
	int64_t n = 0;
	for (int sid = 0; sid < c.n_shapes; ++sid) {
		std::vector<Deferred> deferred;
		for (size_t mi = 0; mi < matched[sid].size(); ++mi) {
			const auto& m = matched[sid][mi];
			Group& g = tables[sid][m.first];
			if (g.erased) continue;
			const int chunk = seed_chunk(c, m.first);
			for (int64_t qp : g.q) {
				if (fingerprint_id(qdata + qp, tdata + m.second) < c.hamming_filter_id) continue;
				const uint32_t qid = qid_of[(size_t)qp];
				const int seed_offset = (int)(qp - qlimits[qid]);
				int score = 0xFFFF;
				const int query_len = (int)(qlimits[qid + 1] - qlimits[qid] - 1);
				if (c.use_ungapped) {
					const int cutoff = ungapped_cutoff(c, query_len);
					if (cutoff) {
						const int window = stage2_window(c, query_len);
						int cb, ce;
						clip_window(qdata + qp - window, 2 * window, window, cb, ce);
						const int window_left = window - cb;
						score = ungapped_window_score(matrix, qdata + qp - window_left, tdata + m.second - window_left, ce - cb);
						if (score > 255) { deferred.push_back(Deferred{ mi, qp, score }); g.need = true; continue; }
						if (score <= cutoff) continue;
					}
				}
				if (!left_most_pair(c, qdata + qp, mask_time.data() + qp, tdata + m.second, seed_offset, sid, chunk, query_len)) continue;
				if (n >= cap) return -1;
				hits[n++] = EmuHit{ qid, seed_offset, m.second, score, 0 };
			}
		}
		if (deferred.empty()) continue;
		// seed_collect_kernel + host sort: joined positions of the flagged seeds, ordered by (seed, position)
		std::vector<std::pair<uint64_t, int64_t>> e;
		for (const auto& m : matched[sid]) if (tables[sid][m.first].need) e.push_back(m);
		std::sort(e.begin(), e.end());
		std::vector<uint64_t> e_loc(e.size());                     // low 40 bits = position, as in the device sort key
		for (size_t i = 0; i < e.size(); ++i) e_loc[i] = (uint64_t)e[i].second;
		// seed_deferred_kernel
		for (const Deferred& d : deferred) {
			const auto& m = matched[sid][d.m];
			const size_t b = (size_t)(std::lower_bound(e.begin(), e.end(), std::make_pair(m.first, (int64_t)INT64_MIN)) - e.begin());
			const size_t en = (size_t)(std::upper_bound(e.begin(), e.end(), std::make_pair(m.first, (int64_t)INT64_MAX)) - e.begin());
			int score = d.score;
			if (simd_batch_size_sorted(c, e_loc.data() + b, (int64_t)(en - b), tdata, qdata + d.qp, m.second) >= 4) score = 255;
			const uint32_t qid = qid_of[(size_t)d.qp];
			const int seed_offset = (int)(d.qp - qlimits[qid]);
			const int query_len = (int)(qlimits[qid + 1] - qlimits[qid] - 1);
			if (score <= ungapped_cutoff(c, query_len)) continue;
			if (!left_most_pair(c, qdata + d.qp, mask_time.data() + d.qp, tdata + m.second, seed_offset, sid, seed_chunk(c, m.first), query_len)) continue;
			if (n >= cap) return -1;
			hits[n++] = EmuHit{ qid, seed_offset, m.second, score, 0 };
		}
	}
	return n;
}

// round 4: key classes of the short-seed pipeline (seed_core.h seed_class) and the partitioned indexing built on them
// (SeedArgs::home / bm1_index, seed_kernels.h, restated here: that header needs the HIP runtime)
extern "C" void emu_seed_classes(const uint64_t* keys, int64_t n, uint64_t slot_mask, uint32_t bm1_words, uint32_t* cls, uint64_t* home, uint32_t* word)
{
	for (int64_t i = 0; i < n; ++i) {
		const uint64_t hh = seed_hash(keys[i]);
		const uint32_t c = seed_class(keys[i]);
		const uint64_t low = slot_mask >> 3;
		cls[i] = c;
		home[i] = (uint64_t)c * (low + 1) | (hh & low);
		const uint32_t w8 = bm1_words >> 3;
		word[i] = c * w8 + bm1_word(seed_hash_a(keys[i]), w8);
	}
}
