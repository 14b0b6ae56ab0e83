"""Synthetic extension-stage workload in the shape of BASELINE config C2 (blastp --fast, 10k queries
x 1M-sequence database), SURVEY.md 8d.

Round 1 models what the reference's extension stage hands to DP::BandedSwipe::swipe after seeding and
chaining (align/gapped_score.cpp:107-180): for every query its true homologs (the members of the family
it was derived from) plus a few spurious seed-hit targets, each with the diagonal band the reference
would build: [d_min - b, d_max + 1 + b) clipped to the matrix, b = Extension::band(qlen, BANDED_FAST)
(align/gapped_score.cpp:41-52). Round 2 re-aligns, with traceback, the targets that survive the
e-value cutoff and top-k culling (max_target_seqs = 25; align/culling.cpp:97-113,189).

Until the seed stage and chaining (a12-a13) are built the band centres come from
this model, not from real seed hits; DESIGN.md states this explicitly."""
import numpy as np
from . import hip, synth

MAX_TARGET_SEQS = 25          # basic/config.h:55


def band_fast(qlen):
    """Extension::band(len, Mode::BANDED_FAST), align/gapped_score.cpp:41-52"""
    qlen = np.asarray(qlen)
    return np.select([qlen < 50, qlen < 100, qlen < 250, qlen < 350], [12, 16, 30, 40], 64)


def sequence_set(data, off):
    """Lays sequences out as the reference's SequenceSet does (data/string_set.h:27-60): 256 x 0x1F perimeter padding,
    then seq, 0x1F, seq, 0x1F, ..., 256 x 0x1F. Returns (data int8[], limits int64[n+1])."""
    lens = np.diff(off)
    limits = 256 + np.concatenate([[0], np.cumsum(lens + 1)])
    out = np.full(int(limits[-1]) + 256, 31, np.int8)
    idx = np.repeat(limits[:-1] - off[:-1], lens) + np.arange(off[-1])
    out[idx] = data
    return out, limits.astype(np.int64)


class Workload:
    def __init__(self, families=100_000, members=10, queries=10_000, spurious=6, seed=20260923):
        self.db, self.doff, self.q, self.qoff, self.qfam = synth.generate(
            families, members=members, queries=queries, seed=seed, family=True)
        self.members = members
        rng = np.random.default_rng(seed & 0xffffffff)
        nq = len(self.qoff) - 1
        ndb = len(self.doff) - 1
        qi_true = np.repeat(np.arange(nq)[self.qfam >= 0], members)
        ti_true = (self.qfam[self.qfam >= 0][:, None] * members + np.arange(members)[None, :]).ravel()
        qi_sp = np.repeat(np.arange(nq), spurious)
        ti_sp = rng.integers(0, ndb, qi_sp.size)
        self.qi = np.concatenate([qi_true, qi_sp])
        self.ti = np.concatenate([ti_true, ti_sp])
        order = np.argsort(self.qi, kind="stable")
        self.qi, self.ti = self.qi[order], self.ti[order]
        true_mask = np.concatenate([np.ones(qi_true.size, bool), np.zeros(qi_sp.size, bool)])[order]
        ql = (self.qoff[self.qi + 1] - self.qoff[self.qi]).astype(np.int64)
        tl = (self.doff[self.ti + 1] - self.doff[self.ti]).astype(np.int64)
        # chain geometry: homologs drift a few diagonals through indels; spurious hits are one short HSP
        centre = np.where(true_mask, rng.integers(-4, 5, self.qi.size), rng.integers(-(tl - 1) // 2, np.maximum(ql // 2, 1)))
        spread = np.where(true_mask, rng.integers(0, 9, self.qi.size), 0)
        b = band_fast(ql)
        d_min, d_max = centre - spread, centre + spread
        self.items = np.zeros(self.qi.size, dtype=hip.DP_TARGET_DTYPE)
        self.items["query_off"] = self.qoff[self.qi]
        self.items["target_off"] = self.doff[self.ti]
        self.items["cbs_off"] = -1
        self.items["query_len"] = ql
        self.items["target_len"] = tl
        self.items["d_begin"] = np.maximum(d_min - b, -(tl - 1))          # gapped_score.cpp:137-138
        self.items["d_end"] = np.minimum(d_max + 1 + b, ql)
        self.n_queries = nq
        self.db_letters = int(self.doff[-1])

    @staticmethod
    def cells(items):
        """DpTarget::cells(): (d_end - d_begin) * banded_cols (dp/dp.h:47-52,121-124) -- the GCUPS denominator."""
        ql, tl = items["query_len"].astype(np.int64), items["target_len"].astype(np.int64)
        d0, d1 = items["d_begin"].astype(np.int64), items["d_end"].astype(np.int64)
        pos = np.maximum(d1 - 1, 0) - (d1 - 1)
        cols = np.minimum(ql - 1 - d0, tl - 1) + 1 - pos
        return (d1 - d0) * cols

    @staticmethod
    def algorithmic_bytes(items):
        """bytes_sw = T*1 + Q*32 + #DpTargets*(32 in + 40 out); T = sum of columns streamed,
        Q = query letters (each distinct query counted once per launch)."""
        ql, tl = items["query_len"].astype(np.int64), items["target_len"].astype(np.int64)
        d0, d1 = items["d_begin"].astype(np.int64), items["d_end"].astype(np.int64)
        pos = np.maximum(d1 - 1, 0) - (d1 - 1)
        cols = np.minimum(ql - 1 - d0, tl - 1) + 1 - pos
        _, first = np.unique(items["query_off"], return_index=True)
        return int(cols.sum() + 32 * ql[first].sum() + 72 * items.size)

    def select_round2(self, params, scores):
        """E-value cutoff + top-k culling between the two swipe rounds (align/culling.cpp:97-113,189):
        order by (evalue asc, score desc, target id asc), keep MAX_TARGET_SEQS per query."""
        ev = hip.evalue_batch(params, scores, self.items["query_len"], self.items["target_len"])
        keep = (scores > 0) & (ev <= params.max_evalue)
        idx = np.nonzero(keep)[0]
        order = np.lexsort((self.ti[idx], -scores[idx].astype(np.int64), ev[idx], self.qi[idx]))
        idx = idx[order]
        q = self.qi[idx]
        start = np.r_[0, np.nonzero(np.diff(q))[0] + 1] if idx.size else np.zeros(0, np.int64)
        rank = np.arange(idx.size) - np.repeat(start, np.diff(np.r_[start, idx.size]))
        return idx[rank < MAX_TARGET_SEQS]
