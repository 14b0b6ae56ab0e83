"""The reference's `--algo auto` choice (dmnd_auto_query_indexed; run/double_indexed.cpp use_query_index): host code, no GPU. The
seed count of a query block close to the 32 Mi-letter limit runs on a thread team (round 4): the answer must not depend on how
the sequences are cut among the threads, and must follow the number of DISTINCT seeds, not of seeds."""
import numpy as np

from diamond_amd import hip, workload


def _block(seqs_flat, n, length):
    off = np.arange(0, (n + 1) * length, length, dtype=np.int64)
    return workload.sequence_set(seqs_flat, off)


def test_auto_algo_counts_distinct_seeds_near_the_limit():
    p = hip.default_params()
    sp, _ = hip.seed_params_preset("default", p)
    rng = np.random.default_rng(3)
    n, length = 100_000, 290                      # 2.9e7 letters: 1.25 x letters rounds up past 32 Mi, so the seeds are counted
    rand = rng.integers(0, 20, n * length).astype(np.int8)
    qd, ql = _block(rand, n, length)
    big_db, small_db = 10 << 30, 100 << 20
    assert hip.auto_query_indexed(sp, qd, ql, small_db) is False          # database below MIN_QUERY_INDEXED_DB_SIZE
    assert hip.auto_query_indexed(sp, qd, ql, big_db) is False            # ~2.8e7 distinct seeds: the table would exceed 32 Mi slots
    # the same number of letters and seeds, but 1000 different sequences repeated: few distinct seeds -> query-indexed
    rep = np.tile(rand[: 1000 * length], n // 1000)
    qd2, ql2 = _block(rep, n, length)
    assert hip.auto_query_indexed(sp, qd2, ql2, big_db) is True
    # a block far below the limit never counts
    qd3, ql3 = _block(rand[: 1000 * length], 1000, length)
    assert hip.auto_query_indexed(sp, qd3, ql3, big_db) is True
