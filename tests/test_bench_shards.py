"""CPU: the decompositions of bench.py's N > 1 runs cover the job exactly once -- every (query, database sequence) pair is searched
by one rank, whatever --shard: database shards, query shards, and (round 6) two query halves x N / 2 database shards. No device:
only the workload's cut (bench.Workload) is built."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


@pytest.mark.parametrize("shard,world", [("db", 2), ("db", 4), ("query", 3), ("2d", 4), ("2d", 8)])
@pytest.mark.parametrize("cfg", ["C2", "C5"])
def test_shards_tile_the_job(cfg, shard, world):
    cover = None
    n_blocks = set()
    for rank in range(world):
        w = bench.Workload(cfg, 400, 120, world, rank, shard)
        if cover is None:
            cover = np.zeros((w.n_queries, w.n_db), dtype=np.int32)
        assert 0 <= w.q_lo < w.q_hi <= w.n_queries
        assert len(w.ql) - 1 == w.q_hi - w.q_lo                      # the rank's query block holds exactly its slice
        for lo, hi, td, tl in w.blocks:
            assert len(tl) - 1 == hi - lo
            cover[w.q_lo:w.q_hi, lo:hi] += 1
        n_blocks.add(w.n_blocks_total)
    assert len(n_blocks) == 1                                        # every rank cut the database the same way (the reference's -b of the parity run)
    assert cover.min() == 1 and cover.max() == 1


def test_2d_needs_an_even_rank_count_of_at_least_four():
    with pytest.raises(AssertionError):
        bench.Workload("C2", 400, 120, 2, 0, "2d")
    with pytest.raises(AssertionError):
        bench.Workload("C2", 400, 120, 6 + 1, 0, "2d")
