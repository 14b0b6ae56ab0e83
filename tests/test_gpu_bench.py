"""-m gpu: bench.py itself, at reduced size: the JSON contract (metric/value/roofline/cpu_baseline on one line), parity_checked
against the reference binary inside the same run, the multi-block configuration (C5: database blocks processed one after the other
and joined as the reference joins reference blocks) and the two-rank database-sharded path (both ranks on the one GPU of the box,
records exchanged over gloo: DMND_BENCH_SHARE_GPU=1)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "diamond_tap")


def _bench(args, env=None, launcher=None):
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                     # ONE json line
    return json.loads(lines[0])


@pytest.mark.parametrize("cfg", ["C2", "C3", "C4", "C5"])
def test_bench_line_and_parity_at_reduced_size(cfg):
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond_tap is missing: under -m gpu the reference binary is the checker, its absence is a failure")
    d = _bench(["--config", cfg, "--queries", "1500", "--families", "3000", "--steps", "2", "--warmup", "1"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "GCUPS" and d["n_gpus"] == 1 and d["steps"] == 2 and d["vs_baseline"] is None and d["value"] > 0
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert d["cpu_baseline"]["kind"] == "reference" and d["cpu_baseline"]["value"] > 0
    assert d["parity_checked"] is True, d.get("parity")
    assert d["parity"]["lines"] > 1000
    # round 5: median over windows beside the mean, and (one block per rank) two database blocks alternating between the steps
    assert "ms_per_step_median" in d and "ms_per_step_windows" in d
    assert d["database_blocks_alternated"] is (cfg != "C5")
    if cfg == "C5":
        assert "8 blocks" in d["config"]["workload"]


def test_c5_full_size_parity():
    """BASELINE config C5 at its full size (100 000 queries, 5 000 000 sequences in 8 database blocks): the joined records of the
    timed run equal the reference binary's output for the same files and the same block size, line for line (md5), inside the
    bench run itself. The other full-size configurations are byte-compared in test_gpu_fullscale.py."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond_tap is missing: under -m gpu the reference binary is the checker, its absence is a failure")
    d = _bench(["--config", "C5", "--steps", "2", "--warmup", "1", "--no-e2e"])
    assert d["config"]["queries"] == 100000 and d["config"]["db_seqs"] == 5000000
    assert d["parity_checked"] is True, d.get("parity")
    assert d["parity"]["lines"] > 100000 and d["parity"]["records_md5"] == d["parity"]["reference_output_md5"]


def test_bench_two_ranks_database_sharded_on_one_gpu():
    """`python bench.py --gpus 2` with NO torchrun: bench.py launches its two ranks itself (both on the one GPU of the box:
    DMND_BENCH_SHARE_GPU=1 maps them to cuda:0 and runs the record exchange over gloo)."""
    one = _bench(["--queries", "1500", "--families", "3000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--queries", "1500", "--families", "3000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(env, DMND_BENCH_SHARE_GPU="1"), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert "all-to-all" in two["config"]["parallelism"]
    # the same fixed job: same seed hits in total; alignments differ only by the per-block culling of the reference's block join
    w1, w2 = one["config"]["workload"], two["config"]["workload"]
    assert w1.split("per step")[1].split("seed hits")[0] == w2.split("per step")[1].split("seed hits")[0]
    # one GPU but --gpus 2 without the sharing hook: refused, not silently run on fewer GPUs
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode != 0


def test_bench_four_ranks_two_query_halves_times_two_database_shards():
    """`--shard 2d` (round 6): rank r searches query half r % 2 against database shard r // 2; the records go through the same
    query-range exchange as the database-sharded run. Four ranks on the one GPU (gloo), the job's records against the reference run
    with the database cut into the same two blocks (`parity_checked`, made on rank 0 behind the teardown of the process group)."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond_tap is missing: under -m gpu the reference binary is the checker, its absence is a failure")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--shard", "2d", "--queries", "1500", "--families", "3000", "--steps", "2", "--warmup", "1",
           "--no-e2e", "--no-masked-step"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(env, DMND_BENCH_SHARE_GPU="1"), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["rccl"]["world_size"] == 4 and d["rccl"]["shard"] == "2d"
    assert d["parity_checked"] is True, d.get("parity")
    assert "2 blocks" in d["parity"]["note"]


def test_bench_e2e_and_hot_path_baselines():
    """The whole-process comparison (diamond-hip against the reference binary on the same files) and the hot-path baseline
    (the reference's seed-stage + extension task timers) are on the bench line, md5-equal outputs for all three command lines."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond_tap is missing: under -m gpu the reference binary is the checker, its absence is a failure")
    d = _bench(["--queries", "1500", "--families", "3000", "--steps", "2", "--warmup", "1", "--with-masking"])
    # --with-masking: the step of the default command line (masking on the device inside the step); its records = the reference's default output
    m = d["masked_step"]
    assert m["ms_per_step"] > 0 and m["stages_back_to_back"]["parts_ms"]["mask_target"] > 0 and m["masked_letters"]["database"] > 0
    assert m["records_equal_back_to_back"] is True          # the pipelined form and the stages one after the other: the same records
    assert m["parity"]["matches"] is True, m["parity"]
    assert d["cpu_baseline"]["hot_path"]["seconds"] > 0 and d["cpu_baseline"]["whole_process"]["seconds"] > d["cpu_baseline"]["hot_path"]["seconds"]
    e = d["e2e"]
    assert set(e["runs"]) == {"default_masking", "masking_off", "stock_command_line"}
    assert e["parity"] is True, {k: v["parity"] for k, v in e["runs"].items()}
    assert all(v["ours_s"] > 0 and v["reference_s"] > 0 for v in e["runs"].values())
    assert all(0 < v["speedup_min"] <= v["speedup"] for v in e["runs"].values()) and e["speedup_min"] > 0
