#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X hot path on BASELINE.json's headline workload.

A "step" = one pass of the extension stage over the C2-shaped synthetic batch
(diamond_amd/workload.py: blastp --fast, 10k queries x 1M-sequence database): round 1 = score-only
banded Smith-Waterman over every (query, target, band) work item, e-value cutoff + top-25 culling on the
host, round 2 = banded Smith-Waterman with traceback over the survivors. Sequence blocks are resident in
HBM before the timed region; work-item descriptors and results cross PCIe inside it.

metric  = GCUPS (DP cells per the reference's definition DpTarget::cells, dp/dp.h:121-124, both rounds)
          / wall seconds of the K timed steps; aligned queries/s is reported beside it.
N > 1   : query sharding (SURVEY.md 8e option 1, bit-identical to one GPU): every rank holds its database
          block in HBM and extends its own 10k-query slice; no collective on the data path, one RCCL
          all_gather of the fixed-size per-query top-k records at the end of each step -> weak scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from diamond_amd import hip, multigpu, workload  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def gather_topk(w, params, hsps, sel, world, rank, device):
    """Per-query top-k records of this rank's query slice, gathered from all ranks with ONE all_gather of a
    fixed-size tensor (diamond_amd/multigpu.py). Returns the number of aligned queries of the whole job."""
    ev = hip.evalue_batch(params, hsps["score"], w.items["query_len"][sel], w.items["target_len"][sel]) if sel.size else np.zeros(0)
    rec = multigpu.topk_records(w.n_queries, w.qi[sel], ev, hsps["score"], w.ti[sel])
    return multigpu.aligned_queries(multigpu.gather_records(rec, device))


def cpu_baseline_reference(args):
    """The GENUINE reference (oracle/_ref/diamond_tap: /root/reference compiled in place; the tap only counts
    DP cells) timed on this box's host cores on a bounded sample of the same synthetic workload:
    `blastp --fast --algo 0` on frac x (queries, database). Returns None if the prebuilt binary is absent."""
    import re
    import shutil
    import subprocess
    import tempfile
    from diamond_amd import synth
    exe = os.path.join(ROOT, "oracle", "_ref", "diamond_tap")
    if not os.path.exists(exe):
        return None
    frac = args.cpu_sample_frac
    nq, nf = max(int(args.queries * frac), 100), max(int(args.families * frac), 1000)
    cores = os.cpu_count() or 1
    tmp = tempfile.mkdtemp(prefix="dmnd_cpu_")
    try:
        db, doff, q, qoff = synth.generate(nf, members=10, queries=nq, seed=20260923)
        synth.write_fasta(os.path.join(tmp, "db.faa"), "t", db, doff)
        synth.write_fasta(os.path.join(tmp, "q.faa"), "q", q, qoff)
        subprocess.run([exe, "makedb", "--in", os.path.join(tmp, "db.faa"), "-d", os.path.join(tmp, "db")],
                       check=True, capture_output=True, timeout=600)
        env = dict(os.environ, DIAMOND_TAP_CELLS=os.path.join(tmp, "cells.json"))
        t0 = time.perf_counter()
        r = subprocess.run([exe, "blastp", "--fast", "--algo", "0", "-q", os.path.join(tmp, "q.faa"), "-d", os.path.join(tmp, "db"),
                            "-o", os.path.join(tmp, "out.tsv"), "-p", str(cores), "--log"], check=True, capture_output=True,
                           text=True, env=env, timeout=1200)
        wall = time.perf_counter() - t0
        cells = json.load(open(os.path.join(tmp, "cells.json")))
        log = r.stdout + r.stderr
        sw = re.search(r"Time \(Smith Waterman\)\s*= ([0-9.eE+-]+)s", log)
        aligned = re.search(r"(\d+) queries aligned", log)
        sw_s = float(sw.group(1)) if sw else float("nan")
        model = open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t") if os.path.exists("/proc/cpuinfo") else "?"
        return {"value": cells["cells"] / wall / 1e9, "unit": "GCUPS", "cores": cores, "kind": "reference",
                "sample": "reference diamond v2.2.2 `blastp --fast --algo 0 -p %d` on %d queries x %d seqs (%.0f%% of the workload), CPU %s: "
                          "%.2f s wall end-to-end, %d DpTargets / %d cells (both rounds), %s queries aligned, %.1f aligned queries/s; "
                          "SW stage alone: %.3f CPU-s => %.2f GCUPS per core"
                          % (cores, nq, nf * 10, frac * 100, model, wall, cells["targets"], cells["cells"],
                             aligned.group(1) if aligned else "?", (int(aligned.group(1)) / wall) if aligned else float("nan"),
                             sw_s, cells["cells"] / sw_s / 1e9)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(w, params, n_items):
    """The CPU port (oracle/banded_swipe.c, a scalar restatement of the reference's banded SWIPE) timed on a
    bounded sample of the same round-1 work items on one host core. Baseline for context, not a target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as orc
    rng = np.random.default_rng(0)
    idx = rng.choice(w.items.size, min(n_items, w.items.size), replace=False)
    M = hip.matrix_of(params)
    cells = int(workload.Workload.cells(w.items[idx]).sum())
    t0 = time.perf_counter()
    for k in idx:
        it = w.items[k]
        orc.banded_swipe(w.q[it["query_off"]: it["query_off"] + it["query_len"]], None,
                         w.db[it["target_off"]: it["target_off"] + it["target_len"]],
                         it["d_begin"], it["d_end"], M, params.gap_open, params.gap_extend, orc.SCORE_ONLY)
    dt = time.perf_counter() - t0
    return {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": 1, "kind": "port",
            "sample": "%d of the %d round-1 DpTargets (score-only banded SW, oracle/banded_swipe.c), %.1f s" % (idx.size, w.items.size, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--families", type=int, default=100_000)
    ap.add_argument("--cpu-items", type=int, default=6000)
    ap.add_argument("--cpu-sample-frac", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    assert world == args.gpus or world == 1

    # every rank: its own seeded query slice + database block (weak scaling: per-GPU work is fixed)
    w = workload.Workload(families=args.families, members=10, queries=args.queries, seed=20260923 + 1000 * rank)
    params = hip.default_params()
    params.db_letters = float(w.db_letters)
    ctx = hip.Context(device=local_rank, params=params)
    ctx.upload_block(hip.QUERY, w.q)
    ctx.upload_block(hip.TARGET, w.db)
    ctx.upload_cbs(np.zeros(0, np.int8))

    cells1 = int(workload.Workload.cells(w.items).sum())
    state = {}

    def step():
        t_a = time.perf_counter()
        r1, _ = ctx.banded_swipe(w.items, hip.SWIPE_SCORE)
        t_b = time.perf_counter()
        ms1 = ctx.last_kernel_ms()[0]
        sel = w.select_round2(params, r1["score"])
        t_c = time.perf_counter()
        r2, tr = ctx.banded_swipe(w.items[sel], hip.SWIPE_TRACEBACK, 510)
        t_d = time.perf_counter()
        ms2, mstb = ctx.last_kernel_ms()
        aligned = gather_topk(w, params, r2, sel, world, rank, device)
        t_e = time.perf_counter()
        state.update(sel=sel, ms1=ms1, ms2=ms2, mstb=mstb, aligned=aligned, n2=sel.size,
                     wall_ms={"round1_call": (t_b - t_a) * 1e3, "culling": (t_c - t_b) * 1e3, "round2_call": (t_d - t_c) * 1e3,
                              "topk_gather": (t_e - t_d) * 1e3})

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    ms1_sum = 0.0
    for _ in range(args.steps):
        step()
        ms1_sum += state["ms1"]
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    cells2 = int(workload.Workload.cells(w.items[state["sel"]]).sum())
    cells_step = cells1 + cells2
    total_cells = cells_step * world       # approximately: every rank has its own seeded slice of the same shape
    gcups = total_cells * args.steps / dt / 1e9

    if rank == 0:
        alg_bytes = workload.Workload.algorithmic_bytes(w.items)
        # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of this same
        # command, summarised by tools/ into profiles/); only valid for the default workload size
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath) and args.queries == 10_000 and args.families == 100_000:
            traffic = json.load(open(tpath))["round1_score_kernels"]["traffic_bytes_fetch_x2"]
        k_ms = ms1_sum / args.steps
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "GCUPS (banded SW extension, blastp --fast 10k queries vs 1M-seq DB shape)",
            "value": gcups, "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "aligned_queries_per_s": state["aligned"] * args.steps / dt,
            "config": {"workload": "C2-shaped extension stage: blastp --fast, %d queries x %d-seq DB%s; round 1 %d DpTargets score-only + "
                                   "round 2 %d DpTargets traceback (top-25, e<=1e-3); band geometry from the workload model, "
                                   "seed stage not yet in the timed path" % (args.queries, args.families * 10,
                                                                             " per GPU" if world > 1 else "", w.items.size, state["n2"]),
                       "queries": args.queries, "db_seqs": args.families * 10, "db_letters": w.db_letters,
                       "cells_per_step": cells_step, "parallelism": "query-shard x%d + RCCL all_gather of top-k records" % world if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "banded_swipe_kernel<P,score-only> (round 1)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": k_ms,
                         "kernel_gcups": cells1 / (k_ms * 1e-3) / 1e9,
                         "note": "integer DP held in VGPRs: VALU-issue bound, not HBM-bound (SURVEY.md 8d); HBM fraction reported as the contract asks"},
            "kernel_ms": {"round1_swipe": state["ms1"], "round2_swipe": state["ms2"], "round2_traceback": state["mstb"]},
            "wall_ms_last_step": state["wall_ms"],
        }
        if not args.no_cpu_baseline:
            ref = cpu_baseline_reference(args)
            port = cpu_baseline(w, params, args.cpu_items)
            out["cpu_baseline"] = ref if ref is not None else port
            out["cpu_baseline_port"] = port
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
