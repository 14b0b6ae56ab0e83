#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X seed-and-extend hot path on BASELINE.json's headline workload
(config C2: blastp --fast, 10k synthetic queries x 1M-sequence synthetic database, SURVEY.md 8d generator).

A "step" = one full pass of the hot path over the query block and the reference block, both resident in HBM before
the timed region: seed stage on the GPU (dmnd_seed_search: query seed table, one stream over the reference block,
complexity masks, Hamming + left-most filters -> stage-2 hits) and extension stage (dmnd_extend: Hauser bias on the GPU,
host x-drop/chaining, round-1 banded Smith-Waterman on the GPU in traceback mode with kept trace rows, e-value cutoff +
top-25 culling, round-2 walk of the kept traces on the GPU, final culling -> match records). It is the work `diamond
blastp --fast --algo 0 --masking 0 --motif-masking 0` does between "Building reference seed array" and the output
writer; results are byte-identical to the reference's (tests/test_gpu_extend.py).

Batches are pipelined (default; --no-pipeline runs them back to back): the seed stage of batch s+1 runs on a second
context (own low-priority stream) and the record gather of batch s-1 on a third thread while batch s is extended --
every batch still passes through the whole path inside the timed region.

metric  = GCUPS: DP cells per the reference's definition (DpTarget::cells, dp/dp.h:121-124, both swipe rounds of the
          reference on this workload -- the same count cpu_baseline uses) / wall seconds of the K timed steps; the cells
          the device actually sweeps (round-2 targets once, not twice) and aligned queries/s are reported beside it.
N > 1   : query sharding (SURVEY.md 8e option 1, bit-identical to one GPU): every rank holds a database block in HBM
          and processes its own 10k-query slice; no collective on the data path, one RCCL all_gather of the fixed-size
          per-query top-k records at the end of each step -> weak scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from diamond_amd import hip, multigpu, synth, workload  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def gather_topk(n_queries, matches, device):
    """Per-query top-k records of this rank's query slice, gathered from all ranks with ONE all_gather of a
    fixed-size tensor (diamond_amd/multigpu.py). Returns the number of aligned queries of the whole job."""
    rec = multigpu.topk_records(n_queries, matches["query"], matches["evalue"], matches["hsp"]["score"], matches["target"], presorted=True)
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
        r = subprocess.run([exe, "blastp", "--fast", "--algo", "0", "--masking", "0", "--motif-masking", "0", "-q", os.path.join(tmp, "q.faa"), "-d", os.path.join(tmp, "db"),
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
                "sample": "reference diamond v2.2.2 `blastp --fast --algo 0 --masking 0 --motif-masking 0 -p %d` on %d queries x %d seqs (%.0f%% of the workload), CPU %s: "
                          "%.2f s wall end-to-end, %d DpTargets / %d cells (both rounds), %s queries aligned, %.1f aligned queries/s; "
                          "SW stage alone: %.3f CPU-s => %.2f GCUPS per core"
                          % (cores, nq, nf * 10, frac * 100, model, wall, cells["targets"], cells["cells"],
                             aligned.group(1) if aligned else "?", (int(aligned.group(1)) / wall) if aligned else float("nan"),
                             sw_s, cells["cells"] / sw_s / 1e9)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--families", type=int, default=100_000)
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--cpu-sample-frac", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="run seed stage and extension stage of a batch back to back on one context")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    # test hook: several ranks on ONE GPU (RCCL refuses that), used to exercise the multi-rank code path on a 1-GPU box:
    # DMND_BENCH_SHARE_GPU=1 maps every rank to cuda:0 and gathers the top-k records over gloo instead of RCCL
    share_gpu = os.environ.get("DMND_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    coll_device = torch.device("cpu") if share_gpu else device
    assert world == args.gpus or world == 1
    threads = args.host_threads or max(1, min(64, (os.cpu_count() or 8) // max(world, 1)))

    # every rank: its own seeded query slice + database block (weak scaling: per-GPU work is fixed)
    db, doff, q, qoff = synth.generate(args.families, members=10, queries=args.queries, seed=20260923 + 1000 * rank)
    qd, ql = workload.sequence_set(q, qoff)
    td, tl = workload.sequence_set(db, doff)
    params = hip.default_params()
    params.db_letters = float(doff[-1])
    ctx = hip.Context(device=local_rank, params=params)
    torch.cuda.synchronize()
    t_up = time.perf_counter()
    ctx.upload_block(hip.QUERY, qd, ql)
    ctx.upload_block(hip.TARGET, td, tl)                    # synchronous H2D of both blocks (pageable host memory)
    upload_ms = (time.perf_counter() - t_up) * 1e3          # outside the timed region: inputs are resident when a step starts
    seed_params = hip.seed_params_fast(threads=8)
    state = {}
    # Batch pipeline (default): a second context holds the same two blocks and runs the seed stage of batch s+1 on its own
    # stream while this one extends batch s -- what a run over many query blocks does (every batch still passes through the
    # whole path inside the timed region; the blocks of both contexts are resident before it starts).
    pipeline = not args.no_pipeline
    ctx_seed = ctx
    if pipeline:
        ctx_seed = hip.Context(device=local_rank, params=params)
        ctx_seed.upload_block(hip.QUERY, qd, ql)
        ctx_seed.upload_block(hip.TARGET, td, tl)

    def step():
        t_a, c_a = time.perf_counter(), time.process_time()
        hits = ctx.seed_search(seed_params)
        t_b, c_b = time.perf_counter(), time.process_time()
        matches, _ = ctx.extend(qd, td, hits, threads=threads)
        t_c, c_c = time.perf_counter(), time.process_time()
        aligned = gather_topk(args.queries, matches, coll_device)
        t_d, c_d = time.perf_counter(), time.process_time()
        state.update(hits=int(hits.size), matches=int(matches.size), aligned=aligned, seed_ms=ctx.seed_kernel_ms(), ext=ctx.extend_stats(),
                     wall_ms={"seed_stage_call": (t_b - t_a) * 1e3, "extension_call": (t_c - t_b) * 1e3, "topk_gather": (t_d - t_c) * 1e3},
                     cpu_ms={"seed_stage_call": (c_b - c_a) * 1e3, "extension_call": (c_c - c_b) * 1e3, "topk_gather": (c_d - c_c) * 1e3})

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def seed_stage():
        torch.cuda.set_device(local_rank)
        hits = ctx_seed.seed_search(seed_params)
        return hits, ctx_seed.seed_kernel_ms()

    def run_pipelined(n_steps, primed=None):
        """n_steps batches in steady state: every step takes the seed hits of its batch (primed: computed by the previous step, or by
        the warm-up for the first timed step), starts the seed stage of the NEXT batch on a worker thread (ctypes releases the GIL),
        extends its own batch and hands the records of the batch to the gather thread. So a run of n steps executes n seed stages,
        n extensions and n gathers; the seed stage started by the last step is awaited before the clock stops.
        Returns (stream kernel ms summed over the seed stages started here, the outstanding seed stage)."""
        stream = 0.0
        fut = primed if primed is not None else seed_pool.submit(seed_stage)
        started = 0 if primed is not None else 1
        wall = [0.0, 0.0, 0.0]
        each = state.setdefault("each_ms", [])
        del each[:]
        gathered = None                                   # the record gather of batch s-1 runs on its own thread during batch s

        def gather_stage(matches):
            torch.cuda.set_device(local_rank)
            return gather_topk(args.queries, matches, coll_device)

        for s in range(n_steps):
            t_a = time.perf_counter()
            hits, seed_ms = fut.result()
            if not (s == 0 and primed is not None):
                stream += seed_ms[1]                      # a seed stage that ran inside this call
            if started < n_steps:
                fut = seed_pool.submit(seed_stage)
                started += 1
            else:
                fut = None
            t_b = time.perf_counter()
            matches, _ = ctx.extend(qd, td, hits, threads=threads)
            t_c = time.perf_counter()
            if gathered is not None:
                state["aligned"] = gathered.result()
            gathered = gather_pool.submit(gather_stage, matches)
            t_d = time.perf_counter()
            for i, x in enumerate((t_b - t_a, t_c - t_b, t_d - t_c)):
                wall[i] += x * 1e3 / n_steps
            each.append(round((t_d - t_a) * 1e3, 2))
            if s == 0:
                state["first_step_ms"] = {"wait_for_seed_stage": (t_b - t_a) * 1e3, "extension_call": (t_c - t_b) * 1e3, "wait_for_previous_gather": (t_d - t_c) * 1e3}
            state.update(hits=int(hits.size), matches=int(matches.size), seed_ms=seed_ms, ext=ctx.extend_stats(),
                         pipe_wall_ms={"wait_for_seed_stage": wall[0], "extension_call": wall[1], "wait_for_previous_gather": wall[2]})
        state["aligned"] = gathered.result()              # the last batch's records are gathered inside the timed region too
        if fut is not None:
            stream += fut.result()[1][1]                  # ... and so is the seed stage the last step started
        return stream, fut

    if pipeline:
        import concurrent.futures
        seed_pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)
        gather_pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)
    for _ in range(args.warmup):
        step()
    primed = None
    if pipeline and args.warmup:
        run_pipelined(2)
        primed = seed_pool.submit(seed_stage)                # fills the pipeline: the first timed step finds its seed hits ready,
        primed.result()                                      # computed before the clock starts (the last timed step computes a batch ahead)
    sync()
    # hipDeviceSynchronize lets the runtime release the hardware queues of idle streams; re-acquiring those of the extension
    # runners would cost the first timed step ~6.5 ms (measured; a streaming caller never synchronizes the whole device)
    ctx.touch_streams()
    if pipeline:
        ctx_seed.touch_streams()
    t0 = time.perf_counter()
    cpu0 = time.process_time()
    stream_ms = 0.0
    if pipeline:
        stream_ms, _ = run_pipelined(args.steps, primed)
    else:
        for _ in range(args.steps):
            step()
            stream_ms += state["seed_ms"][1]
    sync()
    dt = time.perf_counter() - t0
    cpu_ms_per_step = (time.process_time() - cpu0) * 1e3 / args.steps      # CPU time of all threads of this process
    if pipeline:                                             # stage latencies of one batch on an otherwise idle GPU, after the timed region
        state["pipe_ext"] = dict(state["ext"])
        serial = []
        for _ in range(3):
            t_s = time.perf_counter()
            step()
            serial.append((time.perf_counter() - t_s) * 1e3)
        state["serial_ms"] = min(serial)
        state["serial_stream_ms"] = state["seed_ms"][1]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ext = state["ext"]
    # GCUPS numerator = the DP cells of the reference's two swipe rounds on this workload (DpTarget::cells of every round-1 and
    # round-2 target: what the reference computes and what cpu_baseline counts). The device sweeps the round-2 targets only once:
    # round 1 keeps its trace rows and round 2 walks them (round2_swipe_kernel_ms == 0), reported as cells_swept_per_step.
    cells_step = ext["round1_cells"] + ext["round2_cells"]
    cells_swept = ext["round1_cells"] + (ext["round2_cells"] if ext["round2_swipe_kernel_ms"] > 0 else 0.0)
    gcups = cells_step * world * args.steps / dt / 1e9      # every rank runs a slice of the same shape (weak scaling)

    if rank == 0:
        # dominant kernel = the reference stream of the seed stage (seed_stream_fast_kernel), one launch per step.
        # ALGORITHMIC bytes per launch = SURVEY.md 8(d)'s per-unit figure x the units of one launch: the reference side of
        # bytes_seed = S (L 1 + N 8 2 + ...) is 1 B (residue read once) + 16 B (one 8-byte (key32, loc32) seed entry written and
        # read back) per reference letter, N = L for the reference block. Our formulation never materialises the entries: it
        # needs 1 B per letter, reported beside it as design_bytes / frac_design_bytes.
        k_ms = stream_ms / args.steps
        ref_letters = int(tl[-1] - tl[0])
        alg_bytes = 17 * ref_letters
        design_bytes = ref_letters
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        achieved_design = design_bytes / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")      # rocprofv3 --pmc passes of this command (tools/profile_round.sh)
        if os.path.exists(tpath) and args.queries == 10_000 and args.families == 100_000:
            pmc = json.load(open(tpath))
            k = [v for name, v in pmc.items() if "seed_stream_fast_kernel" in name]
            if k:
                # HBM-side bytes per launch: FETCH_SIZE (x2: gfx950 under-reports wide reads, upper bound) + WRITE_SIZE
                traffic = k[0]["FETCH_SIZE_x2_bytes_per_launch"] + k[0]["WRITE_SIZE_bytes_per_launch"]
        # achievable HBM bandwidth on this device: device-to-device copy of 1 GiB (bytes read + written per second)
        buf_a = torch.empty(1 << 30, dtype=torch.uint8, device=device)
        buf_b = torch.empty_like(buf_a)
        buf_b.copy_(buf_a)
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        for _ in range(5):
            buf_b.copy_(buf_a)
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * (1 << 30) / (time.perf_counter() - t_c) / 1e9
        del buf_a, buf_b
        out = {
            "metric": "GCUPS + aligned queries/s, blastp --fast 10k queries vs 1M-seq DB (seed stage + banded SW extension)",
            "value": gcups, "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "aligned_queries_per_s": state["aligned"] * args.steps / dt,
            "config": {"workload": "C2: blastp --fast --algo 0 (no masking), %d queries x %d-seq DB (%d letters)%s; per step %d seed hits, "
                                   "%d round-1 + %d round-2 DpTargets, %d alignments, %d queries aligned"
                                   % (args.queries, args.families * 10, int(doff[-1]), " per GPU" if world > 1 else "", state["hits"],
                                      int(ext["round1_targets"]), int(ext["round2_targets"]), state["matches"], state["aligned"] // world),
                       "queries": args.queries, "db_seqs": args.families * 10, "db_letters": int(doff[-1]), "cells_per_step": cells_step,
                       "cells_swept_per_step": cells_swept, "gcups_on_cells_swept": cells_swept * world * args.steps / dt / 1e9,
                       "cells_note": "value counts the DP cells of the reference's round 1 + round 2 (same definition as cpu_baseline); the device "
                                     "sweeps round-2 targets once (round 1 keeps the trace, round 2 walks it), cells_swept_per_step is what it computes",
                       "host_threads": threads,
                       "parallelism": "query-shard x%d + RCCL all_gather of top-k records" % world if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "seed_stream_fast_kernel (reference block streamed once against the query seed table)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "measured_copy_gbs": copy_gbs,
                         "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_per_unit": 17, "units_per_launch": ref_letters,
                         "kernel_ms": k_ms,
                         # with the batch pipeline the low-priority stream kernel shares the CUs with the previous batch's swipe kernels
                         # inside the timed region; alone (the serial steps after it) it takes kernel_ms_alone
                         "kernel_ms_alone": state.get("serial_stream_ms"),
                         "frac_alone": (alg_bytes / (state["serial_stream_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if state.get("serial_stream_ms") else None,
                         "design_bytes_per_launch": design_bytes, "achieved_design_bytes": achieved_design,
                         "frac_design_bytes": achieved_design / HBM_PEAK_GBS,
                         "note": "achieved = SURVEY 8(d) algorithmic bytes of the join's reference side (17 B per reference letter: residue + "
                                 "one 8-byte seed entry written and read back) / kernel time. The kernel itself reads every letter once "
                                 "(design_bytes = 1 B per letter) and probes a query-side table instead of materialising reference seed "
                                 "entries, so its measured HBM traffic is below the algorithmic bytes; it is bound by one L2 request per "
                                 "reference position (PMC: ~1 TCC request per letter), not by HBM bytes; see DESIGN.md 5"},
            "seed_kernel_ms": dict(zip(["index_queries", "stream_reference", "mask_groups", "pair_filter", "total"], state["seed_ms"])),
            "extension": ext,
            # device time summed over the concurrent runners' launches (they overlap on the GPU: a lower bound of the kernel rate)
            "swipe_kernel_gcups": {"round1": ext["round1_cells"] / max(ext["round1_swipe_kernel_ms"], 1e-9) / 1e6,
                                   "round2": (ext["round2_cells"] / ext["round2_swipe_kernel_ms"] / 1e6) if ext["round2_swipe_kernel_ms"] > 0 else None},
            # seed-stage Gletters/s = (L_q + L_r) x shapes / seed-stage seconds (device time of its kernels)
            "seed_stage_gletters_per_s": (int(ql[-1] - ql[0]) + int(tl[-1] - tl[0])) * seed_params.n_shapes / max(state["seed_ms"][4], 1e-9) / 1e6,
            "wall_ms_last_step": state["wall_ms"],
            "host_cpu_ms_last_step": state.get("cpu_ms"),
            "pipeline": ("seed stage of batch s+1 (second context, own stream) and the record gather of batch s-1 overlap the extension stage of batch s; latency of one batch alone "
                         "%.2f ms, its stream kernel alone %.3f ms" % (state["serial_ms"], state["serial_stream_ms"])) if pipeline else "off",
            "pipeline_wall_ms_per_step": state.get("pipe_wall_ms"),
            "pipeline_extension_last_step": state.get("pipe_ext"),
            "ms_each_step": state.get("each_ms"),
            "first_step_ms": state.get("first_step_ms"),
            "host_cpu_ms_per_step": cpu_ms_per_step,
            # not part of `value`: one-time PCIe upload of both blocks, and the rate if it were paid on every step
            "block_upload_ms": upload_ms,
            "pcie_inclusive_gcups": cells_step * world / ((dt / args.steps + upload_ms * 1e-3)) / 1e9,
        }
        if not args.no_cpu_baseline and world == 1:          # reported on rank 0 at N=1 only
            ref = cpu_baseline_reference(args)
            if ref is not None:
                out["cpu_baseline"] = ref
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
