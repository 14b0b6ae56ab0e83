#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X seed-and-extend hot path on BASELINE.json's workloads.

  --config C2 (default)  blastp --fast,      10k synthetic queries x 1M-sequence synthetic database (the headline config)
  --config C3            blastp --sensitive, same blocks (16 shapes, ungapped + gapped filters)
  --config C4            blastx, 5k synthetic DNA reads of ~1 kb (six frames each) x the same database, default sensitivity
  --config C5            blastp --fast, 100k queries x 5M sequences in 8 database blocks (BASELINE config 5)
  --config C2skew        blastp --fast, 10k queries x 1M sequences in 1000 families of 1000 (the seed stage's overflow / tiled paths)

A "step" = one full pass of the hot path over the query block and the reference block, both resident in HBM before the timed
region: seed stage on the GPU (dmnd_seed_search) and extension stage (dmnd_extend: Hauser bias on the GPU, host chaining in
fixed per-thread query slices, ONE round-1 banded Smith-Waterman launch per band class in traceback mode with kept trace rows,
e-value cutoff + top-25 culling on the host, ONE walk of the kept traces on the GPU, final culling -> match records). It is the
work `diamond blastp --algo 0 --masking 0 --motif-masking 0` does between "Building reference seed array" and the output
writer; the records are byte-identical to the reference's (tests/test_gpu_fullscale.py at this very size; `parity_checked`
below compares this run's records with the reference output produced for `cpu_baseline`).

Batches are pipelined (default; --no-pipeline runs them back to back): the seed stage of batch s+1 runs on a second context
(own low-priority stream) while batch s is extended -- every batch still passes through the whole path in the timed region.
Since round 5 consecutive steps search DIFFERENT memory (N = 1, one block per rank): the config's database block and a second one
holding the same sequences in reverse order alternate, so that no step finds its letters in the 256 MiB Infinity Cache from the step
before (--same-block: the old behaviour); `ms_per_step` is the mean over the timed region, `ms_per_step_median` the median over
windows of steps. After the timed steps the step of the DEFAULT command line (tantan + motif masking of both blocks inside the step)
is timed too and reported as `masked_step`, with its own parity check (--no-masked-step skips it).

value   = GCUPS on the DP cells the device SWEEPS (DpTarget::cells, dp/dp.h:121-124, of every round-1 target; round 2 walks
          the kept traces and sweeps nothing) / wall seconds of the K timed steps, whole job. The reference's own count (both
          rounds swept: what cpu_baseline is quoted on) is reported beside it as `reference_equivalent_gcups`.
N > 1   : STRONG scaling of the fixed job, database-sharded (SURVEY.md 8e option 2, what BASELINE config C5 names): rank g holds
          1/N of the reference block and all queries; e-values against the whole database; per step the ranks' match records go to
          the owners of their query ranges in one all-to-all over RCCL, device memory to device memory, are merged there on the device
          (dmnd_join_blocks_device = the reference's block join) and gathered on rank 0 (multigpu.query_range_join_device; --host-join:
          the host merge of rounds 3-4). --shard query: every rank holds the whole block and 1/N of the queries, no collective on the
          data path (records are concatenated).

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from diamond_amd import hip, multigpu, synth, workload  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CONFIGS = {
    "C2": dict(mode="blastp", sens="fast", flags=["--fast"], what="blastp --fast"),
    "C3": dict(mode="blastp", sens="sensitive", flags=["--sensitive"], what="blastp --sensitive"),
    "C4": dict(mode="blastx", sens="default", flags=[], what="blastx (default sensitivity), 5k reads of ~1 kb, six frames"),
    # BASELINE config 5: 100k queries x 5M sequences cut into 8 database blocks; N GPUs take 8/N blocks each (one after the other
    # on a rank), the records of all blocks are joined as the reference joins reference blocks
    "C5": dict(mode="blastp", sens="fast", flags=["--fast"], what="blastp --fast, database in 8 blocks", blocks=8, queries=100_000, families=500_000),
    # round 5: the same sizes as C2, but 1000 families of 1000 members -- a query's seeds join thousands of reference positions, the
    # joined-position lists outgrow their first buffer, the pair filter runs in its sorted, LDS-tiled form (tests/test_gpu_skew.py)
    "C2skew": dict(mode="blastp", sens="fast", flags=["--fast"], what="blastp --fast, 1000 families of 1000 members", families=1000, members=1000),
}


def name_thread(name):
    """OS name of the calling thread (what /proc/<pid>/task/<tid>/comm shows): host_cpu_by_thread below groups by it"""
    try:
        import ctypes
        ctypes.CDLL(None).prctl(15, name.encode()[:15], 0, 0, 0)      # PR_SET_NAME
    except Exception:
        pass


def thread_cpu_ms():
    """CPU time (user + system, ms) of every thread of this process by OS thread name"""
    out, tick = {}, os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            st = open("/proc/self/task/%s/stat" % tid).read()
        except OSError:
            continue
        comm = st[st.index("(") + 1:st.rindex(")")]
        f = st[st.rindex(")") + 2:].split()
        out[comm] = out.get(comm, 0.0) + (int(f[11]) + int(f[12])) * 1e3 / tick
    return out


def scaling_model(out, ext, cpu_ms_per_step, cpus, step_ms, n_records, n_blocks, ext_contexts, threads):
    """ONE formula for every config and decomposition, evaluated from this N = 1 run; the first SCALE file is a check of it.
    A rank's step = max(GPU term, host term) + exchange:
      GPU term  = overlap x (index + stream + pairs + extension kernels of the rank's share); overlap = this run's step / the same sum at
                  N = 1 (how much of the kernel time the pipeline hides or the launch gaps add), kept for every N
      host term = host CPU-ms per step of the whole job / CPUs of the box (the cgroup quota all ranks share)
      exchange  = 2 x 25 us per collective + this rank's records at 50 GB/s per xGMI link
    Decompositions (the reference's own multi-process mode hands out (query chunk x reference chunk) units,
    /root/reference/src/run/double_indexed.cpp:346-396):
      db     every rank: ALL queries x 1/N of the database -- index whole, stream / pairs / extension by N, records exchanged by query range
      query  every rank: 1/N of the queries x ALL the database -- index and pairs and extension by N, the per-letter stream whole, one gather
      2xN/2  two query halves x N/2 database shards -- index by 2, stream by N/2, pairs and extension by N, exchange inside a half
    """
    sk = out["seed_kernel_ms"]
    index = sk["index_queries"] + sk["mask_groups"]
    stream = sk["stream_reference"]
    pairs = max(sk["total"] - index - stream, 0.0)
    ext_k = ext["round1_swipe_kernel_ms"] + ext["round2_swipe_kernel_ms"] + ext["traceback_kernel_ms"]
    gpu1 = index + stream + pairs + ext_k
    overlap = step_ms / max(gpu1, 1e-9)
    host = cpu_ms_per_step / max(cpus, 1)

    def coll(n, kind):
        if n == 1:
            return 0.0
        payload = 104.0 * n_records / n
        return {"db": 4 * 0.025 + 2 * payload / 50e9 * 1e3, "query": 2 * 0.025 + payload / 50e9 * 1e3, "2d": 4 * 0.025 + 2 * payload / 50e9 * 1e3}[kind]

    def gpu(n, kind):
        if kind == "db":
            return index + (stream + pairs + ext_k) / n
        if kind == "query":
            return index / n + stream + (pairs + ext_k) / n
        return index / 2 + stream / max(n // 2, 1) + (pairs + ext_k) / n          # 2 x n/2

    model = {"what": "predicted ms per step on N GPUs = max(overlap x GPU kernels of a rank's share, host CPU-ms per step / CPUs of the box) + exchange; one formula for "
                     "every config, evaluated for three decompositions from this N = 1 run -- to be checked against SCALE",
             "measured_ms_per_step": step_ms, "kernel_ms_per_step": {"index_queries": index, "stream_reference": stream, "pair_filter_and_stage2": pairs, "extension": ext_k},
             "overlap_factor": overlap, "host_cpu_ms_per_step": cpu_ms_per_step, "host_cpus": cpus, "host_floor_ms": host, "database_blocks": n_blocks,
             "decompositions": {}}
    best = {}
    for kind, label in (("db", "db"), ("query", "query"), ("2d", "2xN/2")):
        pred = {str(n): max(overlap * gpu(n, kind), host if n > 1 else 0.0) + coll(n, kind) for n in (1, 2, 4, 8) if kind != "2d" or n >= 4}
        model["decompositions"][label] = {"predicted_ms_per_step": pred, "predicted_speedup": {n: step_ms / v for n, v in pred.items() if n != "1"},
                                          "bound_at_8": "host" if host > overlap * gpu(8, kind) else "gpu"}
        for n, v in pred.items():
            if n != "1" and (n not in best or v < best[n][1]):
                best[n] = (label, v)
    model["best"] = {n: {"decomposition": k, "predicted_ms_per_step": v, "predicted_speedup": step_ms / v} for n, (k, v) in best.items()}
    model["predicted_speedup"] = {n: step_ms / v for n, (k, v) in best.items()}
    model["note"] = ("host_cpu_ms_per_step is this run's (%d extension contexts, %d host threads); `bench.py --gpus N` shards by --shard (default db: the only one of the "
                     "three a rank count below 4 allows besides query); the default at N > 1 is this model's best decomposition of the config's committed N = 1 line)" % (ext_contexts, threads))
    return model


def sweep_roofline(config, ext):
    """north_star's second kernel, the banded Smith-Waterman sweep (/root/reference/src/dp/swipe/banded_swipe.h:189-351), against the
    roofline that bounds it: VALU issue. Per kernel: VALU wave-instructions per launch (committed PMC pass) / its average launch
    time (committed rocprofv3 kernel statistics of the same configuration) / the device's issue peak: 256 CUs x 4 SIMDs x 2.4 GHz / 2
    = 1229 G wave-instructions/s (a 64-lane wavefront's VALU instruction occupies its SIMD for two cycles) -- `frac`. That peak holds
    for the plain 32-bit VOP2 operations only: tools/probes/valu_probe.hip (profiles/r06_valu_probe.txt) measures 2.4 cycles per
    instruction and SIMD for v_add_u32 / v_and_b32 and 4.2-4.5 for the operations these sweeps are made of (v_pk_*_i16, v_perm_b32,
    DPP moves, v_max, v_lshl_or_b32: 550-590 G/s), so `frac_of_packed_issue_peak` (achieved / 565) is the distance to what this
    instruction mix can issue. Live from this run: the sweeps' device time and cells, and the lane use of the device path's round-1
    DpTargets (band diagonals / the 2 P x lanes diagonals their wavefront or DPP row holds, weighted by steps)."""
    import csv
    PEAK, PACKED_PEAK = 256 * 4 * 2.4 / 2, 565.0
    o = {"bound": "valu_issue", "peak": PEAK, "packed_issue_peak": PACKED_PEAK, "unit": "G wave-instructions/s", "kernels": {}, "achieved": None, "frac": None,
         "frac_of_packed_issue_peak": None,
         "lane_use": (ext["band_diagonal_steps"] / ext["wavefront_diagonal_steps"]) if ext.get("wavefront_diagonal_steps") else None,
         "live": {"round1_sweep_kernel_ms": ext["round1_swipe_kernel_ms"], "round2_sweep_kernel_ms": ext["round2_swipe_kernel_ms"],
                  "round1_gcups": ext["round1_cells"] / max(ext["round1_swipe_kernel_ms"], 1e-9) / 1e6}}
    pmc = next((q for q in (os.path.join(ROOT, "profiles", "r%02d_pmc_summary_%s.json" % (r, config)) for r in (6, 5)) if os.path.exists(q)), None)
    st = next((q for q in (os.path.join(ROOT, "profiles", "r%02d_kernel_stats_%s.csv" % (r, config)) for r in (6, 5)) if os.path.exists(q)), None)
    if not pmc or not st:
        o["source"] = "no committed PMC pass + kernel statistics for this configuration"
        return o
    avg_ns = {r["Name"]: (float(r["AverageNs"]), int(r["TotalDurationNs"])) for r in csv.DictReader(open(st)) if "banded_swipe" in r["Name"]}
    best = None
    for name, v in json.load(open(pmc)).items():
        if "banded_swipe" not in name or "SQ_INSTS_VALU_per_launch" not in v or name not in avg_ns:
            continue
        ach = v["SQ_INSTS_VALU_per_launch"] / avg_ns[name][0]          # wave-instructions per ns = G/s
        short = name[name.index("banded_swipe"):name.index("(")] if "(" in name else name
        o["kernels"][short] = {"valu_wave_instructions_per_launch": v["SQ_INSTS_VALU_per_launch"], "avg_launch_ms": avg_ns[name][0] / 1e6,
                               "share_of_sweep_time": None, "achieved": ach, "frac": ach / PEAK, "frac_of_packed_issue_peak": ach / PACKED_PEAK, "_total_ns": avg_ns[name][1]}
        if best is None or avg_ns[name][1] > best[1]:
            best = (short, avg_ns[name][1])
    total = sum(k["_total_ns"] for k in o["kernels"].values()) or 1
    for k in o["kernels"].values():
        k["share_of_sweep_time"] = k.pop("_total_ns") / total
    if best:
        o["dominant"] = best[0]
        o["achieved"], o["frac"] = o["kernels"][best[0]]["achieved"], o["kernels"][best[0]]["frac"]
        o["frac_of_packed_issue_peak"] = o["kernels"][best[0]]["frac_of_packed_issue_peak"]
    o["source"] = "%s (SQ_INSTS_VALU per launch) / %s (average launch time)" % (os.path.relpath(pmc, ROOT), os.path.relpath(st, ROOT))
    return o


def cgroup_cpus():
    """CPUs of time this process may use: the cgroup quota when there is one (the GPU boxes run the container under
    cpu.max = 16 CPUs for 256 hardware threads), else the visible cores."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    return os.cpu_count() or 1


class Workload:
    """Synthetic blocks of one config (SURVEY.md 8d generator) and, for N > 1, this rank's shard of them."""

    def __init__(self, cfg, families, queries, world, rank, shard):
        self.cfg = CONFIGS[cfg]
        self.db, self.doff, self.q, self.qoff = synth.generate(families, members=self.cfg.get("members", 10), queries=queries, seed=20260923)
        self.n_db, self.db_letters = len(self.doff) - 1, int(self.doff[-1])
        self.source_lens = None
        if self.cfg["mode"] == "blastx":
            n_reads = min(queries, 5000)
            self.dna, self.dna_off = synth.back_translate(self.q[:self.qoff[n_reads]], self.qoff[:n_reads + 1], seed=5)
            self.n_queries = n_reads
            self.source_lens = np.diff(self.dna_off)
        else:
            self.n_queries = queries
        # this rank's part of the job: its database blocks (one, unless the config names a block count) or its query slice
        # "2d" (round 6): two query halves x world / 2 database shards -- rank r searches half r % 2 of the queries against shard r // 2
        # (the reference's own multi-process mode hands out (query chunk x reference chunk) units: run/double_indexed.cpp:346-396)
        self.q_lo, self.q_hi = 0, self.n_queries
        db_ranks, db_rank = (world, rank) if shard == "db" else (world // 2, rank // 2) if shard == "2d" else (1, 0)
        n_blocks = max(self.cfg.get("blocks", 1), db_ranks) if shard in ("db", "2d") else 1
        if world > 1 and shard == "query":
            self.q_lo, self.q_hi = multigpu.shard_range(self.n_queries, world, rank)
        if world > 1 and shard == "2d":
            assert world % 2 == 0 and world >= 4, "--shard 2d needs an even number of at least 4 ranks"
            self.q_lo, self.q_hi = multigpu.shard_range(self.n_queries, 2, rank % 2)
        self.blocks = []                                     # (first sequence, one past the last, letters, limits) of every block of this rank
        # cut as the reference cuts reference blocks (SequenceFile::load_seqs: a block ends with the sequence that reaches the
        # block size), block size = total letters / blocks: the job equals the reference run with that -b
        lens = np.diff(self.doff)
        self.block_letters = -(-self.db_letters // n_blocks)
        cuts, acc = [0], 0
        if n_blocks > 1:
            csum = np.cumsum(lens)
            while cuts[-1] < self.n_db:
                nxt = int(np.searchsorted(csum, acc + self.block_letters, side="left")) + 1
                nxt = min(max(nxt, cuts[-1] + 1), self.n_db)
                cuts.append(nxt)
                acc = int(csum[nxt - 1])
        else:
            cuts.append(self.n_db)
        n_blocks = len(cuts) - 1
        assert n_blocks >= db_ranks, "fewer database blocks than ranks"
        for b in range(db_rank, n_blocks, db_ranks):
            lo, hi = cuts[b], cuts[b + 1]
            t_off = self.doff[lo:hi + 1] - self.doff[lo]
            td, tl = workload.sequence_set(self.db[self.doff[lo]:self.doff[hi]], t_off)
            self.blocks.append((lo, hi, td, tl))
        self.n_blocks_total = n_blocks
        self.t_lo, self.t_hi, self.td, self.tl = self.blocks[0]
        if self.cfg["mode"] == "blastx":
            off = self.dna_off[self.q_lo:self.q_hi + 1] - self.dna_off[self.q_lo]
            self.qd, self.ql = hip.translated_block(self.dna[self.dna_off[self.q_lo]:self.dna_off[self.q_hi]], off)
            self.contexts = 6
        else:
            off = self.qoff[self.q_lo:self.q_hi + 1] - self.qoff[self.q_lo]
            self.qd, self.ql = workload.sequence_set(self.q[self.qoff[self.q_lo]:self.qoff[self.q_hi]], off)
            self.contexts = 1

    def seed_params(self, params):
        if self.cfg["sens"] == "fast":
            sp, gf = hip.seed_params_fast(threads=8), 0.0
        else:
            sp, gf = hip.seed_params_preset(self.cfg["sens"], params, threads=8)
        sp.query_translated = 1 if self.contexts == 6 else 0
        return sp, gf

    def write_fasta(self, d):
        synth.write_fasta(os.path.join(d, "db.faa"), "t", self.db, self.doff)
        if self.cfg["mode"] == "blastx":
            synth.write_dna_fasta(os.path.join(d, "q.fna"), "r", self.dna, self.dna_off)
            return os.path.join(d, "q.fna")
        synth.write_fasta(os.path.join(d, "q.faa"), "q", self.q, self.qoff)
        return os.path.join(d, "q.faa")


# the task timers of the reference's --log output that make up the hot path (what dmnd_seed_search + dmnd_extend replace): the seed
# stage of every shape x index chunk and the extension stage; everything else (open, load, masking, output) is outside it
HOT_TIMERS = ("Building reference histograms", "Building query histograms", "Allocating buffers", "Building reference seed array",
              "Building query seed array", "Computing hash join", "Masking low complexity seeds", "Searching alignments", "Deallocating memory",
              "Deallocating buffers", "Sorting trace points", "Computing partition", "Computing alignments", "Building seed filter",
              "Building query seed set", "Building reference index", "Building query index")


def _timers(log):
    t = {}
    for name, sec in re.findall(r"^([A-Za-z][^\[\n]*?)\.\.\.\s+\[([0-9.eE+-]+)s\]", log, re.M):
        t[name] = t.get(name, 0.0) + float(sec)
    return t


def cpu_baseline_reference(w, cores, e2e=True, parity_only=False):
    """The GENUINE reference (oracle/_ref/diamond_tap: /root/reference compiled in place; the tap only counts DP cells) on
    this box's host cores, on the FULL workload of the config, with as many threads as the cgroup allows.
      hot path  `--algo 0 --masking 0 --motif-masking 0 --log`: cells / (sum of the seed-stage and extension-stage task timers) --
                what bench.py's `value` (blocks resident, seed stage + extension) may be compared with; md5 of its output = parity_checked
      e2e       `diamond blastp ...` against `diamond-hip blastp ...` on the same files, whole processes (open + load the .dmnd,
                upload, mask, search, write the output file), wall clock around the process; default masking, masking off, and the
                stock command line (no --algo); outputs md5-compared (SURVEY.md 8d GCUPS_e2e; target >= 10x)
    Returns (cpu_baseline object, md5 of the hot-path run's output, e2e object | None), or (None, None, None)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "diamond_tap")
    ours_exe = os.path.join(ROOT, "diamond_amd", "diamond-hip")
    if not os.path.exists(exe):
        return None, None, None
    tmp = tempfile.mkdtemp(prefix="dmnd_cpu_")
    try:
        qfile = w.write_fasta(tmp)
        subprocess.run([exe, "makedb", "--in", os.path.join(tmp, "db.faa"), "-d", os.path.join(tmp, "db"), "-p", str(cores)],
                       check=True, capture_output=True, timeout=900)
        env = dict(os.environ, DIAMOND_TAP_CELLS=os.path.join(tmp, "cells.json"))
        blocks = []
        if w.n_blocks_total > 1:                             # the same block cut as ours (-b in billions of letters; +0.5 letter against rounding)
            blocks = ["-b", "%.12f" % ((w.block_letters + 0.5) / 1e9)]

        def run(binary, flags, out, threads=True):
            cmd = [binary, w.cfg["mode"]] + w.cfg["flags"] + flags + ["-q", qfile, "-d", os.path.join(tmp, "db"), "-o", os.path.join(tmp, out)] + blocks
            if threads:
                cmd += ["-p", str(cores)]
            else:
                # every GPU run starts on an idle device: for ~0.2 s after a process that held GBs of HBM has exited the driver is
                # still tearing its memory down, and the next process's HIP start-up waits for it (measured: 0.26 s against 0.51 s
                # wall for back-to-back masked runs of C2)
                time.sleep(1.0)
            t0 = time.perf_counter()
            r = subprocess.run(cmd, check=True, capture_output=True, text=True, env=env, timeout=3000)
            wall = time.perf_counter() - t0
            return wall, r.stdout + r.stderr, hashlib.md5(open(os.path.join(tmp, out), "rb").read()).hexdigest()

        hot_flags = ["--algo", "0", "--masking", "0", "--motif-masking", "0"]
        wall, log, md5 = run(exe, hot_flags + ["--log"], "out.tsv")
        if parity_only:                                      # N > 1: only what the records are compared with (the same block cut: -b above)
            return None, md5, None
        cells = json.load(open(os.path.join(tmp, "cells.json")))
        timers = _timers(log)
        hot_s = sum(v for k, v in timers.items() if k in HOT_TIMERS)
        sw = re.search(r"Time \(Smith Waterman\)\s*= ([0-9.eE+-]+)s", log)
        aligned = re.search(r"(\d+) queries aligned", log)
        total = re.search(r"Total time = ([0-9.eE+-]+)s", log)
        sw_s = float(sw.group(1)) if sw else float("nan")
        model = open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t") if os.path.exists("/proc/cpuinfo") else "?"
        n_aligned = int(aligned.group(1)) if aligned else 0
        base = {"value": cells["cells"] / hot_s / 1e9, "unit": "GCUPS", "cores": cores, "kind": "reference",
                "aligned_queries_per_s": n_aligned / hot_s,
                "hot_path": {"seconds": hot_s, "gcups": cells["cells"] / hot_s / 1e9,
                             "what": "sum of the reference's --log task timers of the seed stage (histograms, seed arrays, hash join, stage-1/2 filters) and the "
                                     "extension stage (sort, partition, 'Computing alignments'): the part of its run that dmnd_seed_search + dmnd_extend replace",
                             "timers_s": {k: round(v, 4) for k, v in timers.items() if k in HOT_TIMERS and v > 0}},
                "whole_process": {"seconds": wall, "gcups": cells["cells"] / wall / 1e9, "own_total_time_s": float(total.group(1)) if total else None},
                "sample": "reference diamond v2.2.2 `%s %s --algo 0 --masking 0 --motif-masking 0 -p %d` on the FULL workload (%d queries x %d seqs), "
                          "%d = the cgroup's CPU quota of this box (%d hardware threads visible), CPU %s; value = cells / hot-path seconds (%.3f s of %.2f s wall); "
                          "%d DpTargets / %d cells (both rounds), %d queries aligned; SW stage alone: %.3f CPU-s => %.2f GCUPS per core"
                          % (w.cfg["mode"], " ".join(w.cfg["flags"]), cores, w.n_queries, w.n_db, cores, os.cpu_count() or 0, model, hot_s, wall,
                             cells["targets"], cells["cells"], n_aligned, sw_s, cells["cells"] / sw_s / 1e9)}
        e2e_obj = None
        if e2e and os.path.exists(ours_exe):
            runs = {}
            plain = os.path.join(ROOT, "oracle", "_ref", "diamond")      # the unmodified reference binary (no tap) for the process timings
            if os.path.exists(plain):
                exe = plain
            for name, flags in (("default_masking", ["--algo", "0"]), ("masking_off", hot_flags), ("stock_command_line", [])):
                ref_wall, ref_log, ref_md5 = run(exe, flags, "ref_%s.tsv" % name)
                ours = [run(ours_exe, flags, "ours_%s.tsv" % name, threads=False) for _ in range(3)]
                ours_wall = sorted(x[0] for x in ours)[1]
                ref_total = re.search(r"Total time = ([0-9.eE+-]+)s", ref_log)
                runs[name] = {"flags": " ".join(w.cfg["flags"] + flags), "reference_md5": ref_md5, "reference_s": ref_wall, "reference_own_total_time_s": float(ref_total.group(1)) if ref_total else None,
                              "ours_s": ours_wall, "ours_runs_s": [round(x[0], 4) for x in ours], "speedup": ref_wall / ours_wall, "speedup_min": ref_wall / max(x[0] for x in ours), "parity": all(x[2] == ref_md5 for x in ours),
                              "ours_log": [l for l in ours[1][1].splitlines() if "[" in l or "Total time" in l]}
            e2e_obj = {"what": "whole processes on the same files (page cache warm): `diamond %s` on %d host threads against `diamond-hip %s` on one MI355X -- open and load the "
                               ".dmnd, upload, masking, seed stage, extension, output file; wall clock around the process (ours: median of 3, HIP start-up included, each run started 1 s after the previous process left the GPU)"
                               % (w.cfg["mode"], cores, w.cfg["mode"]),
                       "runs": runs, "speedup": runs["default_masking"]["speedup"], "speedup_min": min(r["speedup_min"] for r in runs.values()), "parity": all(r["parity"] for r in runs.values()),
                       "gcups_e2e": {"ours": cells["cells"] / runs["masking_off"]["ours_s"] / 1e9, "reference": cells["cells"] / runs["masking_off"]["reference_s"] / 1e9,
                                     "note": "SURVEY 8(d) GCUPS_e2e = the reference's cell count (both rounds) of the masking-off run / whole-process wall seconds"}}
        return base, md5, e2e_obj
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C2")
    ap.add_argument("--queries", type=int, default=None, help="default: the config's (10000; C5: 100000)")
    ap.add_argument("--families", type=int, default=None, help="protein families of 10 members in the database (default 100000; C5: 500000)")
    ap.add_argument("--host-threads", type=int, default=None, help="host threads of the extension stage per rank, divided among the extension contexts (default 12, fewer per rank with several ranks)")
    ap.add_argument("--shard", choices=["db", "query", "2d"], default=None, help="N > 1: database shards (all queries each), query shards (the whole database each), or 2d = two query halves x N/2 "
                    "database shards. Default: what the scaling model of the config's committed N = 1 line predicts to be fastest at this N (profiles/r06_bench_<config>.json scaling_model.best), db without one")
    ap.add_argument("--ext-contexts", type=int, default=None, help="batches extended concurrently, each on its own context and host thread team (default 3; 2 with several database blocks per rank)")
    ap.add_argument("--seed-contexts", type=int, default=None, help="seed stages in flight at the same time (own context and stream each; one with several database blocks per rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the whole-process comparison (diamond-hip against the reference binary on files)")
    ap.add_argument("--with-masking", action="store_true", help="(default since round 5; kept for old command lines) time the step of the default command line too")
    ap.add_argument("--no-masked-step", action="store_true", help="skip `masked_step`: after the timed steps the step of the default command line is timed as well (N=1, one database "
                    "block, blastp): block copies as loaded -> tantan + motif masking of both blocks on the device -> seed stage -> extension, back to back on one context")
    ap.add_argument("--same-block", action="store_true", help="search the SAME database block in every step. Default (N=1, one block per rank): two database blocks at different "
                    "places of HBM alternate between steps -- the block of the config and the same sequences in reverse order -- so that no step finds its 301 MB of "
                    "letters in the 256 MiB Infinity Cache from the step before")
    ap.add_argument("--host-join", action="store_true", help="join the records of several database blocks / ranks on the host (dmnd_join_blocks) instead of on the device")
    ap.add_argument("--no-pipeline", action="store_true", help="run seed stage and extension stage of a batch back to back on one context")
    args = ap.parse_args()

    # `python bench.py --gpus N` launches its own N ranks (one process per GPU over RCCL); under torchrun / torch.distributed.run
    # the ranks exist already. Either way the number of ranks that run must be the number asked for.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d ranks were launched; refusing to report a rank count that did not run" % (args.gpus, world))
    if os.environ.get("DMND_BENCH_LAUNCH_ONLY") == "1":      # tests/test_multigpu_gloo.py: the launcher alone, no GPU needed
        print("launch-only rank %d of %d local %d gpus %d" % (rank, world, local_rank, args.gpus), flush=True)
        return
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    # test hook: several ranks on ONE GPU (RCCL refuses that), used to exercise the multi-rank code path on a 1-GPU box:
    # DMND_BENCH_SHARE_GPU=1 maps every rank to cuda:0 and gathers the records over gloo instead of RCCL
    share_gpu = os.environ.get("DMND_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        sys.exit("bench.py: %d ranks but %d GPU(s) visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    coll_device = torch.device("cpu") if share_gpu else device
    if args.shard is None:
        # the decomposition the scaling model of this config's committed N = 1 line predicts to be fastest at this rank count (the
        # same choice on every rank: it is read from the repository, not measured here)
        args.shard = "db"
        try:
            best = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_%s.json" % args.config)))["scaling_model"]["best"].get(str(world))
            if world > 1 and best:
                args.shard = {"db": "db", "query": "query", "2xN/2": "2d"}[best["decomposition"]]
        except (OSError, KeyError, ValueError):
            pass
        if args.shard == "2d" and (world < 4 or world % 2):
            args.shard = "db"
    # every rank of a database-sharded run extends 1/N of the seed hits: its host part needs correspondingly fewer threads, and N
    # ranks share the node's cores
    threads = max(1, args.host_threads) if args.host_threads else (12 if world == 1 else max(3, 24 // world))
    many_blocks = CONFIGS[args.config].get("blocks", 1) > (world // 2 if args.shard == "2d" else world)
    if many_blocks and not args.host_threads:
        threads = max(threads, min(16, cgroup_cpus()) // max(world, 1))      # the host part of 8 blocks per batch is the bound of C5 (measured: 86 -> 66 ms per step)
    long_seed_stage = CONFIGS[args.config]["sens"] not in ("fast", "default")
    if args.ext_contexts is None:
        # several blocks per batch (C5): the extension of a block is a chain of short host and device phases; four batches in flight
        # with 24 host threads among them (more than the 16-CPU quota: most of the time they wait for the device) gave 50 ms per step
        # against 59 with two batches and 16 threads (profiles/r05_extension_contexts_sweep.txt) -- at the price of a third more CPU time
        # one block per batch: three batches in flight, one seed stage at a time -- except for the sensitive modes, whose seed stage is
        # 16 shapes long: two seed stages at a time and four batches (round 6, tools/gpu_r06v.sh / gpu_r06w.sh, the two database blocks
        # alternating in every variant: C3 145 -> 136 ms per step; C2 3.09 -> 3.00 -- inside the noise, and two stream kernels that
        # share the L2 request rate take twice as long each, which halves `roofline.frac` as it is defined on a launch's duration --;
        # C4 and C2skew unchanged)
        args.ext_contexts = (4 if world == 1 else 2) if many_blocks else (4 if world == 1 and long_seed_stage else 3)
        if many_blocks and world == 1 and not args.host_threads:
            threads = max(threads, 24)

    if args.queries is None:
        args.queries = CONFIGS[args.config].get("queries", 10_000)
    if args.families is None:
        args.families = CONFIGS[args.config].get("families", 100_000)
    w = Workload(args.config, args.families, args.queries, world, rank, args.shard)
    NB = len(w.blocks)
    params = hip.default_params()
    params.db_letters = float(w.db_letters)                  # e-values against the WHOLE database, whatever this rank holds
    seed_params, gf_evalue = w.seed_params(params)

    def make_ctx(b=0):
        c = hip.Context(device=local_rank, params=params)
        c.upload_block(hip.QUERY, w.qd, w.ql)
        if b is not None:
            c.upload_block(hip.TARGET, w.blocks[b][2], w.blocks[b][3])
        c.set_query_contexts(w.contexts)
        c.set_gapped_filter(gf_evalue)
        return c

    torch.cuda.synchronize()
    t_up = time.perf_counter()
    ctxs = [make_ctx(b) for b in range(NB)]                 # synchronous H2D of the blocks (pageable host memory); all stay resident in HBM
    upload_ms = (time.perf_counter() - t_up) * 1e3          # outside the timed region: inputs are resident when a step starts
    pipeline = not args.no_pipeline
    # Several database blocks on this rank (C5): ONE seed context searches them in turn -- every block stays resident in the
    # context that extends it and is aliased (dmnd_share_block), and the query seed index built for the first block of a batch
    # is kept for its other blocks (dmnd_set_query_index_reuse; reset at the start of every batch, so that every step still
    # indexes its query block once, as a run over many query blocks does)
    # --seed-contexts N: N seed stages at the same time, each on its own context. Measured on C2 (round 3): 3.38 / 3.33 / 3.46 ms per
    # step with 1 / 2 / 3 -- the device is the bound (the stream kernels of two batches share the L2 request rate, each takes twice
    # as long), so the default stays 1.
    # Several blocks (round 6): the seed stages of a BATCH (all its blocks, one query index) are one task on one context, and two such
    # tasks run at the same time on two contexts -- a seed stage of a 1.9e8-letter block is ~45 dependent launches and half a dozen
    # host waits around 2 ms of kernels, so one context alone left the device idle half of the time (C5: 54 -> see DESIGN 5.0).
    # One block (round 6, after the extension left the host): two seed stages at a time pay for the sensitive modes only (C3:
    # 145 -> 136 ms per step with four batches in the extension; C2: 3.09 / 3.16 ms per step with one seed stage and 3 / 4 extension
    # contexts, 3.08 / 3.00 with two, 3.46 / 3.02 with three). A first measurement of this without the alternating blocks showed
    # 7-12 % for every config: the second seed context switched the alternation off, and the block stayed in the Infinity Cache.
    if args.seed_contexts is None:
        args.seed_contexts = 2 if (NB > 1 or (world == 1 and long_seed_stage)) else 1
    SC = 1 if not pipeline else max(1, args.seed_contexts)
    ctxs_seed = [make_ctx(None) for _ in range(SC)] if NB > 1 else ([make_ctx(0) for _ in range(SC)] if pipeline else ctxs)
    import queue as queue_mod
    import threading
    seed_free = queue_mod.Queue()
    for c in ctxs_seed:
        seed_free.put(c)
    seed_lock = threading.Lock()
    # E batches are extended at the same time, each on its own context (own streams, buffers and host thread team): while one
    # batch is in a host phase (chaining, culling) the other one's sweep or walk runs, which the seed stage alone did not fill
    E = max(1, args.ext_contexts) if pipeline else 1
    ext_ctxs = [ctxs] + [[make_ctx(b) for b in range(NB)] for _ in range(E - 1)]
    ext_threads = max(1, threads // E)
    ctx, ctx_seed = ctxs[0], ctxs_seed[0]
    join_ctx = hip.Context(device=local_rank, params=params) if (world > 1 or NB > 1) else None      # stream + scratch of the device-side block join
    # (round 6) one GPU, several blocks: the records of a batch's blocks are joined where dmnd_extend left them in HBM
    # (dmnd_join_contexts_device), by the extension thread of the batch, on a join context of its context set
    set_join_ctxs = [hip.Context(device=local_rank, params=params) for _ in range(max(1, args.ext_contexts))] if (world == 1 and NB > 1 and not args.host_join) else None
    state = {"stream_ms": 0.0, "stream_launches": 0}
    # Two database blocks alternate between the steps (round 5): block B holds the sequences of block A in reverse order, at its own
    # place in HBM -- the same work per step (same hits, cells and records up to the target numbers), but a step never streams the
    # letters the step before it has just pulled through the Infinity Cache. Block B has its own seed contexts (as many as block A:
    # SC seed stages run at the same time, of either block) and its own extension context per extension team.
    alternate = world == 1 and NB == 1 and pipeline and not args.same_block
    alt_host = None
    if alternate:
        lens = np.diff(w.doff)
        r_lens = lens[::-1]
        r_off = np.concatenate([[0], np.cumsum(r_lens)]).astype(np.int64)
        idx = np.repeat(w.doff[:-1][::-1] - r_off[:-1], r_lens) + np.arange(int(r_off[-1]), dtype=np.int64)
        alt_host = workload.sequence_set(w.db[idx], r_off)
        del idx

        def make_alt_ctx():
            c = hip.Context(device=local_rank, params=params)
            c.upload_block(hip.QUERY, w.qd, w.ql)
            c.upload_block(hip.TARGET, alt_host[0], alt_host[1])
            c.set_query_contexts(w.contexts)
            c.set_gapped_filter(gf_evalue)
            return c
        alt_ext_ctxs = [make_alt_ctx() for _ in range(E)]
        seed_ctxs_alt = [make_alt_ctx() for _ in range(SC)]
        seed_free_alt = queue_mod.Queue()
        for c in seed_ctxs_alt:
            seed_free_alt.put(c)
        seed_counter = [0]

    def seed_stage(b=0, alt=False, c=None):
        torch.cuda.set_device(local_rank)
        held = c is not None                                 # a batch task brings the context it holds for all its blocks
        if c is None:
            c = (seed_free_alt if alt else seed_free).get()      # block B has its own seed contexts
        try:
            if NB > 1:
                if b == 0:
                    c.set_query_index_reuse(True)               # drops the index of the previous batch
                c.share_block(hip.TARGET, ctxs[b])
            t_s = time.perf_counter()
            hits = c.seed_search(seed_params)
            wall = (time.perf_counter() - t_s) * 1e3
            ms = c.seed_kernel_ms()
        finally:
            if alt:
                seed_free_alt.put(c)
            elif not held:
                seed_free.put(c)
        with seed_lock:
            state.setdefault("seed_wall", []).append(wall)
            state["stream_ms"] += ms[1]                         # every seed stage that ran since the counters were reset
            state["stream_launches"] += seed_params.n_shapes
        return hits, ms, alt

    def finish(parts):
        """What happens to a batch's records: database blocks are joined as the reference joins reference blocks -- the blocks of
        this rank with the other ranks' over RCCL."""
        if isinstance(parts, dict):                          # joined in HBM by the batch's extension thread
            full = parts["joined"]
            q = full["query"]
            state["joined_queries"] = int((q[1:] != q[:-1]).sum() + 1) if q.size else 0
            state["join_form"] = "records joined where dmnd_extend left them in HBM (dmnd_join_contexts_device)"
            return full
        if world > 1 and args.shard in ("db", "2d") or NB > 1:
            # one copy of the records (the concatenation); block ids -> database ordinals in place (2d: and the half's query numbers
            # -> the job's: the exchange below is keyed by the job's query ranges whatever rank searched a query)
            mine = np.concatenate([np.ascontiguousarray(m, dtype=hip.MATCH_DTYPE) for m in parts])
            if args.shard == "2d":
                mine["query"] += np.uint32(w.q_lo)
            at = 0
            for b, m in enumerate(parts):
                mine["target"][at:at + len(m)] += np.uint32(w.blocks[b][0])
                at += len(m)
            # all-to-all keyed by query range, rank g joins queries [g Q/G, (g+1) Q/G), one gather to rank 0.
            # Round 5: over RCCL the records stay in HBM from the first all-to-all to the gather and are merged there
            # (dmnd_join_blocks_device); one GPU with several blocks joins them on the device too. (gloo / --host-join: the host join.)
            if world > 1 and coll_device.type == "cuda" and not args.host_join:
                part, full = multigpu.query_range_join_device(mine, w.n_queries, coll_device, join_ctx)
            elif world == 1 and not args.host_join:
                part = full = join_ctx.join_blocks_device(mine, multigpu.TOPK)
                state["join_form"] = "host records uploaded (dmnd_join_blocks_device_host)"
            else:
                part, full = multigpu.query_range_join(mine, w.n_queries, coll_device, own=True)
            q = part["query"]
            state["joined_queries"] = int((q[1:] != q[:-1]).sum() + 1) if q.size else 0      # (the join returns query order)
            return full if rank == 0 else part
        return parts[0]

    def extend_batch(e, prefetched):
        """The extension stage of one batch on extension context set e: all database blocks of this rank, one after the other.
        prefetched: per block the future (or the result) of its seed stage."""
        torch.cuda.set_device(local_rank)
        t_b = time.perf_counter()
        parts, n_hits, seed_ms, ext_sum, alt_flags = [], 0, None, None, []
        for b in range(NB):
            got = prefetched[b] if prefetched is not None else seed_stage(b)
            hits, ms, alt = got.result() if hasattr(got, "result") else got
            alt_flags.append(bool(alt))
            ec = alt_ext_ctxs[e] if alt else ext_ctxs[e][b]
            m, _ = ec.extend(w.qd, alt_host[0] if alt else w.blocks[b][2], hits, threads=ext_threads)
            parts.append(m)
            n_hits += int(hits.size)
            seed_ms = list(ms) if seed_ms is None else [x + y for x, y in zip(seed_ms, ms)]
            st = ec.extend_stats()
            dv = ec.extend_device_stats()                       # how much of the batch the device half extended, and its sweeps' lane use
            st.update(device_queries=float(dv["queries"]), device_queries_back_to_host=float(dv["queries_back_to_host"]), device_items=float(dv["items"]),
                      band_diagonal_steps=dv["band_diagonal_steps"], wavefront_diagonal_steps=dv["wavefront_diagonal_steps"])
            # DP cells that round 2 SWEEPS (again): the device half's survivors without kept trace rows, and all round-2 targets of the
            # host half when it ran without kept traces (its own round-2 sweep time is what is left of slot [10])
            host_r2_ms = st["round2_swipe_kernel_ms"] - dv["round2_sweep_kernel_ms"]
            st["round2_cells_swept"] = dv["round2_cells_swept_again"] + ((st["round2_cells"] - dv["round2_cells"]) if host_r2_ms > 1e-6 else 0.0)
            ext_sum = dict(st) if ext_sum is None else {k: ext_sum[k] + st[k] for k in st}
        joined = None
        if set_join_ctxs is not None and not any(alt_flags) and all(ext_ctxs[e][b].extend_records_device()[1] >= 0 for b in range(NB)):
            joined = set_join_ctxs[e].join_contexts_device([ext_ctxs[e][b] for b in range(NB)], [w.blocks[b][0] for b in range(NB)], multigpu.TOPK, max_query=max(w.n_queries - 1, 1))
        state.setdefault("ext_wall", []).append((time.perf_counter() - t_b) * 1e3)
        return dict(parts=parts, joined=joined, hits=n_hits, seed_ms=seed_ms, ext=ext_sum, ext_wall_ms=(time.perf_counter() - t_b) * 1e3)

    def step(prefetched=None, done=None):
        """One batch: its extension (here, or already done on an extension thread), then the join of its records."""
        r = done if done is not None else extend_batch(0, prefetched)
        parts = r["parts"]
        joined_in_hbm = r.get("joined")
        state.update(hits=r["hits"], matches=np.concatenate(parts) if NB > 1 else parts[0], seed_ms=r["seed_ms"], ext=r["ext"], ext_wall_ms=r["ext_wall_ms"])

        def do_finish():
            torch.cuda.set_device(local_rank)
            t_f = time.perf_counter()
            state["records"] = finish(dict(joined=joined_in_hbm) if joined_in_hbm is not None else parts)
            state["finish_wall_ms"] = (time.perf_counter() - t_f) * 1e3
        if pipeline and (world > 1 or NB > 1):
            # the exchange + join of this batch's records runs on its own thread while the next batch is extended (the ranks'
            # finish threads issue their collectives in batch order); every pending join is awaited before the clock stops
            state.setdefault("finishing", []).append(finish_pool.submit(do_finish))
        else:
            do_finish()

    def drain():
        for f in state.pop("finishing", []):
            f.result()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if pipeline:
        import concurrent.futures
        seed_pool = concurrent.futures.ThreadPoolExecutor(max_workers=SC, initializer=name_thread, initargs=("bench-seed",))
        finish_pool = concurrent.futures.ThreadPoolExecutor(max_workers=1, initializer=name_thread, initargs=("bench-finish",))
        ext_pools = [concurrent.futures.ThreadPoolExecutor(max_workers=1, initializer=name_thread, initargs=("bench-extend",)) for _ in range(E)]      # a context runs one call at a time

    def seed_batch():
        """the seed stages of all blocks of one batch, on ONE seed context (its query index serves every block)"""
        c = seed_free.get()
        try:
            return [seed_stage(b, c=c) for b in range(NB)]
        finally:
            seed_free.put(c)

    class BatchPart:
        """block b's share of a batch task, with a future's face"""
        def __init__(self, fut, b):
            self.fut, self.b = fut, b

        def result(self):
            return self.fut.result()[self.b]

    def submit_seed_batch():
        fut = seed_pool.submit(seed_batch)
        return [BatchPart(fut, b) for b in range(NB)]

    def submit_seed(b):
        alt = False
        if alternate:
            alt = seed_counter[0] % 2 == 1
            seed_counter[0] += 1
        return seed_pool.submit(seed_stage, b, alt)

    PREFETCH = max(2, E + 1, 2 * SC)        # seed stages in flight or finished ahead of the extension (a bounded prefetch queue, like a data loader's)

    def run(n_steps, queue):
        """n_steps batches: every step takes the oldest seed-stage result of the queue (computed during earlier steps; by the
        warm-up for the first timed ones), submits ONE new seed stage to the seed thread and extends its own batch. n steps
        execute n seed stages and n extensions; the seed stages still queued at the end are awaited before the clock stops."""
        each, inflight = [], []
        t_a = time.perf_counter()

        def retire():                                        # batches complete in order: their records are joined in batch order on every rank
            nonlocal t_a
            step(done=inflight.pop(0).result())
            t_n = time.perf_counter()
            each.append(round((t_n - t_a) * 1e3, 2))
            t_a = t_n
        for s in range(n_steps):
            if pipeline:
                got = [queue.pop(0) for _ in range(NB)]
                if NB > 1:
                    queue.extend(submit_seed_batch())
                else:
                    queue.append(submit_seed(0))
                inflight.append(ext_pools[s % E].submit(extend_batch, s % E, got))
                if len(inflight) >= E:
                    retire()
            else:
                step()
                t_n = time.perf_counter()
                each.append(round((t_n - t_a) * 1e3, 2))
                t_a = t_n
        while inflight:
            retire()
        return each, queue

    # set-up, not warm-up: the first call of a context allocates its device buffers and trace arenas (tens to hundreds of ms);
    # the W warm-up steps only reach the first W of the E extension contexts, so the others are primed here
    for e in range(1, E):
        extend_batch(e, [seed_stage(b) for b in range(NB)])
    if alternate:
        for e in range(E):
            extend_batch(e, [seed_stage(0, True)])
    queue = ([p for _ in range(PREFETCH) for p in submit_seed_batch()] if NB > 1 else [submit_seed(0) for _ in range(PREFETCH)]) if pipeline else []
    _, queue = run(args.warmup, queue)
    for f in queue:
        f.result()                                           # the first timed steps find their seed hits ready
    drain()
    sync()
    # hipDeviceSynchronize lets the runtime release the hardware queues of idle streams; re-acquiring them costs the first
    # timed calls milliseconds (a streaming caller never synchronizes the whole device)
    for c in [x for cs in ext_ctxs for x in cs] + (ctxs_seed if ctxs_seed is not ctxs else []) + (alt_ext_ctxs + seed_ctxs_alt if alternate else []):
        c.touch_streams()
    state["stream_ms"], state["stream_launches"] = 0.0, 0
    state["seed_wall"], state["ext_wall"] = [], []
    multigpu.reset_stats()
    t0 = time.perf_counter()
    cpu0 = time.process_time()
    thr0 = thread_cpu_ms()
    each, queue = run(args.steps, queue)
    for f in queue:
        f.result()
    drain()
    sync()
    dt = time.perf_counter() - t0
    cpu_ms_per_step = (time.process_time() - cpu0) * 1e3 / args.steps      # CPU time of all threads of this process
    thr1 = thread_cpu_ms()
    xstats = dict(multigpu.STATS)                          # the exchanges of the timed steps (the serial steps below exchange again)
    cpu_by_thread = {k: round((v - thr0.get(k, 0.0)) / args.steps, 2) for k, v in sorted(thr1.items()) if v - thr0.get(k, 0.0) > 0}
    stream_ms, stream_launches = state["stream_ms"], state["stream_launches"]

    def pct(v, q):
        v = sorted(v)
        return v[min(len(v) - 1, int(q * len(v)))] if v else None
    lat = {"seed_stage_call_ms": {"p50": pct(state["seed_wall"], 0.5), "p95": pct(state["seed_wall"], 0.95)},
           "extension_of_a_batch_ms": {"p50": pct(state["ext_wall"], 0.5), "p95": pct(state["ext_wall"], 0.95)},
           "note": "wall time of the calls inside the timed, pipelined region (per database block for the seed stage, per batch for the extension): with "
                   "several batches in flight a batch's latency is several steps long; ms_per_step is the throughput figure"}
    pipe_ext = dict(state["ext"])
    # stage latencies of one batch on an otherwise idle GPU, after the timed region
    serial, alone = [], {}
    for _ in range(3):
        t_s = time.perf_counter()
        hs = [seed_stage(b) for b in range(NB)]
        t_m = time.perf_counter()
        step(prefetched=hs)
        drain()
        serial.append(((time.perf_counter() - t_s) * 1e3, (t_m - t_s) * 1e3))
    alone = {"batch_latency_ms": min(x[0] for x in serial), "seed_stage_call_ms": min(x[1] for x in serial),
             "extension_call_ms": state["ext_wall_ms"], "finish_ms": state["finish_wall_ms"], "seed_kernel_ms": list(state["seed_ms"]),
             "extension": dict(state["ext"])}
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    masked_step, masked_records = None, None
    if not args.no_masked_step and world == 1 and NB == 1 and w.contexts == 1:
        # The step of `diamond blastp --algo 0` with masking at its default (tantan on both blocks, motif soft masking for seed
        # generation; cli.cpp does the same calls per block pair). dmnd_mask_block works in place, so every step starts from the
        # letters as loaded: a device-to-device copy from a context that keeps them (it stands in for the block upload -- inputs
        # are resident in HBM when a step starts -- and is inside the timed region). One context, stages back to back.
        hip.load_motif_table()
        raw, mc = make_ctx(0), make_ctx(0)
        qd_m, td_m = w.qd.copy(), w.blocks[0][2].copy()       # the host copies the extension reads: patched with the masked positions
        parts = {k: [] for k in ("copy", "mask_query", "mask_target", "tantan_target_kernel", "tantan_target_call", "motif_target_call", "seed_stage", "extension", "step")}
        n_masked = None
        for s in range(args.warmup + args.steps):
            torch.cuda.synchronize()
            t_0 = time.perf_counter()
            mc.copy_block(hip.QUERY, raw)
            mc.copy_block(hip.TARGET, raw)
            t_1 = time.perf_counter()
            nq = mc.mask_block(hip.QUERY, qd_m)
            mc.soft_mask_block(hip.QUERY)
            t_2 = time.perf_counter()
            nt = mc.mask_block(hip.TARGET, td_m)
            k_ms = mc.mask_kernel_ms()
            t_2b = time.perf_counter()
            mc.soft_mask_block(hip.TARGET)
            t_3 = time.perf_counter()
            hits = mc.seed_search(seed_params)
            t_4 = time.perf_counter()
            masked_records, _ = mc.extend(qd_m, td_m, hits, threads=threads)
            t_5 = time.perf_counter()
            if s >= args.warmup:
                for k, v in zip(("copy", "mask_query", "mask_target", "seed_stage", "extension", "step"), (t_1 - t_0, t_2 - t_1, t_3 - t_2, t_4 - t_3, t_5 - t_4, t_5 - t_0)):
                    parts[k].append(v * 1e3)
                parts["tantan_target_kernel"].append(k_ms)
                parts["tantan_target_call"].append((t_2b - t_2) * 1e3)
                parts["motif_target_call"].append((t_3 - t_2b) * 1e3)
            n_masked = (int(nq), int(nt))
        st = mc.extend_stats()
        dvm = mc.extend_device_stats()
        m_host_r2 = (st["round2_cells"] - dvm["round2_cells"]) if st["round2_swipe_kernel_ms"] - dvm["round2_sweep_kernel_ms"] > 1e-6 else 0.0
        m_cells = st["round1_cells"] + dvm["round2_cells_swept_again"] + m_host_r2
        mean = {k: sum(v) / len(v) for k, v in parts.items()}
        serial_records = masked_records
        # The same work as a pipeline over the batches, as the headline step is one (round 6): three context sets in rotation;
        # while batch s is extended, batch s + 1 is in its seed stage and batch s + 2 is being masked -- the database block by
        # tantan + motifs on the set's context, the query block at the same time on a helper context of the set (its masked
        # letters are then copied over, device to device, and soft-masked in place). Every batch still goes through every stage
        # inside the timed region; the records of the last batch are the ones compared with the reference.
        import concurrent.futures as cf
        sets = [dict(c=mc if i == 0 else make_ctx(0), q=make_ctx(None), qd=qd_m if i == 0 else w.qd.copy(), td=td_m if i == 0 else w.blocks[0][2].copy()) for i in range(3)]
        for st_ in sets:
            st_["q"].upload_block(hip.QUERY, w.qd, w.ql)
        pools = {k: cf.ThreadPoolExecutor(max_workers=1, initializer=name_thread, initargs=("bench-" + k,)) for k in ("mask", "maskq", "seed", "extend")}

        def stage_mask(k):
            torch.cuda.set_device(local_rank)
            S = sets[k]
            def query_side():
                torch.cuda.set_device(local_rank)
                S["q"].copy_block(hip.QUERY, raw)
                return S["q"].mask_block(hip.QUERY, S["qd"])
            fq = pools["maskq"].submit(query_side)
            S["c"].copy_block(hip.TARGET, raw)
            nt = S["c"].mask_block(hip.TARGET, S["td"])
            S["c"].soft_mask_block(hip.TARGET)
            nq = fq.result()
            S["c"].copy_block(hip.QUERY, S["q"])
            S["c"].soft_mask_block(hip.QUERY)
            return int(nq), int(nt)

        def stage_seed(k, masked):
            torch.cuda.set_device(local_rank)
            masked.result()
            return sets[k]["c"].seed_search(seed_params)

        def stage_extend(k, hits):
            torch.cuda.set_device(local_rank)
            S = sets[k]
            return S["c"].extend(S["qd"], S["td"], hits.result(), threads=threads)[0]

        def run_masked(n):
            done, last = [], None
            for s_ in range(n):
                k = s_ % 3
                if len(done) >= 3:
                    last = done.pop(0).result()               # set k is free again once its batch has been extended
                fm = pools["mask"].submit(stage_mask, k)
                fs = pools["seed"].submit(stage_seed, k, fm)
                done.append(pools["extend"].submit(stage_extend, k, fs))
            for f in done:
                last = f.result()
            return last
        run_masked(max(args.warmup, 3))
        torch.cuda.synchronize()
        t_p = time.perf_counter()
        masked_records = run_masked(args.steps)
        torch.cuda.synchronize()
        piped_ms = (time.perf_counter() - t_p) * 1e3 / args.steps
        for pl in pools.values():
            pl.shutdown()
        same = len(masked_records) == len(serial_records) and bool((np.asarray(masked_records).view(np.uint8) == np.asarray(serial_records).view(np.uint8)).all())
        masked_step = {"what": "the work of `diamond blastp --algo 0` per block pair with masking at its default -- block letters as loaded copied device-to-device, tantan + motif "
                               "masking of the query block and of the database block, seed stage, extension -- as a pipeline over the batches (three context sets: batch s is "
                               "extended while batch s + 1 is in its seed stage and batch s + 2 is masked, query block and database block at the same time)",
                       "ms_per_step": piped_ms, "gcups": m_cells / piped_ms / 1e6, "cells_swept_per_step": m_cells,
                       "stages_back_to_back": {"ms_per_step": mean["step"], "what": "the same stages one after the other on one context (the figure of rounds 4 and 5)",
                                               "parts_ms": {k: round(v, 3) for k, v in mean.items() if k != "step"}},
                       "records_equal_back_to_back": same,
                       "masked_letters": {"query": n_masked[0], "database": n_masked[1]}, "records": int(len(masked_records)), "steps": args.steps}
        raw.close()
        for st_ in sets:
            st_["c"].close()
            st_["q"].close()

    # completion intervals as windows: at least five of them when the run has the steps for it, each a multiple of E steps
    WIN = E
    while len(each) // (2 * WIN) >= 5 and WIN < 4 * E:
        WIN += E
    win_ms = [sum(each[i:i + WIN]) / WIN for i in range(0, len(each) - WIN + 1, WIN)]
    win_median = sorted(win_ms)[len(win_ms) // 2] if len(win_ms) >= 8 else None      # (fewer windows: a median of clumped completions says nothing -- the mean is the figure)
    ext = pipe_ext
    # the job's DP cells: with database shards every rank sweeps its own targets, with query shards its own queries
    cells = torch.tensor([ext["round1_cells"], ext["round2_cells_swept"], ext["round2_cells"],
                          float(len(state["matches"])), ext["round1_targets"], ext["round2_targets"], float(state["hits"])], dtype=torch.float64, device=coll_device)
    if world > 1:
        dist.all_reduce(cells, op=dist.ReduceOp.SUM)
    r1_cells, r2_swept, r2_cells, n_matches_all, r1_targets, r2_targets, n_hits_all = [float(x) for x in cells.tolist()]
    cells_swept = r1_cells + r2_swept
    records = state["records"]
    if world > 1:         # every rank holds the records of its own queries (its query shard, or the query range it joined)
        n_aligned = torch.tensor([float(state["joined_queries"] if args.shard in ("db", "2d") else np.unique(records["query"]).size)], dtype=torch.float64, device=coll_device)
        dist.all_reduce(n_aligned, op=dist.ReduceOp.SUM)
        aligned = int(n_aligned.item())
    else:
        aligned = int(np.unique(records["query"]).size)

    closed = False
    pending_line = None
    if rank == 0:
        out = {
            "metric": "GCUPS + aligned queries/s, %s, %d queries vs %d-seq DB (seed stage + banded SW extension)" % (w.cfg["what"], w.n_queries, w.n_db),
            "value": cells_swept * args.steps / dt / 1e9, "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int16", "data": "synthetic",
            "aligned_queries_per_s": aligned * args.steps / dt,
            "reference_equivalent_gcups": (r1_cells + r2_cells) * args.steps / dt / 1e9,
            "config": {"workload": "%s: %s --algo 0 (no masking), %d queries x %d-seq DB (%d letters); per step %d seed hits, %d round-1 + %d round-2 "
                                   "DpTargets, %d alignments, %d queries aligned" % (args.config, w.cfg["what"], w.n_queries, w.n_db, w.db_letters, int(n_hits_all),
                                                                                      int(r1_targets), int(r2_targets), len(records), aligned),
                       "queries": w.n_queries, "db_seqs": w.n_db, "db_letters": w.db_letters,
                       "cells_swept_per_step": cells_swept, "reference_cells_per_step": r1_cells + r2_cells,
                       "cells_note": "value counts the DP cells the device sweeps (round 1 in traceback mode with kept trace rows; round 2 = a walk of the kept "
                                     "traces); the reference sweeps its round-2 targets a second time: reference_equivalent_gcups counts those too, the "
                                     "definition cpu_baseline is quoted on",
                       "dp_arithmetic": "packed int16 (two work items per wavefront), items that saturate re-run in int32",
                       "host_threads": threads,
                       "parallelism": ("%s-shard x%d (strong scaling of the fixed job)%s" % (args.shard, world, " + RCCL all-to-all of match records keyed by query range, rank g joins 1/N of the queries, gather to rank 0" if args.shard in ("db", "2d") else "")) if world > 1 else "single GPU"},
            "extension": ext,
            **({"masked_step": masked_step} if masked_step is not None else {}),
            "swipe_kernel_gcups": {"round1": ext["round1_cells"] / max(ext["round1_swipe_kernel_ms"], 1e-9) / 1e6,
                                   "traceback_kernel_ms": ext["traceback_kernel_ms"]},
            "seed_kernel_ms": dict(zip(["index_queries", "stream_reference", "mask_groups", "pair_filter", "total"], state["seed_ms"])),
            # seed-stage Gletters/s = (L_q + L_r) x shapes / seed-stage seconds (device time of its kernels)
            "seed_stage_gletters_per_s": (NB * int(w.ql[-1] - w.ql[0]) + sum(int(b[3][-1] - b[3][0]) for b in w.blocks)) * seed_params.n_shapes / max(state["seed_ms"][4], 1e-9) / 1e6,
            "pipeline": ("%d seed stage(s) at a time on their own contexts (low-priority streams), up to %d batches ahead of the extension stage; %d batches are extended at the same time "
                         "(own context and a team of %d host threads each)" % (SC, PREFETCH, E, ext_threads)) if pipeline else "off",
            "ms_each_step": each,
            # batches retire in clumps (three are extended at a time): the completion intervals as windows of three steps -- the
            # median window / 3 shows what a host hiccup (one long step) does to the mean that `value` is defined on
            "ms_per_step_median_of_3_step_windows": (sorted(sum(each[i:i + 3]) for i in range(0, len(each) - 2, 3))[len(range(0, len(each) - 2, 3)) // 2] / 3.0) if len(each) >= 24 else None,
            "ms_per_step_median": win_median,
            "ms_per_step_windows": {"window_steps": WIN, "ms_per_step_of_each_window": [round(x, 4) for x in win_ms], "median": win_median,
                                    "note": "the timed steps cut into windows of %d (a multiple of the %d batches that retire together); `ms_per_step` above is the MEAN over the whole "
                                            "timed region (what `value` is defined on), this is the median window" % (WIN, E)},
            "database_blocks_alternated": bool(alternate),
            "latency_in_pipeline": lat,
            "alone": alone,
            "host_cpu_ms_per_step": cpu_ms_per_step,
            "host_cpu_ms_per_step_by_thread": cpu_by_thread,      # by OS thread name (10 ms clock ticks): bench-* = this script's stage threads incl. the library calls they make, dmnd-pool = the library's host workers
            "host_cpu_quota": cgroup_cpus(),
            # what the exchange of the match records was (per step and rank 0's share; N = 1: no exchange, the fields say so): the
            # SCALE record's own evidence that the collective ran over N ranks
            "rccl": {"world_size": (dist.get_world_size() if world > 1 else 1), "backend": (dist.get_backend() if world > 1 else None),
                     "transport": (xstats["transport"] if world > 1 else "none (one rank; the blocks' records: %s)" % state.get("join_form", "one block, no join")),
                     "collectives_per_step": xstats["collectives"] / max(args.steps, 1),
                     "bytes_exchanged_per_step": (xstats["bytes_sent"] + xstats["bytes_received"]) / max(args.steps, 1),
                     "exchange_ms": xstats["exchange_s"] * 1e3 / max(args.steps, 1),
                     "shard": args.shard if world > 1 else None,
                     "note": "counted inside diamond_amd/multigpu.py around its all_to_all_single calls during the timed steps (rank 0's sends + receives; exchange_ms runs on the finish thread beside the next batch)"},
            # not part of `value`: one-time PCIe upload of both blocks, and the rate if it were paid on every step
            "block_upload_ms": upload_ms,
            "pcie_inclusive_gcups": cells_swept / ((dt / args.steps + upload_ms * 1e-3)) / 1e9,
        }
        # dominant kernel = the reference stream of the seed stage (seed_stream_fast_kernel), one launch per shape and step.
        # ALGORITHMIC bytes per launch = SURVEY.md 8(d)'s per-unit figure x the units of one launch: the reference side of
        # bytes_seed = S (L 1 + N 8 2 + ...) is 1 B (residue read once) + 16 B (one 8-byte (key32, loc32) seed entry written and
        # read back) per reference letter, N = L for the reference block. Our formulation never materialises the entries: it needs
        # 1 B per letter (design_bytes). Durations: HIP events around the launches on the seed stream (dmnd_seed_search).
        k_ms = stream_ms / max(stream_launches, 1)
        ref_letters = sum(int(b[3][-1] - b[3][0]) for b in w.blocks) // NB          # letters of one launch = one database block
        alg_bytes = 17 * ref_letters
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        k_alone = alone["seed_kernel_ms"][1] / (seed_params.n_shapes * NB)       # the serial step runs one launch per shape and database block
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
        out["roofline"] = {
            "bound": "hbm", "kernel": "seed_stream_fast_kernel (reference block streamed once per shape against the query seed table)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,      # PMC bytes cannot be collected inside a timed run: filled below from the committed rocprofv3 --pmc pass
            "measured_copy_gbs": copy_gbs, "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_per_unit": 17, "units_per_launch": ref_letters,
            "launches_per_step": seed_params.n_shapes, "kernel_ms": k_ms, "kernel_ms_alone": k_alone,
            "frac_alone": alg_bytes / (k_alone * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "design_bytes_per_launch": ref_letters, "frac_design_bytes": ref_letters / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "achieved = SURVEY 8(d) algorithmic bytes of the join's reference side (17 B per reference letter: residue + one 8-byte seed entry "
                    "written and read back) / average launch duration in the timed region. The kernel reads every letter once (design_bytes = 1 B per "
                    "letter) and probes a query-side table instead of materialising reference seed entries; it is bound by one L2 request per reference "
                    "position, not by HBM bytes (DESIGN.md 5)"}
        # HBM traffic of the dominant kernel per launch: FETCH_SIZE of a separate `rocprofv3 --pmc FETCH_SIZE` pass over this same
        # command (tools/profile_r03.sh), doubled as MI355X_MICROARCH.md prescribes for gfx950, plus WRITE_SIZE; only quoted for the
        # configuration and kernel variant it was measured on
        pmc_path = next((q for q in (os.path.join(ROOT, "profiles", "r%02d_pmc_summary_%s.json" % (r, args.config)) for r in (6, 5, 4, 3, 2)) if os.path.exists(q)), None)
        full_size = args.queries == CONFIGS[args.config].get("queries", 10_000) and args.families == CONFIGS[args.config].get("families", 100_000)
        if not (world == 1 and full_size and pmc_path):
            out["roofline"]["traffic_source"] = "no committed PMC pass applies (N = %d, full size: %s, file: %s)" % (world, full_size, pmc_path)
        if world == 1 and full_size and pmc_path:
            pmc = json.load(open(pmc_path))
            # the stream kernel's entry: of several instantiations the one with the most launches (the config's own)
            k = sorted([v for name, v in pmc.items() if "seed_stream_fast_kernel" in name and "FETCH_SIZE_x2_bytes_per_launch" in v], key=lambda v: -v.get("launches", 0))[:1]
            if len(k) == 1 and "FETCH_SIZE_x2_bytes_per_launch" in k[0]:
                rl = out["roofline"]
                rl["traffic"] = k[0]["FETCH_SIZE_x2_bytes_per_launch"] + k[0].get("WRITE_SIZE_bytes_per_launch", 0.0)
                rl["traffic_source"] = "%s: FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, bytes per launch" % os.path.relpath(pmc_path, ROOT)
                # three different HBM figures, named for what they are
                rl["hbm_measured"] = {"bytes_per_launch": rl["traffic"], "gbs": rl["traffic"] / (k_alone * 1e-3) / 1e9, "frac": rl["traffic"] / (k_alone * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "note": "what the kernel really moves over the fabric (PMC) / its launch time alone; `achieved` above is the SURVEY 8(d) MODEL of the "
                                              "reference's seed-array traffic (17 B per letter), which this kernel never moves"}
                if "TCC_REQ_sum_per_launch" in k[0]:
                    L2_REQ_PEAK = 2.7e11      # 34.5 TB/s of L2 bandwidth (MI355X_MICROARCH.md) / 128-byte lines
                    rl["l2_requests"] = {"bound": "l2 request rate", "requests_per_launch": k[0]["TCC_REQ_sum_per_launch"],
                                         "hits": k[0].get("TCC_HIT_sum_per_launch"), "misses": k[0].get("TCC_MISS_sum_per_launch"),
                                         "achieved": k[0]["TCC_REQ_sum_per_launch"] / (k_alone * 1e-3), "peak": L2_REQ_PEAK, "unit": "requests/s",
                                         "frac": k[0]["TCC_REQ_sum_per_launch"] / (k_alone * 1e-3) / L2_REQ_PEAK,
                                         "note": "the kernel's real limit: one 4-byte probe of the L2-resident query-seed bitmap per reference position = one L2 "
                                                 "request per letter (TCC_REQ of the committed PMC pass / launch time alone)"}
                # ... and as plain fields of the roofline object, for readers that do not descend into the sub-objects
                rl["hbm_measured_frac"] = rl["hbm_measured"]["frac"]
                if "l2_requests" in rl:
                    rl["l2_requests_frac"] = rl["l2_requests"]["frac"]
        out["sweep_roofline"] = sweep_roofline(args.config, ext)
        # SURVEY 8(d)'s whole-pipeline figure: bytes_total = bytes_seed + bytes_sw over the step's wall time
        S = seed_params.n_shapes
        L = ref_letters * NB + int(w.ql[-1] - w.ql[0]) * NB
        bytes_seed = S * (L * 1 + L * 16 + int(n_hits_all) * 15)
        bytes_sw = int((r1_targets + r2_targets) * w.db_letters / w.n_db) + int(w.ql[-1] - w.ql[0]) * 32 + int(r1_targets + r2_targets) * 72      # T ~ DpTargets x mean target length
        out["roofline"]["pipeline_hbm_model"] = {"bytes_per_step": bytes_seed + bytes_sw, "gbs": (bytes_seed + bytes_sw) / (dt / args.steps) / 1e9,
                                                 "frac": (bytes_seed + bytes_sw) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS / max(world, 1),
                                                 "note": "SURVEY 8(d): (bytes_seed + bytes_sw of the reference's data layout) / wall time of a step / (N x 8 TB/s); the joined-position "
                                                         "fingerprint term (P x 48 B) is left out (P is not counted on the device in --fast)"}
        if world == 1:
            out["scaling_model"] = scaling_model(out, ext, cpu_ms_per_step, cgroup_cpus(), dt / args.steps * 1e3, int(n_matches_all), w.n_blocks_total, E, threads)
        if seed_params.n_shapes > 2:
            out["roofline"]["note"] += ("; with short seeds (weight < 10) this kernel also runs the Hamming filter of every joined (query, reference) "
                                        "position pair, so its launch time covers the join AND the stage-1 filter")
        if not args.no_cpu_baseline and world == 1:          # reported on rank 0 at N=1 only
            # the whole-process runs get the GPU to themselves: this process's contexts (resident blocks, trace arenas) go first
            qids = ["%s%d" % ("r" if w.contexts == 6 else "q", i) for i in range(w.n_queries)]
            tids = ["t%d" % i for i in range(w.n_db)]
            text = hip.format_tab(state["records"], qids, tids, w.source_lens)
            masked_text = hip.format_tab(masked_records, qids, tids, w.source_lens) if masked_records is not None else None
            for c in (ctxs_seed if ctxs_seed is not ctxs else []) + [x for cs in ext_ctxs for x in cs] + (alt_ext_ctxs + seed_ctxs_alt if alternate else []) + ([join_ctx] if join_ctx else []) + (set_join_ctxs or []):
                c.close()
            closed = True
            torch.cuda.empty_cache()
            ref, ref_md5, e2e = cpu_baseline_reference(w, cgroup_cpus(), e2e=not args.no_e2e)
            if ref is not None:
                out["cpu_baseline"] = ref
                if e2e is not None:
                    out["e2e"] = e2e
                # parity of THIS run: the records of the last timed step, formatted as the reference's tabular output
                ours = hashlib.md5(text.encode()).hexdigest()
                out["parity_checked"] = ours == ref_md5
                out["parity"] = {"records_md5": ours, "reference_output_md5": ref_md5, "lines": text.count("\n")}
                if masked_text is not None and e2e is not None:
                    want = e2e["runs"]["default_masking"]["reference_md5"]
                    got = hashlib.md5(masked_text.encode()).hexdigest()
                    out["masked_step"]["parity"] = {"records_md5": got, "reference_output_md5": want, "matches": got == want}
        if world > 1:
            pending_line = out                               # printed below, behind the teardown of the process group
        else:
            print_line(out)
    if not closed:
        for c in (ctxs_seed if ctxs_seed is not ctxs else []) + [x for cs in ext_ctxs for x in cs] + (alt_ext_ctxs + seed_ctxs_alt if alternate else []) + ([join_ctx] if join_ctx else []) + (set_join_ctxs or []):
            c.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()                         # all ranks together: rank 0 is about to spend seconds on the reference run
    if pending_line is not None:
        out = pending_line
        if not args.no_cpu_baseline and state.get("records") is not None:
            # N > 1: the job's records (rank 0 holds them all) against the reference run with the same database block cut (-b as the
            # workload cut its blocks); the reference's own timing is an N = 1 matter and not taken here
            qids = ["%s%d" % ("r" if w.contexts == 6 else "q", i) for i in range(w.n_queries)]
            tids = ["t%d" % i for i in range(w.n_db)]
            text = hip.format_tab(state["records"], qids, tids, w.source_lens)
            _, ref_md5, _ = cpu_baseline_reference(w, cgroup_cpus(), e2e=False, parity_only=True)
            if ref_md5 is not None:
                ours = hashlib.md5(text.encode()).hexdigest()
                out["parity_checked"] = ours == ref_md5
                out["parity"] = {"records_md5": ours, "reference_output_md5": ref_md5, "lines": text.count("\n"),
                                 "note": "reference run on rank 0 with the database cut into the same %d blocks (-b)" % w.n_blocks_total}
        print_line(out)


def print_line(out):
    """the last thing on the line (what a tail of the output shows): the figures of this run in one place"""
    out["summary"] = {"ms_per_step": out["ms_per_step"], "value": out["value"], "unit": out["unit"], "parity_checked": out.get("parity_checked"),
                      "masked_step_ms_per_step": (out.get("masked_step") or {}).get("ms_per_step"), "host_cpu_ms_per_step": out["host_cpu_ms_per_step"],
                      "roofline_frac": out["roofline"]["frac"], "roofline_traffic_bytes": out["roofline"].get("traffic"),
                      "sweep_valu_issue_frac": out["sweep_roofline"].get("frac"), "sweep_frac_of_packed_issue_peak": out["sweep_roofline"].get("frac_of_packed_issue_peak"), "sweep_lane_use": out["sweep_roofline"].get("lane_use"),
                      "e2e_speedup_min": (out.get("e2e") or {}).get("speedup_min"), "rccl_world_size": out["rccl"]["world_size"]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
