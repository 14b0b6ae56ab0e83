#!/usr/bin/env python3
"""Full-scale A/B against the genuine reference (oracle/_ref/diamond) on BASELINE configs C2 / C3 / C4: writes the synthetic
FASTA files, runs both binaries with the same flags and compares the tabular output byte for byte.
usage: tools/fullscale_parity.py WORKDIR CONFIG[,CONFIG...] [--algo 0|1] [--threads N]   CONFIG in c2 c3 c4"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diamond_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "diamond")
CLI = os.path.join(ROOT, "diamond_amd", "diamond-hip")


def run(cmd, env=None):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        print("FAILED", " ".join(cmd), r.stderr[-1500:])
        sys.exit(2)
    return dt, r.stdout + r.stderr


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def main():
    work, configs = sys.argv[1], sys.argv[2].split(",")
    algo = sys.argv[sys.argv.index("--algo") + 1] if "--algo" in sys.argv else "0"
    threads = sys.argv[sys.argv.index("--threads") + 1] if "--threads" in sys.argv else "16"
    os.makedirs(work, exist_ok=True)
    db, doff, q, qoff = synth.generate(100_000, members=10, queries=10_000, seed=20260923)
    dbf, qf, rf = os.path.join(work, "db.faa"), os.path.join(work, "q.faa"), os.path.join(work, "reads.fna")
    if not os.path.exists(dbf):
        synth.write_fasta(dbf, "t", db, doff)
        synth.write_fasta(qf, "q", q, qoff)
        dna, off = synth.back_translate(q[:qoff[5000]], qoff[:5001], seed=5)
        synth.write_dna_fasta(rf, "r", dna, off)
    if not os.path.exists(os.path.join(work, "db.dmnd")):
        dt, _ = run([REF, "makedb", "--in", dbf, "-d", os.path.join(work, "db"), "-p", threads])
        print("reference makedb %.1f s" % dt)
    ok = True
    for cfg in configs:
        mode, sens, qfile = {"c2": ("blastp", ["--fast"], qf), "c3": ("blastp", ["--sensitive"], qf), "c4": ("blastx", [], rf)}[cfg]
        common = sens + ["--masking", "0", "-q", qfile, "-d", os.path.join(work, "db.dmnd"), "-p", threads]
        ref_out, hip_out = os.path.join(work, "%s_a%s_ref.tsv" % (cfg, algo)), os.path.join(work, "%s_a%s_hip.tsv" % (cfg, algo))
        t_ref, log = run([REF, mode] + common + ["--algo", algo, "--motif-masking", "0", "-o", ref_out])
        alg = [l for l in log.splitlines() if l.startswith("Algorithm")]
        t_hip, log2 = run([CLI, mode] + common + ["--algo", algo, "-o", hip_out])
        same = open(ref_out, "rb").read() == open(hip_out, "rb").read()
        ok &= same
        print("%s algo %s: reference %.1f s (%s), diamond-hip %.1f s, %d lines, identical=%s md5 %s" % (
            cfg, algo, t_ref, alg[0] if alg else "?", t_hip, sum(1 for _ in open(ref_out)), same, md5(ref_out)))
        if not same:
            a, b = open(ref_out).read().splitlines(), open(hip_out).read().splitlines()
            sa, sb = set(a), set(b)
            print("  lines only in reference: %d, only in diamond-hip: %d" % (len(sa - sb), len(sb - sa)))
            for l in sorted(sa - sb)[:5]:
                print("  ref:", l)
            for l in sorted(sb - sa)[:5]:
                print("  hip:", l)
        print("  hip log:", " | ".join(l for l in log2.splitlines() if "[" in l or "Total" in l)[-600:])
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
