"""SEG low-complexity masking (`--masking seg`; host-only entry points dmnd_seg_ranges / dmnd_seg_mask_block) against the segments the
reference's own SEG finds (tests/golden/seg_golden.tsv.gz, minted by tests/golden/make_seg_golden.sh from the reference's
blast_seg.cpp) on its ctest fixture and on synthetic low-complexity cases."""
import gzip
import math
import os

import numpy as np

from diamond_amd import hip

HERE = os.path.dirname(os.path.abspath(__file__))
AA = "ARNDCQEGHILKMFPSTWYVBJZX*_"
CODE = {c: i for i, c in enumerate(AA)}


def _fasta(text):
    for r in text.split(">")[1:]:
        h, *s = r.strip().split("\n")
        yield h.split()[0], np.array([CODE.get(c, 23) for c in "".join(s).upper()], np.int8)


def _cases():
    yield from _fasta(open(os.path.join(HERE, "golden", "ref_ctest", "data.faa")).read())
    yield from _fasta(gzip.open(os.path.join(HERE, "golden", "seg_cases.faa.gz"), "rt").read())


def _golden():
    g = []
    for line in gzip.open(os.path.join(HERE, "golden", "seg_golden.tsv.gz"), "rt"):
        f = line.rstrip("\n").split("\t")
        g.append((f[0], [tuple(int(x) for x in r.split("-")) for r in f[1:]]))
    return g


def test_segments_equal_the_reference_seg():
    golden = _golden()
    n_seqs = n_segs = 0
    for (name, seq), (gname, want) in zip(_cases(), golden):
        assert name == gname
        assert hip.seg_ranges(seq) == want, name
        n_seqs += 1
        n_segs += len(want)
    assert n_seqs == len(golden) and n_seqs > 600 and n_segs > 1200
    assert max(e - b + 1 for _, w in golden for b, e in w) > 500          # regions far longer than the trim limit of 50


def test_mask_block_writes_the_mask_letter_over_every_segment():
    cases = list(_cases())[:450]
    golden = dict(_golden())
    data = [np.full(256, 31, np.int8)]
    limits = [256]
    for _, s in cases:
        data += [s, np.array([31], np.int8)]
        limits.append(limits[-1] + len(s) + 1)
    data = np.concatenate(data + [np.full(256, 31, np.int8)])
    before = data.copy()
    want = before.copy()
    for (name, s), lo in zip(cases, limits):
        for b, e in golden[name]:
            want[lo + b: lo + e + 1] = 23
    for threads in (1, 5):
        got = before.copy()
        n = hip.seg_mask_block(got, np.array(limits, np.int64), threads=threads)
        assert np.array_equal(got, want)
        assert n == sum(e - b + 1 for name, _ in cases for b, e in golden[name]) > 10000


def test_edge_cases():
    assert hip.seg_ranges(np.zeros(9, np.int8)) == []                     # shorter than the window of 10
    assert hip.seg_ranges(np.zeros(10, np.int8)) == [(0, 9)]
    assert hip.seg_ranges(np.full(60, 23, np.int8)) == []                 # only non-standard letters: every window has too many of them
    assert hip.seg_ranges(np.zeros(0, np.int8)) == []
    s = np.arange(200, dtype=np.int8) % 20                                # every window holds 10 different residues: K2 = log2(10)
    assert hip.seg_ranges(s) == []


def test_ln_factorial_table_and_stirling():
    lib = hip.load()
    import ctypes
    lib.dmnd_seg_lnfact.restype = ctypes.c_double
    lib.dmnd_seg_lnfact.argtypes = [ctypes.c_uint32]
    for n in list(range(0, 300)) + list(range(300, 10001, 7)) + [9999, 10000]:
        assert lib.dmnd_seg_lnfact(n) == float("%.6f" % math.lgamma(n + 1)), n     # the six-decimal table of blast_seg.cpp:53-1308
    for n in (10001, 12345, 1000000):
        assert lib.dmnd_seg_lnfact(n) == (n + 0.5) * math.log(n) - n + 0.9189385332
