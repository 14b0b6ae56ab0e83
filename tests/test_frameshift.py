"""The three-frame banded sweep of frameshift alignment (blastx -F), CPU side:
  * the oracle restatement (oracle/frameshift_swipe.c) against known answers tapped from the genuine reference at its dispatch
    point banded_3frame_swipe (/root/reference/src/dp/dp.h:296; fixtures tests/golden/f3_*.tap, make_frameshift_golden.sh) --
    score-only calls with the reference's 16-channel vector batches (one band geometry per batch), traceback calls with
    coordinates in the read, statistics and transcripts incl. the frameshift operations;
  * the per-item code of the device kernels (diamond_amd/csrc/frameshift_core.h, run by tests/emu) against the oracle on the same
    items and on random ones."""
import ctypes
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as orc  # noqa: E402
from tapfile import read_3frame_tap  # noqa: E402

GOLDEN = os.path.join(HERE, "golden")
KEYS = "score frame q_begin q_end s_begin s_end qs_begin qs_end length identities mismatches positives gap_openings gaps".split()


def _emu():
    lib = ctypes.CDLL(os.path.join(HERE, "emu", "libswipe_emu.so"))
    return lib


def _frames(frames):
    fr = [np.ascontiguousarray(f, dtype=np.int8) for f in frames]
    ptrs = (ctypes.c_void_p * 3)(*[f.ctypes.data for f in fr])
    lens = (ctypes.c_int32 * 3)(*[len(f) for f in fr])
    return fr, ptrs, lens


def emu_score(frames, target, band, i0, i1, pos0, M, go, ge, fs, stride=1):
    fr, ptrs, lens = _frames(frames)
    t = np.ascontiguousarray(target, dtype=np.int8)
    m = np.ascontiguousarray(M, dtype=np.int8)
    mc = ctypes.c_int(0)
    assert i1 - i0 + 1 == band
    s = _emu().emu_3frame_score(ptrs, lens, ctypes.c_void_p(t.ctypes.data), len(t), int(i0), int(i1), int(pos0), ctypes.c_void_p(m.ctypes.data),
                                int(go), int(ge), int(fs), int(stride), ctypes.byref(mc))
    return s, mc.value


def emu_traceback(frames, strand, dna_len, target, d_begin, d_end, M, go, ge, fs):
    fr, ptrs, lens = _frames(frames)
    t = np.ascontiguousarray(target, dtype=np.int8)
    m = np.ascontiguousarray(M, dtype=np.int8)
    out = np.zeros(16, np.int32)
    cap = 2 * len(t) + len(fr[0]) + 64
    tr = np.zeros(cap, np.uint8)
    rc = _emu().emu_3frame_traceback(ptrs, lens, int(strand), int(dna_len), ctypes.c_void_p(t.ctypes.data), len(t), int(d_begin), int(d_end),
                                     ctypes.c_void_p(m.ctypes.data), int(go), int(ge), int(fs), ctypes.c_void_p(out.ctypes.data),
                                     ctypes.c_void_p(tr.ctypes.data), cap)
    d = dict(zip(KEYS + ["transcript_len", "status"], out.tolist()))
    return rc, d, tr[:d["transcript_len"]].copy()


@pytest.mark.parametrize("tap", ["f3_k3.tap", "f3_k1.tap"])
def test_oracle_and_device_code_equal_the_reference(tap):
    hdr, recs = read_3frame_tap(os.path.join(GOLDEN, tap))
    M, go, ge, fs = hdr["matrix8"], hdr["gap_open"], hdr["gap_extend"], hdr["frame_shift"]
    assert fs == 15
    ev = orc.evaluer(hdr["db_letters"], go, ge)
    n_score = n_trace = n_shift = 0
    for r in recs:
        fr = r["frames"]
        by_target = {}
        for h in r["hsps"]:
            by_target.setdefault(h["swipe_target"], []).append(h)
        if r["score_only"]:
            for batch in orc.frameshift_batches(r["targets"]):
                for k, band, i0, i1, pos0 in batch:
                    t = r["targets"][k]
                    s, mc, ov = orc.frameshift_score(fr, t["seq"], band, i0, i1, pos0, M, go, ge, fs)
                    assert not ov
                    assert emu_score(fr, t["seq"], band, i0, i1, pos0, M, go, ge, fs, stride=1 + k % 3) == (s, mc)
                    if orc.evalue(ev, s, len(fr[0]), len(t["seq"])) > hdr["max_evalue"]:
                        continue
                    rg = orc.frameshift_score_range(r["strand"], r["dna_len"], len(fr[0]), band, i0, pos0, mc)
                    out = np.zeros(4, np.int32)
                    _emu().emu_3frame_score_range(r["strand"], r["dna_len"], len(fr[0]), band, i0, pos0, mc, ctypes.c_void_p(out.ctypes.data))
                    assert out.tolist() == [rg["q_begin"], rg["q_end"], rg["qs_begin"], rg["qs_end"]]
                    assert any(h["score"] == s and all(h[x] == rg[x] for x in ("frame", "q_begin", "q_end", "qs_begin", "qs_end")) for h in by_target.get(t["target_idx"], [])), (t["target_idx"], s, rg)
                    n_score += 1
        else:
            for t in r["targets"]:
                rc, o, tr = orc.frameshift_traceback(fr, r["strand"], r["dna_len"], t["seq"], t["d_begin"], t["d_end"], M, go, ge, fs)
                assert rc == 0
                rc2, e, etr = emu_traceback(fr, r["strand"], r["dna_len"], t["seq"], t["d_begin"], t["d_end"], M, go, ge, fs)
                assert rc2 == 0 and e["score"] == o["score"]
                if o["score"] > 0:
                    assert all(e[x] == o[x] for x in KEYS), (e, o)
                    assert np.array_equal(etr, tr)
                if o["score"] <= 0 or orc.evalue(ev, o["score"], len(fr[0]), len(t["seq"])) > hdr["max_evalue"]:
                    continue
                assert any(all(h[x] == o[x] for x in KEYS) and np.array_equal(h["transcript"][:-1], tr) for h in by_target.get(t["target_idx"], [])), (t["target_idx"], o)
                n_trace += 1
                n_shift += int(((tr == 218) | (tr == 219)).sum())
    assert n_trace > 50 and n_shift > 20 and (n_score > 20 or tap == "f3_k3.tap")


def _random_case(rng, M):
    """A read of one strand (three frames cut from a random letter stream, as a translation gives them), a target related to one
    frame up to a frameshift in the middle, a random band."""
    n = int(rng.integers(4, 260))
    dna_len = 3 * n + int(rng.integers(0, 3))
    frames = [rng.integers(0, 21, max((dna_len - f) // 3, 0)).astype(np.int8) for f in range(3)]
    f0 = int(rng.integers(0, 3))
    cut = int(rng.integers(0, len(frames[f0]) + 1))
    f1 = (f0 + int(rng.integers(0, 3))) % 3
    t = np.concatenate([frames[f0][:cut], frames[f1][cut:cut + int(rng.integers(0, 120))], rng.integers(0, 20, int(rng.integers(0, 30))).astype(np.int8)])
    mut = rng.random(len(t)) < 0.25
    t[mut] = rng.integers(0, 20, int(mut.sum()))
    if len(t) == 0:
        t = np.array([3], np.int8)
    if rng.random() < 0.3:
        t = np.concatenate([rng.integers(0, 20, int(rng.integers(1, 40))).astype(np.int8), t])
    qlen, tlen = len(frames[0]), len(t)
    d0 = int(rng.integers(-(tlen - 1) - 3, qlen + 2))
    d1 = d0 + int(rng.integers(1, 90))
    d0, d1 = max(d0, -(tlen - 1)), min(d1, qlen - 1)
    if d1 <= d0:
        d0, d1 = max(-(tlen - 1), -2), min(qlen - 1, 3)
    if d1 <= d0:
        d0, d1 = 0, 1
    return frames, dna_len, t, d0, d1


def test_device_code_equals_oracle_on_random_items():
    rng = np.random.default_rng(5)
    from diamond_amd import hip
    M = hip.matrix_of(hip.default_params())
    n_hit = n_gap = 0
    for it in range(700):
        frames, dna_len, t, d0, d1 = _random_case(rng, M)
        if len(frames[0]) == 0:
            continue
        strand = it % 2
        fs = int(rng.choice([15, 15, 5, 30]))
        rc, o, tr = orc.frameshift_traceback(frames, strand, dna_len, t, d0, d1, M, 11, 1, fs)
        rc2, e, etr = emu_traceback(frames, strand, dna_len, t, d0, d1, M, 11, 1, fs)
        assert rc == 0 and rc2 == 0 and e["score"] == o["score"], (it, rc, rc2)
        if o["score"] > 0:
            assert all(e[x] == o[x] for x in KEYS), (it, e, o)
            assert np.array_equal(etr, tr)
            n_hit += 1
            n_gap += o["gap_openings"] > 0
        # the same target as a channel of a wider batch: band widened downwards, a later start
        widen, late = int(rng.integers(0, 40)), int(rng.integers(0, 25))
        band = d1 - d0 + widen
        i1 = max(max(d1 - 1, 0) - late, 0)
        i0 = i1 + 1 - band
        pos0 = i1 - (d1 - 1)
        s, mc, ov = orc.frameshift_score(frames, t, band, i0, i1, pos0, M, 11, 1, fs)
        assert emu_score(frames, t, band, i0, i1, pos0, M, 11, 1, fs, stride=1 + it % 4) == (s, mc)
    assert n_hit > 300 and n_gap > 20
