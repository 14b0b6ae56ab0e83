"""CPU check of the packed-int16, two-items-per-wavefront sweep (diamond_amd/csrc/swipe16_core.h) through the 64-lane emulator
tests/emu/swipe16_emu.cpp, against the oracle: both items of a pair must reproduce the reference's banded swipe bit for bit
(scores, end/start cells, statistics, transcripts) whatever item shares the wavefront with them -- items of different
lengths, bands and classes, bands leaving the matrix, 1-letter sequences, score ties, and 16-bit saturation."""
import os
import numpy as np
import pytest

import oracle_py as orc
import emu_py as emu
from tapfile import read_tap

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = "score q_begin q_end s_begin s_end length identities mismatches positives gap_openings gaps transcript_len".split()


def _check_pair(a, b, M, go=11, ge=1, force_p=0):
    rc, ra, rb = emu.banded_swipe16(a, b, M, go, ge, True, force_p)
    assert rc == 0
    rc, sa, sb = emu.banded_swipe16(a, b, M, go, ge, False, force_p)
    assert rc == 0
    rc, ca, cb = emu.banded_swipe16(a, b, M, go, ge, "score", force_p)
    assert rc == 0 and ca[0]["score"] == sa[0]["score"] and cb[0]["score"] == sb[0]["score"]
    for x, (e, etr), (s, _) in ((a, ra, sa), (b, rb, sb)):
        rc, o, otr = orc.banded_swipe(x["query"], x.get("cbs"), x["target"], x["d_begin"], x["d_end"], M, go, ge, orc.TRACEBACK)
        assert rc == 0
        assert e["status"] == 0
        assert e["score"] == o["score"] == s["score"], (e, o)
        if o["score"] <= 0:
            continue
        assert (s["q_end"], s["s_end"]) == (o["q_end"], o["s_end"])
        for k in KEYS:
            assert e[k] == o[k], (k, e, o)
        assert np.array_equal(etr, otr)


def _items_of(tap, step):
    hdr, recs = read_tap(os.path.join(GOLDEN, tap))
    items = []
    for rec in recs[::step]:
        for t in rec["targets"]:
            if t["d_end"] - t["d_begin"] <= 512:
                items.append({"query": rec["query"], "cbs": rec["cbs"], "target": t["seq"], "d_begin": t["d_begin"], "d_end": t["d_end"]})
    return hdr, items


@pytest.mark.parametrize("tap", ["swipe_default.tap", "swipe_fast.tap", "swipe_blastx.tap"])
def test_pairs_of_reference_targets(tap):
    hdr, items = _items_of(tap, 1 if "blastx" in tap else 4)
    assert len(items) >= 6
    rng = np.random.default_rng(3)
    # neighbours in the list (similar geometry, what the host's pairing produces) and random partners (anything goes)
    for i in range(0, len(items) - 1, 2):
        _check_pair(items[i], items[i + 1], hdr["matrix8"], hdr["gap_open"], hdr["gap_extend"])
    for _ in range(20):
        i, j = rng.integers(0, len(items), 2)
        _check_pair(items[i], items[j], hdr["matrix8"], hdr["gap_open"], hdr["gap_extend"])


def _random_item(rng, it, wide=False):
    qlen = int(rng.integers(1, 200))
    tlen = int(rng.integers(1, 200))
    q = rng.integers(0, 25, qlen).astype(np.int8)
    if it % 2 == 0:
        t = q.copy()
        mut = rng.random(qlen) < 0.3
        t[mut] = rng.integers(0, 20, int(mut.sum()))
        cut = int(rng.integers(0, qlen))
        t = np.concatenate([t[:cut], rng.integers(0, 20, int(rng.integers(0, 6))).astype(np.int8), t[cut + int(rng.integers(0, 4)):]])
        if len(t) == 0:
            t = q[:1].copy()
        tlen = len(t)
    else:
        t = rng.integers(0, 25, tlen).astype(np.int8)
    if it % 5 == 0:
        q[rng.integers(0, qlen)] |= -128
    d0 = int(rng.integers(-(tlen - 1) - 5, qlen + 3))
    d1 = d0 + int(rng.integers(1, 140 if not wide else 500))
    if d1 <= -(tlen - 1) or d0 >= qlen:
        d0, d1 = -2, 3
    cbs = rng.integers(-3, 3, qlen).astype(np.int8) if it % 3 else None
    return {"query": q, "cbs": cbs, "target": t, "d_begin": d0, "d_end": d1}


def test_random_pairs_every_class():
    hdr, _ = read_tap(os.path.join(GOLDEN, "swipe_fast.tap"), max_records=1)
    M = hdr["matrix8"]
    rng = np.random.default_rng(16)
    for it in range(160):
        a, b = _random_item(rng, it, wide=it % 4 == 3), _random_item(rng, it + 1, wide=it % 8 == 7)
        _check_pair(a, b, M, force_p=(0, 0, 2, 4)[it % 4])
    # an item paired with itself (what the kernel does with the odd item of a class)
    a = _random_item(rng, 0)
    _check_pair(a, a, M)


def test_row_classes():
    """The row classes P = 3 / 5 (an item pair on the 16 lanes of a DPP row, 6 / 10 diagonals per lane, 15-byte trace records):
    bands of up to 96 / 160 diagonals, everything else as in the wavefront classes."""
    hdr, _ = read_tap(os.path.join(GOLDEN, "swipe_fast.tap"), max_records=1)
    M = hdr["matrix8"]
    rng = np.random.default_rng(35)
    n = {3: 0, 5: 0}
    for it in range(400):
        a, b = _random_item(rng, it, wide=it % 4 == 3), _random_item(rng, it + 1, wide=it % 8 == 7)
        width = max(a["d_end"] - a["d_begin"], b["d_end"] - b["d_begin"])
        P = 3 if width <= 96 and it % 3 else 5 if width <= 160 else 0
        if P:
            _check_pair(a, b, M, force_p=P)
            n[P] += 1
    assert n[3] >= 60 and n[5] >= 60
    # bands at the classes' limits, items of very different lengths in one row, an item paired with itself
    for width, P in ((96, 3), (95, 3), (160, 5), (159, 5), (1, 3), (2, 5)):
        a = _random_item(rng, 0); b = _random_item(rng, 2)
        a["d_begin"] = -min(len(a["target"]) - 1, width // 2); a["d_end"] = a["d_begin"] + width
        _check_pair(a, b if b["d_end"] - b["d_begin"] <= 32 * P else a, M, force_p=P)
    hdr, items = _items_of("swipe_default.tap", 4)
    narrow = [x for x in items if x["d_end"] - x["d_begin"] <= 160]
    assert len(narrow) >= 6
    for i in range(0, len(narrow) - 1, 2):
        w = max(narrow[i]["d_end"] - narrow[i]["d_begin"], narrow[i + 1]["d_end"] - narrow[i + 1]["d_begin"])
        _check_pair(narrow[i], narrow[i + 1], hdr["matrix8"], hdr["gap_open"], hdr["gap_extend"], force_p=3 if w <= 96 else 5)


def test_launch_classes_and_trace_layout_arithmetic():
    """Band -> class (with and without the row classes), class <-> class number, and the trace layout of every class the packed
    kernels take: every (pair-step, diagonal pair) of an item has a byte of its own inside the item's trace_bytes."""
    import ctypes
    L = emu.lib()
    L.emu_trace_bytes.restype = ctypes.c_longlong
    L.emu_trace_byte_index.restype = ctypes.c_longlong
    for band in range(1, 700):
        p2 = L.emu_band_class(band, 0)
        assert p2 & (p2 - 1) == 0 and 128 * p2 >= band and (p2 == 1 or 64 * p2 < band)
        pr = L.emu_band_class(band, 1)
        assert pr == (3 if band <= 96 else 1 if band <= 128 else 5 if band <= 160 else p2)
        assert 2 * pr * L.emu_class_lanes(pr) >= band
    seen = set()
    for P in (1, 2, 3, 4, 5, 8, 16, 32, 64, 128, 256, 512):
        c = L.emu_class_index(P)
        assert 0 <= c < 16 and c not in seen and L.emu_class_of_index(c) == P
        seen.add(c)
        assert L.emu_items_per_wave16(P) == (8 if P in (3, 5) else 2) and L.emu_class_lanes(P) == (16 if P in (3, 5) else 64)
    rng = np.random.default_rng(4)
    for P in (1, 2, 3, 4, 5, 16):
        lanes = L.emu_class_lanes(P)
        for _ in range(6):
            qlen, tlen = int(rng.integers(1, 300)), int(rng.integers(1, 300))
            width = int(rng.integers(1, 2 * P * lanes + 1))
            d0 = int(rng.integers(-(tlen - 1), qlen))
            pairs = L.emu_trace_pairs(qlen, tlen, d0, d0 + width)
            size = L.emu_trace_bytes(qlen, tlen, d0, d0 + width, P)
            idx = np.array([[L.emu_trace_byte_index(P, t, x) for x in range(lanes * P)] for t in range(pairs)], dtype=np.int64)
            assert idx.size == 0 or (idx.min() >= 0 and idx.max() < size)
            assert len(np.unique(idx)) == idx.size
            if P < 16 and pairs:      # a lane's bytes of one group of pair-steps are one 16-byte record
                G = 16 // P
                rec = idx[:G, :P] // 16
                assert len(np.unique(rec)) == 1


def test_ties_and_repeats():
    hdr, _ = read_tap(os.path.join(GOLDEN, "swipe_fast.tap"), max_records=1)
    M = hdr["matrix8"]
    rng = np.random.default_rng(21)
    items = []
    for it in range(60):
        motif = rng.integers(0, 20, int(rng.integers(2, 6))).astype(np.int8)
        q = np.resize(motif, int(rng.integers(5, 120)))
        t = np.resize(motif, int(rng.integers(5, 160)))
        if it % 3 == 0:
            q = q.copy(); t = t.copy()
            q[rng.integers(0, len(q), 2)] = 23
            t[rng.integers(0, len(t), 3)] = 23
        d0 = int(rng.integers(-(len(t) - 1), len(q) - 1))
        d1 = d0 + int(rng.integers(1, 200))
        if d1 <= -(len(t) - 1) or d0 >= len(q):
            continue
        items.append({"query": q, "cbs": rng.integers(-2, 2, len(q)).astype(np.int8) if it % 2 else None, "target": t, "d_begin": d0, "d_end": d1})
    for i in range(0, len(items) - 1, 2):
        _check_pair(items[i], items[i + 1], M, force_p=(1, 2, 4)[i % 3])


def test_saturation_is_reported():
    """A self alignment scoring above 32767 saturates: the emulator (like the kernel) reports exactly 32767, which the host
    treats as overflow and re-runs in the 32-bit kernel; the partner item in the other half is unaffected."""
    hdr, _ = read_tap(os.path.join(GOLDEN, "swipe_fast.tap"), max_records=1)
    M = hdr["matrix8"]
    rng = np.random.default_rng(2)
    q = np.full(3400, 17, np.int8)              # W x W = 11 per column
    big = {"query": q, "cbs": None, "target": q.copy(), "d_begin": -20, "d_end": 21}
    small = _random_item(rng, 2)
    rc, (ea, _), (eb, _) = emu.banded_swipe16(big, small, M, 11, 1, False)
    assert rc == 0 and ea["score"] == 32767
    rc, o, _ = orc.banded_swipe(small["query"], small["cbs"], small["target"], small["d_begin"], small["d_end"], M, 11, 1, orc.COORDS)
    assert eb["score"] == o["score"]
    rc, (eb2, _), (ea2, _) = emu.banded_swipe16(small, big, M, 11, 1, False)
    assert ea2["score"] == 32767 and eb2["score"] == o["score"]
