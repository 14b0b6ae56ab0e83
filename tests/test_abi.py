"""CPU checks of the drop-in boundary: libdiamond_hip.so loads, exports every symbol that
include/diamond_hip.h declares, refuses to run without a GPU (no CPU fallback), and its host-side
parameter tables / e-value arithmetic equal the reference's (golden tap header + oracle)."""
import ctypes
import os
import re
import numpy as np
import pytest
import torch

import oracle_py as orc
from tapfile import read_tap
from diamond_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "diamond_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(dmnd_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 14
    lib = hip.load()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(hip.EXPORTS)
    assert lib.dmnd_abi_version() == 13


def test_struct_layouts_match_header():
    assert hip.DP_TARGET_DTYPE.itemsize == 40
    assert hip.HSP_DTYPE.itemsize == 56
    assert ctypes.sizeof(hip.Params) == 1024 + 8 + 9 * 8


def test_default_matrix_is_the_reference_blosum62():
    hdr, _ = read_tap(os.path.join(GOLDEN, "swipe_default.tap"), max_records=1)
    p = hip.default_params()
    assert np.array_equal(hip.matrix_of(p), hdr["matrix8"])
    assert (p.gap_open, p.gap_extend) == (hdr["gap_open"], hdr["gap_extend"])
    assert p.lambda_ == hdr["lambda"] and np.log(p.K) == pytest.approx(hdr["ln_k"], rel=1e-15)
    assert p.max_evalue == hdr["max_evalue"]


def test_host_evalue_matches_reference_values():
    lib = hip.load()
    for tap in ("swipe_default.tap", "swipe_long.tap"):
        hdr, recs = read_tap(os.path.join(GOLDEN, tap))
        p = hip.default_params()
        p.db_letters = hdr["db_letters"]
        n = 0
        for rec in recs:
            tl = {t["target_idx"]: t["true_target_len"] for t in rec["targets"]}
            for h in rec["hsps"]:
                ev = lib.dmnd_evalue_p(ctypes.byref(p), h["score"], len(rec["query"]), tl[h["swipe_target"]])
                # north_star tolerance: e-values within 1e-6 relative
                assert ev == pytest.approx(h["evalue"], rel=1e-6, abs=0)
                assert lib.dmnd_bitscore_p(ctypes.byref(p), float(h["score"])) == pytest.approx(h["bit_score"], rel=1e-12)
                n += 1
        assert n > 0


def test_banded_cols_matches_oracle():
    lib = hip.load()
    rng = np.random.default_rng(1)
    for _ in range(2000):
        q, t = int(rng.integers(1, 500)), int(rng.integers(1, 500))
        d0 = int(rng.integers(-t, q))
        d1 = d0 + int(rng.integers(1, 200))
        assert lib.dmnd_banded_cols(q, t, d0, d1) == orc.banded_cols(q, t, d0, d1)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback_without_gpu():
    with pytest.raises(hip.DiamondHipError):
        hip.Context()


def test_seed_parameter_presets_equal_the_reference_configuration():
    """dmnd_seed_params_fast / _default (host-only, no GPU call) must reproduce the configuration the genuine reference
    ran with, as tapped into the golden headers (shapes, partition bits, Hamming id, complexity cut, cutoff table)."""
    import ctypes, os
    import emu_py as emu
    from tapfile import read_ext_tap
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for tap, make in (("ext_fast.tap", lambda: hip.seed_params_fast(1)),
                      ("ext_default.tap", lambda: hip.seed_params_default(hip.default_params(), 1)),
                      ("ext_sensitive.tap", lambda: hip.seed_params_sensitive(hip.default_params(), 1))):
        cfg, _ = read_ext_tap(os.path.join(golden, tap), max_records=1)
        want, got = emu.seed_params_from_tap(cfg), make()
        for name, _t in hip.SeedParams._fields_:
            a, b = getattr(got, name), getattr(want, name)
            if hasattr(a, "_length_"):
                a, b = bytes(a), bytes(b)
            if tap == "ext_fast.tap" and name in ("short_query_cutoff", "cutoff_table"):
                continue                                             # unused when use_ungapped == 0
            assert a == b, (tap, name)
