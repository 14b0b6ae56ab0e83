"""CPU check of the motif soft masking (diamond_amd/csrc/mask_core.h, run through tests/emu) against the plain-C restatement of
the reference's mask_motifs (oracle/motif_mask.c): same masked letters and same covered-letter count on random sequences with
planted motifs -- overlapping and touching occurrences (ranges merge), ranges longer than max_motif_len (not masked), sequences
where motifs cover half of the letters or more (nothing masked), motifs interrupted by non-standard letters."""
import ctypes
import numpy as np

import emu_py as emu
import oracle_py as orc


def _code(kmer):
    c = 0
    for l in kmer:
        c = c * 20 + int(l)
    return c


def test_motif_masking_equals_the_restatement():
    rng = np.random.default_rng(8)
    motifs = [rng.integers(0, 20, 8).astype(np.int8) for _ in range(300)]
    # motifs that chain (suffix of one = prefix of the next) so that long merged ranges occur
    for i in range(100):
        m = motifs[i].copy()
        motifs.append(np.concatenate([m[1:], rng.integers(0, 20, 1).astype(np.int8)]))
    table = np.array(sorted(set(_code(m) for m in motifs)), dtype=np.uint64)
    seen_masked = seen_half = seen_long = 0
    for it in range(600):
        n = int(rng.integers(1, 260))
        seq = rng.integers(0, 20, n).astype(np.int8)
        for _ in range(int(rng.integers(0, 7 if it % 5 else 40))):
            m = motifs[int(rng.integers(0, len(motifs)))]
            if n > 8:
                p = int(rng.integers(0, n - 8))
                seq[p:p + 8] = m
        if it % 7 == 0 and n > 150:                      # five motifs back to back: one merged range of 40 letters, too long to mask
            p = int(rng.integers(0, n - 41))
            for k in range(5):
                seq[p + 8 * k:p + 8 * k + 8] = motifs[int(rng.integers(0, len(motifs)))]
        if it % 4 == 0 and n > 3:
            seq[rng.integers(0, n, 3)] = rng.integers(20, 25, 3)
        a, b = seq.copy(), seq.copy()
        na = emu.lib().emu_motif_mask(a.ctypes.data_as(ctypes.c_void_p), n, table.ctypes.data_as(ctypes.c_void_p), len(table), 30)
        nb = orc.lib().oracle_motif_mask(b.ctypes.data_as(ctypes.c_void_p), n, table.ctypes.data_as(ctypes.c_void_p), len(table), 30)
        assert na == nb, (it, na, nb)
        assert np.array_equal(a, b), it
        seen_masked += int((a != seq).any())
        seen_half += int(nb == 0 and it % 5 == 0)
        # a covered stretch longer than 30 letters stays unmasked
        seen_long += int(nb > (a != seq).sum() > 0)
    assert seen_masked > 100 and seen_half > 20 and seen_long > 3
