"""-m gpu parity test of tantan masking on the MI355X (dmnd_mask_block, include/diamond_hip.h) through the C ABI: the
masked blocks must equal, letter for letter, what the genuine reference produced for the same sequences
(tests/golden/tantan.tap), and the oracle on sequences with planted repeats and awkward lengths."""
import os
import numpy as np
import pytest
import torch

import oracle_py as orc
from tapfile import read_tantan_tap
from diamond_amd import hip, workload
from test_oracle_seed import blosum62_matrix8

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    assert torch.cuda.is_available()
    c = hip.Context()
    yield c
    c.close()


def _block(seqs):
    lens = np.array([len(s) for s in seqs], np.int64)
    off = np.concatenate([[0], np.cumsum(lens)])
    return workload.sequence_set(np.concatenate(seqs).astype(np.int8), off)


def test_masked_block_equals_reference(ctx):
    hdr, recs = read_tantan_tap(os.path.join(GOLDEN, "tantan.tap"))
    data, limits = _block([r["before"] for r in recs])
    want, _ = _block([r["after"] for r in recs])
    ctx.upload_block(hip.TARGET, data, limits)
    host = data.copy()
    n = ctx.mask_block(hip.TARGET, host)
    assert np.array_equal(host, want)
    assert n >= int((want != data).sum()) > 3000
    assert ctx.mask_kernel_ms() > 0
    # the device copy is masked too: masking again (idempotence is NOT a property of tantan, so compare with the oracle)
    lr = orc.tantan_matrix(blosum62_matrix8())
    twice = want.copy()
    for i in range(len(limits) - 1):
        a, b = int(limits[i]), int(limits[i + 1]) - 1
        twice[a:b] = orc.tantan_mask(want[a:b], lr)[0]
    host2 = host.copy()                 # host_data is the caller's copy of the block as it stands in HBM: the masked positions are patched into it
    ctx.mask_block(hip.TARGET, host2)
    assert np.array_equal(host2, twice)


def test_planted_repeats_and_boundary_lengths_against_oracle(ctx):
    rng = np.random.default_rng(11)
    lr = orc.tantan_matrix(blosum62_matrix8())
    seqs = []
    for n in [1, 2, 15, 16, 17, 49, 50, 51, 63, 64, 65, 127, 128, 129, 191, 192, 193, 1000, 5000] + rng.integers(20, 900, 300).tolist():
        s = rng.integers(0, 20, n).astype(np.int8)
        if n > 40 and rng.random() < 0.7:
            unit = rng.integers(0, 20, int(rng.integers(1, 12))).astype(np.int8)
            a = int(rng.integers(0, n - 30))
            L = int(rng.integers(20, min(200, n - a)))
            s[a:a + L] = np.resize(unit, L)
            flip = rng.random(L) < 0.08
            s[a:a + L][flip] = rng.integers(0, 20, int(flip.sum()))
        seqs.append(s)
    data, limits = _block(seqs)
    ctx.upload_block(hip.QUERY, data, limits)
    host = data.copy()
    n = ctx.mask_block(hip.QUERY, host)
    total = 0
    for i, s in enumerate(seqs):
        a = int(limits[i])
        want, k = orc.tantan_mask(s, lr)
        assert np.array_equal(host[a:a + len(s)], want), (i, len(s))
        total += k
    assert n == total > 2000
    assert (host[:256] == 31).all() and (host[limits[1:] - 1] == 31).all()          # delimiters and padding untouched


def test_masking_a_subset_of_sequences_leaves_the_others_alone(ctx):
    """dmnd_mask_sequences (lazy masking: only the targets that have seed hits): the chosen sequences come out as dmnd_mask_block
    leaves them, every other letter -- in HBM and in the host copy -- is untouched."""
    hdr, recs = read_tantan_tap(os.path.join(GOLDEN, "tantan.tap"))
    data, limits = _block([r["before"] for r in recs])
    want, _ = _block([r["after"] for r in recs])
    n_seq = len(recs)
    rng = np.random.default_rng(5)
    for ids in (rng.permutation(n_seq)[: n_seq // 3], np.array([n_seq - 1, 0], np.int64), np.arange(n_seq)[::-1], np.zeros(0, np.int64)):
        ctx.upload_block(hip.TARGET, data, limits)
        host = data.copy()
        n = ctx.mask_sequences(hip.TARGET, host, ids)
        expect = data.copy()
        changed = 0
        for i in ids:
            a, b = int(limits[i]), int(limits[i + 1])
            expect[a:b] = want[a:b]
            changed += int((want[a:b] != data[a:b]).sum())
        assert np.array_equal(host, expect)
        assert n >= changed
        # the device copy holds the same letters: a second pass over the REST now completes the block
        rest = np.setdiff1d(np.arange(n_seq), ids)
        ctx.mask_sequences(hip.TARGET, host, rest)
        assert np.array_equal(host, want)
    with pytest.raises(Exception):
        ctx.mask_sequences(hip.TARGET, None, np.array([n_seq], np.int64))            # id outside the block


def test_two_contexts_with_different_motif_tables():
    """The motif table is process-wide by default (as the reference's is), read as a snapshot under a lock; a context can carry its
    own (dmnd_set_context_motif_table). Two contexts with different tables, soft-masking the same block from two threads at
    once, each cover exactly their own motifs."""
    import threading
    from diamond_amd import workload
    rng = np.random.default_rng(9)

    def code(letters):
        c = 0
        for x in letters:
            c = c * 20 + int(x)
        return c

    motif_a, motif_b = rng.integers(0, 20, 8), rng.integers(0, 20, 8)
    seqs = []
    for k in range(400):
        s = rng.integers(0, 20, int(rng.integers(60, 300))).astype(np.int8)
        p = int(rng.integers(0, len(s) - 8))
        s[p:p + 8] = motif_a if k % 2 == 0 else motif_b
        seqs.append(s)
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])])
    data, lim = workload.sequence_set(np.concatenate(seqs), off)
    ca, cb, cg = hip.Context(), hip.Context(), hip.Context()
    try:
        for c in (ca, cb, cg):
            c.upload_block(hip.QUERY, data, lim)
        hip.set_motif_table([code(motif_a), code(motif_b)])          # process-wide: both
        ca.set_motif_table([code(motif_a)])
        cb.set_motif_table([code(motif_b)])
        out = {}

        def run(name, c):
            out[name] = [c.soft_mask_block(hip.QUERY) for _ in range(20)]

        th = [threading.Thread(target=run, args=(n, c)) for n, c in (("a", ca), ("b", cb), ("g", cg))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert len(set(out["a"])) == 1 and len(set(out["b"])) == 1 and len(set(out["g"])) == 1
        na, nb, ng = out["a"][0], out["b"][0], out["g"][0]
        assert na >= 200 * 8 and nb >= 200 * 8 and ng >= na + nb - 64 and na < ng and nb < ng      # (chance occurrences of a motif elsewhere: a handful at most)
        ca.set_motif_table(None)
        assert ca.soft_mask_block(hip.QUERY) == ng
        cb.set_motif_table([])
        assert cb.soft_mask_block(hip.QUERY) == 0
    finally:
        for c in (ca, cb, cg):
            c.close()
        hip.load_motif_table()


def test_copy_block_restores_the_letters_as_loaded():
    """dmnd_copy_block: a context that masks in place takes the letters as loaded from a context that keeps them (device to
    device), so that a second masking pass reproduces the first; blocks of different shape are refused."""
    hdr, recs = read_tantan_tap(os.path.join(GOLDEN, "tantan.tap"))
    data, limits = _block([r["before"] for r in recs])
    want, _ = _block([r["after"] for r in recs])
    keep, work = hip.Context(), hip.Context()
    try:
        keep.upload_block(hip.TARGET, data, limits)
        work.upload_block(hip.TARGET, data, limits)
        for _ in range(2):
            host = data.copy()
            work.mask_block(hip.TARGET, host)
            assert np.array_equal(host, want)
            work.copy_block(hip.TARGET, keep)               # back to the unmasked letters
        other, olim = _block([r["before"] for r in recs[:10]])
        keep.upload_block(hip.TARGET, other, olim)
        with pytest.raises(Exception):
            work.copy_block(hip.TARGET, keep)
        with pytest.raises(Exception):
            work.copy_block(hip.QUERY, keep)                # no query block in either
    finally:
        keep.close()
        work.close()
