"""The closed-form Hauser bias (bias_core.h, what hauser_bias_kernel runs per position) against the restated running-sum
loops of the reference (hauser_int8 behind dmnd_extend_plan, itself pinned on the reference's taps in test_extend_plan):
every length from 1 to 130 (all phase combinations of the five loops), long sequences, masked and non-standard letters."""
import numpy as np

import emu_py as emu
from diamond_amd import hip

BG_FREQ = np.array([7.4216205067993410e-02, 5.1614486141284638e-02, 4.4645808512757915e-02, 5.3626000838554413e-02,
                    2.4687457167944848e-02, 3.4259650591416023e-02, 5.4311925684587502e-02, 7.4146941452644999e-02, 2.6212984805266227e-02,
                    6.7917367618953756e-02, 9.8907868497150955e-02, 5.8155682303079680e-02, 2.4990197579643110e-02, 4.7418459742284751e-02,
                    3.8538003320306206e-02, 5.7229029476494421e-02, 5.0891364550287033e-02, 1.3029956129972148e-02, 3.2281512313758580e-02,
                    7.2919098205619245e-02])


def test_closed_form_bias_equals_running_sums():
    params = hip.default_params()
    M = hip.matrix_of(params)
    # ScoreMatrix::init_background_scores (score_matrix.cpp:241-248), accumulated in double as the library does
    bg = np.zeros(20)
    for i in range(20):
        acc = 0.0
        for j in range(20):
            acc += BG_FREQ[j] * float(M[i, j])
        bg[i] = acc
    rng = np.random.default_rng(8)
    lens = list(range(1, 131)) + [200, 301, 999, 4000]
    seqs = []
    for l in lens:
        s = rng.integers(0, 20, l).astype(np.int8)
        if l > 10 and l % 3 == 0:
            s[rng.integers(0, l, max(1, l // 8))] = rng.choice([20, 21, 22, 23, 24], max(1, l // 8))      # B J Z X * inside
        if l % 7 == 0:
            s[: l // 2] = s[0]                                                                                # low complexity
        seqs.append(s)
    off = np.r_[0, np.cumsum(lens)]
    data = np.concatenate(seqs)
    qd, ql = np.concatenate([np.full(256, 31, np.int8)] + [np.r_[s, np.int8(31)] for s in seqs] + [np.full(256, 31, np.int8)]), None
    ql = 256 + np.r_[0, np.cumsum(np.array(lens) + 1)].astype(np.int64)
    cbs, _ = hip.extend_plan(params, qd, ql, qd, ql, np.zeros(0, hip.SEED_HIT_DTYPE))
    for k, s in enumerate(seqs):
        want = cbs[ql[k]:ql[k] + len(s)]
        got = emu.hauser_bias(s, M, bg.astype(np.float32))
        assert np.array_equal(got, want), (len(s), np.nonzero(got != want)[0][:5])
