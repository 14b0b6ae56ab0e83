"""CPU restatements of the arithmetic behind three round-5 kernels, each against the plain definition it replaces (the kernels
themselves are compared with the reference under -m gpu: tests/test_gpu_seed.py, the full-size runs):
  * seed_classify_kernel (seed_kernels.hip): key classes of all 16 windows of a group as XORs of shifted bit planes -- right only
    because seed_class (seed_core.h) is GF(2)-linear -- and window validity as ORs of shifted maps, runs by doubling;
  * stage2_score_regs: clip_window as a 96-bit delimiter mask, ungapped_window_score unrolled over all 96 letters with the letters
    outside the window neutralised;
  * seed_deferred_kernel's lower bound with 64 probes per round."""
import bisect

import numpy as np

M64 = (1 << 64) - 1


def seed_class(key):                                       # seed_core.h seed_class
    x = (key & 0xffffffff) ^ (key >> 32)
    x ^= x >> 16
    x ^= x >> 8
    x ^= x >> 4
    return (x ^ (x >> 3)) & 7


def test_seed_class_is_linear_and_has_two_coefficient_words():
    rng = np.random.default_rng(1)
    unit = [seed_class(1 << b) for b in range(64)]
    assert seed_class(0) == 0
    for key in rng.integers(0, 1 << 63, size=2000, dtype=np.int64).tolist():
        key = (key * 2654435761) & M64
        acc = 0
        for b in range(64):
            if (key >> b) & 1:
                acc ^= unit[b]
        assert acc == seed_class(key)
    # what the launch code groups by: the coefficient word of a nibble position
    words = {tuple(unit[4 * i: 4 * i + 4]) for i in range(16)}
    assert len(words) == 2


def _classify_planes(codes, delim, bad, care, length, hashed, first, last):
    """The kernel's arithmetic for ONE group of 16 windows: codes / delim / bad cover 32 letters (the group and the one behind it)."""
    plane = [sum(((codes[t] >> b) & 1) << t for t in range(32)) for b in range(4)]
    coef = {}
    for i in care:
        coef.setdefault(tuple(seed_class(1 << (4 * i + b)) for b in range(4)), []).append(i)
    cls = [0, 0, 0]
    bad_any = 0
    for word, positions in coef.items():
        s = [0, 0, 0, 0]
        for i in positions:
            for b in range(4):
                s[b] ^= plane[b] >> i
            bad_any |= bad >> i
        for b in range(4):
            for j in range(3):
                if (word[b] >> j) & 1:
                    cls[j] ^= s[b]
    delim_any, bad_span, p = delim, bad, 1
    while 2 * p <= length:
        delim_any |= delim_any >> p
        bad_span |= bad_span >> p
        p *= 2
    delim_any |= delim_any >> (length - p)
    bad_span |= bad_span >> (length - p)
    lo = 0xffff if first <= 0 else 0 if first >= 16 else (0xffff << first) & 0xffff
    hi = 0xffff if last >= 16 else 0 if last <= 0 else (1 << last) - 1
    inside = lo & hi & ~delim_any & 0xffff
    ok = inside & ~(bad_span if hashed else bad_any)
    maps = []
    for c in range(8):
        eq = 0xffffffff
        for j in range(3):
            eq &= cls[j] if (c >> j) & 1 else ~cls[j]
        maps.append(ok & eq & 0xffff)
    return maps, inside & ~ok & 0xffff


def _classify_plain(codes, delim, bad, care, length, hashed, first, last):
    maps, special = [0] * 8, 0
    span = (1 << length) - 1
    care_mask = sum(1 << i for i in care)
    for w in range(16):
        inside = first <= w < last and ((delim >> w) & span) == 0
        ok = inside and ((bad >> w) & (span if hashed else care_mask)) == 0
        key = 0
        for i in care:
            key |= codes[w + i] << (4 * i)
        if ok:
            maps[seed_class(key)] |= 1 << w
        if hashed and inside and not ok:
            special |= 1 << w
    return maps, special if hashed else 0


def test_bit_plane_classifier_equals_the_window_by_window_form():
    rng = np.random.default_rng(7)
    shapes = [([0, 1, 2, 4, 7, 9, 10], 11), ([0, 1, 3, 4, 8, 11, 12, 14], 15), ([0, 2, 3, 5, 8, 9, 11, 13, 15], 16), (list(range(12)), 12), ([0], 1), ([0, 8], 9)]
    for care, length in shapes:
        for hashed in (False, True):
            for _ in range(300):
                codes = rng.integers(0, 11, size=32).tolist()
                delim = int(rng.integers(0, 1 << 32)) & int(rng.integers(0, 1 << 32)) & int(rng.integers(0, 1 << 32)) & int(rng.integers(0, 1 << 32))
                bad = int(rng.integers(0, 1 << 32)) & int(rng.integers(0, 1 << 32)) & int(rng.integers(0, 1 << 32))
                if hashed:                                   # seed_codes_kernel stores code 0 for a mask / stop letter in this mode
                    codes = [0 if (bad >> t) & 1 else c for t, c in enumerate(codes)]
                first, last = int(rng.integers(-3, 6)), int(rng.integers(8, 40))
                got = _classify_planes(codes, delim, bad, care, length, hashed, first, last)
                want = _classify_plain(codes, delim, bad, care, length, hashed, first, last)
                assert got[0] == want[0] and (got[1] if hashed else 0) == want[1], (care, hashed, codes, delim, bad, first, last)


L_DELIM, LETTER_MASK = 31, 31


def _score_plain(q, s, matrix, window):
    """stage2_score's window part as seed_core.h has it: clip_window(q - window, 2 window, window), then ungapped_window_score."""
    seq = q[48 - window: 48 + window]
    b, e = 0, 2 * window
    for i, c in enumerate(seq):
        if c == L_DELIM:
            if i >= window:
                e = i
                break
            b = i + 1
    st = score = 0
    for n in range(48 - window + b, 48 - window + e):
        st = max(st + int(matrix[q[n] & LETTER_MASK, s[n] & LETTER_MASK]), 0)
        score = max(score, st)
    return score


def _score_regs(q, s, matrix, window):
    mask = sum(1 << i for i in range(96) if q[i] == L_DELIM)
    lo, hi = mask & M64, mask >> 64
    before = lo & ((1 << 48) - 1) & ~((1 << (48 - window)) - 1)
    begin = before.bit_length() if before else 48 - window
    behind = ((lo >> 48) | (hi << 16)) & ((1 << window) - 1)
    end = 48 + ((behind & -behind).bit_length() - 1) if behind else 48 + window
    st = score = 0
    for n in range(96):
        m = int(matrix[q[n] & LETTER_MASK, s[n] & LETTER_MASK])
        st += 0 if n < begin else m if n < end else -(1 << 20)
        st = max(st, 0)
        score = max(score, st)
    return score


def test_register_form_of_the_stage2_window_score():
    rng = np.random.default_rng(3)
    matrix = rng.integers(-6, 12, size=(32, 32))
    matrix = (matrix + matrix.T) // 2
    for _ in range(3000):
        window = int(rng.choice([48, 48, 48, 40, 17, 1, 0]))
        q = rng.integers(0, 25, size=96)
        s = rng.integers(0, 25, size=96)
        for p in rng.integers(0, 96, size=int(rng.integers(0, 4))):      # a few delimiters, either side of the anchor
            q[p] = L_DELIM
        if rng.random() < 0.3:
            q[rng.integers(0, 96)] |= 0x80                               # a soft-masked letter is not a delimiter; its code is its low five bits
        q = q.astype(np.uint8).tolist()
        s = s.astype(np.uint8).tolist()
        assert _score_regs(q, s, matrix, window) == _score_plain(q, s, matrix, window), (window, q)


def _lower_bound64(keys, lo, hi, key):
    while hi - lo > 64:
        chunk = (hi - lo + 63) >> 6
        c = sum(1 for lane in range(64) if lo + lane * chunk < hi and keys[lo + lane * chunk] < key)
        new_hi = min(lo + c * chunk, hi) if c < 64 else hi
        if c > 0:
            lo, hi = lo + (c - 1) * chunk + 1, new_hi
        else:
            hi = lo
    return lo + sum(1 for lane in range(64) if lo + lane < hi and keys[lo + lane] < key)


def test_lower_bound_with_64_probes_per_round():
    rng = np.random.default_rng(5)
    for n in (0, 1, 63, 64, 65, 4095, 4096, 4097, 250_001):
        keys = np.sort(rng.integers(0, max(4 * n, 8), size=n)).tolist()
        probes = [-1, 0, max(4 * n, 8) + 1] + rng.integers(0, max(4 * n, 8), size=60).tolist() + (keys[:: max(1, n // 40)] if n else [])
        for key in probes:
            assert _lower_bound64(keys, 0, n, key) == bisect.bisect_left(keys, key)
        if n > 100:                                           # a sub-range, as the kernel's second and third searches use
            assert _lower_bound64(keys, 37, n - 11, keys[n // 2]) == bisect.bisect_left(keys, keys[n // 2], 37, n - 11)
