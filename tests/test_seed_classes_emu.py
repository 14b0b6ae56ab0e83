"""CPU check of round 4's key classes (seed_core.h seed_class; SeedArgs::home / bm1_index restated in tests/emu/seed_emu.cpp):
a class is a function of the table key alone, the eight classes are balanced on realistic keys (care-masked nibble windows of
reduced letters), and class c owns the c-th eighth of the slots and of the level-1 filter's words."""
import ctypes
import os

import numpy as np

EMU = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "libswipe_emu.so")


def _classes(keys, slot_mask, words):
    lib = ctypes.CDLL(EMU)
    n = len(keys)
    cls, home, word = np.zeros(n, np.uint32), np.zeros(n, np.uint64), np.zeros(n, np.uint32)
    lib.emu_seed_classes.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.emu_seed_classes(keys.ctypes.data, n, slot_mask, words, cls.ctypes.data, home.ctypes.data, word.ctypes.data)
    return cls, home, word


def test_key_classes_partition_slots_and_filter_words():
    rng = np.random.default_rng(4)
    n = 400_000
    # windows of 16 reduced letters (11 classes, skewed like amino-acid groups), care positions of a weight-8 shape
    p = np.array([0.16, 0.13, 0.12, 0.11, 0.10, 0.09, 0.08, 0.07, 0.06, 0.05, 0.03])
    nib = rng.choice(11, size=(n, 16), p=p).astype(np.uint64)
    care = [0, 1, 3, 4, 7, 9, 12, 14]
    keys = np.zeros(n, np.uint64)
    for k in care:
        keys |= nib[:, k] << np.uint64(4 * k)
    slots, words = 1 << 24, 3 * (1 << 18)            # 16 M slots, a 3 MB filter (786 432 words: not a power of two)
    cls, home, word = _classes(keys, slots - 1, words)
    assert cls.max() <= 7
    share = np.bincount(cls, minlength=8) / n
    assert share.min() > 0.105 and share.max() < 0.145, share          # eighths, within a sixth
    # the same key always gets the same class / home / word
    again = _classes(keys[:1000].copy(), slots - 1, words)
    assert np.array_equal(again[0], cls[:1000]) and np.array_equal(again[1], home[:1000]) and np.array_equal(again[2], word[:1000])
    # class c owns slots [c S/8, (c+1) S/8) and words [c W/8, (c+1) W/8)
    assert np.array_equal(home // np.uint64(slots // 8), cls.astype(np.uint64))
    assert np.array_equal(word // np.uint32(words // 8), cls)
    assert home.max() < slots and word.max() < words
    # inside a class the homes spread over the whole eighth
    for c in range(8):
        h = home[cls == c] - np.uint64(c * (slots // 8))
        assert h.min() < slots // 8 // 50 and h.max() > slots // 8 - slots // 8 // 50
