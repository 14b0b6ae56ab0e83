"""TEST INFRASTRUCTURE ONLY -- ctypes view of tests/emu/libswipe_emu.so (CPU lane-emulator of the HIP
wavefront schedule built from diamond_amd/csrc/swipe_core.h)."""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "emu", "swipe_emu.cpp")
_CORE = os.path.join(_HERE, "..", "diamond_amd", "csrc", "swipe_core.h")
_SO = os.path.join(_HERE, "emu", "libswipe_emu.so")


class EmuOut(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                "score q_begin q_end s_begin s_end length identities mismatches positives gap_openings gaps "
                "transcript_len status".split()]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or max(os.path.getmtime(_SRC), os.path.getmtime(_CORE)) > os.path.getmtime(_SO):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-o", _SO, _SRC])
        _lib = ctypes.CDLL(_SO)
    return _lib


def banded_swipe(query, cbs, target, d_begin, d_end, matrix8, gap_open, gap_extend, mode, force_p=0, cap=1 << 17):
    p8 = ctypes.POINTER(ctypes.c_int8)
    q = np.ascontiguousarray(query, dtype=np.int8)
    t = np.ascontiguousarray(target, dtype=np.int8)
    c = np.ascontiguousarray(cbs, dtype=np.int8) if cbs is not None else None
    m = np.ascontiguousarray(matrix8, dtype=np.int8)
    out = EmuOut()
    tr = np.zeros(cap, np.uint8)
    rc = lib().emu_banded_swipe(q.ctypes.data_as(p8), len(q), c.ctypes.data_as(p8) if c is not None else None,
                                t.ctypes.data_as(p8), len(t), int(d_begin), int(d_end), m.ctypes.data_as(p8),
                                int(gap_open), int(gap_extend), int(mode), int(force_p), ctypes.byref(out),
                                tr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), cap)
    o = {n: getattr(out, n) for n, _ in EmuOut._fields_}
    return rc, o, tr[:o["transcript_len"]].copy()


def swipe_stats(query, cbs, target, d_begin, d_end, matrix8, gap_open, gap_extend):
    p8 = ctypes.POINTER(ctypes.c_int8)
    q = np.ascontiguousarray(query, dtype=np.int8)
    t = np.ascontiguousarray(target, dtype=np.int8)
    c = np.ascontiguousarray(cbs, dtype=np.int8) if cbs is not None else None
    m = np.ascontiguousarray(matrix8, dtype=np.int8)
    out = EmuOut()
    rc = lib().emu_swipe_stats(q.ctypes.data_as(p8), len(q), c.ctypes.data_as(p8) if c is not None else None,
                               t.ctypes.data_as(p8), len(t), int(d_begin), int(d_end), m.ctypes.data_as(p8),
                               int(gap_open), int(gap_extend), ctypes.byref(out))
    return rc, {n: getattr(out, n) for n, _ in EmuOut._fields_}
