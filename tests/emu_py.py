"""TEST INFRASTRUCTURE ONLY -- ctypes view of tests/emu/libswipe_emu.so (CPU lane-emulator of the HIP
wavefront schedule built from diamond_amd/csrc/swipe_core.h)."""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "emu", "swipe_emu.cpp")
_SRC2 = os.path.join(_HERE, "emu", "seed_emu.cpp")
_CORE = os.path.join(_HERE, "..", "diamond_amd", "csrc", "swipe_core.h")
_CORE2 = os.path.join(_HERE, "..", "diamond_amd", "csrc", "seed_core.h")
_SRC3 = os.path.join(_HERE, "emu", "gapped_emu.cpp")
_CORE3 = os.path.join(_HERE, "..", "diamond_amd", "csrc", "gapped_core.h")
_SRC4 = os.path.join(_HERE, "emu", "mask_emu.cpp")
_CORE4 = os.path.join(_HERE, "..", "diamond_amd", "csrc", "mask_core.h")
_SRC5 = os.path.join(_HERE, "emu", "bias_emu.cpp")
_SRC6 = os.path.join(_HERE, "emu", "swipe16_emu.cpp")
_CORE6 = os.path.join(_HERE, "..", "diamond_amd", "csrc", "swipe16_core.h")
_SO = os.path.join(_HERE, "emu", "libswipe_emu.so")


class EmuOut(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                "score q_begin q_end s_begin s_end length identities mismatches positives gap_openings gaps "
                "transcript_len status".split()]


_lib = None


def lib():
    global _lib
    if _lib is None:
        import glob
        srcs = sorted(glob.glob(os.path.join(_HERE, "emu", "*.cpp")))
        deps = srcs + glob.glob(os.path.join(_HERE, "..", "diamond_amd", "csrc", "*_core.h"))
        if not os.path.exists(_SO) or max(os.path.getmtime(f) for f in deps) > os.path.getmtime(_SO):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-w", "-o", _SO] + srcs)
        _lib = ctypes.CDLL(_SO)
    return _lib


class Emu16Item(ctypes.Structure):
    _fields_ = [("q", ctypes.c_void_p), ("qlen", ctypes.c_int32), ("cbs", ctypes.c_void_p), ("t", ctypes.c_void_p), ("tlen", ctypes.c_int32),
                ("d_begin", ctypes.c_int32), ("d_end", ctypes.c_int32)]


def banded_swipe16(a, b, matrix8, gap_open, gap_extend, trace=True, force_p=0, cap=1 << 17):
    """Two work items (dicts: query, cbs, target, d_begin, d_end) through the packed-int16 two-items-per-wavefront emulator.
    trace: True = traceback mode, False = end cells, "score" = scores only (one packed max per cell instead of the end-cell keys)."""
    keep = []

    def item(x):
        q = np.ascontiguousarray(x["query"], dtype=np.int8)
        t = np.ascontiguousarray(x["target"], dtype=np.int8)
        c = np.ascontiguousarray(x["cbs"], dtype=np.int8) if x.get("cbs") is not None else None
        keep.extend([q, t, c])
        return Emu16Item(q.ctypes.data, len(q), c.ctypes.data if c is not None else None, t.ctypes.data, len(t), int(x["d_begin"]), int(x["d_end"]))

    ia, ib = item(a), item(b)
    m = np.ascontiguousarray(matrix8, dtype=np.int8)
    oa, ob = EmuOut(), EmuOut()
    tra, trb = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
    rc = lib().emu_banded_swipe16(ctypes.byref(ia), ctypes.byref(ib), m.ctypes.data_as(ctypes.c_void_p), int(gap_open), int(gap_extend),
                                  2 if trace == "score" else int(bool(trace)), int(force_p), ctypes.byref(oa), ctypes.byref(ob),
                                  tra.ctypes.data_as(ctypes.c_void_p), trb.ctypes.data_as(ctypes.c_void_p), cap)
    da = {n: getattr(oa, n) for n, _ in EmuOut._fields_}
    db = {n: getattr(ob, n) for n, _ in EmuOut._fields_}
    return rc, (da, tra[:da["transcript_len"]].copy()), (db, trb[:db["transcript_len"]].copy())


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


# ---- seed stage ---------------------------------------------------------------------------------------------------
class SeedParams(ctypes.Structure):
    """Mirror of dmnd::SeedParams (diamond_amd/csrc/seed_core.h) = dmnd_seed_params (include/diamond_hip.h)."""
    _fields_ = [("n_shapes", ctypes.c_int32), ("shape_len", ctypes.c_int32 * 64), ("shape_weight", ctypes.c_int32 * 64),
                ("shape_mask", ctypes.c_uint32 * 64), ("shape_pos", (ctypes.c_int8 * 32) * 64),
                ("reduction", ctypes.c_int8 * 32), ("reduction_size", ctypes.c_int32),
                ("seedp_bits", ctypes.c_int32), ("index_chunks", ctypes.c_int32), ("hamming_filter_id", ctypes.c_int32),
                ("ungapped_window", ctypes.c_int32), ("left_most_interval", ctypes.c_int32),
                ("seed_complexity_cut", ctypes.c_double),
                ("use_ungapped", ctypes.c_int32), ("short_query_max_len", ctypes.c_int32), ("short_query_cutoff", ctypes.c_int32),
                ("cutoff_table", ctypes.c_int32 * 32), ("tile_size", ctypes.c_int32), ("simd_lanes", ctypes.c_int32),
                ("query_translated", ctypes.c_int32), ("seed_encoding", ctypes.c_int32), ("cutoff_table_short", ctypes.c_int32 * 32)]


HIT_DTYPE = np.dtype([("query", "<u4"), ("seed_offset", "<i4"), ("subject", "<i8"), ("score", "<i4"), ("pad", "<i4")])


def seed_params_from_tap(cfg):
    c = SeedParams()
    c.n_shapes = len(cfg["shapes"])
    for i, sh in enumerate(cfg["shapes"]):
        c.shape_len[i], c.shape_weight[i], c.shape_mask[i] = sh["length"], sh["weight"], sh["mask"]
        for k, p in enumerate(sh["positions"]):
            c.shape_pos[i][k] = p
    for i in range(32):
        c.reduction[i] = int(cfg["reduction"][i])
    c.reduction_size = int(max(cfg["reduction"][:20])) + 1
    c.seedp_bits, c.index_chunks, c.hamming_filter_id = cfg["seedp_bits"], cfg["index_chunks"], cfg["hamming_filter_id"]
    c.ungapped_window, c.left_most_interval = 48, 32
    c.seed_complexity_cut = cfg["seed_complexity_cut"]
    import oracle_py
    short, table = oracle_py.ungapped_cutoffs(cfg["ungapped_evalue"])
    c.use_ungapped = 1 if cfg["ungapped_evalue"] > 0 else 0
    c.short_query_max_len, c.short_query_cutoff = 60, short
    for i in range(32):
        c.cutoff_table[i] = table[i]
        c.cutoff_table_short[i] = table[i]            # ungapped_evalue_short == ungapped_evalue in every golden mode
    c.tile_size, c.simd_lanes = 1024, 32
    c.query_translated = 1 if cfg.get("query_contexts", 1) > 1 else 0
    c.seed_encoding = int(cfg.get("seed_encoding", 0))         # 1: taps minted with --algo 1 (the test sets it; the tap header has no such field)
    return c


def seed_search(c, qdata, qlimits, tdata, tlimits, cap=1 << 22, matrix8=None):
    qd = np.ascontiguousarray(qdata, dtype=np.int8)
    td = np.ascontiguousarray(tdata, dtype=np.int8)
    ql = np.ascontiguousarray(qlimits, dtype=np.int64)
    tl = np.ascontiguousarray(tlimits, dtype=np.int64)
    hits = np.zeros(cap, dtype=HIT_DTYPE)
    f = lib().emu_seed_search
    f.restype = ctypes.c_int64
    m = np.ascontiguousarray(matrix8 if matrix8 is not None else np.zeros(1024), dtype=np.int8)
    n = f(ctypes.byref(c), m.ctypes.data_as(ctypes.c_void_p), qd.ctypes.data_as(ctypes.c_void_p), ql.ctypes.data_as(ctypes.c_void_p),
          ctypes.c_int64(len(ql) - 1), td.ctypes.data_as(ctypes.c_void_p), tl.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(len(tl) - 1),
          hits.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(cap))
    assert n >= 0
    return hits[:n].copy()


# ---- gapped filter ----------------------------------------------------------------------------------------------
class GfParams(ctypes.Structure):
    _fields_ = [("diag_score", ctypes.c_int32), ("gap_open", ctypes.c_int32), ("gap_extend", ctypes.c_int32),
                ("window2", ctypes.c_int32), ("use_cbs", ctypes.c_int32), ("contexts", ctypes.c_int32)]


def gapped_filter_hit(p, matrix8, query, cbs, target, hit_i, hit_j, cutoff1, cutoff2):
    """(flag, f1, f2) of one seed hit through the emulated kernel code."""
    m = np.ascontiguousarray(matrix8, np.int8)
    q = np.ascontiguousarray(query, np.int8)
    t = np.ascontiguousarray(target, np.int8)
    c = np.ascontiguousarray(cbs if cbs is not None else np.zeros(len(q)), np.int8)
    f = (ctypes.c_int * 2)()
    v = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    flag = lib().emu_gapped_filter_hit(ctypes.byref(p), v(m), v(q), len(q), v(c), v(t), len(t), int(hit_i), int(hit_j),
                                       int(cutoff1), int(cutoff2), f)
    return flag, f[0], f[1]


# ---- tantan masking ---------------------------------------------------------------------------------------------
class TantanParams(ctypes.Structure):
    _fields_ = [("p_repeat_end", ctypes.c_float), ("b2b", ctypes.c_float), ("f2f", ctypes.c_float), ("p_mask", ctypes.c_float),
                ("d", ctypes.c_float * 50)]


def tantan_params(p_repeat=0.005, p_repeat_end=0.05, growth=1.0 / 0.9, p_mask=0.9):
    f32 = np.float32
    p = TantanParams()
    p.p_repeat_end = p_repeat_end
    p.b2b = float(f32(1.0) - f32(p_repeat))
    p.f2f = float(f32(1.0) - f32(p_repeat_end))
    p.p_mask = p_mask
    g = f32(growth)
    b2f0 = f32(p_repeat) * (f32(1.0) - g) / (f32(1.0) - f32(np.power(g, f32(50.0), dtype=np.float32)))
    d = np.zeros(50, np.float32)
    d[49] = b2f0
    for i in range(48, -1, -1):
        d[i] = d[i + 1] * g
    for i in range(50):
        p.d[i] = float(d[i])
    return p


def tantan_mask(p, lr, seq):
    """Masked copy of seq through the emulated kernel, and the number of masked positions."""
    s = np.ascontiguousarray(seq, np.int8).copy()
    m = np.ascontiguousarray(lr, np.float32)
    n = lib().emu_tantan_mask(ctypes.byref(p), m.ctypes.data_as(ctypes.c_void_p), s.ctypes.data_as(ctypes.c_void_p), len(s))
    return s, n


def hauser_bias(seq, matrix8, bg, window=40):
    """Closed-form Hauser bias of one sequence (bias_core.h: the per-position code of hauser_bias_kernel)."""
    s = np.ascontiguousarray(seq, dtype=np.int8)
    m = np.ascontiguousarray(matrix8, dtype=np.int8)
    b = np.ascontiguousarray(bg, dtype=np.float32)
    out = np.zeros(len(s), np.int8)
    lib().emu_hauser_bias.restype = None
    lib().emu_hauser_bias(s.ctypes.data_as(ctypes.c_void_p), len(s), m.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                          int(window), out.ctypes.data_as(ctypes.c_void_p))
    return out
