"""TEST INFRASTRUCTURE ONLY -- ctypes view of the CPU restatement (oracle/*.c -> oracle/_ref/libswipe_oracle.so).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libswipe_oracle.so")

SCORE_ONLY, COORDS, TRACEBACK, STATS_FWD, STATS_BWD = range(5)

# BLOSUM62, gap open 11 / extend 1: gapped and ungapped Gumbel constants
# (/root/reference/src/stats/matrices/blosum62.h rows {11,1} and 0; the published NCBI values)
BLOSUM62_11_1 = dict(lambda_=0.267, K=0.041, alpha=1.9, alpha_v=42.6028, sigma=43.6362, u_alpha=0.7916, u_alpha_v=4.96466)


class Hsp(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                "score max_col max_band_row cols q_begin q_end s_begin s_end length identities mismatches "
                "positives gap_openings gaps transcript_len".split()]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Evaluer(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in
                "lambda_ K a_I b_I a_J b_J alpha_I beta_I alpha_J beta_J sigma tau vi_y_thr vj_y_thr c_y_thr db_letters ln_k".split()]


_lib = None


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
        if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
            subprocess.check_call(["make", "-s", "-C", _HERE, "restatement"])
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_evalue.restype = ctypes.c_double
        _lib.oracle_bitscore.restype = ctypes.c_double
        _lib.oracle_area.restype = ctypes.c_double
    return _lib


def _p8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)) if a is not None else None


def banded_swipe(query, cbs, target, d_begin, d_end, matrix8, gap_open, gap_extend, mode, transcript_cap=1 << 17):
    """One target through the restated banded sweep; returns (rc, Hsp dict, transcript uint8[])."""
    q = np.ascontiguousarray(query, dtype=np.int8)
    t = np.ascontiguousarray(target, dtype=np.int8)
    c = np.ascontiguousarray(cbs, dtype=np.int8) if cbs is not None else None
    m = np.ascontiguousarray(matrix8, dtype=np.int8)
    out = Hsp()
    tr = np.zeros(transcript_cap if mode == TRACEBACK else 1, np.uint8)
    rc = lib().oracle_banded_swipe(_p8(q), len(q), _p8(c), _p8(t), len(t), int(d_begin), int(d_end), _p8(m),
                                   int(gap_open), int(gap_extend), int(mode), ctypes.byref(out),
                                   tr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), len(tr))
    return rc, out.asdict(), tr[:out.transcript_len].copy()


def set_channel_band(own_d_begin=0, own_d_end=0, penalty=0):
    """Model of one channel of the reference's 8-bit vector pass inside a wider vector band (oracle/banded_swipe.c); penalty 0 = off."""
    lib().oracle_set_channel_band(int(own_d_begin), int(own_d_end), int(penalty))


def swipe_stats(query, cbs, target, d_begin, d_end, matrix8, gap_open, gap_extend, hsp_values):
    q = np.ascontiguousarray(query, dtype=np.int8)
    t = np.ascontiguousarray(target, dtype=np.int8)
    c = np.ascontiguousarray(cbs, dtype=np.int8) if cbs is not None else None
    m = np.ascontiguousarray(matrix8, dtype=np.int8)
    out = Hsp()
    rc = lib().oracle_swipe_stats(_p8(q), len(q), _p8(c), _p8(t), len(t), int(d_begin), int(d_end), _p8(m),
                                  int(gap_open), int(gap_extend), ctypes.c_uint(hsp_values), ctypes.byref(out))
    return rc, out.asdict()


def banded_cols(qlen, tlen, d_begin, d_end):
    return lib().oracle_banded_cols(int(qlen), int(tlen), int(d_begin), int(d_end))


def evaluer(db_letters, gap_open=11, gap_extend=1, consts=BLOSUM62_11_1):
    e = Evaluer()
    d = ctypes.c_double
    lib().oracle_evalue_init(ctypes.byref(e), d(consts["lambda_"]), d(consts["K"]), d(consts["alpha"]),
                             d(consts["alpha_v"]), d(consts["sigma"]), d(consts["u_alpha"]), d(consts["u_alpha_v"]),
                             int(gap_open), int(gap_extend), d(db_letters))
    return e


def evalue(e, score, qlen, slen):
    return lib().oracle_evalue(ctypes.byref(e), int(score), ctypes.c_uint(qlen), ctypes.c_uint(slen))


def bitscore(e, score):
    return lib().oracle_bitscore(ctypes.byref(e), ctypes.c_double(score))


# ---- seed stage -------------------------------------------------------------------------------------------------
class SeedCfg(ctypes.Structure):
    _fields_ = [("seedp_bits", ctypes.c_int32), ("index_chunks", ctypes.c_int32), ("hamming_filter_id", ctypes.c_int32),
                ("n_shapes", ctypes.c_int32), ("shape_len", ctypes.c_int32 * 16), ("shape_weight", ctypes.c_int32 * 16),
                ("shape_mask", ctypes.c_uint32 * 16), ("shape_pos", (ctypes.c_int32 * 32) * 16),
                ("reduction", ctypes.c_int32 * 32), ("reduction_size", ctypes.c_int32),
                ("ungapped_window", ctypes.c_int32), ("left_most_interval", ctypes.c_int32),
                ("seed_complexity_cut", ctypes.c_double),
                ("use_ungapped", ctypes.c_int32), ("short_query_max_len", ctypes.c_int32), ("short_query_cutoff", ctypes.c_int32),
                ("cutoff_table", ctypes.c_int32 * 32), ("tile_size", ctypes.c_int32), ("simd_lanes", ctypes.c_int32),
                ("matrix", ctypes.c_int8 * 1024), ("query_translated", ctypes.c_int32), ("cutoff_table_short", ctypes.c_int32 * 32),
                ("seed_encoding", ctypes.c_int32)]


def ungapped_cutoffs(ungapped_evalue, lambda_=0.267, K=0.041, short_bits=25.0):
    """(short-query cutoff, CutoffTable): rawscore(bits) = ceil((bits*ln2 + ln K)/lambda) (stats/score_matrix.h:130-134),
    CutoffTable[b] = rawscore(-log(evalue / 1e9 / 2^(b-1)) / log 2) (util/scores/cutoff_table.h:30-35)."""
    import math

    def raw(bits):
        return int(math.ceil((bits * 0.69314718055994530941723212145818 + math.log(K)) / lambda_))
    table = [0] * 32
    if ungapped_evalue > 0:
        for b in range(1, 32):
            table[b] = raw(-math.log(ungapped_evalue / 1e9 / (1 << (b - 1))) / math.log(2))
    return raw(short_bits), table


HIT_DTYPE = np.dtype([("query", "<u4"), ("seed_offset", "<i4"), ("subject", "<i8"), ("score", "<i4"), ("pad", "<i4")])


def seed_cfg_from_tap(cfg, matrix8=None):
    """Builds the oracle's configuration from the 'BLK1' header of an extend tap (tapfile.read_ext_tap)."""
    c = SeedCfg()
    c.seedp_bits, c.index_chunks, c.hamming_filter_id = cfg["seedp_bits"], cfg["index_chunks"], cfg["hamming_filter_id"]
    c.n_shapes = len(cfg["shapes"])
    for i, sh in enumerate(cfg["shapes"]):
        c.shape_len[i], c.shape_weight[i], c.shape_mask[i] = sh["length"], sh["weight"], sh["mask"]
        for k, p in enumerate(sh["positions"]):
            c.shape_pos[i][k] = p
    for i in range(32):
        c.reduction[i] = int(cfg["reduction"][i])
    c.reduction_size = int(max(cfg["reduction"][:20])) + 1
    c.ungapped_window, c.left_most_interval = 48, 32
    c.seed_complexity_cut = cfg["seed_complexity_cut"]
    c.use_ungapped = 1 if cfg["ungapped_evalue"] > 0 else 0
    c.short_query_max_len = 60
    short, table = ungapped_cutoffs(cfg["ungapped_evalue"])
    c.short_query_cutoff = short
    for i in range(32):
        c.cutoff_table[i] = table[i]
        c.cutoff_table_short[i] = table[i]
    c.tile_size, c.simd_lanes = 1024, 32
    c.query_translated = 1 if cfg.get("query_contexts", 1) > 1 else 0
    if matrix8 is not None:
        m = np.ascontiguousarray(matrix8, dtype=np.int8).ravel()
        for i in range(1024):
            c.matrix[i] = int(m[i])
    return c


def seed_search(c, qdata, qlimits, tdata, tlimits, cap=1 << 22):
    qd = np.ascontiguousarray(qdata, dtype=np.int8).copy()
    td = np.ascontiguousarray(tdata, dtype=np.int8)
    ql = np.ascontiguousarray(qlimits, dtype=np.int64)
    tl = np.ascontiguousarray(tlimits, dtype=np.int64)
    hits = np.zeros(cap, dtype=HIT_DTYPE)
    f = lib().oracle_seed_search
    f.restype = ctypes.c_int64
    n = f(ctypes.byref(c), qd.ctypes.data_as(ctypes.c_void_p), ql.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(len(ql) - 1),
          td.ctypes.data_as(ctypes.c_void_p), tl.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(len(tl) - 1),
          hits.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(cap))
    if n < 0:
        raise RuntimeError("hit buffer too small")
    return hits[:n].copy()


# ---- gapped filter ----------------------------------------------------------------------------------------------
def cutoff_table2d(evalue_threshold, gap_open=11, gap_extend=1):
    """CutoffTable2D(evalue) as a 32x32 int32 array indexed [bit_length(qlen)][bit_length(slen)]."""
    e = evaluer(1e9, gap_open, gap_extend)
    t = np.zeros(32 * 32, np.int32)
    lib().oracle_cutoff_table2d(ctypes.byref(e), ctypes.c_double(evalue_threshold), t.ctypes.data_as(ctypes.c_void_p))
    return t.reshape(32, 32)


def gapped_filter_target(matrix8, query, cbs, target, hits_ij, cutoff1, cutoff2, window2=200, diag_score=None, gap_open=11, gap_extend=1):
    """True if the target survives the gapped filter for this query (any seed hit passes both stages)."""
    m = np.ascontiguousarray(matrix8, np.int8)
    q = np.ascontiguousarray(query, np.int8)
    t = np.ascontiguousarray(target, np.int8)
    c = None if cbs is None else np.ascontiguousarray(cbs, np.int8)
    hi = np.ascontiguousarray(hits_ij[:, 0], np.int32)
    hj = np.ascontiguousarray(hits_ij[:, 1], np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
    return bool(lib().oracle_gapped_filter_target(p(m), p(q), len(q), p(c), p(t), len(t), p(hi), p(hj), len(hi), int(cutoff1), int(cutoff2),
                                                  int(window2), int(diag_score), int(gap_open), int(gap_extend)))


def gapped_filter_hit(matrix8, query, cbs, target, hit_i, hit_j, band, window, diag_score, gap_open=11, gap_extend=1):
    """diag_alignment(scan_diags<band>(...)) of one seed hit (gapped_filter.cpp:33-41)."""
    m = np.ascontiguousarray(matrix8, np.int8)
    q = np.ascontiguousarray(query, np.int8)
    t = np.ascontiguousarray(target, np.int8)
    c = None if cbs is None else np.ascontiguousarray(cbs, np.int8)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
    return int(lib().oracle_gapped_filter_hit(p(m), p(q), len(q), p(c), p(t), len(t), int(hit_i), int(hit_j), int(band), int(window),
                                              int(diag_score), int(gap_open), int(gap_extend)))


# ---- tantan masking ---------------------------------------------------------------------------------------------
def tantan_matrix(matrix8):
    """32x32 float32 likelihood-ratio matrix exp(lambda * score) as Masking::Masking builds it."""
    m = np.ascontiguousarray(matrix8, np.int8)
    lr = np.zeros(1024, np.float32)
    lib().oracle_tantan_matrix(m.ctypes.data_as(ctypes.c_void_p), lr.ctypes.data_as(ctypes.c_void_p))
    return lr.reshape(32, 32)


def tantan_lambda(matrix8):
    m = np.ascontiguousarray(matrix8, np.int8)
    lib().oracle_tantan_lambda.restype = ctypes.c_double
    return lib().oracle_tantan_lambda(m.ctypes.data_as(ctypes.c_void_p))


def tantan_mask(seq, lr, p_repeat=0.005, p_repeat_end=0.05, repeat_growth=1.0 / 0.9, p_mask=0.9):
    """Hard-masks a copy of seq; returns (masked int8[], number of masked letters)."""
    s = np.ascontiguousarray(seq, np.int8).copy()
    m = np.ascontiguousarray(lr, np.float32)
    f = ctypes.c_float
    n = lib().oracle_tantan_mask(s.ctypes.data_as(ctypes.c_void_p), len(s), m.ctypes.data_as(ctypes.c_void_p),
                                 f(p_repeat), f(p_repeat_end), f(repeat_growth), f(p_mask))
    return s, n


def join_blocks(per_block, max_target_seqs=25):
    """join_query of the reference for one query (output/join_blocks.cpp:180-256): `per_block` = one list per reference block
    of (evalue, score, target_oid) in that block's output order. BlockJoiner keeps one cursor per block and a heap of the
    cursors' heads ordered by JoinRecord::cmp_evalue (:129-137: e-value ascending, score descending, target ordinal
    ascending); GlobalCulling stops after max_target_seqs targets (target_culling.h:70-88). Returns the joined list."""

    def heap_less(a, b):                      # cmp_evalue(lhs, rhs): true when lhs must sink below rhs
        (ea, sa, ta), (eb, sb, tb) = a[0], b[0]
        return ea > eb or (ea == eb and (sa < sb or (sa == sb and tb < ta)))

    heads = [[blk[0], bi, 0] for bi, blk in enumerate(per_block) if len(blk)]
    out = []
    while heads and len(out) < max_target_seqs:
        top = heads[0]
        for h in heads[1:]:                   # top of a max-heap under heap_less = the element no other one sinks
            if heap_less(top, h):
                top = h
        out.append(top[0])
        bi, pos = top[1], top[2] + 1
        heads.remove(top)
        if pos < len(per_block[bi]):
            heads.append([per_block[bi][pos], bi, pos])
    return out


class Hsp3(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                "score frame q_begin q_end s_begin s_end qs_begin qs_end length identities mismatches positives gap_openings gaps "
                "transcript_len".split()]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def _frames(frames):
    fr = [np.ascontiguousarray(f, dtype=np.int8) for f in frames]
    ptrs = (ctypes.POINTER(ctypes.c_int8) * 3)(*[_p8(f) if len(f) else ctypes.cast(0, ctypes.POINTER(ctypes.c_int8)) for f in fr])
    lens = (ctypes.c_int32 * 3)(*[len(f) for f in fr])
    return fr, ptrs, lens


def frameshift_batches(targets, channels=16, band_bin=24, col_bin=400):
    """The vector batches of the reference's score-only three-frame sweep: the DpTargets of one strand in the order
    std::stable_sort(DpTarget::operator<) leaves them (dp/dp.h:105-111: band / band_bin, cols / col_bin, left_i1), `channels` at a
    time (banded_3frame_swipe.cpp:533-551; 16 int16 channels with AVX2). targets: dicts with d_begin, d_end, cols.
    Yields lists of (index into targets, band, i0, i1, pos0)."""
    key = lambda t: ((t["d_end"] - t["d_begin"]) // band_bin, _floordiv_c(t["cols"], col_bin), max(t["d_end"] - 1, 0))
    order = sorted(range(len(targets)), key=lambda k: key(targets[k]))
    for b in range(0, len(order), channels):
        idx = order[b:b + channels]
        band = max(targets[k]["d_end"] - targets[k]["d_begin"] for k in idx)
        i1 = min(max(targets[k]["d_end"] - 1, 0) for k in idx)
        i0 = i1 + 1 - band
        yield [(k, band, i0, i1, i1 - (targets[k]["d_end"] - 1)) for k in idx]


def _floordiv_c(a, b):
    """C integer division (truncation toward zero): DpTarget::cols may be negative here."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def frameshift_score(frames, target, band, i0, i1, pos0, matrix8, gap_open, gap_extend, frame_shift):
    """-> (score, max_col, overflow) of one channel of a score-only batch."""
    fr, ptrs, lens = _frames(frames)
    t = np.ascontiguousarray(target, dtype=np.int8)
    m = np.ascontiguousarray(matrix8, dtype=np.int8)
    mc, ov = ctypes.c_int(0), ctypes.c_int(0)
    s = lib().oracle_3frame_score(ptrs, lens, _p8(t), len(t), int(band), int(i0), int(i1), int(pos0), _p8(m), int(gap_open), int(gap_extend),
                                  int(frame_shift), ctypes.byref(mc), ctypes.byref(ov))
    return s, mc.value, ov.value


def frameshift_score_range(strand, dna_len, qlen, band, i0, pos0, max_col):
    out = Hsp3()
    lib().oracle_3frame_score_range(int(strand), int(dna_len), int(qlen), int(band), int(i0), int(pos0), int(max_col), ctypes.byref(out))
    return out.asdict()


def frameshift_traceback(frames, strand, dna_len, target, d_begin, d_end, matrix8, gap_open, gap_extend, frame_shift):
    """-> (rc, Hsp3 dict, transcript uint8[]) of one target on its own band."""
    fr, ptrs, lens = _frames(frames)
    t = np.ascontiguousarray(target, dtype=np.int8)
    m = np.ascontiguousarray(matrix8, dtype=np.int8)
    out = Hsp3()
    cap = 4 * (len(t) + len(fr[0])) + 64
    tr = np.zeros(cap, np.uint8)
    rc = lib().oracle_3frame_traceback(ptrs, lens, int(strand), int(dna_len), _p8(t), len(t), int(d_begin), int(d_end), _p8(m),
                                       int(gap_open), int(gap_extend), int(frame_shift), ctypes.byref(out),
                                       tr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), cap)
    return rc, out.asdict(), tr[:out.transcript_len].copy()
