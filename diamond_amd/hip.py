"""ctypes host mirror of the C ABI in include/diamond_hip.h (libdiamond_hip.so).

This is plumbing only: it loads the in-tree HIP library and marshals numpy arrays. There is no CPU
fallback -- if the library is missing or no gfx950 device is visible, calls raise DiamondHipError.
The method names mirror the reference's operator interface: `banded_swipe` is
DP::BandedSwipe::swipe (/root/reference/src/dp/dp.h:287)."""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdiamond_hip.so")

SWIPE_SCORE, SWIPE_COORDS, SWIPE_TRACEBACK, SWIPE_STATS = 0, 1, 2, 3
QUERY, TARGET = 0, 1


class DiamondHipError(RuntimeError):
    pass


class Params(ctypes.Structure):
    _fields_ = [("matrix8", ctypes.c_int8 * 1024), ("gap_open", ctypes.c_int32), ("gap_extend", ctypes.c_int32),
                ("lambda_", ctypes.c_double), ("K", ctypes.c_double), ("alpha", ctypes.c_double),
                ("alpha_v", ctypes.c_double), ("sigma", ctypes.c_double), ("u_alpha", ctypes.c_double),
                ("u_alpha_v", ctypes.c_double), ("db_letters", ctypes.c_double), ("max_evalue", ctypes.c_double)]


class SeedParams(ctypes.Structure):
    """dmnd_seed_params (include/diamond_hip.h)."""
    _fields_ = [("n_shapes", ctypes.c_int32), ("shape_len", ctypes.c_int32 * 64), ("shape_weight", ctypes.c_int32 * 64),
                ("shape_mask", ctypes.c_uint32 * 64), ("shape_pos", (ctypes.c_int8 * 32) * 64),
                ("reduction", ctypes.c_int8 * 32), ("reduction_size", ctypes.c_int32),
                ("seedp_bits", ctypes.c_int32), ("index_chunks", ctypes.c_int32), ("hamming_filter_id", ctypes.c_int32),
                ("ungapped_window", ctypes.c_int32), ("left_most_interval", ctypes.c_int32),
                ("seed_complexity_cut", ctypes.c_double),
                ("use_ungapped", ctypes.c_int32), ("short_query_max_len", ctypes.c_int32), ("short_query_cutoff", ctypes.c_int32),
                ("cutoff_table", ctypes.c_int32 * 32), ("tile_size", ctypes.c_int32), ("simd_lanes", ctypes.c_int32),
                ("query_translated", ctypes.c_int32), ("seed_encoding", ctypes.c_int32), ("cutoff_table_short", ctypes.c_int32 * 32)]


SEED_HIT_DTYPE = np.dtype([("query", "<u4"), ("seed_offset", "<i4"), ("subject", "<i8"), ("score", "<i4"), ("pad", "<i4")])

MATCH_DTYPE = np.dtype([("query", "<u4"), ("target", "<u4"), ("ungapped_score", "<i4"), ("d_begin", "<i4"), ("d_end", "<i4"),
                        ("frame", "<i4"), ("read_begin", "<i4"), ("read_end", "<i4"), ("evalue", "<f8"), ("bit_score", "<f8"),
                        ("hsp", [("score", "<i4"), ("q_begin", "<i4"), ("q_end", "<i4"), ("s_begin", "<i4"), ("s_end", "<i4"),
                                 ("length", "<i4"), ("identities", "<i4"), ("mismatches", "<i4"), ("positives", "<i4"),
                                 ("gap_openings", "<i4"), ("gaps", "<i4"), ("transcript_len", "<i4"), ("transcript_off", "<i8")])])
assert MATCH_DTYPE.itemsize == 104

PLAN_DTYPE = np.dtype([("query", "<u4"), ("target", "<u4"), ("d_begin", "<i4"), ("d_end", "<i4"), ("ungapped_score", "<i4")])

DP_TARGET_DTYPE = np.dtype([("query_off", "<i8"), ("target_off", "<i8"), ("cbs_off", "<i8"), ("query_len", "<i4"),
                            ("target_len", "<i4"), ("d_begin", "<i4"), ("d_end", "<i4")], align=True)
HSP_DTYPE = np.dtype([("score", "<i4"), ("q_begin", "<i4"), ("q_end", "<i4"), ("s_begin", "<i4"), ("s_end", "<i4"),
                      ("length", "<i4"), ("identities", "<i4"), ("mismatches", "<i4"), ("positives", "<i4"),
                      ("gap_openings", "<i4"), ("gaps", "<i4"), ("transcript_len", "<i4"), ("transcript_off", "<i8")],
                     align=True)
FS_TARGET_DTYPE = np.dtype([("frame_off", "<i8", (3,)), ("target_off", "<i8"), ("frame_len", "<i4", (3,)), ("target_len", "<i4"), ("d_begin", "<i4"),
                            ("d_end", "<i4"), ("cols", "<i4"), ("strand", "<i4"), ("dna_len", "<i4"), ("group", "<i4")], align=True)
FS_HSP_DTYPE = np.dtype([(n, "<i4") for n in "score frame q_begin q_end s_begin s_end read_begin read_end length identities mismatches positives "
                         "gap_openings gaps transcript_len max_col".split()] + [("transcript_off", "<i8")], align=True)
assert FS_TARGET_DTYPE.itemsize == 72 and FS_HSP_DTYPE.itemsize == 72
HOST_TARGET_DTYPE = np.dtype([("seq", "<u8"), ("len", "<i4"), ("d_begin", "<i4"), ("d_end", "<i4"), ("matrix", "<u8")], align=True)
assert DP_TARGET_DTYPE.itemsize == 40 and HSP_DTYPE.itemsize == 56 and HOST_TARGET_DTYPE.itemsize == 32

_lib = None

EXPORTS = ["dmnd_abi_version", "dmnd_last_error", "dmnd_default_params", "dmnd_create", "dmnd_destroy",
           "dmnd_set_db_letters", "dmnd_upload_block", "dmnd_upload_cbs", "dmnd_banded_swipe",
           "dmnd_banded_swipe_host", "dmnd_banded_cols", "dmnd_evalue", "dmnd_bitscore", "dmnd_evalue_p",
           "dmnd_bitscore_p", "dmnd_evalue_batch", "dmnd_last_kernel_ms", "dmnd_seed_params_fast", "dmnd_seed_params_default", "dmnd_seed_search",
           "dmnd_seed_hits", "dmnd_seed_kernel_ms", "dmnd_extend_plan", "dmnd_extend", "dmnd_extend_stats", "dmnd_extend_plan_stats", "dmnd_extend_device_stats", "dmnd_extend_reserve", "dmnd_extend_records_device", "dmnd_join_contexts_device", "dmnd_format_tab", "dmnd_set_max_target_seqs",
           "dmnd_seed_params_sensitive", "dmnd_set_gapped_filter", "dmnd_gapped_filter", "dmnd_gapped_filter_ms",
           "dmnd_set_query_contexts", "dmnd_translate", "dmnd_format_tab_translated", "dmnd_mask_block", "dmnd_mask_kernel_ms", "dmnd_seed_params_preset", "dmnd_set_comp_based_stats",
           "dmnd_seed_params_set_index_chunks", "dmnd_join_blocks", "dmnd_set_sensitivity", "dmnd_touch_streams",
           "dmnd_seed_params_set_query_indexed", "dmnd_auto_query_indexed", "dmnd_set_motif_table", "dmnd_motif_table_size",
           "dmnd_soft_mask_block", "dmnd_output_fields", "dmnd_format_fields", "dmnd_format_pairwise_intro", "dmnd_format_pairwise",
           "dmnd_format_paf", "dmnd_device_count", "dmnd_set_top_percent", "dmnd_join_blocks_top", "dmnd_set_filters", "dmnd_format_sam", "dmnd_set_query_source_lengths", "dmnd_format_fields_unaligned", "dmnd_format_fields_header", "dmnd_set_query_index_reuse", "dmnd_set_no_self_hits", "dmnd_matrix_params", "dmnd_masking_lambda", "dmnd_translate_opts", "dmnd_set_extension_mode", "dmnd_format_xml_header", "dmnd_format_xml_query_intro", "dmnd_format_xml", "dmnd_format_xml_query_epilog", "dmnd_format_daa_header", "dmnd_format_daa_query", "dmnd_format_daa_match", "dmnd_seg_ranges", "dmnd_seg_mask_block", "dmnd_seg_lnfact", "dmnd_daa_match_read", "dmnd_hsp_from_transcript", "dmnd_hsp_from_transcript_frames", "dmnd_set_format_flags", "dmnd_host_alloc", "dmnd_host_free", "dmnd_share_block", "dmnd_copy_block", "dmnd_init", "dmnd_seed_reserve", "dmnd_mask_sequences", "dmnd_set_max_hsps", "dmnd_rank_targets", "dmnd_rank_update", "dmnd_set_global_ranking",
           "dmnd_upload_matrices", "dmnd_frameshift_swipe", "dmnd_set_frameshift", "dmnd_set_context_motif_table", "dmnd_cbs_composition", "dmnd_cbs_rule", "dmnd_cbs_target_matrix", "dmnd_cbs_ideal_lambda", "dmnd_join_blocks_range", "dmnd_join_blocks_device", "dmnd_join_blocks_device_host", "dmnd_join_ranks", "dmnd_join_ranks_plan", "dmnd_xdrop_ungapped"]


def set_motif_table(codes):
    """Installs the process-wide motif table: uint64 codes of 8-letter motifs (base 20, first letter most significant); the
    reference's table is extracted at build time by tools/make_motif_table.py into diamond_amd/motifs.bin."""
    lib = load()
    a = np.ascontiguousarray(codes, dtype=np.uint64)
    lib.dmnd_set_motif_table.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    if lib.dmnd_set_motif_table(a.ctypes.data, len(a)) != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())


def load_motif_table(path=None):
    """set_motif_table() from diamond_amd/motifs.bin (fails loudly if build() has not produced it)."""
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "motifs.bin")
    set_motif_table(np.fromfile(path, dtype=np.uint64))


def load():
    """Loads libdiamond_hip.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DiamondHipError("libdiamond_hip.so is not built: run `make product` (needs hipcc)")
        lib = ctypes.CDLL(LIB_PATH)
        lib.dmnd_last_error.restype = ctypes.c_char_p
        lib.dmnd_create.restype = ctypes.c_void_p
        lib.dmnd_create.argtypes = [ctypes.c_int, ctypes.POINTER(Params)]
        lib.dmnd_destroy.argtypes = [ctypes.c_void_p]
        lib.dmnd_set_db_letters.argtypes = [ctypes.c_void_p, ctypes.c_double]
        lib.dmnd_upload_block.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
        lib.dmnd_upload_cbs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        lib.dmnd_banded_swipe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint32,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
        lib.dmnd_banded_swipe_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int64, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
        lib.dmnd_banded_cols.restype = ctypes.c_int32
        lib.dmnd_evalue.restype = ctypes.c_double
        lib.dmnd_evalue.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint32]
        lib.dmnd_bitscore.restype = ctypes.c_double
        lib.dmnd_bitscore.argtypes = [ctypes.c_void_p, ctypes.c_double]
        lib.dmnd_evalue_p.restype = ctypes.c_double
        lib.dmnd_evalue_p.argtypes = [ctypes.POINTER(Params), ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint32]
        lib.dmnd_bitscore_p.restype = ctypes.c_double
        lib.dmnd_bitscore_p.argtypes = [ctypes.POINTER(Params), ctypes.c_double]
        lib.dmnd_seed_params_fast.argtypes = [ctypes.POINTER(SeedParams), ctypes.c_int]
        lib.dmnd_seed_params_default.argtypes = [ctypes.POINTER(SeedParams), ctypes.c_int, ctypes.POINTER(Params)]
        lib.dmnd_seed_params_sensitive.argtypes = [ctypes.POINTER(SeedParams), ctypes.c_int, ctypes.POINTER(Params)]
        lib.dmnd_set_gapped_filter.argtypes = [ctypes.c_void_p, ctypes.c_double]
        lib.dmnd_gapped_filter.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.dmnd_gapped_filter_ms.argtypes = [ctypes.c_void_p]
        lib.dmnd_touch_streams.argtypes = [ctypes.c_void_p]
        lib.dmnd_gapped_filter_ms.restype = ctypes.c_double
        lib.dmnd_seed_search.argtypes = [ctypes.c_void_p, ctypes.POINTER(SeedParams), ctypes.POINTER(ctypes.c_int64)]
        lib.dmnd_seed_hits.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        lib.dmnd_seed_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        v = ctypes.c_void_p
        lib.dmnd_extend.argtypes = [v, v, v, v, ctypes.c_int64, ctypes.c_int, ctypes.c_uint32, v, ctypes.c_int64,
                                    ctypes.POINTER(ctypes.c_int64), v, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
        lib.dmnd_format_tab.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int64]
        lib.dmnd_format_tab_translated.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int32, ctypes.c_char_p, ctypes.c_int64]
        lib.dmnd_set_query_contexts.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.dmnd_mask_block.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
        lib.dmnd_mask_kernel_ms.argtypes = [ctypes.c_void_p]
        lib.dmnd_mask_kernel_ms.restype = ctypes.c_double
        lib.dmnd_masking_lambda.restype = ctypes.c_double
        lib.dmnd_masking_lambda.argtypes = [ctypes.c_void_p]
        lib.dmnd_translate.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        lib.dmnd_last_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        lib.dmnd_upload_matrices.argtypes = [v, v, ctypes.c_int64]
        lib.dmnd_frameshift_swipe.argtypes = [v, v, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, v, v, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
        lib.dmnd_cbs_ideal_lambda.restype = ctypes.c_double
        lib.dmnd_cbs_ideal_lambda.argtypes = [ctypes.POINTER(Params)]
        lib.dmnd_cbs_composition.argtypes = [v, ctypes.c_int32, v, ctypes.POINTER(ctypes.c_int32)]
        lib.dmnd_cbs_rule.argtypes = [ctypes.POINTER(Params), ctypes.c_int, v, ctypes.c_int32, v, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]
        lib.dmnd_cbs_target_matrix.argtypes = [ctypes.POINTER(Params), ctypes.c_int, v, ctypes.c_int32, v, ctypes.c_int32, v]
        _lib = lib
    return _lib


def cbs_composition(seq):
    """Stats::composition + count_true_aa (stats/cbs.cpp:52-77) -> (frequencies[20], residues)."""
    seq = np.ascontiguousarray(seq, dtype=np.int8)
    comp, n = np.zeros(20, np.float64), ctypes.c_int32(0)
    rc = load().dmnd_cbs_composition(seq.ctypes.data, len(seq), comp.ctypes.data, ctypes.byref(n))
    if rc != 0:
        raise DiamondHipError("dmnd_cbs_composition: %d" % rc)
    return comp, n.value


def cbs_rule(params, mode, query_comp, query_true_aa, target):
    """Stats::adjust_matrix (stats/cbs.cpp:94-112) -> -1 / 0 / 4."""
    qc, t, r = np.ascontiguousarray(query_comp, dtype=np.float64), np.ascontiguousarray(target, dtype=np.int8), ctypes.c_int32(99)
    rc = load().dmnd_cbs_rule(ctypes.byref(params), int(mode), qc.ctypes.data, int(query_true_aa), t.ctypes.data, len(t), ctypes.byref(r))
    if rc != 0:
        raise DiamondHipError("dmnd_cbs_rule: %d" % rc)
    return r.value


def cbs_target_matrix(params, rule, query_comp, query_true_aa, target):
    """Stats::TargetMatrix::TargetMatrix (stats/cbs.cpp:114-173) -> int8[32, 32], [target letter][query letter]."""
    qc, t = np.ascontiguousarray(query_comp, dtype=np.float64), np.ascontiguousarray(target, dtype=np.int8)
    out = np.zeros((32, 32), np.int8)
    rc = load().dmnd_cbs_target_matrix(ctypes.byref(params), int(rule), qc.ctypes.data, int(query_true_aa), t.ctypes.data, len(t), out.ctypes.data)
    if rc != 0:
        raise DiamondHipError("dmnd_cbs_target_matrix: %d" % rc)
    return out


def default_params():
    p = Params()
    rc = load().dmnd_default_params(ctypes.byref(p))
    if rc != 0:
        raise DiamondHipError(load().dmnd_last_error().decode())
    return p


def matrix_params(name, gap_open=-1, gap_extend=-1, params=None):
    """--matrix NAME --gapopen O --gapextend E (-1 = the matrix's defaults) -> Params (ScoreMatrix ctor, score_matrix.cpp:49-72)."""
    p = params if params is not None else default_params()
    rc = load().dmnd_matrix_params(name.encode(), int(gap_open), int(gap_extend), ctypes.byref(p))
    if rc != 0:
        raise DiamondHipError(load().dmnd_last_error().decode())
    return p


def evalue_batch(p, score, qlen, slen):
    """ScoreMatrix::evalue for arrays (host double precision)."""
    score = np.ascontiguousarray(score, dtype=np.int32)
    qlen = np.ascontiguousarray(qlen, dtype=np.int32)
    slen = np.ascontiguousarray(slen, dtype=np.int32)
    out = np.zeros(score.size, np.float64)
    lib = load()
    rc = lib.dmnd_evalue_batch(ctypes.byref(p), score.ctypes.data_as(ctypes.c_void_p), qlen.ctypes.data_as(ctypes.c_void_p),
                               slen.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(score.size), out.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())
    return out


def seed_params_fast(threads=1):
    """--fast seed configuration as the reference sets it up for `threads` threads (search/setup.cpp)."""
    p = SeedParams()
    rc = load().dmnd_seed_params_fast(ctypes.byref(p), int(threads))
    if rc != 0:
        raise DiamondHipError(load().dmnd_last_error().decode())
    return p


def extend_plan(params, qdata, qlimits, tdata, tlimits, hits, threads=1, query_contexts=1):
    """Host-only: (Hauser int8 bias parallel to qdata, round-1 DpTargets) for seed hits sorted by query.
    query_contexts = 6 for a translated (blastx) query block; PLAN records then carry the frame's block sequence id."""
    lib = load()
    qd = np.ascontiguousarray(qdata, dtype=np.int8)
    td = np.ascontiguousarray(tdata, dtype=np.int8)
    ql = np.ascontiguousarray(qlimits, dtype=np.int64)
    tl = np.ascontiguousarray(tlimits, dtype=np.int64)
    hits = np.ascontiguousarray(hits, dtype=SEED_HIT_DTYPE)
    cbs = np.zeros(qd.size, np.int8)
    cap = max(1024, 4 * hits.size)
    out = np.zeros(cap, dtype=PLAN_DTYPE)
    n = ctypes.c_int64(0)
    v = ctypes.c_void_p
    rc = lib.dmnd_extend_plan(ctypes.byref(params), qd.ctypes.data_as(v), ql.ctypes.data_as(v), ctypes.c_int64(ql.size - 1),
                              td.ctypes.data_as(v), tl.ctypes.data_as(v), ctypes.c_int64(tl.size - 1),
                              hits.ctypes.data_as(v), ctypes.c_int64(hits.size), int(threads), int(query_contexts), cbs.ctypes.data_as(v),
                              out.ctypes.data_as(v), ctypes.c_int64(cap), ctypes.byref(n))
    if rc != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())
    return cbs, out[:n.value].copy()


def format_tab(matches, qids, tids, source_lens=None):
    """BLAST tabular text (-f 6 default columns) of match records, as the reference prints it.
    source_lens: DNA read lengths for translated (blastx) matches -> qstart/qend in read coordinates."""
    lib = load()
    buf = ctypes.create_string_buffer(4096)
    out = []
    for m in matches:
        rec = np.ascontiguousarray(m)
        if source_lens is not None:
            n = lib.dmnd_format_tab_translated(rec.ctypes.data_as(ctypes.c_void_p), qids[int(m["query"])].encode(), tids[int(m["target"])].encode(),
                                               int(source_lens[int(m["query"])]), buf, 4096)
        else:
            n = lib.dmnd_format_tab(rec.ctypes.data_as(ctypes.c_void_p), qids[int(m["query"])].encode(), tids[int(m["target"])].encode(), buf, 4096)
        if n < 0:
            raise DiamondHipError(lib.dmnd_last_error().decode())
        out.append(buf.raw[:n].decode())
    return "".join(out)


class HspView(ctypes.Structure):
    """dmnd_hsp_view: one HSP with everything the output formats read."""
    _fields_ = [("match", ctypes.c_void_p), ("transcript", ctypes.c_void_p), ("qtitle", ctypes.c_char_p), ("stitle", ctypes.c_char_p),
                ("qseq", ctypes.c_void_p), ("qlen", ctypes.c_int32), ("slen", ctypes.c_int32), ("full_sseq", ctypes.c_void_p),
                ("source_seq", ctypes.c_void_p), ("source_len", ctypes.c_int32), ("qnum", ctypes.c_int64), ("snum", ctypes.c_int64),
                ("qframes", ctypes.c_void_p * 3)]


def output_fields(names):
    """Field names of --outfmt 6 -> (ids, needs_transcript); raises with the reference's message for an unknown field."""
    lib = load()
    arr = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
    ids = (ctypes.c_int32 * len(names))()
    need = ctypes.c_int(0)
    if lib.dmnd_output_fields(arr, len(names), ids, ctypes.byref(need)) != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())
    return list(ids), bool(need.value)


class _View:
    """Keeps the numpy buffers of an HspView alive."""

    def __init__(self, match, transcript, qtitle, stitle, qseq, slen, full_sseq=None, source_seq=None, qnum=0, snum=0):
        self.rec = np.ascontiguousarray(match, dtype=MATCH_DTYPE).reshape(1).copy()
        self.tr = None if transcript is None else np.ascontiguousarray(transcript, dtype=np.uint8)
        self.q = np.ascontiguousarray(qseq, dtype=np.int8)
        self.fs = None if full_sseq is None else np.ascontiguousarray(full_sseq, dtype=np.int8)
        self.src = None if source_seq is None else np.ascontiguousarray(source_seq, dtype=np.int8)
        v = HspView()
        v.match = self.rec.ctypes.data
        v.transcript = None if self.tr is None else self.tr.ctypes.data
        v.qtitle, v.stitle = qtitle.encode(), stitle.encode()
        v.qseq, v.qlen, v.slen = self.q.ctypes.data, len(self.q), int(slen)
        v.full_sseq = None if self.fs is None else self.fs.ctypes.data
        v.source_seq = None if self.src is None else self.src.ctypes.data
        v.source_len = 0 if self.src is None else len(self.src)
        v.qnum, v.snum = int(qnum), int(snum)
        self.v = v


def _formatted(fn, *args):
    lib = load()
    cap = 1 << 16
    while True:
        buf = ctypes.create_string_buffer(cap)
        fn.restype = ctypes.c_int64
        n = fn(*args, buf, ctypes.c_int64(cap))
        if n == -5 and cap < (1 << 30):       # DMND_E_CAP
            cap *= 8
            continue
        if n < 0:
            raise DiamondHipError(lib.dmnd_last_error().decode())
        return buf.raw[:n].decode()


def format_fields(ids, match, transcript, qtitle, stitle, qseq, slen, **kw):
    """One `-f 6 FIELD...` line of a match record (transcript: its PackedOperation bytes; qseq: the aligned query context)."""
    view = _View(match, transcript, qtitle, stitle, qseq, slen, **kw)
    arr = (ctypes.c_int32 * len(ids))(*ids)
    return _formatted(load().dmnd_format_fields, ctypes.byref(view.v), arr, len(ids))


def format_pairwise(match, transcript, qtitle, stitle, qseq, slen, matrix8, **kw):
    """The `-f 0` block of one match."""
    view = _View(match, transcript, qtitle, stitle, qseq, slen, **kw)
    m = np.ascontiguousarray(matrix8, dtype=np.int8)
    return _formatted(load().dmnd_format_pairwise, ctypes.byref(view.v), m.ctypes.data_as(ctypes.c_void_p))


def format_pairwise_intro(qtitle, qlen, unaligned=False):
    return _formatted(load().dmnd_format_pairwise_intro, qtitle.encode(), ctypes.c_int32(int(qlen)), ctypes.c_int(1 if unaligned else 0))


def format_paf(match, qtitle, stitle, qseq, slen, **kw):
    view = _View(match, None, qtitle, stitle, qseq, slen, **kw)
    return _formatted(load().dmnd_format_paf, ctypes.byref(view.v), None)


def seed_params_default(scoring, threads=1):
    """Default-sensitivity seed configuration (2 shapes of weight 10, ungapped e-value filter 10000)."""
    p = SeedParams()
    rc = load().dmnd_seed_params_default(ctypes.byref(p), int(threads), ctypes.byref(scoring))
    if rc != 0:
        raise DiamondHipError(load().dmnd_last_error().decode())
    return p


SENS = {"fast": 0, "default": 1, "mid-sensitive": 2, "sensitive": 3, "more-sensitive": 4, "very-sensitive": 5, "ultra-sensitive": 6}


def seed_params_preset(name, scoring, threads=1):
    """(SeedParams, gapped_filter_evalue) of a sensitivity preset: fast, default, mid-sensitive, sensitive, more-sensitive."""
    p = SeedParams()
    gf = ctypes.c_double(0)
    lib = load()
    lib.dmnd_seed_params_preset.argtypes = [ctypes.POINTER(SeedParams), ctypes.c_int, ctypes.c_int, ctypes.POINTER(Params), ctypes.POINTER(ctypes.c_double)]
    rc = lib.dmnd_seed_params_preset(ctypes.byref(p), SENS[name], int(threads), ctypes.byref(scoring), ctypes.byref(gf))
    if rc != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())
    return p, gf.value


def set_index_chunks(p, index_chunks, threads=1):
    """-c / --index-chunks on a SeedParams preset (recomputes seedp_bits as the reference does)."""
    lib = load()
    lib.dmnd_seed_params_set_index_chunks.argtypes = [ctypes.POINTER(SeedParams), ctypes.c_int, ctypes.c_int]
    if lib.dmnd_seed_params_set_index_chunks(ctypes.byref(p), int(index_chunks), int(threads)) != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())
    return p


def set_query_indexed(p, threads=1):
    """The reference's query-indexed algorithm (--algo 1): hashed seed encoding, one index chunk (dmnd_seed_params_set_query_indexed)."""
    lib = load()
    lib.dmnd_seed_params_set_query_indexed.argtypes = [ctypes.POINTER(SeedParams), ctypes.c_int]
    if lib.dmnd_seed_params_set_query_indexed(ctypes.byref(p), int(threads)) != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())
    return p


def auto_query_indexed(p, qdata, qlimits, db_bytes):
    """The reference's --algo auto choice (dmnd_auto_query_indexed): True = query-indexed."""
    lib = load()
    qd = np.ascontiguousarray(qdata, dtype=np.int8)
    ql = np.ascontiguousarray(qlimits, dtype=np.int64)
    out = ctypes.c_int(0)
    lib.dmnd_auto_query_indexed.argtypes = [ctypes.POINTER(SeedParams), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_int)]
    if lib.dmnd_auto_query_indexed(ctypes.byref(p), qd.ctypes.data, ql.ctypes.data, len(ql) - 1, int(db_bytes), ctypes.byref(out)) != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())
    return bool(out.value)


def join_blocks(records, max_target_seqs=25, copy=True):
    """Merge of one query block's records against several reference blocks (dmnd_join_blocks: join_query of the reference):
    `records` = concatenation of the per-block MATCH_DTYPE arrays with database-wide target ordinals. copy=False: the caller's
    array is reordered in place (it must be a contiguous MATCH_DTYPE array the caller owns)."""
    lib = load()
    r = np.ascontiguousarray(records, dtype=MATCH_DTYPE)
    if copy or r is not records or not r.flags["WRITEABLE"]:
        r = r.copy()
    n = ctypes.c_int64(0)
    lib.dmnd_join_blocks.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
    if lib.dmnd_join_blocks(r.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(r.size), int(max_target_seqs), ctypes.byref(n)) != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())
    return r[:n.value]


RANKED_DTYPE = np.dtype([("query", "<u4"), ("target", "<u4"), ("score", "<u2"), ("context", "u1"), ("pad", "u1")])
assert RANKED_DTYPE.itemsize == 12


def rank_update(table, n, records):
    """--global-ranking: merges the records of one block pair (RANKED_DTYPE, grouped by query, targets as database ordinals) into the
    table of the n best targets per query (dmnd_rank_update = merge_hits). table: RANKED_DTYPE[n_queries * n], modified in place."""
    lib = load()
    assert table.dtype == RANKED_DTYPE and table.flags["C_CONTIGUOUS"] and table.size % n == 0
    r = np.ascontiguousarray(records, dtype=RANKED_DTYPE)
    lib.dmnd_rank_update.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
    if lib.dmnd_rank_update(table.ctypes.data, ctypes.c_int64(table.size // n), int(n), r.ctypes.data, ctypes.c_int64(r.size)) != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())
    return table


def join_blocks_top(records, top_percent):
    """The block join of a --top run (dmnd_join_blocks_top): score order, targets within top_percent of a query's best bit score."""
    lib = load()
    r = np.ascontiguousarray(records, dtype=MATCH_DTYPE).copy()
    n = ctypes.c_int64(0)
    lib.dmnd_join_blocks_top.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.POINTER(ctypes.c_int64)]
    if lib.dmnd_join_blocks_top(r.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(r.size), float(top_percent), ctypes.byref(n)) != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())
    return r[:n.value]


def join_ranks(contexts, records, n_queries, max_target_seqs=25, top_percent=-1.0):
    """dmnd_join_ranks: the RCCL merge of the records of several GPUs driven by one process. contexts: one hip.Context per GPU
    (or several on one device: copies instead of RCCL); records: one MATCH_DTYPE array per context. Returns (joined, transport)."""
    lib = load()
    recs = [np.ascontiguousarray(r, dtype=MATCH_DTYPE) for r in records]
    n = len(contexts)
    out = np.empty(max(1, sum(len(r) for r in recs)), dtype=MATCH_DTYPE)
    n_out, transport = ctypes.c_int64(0), ctypes.c_int(0)
    lib.dmnd_join_ranks.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                    ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]
    hs = (ctypes.c_void_p * n)(*[c.h for c in contexts])
    ps = (ctypes.c_void_p * n)(*[r.ctypes.data for r in recs])
    cs = (ctypes.c_int64 * n)(*[len(r) for r in recs])
    if lib.dmnd_join_ranks(hs, n, ps, cs, ctypes.c_int64(int(n_queries)), int(max_target_seqs), float(top_percent), out.ctypes.data_as(ctypes.c_void_p),
                           ctypes.c_int64(out.size), ctypes.byref(n_out), ctypes.byref(transport)) != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())
    return out[:n_out.value], transport.value


def join_blocks_range(records, max_target_seqs=25, top_percent=-1.0, range_cover=50.0):
    """The block join of a --range-culling run (dmnd_join_blocks_range = join_query with RangeCulling): a target is skipped when
    range_cover per cent of its HSPs' read intervals are covered by the alignments kept before it."""
    lib = load()
    r = np.ascontiguousarray(records, dtype=MATCH_DTYPE).copy()
    n = ctypes.c_int64(0)
    lib.dmnd_join_blocks_range.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.POINTER(ctypes.c_int64)]
    if lib.dmnd_join_blocks_range(r.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(r.size), int(max_target_seqs), float(top_percent), float(range_cover), ctypes.byref(n)) != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())
    return r[:n.value]


def format_sam(match, transcript, qtitle, stitle, qseq, slen, **kw):
    view = _View(match, transcript, qtitle, stitle, qseq, slen, **kw)
    return _formatted(load().dmnd_format_sam, ctypes.byref(view.v), None)


def format_xml_header(program, version, database, first_qtitle, first_qlen, matrix="blosum62", gap_open=11, gap_extend=1, max_evalue=0.001):
    """`-f 5`: the file header (XMLFormat::print_header)."""
    return _formatted(load().dmnd_format_xml_header, program.encode(), version.encode(), database.encode(), first_qtitle.encode(), ctypes.c_int32(int(first_qlen)),
                      matrix.encode(), ctypes.c_int(int(gap_open)), ctypes.c_int(int(gap_extend)), ctypes.c_double(float(max_evalue)))


def format_xml_query_intro(qtitle, qnum, qlen):
    return _formatted(load().dmnd_format_xml_query_intro, qtitle.encode(), ctypes.c_int64(int(qnum)), ctypes.c_int32(int(qlen)))


def format_xml(match, transcript, qtitle, stitle, qseq, slen, hit_num, hsp_num, matrix8, **kw):
    """One <Hsp> of the XML format (hsp_num 0 opens the <Hit>)."""
    view = _View(match, transcript, qtitle, stitle, qseq, slen, **kw)
    m = np.ascontiguousarray(matrix8, dtype=np.int8)
    return _formatted(load().dmnd_format_xml, ctypes.byref(view.v), ctypes.c_int32(int(hit_num)), ctypes.c_int32(int(hsp_num)), m.ctypes.data_as(ctypes.c_void_p))


def format_xml_query_epilog(unaligned, db_seqs, db_letters, K, lambda_):
    return _formatted(load().dmnd_format_xml_query_epilog, ctypes.c_int(1 if unaligned else 0), ctypes.c_int64(int(db_seqs)), ctypes.c_int64(int(db_letters)),
                      ctypes.c_double(float(K)), ctypes.c_double(float(lambda_)))


XML_FOOTER = "</BlastOutput_iterations>\n</BlastOutput>"


class DaaHeader(ctypes.Structure):
    """dmnd_daa_header (include/diamond_hip.h)."""
    _fields_ = [("build", ctypes.c_int64), ("db_seqs", ctypes.c_int64), ("db_letters", ctypes.c_int64), ("db_seqs_used", ctypes.c_int64),
                ("query_records", ctypes.c_int64), ("mode", ctypes.c_int32), ("gap_open", ctypes.c_int32), ("gap_extend", ctypes.c_int32),
                ("K", ctypes.c_double), ("lambda_", ctypes.c_double), ("max_evalue", ctypes.c_double), ("matrix", ctypes.c_char_p),
                ("finished", ctypes.c_int32), ("alignment_bytes", ctypes.c_int64), ("ref_name_bytes", ctypes.c_int64)]


def _binary(fn, *args):
    lib = load()
    cap = 1 << 16
    while True:
        buf = ctypes.create_string_buffer(cap)
        fn.restype = ctypes.c_int64
        n = fn(*args, buf, ctypes.c_int64(cap))
        if n == -5 and cap < (1 << 30):       # DMND_E_CAP
            cap *= 8
            continue
        if n < 0:
            raise DiamondHipError(lib.dmnd_last_error().decode())
        return buf.raw[:n]


def format_daa_header(**kw):
    h = DaaHeader()
    for k, v in kw.items():
        setattr(h, k, v.encode() if k == "matrix" else v)
    return _binary(load().dmnd_format_daa_header, ctypes.byref(h))


def format_daa_query(qtitle, seq, dna=False):
    s = np.ascontiguousarray(seq, dtype=np.int8)
    return _binary(load().dmnd_format_daa_query, qtitle.encode(), s.ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(s.size), ctypes.c_int(1 if dna else 0))


def format_daa_match(match, transcript, qtitle, stitle, qseq, slen, dict_id, **kw):
    view = _View(match, transcript, qtitle, stitle, qseq, slen, **kw)
    return _binary(load().dmnd_format_daa_match, ctypes.byref(view.v), ctypes.c_uint32(int(dict_id)))


def seg_ranges(seq):
    """SEG segments of one sequence (int8 letters) as [(begin, end)] inclusive (dmnd_seg_ranges)."""
    lib = load()
    s = np.ascontiguousarray(seq, dtype=np.int8)
    cap = max(16, s.size // 4)
    out = np.zeros(2 * cap, np.int32)
    n = ctypes.c_int32(0)
    lib.dmnd_seg_ranges.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]
    if lib.dmnd_seg_ranges(s.ctypes.data, s.size, out.ctypes.data, cap, ctypes.byref(n)) != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n.value)]


def seg_mask_block(data, limits, threads=1):
    """Hard-masks a SequenceSet block in place with SEG (dmnd_seg_mask_block); returns the number of masked letters."""
    lib = load()
    assert data.dtype == np.int8 and data.flags["C_CONTIGUOUS"]
    lim = np.ascontiguousarray(limits, dtype=np.int64)
    n = ctypes.c_int64(0)
    lib.dmnd_seg_mask_block.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
    if lib.dmnd_seg_mask_block(data.ctypes.data, lim.ctypes.data, len(lim) - 1, int(threads), ctypes.byref(n)) != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())
    return n.value


FMT_XML_BLORD, FMT_NO_PARSE_SEQIDS, FMT_SAM_QUERY_LEN = 1, 2, 4


def set_format_flags(flags):
    """--xml-blord-format / --no-parse-seqids / --sam-query-len (process-wide, dmnd_set_format_flags)."""
    lib = load()
    lib.dmnd_set_format_flags.argtypes = [ctypes.c_uint32]
    if lib.dmnd_set_format_flags(int(flags)) != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())


def seed_params_sensitive(scoring, threads=1):
    """--sensitive seed configuration (16 shapes of weight 8, ungapped e-value filter 10000); pair with set_gapped_filter(1.0)."""
    p = SeedParams()
    rc = load().dmnd_seed_params_sensitive(ctypes.byref(p), int(threads), ctypes.byref(scoring))
    if rc != 0:
        raise DiamondHipError(load().dmnd_last_error().decode())
    return p


def translate(dna, gencode=1, strands=3, min_orf=0):
    """Six-frame translation of one DNA read (int8 letters 0-4 = ACGTN) as the reference loads a blastx query
    (--query-gencode, --strand as a mask 1 plus / 2 minus / 3 both, --min-orf). Returns a list of six int8 arrays (frames 0-2
    forward, 3-5 reverse)."""
    lib = load()
    dna = np.ascontiguousarray(dna, dtype=np.int8)
    n = dna.size // 3
    bufs = [np.zeros(max(n, 1), np.int8) for _ in range(6)]
    ptrs = (ctypes.c_void_p * 6)(*[b.ctypes.data for b in bufs])
    lens = (ctypes.c_int32 * 6)()
    rc = lib.dmnd_translate_opts(dna.ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(dna.size), int(gencode), int(strands), int(min_orf), ptrs, lens)
    if rc != 0:
        raise DiamondHipError(lib.dmnd_last_error().decode())
    return [bufs[f][:lens[f]].copy() for f in range(6)]


def translated_block(dna, off):
    """SequenceSet block (data, limits) of the six frames of every read, in the reference's order (read-major)."""
    from . import workload
    frames = []
    for i in range(len(off) - 1):
        frames.extend(translate(dna[off[i]:off[i + 1]]))
    lens = np.array([len(f) for f in frames], np.int64)
    o = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    data = np.concatenate(frames) if frames else np.zeros(0, np.int8)
    return workload.sequence_set(data, o)


def matrix_of(p):
    return np.frombuffer(bytes(p.matrix8), dtype=np.int8).reshape(32, 32).copy()


class Context:
    """One dmnd_ctx = one MI355X."""

    def __init__(self, device=-1, params=None):
        self.lib = load()
        self.params = params if params is not None else default_params()
        self.h = self.lib.dmnd_create(device, ctypes.byref(self.params))
        if not self.h:
            raise DiamondHipError(self.lib.dmnd_last_error().decode())

    def close(self):
        if self.h:
            self.lib.dmnd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise DiamondHipError("error %d: %s" % (rc, self.lib.dmnd_last_error().decode()))

    def set_db_letters(self, n):
        self._check(self.lib.dmnd_set_db_letters(self.h, float(n)))

    def upload_block(self, which, data, limits=None):
        data = np.ascontiguousarray(data, dtype=np.int8)
        lim = np.ascontiguousarray(limits, dtype=np.int64) if limits is not None else None
        self._check(self.lib.dmnd_upload_block(self.h, which, data.ctypes.data, data.size,
                                               lim.ctypes.data if lim is not None else None,
                                               (lim.size - 1) if lim is not None else 0))

    def xdrop_ungapped(self, hits, use_bias=False, xdrop=0):
        """dmnd_xdrop_ungapped: the x-drop ungapped extension of every seed hit (SEED_HIT_DTYPE) of the resident block pair.
        Returns an array with fields i (query sequence position), j (reference block offset), len, score."""
        h = np.ascontiguousarray(hits, dtype=SEED_HIT_DTYPE)
        out = np.zeros(len(h), dtype=np.dtype([("i", "<i4"), ("j", "<i8"), ("len", "<i4"), ("score", "<i4")], align=True))
        assert out.dtype.itemsize == 24
        self.lib.dmnd_xdrop_ungapped.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        self._check(self.lib.dmnd_xdrop_ungapped(self.h, h.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(len(h)), int(bool(use_bias)), int(xdrop), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def join_blocks_device(self, records, max_target_seqs=25, top_percent=-1.0):
        """The block join on the device for records in host memory (dmnd_join_blocks_device_host: upload, three radix sorts of a
        permutation, top-k per query, download of the survivors). `records`: concatenated per-block MATCH_DTYPE arrays with
        database-wide target ordinals, one record per (query, target). Returns the joined records (a new array)."""
        r = np.ascontiguousarray(records, dtype=MATCH_DTYPE).copy()
        n = ctypes.c_int64(0)
        self.lib.dmnd_join_blocks_device_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_int64)]
        self._check(self.lib.dmnd_join_blocks_device_host(self.h, r.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(r.size), int(max_target_seqs), float(top_percent), ctypes.byref(n)))
        return r[:n.value]

    def extend_records_device(self):
        """(device pointer, n) of the last extend()'s records where they lie in HBM, or (0, -1) when part of them exists on the host only"""
        ptr, n = ctypes.c_void_p(0), ctypes.c_int64(0)
        self.lib.dmnd_extend_records_device.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]
        self._check(self.lib.dmnd_extend_records_device(self.h, ctypes.byref(ptr), ctypes.byref(n)))
        return (ptr.value or 0), n.value

    def join_contexts_device(self, contexts, target_offsets, max_target_seqs=25, top_percent=-1.0, max_query=0):
        """dmnd_join_contexts_device: the block join over the device-resident records of the contexts' last extend() (one context per
        reference block, on this context's device); returns the joined records."""
        n_ctx = len(contexts)
        hs = (ctypes.c_void_p * n_ctx)(*[c.h for c in contexts])
        offs = (ctypes.c_uint32 * n_ctx)(*[int(x) for x in target_offsets])
        total = sum(max(c.extend_records_device()[1], 0) for c in contexts)
        out = np.zeros(max(total, 1), dtype=MATCH_DTYPE)
        n = ctypes.c_int64(0)
        self.lib.dmnd_join_contexts_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_uint32, ctypes.c_void_p,
                                                       ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
        self._check(self.lib.dmnd_join_contexts_device(self.h, hs, offs, n_ctx, int(max_target_seqs), float(top_percent), ctypes.c_uint32(int(max_query)),
                                                       out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(out.size), ctypes.byref(n)))
        return out[:n.value]

    def join_blocks_device_ptr(self, records_ptr, n, out_ptr, max_target_seqs=25, top_percent=-1.0, max_query=0):
        """dmnd_join_blocks_device on device pointers (e.g. torch tensors' data_ptr() on this context's device): n records at
        records_ptr are joined into out_ptr (room for n records); returns the number of survivors."""
        n_out = ctypes.c_int64(0)
        self.lib.dmnd_join_blocks_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
        self._check(self.lib.dmnd_join_blocks_device(self.h, ctypes.c_void_p(int(records_ptr)), ctypes.c_int64(int(n)), int(max_target_seqs), float(top_percent), ctypes.c_uint32(int(max_query)),
                                                     ctypes.c_void_p(int(out_ptr)), ctypes.byref(n_out)))
        return n_out.value

    def share_block(self, which, src):
        """Alias block `which` of the context `src` (no copy; `src` must stay alive): dmnd_share_block."""
        self.lib.dmnd_share_block.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        self._check(self.lib.dmnd_share_block(self.h, int(which), src.h))

    def copy_block(self, which, src):
        """Overwrite the letters of block `which` with those of `src`'s block of the same shape (device to device): dmnd_copy_block."""
        self.lib.dmnd_copy_block.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        self._check(self.lib.dmnd_copy_block(self.h, int(which), src.h))

    def frameshift_swipe(self, items, score_only, frame_shift=15, channels=16, with_transcripts=True):
        """banded_3frame_swipe for many (query strand, target) items (FS_TARGET_DTYPE) -> (FS_HSP_DTYPE[], transcript bytes or None)."""
        items = np.ascontiguousarray(items, dtype=FS_TARGET_DTYPE)
        out = np.zeros(len(items), dtype=FS_HSP_DTYPE)
        tr, used = None, ctypes.c_int64(0)
        if not score_only and with_transcripts:
            tr = np.zeros(int((2 * items["target_len"].astype(np.int64) + items["frame_len"][:, 0] + 65).sum()) + 16, np.uint8)
        self._check(self.lib.dmnd_frameshift_swipe(self.h, items.ctypes.data, len(items), int(bool(score_only)), int(frame_shift), int(channels),
                                                   out.ctypes.data, tr.ctypes.data if tr is not None else None, tr.size if tr is not None else 0,
                                                   ctypes.byref(used)))
        return out, (tr[:used.value] if tr is not None else None)

    def upload_matrices(self, matrices):
        """n composition-adjusted matrices (int8[n, 32, 32]) for the items whose cbs_off is -2 - number."""
        m = np.ascontiguousarray(matrices, dtype=np.int8).reshape(-1, 32, 32)
        self._check(self.lib.dmnd_upload_matrices(self.h, m.ctypes.data if len(m) else None, len(m)))

    def upload_cbs(self, cbs):
        cbs = np.ascontiguousarray(cbs, dtype=np.int8)
        self._check(self.lib.dmnd_upload_cbs(self.h, cbs.ctypes.data if cbs.size else None, cbs.size))

    def banded_swipe(self, items, mode, hsp_values=0, transcript_cap=None):
        """items: structured array of DP_TARGET_DTYPE. Returns (hsps[HSP_DTYPE], transcript bytes or None)."""
        items = np.ascontiguousarray(items, dtype=DP_TARGET_DTYPE)
        out = np.zeros(items.size, dtype=HSP_DTYPE)
        tr = None
        used = ctypes.c_int64(0)
        if mode == SWIPE_TRACEBACK:
            if transcript_cap is None:
                transcript_cap = int((items["query_len"].astype(np.int64) + items["target_len"] + 2).sum()) + 16
            tr = np.zeros(transcript_cap, np.uint8)
        self._check(self.lib.dmnd_banded_swipe(self.h, items.ctypes.data, items.size, mode, hsp_values, out.ctypes.data,
                                               tr.ctypes.data if tr is not None else None,
                                               tr.size if tr is not None else 0, ctypes.byref(used)))
        return out, (tr[:used.value] if tr is not None else None)

    def banded_swipe_host(self, query, cbs, targets, mode, hsp_values=0):
        """targets: list of (seq int8[], d_begin, d_end[, matrix int8[26, 32] or None]): the literal reference call shape (one query)."""
        q = np.ascontiguousarray(query, dtype=np.int8)
        c = np.ascontiguousarray(cbs, dtype=np.int8) if cbs is not None else None
        seqs = [np.ascontiguousarray(t[0], dtype=np.int8) for t in targets]
        mats = [np.ascontiguousarray(t[3], dtype=np.int8) if len(t) > 3 and t[3] is not None else None for t in targets]
        ht = np.zeros(len(targets), dtype=HOST_TARGET_DTYPE)
        for k, (s, t) in enumerate(zip(seqs, targets)):
            assert mats[k] is None or mats[k].size >= 26 * 32
            ht[k] = (s.ctypes.data, s.size, t[1], t[2], mats[k].ctypes.data if mats[k] is not None else 0)
        out = np.zeros(len(targets), dtype=HSP_DTYPE)
        tr = None
        used = ctypes.c_int64(0)
        if mode == SWIPE_TRACEBACK:
            tr = np.zeros(sum(s.size + q.size + 2 for s in seqs) + 16, np.uint8)
        self._check(self.lib.dmnd_banded_swipe_host(self.h, q.ctypes.data, q.size, c.ctypes.data if c is not None else None,
                                                    ht.ctypes.data, len(targets), mode, hsp_values, out.ctypes.data,
                                                    tr.ctypes.data if tr is not None else None,
                                                    tr.size if tr is not None else 0, ctypes.byref(used)))
        return out, (tr[:used.value] if tr is not None else None)

    def seed_search(self, seed_params):
        """Search::search_shape for all shapes/chunks on the uploaded blocks. Returns hits sorted by (query, subject, seed_offset)."""
        n = ctypes.c_int64(0)
        self._check(self.lib.dmnd_seed_search(self.h, ctypes.byref(seed_params), ctypes.byref(n)))
        hits = np.empty(n.value, dtype=SEED_HIT_DTYPE)
        self._check(self.lib.dmnd_seed_hits(self.h, hits.ctypes.data if n.value else None, n.value))
        return hits

    def mask_block(self, which, host_data=None):
        """tantan repeat masking of the uploaded block in HBM (hard mask, letter 23). host_data: the int8 array the block was
        uploaded from (C-contiguous); it receives the masked letters. Returns the number of masked positions."""
        n = ctypes.c_int64(0)
        ptr = None
        if host_data is not None:
            assert host_data.dtype == np.int8 and host_data.flags["C_CONTIGUOUS"]
            ptr = host_data.ctypes.data
        self._check(self.lib.dmnd_mask_block(self.h, int(which), ptr, ctypes.byref(n)))
        return n.value

    def mask_sequences(self, which, host_data, seq_ids):
        """tantan on the given sequences of the block only (lazy masking); host_data as for mask_block. Returns the masked positions."""
        ids = np.ascontiguousarray(seq_ids, dtype=np.int32)
        n = ctypes.c_int64(0)
        ptr = None
        if host_data is not None:
            assert host_data.dtype == np.int8 and host_data.flags["C_CONTIGUOUS"]
            ptr = host_data.ctypes.data
        self.lib.dmnd_mask_sequences.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
        self._check(self.lib.dmnd_mask_sequences(self.h, int(which), ptr, ids.ctypes.data, ctypes.c_int64(ids.size), ctypes.byref(n)))
        return n.value

    def mask_kernel_ms(self):
        return float(self.lib.dmnd_mask_kernel_ms(self.h))

    def soft_mask_block(self, which):
        """Motif soft masking of the uploaded block (after mask_block): builds the view that seeds are generated from; the
        block itself is unchanged. Needs set_motif_table() once per process. Returns the number of letters inside motif ranges."""
        n = ctypes.c_int64(0)
        self.lib.dmnd_soft_mask_block.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        self._check(self.lib.dmnd_soft_mask_block(self.h, int(which), ctypes.byref(n)))
        return n.value

    def set_motif_table(self, codes):
        """This context's own motif table (None: back to the process-wide one; an empty list: no soft masking here)."""
        self.lib.dmnd_set_context_motif_table.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        if codes is None:
            self._check(self.lib.dmnd_set_context_motif_table(self.h, None, -1))
            return
        a = np.ascontiguousarray(codes, dtype=np.uint64)
        self._check(self.lib.dmnd_set_context_motif_table(self.h, a.ctypes.data if len(a) else None, len(a)))

    def set_comp_based_stats(self, mode):
        """1 = Hauser composition bias (default), 0 = none."""
        self.lib.dmnd_set_comp_based_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self._check(self.lib.dmnd_set_comp_based_stats(self.h, int(mode)))

    def set_query_index_reuse(self, on=True):
        """Keep the query seed index between seed_search calls on the same query block (one query block, many reference blocks)."""
        self.lib.dmnd_set_query_index_reuse.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self._check(self.lib.dmnd_set_query_index_reuse(self.h, 1 if on else 0))

    def set_top_percent(self, percent):
        """--top PERCENT (None or a negative value: off): targets within PERCENT of the best bit score instead of the first -k."""
        self.lib.dmnd_set_top_percent.argtypes = [ctypes.c_void_p, ctypes.c_double]
        self._check(self.lib.dmnd_set_top_percent(self.h, -1.0 if percent is None else float(percent)))

    def set_query_contexts(self, contexts):
        """1 = blastp, 6 = blastx (the query block holds the six frames of every read consecutively)."""
        self._check(self.lib.dmnd_set_query_contexts(self.h, int(contexts)))

    def set_gapped_filter(self, evalue):
        """Search::Config::gapped_filter_evalue (1.0 for --sensitive; 0 switches the filter off)."""
        self._check(self.lib.dmnd_set_gapped_filter(self.h, float(evalue)))

    def gapped_filter(self, hits, use_cbs=True, with_scores=False):
        """Per seed hit: 1 if it passes both gapped-filter stages. Returns flags (uint8) [, scores (n x 2 int32: f1, f2)]."""
        hits = np.ascontiguousarray(hits, dtype=SEED_HIT_DTYPE)
        flags = np.zeros(hits.size, np.uint8)
        scores = np.zeros((hits.size, 2), np.int32) if with_scores else None
        self._check(self.lib.dmnd_gapped_filter(self.h, hits.ctypes.data if hits.size else None, hits.size, 1 if use_cbs else 0,
                                                flags.ctypes.data if hits.size else None,
                                                scores.ctypes.data if with_scores and hits.size else None))
        return (flags, scores) if with_scores else flags

    def gapped_filter_ms(self):
        return float(self.lib.dmnd_gapped_filter_ms(self.h))

    def seed_kernel_ms(self):
        ms = (ctypes.c_double * 5)()
        self._check(self.lib.dmnd_seed_kernel_ms(self.h, ms))
        return list(ms)

    def extend(self, qdata, tdata, hits, threads=8, hsp_values=510, with_transcripts=False):
        """Extension::extend for every query of the block (hits sorted by query). Returns (matches, transcripts|None)."""
        qd = np.ascontiguousarray(qdata, dtype=np.int8)
        td = np.ascontiguousarray(tdata, dtype=np.int8)
        hits = np.ascontiguousarray(hits, dtype=SEED_HIT_DTYPE)
        cap = max(1024, hits.size)
        v = ctypes.c_void_p
        tr = np.zeros(max(1 << 20, 64 * hits.size) if with_transcripts else 0, np.uint8)
        while True:
            out = np.empty(cap, dtype=MATCH_DTYPE)
            n, used = ctypes.c_int64(0), ctypes.c_int64(0)
            rc = self.lib.dmnd_extend(self.h, qd.ctypes.data_as(v), td.ctypes.data_as(v), hits.ctypes.data_as(v),
                                      ctypes.c_int64(hits.size), int(threads), ctypes.c_uint32(hsp_values),
                                      out.ctypes.data_as(v), ctypes.c_int64(cap), ctypes.byref(n),
                                      tr.ctypes.data_as(v) if with_transcripts else None, ctypes.c_int64(tr.size),
                                      ctypes.byref(used))
            if rc != 0 and n.value > cap:          # several HSPs per target (set_max_hsps): more records than seed hits
                cap = n.value
                continue
            self._check(rc)
            return out[:n.value], (tr[:used.value] if with_transcripts else None)

    def rank_targets(self, qdata, tdata, hits, threads=4):
        """--global-ranking: per (query, target) of the seed hits the best x-drop ungapped score and its context (dmnd_rank_targets)."""
        qd = np.ascontiguousarray(qdata, dtype=np.int8)
        td = np.ascontiguousarray(tdata, dtype=np.int8)
        hits = np.ascontiguousarray(hits, dtype=SEED_HIT_DTYPE)
        out = np.zeros(max(1, hits.size), RANKED_DTYPE)
        n = ctypes.c_int64(0)
        v = ctypes.c_void_p
        self.lib.dmnd_rank_targets.argtypes = [v, v, v, v, ctypes.c_int64, ctypes.c_int, v, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
        self._check(self.lib.dmnd_rank_targets(self.h, qd.ctypes.data, td.ctypes.data, hits.ctypes.data, ctypes.c_int64(hits.size), int(threads),
                                               out.ctypes.data, ctypes.c_int64(out.size), ctypes.byref(n)))
        return out[:n.value]

    def set_global_ranking(self, n):
        self._check(self.lib.dmnd_set_global_ranking(self.h, int(n)))

    def set_max_hsps(self, n):
        """--max-hsps: HSPs reported per target (default 1; 0 = all). The records of a target then follow each other."""
        self._check(self.lib.dmnd_set_max_hsps(self.h, int(n)))

    def touch_streams(self):
        """Re-acquires the hardware queues of the context's streams after a device-wide synchronize (see diamond_hip.h)."""
        self._check(self.lib.dmnd_touch_streams(self.h))

    def extend_stats(self):
        st = (ctypes.c_double * 12)()
        self.lib.dmnd_extend_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        self._check(self.lib.dmnd_extend_stats(self.h, st))
        keys = ["round1_targets", "round2_targets", "round1_cells", "round2_cells", "host_hauser_ms", "host_chaining_ms",
                "round1_call_ms", "host_culling_ms", "round2_call_ms", "round1_swipe_kernel_ms", "round2_swipe_kernel_ms",
                "traceback_kernel_ms"]
        return dict(zip(keys, list(st)))

    def extend_plan_stats(self):
        """(groups found on the device, groups left to the host, bands planned) of the last extend(); zeros: the host planned"""
        st = (ctypes.c_double * 3)()
        self.lib.dmnd_extend_plan_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        self._check(self.lib.dmnd_extend_plan_stats(self.h, st))
        return dict(groups=int(st[0]), groups_on_host=int(st[1]), bands=int(st[2]))

    def extend_device_stats(self):
        """(queries extended on the device, of them handed back to the host, round-1 DpTargets, records) of the last extend()"""
        st = (ctypes.c_double * 10)()
        self.lib.dmnd_extend_device_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        self._check(self.lib.dmnd_extend_device_stats(self.h, st))
        return dict(queries=int(st[0]), queries_back_to_host=int(st[1]), items=int(st[2]), records=int(st[3]), band_diagonal_steps=st[4], wavefront_diagonal_steps=st[5],
                    round2_cells=st[6], round2_cells_swept_again=st[7], round2_sweep_kernel_ms=st[8])

    def last_kernel_ms(self):
        a, b = ctypes.c_double(0), ctypes.c_double(0)
        self._check(self.lib.dmnd_last_kernel_ms(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def evalue(self, score, qlen, slen):
        return self.lib.dmnd_evalue(self.h, int(score), int(qlen), int(slen))

    def bitscore(self, score):
        return self.lib.dmnd_bitscore(self.h, float(score))
