"""TEST INFRASTRUCTURE ONLY -- reader for the known-answer tap files written by
oracle/_ref/diamond_tap (oracle/ref_tap.cpp documents the record layout).  Each record is one
genuine call of the reference's DP::BandedSwipe::swipe (dp/dp.h:287)."""
import struct
import numpy as np

MAGIC = 0x31505753
MAGIC_MTX = 0x3158544D
MAGIC2 = 0x32505753          # 'SWP2': DpTargets carry their adjusted matrix

HSP_FIELDS = ("swipe_target", "swipe_bin", "score", "frame", "d_begin", "d_end", "q_begin", "q_end",
              "s_begin", "s_end", "length", "identities", "mismatches", "positives", "gap_openings",
              "gaps", "backtraced")


def read_tap(path, max_records=None):
    """Returns (header, records); header holds the reference's score matrix + statistics constants."""
    buf = open(path, "rb").read()
    pos = 0
    out = []

    def i32():
        nonlocal pos
        v = struct.unpack_from("<i", buf, pos)[0]
        pos += 4
        return v

    def f64():
        nonlocal pos
        v = struct.unpack_from("<d", buf, pos)[0]
        pos += 8
        return v

    def raw(n, dtype):
        nonlocal pos
        a = np.frombuffer(buf, dtype=dtype, count=n, offset=pos).copy()
        pos += n
        return a

    header = None
    while pos < len(buf) and (max_records is None or len(out) < max_records):
        magic = i32()
        if magic == MAGIC_MTX:
            header = {"gap_open": i32(), "gap_extend": i32(), "lambda": f64(), "ln_k": f64(),
                      "db_letters": f64(), "max_evalue": f64()}
            header["matrix8"] = raw(1024, np.int8).reshape(32, 32)
            continue
        assert magic in (MAGIC, MAGIC2), "bad tap record magic at %d" % (pos - 4)
        rec = {"flags": i32(), "hsp_values": i32(), "frame": i32(), "query_source_len": i32()}
        qlen = i32()
        has_cbs = i32()
        rec["query"] = raw(qlen, np.int8)
        rec["cbs"] = raw(qlen, np.int8) if has_cbs else None
        targets = []
        for _ in range(i32()):
            t = {"bin": i32(), "target_idx": i32(), "d_begin": i32(), "d_end": i32(), "cols": i32(),
                 "true_target_len": i32()}
            t["seq"] = raw(i32(), np.int8)
            t["matrix"] = None
            if magic == MAGIC2 and i32():
                t["matrix"] = raw(32 * 26, np.int8).reshape(26, 32)       # [target letter][query letter]
            targets.append(t)
        rec["targets"] = targets
        hsps = []
        n_hsps = i32()
        rec["refused"] = n_hsps < 0                 # the reference threw inside this call
        for _ in range(max(n_hsps, 0)):
            h = {k: i32() for k in HSP_FIELDS}
            h["evalue"] = f64()
            h["bit_score"] = f64()
            h["transcript"] = raw(i32(), np.uint8)
            hsps.append(h)
        rec["hsps"] = hsps
        out.append(rec)
    return header, out


MAGIC_BLK = 0x314B4C42
MAGIC_EXT = 0x31545845


def read_ext_tap(path, max_records=None):
    """Reader for $DIAMOND_TAP_EXT files (oracle/ref_tap.cpp, second seam: Extension::extend).
    Returns (cfg, records): cfg holds the seed-stage configuration and both sequence blocks;
    records[i] = {query_id, hits (structured array), matches}."""
    buf = open(path, "rb").read()
    pos = 0

    def i32():
        nonlocal pos
        v = struct.unpack_from("<i", buf, pos)[0]
        pos += 4
        return v

    def i64():
        nonlocal pos
        v = struct.unpack_from("<q", buf, pos)[0]
        pos += 8
        return v

    def f64():
        nonlocal pos
        v = struct.unpack_from("<d", buf, pos)[0]
        pos += 8
        return v

    def raw(n, dtype):
        nonlocal pos
        a = np.frombuffer(buf, dtype=dtype, count=n, offset=pos).copy()
        pos += n * np.dtype(dtype).itemsize
        return a

    hit_dtype = np.dtype([("query", "<u4"), ("subject", "<i8"), ("seed_offset", "<i4"), ("score", "<i4")])
    cfg, out = None, []
    while pos < len(buf) and (max_records is None or len(out) < max_records):
        magic = i32()
        if magic == MAGIC_BLK:
            cfg = {"seedp_bits": i32(), "index_chunks": i32(), "hamming_filter_id": i32()}
            shapes = []
            for _ in range(i32()):
                sh = {"length": i32(), "weight": i32(), "mask": i32()}
                sh["positions"] = [i32() for _ in range(sh["weight"])]
                shapes.append(sh)
            cfg["shapes"] = shapes
            cfg["reduction"] = np.array([i32() for _ in range(32)], dtype=np.int32)
            cfg["seed_complexity_cut"], cfg["ungapped_evalue"], cfg["gapped_filter_evalue"] = f64(), f64(), f64()
            cfg["query_contexts"] = i32()
            for name in ("query", "target"):
                n = i32()
                raw_len = i64()
                data = raw(raw_len, np.int8)
                limits = raw(n + 1, np.int64)
                cfg[name] = {"n": n, "data": data, "limits": limits}
            continue
        assert magic == MAGIC_EXT, "bad ext tap magic at %d" % (pos - 4)
        rec = {"query_id": i32()}
        nh = i32()
        rec["hits"] = np.frombuffer(buf, dtype=hit_dtype, count=nh, offset=pos).copy()
        pos += nh * hit_dtype.itemsize
        matches = []
        for _ in range(i32()):
            m = {"target_block_id": i32(), "filter_score": i32(), "filter_evalue": f64(), "ungapped_score": i32()}
            hsps = []
            for _ in range(i32()):
                h = {k: i32() for k in HSP_FIELDS}
                h["evalue"] = f64()
                h["bit_score"] = f64()
                h["transcript"] = raw(i32(), np.uint8)
                hsps.append(h)
            m["hsps"] = hsps
            matches.append(m)
        rec["matches"] = matches
        out.append(rec)
    return cfg, out


def read_gf_tap(path, max_records=None):
    """Reader for $DIAMOND_TAP_GF files (oracle/ref_tap.cpp, third seam: Extension::gapped_filter).
    records[i] = {gapped_filter_evalue, gapped_filter_evalue1, diag_score, window, gap_open, gap_extend, query_offset,
    qlen, cbs (int8 array or None), targets: [{block_id, cutoff1, cutoff2, hits (n x 4 int32: i j score frame)}], out}."""
    buf = open(path, "rb").read()
    pos, recs = 0, []
    while pos < len(buf) and (max_records is None or len(recs) < max_records):
        magic, = struct.unpack_from("<i", buf, pos)
        assert magic == 0x314c4647, hex(magic)
        ev, ev1, diag, window, go, ge, qoff, qlen, has_cbs = struct.unpack_from("<ddiiiiqii", buf, pos + 4)
        pos += 4 + struct.calcsize("<ddiiiiqii")
        cbs = None
        if has_cbs:
            cbs = np.frombuffer(buf, np.int8, qlen, pos).copy()
            pos += qlen
        n, = struct.unpack_from("<i", buf, pos)
        pos += 4
        targets = []
        for _ in range(n):
            bid, c1, c2, nh = struct.unpack_from("<iiii", buf, pos)
            pos += 16
            hits = np.frombuffer(buf, "<i4", nh * 4, pos).reshape(nh, 4).copy()
            pos += nh * 16
            targets.append(dict(block_id=bid, cutoff1=c1, cutoff2=c2, hits=hits))
        n_out, = struct.unpack_from("<i", buf, pos)
        pos += 4
        out = np.frombuffer(buf, "<i4", n_out, pos).copy()
        pos += 4 * n_out
        recs.append(dict(gapped_filter_evalue=ev, gapped_filter_evalue1=ev1, diag_score=diag, window=window, gap_open=go,
                         gap_extend=ge, query_offset=qoff, qlen=qlen, cbs=cbs, targets=targets, out=out))
    return recs


def read_tantan_tap(path, max_records=None):
    """Reader for $DIAMOND_TAP_TANTAN files (oracle/ref_tap.cpp, fourth seam: Util::tantan::mask).
    Returns (hdr, records): hdr = {lr (32x32 float32), p_repeat, p_repeat_end, repeat_growth, p_mask};
    records[i] = {mask_mode, before (int8[]), after (int8[]), ranges (n x 2 int32)}."""
    buf = open(path, "rb").read()
    magic, = struct.unpack_from("<i", buf, 0)
    assert magic == 0x484e4154, hex(magic)
    lr = np.frombuffer(buf, "<f4", 1024, 4).reshape(32, 32).copy()
    p = np.frombuffer(buf, "<f4", 4, 4 + 4096)
    hdr = dict(lr=lr, p_repeat=p[0], p_repeat_end=p[1], repeat_growth=p[2], p_mask=p[3])
    pos, recs = 4 + 4096 + 16, []
    while pos < len(buf) and (max_records is None or len(recs) < max_records):
        magic, n, mode = struct.unpack_from("<iii", buf, pos)
        assert magic == 0x314e4154, hex(magic)
        pos += 12
        before = np.frombuffer(buf, np.int8, n, pos).copy()
        after = np.frombuffer(buf, np.int8, n, pos + n).copy()
        pos += 2 * n
        nr, = struct.unpack_from("<i", buf, pos)
        pos += 4
        ranges = np.frombuffer(buf, "<i4", 2 * nr, pos).reshape(nr, 2).copy()
        pos += 8 * nr
        recs.append(dict(mask_mode=mode, before=before, after=after, ranges=ranges))
    return hdr, recs


def read_cbs_tap(path, max_records=None):
    """Reader for $DIAMOND_TAP_CBS files (oracle/ref_tap.cpp, fifth seam: Stats::adjust_matrix and Stats::TargetMatrix).
    Returns (hdr, adj, tmx): hdr = {cbs, scale, ideal_lambda, ungapped_lambda, angle, joint_probs (20x20), background (20), matrix8};
    adj[i] = {query_comp, query_len, cbs, target, rule}; tmx[i] = {query_comp, query_len, cbs, rule, target, scores (26x32), score_min, score_max}."""
    buf = open(path, "rb").read()
    pos, hdr, adj, tmx = 0, None, [], []
    while pos < len(buf) and (max_records is None or len(adj) + len(tmx) < max_records):
        magic, = struct.unpack_from("<i", buf, pos)
        pos += 4
        if magic == 0x31484243:
            cbs, scale, il, ul, angle = struct.unpack_from("<iiddd", buf, pos)
            pos += 32
            jp = np.frombuffer(buf, "<f8", 400, pos).reshape(20, 20).copy(); pos += 3200
            bg = np.frombuffer(buf, "<f8", 20, pos).copy(); pos += 160
            m8 = np.frombuffer(buf, np.int8, 1024, pos).reshape(32, 32).copy(); pos += 1024
            hdr = dict(cbs=cbs, scale=scale, ideal_lambda=il, ungapped_lambda=ul, angle=angle, joint_probs=jp, background=bg, matrix8=m8)
            continue
        comp = np.frombuffer(buf, "<f8", 20, pos).copy(); pos += 160
        if magic == 0x314a4441:
            qlen, cbs, tlen = struct.unpack_from("<iii", buf, pos); pos += 12
            t = np.frombuffer(buf, np.int8, tlen, pos).copy(); pos += tlen
            rule, = struct.unpack_from("<i", buf, pos); pos += 4
            adj.append(dict(query_comp=comp, query_len=qlen, cbs=cbs, target=t, rule=rule))
        else:
            assert magic == 0x31584d54, hex(magic)
            qlen, cbs, rule, tlen = struct.unpack_from("<iiii", buf, pos); pos += 16
            t = np.frombuffer(buf, np.int8, tlen, pos).copy(); pos += tlen
            sc = np.frombuffer(buf, np.int8, 32 * 26, pos).reshape(26, 32).copy(); pos += 32 * 26
            smin, smax = struct.unpack_from("<ii", buf, pos); pos += 8
            tmx.append(dict(query_comp=comp, query_len=qlen, cbs=cbs, rule=rule, target=t, scores=sc, score_min=smin, score_max=smax))
    return hdr, adj, tmx


F3_HSP_FIELDS = ("swipe_target", "score", "frame", "q_begin", "q_end", "s_begin", "s_end", "qs_begin", "qs_end", "length", "identities",
                 "mismatches", "positives", "gap_openings", "gaps")


def read_3frame_tap(path, max_records=None):
    """Reader for $DIAMOND_TAP_3F files (oracle/ref_tap.cpp, sixth seam: banded_3frame_swipe, blastx -F).
    Returns (hdr, records): hdr = {gap_open, gap_extend, frame_shift, db_letters, max_evalue, matrix8};
    records[i] = {strand, score_only, dna_len, frames [3 x int8[]], targets [{target_idx, d_begin, d_end, cols, seq}], hsps [dict]}."""
    buf = open(path, "rb").read()
    pos, hdr, recs = 0, None, []

    def i32():
        nonlocal pos
        v = struct.unpack_from("<i", buf, pos)[0]
        pos += 4
        return v

    def f64():
        nonlocal pos
        v = struct.unpack_from("<d", buf, pos)[0]
        pos += 8
        return v

    def raw(n, dtype):
        nonlocal pos
        a = np.frombuffer(buf, dtype=dtype, count=n, offset=pos).copy()
        pos += n
        return a

    while pos < len(buf) and (max_records is None or len(recs) < max_records):
        magic = i32()
        if magic == 0x31483346:
            hdr = {"gap_open": i32(), "gap_extend": i32(), "frame_shift": i32(), "db_letters": f64(), "max_evalue": f64()}
            hdr["matrix8"] = raw(1024, np.int8).reshape(32, 32)
            continue
        assert magic == 0x31533346, hex(magic)
        rec = {"strand": i32(), "score_only": i32(), "dna_len": i32()}
        rec["frames"] = [raw(i32(), np.int8) for _ in range(3)]
        rec["targets"] = []
        for _ in range(i32()):
            t = {"target_idx": i32(), "d_begin": i32(), "d_end": i32(), "cols": i32()}
            t["seq"] = raw(i32(), np.int8)
            rec["targets"].append(t)
        rec["hsps"] = []
        for _ in range(i32()):
            h = {k: i32() for k in F3_HSP_FIELDS}
            h["evalue"], h["bit_score"] = f64(), f64()
            h["transcript"] = raw(i32(), np.uint8)
            rec["hsps"].append(h)
        recs.append(rec)
    return hdr, recs
