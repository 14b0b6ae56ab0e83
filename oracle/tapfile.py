"""TEST INFRASTRUCTURE ONLY -- reader for the known-answer tap files written by
oracle/_ref/diamond_tap (oracle/ref_tap.cpp documents the record layout).  Each record is one
genuine call of the reference's DP::BandedSwipe::swipe (dp/dp.h:287)."""
import struct
import numpy as np

MAGIC = 0x31505753
MAGIC_MTX = 0x3158544D

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
        assert magic == MAGIC, "bad tap record magic at %d" % (pos - 4)
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
            targets.append(t)
        rec["targets"] = targets
        hsps = []
        for _ in range(i32()):
            h = {k: i32() for k in HSP_FIELDS}
            h["evalue"] = f64()
            h["bit_score"] = f64()
            h["transcript"] = raw(i32(), np.uint8)
            hsps.append(h)
        rec["hsps"] = hsps
        out.append(rec)
    return header, out
