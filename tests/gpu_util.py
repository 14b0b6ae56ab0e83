"""Helpers shared by the -m gpu parity tests: pack reference tap records into HBM-resident blocks and
dmnd_dp_target items."""
import numpy as np
from diamond_amd import hip


def pack_records(recs):
    """Concatenates all queries / bias vectors / target sequences of tap records.
    Returns (qblock, tblock, cbs, items, meta) where meta[i] = (rec, target dict)."""
    return pack_records_m(recs)[:5]


def pack_records_m(recs):
    """pack_records + the adjusted matrices of the targets that carry one (t["matrix"], 26 or 32 rows of 32): such an item names its
    matrix in cbs_off (-2 - number) and gets no bias. Returns (qblock, tblock, cbs, items, meta, matrices int8[n, 32, 32])."""
    qparts, tparts, cparts, rows, meta, mats = [], [], [], [], [], []
    qoff = toff = coff = 0
    for rec in recs:
        q = rec["query"]
        qparts.append(q)
        this_q = qoff
        qoff += len(q)
        this_c = -1
        if rec["cbs"] is not None:
            cparts.append(rec["cbs"])
            this_c = coff
            coff += len(q)
        for t in rec["targets"]:
            tparts.append(t["seq"])
            c_off = this_c
            if t.get("matrix") is not None:
                m = np.full((32, 32), -128, np.int8)
                m[:len(t["matrix"])] = t["matrix"]
                c_off = -2 - len(mats)
                mats.append(m)
            rows.append((this_q, toff, c_off, len(q), len(t["seq"]), t["d_begin"], t["d_end"]))
            meta.append((rec, t))
            toff += len(t["seq"])
    items = np.array(rows, dtype=hip.DP_TARGET_DTYPE)
    qblock = np.concatenate(qparts).astype(np.int8)
    tblock = np.concatenate(tparts).astype(np.int8)
    cbs = np.concatenate(cparts).astype(np.int8) if cparts else np.zeros(0, np.int8)
    return qblock, tblock, cbs, items, meta, (np.stack(mats) if mats else np.zeros((0, 32, 32), np.int8))
