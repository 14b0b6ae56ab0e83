"""Block layout helper: sequences in the reference's SequenceSet form (the layout dmnd_upload_block expects)."""
import numpy as np


def sequence_set(data, off):
    """Lays sequences out as the reference's SequenceSet does (data/string_set.h:27-60): 256 x 0x1F perimeter padding,
    then seq, 0x1F, seq, 0x1F, ..., 256 x 0x1F. Returns (data int8[], limits int64[n+1])."""
    lens = np.diff(off)
    limits = 256 + np.concatenate([[0], np.cumsum(lens + 1)])
    out = np.full(int(limits[-1]) + 256, 31, np.int8)
    idx = np.repeat(limits[:-1] - off[:-1], lens) + np.arange(off[-1])
    out[idx] = data
    return out, limits.astype(np.int64)
