"""ctypes binding of the deterministic synthetic workload generator (csrc/synth.c)."""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class _Cfg(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("families", ctypes.c_int64), ("members", ctypes.c_int),
                ("queries", ctypes.c_int64), ("decoy_frac", ctypes.c_double),
                ("len_mean", ctypes.c_double), ("len_sd", ctypes.c_double),
                ("len_min", ctypes.c_int), ("len_max", ctypes.c_int),
                ("sub_lo", ctypes.c_double), ("sub_hi", ctypes.c_double),
                ("q_lo", ctypes.c_double), ("q_hi", ctypes.c_double), ("indel", ctypes.c_double)]


def _lib():
    path = os.path.join(_HERE, "libdmnd_synth.so")
    if not os.path.exists(path):
        raise RuntimeError("libdmnd_synth.so missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(path)
    lib.synth_generate.restype = ctypes.c_int
    lib.synth_write_fasta.restype = ctypes.c_int
    return lib


def generate(families, members=10, queries=1000, seed=20260923, decoy_frac=0.05, len_mean=300.0, len_sd=80.0,
             len_min=50, len_max=2000, sub=(0.1, 0.6), qsub=(0.2, 0.5), indel=0.02, family=False):
    """Returns (db_letters int8[], db_offsets int64[n+1], q_letters, q_offsets); with family=True also the
    family index each query was derived from (-1 = decoy)."""
    lib = _lib()
    cfg = _Cfg(seed, families, members, queries, decoy_frac, len_mean, len_sd, len_min, len_max,
               sub[0], sub[1], qsub[0], qsub[1], indel)
    dd = ctypes.POINTER(ctypes.c_int8)()
    do = ctypes.POINTER(ctypes.c_int64)()
    qd = ctypes.POINTER(ctypes.c_int8)()
    qo = ctypes.POINTER(ctypes.c_int64)()
    qf = ctypes.POINTER(ctypes.c_int64)()
    dn = ctypes.c_int64()
    qn = ctypes.c_int64()
    rc = lib.synth_generate(ctypes.byref(cfg), ctypes.byref(dd), ctypes.byref(do), ctypes.byref(dn),
                            ctypes.byref(qd), ctypes.byref(qo), ctypes.byref(qn), ctypes.byref(qf))
    if rc != 0:
        raise MemoryError("synth_generate failed")
    try:
        db_off = np.ctypeslib.as_array(do, shape=(dn.value + 1,)).copy()
        q_off = np.ctypeslib.as_array(qo, shape=(qn.value + 1,)).copy()
        db = np.ctypeslib.as_array(dd, shape=(int(db_off[-1]),)).copy()
        q = np.ctypeslib.as_array(qd, shape=(int(q_off[-1]),)).copy()
        fam = np.ctypeslib.as_array(qf, shape=(qn.value,)).copy()
    finally:
        for p in (dd, do, qd, qo, qf):
            lib.synth_free(p)
    if family:
        return db, db_off, q, q_off, fam
    return db, db_off, q, q_off


def write_fasta(path, prefix, data, off):
    lib = _lib()
    data = np.ascontiguousarray(data, dtype=np.int8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    rc = lib.synth_write_fasta(path.encode(), prefix.encode(), data.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)),
                               off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.c_int64(len(off) - 1))
    if rc != 0:
        raise OSError("cannot write " + path)


# ---- DNA reads for blastx (BASELINE config C4) ------------------------------------------------------------------------
_AA = "ARNDCQEGHILKMFPSTWYVBJZX*_"
_STD_CODE = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"      # NCBI table 1, base order TCAG
_NT = "ACGTN"


def back_translate(q, q_off, seed=1, flank=(0, 60), reverse_frac=0.5):
    """Synthetic DNA reads: every protein query is written with uniformly chosen synonymous codons (standard code),
    wrapped in random flanks (so that the coding frame varies) and reverse-complemented with probability reverse_frac.
    Returns (dna int8[] in ACGTN = 0..4 coding, offsets int64[n+1])."""
    rng = np.random.default_rng(seed)
    base = {"T": 3, "C": 1, "A": 0, "G": 2}
    order = "TCAG"
    codons = [[] for _ in range(20)]
    for i, aa in enumerate(_STD_CODE):
        if aa != "*":
            codons[_AA.index(aa)].append((base[order[i // 16]], base[order[(i // 4) % 4]], base[order[i % 4]]))
    n_cod = np.array([len(c) for c in codons])
    table = np.zeros((20, 6, 3), np.int8)
    for a, cs in enumerate(codons):
        for k, c in enumerate(cs):
            table[a, k] = c
    q = np.asarray(q)
    aa = np.where(q < 20, q, 0).astype(np.int64)                 # ambiguity letters do not occur in the generator's output
    pick = (rng.random(aa.size) * n_cod[aa]).astype(np.int64)
    coding = table[aa, pick].reshape(-1)                          # 3 nt per residue, concatenated over all queries
    n = len(q_off) - 1
    left = rng.integers(flank[0], flank[1] + 1, n)
    right = rng.integers(flank[0], flank[1] + 1, n)
    rev = rng.random(n) < reverse_frac
    lens = 3 * np.diff(q_off) + left + right
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    out = rng.integers(0, 4, int(off[-1])).astype(np.int8)        # flanks = random bases
    for i in range(n):
        a, b = 3 * int(q_off[i]), 3 * int(q_off[i + 1])
        s = int(off[i]) + int(left[i])
        out[s:s + (b - a)] = coding[a:b]
        if rev[i]:
            r = out[off[i]:off[i + 1]][::-1]
            out[off[i]:off[i + 1]] = np.where(r < 4, 3 - r, r)
    return out, off


def indel_reads(dna, off, seed=1, deletion=0.003, insertion=0.003, substitution=0.01):
    """Reads with sequencing errors that break the reading frame (what blastx -F is for): every base is dropped with probability
    `deletion`, preceded by a random base with probability `insertion`, replaced with probability `substitution`.
    Returns (dna, offsets)."""
    rng = np.random.default_rng(seed)
    reads = []
    for i in range(len(off) - 1):
        r = dna[off[i]:off[i + 1]]
        x = rng.random(len(r))
        keep = x >= deletion
        ins = (x >= deletion) & (x < deletion + insertion)
        sub = rng.random(len(r)) < substitution
        r = np.where(sub, rng.integers(0, 4, len(r)).astype(r.dtype), r)
        parts = np.empty(2 * len(r), r.dtype)
        parts[0::2] = rng.integers(0, 4, len(r))
        parts[1::2] = r
        mask = np.empty(2 * len(r), bool)
        mask[0::2] = ins
        mask[1::2] = keep
        reads.append(parts[mask])
    o = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
    return np.concatenate(reads), o


def write_dna_fasta(path, prefix, dna, off):
    lut = np.frombuffer(_NT.encode(), np.uint8)
    with open(path, "wb") as f:
        for i in range(len(off) - 1):
            f.write((">%s%d\n" % (prefix, i)).encode())
            f.write(lut[dna[off[i]:off[i + 1]]].tobytes())
            f.write(b"\n")
