"""Six-frame translation (dmnd_translate, host-only C ABI entry; SURVEY 8f blastx) against the translated query block the
genuine reference built for the same reads (header of tests/golden/ext_blastx.tap): frame order, reverse strand, stop
codons, short-ORF masking. CPU only (no device call)."""
import os
import numpy as np

from tapfile import read_ext_tap
from diamond_amd import hip

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NT = {c: i for i, c in enumerate("ACGTN")}


def _reads():
    seqs = [l.strip() for l in open(os.path.join(GOLDEN, "blastx_reads.fna")) if not l.startswith(">")]
    lens = np.array([len(s) for s in seqs], np.int64)
    off = np.concatenate([[0], np.cumsum(lens)])
    dna = np.array([NT[c] for s in seqs for c in s], np.int8)
    return dna, off


def test_translated_block_equals_reference_block():
    cfg, _ = read_ext_tap(os.path.join(GOLDEN, "ext_blastx.tap"), max_records=1)
    assert cfg["query_contexts"] == 6
    dna, off = _reads()
    qd, ql = hip.translated_block(dna, off)
    assert np.array_equal(ql, cfg["query"]["limits"])
    n = int(ql[-1])
    assert np.array_equal(qd[:n], cfg["query"]["data"][:n])
    assert (qd[:n] == 24).any() and (qd[:n] == 23).any()          # stop codons and masked short ORFs occur


def test_translation_edge_cases():
    assert [len(f) for f in hip.translate(np.zeros(2, np.int8))] == [0] * 6        # shorter than a codon
    assert [len(f) for f in hip.translate(np.zeros(10, np.int8))] == [3, 3, 2, 3, 3, 2]
    f = hip.translate(np.array([0, 3, 2, 4, 4, 4, 2, 1, 4], np.int8))               # ATG NNN GCN
    assert f[0].tolist() == [12, 23, 0]                                            # M X A (wobble N that cannot change the residue)
    f = hip.translate(np.array([3, 0, 0], np.int8))                                # TAA = stop
    assert f[0].tolist() == [24]
    assert f[3].tolist() == [hip.translate(np.array([3, 3, 0], np.int8))[0][0]]    # reverse complement of TAA is TTA
