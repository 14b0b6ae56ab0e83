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


def test_translation_options_strand_gencode_min_orf():
    """dmnd_translate_opts: --strand (frames of the other strand are all mask letters, same lengths), --query-gencode (codons that
    differ between NCBI tables), --min-orf, and the reference's error for a table number it does not have."""
    import pytest
    dna, off = _reads()
    read = dna[off[0]:off[1]]
    both = hip.translate(read)
    plus, minus = hip.translate(read, strands=1), hip.translate(read, strands=2)
    for f in range(6):
        assert len(plus[f]) == len(minus[f]) == len(both[f])
        assert np.array_equal(plus[f], both[f]) if f < 3 else (plus[f] == 23).all()
        assert np.array_equal(minus[f], both[f]) if f >= 3 else (minus[f] == 23).all()
    AA = "ARNDCQEGHILKMFPSTWYVBJZX*_"
    codon = lambda s: np.array([NT[c] for c in s], np.int8)
    # TGA: stop in table 1, W in 4; AGA: R in 1, stop in 2, S in 5, G in 13; CTG: L in 1, S in 12, A in 26; TAG: stop in 1, L in 16 / 22, Q in 6
    for code, dna3, want in ((1, "TGA", "*"), (4, "TGA", "W"), (11, "TGA", "*"), (25, "TGA", "G"), (1, "AGA", "R"), (2, "AGA", "*"), (5, "AGA", "S"), (13, "AGA", "G"),
                             (12, "CTG", "S"), (26, "CTG", "A"), (16, "TAG", "L"), (6, "TAG", "Q"), (3, "CTA", "T"), (9, "AAA", "N"), (14, "TAA", "Y"), (24, "AGG", "K"),
                             (21, "ATA", "M"), (23, "TTA", "*"), (22, "TCA", "*"), (10, "TGA", "C")):
        assert AA[hip.translate(codon(dna3), gencode=code)[0][0]] == want, (code, dna3)
    # reverse strand uses the same table: TCA reversed-complemented is TGA
    assert AA[hip.translate(codon("TCA"), gencode=4)[3][0]] == "W"
    for bad in (0, 7, 8, 15, 17, 20, 27, 33):
        with pytest.raises(hip.DiamondHipError, match="Invalid genetic code id"):
            hip.translate(read, gencode=bad)
    # --min-orf: an ORF of 25 residues between stops survives the automatic rule of a 300-letter frame (40) only when asked for
    orf = "ATG" + "GCT" * 24
    s = codon("TAA" + orf + "TAA" + "GCT" * 250)
    auto, asked = hip.translate(s)[0], hip.translate(s, min_orf=20)[0]
    assert (auto[1:26] == 23).all() and asked[1] == 12 and (asked[2:26] == 0).all()
    assert (hip.translate(s, min_orf=300)[0][:300] != 0).all()
