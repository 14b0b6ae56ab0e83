#!/usr/bin/env python3
"""Extracts the reference's motif table (src/masking/motifs.cpp: ~8000 abundant 8-mers that DIAMOND soft-masks during seed
enumeration) into diamond_amd/motifs.bin: little-endian uint64 Kmer<8> codes (base-20 polynomial over ARNDCQEGHILKMFPSTWYV).
The table is reference DATA: it is generated at build time where /root/reference exists (this container) and travels to the
GPU box next to the built libraries; it is not committed (.gitignore). Without it motif masking is off (--motif-masking 0).
usage: tools/make_motif_table.py [REFERENCE_ROOT] [OUT]"""
import os
import re
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "diamond_amd", "motifs.bin")
src = os.path.join(ref, "src", "masking", "motifs.cpp")
if not os.path.exists(src):
    print("make_motif_table: %s not found, leaving %s as it is" % (src, out))
    sys.exit(0)
AA = "ARNDCQEGHILKMFPSTWYV"
codes = set()
text = open(src).read()
text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)        # part of the array is commented out in the reference
text = re.sub(r"//[^\n]*", "", text)
for m in re.finditer(r'"([A-Z]{8})"', text):
    c = 0
    for ch in m.group(1):
        c = c * 20 + AA.index(ch)
    codes.add(c)
with open(out, "wb") as f:
    for c in sorted(codes):
        f.write(struct.pack("<Q", c))
print("make_motif_table: %d motifs -> %s" % (len(codes), out))
