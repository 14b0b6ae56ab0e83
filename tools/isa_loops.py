#!/usr/bin/env python3
"""Per-kind instruction counts of every innermost loop of the kernels in hipcc's -save-temps .s output.
usage: tools/isa_loops.py file.s KERNEL_SYMBOL_SUBSTRING ...   (writes nothing; prints one line per loop)"""
import re
import sys

text = open(sys.argv[1]).read().splitlines()
for pat in sys.argv[2:]:
    start = next(i for i, l in enumerate(text) if l.startswith("_Z") and pat in l and ": " in l and "@" in l)
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    body = text[start:end + 1]
    for h in [i for i, l in enumerate(body) if "Inner Loop Header" in l]:
        lab = h
        while not body[lab].startswith(".LBB"):
            lab -= 1
        name = body[lab].split(":")[0]
        back = [i for i, l in enumerate(body) if re.search(r"s_c?branch\w*\s+" + re.escape(name) + r"\b", l)]
        if not back:
            continue
        ins = [l.strip() for l in body[h:max(back) + 1] if re.match(r"^\s+(v_|s_|ds_|global_|buffer_|flat_)", l)]
        k = {"valu": sum(l.startswith("v_") for l in ins),
             "salu": sum(l.startswith("s_") and not l.startswith(("s_waitcnt", "s_nop")) for l in ins),
             "vmem": sum(l.startswith(("global_", "flat_", "buffer_")) for l in ins), "lds": sum(l.startswith("ds_") for l in ins),
             "waitcnt": sum(l.startswith("s_waitcnt") for l in ins), "nop": sum(l.startswith("s_nop") for l in ins)}
        print(pat, name, len(ins), k)
