#!/usr/bin/env python3
"""Instruction mix of every loop (a backward conditional branch) of at least 100 instructions in the dmnd:: kernels of a hipcc
-save-temps .s file: total, VALU, packed, DPP, LDS, VMEM, waitcnt, nop.   usage: tools/isa_loops2.py file.s [substring]"""
import re
import sys

text = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [i for i, l in enumerate(text) if l.startswith("_ZN4dmnd") and ": " in l and "@" in l and want in l]
for st in starts:
    end = next(i for i in range(st, len(text)) if "s_endpgm" in text[i])
    body = text[st:end + 1]
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if l.startswith(".LBB")}
    name = re.search(r"\d+(\w+?kernel)(I[\w]+?E)Ev", body[0])
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", l)
        if not (m and m.group(1) in labels and labels[m.group(1)] < i):
            continue
        ins = [x.strip() for x in body[labels[m.group(1)]:i + 1] if re.match(r"^\s+(v_|s_|ds_|global_|buffer_|flat_)", x)]
        if len(ins) < 100:
            continue
        k = {"valu": sum(x.startswith("v_") for x in ins), "pk": sum(x.startswith("v_pk") for x in ins), "dpp": sum("dpp" in x for x in ins),
             "lds": sum(x.startswith("ds_") for x in ins), "vmem": sum(x.startswith(("global_", "flat_", "buffer_")) for x in ins),
             "wait": sum(x.startswith("s_waitcnt") for x in ins), "nop": sum(x.startswith("s_nop") for x in ins)}
        print((name.group(1) + name.group(2)) if name else body[0][:50], len(ins), k)
