#!/usr/bin/env python3
"""Instruction count of the innermost loops of a kernel in hipcc's -S output.
usage: tools/isa_loop.py file.s SUBSTRING_OF_KERNEL_SYMBOL ..."""
import re
import sys

text = open(sys.argv[1]).read().splitlines()
for pat in sys.argv[2:]:
    start = next(i for i, l in enumerate(text) if l.startswith("_Z") and pat in l and l.rstrip().split(":")[0].endswith("E") and ":" in l)
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    body = text[start:end + 1]
    headers = [i for i, l in enumerate(body) if "=>This Inner Loop Header" in l or "=>This Loop Header" in l]
    out = []
    for h in headers:
        name = body[h].split(":")[0].lstrip(".L")
        member = [i for i, l in enumerate(body) if i == h or ("Header=" + name + " ") in l or l.rstrip().endswith("Header=" + name)]
        a, b = min(member), max(member)
        while b + 1 < len(body) and not body[b + 1].startswith(".LBB"):
            b += 1
        ins = [l for l in body[a:b + 1] if re.match(r"^\s+(v_|s_|ds_|global_|buffer_|flat_)", l)]
        out.append((len(ins), sum("s_cbranch" in l for l in ins), sum(l.strip().startswith(("global_", "flat_", "buffer_")) for l in ins),
                    sum(l.strip().startswith("ds_") for l in ins), sum("s_waitcnt" in l for l in ins)))
    big = max(out) if out else None
    print(pat, "-> largest loop: %d instructions, %d branches, %d vmem, %d lds, %d waitcnt" % big if big else "no loop")
