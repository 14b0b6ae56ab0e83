#!/bin/bash
# round 4: PMC passes of the tantan lane kernel on 3.0e8 letters (each group in its own rocprofv3 --kernel-trace --pmc run)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/mask"; mkdir -p "$OUT"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p$i" -o pmc -- python -m pytest "$ROOT/tests/test_gpu_c2_scale.py" -m gpu -x -q -s -k "mask" > "$OUT/p$i.log" 2>&1 || tail -3 "$OUT/p$i.log"
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tantan_lanes_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(k, "%.4g" % tot[k])
import json
json.dump({"kernel": "tantan_lanes_kernel, one launch over the 3.0e8-letter C2 reference block (tests/test_gpu_c2_scale.py -k mask)", "counters": dict(tot),
           "note": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; SQ_* cycle counters in quad-cycles, summed over the XCDs"}, open(out + "/pmc_tantan.json", "w"), indent=1, sort_keys=True)
PY
rm -rf "$OUT"/p[0-9]*
