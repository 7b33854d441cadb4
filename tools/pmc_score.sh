#!/bin/bash
# SQ counters of the v3 score kernels: tools/pmc_score.sh "<counters>" [ENV=..]...
ROOT=$(pwd); CNT="$1"; shift
OUT=$ROOT/gpurun_out/pmc_score; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
STEPS=5 PRE=5 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$OUT" -- python3 "$ROOT/tools/score_only.py" > "$OUT/out.txt" 2> "$OUT/err.log"
f=$(find "$OUT" -name '*counter_collection.csv' | head -1)
python3 - "$f" <<PY
import csv,sys
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(float)); n=defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].replace("(anonymous namespace)::","")[:40]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k,v in acc.items():
    if "score_groups_all" in k or "prep_binned" in k or "score4" in k:
        print(k, "dispatches", len(n[k]))
        for c,x in sorted(v.items()): print("    %-28s %14.0f per dispatch" % (c, x/len(n[k])))
PY
