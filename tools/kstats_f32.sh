#!/bin/bash
# per-kernel average durations of the bench's Float32-cloud leg under rocprofv3 (every kernel, sorted by total time)
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstats_f32; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3 --no-cpu --no-cfg5 --no-cfg2 --no-e2e --detail-out "$OUT/detail.json" "$@" > "$OUT/line.json" 2> "$OUT/err.log"
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv,sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    n=r["Name"].replace("(anonymous namespace)::","")
    if i < 24: print("%-90s calls %5s avg %8.2f us" % (n[:90], r["Calls"], float(r["AverageNs"])/1e3))
PY
python3 -c "
import json; d=json.load(open('$OUT/line.json')); print('ms_per_step', d['ms_per_step'], 'f32_ms', d.get('f32_ms'))"
