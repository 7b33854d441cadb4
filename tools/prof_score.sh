#!/bin/bash
# per-kernel times of the score step under rocprofv3 (run through gpurun from the repo root): tools/prof_score.sh [tag] [env...]
ROOT=$(pwd); TAG=${1:-s}; shift
OUT=$ROOT/gpurun_out/prof_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --no-cpu --no-e2e --no-cfg5 --no-cfg2 --steps 100 --warmup 10 > "$OUT/bench.json" 2> "$OUT/err.log"
find "$OUT" -name '*kernel_trace.csv' -delete
f=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:12]:
    print("%-90s calls %6s avg %9.1f us total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
