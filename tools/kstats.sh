#!/bin/bash
# per-kernel average durations of the bench's score / masks legs under rocprofv3: tools/kstats.sh <workload> [extra bench args]
W=${1:-cfg3}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstats_$W; rm -rf "$OUT"; mkdir -p "$OUT"
EX=""; [ "$W" = cfg5 ] && EX="--steps 60 --warmup 10"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu --no-cfg5 --no-cfg2 --no-f32 --no-e2e $EX --detail-out "$OUT/detail.json" "$@" > "$OUT/line.json" 2> "$OUT/err.log"
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"].replace("(anonymous namespace)::","")
    if any(k in n for k in ("score4_kernel","unpermute","clear_","prep_binned","refit_mask")):
        print("$W %-64s calls %4s avg %8.2f us" % (n[:64], r["Calls"], float(r["AverageNs"])/1e3))
PY
