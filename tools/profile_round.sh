#!/bin/bash
# Collects the round's profiles on a GPU box (run through gpurun from the repo root):
#   kernel-trace + stats of the bench (score only, and with the end-to-end legs), and two PMC passes
#   (FETCH_SIZE, WRITE_SIZE; counters in their own runs, kernel-trace only).  Output: gpurun_out/prof/
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
python bench.py > "$OUT/bench_default_unprofiled.json" 2> "$OUT/bench_default.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/score_only" -- python3 "$ROOT/bench.py" --no-cpu --no-e2e --no-cfg5 --no-cfg2 > "$OUT/bench_score_only_under_rocprof.json" 2> "$OUT/score_only.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/with_e2e" -- python3 "$ROOT/bench.py" --no-cpu --no-cfg5 --no-cfg2 > "$OUT/bench_under_rocprof.json" 2> "$OUT/with_e2e.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --no-cpu --no-e2e --no-cfg5 --no-cfg2 --steps 3 --warmup 1 > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" --no-cpu --no-e2e --no-cfg5 --no-cfg2 --steps 3 --warmup 1 > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
# the traces are large: keep the stats and the counter tables only
find "$OUT" -name '*kernel_trace.csv' -path '*score_only*' -delete
find "$OUT" -name '*kernel_trace.csv' -path '*with_e2e*' -delete
find "$OUT" -name '*kernel_trace.csv' -path '*pmc_*' -delete
ls -R "$OUT" | head -50
