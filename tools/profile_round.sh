#!/bin/bash
# Collects a round's profiles on a GPU box (run through gpurun from the repo root):
#     tools/profile_round.sh r4 [default|cfg5]
#   default: 1. the default bench line, un-profiled;
#            2. rocprofv3 --kernel-trace --stats of the bench (score + masks legs only with ONE batch in flight, so that a launch's
#               duration is its own and not that of two launches sharing the chip; and the default line with the end-to-end legs);
#            3. PMC passes, each in its own run with --kernel-trace only: FETCH_SIZE, WRITE_SIZE, three SQ passes of 8 counters
#               (ONLY=sq3: the third one alone, the instruction classes).
#   cfg5:    the same stats + PMC passes for `--workload cfg5` (50M points, cones; the 50M-point refit scan).
#   cfg2:    the same for `--workload cfg2` (BASELINE configs[1]: 1M points, 6 dense primitives).
#   stats:   the diag build's event counters of one score launch per workload (tools/s4_stats.py -> s4_stats_<wl>.json) and the
#            issue-rate microbenchmark (tools/ubench/count_seq): the dynamic side and the prices of tools/isa_account.py.
# The program itself follows `--` (no env / bash -c hop: the profiler's library has initialised the GPU by then).
# Output: gpurun_out/prof_<round>/ (tools/collect_profiles.py copies what is to be judged into profiles/<round>/).
set -e
ROOT=$(pwd)
RND=${1:-r3}
WHAT=${2:-default}
OUT=$ROOT/gpurun_out/prof_$RND
mkdir -p "$OUT"
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"
SQ3="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM"
SQ2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY"
# what the default line's score launch looks like for a workload (super-tile lists or not: the library's rule with batches in flight):
# the profiled passes run ONE batch at a time and are told to launch the same instantiation (bench.py --lists on|off)
lists_of() {   # $1 = workload
    python3 - "$1" "$OUT" <<'PYEOF'
import json, subprocess, sys, os
wl, out = sys.argv[1], sys.argv[2]
f = os.path.join(out, "lists_probe_%s.json" % wl)
if not os.path.exists(f):
    r = subprocess.run([sys.executable, "bench.py", "--workload", wl, "--no-cpu", "--no-cpu-baseline", "--no-cfg5", "--no-cfg2", "--no-f32", "--no-per-kind", "--no-e2e",
                        "--steps", "40", "--warmup", "5", "--spread-regions", "0", "--detail-out", os.path.join(out, "lists_probe_%s_detail.json" % wl)],
                       capture_output=True, text=True, cwd=os.environ.get("RH_ROOT", "."))
    open(f, "w").write(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "{}")
try:
    print("on" if json.load(open(f)).get("score_lists") else "off")
except Exception:
    print("off")
PYEOF
}
export RH_ROOT=$ROOT
if [ "$WHAT" = "stats" ]; then
    for w in cfg2 cfg3 cfg5; do LISTS=$(lists_of $w) WL=$w python tools/s4_stats.py > "$OUT/s4_stats_$w.log" 2>&1; cp gpurun_out/s4_stats_$w.json "$OUT/"; echo "stats $w done"; done
    timeout -k 10 300 tools/ubench/count_seq > "$OUT/ubench_count_seq.txt" 2>&1
    echo "ubench done"; exit 0
fi
if [ "$ONLY" = "sq3" ]; then
    cd /tmp && export TMPDIR=/tmp
    if [ "$WHAT" = "default" ]; then LI=$(lists_of cfg3); B="$ROOT/bench.py --no-cpu --no-cfg5 --no-cfg2 --no-f32 --no-per-kind"; SFX=""; else LI=$(lists_of $WHAT); B="$ROOT/bench.py --workload $WHAT --no-cpu --no-cfg5 --no-cfg2 --no-f32 --no-per-kind"; SFX="_$WHAT"; fi
elif [ "$WHAT" = "default" ]; then
    python bench.py --detail-out "$OUT/bench_default_detail.json" > "$OUT/bench_default_unprofiled.json" 2> "$OUT/bench_default.err"
    echo "bench done"
    LI=$(lists_of cfg3)
    echo "lists: $LI"
    cd /tmp && export TMPDIR=/tmp
    B="$ROOT/bench.py --no-cpu --no-cfg5 --no-cfg2 --no-f32 --no-per-kind"
    SFX=""
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/score_only" -- python3 $B --no-e2e --in-flight 1 --lists $LI --detail-out "$OUT/bench_score_only_under_rocprof_detail.json" > "$OUT/bench_score_only_under_rocprof.json" 2> "$OUT/score_only.err"
    echo "stats 1 done"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/with_e2e" -- python3 $B --detail-out "$OUT/bench_under_rocprof_detail.json" > "$OUT/bench_under_rocprof.json" 2> "$OUT/with_e2e.err"
    echo "stats 2 done"
else
    LI=$(lists_of $WHAT)
    echo "lists: $LI"
    cd /tmp && export TMPDIR=/tmp
    B="$ROOT/bench.py --workload $WHAT --no-cpu --no-cfg5 --no-cfg2 --no-f32 --no-per-kind"
    SFX="_$WHAT"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/score_only$SFX" -- python3 $B --no-e2e --in-flight 1 --lists $LI --steps 60 --warmup 10 --detail-out "$OUT/bench_score_only_under_rocprof${SFX}_detail.json" > "$OUT/bench_score_only_under_rocprof$SFX.json" 2> "$OUT/score_only$SFX.err"
    echo "stats $WHAT done"
fi
S="$B --no-e2e --in-flight 1 --lists $LI --steps 3 --warmup 1 --prewarm-ms 0 --detail-out $OUT/pmc_detail$SFX.json"
rocprofv3 --kernel-trace --pmc $SQ3 --output-format csv -d "$OUT/pmc_sq3$SFX" -- python3 $S > "$OUT/pmc_sq3$SFX.json" 2> "$OUT/pmc_sq3$SFX.err"
echo "pmc sq3 done"
if [ "$ONLY" = "sq3" ]; then find "$OUT" -name '*kernel_trace.csv' -delete; exit 0; fi
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch$SFX" -- python3 $S > "$OUT/pmc_fetch$SFX.json" 2> "$OUT/pmc_fetch$SFX.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write$SFX" -- python3 $S > "$OUT/pmc_write$SFX.json" 2> "$OUT/pmc_write$SFX.err"
echo "pmc hbm done"
rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d "$OUT/pmc_sq1$SFX" -- python3 $S > "$OUT/pmc_sq1$SFX.json" 2> "$OUT/pmc_sq1$SFX.err"
rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d "$OUT/pmc_sq2$SFX" -- python3 $S > "$OUT/pmc_sq2$SFX.json" 2> "$OUT/pmc_sq2$SFX.err"
echo "pmc sq done"
# the traces are large: keep the stats and the counter tables only
find "$OUT" -name '*kernel_trace.csv' -delete
cd "$ROOT"
echo "now run: python3 tools/collect_profiles.py $RND (in the build container, after gpurun merged gpurun_out/)"
ls "$OUT" | head -60
