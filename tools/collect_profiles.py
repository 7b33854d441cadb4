#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<round>/ (tools/profile_round.sh) into the files kept under profiles/<round>/:
kernel_stats_*.csv, bench_*.json, pmc_hbm_traffic.json (KB per dispatch), pmc_sq_counters.json (per dispatch, summed over
the chip) and pmc_meta.json (source hash of the library the passes were taken on -- bench.py marks a replay stale when
the library has changed since)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    """the files of the LATEST pass only: gpurun merges a run's output into gpurun_out/ beside what earlier runs left
    there (other pids in the names), so everything older than the newest match by more than ten minutes is dropped"""
    files = glob.glob(pattern, recursive=True)
    if not files:
        return []
    t = max(os.path.getmtime(f) for f in files)
    return [f for f in files if os.path.getmtime(f) > t - 600]


def counters(dirs):
    acc = defaultdict(list)
    for d in dirs:
        for f in newest(d + "/**/*counter_collection.csv"):
            per = defaultdict(float)
            for r in csv.DictReader(open(f)):
                per[(r["Kernel_Name"], r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])   # one row per XCD
            for (k, c, _), v in per.items():
                acc[(k, c)].append(v)
    return acc


def main(rnd):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + rnd)
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    for name in ("bench_default_unprofiled.json", "bench_score_only_under_rocprof.json", "bench_under_rocprof.json",
                 "bench_score_only_under_rocprof_cfg5.json"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, name))
    for sub, out in (("score_only", "kernel_stats_bench_score_only.csv"), ("with_e2e", "kernel_stats_bench_with_e2e.csv"),
                     ("score_only_cfg5", "kernel_stats_bench_score_only_cfg5.csv")):
        f = newest(os.path.join(src, sub, "**", "*kernel_stats.csv"))
        if f:
            shutil.copy(max(f, key=os.path.getmtime), os.path.join(dst, out))
    rows = []
    for sfx in ("", "_cfg5"):     # the default workload's passes and the ones taken with --workload cfg5
        acc = counters([os.path.join(src, "pmc_fetch" + sfx), os.path.join(src, "pmc_write" + sfx)])
        if acc:
            rows = [{"kernel": k, "counter": c, "dispatches": len(v), "mean_KB": sum(v) / len(v), "max_KB": max(v)} for (k, c), v in acc.items()]
            rows.sort(key=lambda r: (r["counter"], r["kernel"]))
            json.dump(rows, open(os.path.join(dst, "pmc_hbm_traffic%s.json" % sfx), "w"), indent=1)
        acc = counters([os.path.join(src, "pmc_sq1" + sfx), os.path.join(src, "pmc_sq2" + sfx)])
        if acc:
            rows = [{"kernel": k, "counter": c, "dispatches": len(v), "mean": sum(v) / len(v), "max": max(v)} for (k, c), v in acc.items()]
            rows.sort(key=lambda r: (r["kernel"], r["counter"]))
            json.dump(rows, open(os.path.join(dst, "pmc_sq_counters%s.json" % sfx), "w"), indent=1)
            for r in rows:
                if "score4_kernel" in r["kernel"] and r["counter"] in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES"):
                    print(sfx or "default", r["kernel"][:60], r["counter"], r["mean"])
    import importlib.util
    spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "ransac.jl_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    json.dump({"lib_source_hash": b.source_hash(), "command": "tools/profile_round.sh %s default; tools/profile_round.sh %s cfg5" % (rnd, rnd),
               "passes": "bench.py [--workload cfg5] --no-cpu --no-cfg5 --no-cfg2 --no-f32 --no-e2e --steps 3 --warmup 1 --prewarm-ms 0, one rocprofv3 --pmc "
                         "run per counter group, the program itself after `--`"},
              open(os.path.join(dst, "pmc_meta.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r3")
