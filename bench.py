#!/usr/bin/env python3
"""bench.py -- candidates scored / s (+ shapes / s) on the BASELINE.json workload.

A "step" = one pass of the hot path over one batch: B = 4096 candidate shapes (ground-truth
primitives jittered by 1 %) scored against subset 1 (S = N/32 points) of the
10M-point, 40-primitive, 30 %-outlier synthetic cloud (BASELINE.json configs[2], the config the
metric is quoted on), everything resident in HBM when the timed region starts.  With --gpus N
every rank holds a replica of the cloud, scores its own 4096-candidate slice of a global
N x 4096 batch and one RCCL int32 sum all-reduce per step gives every rank every score
(weak scaling; value = candidates all ranks scored / max-over-ranks time).

Started plainly with --gpus N > 1 (no torchrun), the script launches its N ranks itself -- before the parent
has made any GPU call -- and relays rank 0's line; it never falls back to fewer ranks than asked for.

One JSON line on rank 0.  Extra objects: `roofline` (the dominant score kernel against the bound it really
hits: FP64 vector-ALU issue slots; live HIP-event time, counter values replayed from profiles/rN and labelled
so), `roofline_refit` (the HBM-bound full-cloud scan, everything measured live), `masks_out` (the same step
with inlier masks written), `cpu_baseline` (the oracle, 1 thread, bounded sample; `cpu_baseline_mt`: OpenMP
steelman on this GPU's share of the host cores), `end_to_end` (rh_ransac on the same cloud, median of 5 runs),
`cfg2` / `cfg5` (BASELINE configs[1] and [4] on this GPU, child processes; cfg5 with an oracle check).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 FMA lanes x 2 x 2.4 GHz
SCORE_BYTES_PER_TEST = 48.25   # 6 x 8 B point+normal + 1 bit enabled + 1 bit mask
REFIT_BYTES_PER_POINT = 48.125
# FP64 vector instructions per (candidate, point) test in score_kernel's inner loop, counted from
# the gfx950 ISA of this build: v_add/v_mul/v_fma/v_cmp/v_rcp/v_rsq ... _f64
VALU_F64_PER_TEST = {"plane": 15, "sphere": 45, "cylinder": 59, "cone": 247}
# useful flops per test in the reference's arithmetic
FLOPS_PER_TEST = {"plane": 13, "sphere": 20, "cylinder": 34, "cone": 140}
KINDS = ["plane", "sphere", "cylinder", "cone"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm-ms", type=float, default=40.0,
                    help="before the W warm-up steps, repeat the step for about this many ms so that the GPU has settled at "
                         "its clocks (a 20-step timed region lasts 4 ms, less than the clock ramp); 0 disables it")
    ap.add_argument("--points", type=int, default=None, help="cloud size (default: the workload's)")
    ap.add_argument("--workload", choices=["cfg2", "cfg3", "cfg5"], default="cfg3",
                    help="cfg3 = the config the metric is quoted on (default); cfg2 = 1M points, 6 primitives; "
                         "cfg5 = 50M points with cones")
    ap.add_argument("--shard", choices=["candidates", "points"], default="candidates",
                    help="N > 1 partitioning: candidates (each rank scores its 4096 of the N x 4096 batch on a replica "
                         "of subset 1; default, BASELINE north_star) or points (each rank scores all N x 4096 candidates "
                         "on its 1/N slice of subset 1; the all-reduce is a true sum)")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="N = 1: batches in flight (rh_set_option batches_in_flight): each batch's launches start while the previous "
                         "batch's launch drains; 1 = one batch at a time")
    ap.add_argument("--lists", choices=["auto", "on", "off"], default="auto",
                    help="rh_set_option st_cull on the scored cloud: super-tile candidate lists in front of the score kernel by the "
                         "library's own rule (auto), always, never.  Profiling passes that run one batch at a time pass what the default run took")
    ap.add_argument("--spread-regions", type=int, default=5,
                    help="after the timed region, repeat the same K-step region this many times and report min / median / max of the "
                         "metric as `value_spread` (the headline `value` stays the first region)")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the 50M-point / cones leg (a child process at N = 1)")
    ap.add_argument("--no-cfg2", action="store_true", help="skip the 1M-point / 6-primitive leg (a child process at N = 1)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--oracle-check", type=int, default=0,
                    help="compare the GPU counts of this many candidates (evenly spread over the batch, so every kind is "
                         "in) with the oracle (OpenMP over candidates) and fail on a mismatch; independent of --no-cpu")
    ap.add_argument("--e2e-runs", type=int, default=5, help="timed rh_ransac runs of the end-to-end leg (median reported)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end ransac leg")
    ap.add_argument("--no-f32", action="store_true", help="skip the Float32-cloud leg")
    ap.add_argument("--no-per-kind", action="store_true", help="skip the diagnostic launches of the score kernel over one kind at a time (profiling passes: "
                    "they carry the product launch's kernel name)")
    ap.add_argument("--e2e-iters", type=int, default=16384, help="itermax of the end-to-end ransac leg")
    ap.add_argument("--e2e-cpu-iters", type=int, default=768, help="iterations of the oracle's end-to-end prefix")
    ap.add_argument("--e2e-octree-iters", type=int, default=256, help="itermax of the octree-sampling end-to-end leg")
    ap.add_argument("--collective", choices=["torch", "lib"], default="torch",
                    help="N > 1: who issues the score all-reduce -- torch.distributed (RCCL through PyTorch, dist.ShardedScorer) or the "
                         "library itself (rh_comm_* of the C ABI: librccl directly, the path a Julia / C host uses)")
    ap.add_argument("--no-e2e-octree", action="store_true", help="skip the octree-sampling end-to-end leg")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the cpu_baseline legs of the score step (the end-to-end prefix check stays)")
    ap.add_argument("--detail-out", default=DETAIL_DEFAULT,
                    help="where rank 0 writes the full result document (notes, per-kind diagnostics, nested cfg2/cfg5 legs, counter "
                         "provenance); stdout carries only the compact line (< 4 KB)")
    return ap.parse_args()


def _newest_profile(fname):
    """profiles/rN/<fname> of the highest round number N (numeric, so r10 > r2), or None"""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*", fname)):
        m = re.search(r"profiles/r(\d+)/", f.replace(os.sep, "/"))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


def _lib_hash():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_rh_build", os.path.join(ROOT, "ransac.jl_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b.source_hash()


def pmc_replay(kernel_key, enabled, suffix=""):
    """Per-launch hardware counters of one kernel REPLAYED from the committed PMC passes of the newest round
    (profiles/rN/pmc_hbm_traffic.json, pmc_sq_counters.json: separate rocprofv3 --pmc runs of this same command;
    tools/profile_round.sh).  They are not measured in this run -- the result says so (`replayed_from`) and
    carries `stale: true` when the library has been rebuilt from other sources since the passes were taken
    (profiles/rN/pmc_meta.json holds the source hash).  gfx950's FETCH_SIZE counts half the bytes of a
    coalesced stream (MI355X_MICROARCH.md; calibrated on the refit scan, profiles/r1/README.md): HBM bytes =
    (2 x FETCH_SIZE + WRITE_SIZE) KB."""
    out = {"traffic": None, "sq": {}, "replayed_from": None, "stale": None}
    if not enabled:
        return out
    import re
    pat = re.compile(kernel_key)          # a regular expression over the kernel's demangled name

    def hit(name):
        return pat.search(name) is not None
    # (suffix: "" = the default workload's passes, "_cfg5" = the passes taken with --workload cfg5)
    ft, fs, fm = _newest_profile("pmc_hbm_traffic%s.json" % suffix), _newest_profile("pmc_sq_counters%s.json" % suffix), _newest_profile("pmc_meta.json")
    files = []
    if ft:
        rows = json.load(open(ft))
        rd = [r["mean_KB"] for r in rows if r["counter"] == "FETCH_SIZE" and hit(r["kernel"])]
        wr = [r["mean_KB"] for r in rows if r["counter"] == "WRITE_SIZE" and hit(r["kernel"])]
        if rd and wr:
            out["traffic"] = (2.0 * rd[0] + wr[0]) * 1024.0
            files.append(os.path.relpath(ft, ROOT))
    if fs:
        for r in json.load(open(fs)):
            if hit(r["kernel"]):
                out["sq"][r["counter"]] = r["mean"]
        if out["sq"]:
            files.append(os.path.relpath(fs, ROOT))
    if files:
        out["replayed_from"] = files
        try:
            out["stale"] = (json.load(open(fm)).get("lib_source_hash") != _lib_hash()) if fm and \
                os.path.dirname(fm) == os.path.dirname(os.path.join(ROOT, files[0])) else True
        except Exception:
            out["stale"] = True
    return out


DETAIL_DEFAULT = os.path.join(ROOT, "bench_detail.json")
LINE_MAX_BYTES = 4096        # the driver keeps a bounded tail of stdout: the line it parses must fit with room to spare
LINE_MAX_STR = 120


def _num(x, digits=6):
    """a float rounded to `digits` significant digits (None stays None, ints stay ints)"""
    if x is None or isinstance(x, (bool, int)):
        return x
    try:
        return float("%.*g" % (digits, float(x)))
    except (TypeError, ValueError):
        return None


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_line(out, detail_file=None):
    """The ONE line rank 0 prints: the contract's keys, `roofline` and `cpu_baseline`, and flat scalars for the
    legs (no prose, no nesting beyond those three objects).  Everything else stays in the detail file.  Pure:
    tests/test_bench_line.py builds it from a canned result."""
    rf, cb, cfg = out.get("roofline") or {}, out.get("cpu_baseline") or {}, out.get("config") or {}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _num(line["value"], 7), _num(line["ms_per_step"], 6)
    line["config"] = {k: cfg.get(k) for k in ("workload", "points", "subset_points", "candidates_per_step", "score_mode", "parallelism")}
    line["roofline"] = {"kernel": rf.get("kernel"), "bound": rf.get("bound"), "achieved": _num(rf.get("achieved")),
                        "peak": _num(rf.get("peak")), "unit": rf.get("unit"), "frac": _num(rf.get("frac"), 4),
                        "frac_upper": _num(rf.get("frac_upper"), 4), "frac_guide": _num(rf.get("frac_guide"), 4),
                        "frac_necessary": _num(rf.get("frac_necessary"), 4), "frac_necessary_guide": _num(rf.get("frac_necessary_guide"), 4),
                        "traffic": _num(rf.get("traffic")), "ms_per_launch": _num(rf.get("ms_per_launch")),
                        "frac_over_step_in_flight": _num(rf.get("frac_over_step_in_flight"), 4),
                        "frac_necessary_over_step_in_flight": _num(rf.get("frac_necessary_over_step_in_flight"), 4),
                        "counters": rf.get("counters")}
    line["cpu_baseline"] = ({"value": _num(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                             "sample": cb.get("sample")} if cb else None)
    line["oracle_checked"] = out.get("oracle_checked")
    vs = out.get("value_spread") or {}
    if vs:
        line["value_spread"] = [_num(vs.get("min"), 5), _num(vs.get("median"), 5), _num(vs.get("max"), 5)]
    line["batches_in_flight"] = cfg.get("batches_in_flight")
    line["score_lists"] = cfg.get("score_lists")
    line["one_batch_in_flight_ms"] = _get(out, "one_batch_in_flight", "ms_per_step")
    flat = {
        "rccl_ranks_seen": out.get("rccl_ranks_seen"),
        "cpu_mt_value": _get(out, "cpu_baseline_mt", "value"), "cpu_mt_cores": _get(out, "cpu_baseline_mt", "cores"),
        "masks_ms": _get(out, "masks_out", "ms_per_step"),
        "pcie_ms": _get(out, "pcie_inclusive", "ms_per_step"),
        "f32_ms": _get(out, "float32", "ms_per_step"),
        "refit_scan_ms": _get(out, "roofline_refit", "ms_per_launch"), "refit_scan_gbs": _get(out, "roofline_refit", "achieved"),
        "refit_scan_frac_hbm": _get(out, "roofline_refit", "frac"), "refit_culled_ms": _get(out, "refit_culled", "ms_per_refit_scan"),
        "e2e_shapes": _get(out, "end_to_end", "shapes"), "e2e_seconds": _get(out, "end_to_end", "seconds"),
        "e2e_seconds_to_last_extraction": _get(out, "end_to_end", "seconds_to_last_extraction"),
        "e2e_shapes_per_sec": _get(out, "end_to_end", "shapes_per_sec_to_last_extraction"),
        "e2e_cpu_sets_per_sec": _get(out, "end_to_end", "cpu_baseline", "minimal_sets_per_sec"),
        "e2e_sets_per_sec": _get(out, "end_to_end", "minimal_sets_per_sec"),
        "octree_seconds": _get(out, "end_to_end_octree", "seconds"), "octree_seconds_max": _get(out, "end_to_end_octree", "seconds_max"),
        "octree_shapes": _get(out, "end_to_end_octree", "shapes"),
        "e2e_sharded_seconds": _get(out, "end_to_end_sharded", "seconds"), "e2e_sharded_speedup": _get(out, "end_to_end_sharded", "speedup_vs_one_gpu"),
        "cloud_create_ms": _get(out, "cloud_create", "ms_total"),
        "cfg2_value": _get(out, "cfg2", "value"), "cfg2_ms": _get(out, "cfg2", "ms_per_step"),
        "cfg2_oracle_checked": _get(out, "cfg2", "oracle_checked"),
        "cfg2_frac": _get(out, "cfg2", "roofline", "frac"), "cfg2_frac_necessary": _get(out, "cfg2", "roofline", "frac_necessary"),
        "cfg5_frac_necessary": _get(out, "cfg5", "roofline", "frac_necessary"),
        "cfg5_value": _get(out, "cfg5", "value"), "cfg5_ms": _get(out, "cfg5", "ms_per_step"),
        "cfg5_frac": _get(out, "cfg5", "roofline", "frac"), "cfg5_frac_upper": _get(out, "cfg5", "roofline", "frac_upper"),
        "cfg5_masks_ms": _get(out, "cfg5", "masks_out", "ms_per_step"),
        "cfg5_refit_scan_ms": _get(out, "cfg5", "roofline_refit", "ms_per_launch"),
        "cfg5_refit_scan_frac_hbm": _get(out, "cfg5", "roofline_refit", "frac"),
        "cfg5_oracle_checked": _get(out, "cfg5", "oracle_checked"),
        "cfg5_e2e_shapes": _get(out, "cfg5", "end_to_end", "shapes"), "cfg5_e2e_seconds": _get(out, "cfg5", "end_to_end", "seconds"),
        "cfg5_cloud_create_ms": _get(out, "cfg5", "cloud_create", "ms_total"),
    }
    for k, v in flat.items():
        if v is not None:
            line[k] = _num(v)
    errs = [k for k in ("masks_out", "float32", "cfg2", "cfg5", "end_to_end_sharded", "hbm_measured") if _get(out, k, "error")]
    if errs:
        line["legs_failed"] = errs
    if detail_file:
        line["detail"] = os.path.basename(detail_file)
    def clip(o):
        if isinstance(o, dict):
            return {k: clip(v) for k, v in o.items()}
        return o[:LINE_MAX_STR] if isinstance(o, str) else o
    text = json.dumps(clip(line), separators=(",", ":"))
    if len(text) >= LINE_MAX_BYTES:
        raise RuntimeError("bench line is %d bytes (limit %d)" % (len(text), LINE_MAX_BYTES))
    return text


def write_detail(out, path):
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    os.replace(tmp, path)


def self_launch(args):
    """`python bench.py --gpus N` without torchrun: start the N ranks here.  The parent has not imported torch or
    touched HIP (and never does); the children are fresh interpreters with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, exactly what `torch.distributed.run --nproc-per-node N` would give them.  Rank 0's stdout is
    relayed; any rank failing fails the run -- there is no fallback to fewer ranks."""
    import socket
    import subprocess
    import tempfile
    if not os.environ.get("RH_BENCH_SHARE_GPU0"):
        # count the GPUs from the kernel driver's topology (sysfs): no torch, no HIP, no HSA in this process -- the
        # launcher must never hold the devices its ranks are about to open.  (Every rank checks its own device again.)
        have = 0
        topo = "/sys/class/kfd/kfd/topology/nodes"
        try:
            for node in os.listdir(topo):
                with open(os.path.join(topo, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    have += 1
        except OSError:
            have = 0   # no kfd driver in this system: no GPU
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s); one rank per GPU, no fallback to fewer ranks "
                             "(RH_BENCH_SHARE_GPU0=1 RH_BENCH_BACKEND=gloo rehearses the N > 1 flow on one GPU)"
                             % (args.gpus, have))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RH_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    # wait for all; the first rank that fails takes the others down (they would wait for it in the rendezvous forever)
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for i, q in enumerate(procs):
            if codes[i] is None:
                codes[i] = q.poll()
        if any(c not in (None, 0) for c in codes):
            for i, q in enumerate(procs):
                if codes[i] is None:
                    q.kill()            # exactly the processes started above
                    codes[i] = q.wait()
            break
        time.sleep(0.1)
    if any(codes):
        raise SystemExit("bench.py --gpus %d: rank exit codes %s -- no result (it never falls back to fewer ranks)"
                         % (args.gpus, codes))
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()


def shapes_to_c(R, L, cands):
    arr = (L.Shape * max(1, len(cands)))()
    kmap = {"plane": L.PLANE, "sphere": L.SPHERE, "cylinder": L.CYLINDER, "cone": L.CONE}
    for i, (name, outw, v) in enumerate(cands):
        arr[i].kind = kmap[name]
        arr[i].outwards = int(outw)
        for j, x in enumerate(v):
            arr[i].v[j] = float(x)
        R.lib().rh_shape_finalize(C.byref(arr[i]))
    return arr


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:       # plain start: be the launcher (before anything here touches the GPU)
            return self_launch(args)
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: launch as `python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node %d --master-addr 127.0.0.1 bench.py --gpus %d ...`, or plainly as `python bench.py "
                         "--gpus %d` (it then starts its own ranks)" % (args.gpus, os.environ["WORLD_SIZE"], args.gpus,
                                                                         args.gpus, args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist

    import ransac_jl_amd as R
    from ransac_jl_amd import _lib as L
    from ransac_jl_amd import dist as rdist
    from ransac_jl_amd import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if os.environ.get("RH_BENCH_SHARE_GPU0"):   # rehearsal of the N > 1 flow on a one-GPU box (use with gloo)
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs GPU %d but this node shows %d GPU(s); one rank per GPU, no fallback "
                         "(RH_BENCH_SHARE_GPU0=1 RH_BENCH_BACKEND=gloo rehearses the N > 1 flow on one GPU)"
                         % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    t0 = time.perf_counter()
    torch.zeros(1, device="cuda")       # the process's first HIP call creates the context (60-300 ms on these boxes): before anything is timed
    torch.cuda.synchronize()
    t_hip_context = time.perf_counter() - t0
    # RH_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, pipelined scorer, all-reduce) with
    # one rank -- the RCCL rehearsal a one-GPU box allows
    multi = world > 1 or bool(os.environ.get("RH_BENCH_FORCE_DIST"))
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        backend = os.environ.get("RH_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    lib = R.lib()
    ranks_seen = 1
    if multi:   # every rank adds one: the collective really spans `world` processes
        one = torch.ones(1, dtype=torch.int32, device="cuda")
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(one.item())
        if ranks_seen != world:
            raise SystemExit("bench.py: the all-reduce saw %d ranks, expected %d" % (ranks_seen, world))

    # ---- workload: BASELINE cfg3 (cfg4 when sharded) -----------------------------------
    prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12
    types = [R.FittedPlane, R.FittedSphere, R.FittedCylinder]
    wseed, wname = 3, "cfg3: 10M-point 40-primitive cloud, 30% outliers, r=32 subsets, B=4096 candidates/GPU/step"
    n_default = 10_000_000
    outlier_frac = 0.30
    if args.workload == "cfg2":
        prim = ["plane", "plane", "sphere", "sphere", "cylinder", "cylinder"]
        wseed, wname = 2, "cfg2: 1M-point 6-primitive cloud, no outliers, r=32 subsets, B=4096 candidates/GPU/step"
        n_default, outlier_frac = 1_000_000, 0.0
    if args.workload == "cfg5":
        prim = prim + ["cone"] * 8
        types = types + [R.FittedCone]
        wseed, wname = 5, "cfg5: 50M-point 48-primitive scanner-style cloud with cones, 30% outliers, r=32 subsets, B=4096 candidates/GPU/step"
        n_default = 50_000_000
    n = args.points or n_default
    t0 = time.time()
    xyz, nrm, truth = synth.make_cloud(n, prim, outlier_frac, seed=wseed,
                                       scanner=[synth.BOX / 2] * 3 if args.workload == "cfg5" else None)
    subs = synth.make_subsets(n, 32, seed=wseed)
    S = subs[0].size
    t_scene = time.time() - t0
    pc = R.RANSACCloud(xyz, nrm, subs, device=local_rank)
    t_setup = time.time() - t0
    cms = (C.c_double * 4)()
    L.check(R.lib().rh_cloud_create_ms(pc._h, cms))
    cloud_create = {"ms_total": cms[0], "ms_subset_order": cms[1], "ms_before_it": cms[2], "ms_after_it": cms[3],
                    "scene_generation_s": t_scene, "hip_context_ms_before": 1e3 * t_hip_context,
                    "note": "rh_cloud_create alone, first cloud of the process, HIP context already there (setup_seconds also "
                            "holds numpy's scene generation): total wall time, of which the k-d leaf order of subset 1 (on the "
                            "device: one radix sort per level, kdorder.hip), the "
                            "part before it (allocations, uploads, AoS -> SoA, bounding box, Morton order of the cloud) and after it"}
    params = R.ransacparameters(types)
    cp = R.params_to_c(params, score_mode=L.SCORE_F64)   # Int64 score wraps at this size

    b_global = B_PER_GPU * world
    cands = synth.jittered_candidates(truth, b_global, seed=0)
    if os.environ.get("RH_BENCH_SORT_CANDS"):   # experiment: the copies of one primitive next to each other
        cands = [cands[i] for i in sorted(range(b_global), key=lambda i: i % len(truth))]
    arr = shapes_to_c(R, L, cands)
    batch = rdist.DeviceBatch(pc, arr, b_global)
    counts = torch.zeros(b_global, dtype=torch.int32, device="cuda")
    # N > 1: the cloud shares torch's stream, so fill -> score -> all-reduce are ordered on the device
    # and a step has no host synchronisation (RH_BENCH_SPLIT_STREAMS=1: the library's own stream + two waits)
    same_stream = multi and not os.environ.get("RH_BENCH_SPLIT_STREAMS")
    if same_stream:
        # an explicit (non-default) stream: the legacy null stream would serialise against the
        # collective's stream and undo the overlap
        compute_stream = torch.cuda.Stream()
        torch.cuda.set_stream(compute_stream)
        pc.set_stream(compute_stream.cuda_stream)
    local = rdist.gpu_local_score(pc, batch, cp, wait=not same_stream)
    lo, hi = rdist.shard_bounds(b_global, rank, world)
    points_mode = multi and args.shard == "points"
    tcloud = pc
    if points_mode:   # a second cloud: just this rank's slice of subset 1; pc stays for the rank-0 diagnostics
        sx, sn, ssub, _ = rdist.point_shard_subset(xyz, nrm, subs[0], None, rank, world)
        pcs = R.RANSACCloud(sx, sn, [ssub], device=local_rank)
        sbatch = rdist.DeviceBatch(pcs, arr, b_global)
        if same_stream:
            pcs.set_stream(compute_stream.cuda_stream)
        local = rdist.gpu_local_score(pcs, sbatch, cp, wait=not same_stream)
        tcloud = pcs

    # N > 1: two batches in flight -- batch i's all-reduce overlaps batch i + 1's score launch
    # (RH_BENCH_NO_OVERLAP=1: one batch at a time)
    scorer = None
    if multi and not os.environ.get("RH_BENCH_NO_OVERLAP"):
        scorer = rdist.ShardedScorer(b_global, rank, world, local, "cuda", same_stream=same_stream, points=points_mode)

    # --collective lib: the all-reduce inside the C ABI (rh_score_batch_allreduce_dev), two count buffers in flight
    libcomm = None
    if multi and args.collective == "lib" and not points_mode:
        libcomm = rdist.LibComm(pc, rank, world)
        scorer = None
        lib_bufs = [torch.zeros(b_global, dtype=torch.int32, device="cuda") for _ in range(2)]
        torch.cuda.synchronize()
        lib_k = [0]

    # N = 1: --in-flight F batches in flight, F count buffers in turn (batches_in_flight, include/ransac_hip.h)
    in_flight = 1 if multi else max(1, min(4, args.in_flight))
    ring = [counts] + [torch.zeros_like(counts) for _ in range(in_flight - 1)]
    ring_k = [0]
    if in_flight > 1:
        R.set_option("batches_in_flight", in_flight, cloud=pc)
    if args.lists != "auto":
        R.set_option("st_cull", 1 if args.lists == "on" else 2, cloud=pc)

    def step():
        if libcomm is not None:
            buf = lib_bufs[lib_k[0] & 1]
            lib_k[0] += 1
            libcomm.score_allreduce(batch.slice_ptr(lo), hi - lo, lo, b_global, cp, buf.data_ptr())
        elif scorer is not None:
            scorer.submit()
        elif points_mode:
            rdist.score_batch_point_sharded(lambda out: local(0, b_global, out), counts)
        elif multi:
            rdist.score_batch_sharded(b_global, rank, world, local, counts, same_stream=same_stream)
        else:   # no collective, no host sync inside the timed region
            buf = ring[ring_k[0] % in_flight]
            ring_k[0] += 1
            L.check(lib.rh_score_batch_dev(pc._h, batch.slice_ptr(0), b_global, C.byref(cp),
                                           C.c_void_p(buf.data_ptr()), None))

    def fence():
        if libcomm is not None:
            libcomm.sync()      # every batch's collective has finished before the clock stops
        if scorer is not None:
            scorer.drain()      # every batch's collective has been waited for before the clock stops
        L.check(lib.rh_cloud_sync(pc._h))
        L.check(lib.rh_cloud_sync(tcloud._h))
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    if args.prewarm_ms > 0:   # untimed, before the warm-up steps: bring the clocks up (reported as config.prewarm_ms)
        for _ in range(int(args.prewarm_ms * 14)):   # ~0.07 ms per step at N = 1 (more at N > 1); a fixed count, so every rank does the same
            step()
        fence()
    for _ in range(args.warmup):
        step()
    fence()
    ring_k[0] = 0     # (a fence ends a run of batches in flight: the next call is slot 0 again)
    t0 = time.perf_counter()
    L.check(lib.rh_timer_start(tcloud._h))
    for _ in range(args.steps):
        step()
    ev_ms = C.c_float()
    L.check(lib.rh_timer_stop(tcloud._h, C.byref(ev_ms)))
    fence()
    dt = time.perf_counter() - t0
    dt_rank = dt                      # this rank's own clock (the line quotes the maximum over the ranks)
    if multi:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = 1e3 * dt / args.steps
    value = b_global * args.steps / dt
    # the same K-step region a few more times (each with its own fence on both sides): the spread of `value` on this box.
    # `value` itself is the FIRST region above, exactly as the contract times it.
    spread_vals = []
    for _ in range(max(0, args.spread_regions)):
        fence()
        ring_k[0] = 0
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt1 = time.perf_counter() - t1
        if multi:
            tm1 = torch.tensor([dt1], dtype=torch.float64, device="cuda")
            dist.all_reduce(tm1, op=dist.ReduceOp.MAX)
            dt1 = float(tm1.item())
        spread_vals.append(b_global * args.steps / dt1)
    # what the step's score launch looked like: rows of R chunks, with or without super-tile lists (the library's rule, or --lists)
    linfo = (C.c_int32 * 4)()
    L.check(lib.rh_score_launch_info(pc._h, linfo))
    launch_R, lists_used = int(linfo[0]), bool(linfo[1])
    one_in_flight = None
    if in_flight > 1:   # every buffer of the ring holds the same batch's counts; then the same region one batch at a time
        for r in ring[1:]:
            if args.steps >= in_flight and not torch.equal(r, counts):
                raise SystemExit("PARITY FAILURE: count buffers of the batches in flight differ")
        R.set_option("batches_in_flight", 1, cloud=pc)
        vals = []
        for _ in range(3):
            fence()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                L.check(lib.rh_score_batch_dev(pc._h, batch.slice_ptr(0), b_global, C.byref(cp), C.c_void_p(counts.data_ptr()), None))
            fence()
            vals.append(1e3 * (time.perf_counter() - t1) / args.steps)
        one_in_flight = {"ms_per_step": sorted(vals)[1], "value": b_global / (1e-3 * sorted(vals)[1]),
                         "note": "the same K-step region with batches_in_flight = 1 (median of 3): one batch's prepare + score launches at a time"}
        R.set_option("batches_in_flight", in_flight, cloud=pc)
    counts_h = (lib_bufs[(lib_k[0] - 1) & 1] if libcomm is not None else
                scorer.result(scorer.k - 1) if scorer is not None else counts).cpu().numpy()

    out = {
        "metric": "candidates_scored_per_sec", "value": value, "unit": "candidates/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": wname if n == n_default else "custom (%s primitives)" % args.workload,
                   "points": n, "subset_points": int(S), "candidates_per_step": b_global,
                   "kinds": "%s (cycled over the %d ground-truth primitives, 1%% jitter)"
                            % ("/".join(sorted(set(prim), key=KINDS.index)), len(prim)),
                   "score_mode": "f64", "prewarm_ms": args.prewarm_ms, "batches_in_flight": in_flight,
                   "score_rows": launch_R, "score_lists": lists_used,
                   "parallelism": ("point-sharded x%d (1/%d of subset 1 per GPU, every GPU scores the whole batch), "
                                   "int32 sum all-reduce" % (world, world)) if points_mode
                   else "candidate-sharded x%d, int32 sum all-reduce" % world},
        "tests_per_sec": value * S,
        "one_batch_in_flight": one_in_flight,
        "value_spread": ({"regions": len(spread_vals), "min": min(spread_vals), "median": sorted(spread_vals)[len(spread_vals) // 2],
                          "max": max(spread_vals), "note": "%d further timed regions of %d steps each, same fences; `value` is the first region"
                                                           % (len(spread_vals), args.steps)} if spread_vals else None),
        "rccl_ranks_seen": ranks_seen,
        "collective_backend": (os.environ.get("RH_BENCH_BACKEND", "nccl") if multi else None),
        "collective_issued_by": (("libransac_hip (rh_score_batch_allreduce_dev: librccl directly)" if libcomm is not None else
                                 "torch.distributed (dist.ShardedScorer)") if multi else None),
    }

    # ---- N > 1: what each rank saw, so that the first run on a real multi-GPU node can be read -- per rank the timed region's
    # step, the score launch alone, the all-reduce alone (both with a host wait each: latencies, not the pipelined step)
    per_rank = None
    if multi:
        try:
            kd = max(5, min(20, args.steps))
            fence()
            t1 = time.perf_counter()
            for _ in range(kd):
                if hi > lo:
                    L.check(lib.rh_score_batch_dev(tcloud._h, (sbatch if points_mode else batch).slice_ptr(0 if points_mode else lo),
                                                   (b_global if points_mode else hi - lo), C.byref(cp),
                                                   C.c_void_p(counts.data_ptr() + (0 if points_mode else 4 * lo)), None))
            L.check(lib.rh_cloud_sync(tcloud._h))
            torch.cuda.synchronize()
            t_score_alone = (time.perf_counter() - t1) / kd
            dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(kd):
                dist.all_reduce(counts, op=dist.ReduceOp.SUM)
                torch.cuda.synchronize()
            t_ar_alone = (time.perf_counter() - t1) / kd
            mine = torch.tensor([float(rank), 1e3 * dt_rank / args.steps, 1e3 * t_score_alone, 1e3 * t_ar_alone], dtype=torch.float64, device="cuda")
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = [{"rank": int(v[0].item()), "ms_per_step_timed_region": float(v[1].item()), "ms_score_launch_alone": float(v[2].item()),
                         "ms_allreduce_alone": float(v[3].item())} for v in allr]
            fence()
        except Exception as e:   # diagnostics never take the headline down
            per_rank = [{"error": repr(e)[:300]}]
            fence()
    out["per_rank"] = per_rank

    # ---- N > 1, end to end: ONE scene run by all ranks together (rh_ransac_mp: the minimal sets of every iteration are
    # dealt round-robin to the ranks, windows' candidate lists exchanged through host shared memory, extractions
    # replicated).  Strong scaling: the same scene, the same result as one GPU, bit for bit.
    e2e_sharded = None
    if world > 1 and not args.no_e2e:
        try:
            import hashlib
            rp = R.ransacparameters(types, iteration={"minsubsetN": 4096, "itermax": args.e2e_iters, "τ": 900, "prob_det": 0.9})
            rcp = R.params_to_c(rp, score_mode=L.SCORE_F64, sphere_uses_enabled=True, sampling_streams=1)

            def digest(got_, st_):
                h_ = hashlib.sha256()
                for g_ in got_:
                    h_.update(bytes(g_.c_shape)); h_.update(np.ascontiguousarray(g_.inpoints).tobytes())
                h_.update(np.int64([len(got_), st_["iterations"], st_["candidates_scored"], st_["draws"]]).tobytes())
                return int.from_bytes(h_.digest()[:7], "little")
            pc.enable_all()
            g1, _, s1 = R.ransac(pc, rcp, seed=1234, return_stats=True)     # one GPU, this rank alone: warm-up and the reference result
            d1 = digest(g1, s1)
            del g1
            grp = R.MpGroup("/rh_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("RH_BENCH_TAG", "e2e")), rank, world)
            times = []
            for _ in range(max(1, args.e2e_runs)):
                pc.enable_all()
                fence()
                t0 = time.perf_counter()
                gm, _, sm = R.ransac(pc, rcp, seed=1234, return_stats=True, mp=grp)
                times.append(time.perf_counter() - t0)
            dm = digest(gm, sm)
            nshapes = len(gm)
            split = torch.tensor([float(rank), times[-1], sm["seconds"], sm["seconds_score"], sm["seconds_extract"], sm["seconds_host"]],
                                 dtype=torch.float64, device="cuda")
            splits = [torch.zeros_like(split) for _ in range(world)]
            dist.all_gather(splits, split)
            del gm
            grp.close()
            tt = torch.tensor(times, dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)                        # per run: the slowest rank
            same = torch.tensor([1 if dm == d1 else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            if int(same.item()) != 1:
                raise SystemExit("PARITY FAILURE: rh_ransac_mp on %d ranks differs from the single-GPU rh_ransac" % world)
            tmed = float(torch.sort(tt).values[len(times) // 2].item())
            pc.enable_all()
            t0 = time.perf_counter()
            g1, _, s1 = R.ransac(pc, rcp, seed=1234, return_stats=True)
            t_one = time.perf_counter() - t0
            del g1
            e2e_sharded = {"metric": "shapes_per_sec", "value": nshapes / tmed, "n_gpus": world, "shapes": nshapes, "seconds": tmed,
                           "runs": len(times), "seconds_all_runs_slowest_rank": [float(x) for x in tt.tolist()],
                           "seconds_one_gpu_same_scene": t_one, "speedup_vs_one_gpu": t_one / tmed, "scaling": "strong",
                           "iterations": sm["iterations"], "candidates_scored": sm["candidates_scored"],
                           "per_rank_last_run": [{"rank": int(v[0].item()), "wall_s": float(v[1].item()), "rh_ransac_s": float(v[2].item()),
                                                  "windows_score_s": float(v[3].item()), "extractions_s": float(v[4].item()),
                                                  "sample_fit_and_exchange_s": float(v[5].item())} for v in splits],
                           "note": "ONE scene, all ranks together (rh_ransac_mp): every rank returned the single-GPU result bit for "
                                   "bit (checked in this run); median over runs of the slowest rank's wall time"}
            pc.enable_all()
            fence()
        except Exception as e:   # a wrong result raises SystemExit above; anything else must not take the headline down
            e2e_sharded = {"error": repr(e)[:400]}
            fence()


    if rank == 0:
        # ---- per-kind kernel time (HIP events) and rooflines -----------------------------
        per_kind = {}
        reps = 30
        acc = [0.0] * 5
        list_ms_acc = 0.0
        msk = (C.c_float * 5)()
        for _ in range(reps):   # [4]: the launch of the timed steps (all kinds, one kernel); [0..3]: one launch per kind
            if args.no_per_kind:
                msk[0] = -1.0   # (the library's sign for "the product launch only")
            L.check(lib.rh_score_batch_dev_timed(pc._h, batch.slice_ptr(lo), hi - lo, C.byref(cp),
                                                 C.c_void_p(counts.data_ptr() + 4 * lo), None, msk))
            for k in range(5):
                acc[k] += msk[k]
            lm = C.c_float()
            L.check(lib.rh_last_list_launch_ms(pc._h, C.byref(lm)))
            list_ms_acc += lm.value
        for ki, k in enumerate(KINDS):
            nk = sum(1 for c in cands[lo:hi] if c[0] == k)
            if nk:
                per_kind[k] = {"candidates": nk, "ms_separate_launch": acc[ki] / reps,
                               "note": "diagnostic: the same kernel launched over this kind's candidates alone"}
        # the culled kernel (score4.hip: all kinds in one launch) from 8192 subset points on, else the brute-force kernel per kind
        forced_path = R.get_option("score_path")            # rh_set_option; None / 0: the library's own choice
        culled = forced_path in (None, 0, 2) and S >= 8192
        if culled:
            kname = "score4_kernel<%d%s> (plane+sphere+cylinder%s in one launch%s)" % (
                launch_R, ", lists" if lists_used else "", "+cone" if "cone" in per_kind else "",
                "; st_cull_kernel in front of it: list_launch_ms" if lists_used else "")
            sec = acc[4] / reps * 1e-3
            kinds_in = list(per_kind)
            # counts only, Float64: the timed step's launch (rows of 4, 8, 12 or 16 chunks, picked by the grid's size)
            pmc_key = r"score4_kernel<\d+, false, false, false, %s>" % ("true" if lists_used else "false")
        else:   # per-kind launches: the dominant one
            dom = max(per_kind, key=lambda k: per_kind[k]["ms_separate_launch"])
            kname = "score kernel <%s>" % dom
            sec = per_kind[dom]["ms_separate_launch"] * 1e-3
            kinds_in = [dom]
            pmc_key = r"score_kernel<%d," % KINDS.index(dom)
        ncand = sum(per_kind[k]["candidates"] for k in kinds_in)
        tests = ncand * S
        # the committed PMC passes were taken on the default workloads and batch split
        pmc_ok = n == n_default and world == 1 and forced_path in (None, 0)
        pmc = pmc_replay(pmc_key, pmc_ok, "" if args.workload == "cfg3" else "_" + args.workload)
        sq = pmc["sq"]
        # ---- work-based roofline (round 5).  The batched score is bound by vector-instruction ISSUE, not by HBM (every
        # point is reused across the batch).  What a launch issues is ACCOUNTED, not assumed: tools/isa_account.py multiplies the
        # static opcode histogram of every region of the kernel's gfx950 ISA with how often the region runs (the diag build's event
        # counters of this very workload, tools/s4_stats.py), prices every opcode -- MI355X_MICROARCH.md's rates (frac_guide) and
        # the rates measured on this GPU (frac; frac_upper - frac = the classes priced by analogy) -- and is held against the
        # hardware's per-class instruction counters of the same launch (isa_account_<wl>.json: `validation`).  frac_necessary =
        # the floor any kernel with this culling granularity has to issue (one box test per (candidate, group) + the classifier's
        # own per-point mix over the 64 points of every pair whose group really holds a band point) over the same denominator.
        # The priced cycles per launch are REPLAYED from the committed accounting of the newest round, the launch time is live.
        simd_cycles_per_s = 1024 * 2.4e9
        acc_file = _newest_profile("isa_account_%s.json" % args.workload) if pmc_ok else None
        acc = json.load(open(acc_file)) if acc_file else None
        if acc is not None and bool(acc.get("lists")) != lists_used:   # the committed accounting is of the other instantiation
            acc, acc_file = None, None
        acc_stale = None
        if acc_file:
            fm = os.path.join(os.path.dirname(acc_file), "pmc_meta.json")
            try:
                acc_stale = json.load(open(fm)).get("lib_source_hash") != _lib_hash()
            except Exception:
                acc_stale = True
        alg_bytes = tests * SCORE_BYTES_PER_TEST + ncand * (64 + 4)
        flops = sum(FLOPS_PER_TEST[k] * per_kind[k]["candidates"] * S for k in kinds_in)
        rf = {"kernel": kname, "bound": "valu_issue", "achieved": None, "peak": simd_cycles_per_s, "unit": "SIMD vector-issue cycles/s",
              "frac": None, "frac_upper": None, "frac_guide": None, "frac_necessary": None, "frac_necessary_guide": None}
        if acc:
            scl = acc.get("scale_to_hardware_valu", 1.0)
            m = acc["model"]
            den = sec * simd_cycles_per_s
            rf.update({
                "achieved": scl * m["valu_issue_cycles_measured_lower"] / sec,
                "frac": scl * m["valu_issue_cycles_measured_lower"] / den, "frac_upper": scl * m["valu_issue_cycles_measured"] / den,
                "frac_guide": scl * m["valu_issue_cycles_guide"] / den,
                "frac_necessary": acc["necessary_cycles_measured"] / den, "frac_necessary_guide": acc["necessary_cycles_guide"] / den,
                "frac_scalar_issue": m["salu_issue_cycles"] / den,
                "priced_directly_share": m["priced_directly_share"], "model_vs_hardware_valu": 1.0 / scl,
                "valu_insts_per_launch": scl * m["valu_instructions"], "salu_insts_per_launch": m["salu_instructions"],
                "accounting": os.path.relpath(acc_file, ROOT), "accounting_is_stale": acc_stale})
            if in_flight > 1 and not multi:
                # the same priced cycles over the timed region's time per batch: launches of consecutive batches overlap there
                # (batches_in_flight), so one batch costs less wall time than a launch on its own -- the chip's issue
                # utilisation over the whole step, prepare launch and launch gaps included in the time (not in the work)
                rf["frac_over_step_in_flight"] = scl * m["valu_issue_cycles_measured_lower"] / (ms_per_step * 1e-3 * simd_cycles_per_s)
                rf["frac_over_step_in_flight_upper"] = scl * m["valu_issue_cycles_measured"] / (ms_per_step * 1e-3 * simd_cycles_per_s)
                rf["frac_necessary_over_step_in_flight"] = acc["necessary_cycles_measured"] / (ms_per_step * 1e-3 * simd_cycles_per_s)
        src = acc_file or (os.path.join(ROOT, pmc["replayed_from"][0]) if pmc["replayed_from"] else None)
        rf.update({
            "sq_wave_cycles": sq.get("SQ_WAVE_CYCLES"), "sq_wait_any": sq.get("SQ_WAIT_ANY"),
            "traffic": pmc["traffic"],
            "counters": None if src is None else ("replayed:" + os.path.dirname(os.path.relpath(src, ROOT)) + (" (stale)" if (acc_stale if acc_file else pmc["stale"]) else "")),
            "traffic_frac_of_hbm_peak": None if pmc["traffic"] is None else pmc["traffic"] / sec / 1e9 / HBM_PEAK_GBS,
            "ms_per_launch": sec * 1e3, "ms_source": "HIP events on the library's stream, this run",
            "list_launch_ms": (list_ms_acc / reps) if culled and lists_used else None,
            "replayed_from": pmc["replayed_from"], "replay_is_stale": pmc["stale"],
            "effective_algorithmic": {
                "GBs": alg_bytes / sec / 1e9, "bytes_per_launch": alg_bytes, "tests_per_launch": tests,
                "TFLOPs_reference_flops": flops / sec / 1e12,
                "note": "NOT a roofline: 48.25 B (SURVEY.md 8d) x every (candidate, point) pair of the batch / time.  The "
                        "kernel never streams those bytes -- tiles are staged once and box tests on the k-d leaves reject "
                        "~90 % of the (candidate, group) pairs, bit-exactly -- so this exceeds the HBM peak by design"},
            "note": "frac = priced vector-issue cycles of the launch / (time x 1024 SIMDs x 2.4 GHz), every opcode priced with the issue "
                    "cost measured on this GPU (tools/ubench: binary32 fma 2.45, add / mul 2.3; min / max / compare / select / most integer "
                    "4.2; transcendental 8.15; binary64 4.2-4.75 / 16.2), the few classes priced by analogy at 2.3 (frac) .. 4.2 "
                    "(frac_upper); frac_guide = the same with MI355X_MICROARCH.md's rates (32-bit 2, binary64 4, transcendental 8 / 16); "
                    "frac_necessary = the necessary-work floor (box test per (candidate, group) + the classifier over the pairs that "
                    "hold a band point) over the same denominator.  Instruction counts: tools/isa_account.py (static ISA histogram x "
                    "event counters, validated against the SQ_INSTS_* class counters: model_vs_hardware_valu); launch time: this run"})
        out["roofline"] = rf
        # the host-buffer form of the same step (rh_score_batch: H2D of the shapes, D2H of the counts, one sync)
        hcounts = np.zeros(hi - lo, dtype=np.int32)
        harr = (L.Shape * (hi - lo)).from_buffer_copy(bytes(arr)[C.sizeof(L.Shape) * lo:C.sizeof(L.Shape) * hi])
        for _ in range(2):
            L.check(lib.rh_score_batch(pc._h, harr, hi - lo, C.byref(cp), hcounts.ctypes.data_as(C.POINTER(C.c_int32)), None))
        t0 = time.perf_counter()
        for _ in range(10):
            L.check(lib.rh_score_batch(pc._h, harr, hi - lo, C.byref(cp), hcounts.ctypes.data_as(C.POINTER(C.c_int32)), None))
        t_host = (time.perf_counter() - t0) / 10
        out["pcie_inclusive"] = {"ms_per_step": 1e3 * t_host, "candidates_per_sec": (hi - lo) / t_host,
                                 "note": "rh_score_batch with host buffers (never `value`)"}
        if not np.array_equal(hcounts, counts_h[lo:hi]):
            raise SystemExit("PARITY FAILURE: rh_score_batch and rh_score_batch_dev disagree")
        # the same step with the inlier masks written (scorecandidate's contract is (score, inpoints), plane.jl:61-71):
        # MASK = true instantiation (exact test on every surviving group, no pair queue) + the un-permutation of the
        # k-d-leaf-ordered masks back to subset order; masks stay in HBM (b x ceil(S/64) words)
        try:
            swords = (S + 63) // 64
            nb_m = hi - lo
            # (F batches in flight here too: batch k's un-permutation, HBM-bound, under batch k + 1's score launch, issue-bound)
            dmasks = [torch.empty(nb_m * swords, dtype=torch.int64, device="cuda") for _ in range(in_flight)]
            mcs = [torch.zeros(nb_m, dtype=torch.int32, device="cuda") for _ in range(in_flight)]
            torch.cuda.synchronize()
            dmask, mcounts = dmasks[0], mcs[0]
            mreps = 60

            def mregion(F):
                for k in range(mreps):
                    L.check(lib.rh_score_batch_dev(pc._h, batch.slice_ptr(lo), nb_m, C.byref(cp),
                                                   C.c_void_p(mcs[k % F].data_ptr()), C.c_void_p(dmasks[k % F].data_ptr())))
            mregion(in_flight)       # warm-up: every slot's lists allocated
            L.check(lib.rh_cloud_sync(pc._h))
            mev = C.c_float()

            def mtimed(F):
                R.set_option("batches_in_flight", F, cloud=pc)
                L.check(lib.rh_timer_start(pc._h))
                mregion(F)
                L.check(lib.rh_timer_stop(pc._h, C.byref(mev)))
                return mev.value * 1e-3 / mreps
            # (the two forms in turn, twice: the faster region of each -- a 50M-point mask step moves 2.4 GB and the clocks move with it)
            t_f, t_1 = [], []
            for _ in range(2):
                t_f.append(mtimed(in_flight))
                if in_flight > 1:
                    t_1.append(mtimed(1))
            R.set_option("batches_in_flight", in_flight if in_flight > 1 else None, cloud=pc)
            if in_flight > 1:   # (the count / mask buffers checked below: written by the last pipelined region's order again)
                mregion(in_flight)
                L.check(lib.rh_cloud_sync(pc._h))
            t_m = min(t_f)
            t_m1 = min(t_1) if t_1 else t_m
            for mc_k, dm_k in zip(mcs, dmasks):
                mh = mc_k.cpu().numpy()
                if not np.array_equal(mh, counts_h[lo:hi]):
                    raise SystemExit("PARITY FAILURE: counts of the mask-writing launch differ from the counts-only launch")
                # popcount of the masks == the counts (checksum of checksums), on a slice to bound the host work
                mslice = dm_k[: 64 * swords].cpu().numpy().view(np.uint64).reshape(64, swords)
                if not np.array_equal(np.bitwise_count(mslice).sum(axis=1, dtype=np.int64), mh[:64]):
                    raise SystemExit("PARITY FAILURE: mask popcounts differ from the counts")
                if not torch.equal(dm_k, dmasks[0]):
                    raise SystemExit("PARITY FAILURE: mask buffers of the batches in flight differ")
            out["masks_out"] = {"metric": "candidates_scored_per_sec_with_masks", "value": nb_m / t_m, "unit": "candidates/s",
                                "ms_per_step": 1e3 * t_m, "ms_per_step_one_in_flight": 1e3 * t_m1, "batches_in_flight": in_flight,
                                "mask_bytes_per_step": nb_m * swords * 8,
                                "note": "rh_score_batch_dev with d_masks: counts AND the per-candidate inlier bit masks over "
                                        "subset 1 in subset order, resident in HBM; HIP events over %d steps, the faster of two regions "
                                        "per form (batches in flight / one at a time, in turn); counts equal the headline launch's, "
                                        "popcounts of 64 mask rows equal their counts" % mreps}
            del dmasks, mcs
            del dmask, mcounts
        except SystemExit:
            raise
        except Exception as e:   # never fatal for the headline
            out["masks_out"] = {"error": repr(e)[:300]}
        out["per_kind"] = per_kind
        out["score_path"] = {None: "groups (culled)", 0: "groups (culled)", 1: "brute", 2: "groups (culled)"}[forced_path]
        out["library_variant"] = L.variant()
        out["event_ms_per_step"] = ev_ms.value / args.steps

        # refit: the streaming scan (HBM-bound: the roofline object) and the culled scan the library takes at this size
        t = truth[0]
        plane = R.FittedPlane(t["point"], t["normal"]).to_c()

        def time_refit(handle, cshape, path, reps=5):
            """rh_refit `reps` times with the option refit_path = path on that cloud: (scan ms from HIP events, compaction ms,
            host wall s, index list)"""
            check_set = lib.rh_set_option(handle, b"refit_path", L.REFIT_PATH[path])
            L.check(check_set)
            try:
                idx = np.zeros(n, dtype=np.int64)
                nout = C.c_int64()
                L.check(lib.rh_refit(handle, C.byref(cshape), C.byref(cp), idx.ctypes.data_as(C.POINTER(C.c_int64)), n, C.byref(nout)))
                t0 = time.perf_counter()
                scan_ms, comp_ms = [], []
                for _ in range(reps):
                    L.check(lib.rh_refit(handle, C.byref(cshape), C.byref(cp), idx.ctypes.data_as(C.POINTER(C.c_int64)), n, C.byref(nout)))
                    a, b = C.c_float(), C.c_float()
                    L.check(lib.rh_last_refit_ms(handle, C.byref(a), C.byref(b)))
                    scan_ms.append(a.value); comp_ms.append(b.value)
                wall = (time.perf_counter() - t0) / reps
                return sum(scan_ms) / len(scan_ms), sum(comp_ms) / len(comp_ms), wall, idx[:nout.value].copy()
            finally:
                L.check(lib.rh_set_option(handle, b"refit_path", L.OPTION_UNSET))

        scan_ms, comp_ms, t_refit, list_scan = time_refit(pc._h, plane, "scan")
        t_scan = 1e-3 * scan_ms
        rbytes = n * REFIT_BYTES_PER_POINT
        rpmc = pmc_replay(r"refit_mask_kernel<0>", pmc_ok)
        out["roofline_refit"] = {"kernel": "refit_mask_kernel<plane>", "bound": "hbm", "achieved": rbytes / t_scan / 1e9,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rbytes / t_scan / 1e9 / HBM_PEAK_GBS,
                                 "traffic": rpmc["traffic"], "traffic_replayed_from": rpmc["replayed_from"],
                                 "traffic_replay_is_stale": rpmc["stale"], "ms_per_launch": 1e3 * t_scan,
                                 "algorithmic_bytes_per_launch": rbytes,
                                 "compaction_ms": comp_ms, "rh_refit_host_wall_ms": 1e3 * t_refit,
                                 "inliers": int(len(list_scan)),
                                 "note": "the HBM-bound kernel of the path (option refit_path = scan): one pass over the whole cloud "
                                         "in original order (48.125 B/point); HIP events on the library's stream; host wall "
                                         "adds the compaction, two syncs and the D2H of the index list.  Clouds of 2^21 "
                                         "points and more take the culled scan instead: `refit_culled`"}
        cul_ms, cul_comp, cul_wall, list_cul = time_refit(pc._h, plane, "culled")
        if not np.array_equal(list_scan, list_cul):
            raise SystemExit("PARITY FAILURE: the culled refit scan and the streaming scan disagree")
        out["refit_culled"] = {"ms_per_refit_scan": cul_ms, "speedup_vs_streaming_scan": scan_ms / cul_ms,
                               "compaction_ms": cul_comp, "rh_refit_host_wall_ms": 1e3 * cul_wall,
                               "kernels": "refitk_boxes_kernel + refitk_groups_kernel + refitk_flags_kernel (korder.hip)",
                               "note": "the same refit through the Morton-ordered copy of the cloud: box test per 64-point "
                                       "group, exact test on the surviving groups only (a few percent of the cloud), index "
                                       "list identical to the streaming scan's (checked in this run)"}

        # ---- the same workload as a Float32 cloud (RANSACCloud(...; force_eltype = Float32), octree.jl:102-109): every
        # per-point operation in binary32, the refit scan streams 24 bytes per point instead of 48
        if world == 1 and not args.no_f32:
            try:
                from oracle import oracle as orc
                t0 = time.time()
                x32, n32 = xyz.astype(np.float32), nrm.astype(np.float32)
                pc32 = R.RANSACCloud(x32, n32, subs, device=local_rank, force_eltype=np.float32)
                arr32 = (L.Shape * b_global).from_buffer_copy(bytes(arr))
                for i in range(b_global):
                    lib.rh_shape_finalize_f32(C.byref(arr32[i]))
                batch32 = rdist.DeviceBatch(pc32, arr32, b_global)
                # (the same batches in flight as the headline step: F count buffers in turn)
                ring32 = [torch.zeros(b_global, dtype=torch.int32, device="cuda") for _ in range(in_flight)]
                torch.cuda.synchronize()
                c32 = ring32[0]
                if in_flight > 1:
                    R.set_option("batches_in_flight", in_flight, cloud=pc32)
                k32 = [0]

                def step32():
                    buf = ring32[k32[0] % in_flight]
                    k32[0] += 1
                    L.check(lib.rh_score_batch_dev(pc32._h, batch32.slice_ptr(0), b_global, C.byref(cp), C.c_void_p(buf.data_ptr()), None))
                for _ in range(100):
                    step32()
                L.check(lib.rh_cloud_sync(pc32._h))
                k32[0] = 0
                L.check(lib.rh_timer_start(pc32._h))
                for _ in range(100):
                    step32()
                ev32 = C.c_float()
                L.check(lib.rh_timer_stop(pc32._h, C.byref(ev32)))
                for r32 in ring32[1:]:
                    if not torch.equal(r32, c32):
                        raise SystemExit("PARITY FAILURE: Float32 count buffers of the batches in flight differ")
                c32h = c32.cpu().numpy()
                # oracle (binary32 twin) on a spread sample of the batch
                sel = list(range(0, b_global, max(1, b_global // 256)))[:256]
                oc32 = orc.Cloud32(x32, n32, subs[0])
                o32 = (orc.Shape * len(sel))()
                for j, i in enumerate(sel):
                    o32[j] = orc.Shape.from_buffer_copy(bytes(arr32[i]))
                chk = oc32.score_batch(o32, orc.Params.from_buffer_copy(bytes(cp)), nthreads=16)
                if not np.array_equal(chk, c32h[sel]):
                    raise SystemExit("PARITY FAILURE: Float32 cloud counts differ from the binary32 oracle")
                cs = R.shape_f32(R.FittedPlane(truth[0]["point"], truth[0]["normal"]))
                scan32_ms, _c, _w, list32 = time_refit(pc32._h, cs, "scan")
                cul32_ms, _c, _w, list32c = time_refit(pc32._h, cs, "culled")
                oref = oc32.refit(orc.Shape.from_buffer_copy(bytes(cs)), orc.Params.from_buffer_copy(bytes(cp)))
                if not (np.array_equal(oref, list32) and np.array_equal(oref, list32c)):
                    raise SystemExit("PARITY FAILURE: Float32 refit list differs from the binary32 oracle")
                n32o = C.c_int64(len(list32))
                t_scan32 = 1e-3 * scan32_ms
                b32 = n * 24.125
                out["float32"] = {
                    "candidates_per_sec": b_global / (ev32.value * 1e-3 / 100), "ms_per_step": ev32.value / 100,
                    "oracle_checked": len(sel),
                    "roofline_refit": {"kernel": "refit32_mask_kernel<plane>", "bound": "hbm", "achieved": b32 / t_scan32 / 1e9,
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b32 / t_scan32 / 1e9 / HBM_PEAK_GBS,
                                       "ms_per_launch": 1e3 * t_scan32, "algorithmic_bytes_per_launch": b32, "inliers": int(n32o.value),
                                       "traffic": None},
                    "refit_culled_ms": cul32_ms,
                    "setup_seconds": time.time() - t0,
                    "note": "the same cloud and batch as Float32 (shapes rounded to binary32): culled score kernel with the exact "
                            "test in binary32, refit scan over 24.125 B per point; 256 counts and the refit list checked against the "
                            "oracle's binary32 twin (oracle/orc_f32.c) in this run"}
                batch32.free()
                del pc32, oc32
            except SystemExit:
                raise
            except Exception as e:
                out["float32"] = {"error": repr(e)[:300]}

        # ---- what this box's HBM delivers to simple streaming kernels (torch ops as the probe) ------
        try:
            nbytes = 1 << 30
            a_ = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").normal_()
            b_ = torch.empty_like(a_)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def _time(fn, reps=10):
                fn(); torch.cuda.synchronize()
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record(); torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e-3 / reps
            t_copy = _time(lambda: b_.copy_(a_))
            t_read = _time(lambda: a_.sum())
            out["hbm_measured"] = {"copy_GBs": 2 * nbytes / t_copy / 1e9, "read_GBs": nbytes / t_read / 1e9, "bytes": nbytes,
                                   "note": "1-GiB float32 tensor on this box: b.copy_(a) counts read + write bytes, a.sum() "
                                           "read bytes; the refit scan's achieved GB/s can be read against these as well as "
                                           "against the 8 TB/s spec"}
            del a_, b_
        except Exception as e:   # a probe, never fatal
            out["hbm_measured"] = {"error": repr(e)[:200]}

        # ---- cpu_baseline: the oracle (port of the reference's single-threaded path) ------
        if world > 1:
            args.no_cpu = True   # the CPU legs belong to the N = 1 line (the other ranks would sit in the barrier for them)
        if not args.no_cpu and not args.no_cpu_baseline:
            from oracle import oracle as orc
            oc = orc.Cloud(xyz, nrm, subs[0])
            nb = 768
            oarr = (orc.Shape * nb)()
            C.memmove(oarr, arr, C.sizeof(L.Shape) * nb)
            op = orc.Params.from_buffer_copy(bytes(cp))
            t0 = time.perf_counter()
            oc_counts = oc.score_batch(oarr, op)
            t_cpu = time.perf_counter() - t0
            if not np.array_equal(oc_counts, counts_h[:nb]):
                raise SystemExit("PARITY FAILURE: GPU counts differ from the oracle on the cpu_baseline sample")
            out["cpu_baseline"] = {"value": nb / t_cpu, "unit": "candidates/s", "cores": 1, "kind": "port",
                                   "sample": "first %d candidates of the batch, S=%d, %.1f s, counts equal the GPU's" % (nb, S, t_cpu),
                                   "host_cpus": os.cpu_count()}
            # steelman: the same passes spread over this GPU's share of the host cores (OpenMP over candidates).  A GPU box
            # hands one GPU's job 16 of the host's CPUs; the affinity mask says what this process may really use.
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            nthr = max(1, min(16, avail))
            nb_mt = min(b_global, 256 * nthr)   # 16 threads: the whole 4096-candidate batch is checked against the oracle
            oarr_mt = (orc.Shape * nb_mt)()
            C.memmove(oarr_mt, arr, C.sizeof(L.Shape) * nb_mt)
            t0 = time.perf_counter()
            mt_counts = oc.score_batch_mt(oarr_mt, op, nthr)
            t_mt = time.perf_counter() - t0
            if not np.array_equal(mt_counts, counts_h[:nb_mt]):
                raise SystemExit("PARITY FAILURE: GPU counts differ from the oracle on the multi-thread sample")
            out["cpu_baseline_mt"] = {"value": nb_mt / t_mt, "unit": "candidates/s", "cores": nthr, "kind": "port",
                                      "host_cpus": os.cpu_count(), "cpus_available_to_this_process": avail,
                                      "sample": "first %d candidates of the same batch, OpenMP over candidates on %d threads "
                                                "(one GPU's share of the host, capped at 16; the reference itself is "
                                                "single-threaded), %.1f s; counts checked equal to the GPU's"
                                                % (nb_mt, nthr, t_mt)}
            out["oracle_checked"] = int(max(nb, nb_mt))
            del oc

        if args.oracle_check > 0 and world == 1:
            from oracle import oracle as orc
            k_chk = min(args.oracle_check, b_global)
            sel = sorted({int(round(i * (b_global - 1) / max(1, k_chk - 1))) for i in range(k_chk)})
            oc = orc.Cloud(xyz, nrm, subs[0])
            oarr_c = (orc.Shape * len(sel))()
            for j, i in enumerate(sel):
                oarr_c[j] = orc.Shape.from_buffer_copy(bytes(arr[i]))
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            t0 = time.perf_counter()
            chk = oc.score_batch_mt(oarr_c, orc.Params.from_buffer_copy(bytes(cp)), max(1, min(16, avail)))
            t_chk = time.perf_counter() - t0
            if not np.array_equal(chk, counts_h[sel]):
                bad = [sel[j] for j in range(len(sel)) if chk[j] != counts_h[sel[j]]]
                raise SystemExit("PARITY FAILURE: GPU counts differ from the oracle on candidates %s" % bad[:10])
            kinds_chk = sorted({cands[i][0] for i in sel}, key=KINDS.index)
            out["oracle_checked"] = max(int(out.get("oracle_checked", 0)), len(sel))
            out["oracle_check"] = {"candidates": len(sel), "kinds": kinds_chk, "seconds": t_chk,
                                   "inliers": int(chk.sum()), "note": "counts of %d candidates spread evenly over the batch, "
                                   "oracle (OpenMP) vs the timed launch's: identical" % len(sel)}
            del oc

        # ---- end to end: shapes / s of the whole ransac() loop on the same cloud ----------
        if not args.no_e2e:
            e2e = R.ransacparameters(types,
                                     iteration={"minsubsetN": 4096, "itermax": args.e2e_iters, "τ": 900, "prob_det": 0.9})
            ecp = R.params_to_c(e2e, score_mode=L.SCORE_F64, sphere_uses_enabled=True, sampling_streams=1)
            pc.enable_all()
            R.ransac(pc, ecp, seed=99)                      # warm-up: a whole run (the cloud's one-time allocations:
            pc.enable_all()                                 # select list, compact records, windows, pinned arena)

            def prewarm():   # the oracle legs above left the GPU idle for seconds: settle the clocks again
                for _ in range(int(args.prewarm_ms * 5)):
                    L.check(lib.rh_score_batch_dev(pc._h, batch.slice_ptr(lo), hi - lo, C.byref(cp),
                                                   C.c_void_p(counts.data_ptr() + 4 * lo), None))
                L.check(lib.rh_cloud_sync(pc._h))
            runs = []
            for _ in range(max(1, args.e2e_runs)):   # the same run (same seed, same result) timed several times: median
                pc.enable_all()
                prewarm()
                t0 = time.perf_counter()
                got, secs, st = R.ransac(pc, ecp, seed=1234, return_stats=True)
                runs.append((time.perf_counter() - t0, st))
            order = sorted(range(len(runs)), key=lambda i: runs[i][0])
            t_e2e, st = runs[order[len(order) // 2]]
            last_it = max([g.iteration for g in got], default=0)
            out["end_to_end"] = {"metric": "shapes_per_sec", "value": len(got) / t_e2e, "shapes": len(got), "seconds": t_e2e,
                                 "runs": len(runs), "seconds_all_runs": [r[0] for r in runs],
                                 "seconds_rh_ransac": st["seconds"], "iterations": st["iterations"], "minimal_sets": st["iterations"] * 4096,
                                 "minimal_sets_per_sec": st["iterations"] * 4096 / t_e2e,
                                 "candidates_scored": st["candidates_scored"], "last_extraction_iteration": last_it,
                                 "seconds_to_last_extraction": st.get("seconds_to_last_extraction"),
                                 "shapes_per_sec_to_last_extraction": (len(got) / st["seconds_to_last_extraction"]) if st.get("seconds_to_last_extraction") else None,
                                 "breakdown_s": {"sample_fit": st["seconds_host"], "score": st["seconds_score"],
                                                 "extract": st["seconds_extract"]},
                                 "largest_shapes": sorted((len(g.inpoints) for g in got), reverse=True)[:5],
                                 "note": "MEDIAN of %d rh_ransac calls (same seed; value = shapes / median wall time through the "
                                         "Python wrapper): minsubsetN=4096, itermax=%d, root-cell sampling like the "
                                         "reference, f64 score mode, per-set random streams (sampling + fits + scoring on "
                                         "the device, iterations speculated in pipelined windows of up to 512; after a warm-up "
                                         "run of the same length)" % (len(runs), args.e2e_iters)}
            # fixed behaviour: level-weighted octree sampling (docs/src/ransac.md:73-96)
            ocp = R.params_to_c(e2e, score_mode=L.SCORE_F64, sphere_uses_enabled=True, sampling_streams=1,
                                octree_sampling=True)
            if not args.no_e2e_octree:
                pc.enable_all()
                ocp.itermax = 4
                R.ransac(pc, ocp, seed=99)                      # builds + caches the linear octree (setup)
                pc.enable_all()
                ocp.itermax = args.e2e_octree_iters
                R.ransac(pc, ocp, seed=99)                      # warm-up: a whole run of the same length, like the root-cell leg's
                oruns = []
                for _ in range(max(1, args.e2e_runs)):
                    pc.enable_all()
                    prewarm()
                    t0 = time.perf_counter()
                    goto_, _, sto = R.ransac(pc, ocp, seed=1234, return_stats=True)
                    oruns.append((time.perf_counter() - t0, sto))
                t_oct, sto = sorted(oruns, key=lambda r: r[0])[len(oruns) // 2]
                out["end_to_end_octree"] = {
                    "metric": "shapes_per_sec", "value": len(goto_) / t_oct, "shapes": len(goto_), "seconds": t_oct, "seconds_rh_ransac": sto["seconds"],
                    "runs": len(oruns), "seconds_all_runs": [r[0] for r in oruns],
                    "iterations": sto["iterations"], "candidates_scored": sto["candidates_scored"],
                    "last_extraction_iteration": max([g.iteration for g in goto_], default=0),
                    "seconds_to_last_extraction": sto.get("seconds_to_last_extraction"),
                    "breakdown_s": {"sample_fit": sto["seconds_host"], "score": sto["seconds_score"], "extract": sto["seconds_extract"]},
                    "note": "median of %d runs after a warm-up run of the same length; same cloud, octree_sampling=1 (level-weighted cells of a linear Morton octree; not what the "
                            "reference executes, SURVEY.md 0.5), minsubsetN=4096, itermax=%d.  Every iteration's scores move the level "
                            "distribution the next one samples from, so the iterations run as CHAINED windows: sampling, fits, scoring, "
                            "the level update and the extraction test of up to 8 iterations are queued back to back on the device, the "
                            "host replays iteration i while the device runs i + 1 (and checks the device's level distribution against "
                            "its own, bit for bit); the candidate store (~100k entries at an extraction) is compacted on the device"
                            % (len(oruns), args.e2e_octree_iters)}
            if not args.no_cpu:
                # CPU side of the same loop on a bounded prefix, and a parity check of that prefix
                from oracle import oracle as orc
                nit = args.e2e_cpu_iters
                ecp.itermax = nit
                pc.enable_all()
                gp, _, sp = R.ransac(pc, ecp, seed=1234, return_stats=True)
                oc = orc.Cloud(xyz, nrm, subs[0])
                t0 = time.perf_counter()
                eo = oc.ransac(orc.Params.from_buffer_copy(bytes(ecp)), seed=1234)
                t_cpu = time.perf_counter() - t0
                same = (len(gp) == len(eo["shapes"]) and sp["draws"] == eo["draws"] and
                        all(bytes(g.c_shape) == bytes(e["shape"]) and np.array_equal(g.inpoints, e["inpoints"])
                            for g, e in zip(gp, eo["shapes"])))
                if not same:
                    raise SystemExit("PARITY FAILURE: rh_ransac differs from the oracle on the %d-iteration prefix" % nit)
                out["end_to_end"]["cpu_baseline"] = {
                    "kind": "port", "cores": 1, "iterations": nit, "seconds": t_cpu,
                    "minimal_sets_per_sec": nit * 4096 / t_cpu, "shapes": len(eo["shapes"]),
                    "sample": "the oracle's ransac() on the first %d iterations of the same run; extracted shapes, "
                              "index sets and draw counts checked identical to rh_ransac's" % nit}
        if e2e_sharded is not None:
            out["end_to_end_sharded"] = e2e_sharded
        out["setup_seconds"] = t_setup
        out["cloud_create"] = cloud_create
        # ---- the other single-GPU BASELINE configs on this GPU: child processes of the same script
        def child_leg(name, extra, note):
            import subprocess
            try:
                dpath = os.path.splitext(args.detail_out)[0] + "_%s.json" % name
                if os.path.exists(dpath):
                    os.remove(dpath)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", name, "--no-cfg5", "--no-cfg2", "--no-f32",
                                    "--detail-out", dpath] +
                                   ([] if "--e2e-iters" in extra else ["--no-e2e", "--no-cpu"]) + extra, capture_output=True, text=True, timeout=900)
                if r.returncode != 0:
                    if "PARITY FAILURE" in (r.stderr or ""):   # a wrong result is never just a missing leg
                        raise SystemExit("%s leg: %s" % (name, r.stderr.strip().splitlines()[-1]))
                    return {"error": "exit code %d: %s" % (r.returncode, (r.stderr or "").strip()[-300:])}
                json.loads(r.stdout.strip().splitlines()[-1])      # the child's own compact line parses
                c5 = json.load(open(dpath))                         # its full document
                leg = {"config": c5["config"], "value": c5["value"], "unit": c5["unit"], "ms_per_step": c5["ms_per_step"],
                       "tests_per_sec": c5["tests_per_sec"], "per_kind": c5["per_kind"],
                       "score_kernel_ms": c5["roofline"]["ms_per_launch"],
                       "roofline_refit": {k: c5["roofline_refit"][k] for k in
                                          ("kernel", "bound", "achieved", "peak", "unit", "frac", "ms_per_launch",
                                           "algorithmic_bytes_per_launch", "inliers")},
                       "setup_seconds": c5["setup_seconds"], "note": note}
                for k in ("oracle_checked", "oracle_check", "masks_out", "refit_culled", "cloud_create", "end_to_end", "roofline"):
                    if k in c5:
                        leg[k] = c5[k]
                return leg
            except SystemExit:
                raise
            except Exception as e:   # the headline line must not depend on these legs
                return {"error": repr(e)[:300]}
        if world == 1 and args.workload == "cfg3" and n == n_default and not args.no_cfg2:
            out["cfg2"] = child_leg("cfg2", ["--oracle-check", "4096"],
                                    "python bench.py --workload cfg2 --no-cpu --no-e2e --oracle-check 4096 (child process; the whole timed batch against the oracle): BASELINE "
                                    "configs[1], 1M points = 2 planes + 2 spheres + 2 cylinders without outliers, S = 31 250, B = 4096")
        if world == 1 and args.workload == "cfg3" and n == n_default and not args.no_cfg5:
            out["cfg5"] = child_leg("cfg5", ["--steps", "60", "--warmup", "10", "--oracle-check", "1024", "--e2e-iters", "4096", "--e2e-runs", "5",
                                             "--e2e-cpu-iters", "48", "--no-e2e-octree", "--no-cpu-baseline"],
                                    "python bench.py --workload cfg5 --steps 60 --warmup 10 --oracle-check 1024 --e2e-iters 4096 --e2e-runs 5 "
                                    "--e2e-cpu-iters 48 --no-e2e-octree --no-cpu-baseline (child process): one replica of the 50M-point cloud on "
                                    "this GPU, S = 1 562 500, cones in the batch; 1024 candidates of all four kinds (spread evenly over the timed batch) checked against the "
                                    "oracle; the refit scan streams 2.4 GB; a bounded end-to-end leg (4096 iterations, the oracle's loop "
                                    "compared on a 48-iteration prefix)")
        write_detail(out, args.detail_out)
        sys.stderr.write("bench.py: full result document -> %s\n" % args.detail_out)
        sys.stderr.flush()
        print(compact_line(out, args.detail_out), flush=True)
    batch.free()
    if points_mode:
        sbatch.free()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
