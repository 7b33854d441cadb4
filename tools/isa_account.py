#!/usr/bin/env python3
"""Work-based accounting of the culled score kernel (score4_kernel<R, false, false, false>, csrc/score4.hip):

    dynamic instruction histogram  =  sum over the kernel's REGIONS of  (static opcode histogram of the region's ISA)
                                                                      x (how often the region runs in one launch)

The static side is the gfx950 ISA hipcc emits for the product flags (no GPU needed): the kernel is cut into basic blocks,
LLVM's own loop annotations give the loop tree (per kind body: the batch loop with its two point loops and the second-pass /
compaction loop), and every instruction gets a price by OPCODE -- two columns: MI355X_MICROARCH.md's rates and the rates
measured on this GPU (tools/ubench/valu_rates.hip, count_seq.hip).  The dynamic side is the diag build's event counters of
one launch (tools/s4_stats.py -> s4_stats_<WL>.json: chunk visits, batches, second-pass pairs, ring drains per kind).  The
result is checked against the hardware's own per-class instruction counters of the same launch (profiles/rN/pmc_sq_counters*.json)
when they are there: the model has to reproduce SQ_INSTS_VALU / _SALU / the FMA / ADD / MUL / TRANS / F64 class counts.

Also: the NECESSARY-work floor.  Any algorithm with this culling granularity has to (1) test every (candidate, 64-point group)
box once and (2) classify the 64 points of every pair whose group holds a band point (rh_dbg_cls_soundness counts them);
priced with the classifier's own per-point instruction mix (the point loop bodies of this ISA) that is a lower bound of the
issue cycles a launch needs, and  frac_necessary = floor / (launch time x 1024 SIMDs x 2.4 GHz).

    python tools/isa_account.py --stats gpurun_out/s4_stats_cfg3.json --rows 12 [--pmc profiles/r5/pmc_sq_counters.json]
                                [--ms 0.0815] --out profiles/r5/isa_account_cfg3.json
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = ["cone", "cylinder", "sphere", "plane"]          # the order of the four per-kind bodies in the kernel (expensive first)
SIMD_CYCLES_PER_S = 1024 * 2.4e9


# ---------------------------------------------------------------------------------------------------------- prices ----
def classify(op):
    """opcode -> class name (what the two price columns key on)"""
    o = op
    if o.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache", "s_memtime", "s_memrealtime")):
        return "smem"
    if o.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_endpgm", "s_setprio", "s_branch", "s_cbranch", "s_setpc", "s_swappc", "s_getpc")):
        return "s_ctrl"
    if o.startswith("s_"):
        return "salu"
    if o.startswith("ds_"):
        return "lds"
    if o.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if o.startswith("v_mfma") or o.startswith("v_smfma"):
        return "mfma"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_f64", o):
        return "trans_f64"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)(_iflag|_legacy)?_f(32|16)", o):
        return "trans_f32"
    if re.match(r"v_(fma|fmac|mad|mac)_f64", o) or o.startswith("v_div_fmas_f64") or o.startswith("v_div_fixup_f64"):
        return "fma_f64"
    if re.match(r"v_mul_f64", o) or o.startswith("v_ldexp_f64"):
        return "mul_f64"
    if re.match(r"v_(add|sub|min|max|cmp\w*|cmpx\w*|div_scale|trig_preop|frexp\w*|fract|floor|ceil|rndne|trunc)_f64", o) or (o.startswith("v_cmp") and o.endswith("_f64")):
        return "add_f64"
    if o.startswith("v_cvt"):
        return "cvt"
    if re.match(r"v_(fma|fmac|mad|mac|fmaak|fmamk|madak|madmk)_f32", o) or o.startswith("v_pk_fma_f32"):
        return "fma_f32"
    if re.match(r"v_(add|sub|subrev)_f32", o) or o.startswith("v_pk_add_f32"):
        return "add_f32"
    if re.match(r"v_mul(_legacy)?_f32", o) or o.startswith("v_pk_mul_f32"):
        return "mul_f32"
    # everything below: no class counter of its own on the hardware
    if re.match(r"v_(min|max|min3|max3|med3)_(f32|f16|i32|u32|i16|u16)", o):
        return "minmax"
    if o.startswith("v_cmp") or o.startswith("v_cmpx"):
        return "cmp"
    if o.startswith("v_cndmask"):
        return "cndmask"
    if o.startswith(("v_alignbit", "v_alignbyte", "v_perm", "v_bfe", "v_bfi", "v_bcnt", "v_lshl_or", "v_lshl_add", "v_add_lshl", "v_and_or", "v_or3", "v_xad",
                     "v_add3", "v_bitop3", "v_mad_u32", "v_mad_i32", "v_mul_u32_u24", "v_mul_i32_i24", "v_mad_u64", "v_mad_i64", "v_mul_lo", "v_mul_hi",
                     "v_bfrev", "v_ffbh", "v_ffbl", "v_sad", "v_addc", "v_subb", "v_subbrev", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")):
        return "int_slow"
    if o.startswith(("v_mbcnt",)):
        return "mbcnt"
    if o.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane"
    if o.startswith("v_mov") or o.startswith("v_accvgpr") or o.startswith("v_swap"):
        return "mov"
    if re.match(r"v_(lshlrev|lshrrev|ashrrev)_(b32|i32|b16|i16)", o):
        return "int_slow"                       # (shifts hold the port 4.1-4.2 cycles like the other slow integer operations: ubench_count_seq.txt)
    if re.match(r"v_(and|or|xor|not|add|sub|subrev|add_co|sub_co|subrev_co)_(b32|u32|i32|nc_u32|u16|i16|b16)", o) or \
            re.match(r"v_(add|sub|subrev)_(co_)?u32", o) or o.startswith(("v_add_u32", "v_sub_u32", "v_subrev_u32")):
        return "int_fast"
    if o.startswith("v_"):
        return "valu_other"
    return "other"


# cycles a wave64 instruction holds a SIMD's vector-issue port (scalar: the SIMD's scalar issue)
PRICE_GUIDE = {   # /opt/skills/guides/MI355X_MICROARCH.md: 32-bit VALU 2 (several waves per SIMD), FP64 4, transcendental 8 / 16
    "fma_f32": 2, "add_f32": 2, "mul_f32": 2, "minmax": 2, "cmp": 2, "cndmask": 2, "int_slow": 2, "int_fast": 2, "mbcnt": 2, "lane": 2, "mov": 2,
    "valu_other": 2, "cvt": 2, "trans_f32": 8, "fma_f64": 4, "mul_f64": 4, "add_f64": 4, "trans_f64": 16, "mfma": 16}
PRICE_MEASURED = {   # tools/ubench/valu_rates.hip + count_seq.hip on this GPU, 8 waves per SIMD (profiles/r3/ubench_valu_rates.txt, r5/ubench_count_seq.txt)
    "fma_f32": 2.45, "add_f32": 2.3, "mul_f32": 2.3, "minmax": 4.2, "cmp": 4.2, "cndmask": 4.2, "int_slow": 4.2, "int_fast": 2.45, "mbcnt": 4.2, "lane": 4.2,
    "mov": 2.3, "valu_other": 4.2, "cvt": 4.2, "trans_f32": 8.15, "fma_f64": 4.75, "mul_f64": 4.3, "add_f64": 4.2, "trans_f64": 16.2, "mfma": 16}
# classes measured directly on this GPU (round 5 added moves, v_mbcnt, v_readlane / v_readfirstlane, DPP moves, shifts, v_add3, v_and, v_add_u32 and
# v_cvt_f32_f64 to count_seq.hip); `valu_other` -- whatever opcode none of the patterns above knows -- is priced by analogy: the residual error bar
MEASURED_DIRECTLY = {"fma_f32", "add_f32", "mul_f32", "minmax", "cmp", "cndmask", "int_slow", "int_fast", "cvt", "trans_f32", "fma_f64", "mul_f64", "add_f64", "trans_f64",
                     "mov", "mbcnt", "lane"}
VALU_CLASSES = set(PRICE_GUIDE)
SALU_PRICE = 4.2   # a scalar ALU instruction, per SIMD (one scalar unit per CU shared by four SIMDs): valu_rates.hip


# ------------------------------------------------------------------------------------------------------------ ISA ----
def build_isa():
    sys.path.insert(0, os.path.join(ROOT, "ransac.jl_amd"))
    import build
    flags = [f for f in build.FLAGS if f not in ("-fPIC", "-shared", "-Wall")]
    out = os.path.join(tempfile.mkdtemp(prefix="isa_"), "score4.s")
    subprocess.check_call([build.hipcc()] + flags + ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "ransac.jl_amd", "csrc"), "-x", "hip",
                           os.path.join(ROOT, "ransac.jl_amd", "csrc", "score4.hip"), "--cuda-device-only", "-S", "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read()


def kernel_text(isa, rows, mask=False, f32=False, tail=False, lists=False):
    name = "score4_kernelILi%dELb%dELb%dELb%dELb%dE" % (rows, mask, f32, tail, lists)
    m = re.search(r"^(\S*%s\S*):.*$" % re.escape(name), isa, flags=re.M)
    if not m:
        raise SystemExit("kernel %s not found in the ISA" % name)
    end = isa.index(".Lfunc_end", m.end())
    return m.group(1), isa[m.end():end]


class Block:
    def __init__(self, name, note):
        self.name, self.note, self.ins = name, note, []
        self.l1 = self.l2 = None      # headers of the depth-1 / depth-2 loops this block sits in
        self.is_l1_header = self.is_l2_header = False


def parse_blocks(text):
    blocks = [Block("entry", "")]
    lines = text.split("\n")
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", ln) or re.match(r"^; (%bb\.\d+):\s*(;.*)?$", ln)
        if m:
            note = m.group(2) or ""
            if i + 1 < len(lines) and lines[i + 1].strip().startswith(";") and "Loop" in lines[i + 1]:
                note += " " + lines[i + 1].strip()
            b = Block(m.group(1).lstrip("."), note)
            blocks.append(b)
            continue
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if re.match(r"^[a-z_0-9]+$", op):
            blocks[-1].ins.append(op)
    l2_parent = {}
    for b in blocks:
        n = b.note
        m = re.search(r"This Loop Header: Depth=1", n)
        if m:
            b.is_l1_header, b.l1 = True, b.name.replace("L", "")
        m = re.search(r"Parent Loop (BB\d+_\d+) Depth=1", n)
        if m and "Inner Loop Header" in n:
            b.is_l2_header, b.l1, b.l2 = True, m.group(1), b.name.replace("L", "")
            l2_parent[b.l2] = b.l1
    for b in blocks:
        if b.l1 or b.l2:
            continue
        m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d)", b.note)
        if m:
            if m.group(2) == "1":
                b.l1 = m.group(1)
            else:
                b.l2 = m.group(1)
    for b in blocks:   # depth-2 members: their depth-1 parent
        if b.l2 and not b.l1:
            b.l1 = l2_parent.get(b.l2)
    return blocks


def has_f64(b):
    return any(op.endswith("_f64") or "_f64_" in op for op in b.ins)


def regions(blocks, rows):
    """region name -> list of blocks.  Per kind body k (in the kernel's order): pre (everything between the previous body's batch
    loop and this one: staging + the unrolled chunk visits of stage 1), batch (the batch loop outside its inner loops), pl (the
    two point loops), p2 (the third inner loop: second pass, or the cone's compaction rounds), drain_in (its blocks with binary64
    arithmetic: a full ring drained inside the loop), drain_end (binary64 blocks behind it: the final drain), p2_tail (the
    other blocks that only run for batches with an undecided point)."""
    l1_heads = [b.l1 for b in blocks if b.is_l1_header]
    if len(l1_heads) != 4:
        raise SystemExit("expected the four per-kind batch loops, found %d depth-1 loops" % len(l1_heads))
    kind_of = {h: KINDS[i] for i, h in enumerate(l1_heads)}
    reg = defaultdict(list)
    cur = 0            # index of the next body whose batch loop has not started yet
    seen_l1 = None
    for b in blocks:
        if b.l1 is None:
            if seen_l1 is not None and cur < 4 and kind_of.get(seen_l1) == KINDS[cur]:
                cur += 1
                seen_l1 = None
            reg["pre_%s" % KINDS[cur] if cur < 4 else "epilogue"].append(b)
            continue
        seen_l1 = b.l1
        k = kind_of[b.l1]
        l2s = [x.l2 for x in blocks if x.is_l2_header and x.l1 == b.l1]
        if b.l2 is None:
            # behind the third inner loop? (textual order: blocks of the batch loop after the last inner header)
            reg["batch_%s" % k].append(b)
        elif b.l2 in l2s[:2]:
            reg["pl_%s" % k].append(b)
        else:
            reg[("drain_in_%s" if has_f64(b) else "p2_%s") % k].append(b)
    # the kernel's own prologue (the kinds' counts, the row's chunk range): the blocks of the first body's pre region in front
    # of the first load of points (global_load_dwordx2: the staging's binary64 loads)
    first = "pre_%s" % KINDS[0]
    idx = next((i for i, b in enumerate(reg[first]) if any(o.startswith("global_load_dwordx2") for o in b.ins)), 0)
    reg["prologue"] = reg[first][:idx]
    reg[first] = reg[first][idx:]
    # a pre region = the staging (up to its barrier) + the unrolled chunk visits of stage 1: box tests ("test") and, behind
    # the `no survivor` branch, the pair-list bookkeeping ("append": blocks that hold its LDS atomic, lane counts or list stores)
    for k in KINDS:
        blks = reg.pop("pre_%s" % k)
        bar = next((i for i, b in enumerate(blks) if "s_barrier" in b.ins), len(blks) - 1)
        reg["stage_%s" % k] = blks[:bar + 1]
        for b in blks[bar + 1:]:
            app = any(o in ("ds_write_b16", "ds_add_rtn_u32") or o.startswith("v_mbcnt") for o in b.ins)
            reg[("visit_append_%s" if app else "visit_test_%s") % k].append(b)
    # split the batch region: blocks behind the third loop with binary64 arithmetic = the final drain
    for k in KINDS:
        keep = []
        third_seen = False
        order = [b for b in blocks if b.l1 is not None and kind_of[b.l1] == k]
        l2s = [x.l2 for x in order if x.is_l2_header]
        for b in order:
            if b.l2 is not None and len(l2s) >= 3 and b.l2 == l2s[2]:
                third_seen = True
        after_third = False
        for b in reg["batch_%s" % k]:
            idx = order.index(b)
            after_third = any(x.l2 is not None and len(l2s) >= 3 and x.l2 == l2s[2] for x in order[:idx])
            if after_third and has_f64(b):
                reg["drain_end_%s" % k].append(b)
            else:
                keep.append(b)
        reg["batch_%s" % k] = keep
    return reg


def hist(blks):
    c = Counter()
    for b in blks:
        c.update(b.ins)
    return c


# ---------------------------------------------------------------------------------------------------------- model ----
def region_counts(stats, rows):
    """how often each region runs in the launch the counters were taken on (wave-level)"""
    out = {}
    cpw = max(1, (rows + 3) // 4)
    for k in KINDS:
        p = stats["per_kind"][k]
        seg = p["segments"]
        visits = p["visits_h0"] + p["visits_h1"] + p["visits_h2"] + p["visits_h3p"]
        out["stage_%s" % k] = {"count": seg}
        out["visit_test_%s" % k] = {"count": visits / float(cpw)}                                   # per unrolled copy
        out["visit_append_%s" % k] = {"count": (visits - p["visits_without_survivor"]) / float(cpw)}
        out["batch_%s" % k] = {"count": p["batches"]}
        out["pl_%s" % k] = {"count": None, "batches": p["batches"]}              # iterations from the unroll factor
        if k == "cone":
            out["p2_%s" % k] = {"count": p["cone_compaction_rounds"]}
        else:
            out["p2_%s" % k] = {"count": p["second_pass_pairs"]}
        out["drain_in_%s" % k] = {"count": p["ring_drains_full"]}
        out["drain_end_%s" % k] = {"count": p["ring_drains_final"]}
    out["epilogue"] = {"count": stats["global"]["blocks_with_tile"]}
    out["prologue"] = {"count": stats["global"]["blocks_with_tile"]}       # (the counters are per wave)
    return out


def account(isa, stats, rows, lists=False):
    sym, text = kernel_text(isa, rows, lists=lists)
    blocks = parse_blocks(text)
    reg = regions(blocks, rows)
    rc = region_counts(stats, rows)
    dyn = Counter()
    per_region = {}
    point_loop = {}
    for name, blks in sorted(reg.items()):
        h = hist(blks)
        n_ins = sum(h.values())
        info = rc.get(name, {"count": 0})
        cnt = info["count"]
        if name.startswith("pl_"):
            k = name[3:]
            # the two point loops cover 32 points each; a body of U points runs 32 / U times per loop and batch
            per_loop = []
            l2s = sorted({b.l2 for b in blks})
            total = Counter()
            for l2 in l2s:
                hb = hist([b for b in blks if b.l2 == l2])
                nbits = sum(v for o, v in hb.items() if o.startswith("v_alignbit"))
                u = max(1, nbits // (2 if k == "cone" else 1))
                iters = 32.0 / u
                per_loop.append({"loop": l2, "instructions_per_iteration": sum(hb.values()), "points_per_iteration": u})
                for o, v in hb.items():
                    total[o] += v * iters
            point_loop[k] = {"per_batch": dict(total), "loops": per_loop}
            for o, v in total.items():
                dyn[o] += v * info["batches"]
            per_region[name] = {"static_instructions": n_ins, "runs": info["batches"], "dynamic_instructions": sum(total.values()) * info["batches"]}
            continue
        scale = 1.0
        for o, v in h.items():
            dyn[o] += v * cnt * scale
        per_region[name] = {"static_instructions": n_ins, "runs": cnt, "dynamic_instructions": n_ins * cnt * scale}
    return sym, dyn, per_region, point_loop, reg


def price(dyn):
    by_class = Counter()
    for o, v in dyn.items():
        by_class[classify(o)] += v
    valu = {c: v for c, v in by_class.items() if c in VALU_CLASSES}
    cyc_g = sum(v * PRICE_GUIDE[c] for c, v in valu.items())
    cyc_m = sum(v * PRICE_MEASURED[c] for c, v in valu.items())
    n_valu = sum(valu.values())
    direct = sum(v for c, v in valu.items() if c in MEASURED_DIRECTLY)
    # the error bar of the measured column: the classes priced by analogy at 2.3 (lower) / their table value (upper)
    cyc_m_lo = sum(v * (PRICE_MEASURED[c] if c in MEASURED_DIRECTLY else 2.3) for c, v in valu.items())
    return {"by_class": dict(by_class), "valu_instructions": n_valu, "salu_instructions": by_class.get("salu", 0),
            "lds_instructions": by_class.get("lds", 0), "vmem_instructions": by_class.get("vmem", 0), "smem_instructions": by_class.get("smem", 0),
            "valu_issue_cycles_guide": cyc_g, "valu_issue_cycles_measured": cyc_m, "valu_issue_cycles_measured_lower": cyc_m_lo,
            "priced_directly_share": direct / max(1.0, n_valu), "salu_issue_cycles": by_class.get("salu", 0) * SALU_PRICE}


def necessary(stats, point_loop, reg):
    """floor: one box test per (candidate, group) + the classifier over the 64 points of every pair that holds a band point"""
    tot_g = tot_m = 0.0
    detail = {}
    for k in KINDS:
        c = stats["census"][k]
        if not c["pairs"]:
            continue
        pl = Counter(point_loop[k]["per_batch"])          # one batch = 64 pairs x 64 points
        pb = price(pl)
        per_batch_g, per_batch_m = pb["valu_issue_cycles_guide"], pb["valu_issue_cycles_measured"]
        batches = c["pairs_with_band_point"] / 64.0
        # the box test: the arithmetic of box_skip32 alone (fma / mul / add / sub / max + one compare per test), from the pre region
        pre = hist(reg["visit_test_%s" % k] + reg["visit_append_%s" % k])
        arith = Counter({o: v for o, v in pre.items() if classify(o) in ("fma_f32", "add_f32", "mul_f32", "minmax", "cmp")})
        copies = max(1, sum(1 for o in pre.elements() if o == "ds_add_rtn_u32"))      # unrolled chunk visits in the region
        pa = price(arith)
        per_visit_g, per_visit_m = pa["valu_issue_cycles_guide"] / copies, pa["valu_issue_cycles_measured"] / copies   # 64 candidates x 4 groups
        visits = c["pairs"] / 256.0
        g = batches * per_batch_g + visits * per_visit_g
        m = batches * per_batch_m + visits * per_visit_m
        tot_g += g
        tot_m += m
        detail[k] = {"band_pairs": c["pairs_with_band_point"], "pairs": c["pairs"], "cycles_per_batch_guide": per_batch_g, "cycles_per_batch_measured": per_batch_m,
                     "cycles_per_chunk_visit_guide": per_visit_g, "cycles_per_chunk_visit_measured": per_visit_m,
                     "valu_per_point": sum(v for o, v in pl.items() if classify(o) in VALU_CLASSES) / 64.0}
    return tot_g, tot_m, detail


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", required=True)
    ap.add_argument("--rows", type=int, required=True, help="R of the launch: 12 at cfg3, 16 at cfg5, 4 (or what the dispatch picks) at cfg2")
    ap.add_argument("--lists", action="store_true", help="the LIST instantiation (the rows walk their super-tile's candidate list: st_cull)")
    ap.add_argument("--pmc", default=None, help="pmc_sq_counters*.json of the same launch (validation)")
    ap.add_argument("--ms", type=float, default=None, help="launch time in ms (rocprof / HIP events)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    stats = json.load(open(a.stats))
    isa = build_isa()
    sym, dyn, per_region, point_loop, reg = account(isa, stats, a.rows, lists=a.lists)
    pr = price(dyn)
    nec_g, nec_m, nec_detail = necessary(stats, point_loop, reg)
    out = {"kernel": sym, "rows": a.rows, "lists": bool(a.lists), "workload": stats.get("workload"), "regions": per_region, "model": pr, "necessary": nec_detail,
           "necessary_cycles_guide": nec_g, "necessary_cycles_measured": nec_m,
           "top_opcodes": dict(Counter({o: v for o, v in dyn.items()}).most_common(40))}
    hw = {}
    if a.pmc and os.path.exists(a.pmc):
        pat = re.compile(r"score4_kernel<%d, false, false, false, %s>" % (a.rows, "true" if a.lists else "false"))
        for r in json.load(open(a.pmc)):
            if pat.search(r["kernel"]):
                hw[r["counter"]] = r["mean"]
        bc = pr["by_class"]
        cmp_ = {"SQ_INSTS_VALU": pr["valu_instructions"], "SQ_INSTS_SALU": bc.get("salu", 0), "SQ_INSTS_LDS": bc.get("lds", 0),
                "SQ_INSTS_SMEM": bc.get("smem", 0), "SQ_INSTS_VALU_FMA_F32": bc.get("fma_f32", 0), "SQ_INSTS_VALU_ADD_F32": bc.get("add_f32", 0),
                "SQ_INSTS_VALU_MUL_F32": bc.get("mul_f32", 0), "SQ_INSTS_VALU_TRANS_F32": bc.get("trans_f32", 0), "SQ_INSTS_VALU_CVT": bc.get("cvt", 0),
                "SQ_INSTS_VALU_FMA_F64": bc.get("fma_f64", 0), "SQ_INSTS_VALU_MUL_F64": bc.get("mul_f64", 0), "SQ_INSTS_VALU_ADD_F64": bc.get("add_f64", 0),
                "SQ_INSTS_VALU_TRANS_F64": bc.get("trans_f64", 0)}
        out["validation"] = {k: {"model": v, "hardware": hw.get(k), "ratio": (v / hw[k]) if hw.get(k) else None} for k, v in cmp_.items()}
        if hw.get("SQ_INSTS_VALU"):      # the model's MIX scaled to the hardware's total: what the fractions below are quoted on
            sc = hw["SQ_INSTS_VALU"] / max(1.0, pr["valu_instructions"])
            out["scale_to_hardware_valu"] = sc
    if a.ms:
        den = a.ms * 1e-3 * SIMD_CYCLES_PER_S
        sc = out.get("scale_to_hardware_valu", 1.0)
        out["ms_per_launch"] = a.ms
        out["frac_guide"] = sc * pr["valu_issue_cycles_guide"] / den
        out["frac_measured"] = sc * pr["valu_issue_cycles_measured"] / den
        out["frac_measured_lower"] = sc * pr["valu_issue_cycles_measured_lower"] / den
        out["frac_necessary_guide"] = nec_g / den
        out["frac_necessary_measured"] = nec_m / den
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)
    print("kernel", sym)
    print("regions (static instructions x runs):")
    for k, v in per_region.items():
        if v["runs"]:
            print("  %-18s %6d x %10.0f = %12.0f" % (k, v["static_instructions"], v["runs"], v["dynamic_instructions"]))
    bc = pr["by_class"]
    print("model: VALU %.3e  SALU %.3e  LDS %.3e  VMEM %.3e  SMEM %.3e" % (pr["valu_instructions"], bc.get("salu", 0), bc.get("lds", 0), bc.get("vmem", 0), bc.get("smem", 0)))
    print("VALU by class:", {c: "%.3e" % v for c, v in sorted(bc.items(), key=lambda kv: -kv[1]) if c in VALU_CLASSES})
    print("priced directly (class measured on this GPU): %.1f %% of the VALU instructions" % (100 * pr["priced_directly_share"]))
    if "validation" in out:
        for k, v in out["validation"].items():
            if v["hardware"]:
                print("  %-26s model %.3e  hardware %.3e  ratio %.3f" % (k, v["model"], v["hardware"], v["ratio"]))
    for k in ("frac_guide", "frac_measured_lower", "frac_measured", "frac_necessary_guide", "frac_necessary_measured"):
        if k in out:
            print("%-26s %.4f" % (k, out[k]))
    for k, v in nec_detail.items():
        print("  necessary %-8s band pairs %9d of %10d, %.1f VALU per point" % (k, v["band_pairs"], v["pairs"], v["valu_per_point"]))


if __name__ == "__main__":
    main()
