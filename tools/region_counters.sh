#!/bin/bash
# Per-region instruction counters of the v4 score kernel on the cfg3 batch (profiles/rN/README.md): the whole batch, the
# skeleton alone (RH_G2_DBG=1: staging + box tests, no pair survives), and the batch restricted to one kind at a time.
# Each line is one rocprofv3 --pmc run of tools/score_only.py (5 launches).   bash tools/region_counters.sh > out.txt
ROOT=$(pwd)
C1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"
C2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
run() {   # label, env assignments...
    label=$1; shift
    for cs in "$C1" "$C2"; do
        echo "== $label"
        bash tools/pmc_score.sh "$cs" "$@" | grep -A9 "score4_kernel"
        grep ms_per gpurun_out/pmc_score/out.txt
    done
}
if [ "${WL:-cfg3}" = "cfg5" ]; then   # WL=cfg5 bash tools/region_counters.sh: the 50M-point workload, per kind incl. cones
    run "cfg5 whole batch" WL=cfg5
    run "cfg5 skeleton only (RH_G2_DBG=1)" WL=cfg5 RH_G2_DBG=1
    run "cfg5 planes only" WL=cfg5 KINDS=plane
    run "cfg5 spheres only" WL=cfg5 KINDS=sphere
    run "cfg5 cylinders only" WL=cfg5 KINDS=cylinder
    run "cfg5 cones only" WL=cfg5 KINDS=cone
    exit 0
fi
run "whole batch" X=1
run "skeleton only (RH_G2_DBG=1)" RH_G2_DBG=1
run "planes only (1648 candidates)" KINDS=plane
run "spheres only (1224)" KINDS=sphere
run "cylinders only (1224)" KINDS=cylinder
