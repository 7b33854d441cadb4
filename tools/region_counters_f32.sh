#!/bin/bash
# tools/region_counters.sh's regions for a Float32 cloud (F32=1), next to the Float64 whole batch
C1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"
C2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
run() {
    label=$1; shift
    for cs in "$C1" "$C2"; do
        echo "== $label"
        bash tools/pmc_score.sh "$cs" "$@" | grep -A9 "score4_kernel"
        grep ms_per gpurun_out/pmc_score/out.txt
    done
}
run "f64 whole batch" X=1
run "f32 whole batch" F32=1
run "f32 skeleton only" F32=1 RH_G2_DBG=1
run "f32 planes only" F32=1 KINDS=plane
run "f32 spheres only" F32=1 KINDS=sphere
run "f32 cylinders only" F32=1 KINDS=cylinder
