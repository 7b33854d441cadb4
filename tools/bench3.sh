#!/bin/bash
# the score step of the three single-GPU workloads, counts + masks + refit only (no CPU legs): one compact line each
set -e
tag=${1:-x}
for w in ${WORKLOADS:-cfg3 cfg2 cfg5}; do
  extra=""
  [ "$w" = cfg5 ] && extra="--steps 60 --warmup 10"
  python bench.py --workload $w --no-cpu --no-e2e --no-cfg2 --no-cfg5 --no-f32 --oracle-check 64 $extra --detail-out gpurun_out/b3_${tag}_$w.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['ms_per_launch'], 'masks_ms', d.get('masks_ms'), 'cloud_create_ms', d.get('cloud_create_ms'), 'checked', d['oracle_checked'])"
done
