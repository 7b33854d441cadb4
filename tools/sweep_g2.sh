# grid sweep of the merged score kernel on the bench workload: chunks per block x block budget
set -e
mkdir -p gpurun_out
for cpb in 4 6 8 10 12 16; do
  for blocks in 8192 16384 32768; do
    RH_G2_CPB=$cpb RH_G2_BLOCKS=$blocks timeout -k 10 120 python bench.py --no-cpu --no-e2e --no-cfg5 --no-cfg2 > gpurun_out/sw.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1])
print("cpb=$cpb blocks=$blocks ms_per_step=%.4f kernel=%.4f"%(d['ms_per_step'], d['roofline']['ms_per_launch']), flush=True)
PY
  done
done
