set -e
for cpb in 2 4 6 8 12 16 32; do
  for blocks in 16384 65536; do
    RH_G2_CPB=$cpb RH_G2_BLOCKS=$blocks timeout -k 10 120 python bench.py --no-cpu --no-e2e --steps 30 --warmup 5 > gpurun_out/sw.json
    python - <<PY
import json
d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1])
print("cpb=$cpb blocks=$blocks ms_per_step=%.4f kernel=%.4f"%(d['ms_per_step'], d['roofline']['ms_per_launch']))
PY
  done
done
