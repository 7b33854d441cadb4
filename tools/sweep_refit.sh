for b in 512 1024 2048 3072 4096 8192 16384; do
  RH_LIB_VARIANT=diag RH_REFIT_BLOCKS=$b timeout -k 10 120 python bench.py --no-cpu --no-e2e --no-cfg5 --steps 3 --warmup 1 > gpurun_out/rs.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/rs.json').read().strip().splitlines()[-1])
r=d['roofline_refit']; print("blocks=$b refit ms %.4f GB/s %.0f frac %.3f"%(r['ms_per_launch'], r['achieved'], r['frac']))
PY
done
