#!/bin/bash
# cfg5: likelihood grid 768 (3 workgroups per CU, the old geometry) against the occupancy-derived grid
mkdir -p gpurun_out/r3v
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],3), {k:(round(v["avg_us"],2), round(v["pct"],1)) for k,v in d.get("roofline_kernels",{}).items()})
PY
}
for rep in 1 2 3; do
  for g in default 768 1280; do
    if [ $g = default ]; then unset PGB_LL_GRID; else export PGB_LL_GRID=$g; fi
    timeout 300 python bench.py --workload cfg5 --steps 10 --warmup 2 --repeats 8 --no-cpu-baseline --no-multichain --no-extras > gpurun_out/r3v/c5_${g}_$rep.json 2>gpurun_out/r3v/c5_${g}_$rep.err
    show gpurun_out/r3v/c5_${g}_$rep.json
  done
done
