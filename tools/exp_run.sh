#!/bin/bash
# GPU box: quick headline bench of several library variants + slot anatomies.
# usage: tools/exp_run.sh TAG "variant1 variant2 ..." ["trace_lib1 trace_lib2 ..."] [extra bench args]
TAG=$1; VARS=$2; TRACES=$3; shift 3
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; C=$R/pymc_bart_amd/csrc; V=$R/build/variants
Q="--steps 20 --warmup 5 --repeats 5 --no-extras --no-cpu-baseline --no-multichain $@"
for v in $VARS; do
  L=$V/libpgbart_hip_$v.so; [ $v = default ] && L=$C/libpgbart_hip.so
  PGBART_HIP_LIB=$L timeout 300 python bench.py $Q > $O/bench_$v.json 2> $O/bench_$v.err
  python3 - <<PY
import json
try:
    d=json.load(open("$O/bench_$v.json"))
    k=d.get("roofline_kernels") or {}
    print("%-12s %.4f M  (%.4f..%.4f)  ms %.4f  " % ("$v", d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d["ms_per_step"]),
          {a:(round(b["avg_us"],2)) for a,b in k.items()})
except Exception as e:
    print("$v", "FAILED", e)
PY
done
for t in $TRACES; do
  echo "== trace $t"
  timeout 300 python tools/trace_slot.py --lib $V/libpgbart_hip_$t.so > $O/trace_$t.txt 2>&1; tail -6 $O/trace_$t.txt
done
