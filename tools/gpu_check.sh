#!/bin/bash
# Run on the GPU box from the repo root (gpurun): GPU tests, the default bench line, a kernel trace.
# usage: tools/gpu_check.sh <tag> [quick]
TAG=${1:-x}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
if [ "$2" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -3 $O/pytest_gpu.log
fi
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-multichain --no-extras > $O/under_rocprof_cfg2.json 2> /tmp/prof_$TAG.err
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-multichain --no-extras" > $O/kernel_stats_cfg2.txt
python3 $R/tools/rocpd_summary.py $DB --gaps >> $O/kernel_stats_cfg2.txt 2>&1
head -6 $O/kernel_stats_cfg2.txt
python3 - <<PY
import json
d=json.load(open("$O/bench_cfg2.json"))
print("value", d["value"], "min/max", d["value_min"], d["value_max"], "ms", d["ms_per_step"])
for k in ("astep_path","tune1","concurrent_chains"):
    if k in d: print(k, {a:b for a,b in d[k].items() if a!="note"})
print("kernels", d.get("roofline_kernels"))
print("roofline", {a:b for a,b in d.get("roofline",{}).items() if a not in("note",)})
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("all_cores",{}).get("value"))
PY
