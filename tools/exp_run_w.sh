#!/bin/bash
# GPU box: bench a workload against several library variants.  usage: tools/exp_run_w.sh TAG WORKLOAD "v1 v2 ..."
TAG=$1; W=$2; VARS=$3; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; C=$R/pymc_bart_amd/csrc; V=$R/build/variants
for v in $VARS; do
  L=$V/libpgbart_hip_$v.so; [ $v = default ] && L=$C/libpgbart_hip.so
  PGBART_HIP_LIB=$L timeout 600 python bench.py --workload $W --steps 5 --warmup 1 --burnin 10 --repeats 3 --no-multichain --no-cpu-baseline --no-extras > $O/${W}_$v.json 2> $O/${W}_$v.err
  echo -n "$v  "; python3 tools/show_bench.py $O/${W}_$v.json
done
