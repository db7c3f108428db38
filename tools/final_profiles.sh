#!/bin/bash
# Final-build evidence of a round (run on the GPU box from the repo root; results under gpurun_out/fin,
# to be copied into profiles/): PMC passes first (bench.py reads the traffic / instruction counts from
# profiles/r03_pmc_<workload>.json), then the default bench line, then rocprofv3 kernel traces.
R=$PWD; O=$R/gpurun_out/fin; mkdir -p $O
for W in cfg2 cfg4 cfg5; do
  tools/pmc_collect.sh $W > $O/pmc_$W.log 2>&1
  cp $R/gpurun_out/r03_pmc_$W.json $R/profiles/r03_pmc_$W.json
  cp $R/gpurun_out/r03_pmc_$W.json $O/r03_pmc_$W.json
done
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03_bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --chains-per-gpu 2 --no-extras --no-cpu-baseline --no-multichain --no-roofline > $O/r03_bench_cfg2_chains2.json 2>/dev/null
for w in cfg2 cfg4 cfg5; do
  if [ $w = cfg2 ]; then A="--steps 20 --warmup 5 --repeats 5"; else A="--workload $w --steps 10 --warmup 2 --repeats 3"; fi
  A="$A --no-cpu-baseline --no-multichain --no-extras --no-workloads"
  tools/ktrace.sh gpurun_out/fin/r03_${w}_kernel_stats.txt $A
  cp /tmp/ktrace_bench.json $O/r03_bench_${w}_under_rocprof.json
  tools/pmc_busy.sh gpurun_out/fin/r03_${w}_issue_wait.txt $A --no-roofline > /dev/null 2>&1
done
python tools/latency_guard.py > $O/r03_latency_guard.txt 2>&1; echo "latency guard rc=$?" >> $O/r03_latency_guard.txt; tail -3 $O/r03_latency_guard.txt
ls -la $O
