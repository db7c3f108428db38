#!/bin/bash
# Final-build evidence of a round (run on the GPU box from the repo root; results under gpurun_out/fin,
# to be copied into profiles/): PMC passes first (bench.py reads the traffic / instruction counts from
# profiles/<round>_pmc_<workload>.json; ROUND=r06 by default), then the default bench line, then rocprofv3 kernel traces.
R=$PWD; O=$R/gpurun_out/fin; mkdir -p $O; export ROUND=${ROUND:-r06}
for W in cfg2 cfg4 cfg5; do
  tools/pmc_collect.sh $W > $O/pmc_$W.log 2>&1
  cp $R/gpurun_out/${ROUND}_pmc_$W.json $R/profiles/${ROUND}_pmc_$W.json
  cp $R/gpurun_out/${ROUND}_pmc_$W.json $O/${ROUND}_pmc_$W.json
done
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${ROUND}_bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --chains-per-gpu 2 --no-extras --no-cpu-baseline --no-multichain --no-roofline > $O/${ROUND}_bench_cfg2_chains2.json 2>/dev/null
for w in cfg2 cfg4 cfg5; do
  if [ $w = cfg2 ]; then A="--steps 20 --warmup 5 --repeats 5"; else A="--workload $w --steps 10 --warmup 2 --repeats 3"; fi
  A="$A --no-cpu-baseline --no-multichain --no-extras --no-workloads"
  tools/ktrace.sh gpurun_out/fin/${ROUND}_${w}_kernel_stats.txt $A
  cp /tmp/ktrace_bench.json $O/${ROUND}_bench_${w}_under_rocprof.json
  tools/pmc_busy.sh gpurun_out/fin/${ROUND}_${w}_issue_wait.txt $A --no-roofline > /dev/null 2>&1
  tools/pmc_mix.sh gpurun_out/fin/${ROUND}_${w}_instruction_mix.txt $A --no-roofline > /dev/null 2>&1
done
python tools/image_timing.py > $O/${ROUND}_image_timing.json 2> /dev/null
# one rank under the launcher: RCCL initialised, the end-of-run gather (100 draws = 80 MB) through the nccl backend
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-multichain > $O/${ROUND}_bench_rccl_world1.json 2> /dev/null
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) > $O/${ROUND}_final_gpu_tests.txt
python tools/occupancy_guard.py > $O/${ROUND}_occupancy.txt 2>&1
python tools/latency_guard.py > $O/${ROUND}_latency_guard.txt 2>&1; echo "latency guard rc=$?" >> $O/${ROUND}_latency_guard.txt; tail -3 $O/${ROUND}_latency_guard.txt
ls -la $O
