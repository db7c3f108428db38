#!/bin/bash
# Final-build evidence of a round (run on the GPU box from the repo root; results under gpurun_out/fin,
# to be copied into profiles/): PMC passes first (bench.py reads the traffic / instruction counts from
# profiles/r02_pmc_<workload>.json), then the bench lines, then rocprofv3 kernel traces.
R=$PWD; O=$R/gpurun_out/fin; mkdir -p $O
for W in cfg2 cfg4 cfg5; do
  tools/pmc_collect.sh $W > $O/pmc_$W.log 2>&1
  cp $R/gpurun_out/r02_pmc_$W.json $R/profiles/r02_pmc_$W.json
  cp $R/gpurun_out/r02_pmc_$W.json $O/r02_pmc_$W.json
done
python bench.py --steps 20 --warmup 5 > $O/r02_bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --steps 20 --warmup 5 --burnin 0 --no-extras --no-cpu-baseline --no-multichain > $O/r02_bench_cfg2_chain_start.json 2>/dev/null
python bench.py --workload cfg4 --steps 5 --warmup 1 --burnin 10 --repeats 3 --no-multichain --cpu-budget 8 > $O/r02_bench_cfg4.json 2> $O/bench_cfg4.err
python bench.py --workload cfg5 --steps 5 --warmup 1 --burnin 10 --repeats 3 --no-multichain --cpu-budget 8 > $O/r02_bench_cfg5.json 2> $O/bench_cfg5.err
python bench.py --steps 20 --warmup 5 --chains-per-gpu 2 --no-extras --no-cpu-baseline --no-multichain --no-roofline > $O/r02_bench_cfg2_chains2.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for w in cfg2 cfg4 cfg5; do
  if [ $w = cfg2 ]; then A="--steps 20 --warmup 5"; else A="--workload $w --steps 3 --warmup 1 --burnin 5 --repeats 2"; fi
  A="$A --no-cpu-baseline --no-multichain --no-extras"
  rm -rf /tmp/prof_$w
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -- python3 $R/bench.py $A > $O/r02_bench_${w}_under_rocprof.json 2> /tmp/prof_$w.err
  DB=$(find /tmp/prof_$w -name "*.db" | head -1)
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py $A" > $O/r02_${w}_kernel_stats.txt
  python3 $R/tools/rocpd_summary.py $DB --gaps >> $O/r02_${w}_kernel_stats.txt 2>&1
done
ls -la $O
