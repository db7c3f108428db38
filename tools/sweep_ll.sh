#!/bin/bash
# GPU box: sweep the likelihood-pass launch knobs on a workload.  usage: tools/sweep_ll.sh TAG WORKLOAD
O=gpurun_out/${1:-sweepll}; W=${2:-cfg4}; mkdir -p $O
Q="--workload $W --steps 5 --warmup 1 --burnin 10 --repeats 3 --no-multichain --no-cpu-baseline --no-extras"
run() { n=$1; shift; env "$@" timeout 600 python bench.py $Q > $O/$n.json 2>/dev/null; echo -n "$n  "; python3 tools/show_bench.py $O/$n.json | cut -c1-60,100-330; }
run base X=1
for t in 512 768 1280 2048; do run lltarget_$t PGB_LL_TARGET=$t; done
for g in 768 1280 1536 2048; do run llgrid_$g PGB_LL_GRID=$g PGB_LL_TARGET=$g; done
for t in 768 1280 2048; do run rows_target_$t PGB_ROWS_TARGET=$t PGB_ROWS_TARGET_INIT=$t; done
run base2 X=1
