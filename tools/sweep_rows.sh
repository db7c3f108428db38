#!/bin/bash
# GPU box: sweep the row-pass launch knobs (environment, read at pgb_create) on the cfg2 headline.
O=gpurun_out/${1:-sweep}; mkdir -p $O
Q="--steps 20 --warmup 5 --repeats 5 --no-extras --no-cpu-baseline --no-multichain --no-roofline"
run() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py $Q > $O/$n.json 2>/dev/null
  python3 -c "
import json
d=json.load(open('$O/$n.json')); print('%-34s %.4f M (%.4f..%.4f)' % ('$n', d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6))"
}
run base X=1
for t in 384 512 768 896; do run target_$t PGB_ROWS_TARGET=$t; done
for t in 512 640 1024 1280; do run init_$t PGB_ROWS_TARGET_INIT=$t; done
for g in 512 768; do run grid_$g PGB_ROWS_GRID=$g; done
run base2 X=1
