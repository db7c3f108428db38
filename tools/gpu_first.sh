#!/bin/bash
# round-3 first GPU contact: GPU suite, the default bench line, FETCH/WRITE calibration, counter names
R=$PWD; O=$R/gpurun_out/r3a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; echo "bench rc=$?"; tail -3 $O/bench.time
tail -5 $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1; grep -i "TCC_EA0_RD\|TCC_EA0_WR\|FETCH_SIZE\|WRITE_SIZE" $O/counters.txt | head -40
timeout 600 python3 $R/tools/microbench/fetch_calib.py $O/fetch_calibration.json > $O/fetch_calib.log 2>&1; echo "calib rc=$?"; tail -60 $O/fetch_calib.log
