#!/bin/bash
# rocprofv3 kernel trace of a bench.py run, summarised (per-kernel table, duration histograms, gaps).
# usage (GPU box, repo root): tools/ktrace.sh OUTFILE <bench.py args...>
OUT=$1; shift; R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktrace_prof
rocprofv3 --kernel-trace --stats -d /tmp/ktrace_prof -- python3 $R/bench.py "$@" > /tmp/ktrace_bench.json 2> /tmp/ktrace.err
DB=$(find /tmp/ktrace_prof -name "*.db" | head -1)
echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py $*" > $R/$OUT
python3 $R/tools/rocpd_summary.py $DB --gaps --hist >> $R/$OUT 2>&1
