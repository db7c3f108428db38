#!/bin/bash
# Issue / wait accounting of the hot kernels (SQ counters, one pass): per kernel, all launches and the
# heaviest 15 % (the round-0 slots).  usage (GPU box): tools/pmc_busy.sh OUT.txt <bench.py args>
OUT=$1; shift; R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_busy
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
  -d /tmp/pmc_busy --output-format csv -- python3 $R/bench.py "$@" > /dev/null 2> /tmp/pmc_busy.err || tail -3 /tmp/pmc_busy.err
python3 - "$R/$OUT" "$*" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
per = defaultdict(lambda: defaultdict(dict))
for f in glob.glob("/tmp/pmc_busy/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]); k = re.sub(r"^void ", "", k)
        per[k][int(row["Dispatch_Id"])][row["Counter_Name"]] = per[k][int(row["Dispatch_Id"])].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
out = open(sys.argv[1], "w")
out.write("# rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -- python3 bench.py " + sys.argv[2] + "\n")
for k, d in per.items():
    if not k.startswith("k_"): continue
    ds = sorted(d.values(), key=lambda c: c.get("SQ_WAVE_CYCLES", 0))
    for label, sel in (("all", ds), ("heaviest 15%", ds[int(len(ds) * 0.85):])):
        if not sel: continue
        s = defaultdict(float)
        for c in sel:
            for n, v in c.items(): s[n] += v
        wc = s["SQ_WAVE_CYCLES"] or 1
        out.write(f"{k[:44]:44s} {label:13s} n={len(sel):5d} waves/launch={s['SQ_WAVES']/len(sel):9.0f} valu_insts/launch={s['SQ_INSTS_VALU']/len(sel):12.0f} "
                  f"wave-cycles: active_any {100*s['SQ_ACTIVE_INST_ANY']/wc:5.1f}% active_valu {100*s['SQ_ACTIVE_INST_VALU']/wc:5.1f}% "
                  f"wait_inst {100*s['SQ_WAIT_INST_ANY']/wc:5.1f}% wait_any {100*s['SQ_WAIT_ANY']/wc:5.1f}%  "
                  f"valu cycles/inst {s['SQ_ACTIVE_INST_VALU']/max(s['SQ_INSTS_VALU'],1):5.2f}  busy_cycles/launch {s['SQ_BUSY_CYCLES']/len(sel):10.0f}\n")
out.close()
print(open(sys.argv[1]).read())
PY
