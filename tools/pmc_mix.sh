#!/bin/bash
# Instruction mix / LDS accounting of the hot kernels: one rocprofv3 --pmc pass per counter group (counters only,
# no trace domains), per kernel over all launches and over the heaviest 15 % (the slots that start a tree).
# usage (GPU box, repo root): tools/pmc_mix.sh OUT.txt <bench.py args>
OUT=$1; shift; R=$PWD
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
G2="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE"
i=0
for G in "$G1" "$G2"; do
  i=$((i+1)); rm -rf /tmp/pmc_mix_$i
  rocprofv3 --pmc $G -d /tmp/pmc_mix_$i --output-format csv -- python3 $R/bench.py "$@" > /dev/null 2> /tmp/pmc_mix_$i.err || tail -3 /tmp/pmc_mix_$i.err
done
python3 - "$R/$OUT" "$*" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
out = open(sys.argv[1], "w")
out.write("# rocprofv3 --pmc <group> --output-format csv -- python3 bench.py " + sys.argv[2] + "   (two passes, see tools/pmc_mix.sh)\n")
for g in (1, 2):
    per = defaultdict(lambda: defaultdict(dict))
    for f in glob.glob(f"/tmp/pmc_mix_{g}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"]); k = re.sub(r"^void ", "", k)
            c = per[k][int(row["Dispatch_Id"])]
            c[row["Counter_Name"]] = c.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    for k, d in sorted(per.items()):
        if not k.startswith("k_") or len(d) < 50: continue
        ds = sorted(d.values(), key=lambda c: c.get("SQ_WAVE_CYCLES", 0))
        for label, sel in (("all", ds), ("heaviest 15%", ds[int(len(ds) * 0.85):])):
            s = defaultdict(float)
            for c in sel:
                for n, v in c.items(): s[n] += v
            wc = s["SQ_WAVE_CYCLES"] or 1
            txt = f"{k[:40]:40s} {label:13s} n={len(sel):5d} "
            for n in sorted(s):
                if n in ("SQ_WAVE_CYCLES",): continue
                if n.startswith("SQ_INSTS") or n == "SQ_WAVES": txt += f"{n[3:]}/launch={s[n]/len(sel):.0f} "
                else: txt += f"{n[3:]}={100*s[n]/wc:.1f}%wc "
            out.write(txt + "\n")
out.close()
print(open(sys.argv[1]).read())
PY
