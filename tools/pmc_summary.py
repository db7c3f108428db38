#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel launch.

usage: tools/pmc_summary.py [--tail F] <dir with *counter_collection.csv> [more dirs...]
Prints JSON {kernel: {counter: avg per launch, "launches": n}} (kernel names shortened).
--tail F (0 < F <= 1): average over the LAST fraction F of each kernel's dispatches only -- the steady-state,
tune=0 part of a bench.py run whose first dispatches are the tune=1 burn-in."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = defaultdict(dict)
argv = sys.argv[1:]
tail = 1.0
if argv and argv[0] == "--tail":
    tail = float(argv[1])
    argv = argv[2:]
for d in argv:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))  # kernel -> counter -> dispatch -> value
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"])
            k = re.sub(r"^void ", "", k)
            per[k][row["Counter_Name"]][int(row.get("Dispatch_Id", row.get("Correlation_Id")))] += float(row["Counter_Value"])
        for k in per:
            for c in per[k]:
                ids = sorted(per[k][c])
                keep = ids[len(ids) - max(1, int(round(len(ids) * tail))):]
                out[k][c + "_avg_per_launch"] = sum(per[k][c][i] for i in keep) / len(keep)
                out[k]["launches"] = len(keep)
                out[k]["launches_total"] = len(ids)
print(json.dumps(out, indent=1))
