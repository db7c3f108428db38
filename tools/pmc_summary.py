#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel launch.

usage: tools/pmc_summary.py <dir with *counter_collection.csv> [more dirs...]
Prints JSON {kernel: {counter: avg per launch, "launches": n}} (kernel names shortened)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = defaultdict(dict)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(lambda: defaultdict(set))
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"])
            k = re.sub(r"^void ", "", k)
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k][row["Counter_Name"]].add(row.get("Dispatch_Id", row.get("Correlation_Id")))
        for k in acc:
            for c in acc[k]:
                n = len(disp[k][c])
                out[k][c + "_avg_per_launch"] = acc[k][c] / max(n, 1)
                out[k]["launches"] = n
print(json.dumps(out, indent=1))
