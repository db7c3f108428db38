#!/usr/bin/env python3
"""One-line summary of a bench.py JSON line: python tools/show_bench.py FILE..."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:  # noqa: BLE001
        print(f, "unreadable:", e)
        continue
    k = d.get("roofline_kernels") or {}
    r = d.get("roofline") or {}
    print("%s: %.4f M (%.4f..%.4f) %s/s, %.3f ms/astep | %s | roofline %s %s frac %s" % (
        d.get("config", {}).get("workload", f), d["value"] / 1e6, d.get("value_min", 0) / 1e6, d.get("value_max", 0) / 1e6,
        d.get("unit", ""), d["ms_per_step"], {a: (round(b["avg_us"], 2), round(b["pct"], 1)) for a, b in k.items()},
        r.get("kernel"), r.get("bound"), None if r.get("frac") is None else round(r["frac"], 3)))
