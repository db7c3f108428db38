#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/microbench/fetch_calib.hip).

usage (GPU box, repo root):  python3 tools/microbench/fetch_calib.py OUT.json
Runs the microbenchmark three times: plain (for the byte counts it prints) and under
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, counters only), then writes, per kernel,
bytes moved / counter value (counters are reported in KB) = the factor a raw counter must be multiplied by."""
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "tools", "microbench", "fetch_calib")


def counter(cname):
    d = f"/tmp/fcal_{cname}"
    subprocess.call(["rm", "-rf", d])
    subprocess.check_call(["rocprofv3", "--pmc", cname, "-d", d, "--output-format", "csv", "--", EXE],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp",
                          env=dict(os.environ, TMPDIR="/tmp"))
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"])
            k = re.sub(r"^void ", "", k).replace(" ", "")
            out[k] = out.get(k, 0.0) + float(row["Counter_Value"])
    return out


def main():
    moved = json.loads(subprocess.check_output([EXE], text=True).strip().splitlines()[-1])
    res = {"note": "factor = bytes the kernel moved / (counter x 1024); FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports "
                   "them; buffer 2 GiB (beyond the 256 MiB Infinity Cache), each kernel streams it once",
           "kernels": {}}
    fetch, write = counter("FETCH_SIZE"), counter("WRITE_SIZE")
    for name, b in moved.items():
        if name == "rd_tile_lines_bytes":
            continue
        k = name.replace(" ", "")
        e = {"bytes_moved": b}
        if k.startswith("rd") and fetch.get(k):
            e["FETCH_SIZE_KB"] = fetch[k]
            e["fetch_factor"] = b / (fetch[k] * 1024.0)
            if k == "rd_tile":
                e["fetch_factor_vs_128B_lines_touched"] = moved["rd_tile_lines_bytes"] / (fetch[k] * 1024.0)
        if k.startswith("wr") and write.get(k):
            e["WRITE_SIZE_KB"] = write[k]
            e["write_factor"] = b / (write[k] * 1024.0)
        res["kernels"][name] = e
    json.dump(res, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main()
