#!/bin/bash
# HBM traffic / VALU instruction counts of the hot kernels, per launch, from separate rocprofv3 --pmc passes
# (counters only: no trace domains), written to gpurun_out/${ROUND:-r06}_pmc_<workload>.json -- copy it to profiles/,
# where bench.py reads `roofline.traffic` / the instruction counts from.  Run on the GPU box from the repo root.
# The run under the counters follows the HEADLINE protocol (100 tune=1 asteps of burn-in, then tune=0); the
# averages are taken over the LAST 15 % of each kernel's dispatches, i.e. the steady-state tune=0 part.
# FETCH_SIZE x2 / WRITE_SIZE x1: calibrated on known byte counts for every stream width the passes use
# (tools/microbench/fetch_calib.*, profiles/r03_fetch_calibration.json).
# usage: tools/pmc_collect.sh cfg2|cfg4|cfg5
W=${1:-cfg2}; R=$PWD; cd /tmp; export TMPDIR=/tmp
if [ $W = cfg2 ]; then A="--steps 20 --warmup 5 --burnin 100 --repeats 2"; else A="--workload $W --steps 10 --warmup 2 --burnin 100 --repeats 2"; fi
A="$A --no-extras --no-cpu-baseline --no-roofline --no-multichain --no-workloads"
DIRS=""
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_WAVES; do
  rm -rf /tmp/pmc_${W}_$C
  rocprofv3 --pmc $C -d /tmp/pmc_${W}_$C --output-format csv -- python3 $R/bench.py $A > /dev/null 2> /tmp/pmc_${W}_$C.err || tail -3 /tmp/pmc_${W}_$C.err
  DIRS="$DIRS /tmp/pmc_${W}_$C"
done
python3 $R/tools/pmc_summary.py --tail 0.15 $DIRS > /tmp/pmc_${W}.json
python3 - <<PY
import importlib.util, json
spec = importlib.util.spec_from_file_location("bench_mod", "$R/bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
raw = json.load(open("/tmp/pmc_${W}.json"))
out = {"kernel_source_sha256": bench.kernel_source_sha256(),
       "kernel_source": "sha256 over pymc_bart_amd/csrc/*.hip, *.h, include/*.h and the hipcc flags (bench.kernel_source_sha256; bench.py compares it with the tree it runs from: roofline.traffic_stale)",
       "command": "rocprofv3 --pmc <C> --output-format csv -- python3 bench.py $A   (one pass per counter C in FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU, SQ_WAVES; averaged over the last 15 % of each kernel's dispatches by tools/pmc_summary.py --tail 0.15)",
       "workload": "$W",
       "note": "FETCH_SIZE on gfx950 counts a 128-B line fetched from the fabric as 64 B: x2 (exact to 4 digits for 1 / 4 / 16 / 32 B-per-lane streams, profiles/r03_fetch_calibration.json); WRITE_SIZE exact; units KB per launch as the counters report them; SQ_INSTS_VALU counts wave-instructions",
       "raw": {k: v for k, v in raw.items() if k.startswith("k_")}}
for k, v in raw.items():
    short = k.split("<")[0]
    f, w = v.get("FETCH_SIZE_avg_per_launch"), v.get("WRITE_SIZE_avg_per_launch")
    e = out.setdefault(short, {})
    if f is not None and w is not None and v.get("launches", 0) > e.get("launches", 0):
        e.update(launches=v["launches"], hbm_bytes_per_launch_corrected=(2.0 * f + w) * 1024.0,
                 fetch_KB_per_launch=f, write_KB_per_launch=w)
    if v.get("SQ_INSTS_VALU_avg_per_launch") is not None and v.get("launches", 0) >= e.get("valu_launches", 0):
        e.update(valu_launches=v["launches"], valu_wave_insts_per_launch=v["SQ_INSTS_VALU_avg_per_launch"],
                 waves_per_launch=v.get("SQ_WAVES_avg_per_launch"))
json.dump(out, open("$R/gpurun_out/${ROUND:-r06}_pmc_${W}.json", "w"), indent=1)
print({k: v for k, v in out.items() if k.startswith("k_")})
PY
