#!/bin/bash
# Parity hunts of a round on the GPU box (from the repo root): random configurations through both backends, bit for
# bit -- plain, with the chain migrating between the backends through the chain image, large n, the upstream-semantics
# switches, K-vector configurations, the 16-bit order keys forced on -- and the soak of every stepping API.
# usage: tools/hunts.sh OUTDIR [SECONDS_PER_LEG] [FIRST_SEED]     (logs under OUTDIR; copy the ones to keep into profiles/)
O=${1:-gpurun_out/hunts}; T=${2:-150}; B=${3:-600000}; mkdir -p $O   # B: first seed of the call (every leg takes its own range above it)
python tools/fuzz_hunt.py $((B + 10000)) 100000 $T migrate            > $O/fuzz_migrate.log 2>&1
python tools/fuzz_hunt.py $((B + 20000)) 100000 $T migrate compat     > $O/fuzz_migrate_compat.log 2>&1
python tools/fuzz_hunt.py $((B + 30000)) 100000 $T migrate mk         > $O/fuzz_migrate_mk.log 2>&1
python tools/fuzz_hunt.py $((B + 40000)) 100000 $T large migrate      > $O/fuzz_migrate_large.log 2>&1
python tools/fuzz_hunt.py $((B + 50000)) 100000 $T                    > $O/fuzz_default.log 2>&1
PGB_X32_MIN_MB=0 python tools/fuzz_hunt.py $((B + 60000)) 100000 $T migrate > $O/fuzz_migrate_keys_forced.log 2>&1
PGB_X32_MIN_MB=0 python tools/fuzz_hunt.py $((B + 70000)) 100000 $T mk compat > $O/fuzz_mk_compat_keys_forced.log 2>&1
python tools/soak_parity.py $T 6 > $O/soak.log 2>&1
tail -n 2 $O/*.log
