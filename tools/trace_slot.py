#!/usr/bin/env python3
"""Slot anatomy from the in-kernel device-clock stamps (-DPGB_TRACE build of the HIP library).

Builds build/variants/libpgbart_hip_trace.so (same flags as __graft_entry__.build plus
-DPGB_TRACE) when run with --build (no GPU needed), and on a GPU box runs cfg2 for a few asteps
and prints, per stage stamp, the median offset from the entry of k_ctrl (workgroup 1), for plain
SMC rounds and for the slots that start a tree.  Stamps (k_ctrl.h / k_rows.h):
  0 entry | 1 control word here | 2 finish stage done (wave 0) | 3 ancestor known by all waves
  4 node-table copy issued / end-of-tree bookkeeping done | 5 popped, prior coin | 6 split-row selection starts
  20 chunk counts scanned | 23 chunk known | 21 row known | 22 split value here
  7 split row found | 8 job written | 9,10 pre-draw waves done | 11 control word written (workgroup 0)
  12 k_rows entry (workgroup 0) | 13 jobs listed | 14 rows loaded + quantised (last item) | 15 item loop done
usage: python tools/trace_slot.py --build            (here)
       python tools/trace_slot.py [--asteps 10]      (GPU box)
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRACE_SO = os.path.join(ROOT, "build", "variants", "libpgbart_hip_trace.so")


def build(extra=(), out=TRACE_SO):
    import __graft_entry__ as g

    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *g.HIPCC_FLAGS, "-DPGB_TRACE", *extra, g.HIP_SRC, "-o", out]
    subprocess.check_call(cmd)
    print("built", out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--asteps", type=int, default=10)
    ap.add_argument("--burnin", type=int, default=40)
    ap.add_argument("--define", action="append", default=[])
    ap.add_argument("--stamps", action="store_true", help="also per-workgroup first/last clock readings of the row pass")
    ap.add_argument("--lib", default=TRACE_SO, help="trace build to load / write")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg4", "cfg5"])
    ap.add_argument("--list", action="store_true", help="the library was built with -DPGB_TRACE_LIST: stamps 27..29 are inside k_loglik's job list")
    ap.add_argument("--deltas", action="store_true", help="also the p10 / p50 / p90 of the stage-to-stage differences of k_loglik's stamps")
    ap.add_argument("--ll", action="store_true", help="the library was built with -DPGB_STAMP_LL: the per-workgroup stamps are k_loglik's")
    a = ap.parse_args()
    if a.build:
        build([f"-D{d}" for d in a.define], a.lib)
        return
    import torch  # noqa: F401  (one HIP runtime per process)

    from pymc_bart_amd import _abi, workloads
    from pymc_bart_amd._device import TorchHipMemory
    from pymc_bart_amd.sampler import Backend, PyBartSettings, PySampler

    lib = _abi.PGBLibrary(a.lib)
    lib.lib.pgb_debug_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    be = Backend(lib=lib, mem=TorchHipMemory(0))
    w = getattr(workloads, a.workload)()
    X, Y = w["X"], w["Y"]
    st = PyBartSettings.from_data(X, Y, m=w["m"], num_particles=w["num_particles"], seed=3415, family=w["family"],
                                  n_outputs=w.get("K", 1))
    s = PySampler(st, X, Y, np.zeros(X.shape[1], np.int32), np.ones(X.shape[1]), backend=be)
    s.set_likelihood([1.0] if w["family"] == "normal" else [])
    for _ in range(a.burnin):
        s.step(True)
    c0 = s.counters.as_dict()
    if a.stamps:
        s.profile(True)
    import time
    t0 = time.perf_counter()
    for _ in range(a.asteps):
        s.step(False)
    dt = time.perf_counter() - t0
    c1 = s.counters.as_dict()
    print({k: c1[k] - c0[k] for k in c1}, "sync astep ms", 1e3 * dt / a.asteps)
    NS = 4096
    buf = np.zeros((NS, 40), np.int64)
    lib.check(lib.lib.pgb_debug_trace(s._h, buf.ctypes.data, NS), "trace")
    # keep slots with a full set of k_ctrl stamps; stamp 15 = round of the proposal, 14 = attempt
    t = buf.astype(np.float64) * 0.01  # 100 MHz ticks -> us
    ok = (buf[:, 0] > 0) & (buf[:, 8] > buf[:, 0])
    order = np.argsort(buf[:, 0])
    order = order[ok[order]]
    t = t[order]
    rnd = buf[order, 17]
    att = buf[order, 16]
    nxt = np.roll(t[:, 0], -1) - t[:, 0]
    nxt[-1] = np.nan

    def show(name, sel):
        if sel.sum() < 5:
            print(name, "too few slots", int(sel.sum()))
            return
        parts = []
        # (24..30: the likelihood pass, workgroup 100: entry | job list | INIT part done | first item: labels asked |
        #  its first particle done | its particles done | its sums out -- the LAST item's stamps when there are two)
        # (32 / 33: the pre-draw waves have their Philox draws, 34 / 35: their split variable)
        for i in (1, 32, 33, 34, 35, 9, 10, 2, 3, 4, 5, 6, 20, 23, 21, 22, 7, 8, 11, 12, 13, 14, 15, 24, 36, 37, 38, 25, 26, 39, 27, 28, 29, 30, 31):
            d = t[sel, i] - t[sel, 0]
            d = d[(t[sel, i] > 0) & (d > -1) & (d < 200)]
            if d.size:
                parts.append(f"{i}:{np.median(d):.2f}")
        v = nxt[sel]
        v = v[np.isfinite(v) & (v < 200)]
        print(f"{name:28s} n={int(sel.sum()):5d}  " + " ".join(parts) + f"  | next k_ctrl {np.median(v):.2f} (mean {v.mean():.2f})")
        if a.deltas:  # distribution of the stage-to-stage differences of the likelihood pass (a median of offsets hides a two-humped stage)
            seq = (24, 36, 37, 27, 28, 29, 38, 25, 26, 39, 30, 31) if a.list else (24, 36, 37, 38, 25, 26, 39, 27, 29, 30, 31)
            out = []
            for i0, i1 in zip(seq[:-1], seq[1:]):
                m = (t[sel, i0] > 0) & (t[sel, i1] > 0)
                d = (t[sel, i1] - t[sel, i0])[m]
                d = d[(d > -50) & (d < 200)]
                if d.size:
                    q = np.percentile(d, [10, 50, 90])
                    out.append(f"{i0}>{i1}: {q[0]:.2f}/{q[1]:.2f}/{q[2]:.2f} (n={d.size})")
            print("    deltas p10/p50/p90  " + "  ".join(out))

    fresh = buf[order, 18] != 0
    stop = buf[order, 19] != 0
    show("plain round, attempt", (rnd >= 2) & (att == 1) & ~fresh & ~stop)
    show("plain round, no attempt", (rnd >= 2) & (att == 0) & ~fresh & ~stop)
    show("round 1 (after tree start)", (rnd == 1) & ~fresh & ~stop)
    show("tree end + next tree start", fresh & stop)
    show("step start (begin)", fresh & ~stop)
    if a.stamps:
        s.profile(False)
        lib.lib.pgb_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int]
        sb = np.zeros((NS, 1024, 2), np.int64)
        lib.check(lib.lib.pgb_debug_stamps(s._h, sb.ctypes.data, 0, NS), "stamps")
        sb = sb[order].astype(np.float64) * 0.01  # same ring index as the trace records

        def rows(name, sel):
            idx = np.nonzero(sel)[0]
            out = []
            for i in idx:
                st, en = sb[i, :, 0], sb[i, :, 1]
                m = (st > 0) & (en > st)
                if m.sum() < 8 or abs(st[m].min() - t[i, 12]) > (400 if a.ll else 50):
                    continue  # stale ring entry
                t0 = st[m].min()
                dur = en[m] - st[m]
                work = dur > 0.6  # workgroups that had an item
                nx = t[i + 1, 0] if i + 1 < t.shape[0] else np.nan
                out.append((m.sum(), work.sum(), st[m].max() - t0, en[m].max() - t0, np.median(dur[work]) if work.any() else 0,
                            dur.max(), t[i, 15] - t0, nx - en[m].max(), t0 - t[i, 8]))
            if len(out) < 3:
                print(name, "too few launches with stamps", len(out))
                return
            o = np.array(out)
            md = np.nanmedian(o, axis=0)
            print(f"{name:28s} n={len(out):4d} wgs {md[0]:.0f} with work {md[1]:.0f} | last start +{md[2]:.2f} | span {md[3]:.2f} | "
                  f"wg median {md[4]:.2f} longest {md[5]:.2f} | wg0 done +{md[6]:.2f} | end -> next k_ctrl {md[7]:.2f} | k_ctrl wg1 job -> first wg {md[8]:.2f}")

        def by_block(name, sel):
            idx = np.nonzero(sel)[0]
            acc_d, acc_s, acc_e, cnt, nbs = np.zeros(1024), np.zeros(1024), np.zeros(1024), 0, 0
            for i in idx:
                st, en = sb[i, :, 0], sb[i, :, 1]
                m = (st > 0) & (en > st)
                if m.sum() < 256 or abs(st[m].min() - t[i, 12]) > (400 if a.ll else 50):
                    continue
                nb = int(np.nonzero(m)[0].max()) + 1  # (the grid of the launch: the likelihood pass has fewer than 1024 workgroups)
                t0 = st[m].min()
                acc_d[:nb] += (en - st)[:nb]
                acc_s[:nb] += (st - t0)[:nb]
                acc_e[:nb] += (en - t0)[:nb]
                nbs = max(nbs, nb)
                cnt += 1
            if cnt == 0:
                return
            d, st_, en_ = acc_d / cnt, acc_s / cnt, acc_e / cnt
            print(f"{name}: mean per block over {cnt} launches")
            print("  blocks 0..15 duration", np.round(d[:16], 2))
            print("  blocks 0..15 start   ", np.round(st_[:16], 2))
            for lo in (0, 64, 128, 256, 384, 512, 640, 768, 896):
                hi = min(lo + (64 if lo < 128 else 128), nbs)
                if lo >= nbs:
                    break
                print(f"  blocks {lo:4d}..{hi - 1:4d}: start {st_[lo:hi].mean():.2f}  duration {d[lo:hi].mean():.2f} (max {d[lo:hi].max():.2f})  end {en_[lo:hi].mean():.2f} (max {en_[lo:hi].max():.2f})")
            d, en_ = d[:nbs], en_[:nbs]
            print("  by blockIdx % 8: duration", np.round([d[k::8].mean() for k in range(8)], 2), "end", np.round([en_[k::8].max() for k in range(8)], 2))
            print("  duration deciles over the blocks:", np.round(np.percentile(d, [0, 10, 25, 50, 75, 90, 100]), 2))

        def tails(name, sel, show_n=6):
            idx = np.nonzero(sel)[0]
            shown = 0
            pct = []
            for i in idx:
                st, en = sb[i, :, 0], sb[i, :, 1]
                m = (st > 0) & (en > st)
                if m.sum() < 512 or abs(st[m].min() - t[i, 12]) > (400 if a.ll else 50):
                    continue
                dur = np.where(m, en - st, 0.0)
                q = np.percentile(dur[m], [50, 90, 99, 100])
                pct.append(q)
                if shown < show_n:
                    top = np.argsort(dur)[-5:][::-1]
                    print(f"  {name} launch {i}: p50 {q[0]:.2f} p90 {q[1]:.2f} p99 {q[2]:.2f} max {q[3]:.2f} | slowest blocks "
                          + " ".join(f"{b}:{dur[b]:.1f}(start {st[b] - st[m].min():.1f})" for b in top))
                    shown += 1
            if pct:
                pq = np.median(np.array(pct), axis=0)
                print(f"  {name}: median over {len(pct)} launches of p50 / p90 / p99 / max = {pq[0]:.2f} / {pq[1]:.2f} / {pq[2]:.2f} / {pq[3]:.2f}")

        tails("plain round", (rnd >= 2) & ~fresh & ~stop)
        tails("round 1", (rnd == 1) & ~fresh & ~stop)
        by_block("plain round", (rnd >= 2) & ~fresh & ~stop)
        by_block("tree start", fresh & stop)
        print("-- row pass, per-workgroup stamps (us)")
        rows("plain round", (rnd >= 2) & ~fresh & ~stop)
        rows("round 1", (rnd == 1) & ~fresh & ~stop)
        rows("tree end + next tree start", fresh & stop)
    v = nxt[np.isfinite(nxt) & (nxt < 200)]
    print("all slots: mean period %.2f us, median %.2f us, n=%d" % (v.mean(), np.median(v), v.size))


if __name__ == "__main__":
    main()
