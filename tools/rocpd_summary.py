#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (sqlite) result: per-kernel count / total / avg / min / max.

usage: tools/rocpd_summary.py results.db [--gaps]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    scol = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    namecol = "kernel_name" if "kernel_name" in scol else "display_name"
    q = (f"select s.{namecol}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
         f"max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.{namecol} "
         "order by 3 desc")
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print(f"{'kernel':60s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for name, n, s, a, mn, mx in rows:
        print(f"{name[:60]:60s} {n:8d} {s/1e6:10.3f} {a/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.1f}")
    if "--hist" in sys.argv:  # where each hot kernel's time goes: launches and time by duration bucket
        edges = [0, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 1 << 30]
        for name, n, s_, a, mn, mx in rows[:4]:
            durs = [r[0] / 1e3 for r in c.execute(
                f"select d.end-d.start from {kd} d join {ks} s on d.kernel_id = s.id where s.{namecol} = ?", (name,))]
            tot_k = sum(durs) or 1.0
            print(f"histogram {name[:50]}")
            for lo, hi in zip(edges[:-1], edges[1:]):
                sel = [d for d in durs if lo <= d < hi]
                if sel:
                    print(f"   [{lo:4d},{hi if hi < 1 << 30 else 9999:5d}) us: {len(sel):6d} launches {100 * len(sel) / len(durs):5.1f} %  "
                          f"time {100 * sum(sel) / tot_k:5.1f} %  avg {sum(sel) / len(sel):7.2f} us")
    if "--gaps" in sys.argv:
        ev = list(c.execute(f"select d.start, d.end from {kd} d order by d.start"))
        gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
        gaps = [g for g in gaps if 0 <= g < 1e6]
        gaps.sort()
        if gaps:
            print(f"inter-kernel gaps: n={len(gaps)} median={gaps[len(gaps)//2]/1e3:.2f}us "
                  f"mean={sum(gaps)/len(gaps)/1e3:.2f}us p90={gaps[int(.9*len(gaps))]/1e3:.2f}us")
            span = ev[-1][1] - ev[0][0]
            print(f"span={span/1e6:.3f}ms busy={tot/1e6:.3f}ms")


if __name__ == "__main__":
    main()
