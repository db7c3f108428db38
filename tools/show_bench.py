#!/usr/bin/env python3
"""Views of a bench.py JSON line.

  python tools/show_bench.py FILE...            one line per file: value, ms / astep, kernels, roofline
  python tools/show_bench.py --table FILE       BASELINE.md's results table (section 3), filled from the ONE line
                                                `python bench.py` prints (BENCH_rNN.json's "parsed" object works too)
README.md's table is the output of --table on the committed line (profiles/rNN_bench.json)."""
import json
import sys


def load(path):
    d = json.load(open(path))
    return d.get("parsed", d)  # the driver's record wraps the line


def fmt(x, unit=""):
    if x is None:
        return "—"
    x = float(x)
    if x >= 1e6:
        return f"{x / 1e6:.3g} M{unit}"
    if x >= 1e4:
        return f"{x / 1e3:.3g} k{unit}"
    return f"{x:.4g}{unit}"


def table(d):
    rows = []
    hdr = ("| cfg | backend | devices / cores | particle-steps/s | tree-updates/s | algorithmic GB/s (SURVEY 8d) | roofline fraction "
           "| speed-up vs restated CPU |")
    rows += [hdr, "|---|---|---|---|---|---|---|---|"]
    c1 = d.get("cfg1")
    if c1:
        rows.append(f"| 1 | CPU restatement | 1 core | {fmt(c1['value'])} | {fmt(c1['tree_updates_per_s'])} | n/a | n/a | 1× |")
    cpu = d.get("cpu_baseline") or {}
    if cpu:
        rows.append(f"| 2 | CPU restatement | 1 core | {fmt(cpu['value'])} | {fmt(cpu.get('tree_updates_per_s'))} | n/a | n/a | 1× |")
        if "all_cores" in cpu:
            ac = cpu["all_cores"]
            rows.append(f"| 2 | CPU restatement | {ac['cores']} chains on {ac['cores']} cores | {fmt(ac['value'])} | — | n/a | n/a | "
                        f"{ac['value'] / cpu['value']:.1f}× |")

    def gpu_row(cfg, leg, label="1 GPU", note=""):
        rf = leg.get("roofline") or {}
        frac = rf.get("frac")
        wf = rf.get("whole_step_frac")
        cb = leg.get("cpu_baseline") or {}
        sp = leg.get("speedup_vs_cpu_baseline")
        sp8 = (leg["value"] / cb["all_cores"]["value"]) if cb.get("all_cores") else leg.get("speedup_vs_cpu_all_cores")
        rows.append(
            f"| {cfg} | HIP gfx950{note} | {label} | **{fmt(leg['value'])}** ({leg['ms_per_step']:.3g} ms / astep) | "
            f"{fmt(leg.get('tree_updates_per_s'))} | {fmt(leg.get('algorithmic_GBps_whole_step'))} | "
            f"{rf.get('kernel', '—')} {'' if frac is None else f'{frac:.2f}'} of its {rf.get('bound', '')} peak; whole step "
            f"{'—' if wf is None else f'{wf:.2f}'} of 8 TB/s | "
            f"{'—' if sp is None else f'{sp:.0f}× (1 core)'}{'' if not sp8 else f', {sp8:.0f}× (8 cores)'} |")

    if d.get("n_gpus", 1) == 1:
        gpu_row(2, d)
        t1 = (d.get("tune1") or {}).get("astep")
        if t1:
            rows.append(f"| 2 | HIP gfx950, tune=1 | 1 GPU | {fmt(t1['value'])} ({t1['ms_per_step']:.3g} ms / astep) | "
                        f"{fmt(t1['tree_updates_per_s'])} | — | — | — |")
        cc = d.get("concurrent_chains")
        if cc:
            rows.append(f"| 2 | HIP gfx950, {cc['chains_per_gpu']} chains on the one GPU (resident) | 1 GPU | {fmt(cc['value'])} "
                        f"aggregate | — | — | — | — |")
    else:
        gpu_row(3 if "cfg2" in d["config"]["workload"] else 5, d, f"{d['n_gpus']} GPUs (1 chain each)")
    for wn, leg in (d.get("workloads") or {}).items():
        gpu_row(wn[-1], leg)
        cb = leg.get("cpu_baseline") or {}
        if cb:
            ac = cb.get("all_cores") or {}
            rows.append(f"| {wn[-1]} | CPU restatement | 1 core{'' if not ac else ' / 8 chains on 8 cores'} | {fmt(cb['value'])}"
                        f"{'' if not ac else ' / ' + fmt(ac['value'])} | — | n/a | n/a | 1× |")
    return "\n".join(rows)


def one_line(f, d):
    k = d.get("roofline_kernels") or {}
    r = d.get("roofline") or {}
    print("%s: %.4f M (%.4f..%.4f) %s, %.3f ms/astep | %s | roofline %s %s frac %s" % (
        d.get("config", {}).get("workload", f), d["value"] / 1e6, d.get("value_min", 0) / 1e6, d.get("value_max", 0) / 1e6,
        d.get("unit", ""), d["ms_per_step"], {a: (round(b["avg_us"], 2), round(b["pct"], 1)) for a, b in k.items()},
        r.get("kernel"), r.get("bound"), None if r.get("frac") is None else round(r["frac"], 3)))


def main():
    args = sys.argv[1:]
    as_table = bool(args) and args[0] == "--table"
    for f in args[1:] if as_table else args:
        try:
            d = load(f)
        except Exception as e:  # noqa: BLE001
            print(f, "unreadable:", e)
            continue
        if as_table:
            print(table(d))
        else:
            one_line(f, d)


if __name__ == "__main__":
    main()
