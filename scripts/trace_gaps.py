#!/usr/bin/env python3
"""GPU idle time inside the traced steps of a rocprofv3 kernel trace (rocpd sqlite): the union of the kernel intervals
against the span, per step (steps are cut at the optimizer kernel), and the largest gaps with the kernels around them.
  python scripts/trace_gaps.py p_results.db [--skip 5]"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--skip", type=int, default=5, help="leading steps to ignore (warm-up)")
ap.add_argument("--end", default="sgd_step_kernel", help="kernel that ends a step")
args = ap.parse_args()
db = sqlite3.connect(args.db)
rows = db.execute("select name, start, end from kernels order by start").fetchall()
steps, cur = [], []
for r in rows:
    cur.append(r)
    if args.end in r[0]:
        steps.append(cur)
        cur = []
steps = steps[args.skip:]
tot_span = tot_busy = 0
gaps = []
for st in steps:
    s0, e_max, busy = st[0][1], st[0][1], 0
    for name, s, e in st:
        if s > e_max:
            gaps.append((s - e_max, prev, name))
            busy += 0
            cover_start = s
        busy += max(0, e - max(s, e_max))
        if e > e_max:
            e_max, prev = e, name
    tot_span += e_max - s0
    tot_busy += busy
n = len(steps)
print("steps %d: span %.3f ms, some kernel running %.3f ms, idle %.3f ms per step" % (n, tot_span / n / 1e6, tot_busy / n / 1e6,
                                                                                       (tot_span - tot_busy) / n / 1e6))
agg = {}
for g, a, b in gaps:
    k = (a[:50], b[:50])
    agg.setdefault(k, [0, 0])
    agg[k][0] += g
    agg[k][1] += 1
for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print("%8.1f us/step  x%5.1f  %-50s -> %s" % (g / n / 1e3, c / n, a, b))
