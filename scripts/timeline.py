#!/usr/bin/env python3
"""Where does the WALL time of a step go?  Sweep over the kernel intervals of a rocprofv3 kernel trace (rocpd
database): time with no kernel running (idle: launch gaps, host-bound stretches), time with exactly one kernel
(attributed to it) and time with several (split evenly between them).  Per-kernel attributed wall time per step.

  python scripts/timeline.py p_results.db --steps K [--skip S]   (the first S steps' share of the trace is dropped)
"""
import argparse
import re
import sqlite3
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, required=True, help="steps in the trace (warm-up + timed)")
    ap.add_argument("--skip", type=float, default=0.5, help="fraction of the trace (by time) to drop at the start")
    ap.add_argument("--top", type=int, default=30)
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    # step boundaries = ends of the optimizer's last launch; keep the last `steps` whole steps
    marks = [r[2] for r in rows if "pack_dgrad_batched" in r[0]]
    if len(marks) > a.steps:
        lo, hi = marks[-a.steps - 1], marks[-1]
        rows = [r for r in rows if r[1] >= lo and r[2] <= hi]
        a.skip = 0.0
    else:
        t_lo, t_hi = rows[0][1], max(r[2] for r in rows)
        cut = t_lo + a.skip * (t_hi - t_lo)
        rows = [r for r in rows if r[1] >= cut]
    ev = []
    for i, (n, s, e) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()
    live = set()
    att = defaultdict(float)
    idle = 0.0
    multi = 0.0
    prev = ev[0][0]
    for t, kind, i in ev:
        dt = t - prev
        if dt > 0:
            if not live:
                idle += dt
            else:
                if len(live) > 1:
                    multi += dt
                for j in live:
                    att[short(rows[j][0])] += dt / len(live)
        prev = t
        if kind:
            live.add(i)
        else:
            live.discard(i)
    span = ev[-1][0] - ev[0][0]
    steps = a.steps * (1 - a.skip)
    print("span %.3f ms = %.3f ms/step over ~%.1f steps; idle %.3f ms/step (%.1f%%); >=2 kernels running %.3f ms/step" % (
        span / 1e6, span / 1e6 / steps, steps, idle / 1e6 / steps, 100 * idle / span, multi / 1e6 / steps))
    tot = sum(att.values())
    for n, v in sorted(att.items(), key=lambda kv: -kv[1])[:a.top]:
        print("%-72s %8.3f ms/step %5.1f%%" % (n, v / 1e6 / steps, 100 * v / span))


if __name__ == "__main__":
    main()
