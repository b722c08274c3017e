#!/usr/bin/env python3
"""Ordered kernel timeline of ONE steady training step from a rocprofv3 rocpd database (kernel trace).

  python scripts/step_timeline.py gpurun_out/tl/p_results.db [--marker sgd_step_kernel] [--step -2]

A step is the interval between two launches of the marker kernel (the optimizer's).  For every launch in it: start
offset, duration, the idle gap since the previous kernel END on the same queue and since the previous end on ANY queue
(the second is what the step pays for: time in which the GPU ran nothing).  The footer sums busy time and gaps per
queue, and the union busy time.
"""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:64]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--marker", default="sgd_step_kernel")
    ap.add_argument("--step", type=int, default=-2, help="which marker interval (negative: from the end)")
    ap.add_argument("--quiet", action="store_true", help="footer only")
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
    sel = "select name, start, end, %s from kernels order by start" % (qcol or "0")
    rows = db.execute(sel).fetchall()
    marks = [i for i, r in enumerate(rows) if args.marker in r[0]]
    if len(marks) < 3:
        raise SystemExit("fewer than three %s launches in the trace" % args.marker)
    a, b = marks[args.step - 1] + 1, marks[args.step] + 1
    step = rows[a:b]
    t0 = rows[a - 1][2]  # the previous step's marker END: this step's first gap counts
    last_q, last_any = {}, t0
    busy_q, gap_q, n_q = {}, {}, {}
    idle = 0.0
    if not args.quiet:
        print("%5s %10s %9s %8s %8s %-4s %s" % ("#", "start us", "dur us", "gap(q)", "idle", "q", "kernel"))
    for i, (name, s, e, q) in enumerate(step):
        gq = (s - last_q[q]) / 1e3 if q in last_q else 0.0
        ga = max(0.0, (s - last_any) / 1e3)
        idle += ga
        if not args.quiet:
            print("%5d %10.1f %9.1f %8.1f %8.1f %-4s %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, gq, ga, q, short(name)))
        last_q[q] = e
        last_any = max(last_any, e)
        busy_q[q] = busy_q.get(q, 0.0) + (e - s) / 1e3
        gap_q[q] = gap_q.get(q, 0.0) + max(gq, 0.0)
        n_q[q] = n_q.get(q, 0) + 1
    span = (step[-1][2] - t0) / 1e3
    print("step span %.1f us, %d launches, GPU idle (no kernel on any queue) %.1f us" % (span, len(step), idle))
    for q in sorted(busy_q, key=lambda k: -busy_q[k]):
        print("  queue %-6s %4d launches  busy %8.1f us  gaps between its kernels %8.1f us" % (q, n_q[q], busy_q[q], gap_q[q]))


if __name__ == "__main__":
    main()
