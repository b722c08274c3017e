#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel table.

  python scripts/prof_summary.py gpurun_out/prof2/r01b_results.db [--steps K] [--csv out.csv]

Prints name, calls, total ms, average us, share; with --steps also ms per step (total / K).
Only reads the sqlite file rocprofv3 wrote; used to produce the summaries under profiles/.
"""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    if len(name) > 110:
        name = name[:107] + "..."
    return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, default=0, help="profiled steps (warm-up + timed) for a per-step column")
    ap.add_argument("--csv", default=None)
    ap.add_argument("--top", type=int, default=60)
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    rows = db.execute("select name, count(*), sum(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    span = db.execute("select min(start), max(end) from kernels").fetchone()
    total = sum(r[2] for r in rows)
    # the median duration beside the mean: the mean of a short traced run carries its warm-up launches (cold clocks,
    # first-touch work), the median is what a steady step sees and what bench.py's live timing should be compared with
    med = {}
    for name, dur in db.execute("select name, duration from kernels order by name, duration"):
        med.setdefault(name, []).append(dur)
    med = {k: v[len(v) // 2] for k, v in med.items()}
    lines = ["name,calls,total_ns,avg_ns,min_ns,max_ns,pct,vgpr,agpr,sgpr,lds,median_ns"]
    print("%-112s %6s %10s %10s %6s %5s %5s %7s %9s" % ("kernel", "calls", "total ms", "avg us", "%", "vgpr", "agpr", "lds",
                                                        "median us"))
    for i, (name, n, tot, mn, mx, vg, ag, sg, lds) in enumerate(rows):
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.3f,%d,%d,%d,%d,%d' % (name, n, tot, tot / n, mn, mx, 100.0 * tot / total,
                                                                 vg or 0, ag or 0, sg or 0, lds or 0, med[name]))
        if i < args.top:
            print("%-112s %6d %10.3f %10.1f %6.2f %5d %5d %7d %9.1f" % (short(name), n, tot / 1e6, tot / n / 1e3,
                                                                       100.0 * tot / total, vg or 0, ag or 0, lds or 0,
                                                                       med[name] / 1e3))
    print("kernels: %d launches, busy %.3f ms, first-to-last span %.3f ms" % (sum(r[1] for r in rows), total / 1e6,
                                                                              (span[1] - span[0]) / 1e6))
    if args.steps:
        print("per step over %d steps: busy %.3f ms" % (args.steps, total / 1e6 / args.steps))
    if args.csv:
        with open(args.csv, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
