#!/usr/bin/env python3
"""Runs of identical consecutive launches in a rocprofv3 kernel trace: count, median duration, median gap to the
next launch.  For micro-benchmarks that launch one shape many times in a row (scripts/bench_pw2.py).
  python scripts/prof_runs.py gpurun_out/x/p_results.db [--min 10]"""
import argparse
import re
import sqlite3
import statistics


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--min", type=int, default=10)
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
    extra = (", %s, %s, %s" % (gx, gx.replace("_x", "_y"), gx.replace("_x", "_z"))) if gx else ", 0, 0, 0"
    extra += (", %s" % wx) if wx else ", 1"
    rows = db.execute("select name, start, end %s from kernels order by start" % extra).fetchall()
    runs = []
    for r in rows:
        key = (r[0], r[3], r[4], r[5])
        if runs and runs[-1][0] == key:
            runs[-1][1].append(r)
        else:
            runs.append((key, [r]))
    for key, rs in runs:
        if len(rs) < a.min:
            continue
        d = [r[2] - r[1] for r in rs]
        gaps = [rs[i + 1][1] - rs[i][2] for i in range(len(rs) - 1)]
        name = re.sub(r"\(anonymous namespace\)::|^void ", "", key[0])[:70]
        wg = rs[0][6] or 1
        print("%-70s grid %6d x %3d x %3d  n %3d  dur med %7.1f us min %7.1f  gap med %6.1f us" % (
            name, (key[1] or 0) // wg, key[2] or 0, key[3] or 0, len(rs), statistics.median(d) / 1e3, min(d) / 1e3,
            statistics.median(gaps) / 1e3 if gaps else 0.0))


if __name__ == "__main__":
    main()
