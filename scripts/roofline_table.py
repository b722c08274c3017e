#!/usr/bin/env python3
"""Joins a kernel-trace summary (scripts/prof_summary.py --csv) with the PMC per-kernel means
(scripts/pmc_summary.py --csv) into the two rooflines the task asks for:
  MFMA kernels      : MFMA-pipe busy share (SQ_VALU_MFMA_BUSY_CYCLES over GRBM_GUI_ACTIVE x SIMDs)
  bandwidth kernels : achieved HBM GB/s = (FETCH_SIZE x 2 + WRITE_SIZE) bytes / average duration, against 8 TB/s
usage: roofline_table.py kernel_stats.csv pmc_per_kernel.csv"""
import csv
import re
import sys

PEAK_HBM = 8.0e12


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n[:60]


def main():
    ks = {r["name"]: r for r in csv.DictReader(open(sys.argv[1]))}
    pm = {r["kernel"]: r for r in csv.DictReader(open(sys.argv[2]))}
    rows = []
    for name, k in ks.items():
        p = pm.get(name)
        if p is None:
            continue
        avg_s = float(k["avg_ns"]) * 1e-9
        b = float(p["hbm_read_B"]) + float(p["hbm_write_B"])
        gui = float(p["gui"]) / 8.0
        busy = float(p["mfma_busy"]) / (gui * 1024) if gui else 0.0
        rows.append((float(k["total_ns"]), short(name), int(k["calls"]), avg_s * 1e6, busy, b / 1e6, b / avg_s / 1e9 if avg_s else 0.0))
    rows.sort(reverse=True)
    print("%-60s %6s %9s %10s %10s %10s %8s" % ("kernel", "calls", "avg us", "MFMA busy", "HBM MB", "HBM GB/s", "of 8TB/s"))
    for tot, n, c, us, busy, mb, gbs in rows[:40]:
        print("%-60s %6d %9.1f %10.3f %10.2f %10.0f %7.1f%%" % (n, c, us, busy, mb, gbs, 100.0 * gbs * 1e9 / PEAK_HBM))


if __name__ == "__main__":
    main()
