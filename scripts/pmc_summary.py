#!/usr/bin/env python3
"""Per-kernel means of the rocprofv3 --pmc CSV passes written by scripts/pmc_check.sh.

  python scripts/pmc_summary.py gpurun_out/pmc_TAG [--csv out.csv]

HBM bytes per launch follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes for wide coalesced reads, so the read side is
doubled (an upper bound for narrow accesses); WRITE_SIZE is exact for 16-byte-per-lane stores and float atomics."""
import argparse
import collections
import csv
import glob
import os
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--csv", default=None)
    ap.add_argument("--top", type=int, default=30)
    args = ap.parse_args()
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(args.dir, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name") or row.get("Kernel Name")
                c = row.get("Counter_Name")
                v = float(row.get("Counter_Value") or 0)
                a = acc[k][c]
                a[0] += v
                a[1] += 1
    rows = []
    for k, cs in acc.items():
        m = {c: a[0] / max(a[1], 1) for c, a in cs.items()}
        n = max(a[1] for a in cs.values())
        rd = 2.0 * m.get("FETCH_SIZE", 0.0) * 1024
        wr = m.get("WRITE_SIZE", 0.0) * 1024
        wave = m.get("SQ_WAVE_CYCLES", 0.0) * 4  # quad-cycles
        rows.append(dict(kernel=k, launches=n, hbm_read_B=rd, hbm_write_B=wr, mfma_busy=m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0),
                         busy=m.get("SQ_BUSY_CYCLES", 0.0), gui=m.get("GRBM_GUI_ACTIVE", 0.0), wave_cycles=wave,
                         wait_any=m.get("SQ_WAIT_ANY", 0.0) * 4, wait_inst=m.get("SQ_WAIT_INST_ANY", 0.0) * 4,
                         active_inst=m.get("SQ_ACTIVE_INST_ANY", 0.0) * 4, lds_conf=m.get("SQ_LDS_BANK_CONFLICT", 0.0),
                         lds_active=m.get("SQ_LDS_IDX_ACTIVE", 0.0)))
    rows.sort(key=lambda r: -(r["gui"] * r["launches"]))
    print("%-90s %5s %10s %10s %9s %9s %8s %8s" % ("kernel (per-launch means)", "n", "HBM rd MB", "HBM wr MB", "kclk", "mfma busy",
                                                     "wait%", "ldsconf%"))
    for r in rows[:args.top]:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs (checked against the kernel-trace durations);
        # SQ_VALU_MFMA_BUSY_CYCLES over the chip's 1024 SIMDs: MFMA-pipe busy share = busy / (gui / 8 * 1024)
        util = r["mfma_busy"] / (r["gui"] / 8.0 * 1024) if r["gui"] else 0.0
        print("%-90s %5d %10.2f %10.2f %9.1f %9.3f %8.1f %8.1f" % (
            short(r["kernel"]), r["launches"], r["hbm_read_B"] / 1e6, r["hbm_write_B"] / 1e6, r["gui"] / 8e3, util,
            100.0 * r["wait_any"] / r["wave_cycles"] if r["wave_cycles"] else 0.0,
            100.0 * r["lds_conf"] / r["lds_active"] if r["lds_active"] else 0.0))
    if args.csv:
        with open(args.csv, "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)


if __name__ == "__main__":
    main()
