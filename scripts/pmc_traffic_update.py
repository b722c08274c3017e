#!/usr/bin/env python3
"""Rewrite profiles/pmc_traffic.json (what bench.py's roofline.traffic reads) from a pmc_per_kernel.csv of
scripts/pmc_summary.py.  Only kernels that run ONE launch geometry in the profiled step get a record (the csv holds
per-kernel means): the stem kernels and layer1's halo kernel (its 4 forward + 4 data-gradient launches share one
geometry and one traffic pattern).

  python scripts/pmc_traffic_update.py profiles/r03_v5_pmc_per_kernel.csv"""
import csv
import json
import os
import sys

GEOM = {  # kernel-name fragment -> (records, launch geometry string as bench.py's key prints it)
    "stem_fwd_kernel": (("stem_fwd",), "64x16x112x112x64"),
    "stem_wgrad_kernel": (("stem_wgrad",), "64x16x112x112x64"),
    "halo_conv_kernel<4, 32, true, 7>": (("fwd", "dgrad"), "64x64x8x28x28x64x3x3x3x1x1x1"),
}


def main():
    src = sys.argv[1]
    out = {"_source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, scripts/pmc_check.sh; read side doubled per "
                      "MI355X_MICROARCH.md HBM section); each record holds for ONE launch geometry (\"launch\"): bench.py "
                      "reports it only when its dominant launch is that one" % src}
    for row in csv.DictReader(open(src)):
        for frag, (recs, launch) in GEOM.items():
            if frag in row["kernel"]:
                rd, wr = float(row["hbm_read_B"]), float(row["hbm_write_B"])
                for r in recs:
                    out[r] = {"kernel": frag, "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
                              "launch": launch, "launches_averaged": int(row["launches"])}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote", path, sorted(k for k in out if k != "_source"))


if __name__ == "__main__":
    main()
