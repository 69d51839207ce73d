#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes of `bench.py` (WRITE_SIZE and FETCH_SIZE, separate runs) into
profiles/pmc_traffic.json, the `roofline.traffic` figure bench.py reports.

Units and gfx950 corrections follow MI355X_MICROARCH.md (HBM section): the counters are in KiB; WRITE_SIZE reads
exact for 16-byte-per-lane streaming stores; FETCH_SIZE reports half the bytes of a wide coalesced read, so it is
doubled (an upper bound here: this kernel's reads are byte/dword table loads, a width the guide calls uncalibrated --
they are 0.4 % of the traffic either way).

usage: pmc_traffic.py <write counter_collection.csv> <fetch counter_collection.csv> --samples S --tile-rows R --what W
"""
import argparse
import csv
import json
import os
import statistics


def per_launch(path, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if ("rowrun_kernel" in r["Kernel_Name"] or "rowgen_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == counter]
    if not vals:
        raise SystemExit("no %s rows for the row kernel in %s" % (counter, path))
    return statistics.mean(vals), len(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("write_csv")
    ap.add_argument("fetch_csv")
    ap.add_argument("--samples", type=int, default=20000)
    ap.add_argument("--tile-rows", type=int, default=250000)
    ap.add_argument("--what", default="text+packed")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                  "profiles", "pmc_traffic.json"))
    a = ap.parse_args()
    w_kib, nw = per_launch(a.write_csv, "WRITE_SIZE")
    f_kib, nf = per_launch(a.fetch_csv, "FETCH_SIZE")
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    doc = {
        "kernel_source_sha16": bench.kernel_source_hash(),  # bench.py reports the figure only for these very sources
        "samples": a.samples, "tile_rows": a.tile_rows, "what": a.what,
        "write_bytes_per_launch": w_kib * 1024, "fetch_bytes_per_launch_raw": f_kib * 1024,
        "fetch_bytes_per_launch_corrected": 2 * f_kib * 1024,
        "hbm_bytes_per_launch": w_kib * 1024 + 2 * f_kib * 1024,
        "launches": {"WRITE_SIZE": nw, "FETCH_SIZE": nf},
        "method": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE in separate passes of `python3 bench.py --steps 4 "
                  "--warmup 1 --cpu-seconds 0`; KiB -> bytes; FETCH_SIZE doubled (gfx950)",
    }
    json.dump(doc, open(a.out, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
