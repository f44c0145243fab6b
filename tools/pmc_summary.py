"""Summarise rocprofv3 --pmc counter_collection CSVs into per-kernel HBM bytes per launch.
  python tools/pmc_summary.py <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass> > profiles/rNN_pmc_traffic.json
hbm bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half of a wide coalesced
streaming read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as reported.  Both are in KiB."""
import csv
import glob
import json
import os
import re
import sys


def collect(d, counter):
    acc = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
                s = acc.setdefault(name, [0.0, 0])
                s[0] += float(row["Counter_Value"])
                s[1] += 1
    return acc


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 5 --no-cpu "
                   "--no-cmax`; hbm bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reports half "
                   "of a wide coalesced streaming read, MI355X_MICROARCH.md HBM section; WRITE_SIZE as reported). "
                   "Summarised by tools/pmc_summary.py.",
           "kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        if "evk::" not in name:
            continue
        f, w = fetch.get(name, [0.0, 0]), write.get(name, [0.0, 0])
        fa, wa = (f[0] / f[1] if f[1] else 0.0), (w[0] / w[1] if w[1] else 0.0)
        out["kernels"][name] = {"FETCH_SIZE_KB_avg": round(fa, 1), "WRITE_SIZE_KB_avg": round(wa, 1),
                                "calls": max(f[1], w[1]),
                                "hbm_bytes_per_launch_corrected": int(round((2 * fa + wa) * 1024))}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
