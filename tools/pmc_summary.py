"""Summarise the rocprofv3 --pmc passes of tools/profile_round.sh into per-kernel HBM bytes per launch.
  python tools/pmc_summary.py <dir holding pmc_fetch_<tag>/ and pmc_write_<tag>/ for tag in c2, c5_share>
hbm bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half of a wide coalesced
streaming read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as reported.  Both are in KiB.  FETCH_SIZE and
WRITE_SIZE come from SEPARATE passes (they do not fit one pass, same section)."""
import csv
import glob
import json
import os
import re
import sys

ALG = {"c2": 16 * 10_000_000 + 5 * 480 * 640 * 4, "c5_share": 16 * 50_000_000 + 5 * 720 * 1280 * 4,
       "img_nearest": 12 * 10_000_000 + 480 * 640 * 4, "img_bilinear": 12 * 10_000_000 + 480 * 640 * 4,
       "prebucketed": 16 * 10_000_000 + 5 * 480 * 640 * 4, "img_timestamp": 16 * 10_000_000 + 4 * 481 * 641 * 4}


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
    root = sys.argv[1]
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/pmc_workload.py <tag> "
                   "(8 calls of the voxel / event-image path alone); hbm bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: "
                   "FETCH_SIZE reports half of a wide coalesced streaming read, MI355X_MICROARCH.md HBM section; WRITE_SIZE "
                   "as reported).  whole_call_bytes = sum over the kernels of one call; algorithmic_bytes = 16 B/event + "
                   "the grid, 12 B/event + the image for img_* (SURVEY.md 8(d)).  Summarised by tools/pmc_summary.py."}
    for tag in ("c2", "c5_share", "img_nearest", "img_bilinear", "img_timestamp", "prebucketed"):
        if not os.path.isdir(os.path.join(root, "pmc_fetch_" + tag)):
            continue
        fetch = collect(os.path.join(root, "pmc_fetch_" + tag), "FETCH_SIZE")
        write = collect(os.path.join(root, "pmc_write_" + tag), "WRITE_SIZE")
        ks, total = {}, 0
        for name in sorted(set(fetch) | set(write)):
            if "evk::" not in name:
                continue
            f, w = fetch.get(name, [0.0, 0]), write.get(name, [0.0, 0])
            fa, wa = (f[0] / f[1] if f[1] else 0.0), (w[0] / w[1] if w[1] else 0.0)
            b = int(round((2 * fa + wa) * 1024))
            ks[name] = {"FETCH_SIZE_KB_avg": round(fa, 1), "WRITE_SIZE_KB_avg": round(wa, 1), "calls": max(f[1], w[1]),
                        "hbm_bytes_per_launch_corrected": b}
            total += b
        if tag == "prebucketed":   # the bucketing ran once, outside the timed kernel: the call is k_voxel_tiled alone
            ks = {k: v for k, v in ks.items() if "k_voxel_tiled" in k}
            total = sum(v["hbm_bytes_per_launch_corrected"] for v in ks.values())
        out[tag] = {"kernels": ks, "whole_call_bytes": total, "algorithmic_bytes": ALG[tag],
                    "amplification": round(total / ALG[tag], 3) if total else None}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
