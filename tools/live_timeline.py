"""Timeline of the last calls in a rocprofv3 kernel-trace CSV: start / end of every kernel relative to the call's partition."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows), key=lambda e: e[0])
parts = [i for i, e in enumerate(ev) if "k_part_sorted" in e[2]]
for i in parts[-9:]:
    t0 = ev[i][0]
    line = []
    for e in ev[i: i + 4]:
        if e[0] - t0 > 400000:
            break
        nm = "part" if "part_sorted" in e[2] else ("live" if "voxel_live" in e[2] else ("tiles" if "voxel_tiles2" in e[2] else e[2][:20]))
        line.append("%s %.1f..%.1f" % (nm, (e[0] - t0) / 1e3, (e[1] - t0) / 1e3))
    print(" | ".join(line))
