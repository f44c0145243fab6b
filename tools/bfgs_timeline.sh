#!/bin/bash
# kernel timeline of optimize_contrast(optimizer='evk_bfgs'): gaps between the kernels of a pass and between passes, and the
# mean duration of every kernel of the loop.       bash tools/bfgs_timeline.sh [N H W]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf gpurun_out/tl; timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python tools/bfgs_profile.py "$@" > gpurun_out/tl.log 2>&1
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
from collections import defaultdict
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows = [r for r in rows if "evk::" in r[2]]
rows.sort()
# the last optimize call: take the last 200 kernels
rows = rows[-200:]
intra, inter, durs = [], [], []
per = defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gap = (s1 - e0) / 1e3
    if "k_reduce_final" in n0:
        inter.append(gap)
    else:
        intra.append(gap)
    durs.append((e0 - s0) / 1e3)
    per[re.sub(r"^void evk::|\(.*$", "", n0)[:60]].append((e0 - s0) / 1e3)
import statistics as st
print("kernels %d: mean duration %.1f us; gaps inside a pass: median %.1f us (n=%d, mean %.1f); gaps between passes: median %.1f us (n=%d, mean %.1f)" % (
    len(rows), st.mean(durs), st.median(intra), len(intra), st.mean(intra), st.median(inter), len(inter), st.mean(inter)))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print("  %-60s n=%3d mean %.1f us" % (k, len(v), st.mean(v)))
PY
rm -rf gpurun_out/tl
