#!/bin/bash
# rocprofv3 kernel-trace stats of a command, evk kernels only:  bash tools/kstats.sh <tag> <command ...>
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
tag=$1; shift
rm -rf gpurun_out/ks_$tag
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$tag -- "$@" > gpurun_out/ks_$tag.log 2>&1
f=$(find gpurun_out/ks_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "evk" in n or "k_" in n:
        print("%-70s calls %5s avg %9.1f us  min %9.1f" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
rm -rf gpurun_out/ks_$tag
