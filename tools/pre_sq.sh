#!/bin/bash
# SQ counters of the pre-bucketed voxel kernel (k_voxel_tiled, bench.py -> prebucketed): 10 M events 640x480
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_pre; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python tools/pmc_workload.py prebucketed 60 > $OUT/p$i.log 2>&1
done
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python tools/pmc_workload.py prebucketed 60 > $OUT/st.log 2>&1
python3 - <<'PY'
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob("gpurun_out/pmc_pre/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
        if "k_voxel_tiled" not in name: continue
        a = acc[name][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
dur = {}
for path in glob.glob("gpurun_out/pmc_pre/st/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        name = re.sub(r"\(.*", "", row["Name"]).replace("void ", "").strip()
        if "k_voxel_tiled" in name: dur[name] = float(row["AverageNs"]) / 1e3
for k, cs in sorted(acc.items()):
    d = {c: s / n for c, (s, n) in sorted(cs.items())}
    us = dur.get(k, 0)
    cap = 1024 * us * 1e-6 * 2.1e9 / 4 if us else 0      # wave-instruction slots of 1024 SIMDs at 2.1 GHz
    print(k, "avg %.1f us" % us)
    print("   VALU %.3g (%.0f %% of the SIMDs' slots)  SALU %.3g  LDS insts %.3g  LDS busy %.1f us/CU (conflicts %.0f %%)" % (
        d.get("SQ_INSTS_VALU", 0), 100 * d.get("SQ_INSTS_VALU", 0) / cap if cap else 0, d.get("SQ_INSTS_SALU", 0), d.get("SQ_INSTS_LDS", 0),
        d.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / 2100.0, 100 * d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
rm -rf $OUT
