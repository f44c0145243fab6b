#!/bin/bash
# LDS / VALU counters of the tiled IWE kernels at C4 (50 M events, 1280x720, moving-edge scene; tools/c4_modes.py runs
# every accumulator mode: k_iwe_tiled<MODE, FIXED, COMPACT>, FIXED 1 = 64-bit cells, 2 = packed pairs) ->
# gpurun_out/prof/r02_c4_iwe_lds_counters.json.  Counter passes carry no trace options; the durations come from a
# separate --kernel-trace --stats pass.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
R=${ROUND:-r02}; OUT=gpurun_out/pmc_iwe; rm -rf $OUT; mkdir -p $OUT gpurun_out/prof
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python tools/c4_modes.py > $OUT/p$i.log 2>&1
done
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python tools/c4_modes.py > $OUT/st.log 2>&1
python - "$R" <<'PY'
import csv, glob, json, re, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob("gpurun_out/pmc_iwe/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
        if "k_iwe_tiled" not in name: continue
        a = acc[name][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
dur = {}
for path in glob.glob("gpurun_out/pmc_iwe/st/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        name = re.sub(r"\(.*", "", row["Name"]).replace("void ", "").strip()
        if "k_iwe_tiled" in name: dur[name] = float(row["AverageNs"]) / 1e3
out = {"note": "rocprofv3 --pmc (two counter passes, no trace options) over tools/c4_modes.py: 50 M events, 1280x720, moving-edge "
               "scene, balanced plan.  k_iwe_tiled<MODE, FIXED, COMPACT>: MODE 0 IWE, 1 IWE + dIWE, 2 three flows; FIXED 1 "
               "64-bit fixed-point cells, 2 packed 32-bit pairs (returning atomics).  SQ_* summed over the chip, per launch; "
               "lds_active_us_per_cu = SQ_LDS_IDX_ACTIVE / 256 CUs / 2.4 GHz; avg_us from a separate --kernel-trace --stats pass.",
       "kernels": {}}
for k, cs in sorted(acc.items()):
    d = {c: s / n for c, (s, n) in sorted(cs.items())}
    if "SQ_LDS_IDX_ACTIVE" in d: d["lds_active_us_per_cu"] = d["SQ_LDS_IDX_ACTIVE"] / 256 / 2400.0
    if "SQ_LDS_BANK_CONFLICT" in d and d.get("SQ_LDS_IDX_ACTIVE"): d["bank_conflict_share_of_lds_cycles"] = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]
    if k in dur: d["avg_us"] = dur[k]
    out["kernels"][k] = d
json.dump(out, open("gpurun_out/prof/%s_c4_iwe_lds_counters.json" % sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT
