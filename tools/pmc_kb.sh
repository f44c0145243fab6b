#!/bin/bash
# SQ counters of the voxel kernels (10 M events VGA): where do the waves spend their cycles?
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_kb; rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python tools/pmc_workload.py ${1:-c2} > $OUT/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob("gpurun_out/pmc_kb/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
        if "evk::" not in name: continue
        a = acc[name][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print(k)
    for c, (s, n) in sorted(cs.items()):
        print("   %-24s %.4g per launch" % (c, s / n))
PY
