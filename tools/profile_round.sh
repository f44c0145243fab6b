#!/bin/bash
# Round-end profile refresh on the GPU box (run through gpurun): the bench line, rocprofv3 kernel-trace stats of the same
# command, and the PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, MI355X_MICROARCH.md) of the two voxel
# workloads, summarised into gpurun_out/prof (copy the results into profiles/ afterwards).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=${ROUND:-r06}
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
t0=$SECONDS
timeout -s KILL 500 python bench.py > $OUT/${R}_bench.json 2> $OUT/bench.err
echo "python bench.py (defaults, cpu_baseline included): $((SECONDS - t0)) s wall" > $OUT/${R}_bench_wall.txt
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 20 --no-cpu > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/${R}_bench_kernel_stats.csv \;
rm -rf $OUT/stats
for tag in c2 c5_share img_nearest img_bilinear img_timestamp prebucketed; do
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$tag -- python tools/pmc_workload.py $tag > $OUT/pmc_fetch_$tag.log 2>&1
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$tag -- python tools/pmc_workload.py $tag > $OUT/pmc_write_$tag.log 2>&1
  # (kernel durations at steady clocks: a few hundred calls -- the first milliseconds of a burst run 15-20 % slower)
  calls=400; [ $tag = c5_share ] && calls=150
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$tag -- python tools/pmc_workload.py $tag $calls > $OUT/st_$tag.log 2>&1
  find $OUT/st_$tag -name "*kernel_stats.csv" -exec cp {} $OUT/${R}_${tag}_kernel_stats.csv \;
done
python tools/pmc_summary.py $OUT > $OUT/${R}_pmc_traffic.json
rm -rf $OUT/pmc_fetch_* $OUT/pmc_write_* $OUT/st_*
tail -c 400 $OUT/${R}_bench.json; echo; cat $OUT/${R}_pmc_traffic.json | head -60
