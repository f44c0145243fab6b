#!/bin/bash
# Round-end profile refresh on the GPU box (run through gpurun): kernel-trace stats and the two PMC passes of the
# default bench command, summarised into gpurun_out/ (copy the results into profiles/ afterwards).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 20 --no-cpu > $OUT/stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 5 --no-cpu --no-cmax > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps 5 --no-cpu --no-cmax > $OUT/pmc_write.log 2>&1
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/stats
tail -c 600 $OUT/bench.json; echo; head -c 1500 $OUT/pmc_traffic.json
