#!/bin/bash
# Round evidence in one GPU call: usage  collect_profiles.sh <tag>   (from the repo root on the GPU box)
# Writes gpurun_out/prof_<tag>/: bench line, breakdowns, rocprofv3 kernel stats, PMC traffic of the dominant kernel.
set -e
TAG=$1; OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 3 --breakdown $OUT/breakdown_two_lanes.txt > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"; tail -c 600 $OUT/bench.json; echo
python3 scratch/prof_detail.py 256 > $OUT/per_shape_single_lane.txt 2>/dev/null
echo "detail done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats.csv
echo "kernel stats done"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmcF -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmcF.log 2>&1
echo "fetch pass done"
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmcW -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmcW.log 2>&1
echo "write pass done"
python3 scratch/collect_traffic.py $OUT/pmcF $OUT/pmcW $OUT/pmc_traffic.json
rm -rf $OUT/kt $OUT/pmcF $OUT/pmcW
ls -la $OUT
python3 bench.py --single-lane --steps 4 --warmup 2 --no-cpu-baseline --breakdown $OUT/breakdown_single_lane.txt > $OUT/bench_single_lane.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kts -o kt -- python3 bench.py --single-lane --steps 2 --warmup 1 --no-cpu-baseline > $OUT/kts.log 2>&1
cp $(find $OUT/kts -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_single_lane.csv
rm -rf $OUT/kts
ls -la $OUT
