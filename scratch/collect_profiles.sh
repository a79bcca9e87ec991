#!/bin/bash
# Round evidence in one GPU call: usage  collect_profiles.sh <tag>   (from the repo root on the GPU box, TMPDIR=/tmp)
# Writes gpurun_out/prof_<tag>/: PMC traffic of the dominant kernel, bench line (carrying that traffic), breakdowns,
# rocprofv3 kernel stats (two-lane = the default command, and single-lane), other configurations, secondary workloads.
set -e
TAG=$1; OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=1   # what bench.py sets itself; under rocprofv3 the runtime is up before the script runs
# 1. HBM traffic of the dominant kernel: separate --pmc passes, nothing else traced
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmcF -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmcF.log 2>&1
echo "fetch pass done"
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmcW -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmcW.log 2>&1
echo "write pass done"
python3 scratch/collect_traffic.py $OUT/pmcF $OUT/pmcW $OUT/pmc_traffic.json
cp $OUT/pmc_traffic.json profiles/${TAG}_pmc_traffic_conv_igemm_b256_bf16.json   # bench.py replays the newest one
rm -rf $OUT/pmcF $OUT/pmcW
# 2. the bench line (default command, CPU baseline included) + two-lane breakdown
python3 bench.py --steps 20 --warmup 3 --breakdown $OUT/breakdown_two_lanes.txt > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"; tail -c 600 $OUT/bench.json; echo
# 3. per-shape and per-class tables, lanes serialised
python3 scratch/prof_detail.py 256 > $OUT/per_shape_single_lane.txt 2>/dev/null
python3 bench.py --single-lane --steps 4 --warmup 2 --no-cpu-baseline --breakdown $OUT/breakdown_single_lane.txt > $OUT/bench_single_lane.json 2> /dev/null
echo "detail done"
# 4. rocprofv3 kernel stats: default command, and single-lane (what roofline.achieved is measured on)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats.csv
rm -rf $OUT/kt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kts -o kt -- python3 bench.py --single-lane --steps 2 --warmup 1 --no-cpu-baseline > $OUT/kts.log 2>&1
cp $(find $OUT/kts -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_single_lane.csv
rm -rf $OUT/kts
echo "kernel stats done"
# 5. other configurations of the same step + the secondary workloads, one JSON line each (no CPU baseline)
: > $OUT/bench_variants.jsonl
for extra in "--dtype f16" "--dtype f32 --steps 4 --warmup 1" "--batch 512 --steps 6 --warmup 2" "--batch 128" \
             "--img 448 --batch 128 --dtype f16 --steps 6 --warmup 2" "--metadata-dim 20" "--target-momentum 0.99" \
             "--workload linear_probe" "--workload inference" "--workload mlc_train"; do
  python3 bench.py --no-cpu-baseline $extra 2>/dev/null >> $OUT/bench_variants.jsonl || echo "{\"failed\": \"$extra\"}" >> $OUT/bench_variants.jsonl
done
cut -c1-230 $OUT/bench_variants.jsonl
ls -la $OUT
