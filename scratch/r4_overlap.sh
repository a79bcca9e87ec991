#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4overlap; mkdir -p $O
SM3_BENCH_REGION_EVENTS=none rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-dtypes > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
ls -la $f
python3 scratch/trace_overlap.py $f | tee $O/two_lane_overlap.txt
rm -rf $O/kt
