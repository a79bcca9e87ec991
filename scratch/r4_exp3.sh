#!/bin/bash
set -o pipefail
O=gpurun_out/r4e3; mkdir -p $O
SM3_LIBRARY=scratch/_stamp/libsm3hip_stamp.so timeout -k 10 300 python scratch/stamp_phases.py > $O/stamp_phases.txt 2>&1; echo "stamp rc=$?" | tee -a $O/summary.txt
cat $O/stamp_phases.txt
timeout -k 10 900 python -m pytest tests/test_round4_gpu.py tests/test_p2p_gpu.py "tests/test_dp_gpu.py::test_two_rank_dp_bf16_batchnorm_by_linearity_syncs_like_the_two_pass_form" tests/test_round3_gpu.py -q -s -m gpu -k "config4 or p2p or missing or exchange or linear or fp16 or 16bit or forms" > $O/new_tests.log 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|PASS|FAIL|Error|config 4|448x448|224x224|B=16|B=4|mailbox|rank [01]:" $O/new_tests.log | tail -40
timeout -k 10 600 python scratch/t2_stream.py > $O/t2_stream.txt 2>&1; echo "t2 rc=$?" | tee -a $O/summary.txt
cat $O/t2_stream.txt | tail -12
