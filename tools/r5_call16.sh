#!/bin/bash
# Run ON THE GPU BOX (round 5, call 16): the whole GPU test suite on the final kernels, the content sweep, export timing.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
( time AECM_SANITIZER_LOG=$PWD/$O/r5_ubsan_gpu.log python -m pytest tests -m gpu -x -q --durations=8 ) > $O/r5_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r5_pytest.log
( python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "65536_streams_is_fast" 2>&1 | grep "65 536 states" ) > $O/r5_export_timing.txt 2>&1
bash tools/content_sweep.sh > $O/r5_content_sweep.txt 2>&1
tail -14 $O/r5_pytest.log; cat $O/r5_export_timing.txt $O/r5_content_sweep.txt
