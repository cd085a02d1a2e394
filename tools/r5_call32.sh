#!/bin/bash
# Run ON THE GPU BOX (round 5, call 32): launches of one or two blocks keep one wavefront per stream -- tests.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
{
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 )
} > $O/r5_call32.log 2>&1
cat $O/r5_call32.log
