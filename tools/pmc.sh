#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc.sh <tag> "<counters>" <python script and args...>
# one counter pass (no tracing besides the kernel trace); per-kernel averages go to gpurun_out/<tag>_pmc.txt
tag=$1; shift
ctrs=$1; shift
export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
timeout -s KILL 240 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc_$tag -o $tag -- python3 "$@" > gpurun_out/${tag}_pmc.log 2>&1 || { tail -20 gpurun_out/${tag}_pmc.log; exit 1; }
python3 tools/pmc_any.py $(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1) > gpurun_out/${tag}_pmc.txt 2>&1
cat gpurun_out/${tag}_pmc.txt
