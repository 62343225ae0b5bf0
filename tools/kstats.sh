#!/bin/bash
# usage (GPU box, repo root): tools/kstats.sh <tag> "<quick_bench args>" [lib.so]  - per-kernel durations of one quick bench run
tag=$1; args=$2; lib=${3:-}
export TMPDIR=/tmp
[ -n "$lib" ] && export NKML_LIBRARY=$lib
rm -rf /tmp/ks_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$tag -o $tag -- python3 tools/quick_bench.py $args > gpurun_out/ks_${tag}.log 2>&1 || exit 1
python3 tools/rocpd_summary.py $(find /tmp/ks_$tag -name "*.db" | head -1) | grep "kernel \|k_pt_\|k_hp_\|k_pack\|k_mark\|k_reduce\|k_decide\|k_count" > gpurun_out/ks_${tag}.txt
cat gpurun_out/ks_${tag}.txt | cut -c1-60,95-
