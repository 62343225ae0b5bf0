#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> <python script and args...>
# kernel-trace profile of one python run; summary goes to gpurun_out/<tag>_stats.txt, program output to gpurun_out/<tag>.log
tag=$1; shift
export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag -- python3 "$@" > gpurun_out/$tag.log 2>&1 || { tail -20 gpurun_out/$tag.log; exit 1; }
python3 tools/rocpd_summary.py $(find /tmp/prof_$tag -name "*.db" | head -1) > gpurun_out/${tag}_stats.txt 2>&1
grep -v "rocprofv3\|^W2026\|^E2026" gpurun_out/$tag.log | tail -${TAIL:-16}
head -${HEAD:-28} gpurun_out/${tag}_stats.txt
