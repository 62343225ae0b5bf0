#!/bin/bash
# usage (on the GPU box, from the repo root): tools/round_profiles.sh <tag> [bench.py arguments...]
# The profile set of one bench configuration, every pass a run of its own:
#   gpurun_out/<tag>_bench.json         the bench line without a profiler attached
#   gpurun_out/<tag>_kernel_stats.txt   rocprofv3 --kernel-trace --stats of the same command (per-kernel durations)
#   gpurun_out/<tag>_fetch_pmc.txt      rocprofv3 --pmc FETCH_SIZE   (per-kernel averages, KiB per call)
#   gpurun_out/<tag>_write_pmc.txt      rocprofv3 --pmc WRITE_SIZE
tag=$1; shift
export TMPDIR=/tmp
set -e
timeout -k 10 400 python3 bench.py "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 1500 gpurun_out/${tag}_bench.json
rm -rf /tmp/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag -- python3 bench.py "$@" --no-cpu-baseline > gpurun_out/${tag}_prof.log 2>&1
python3 tools/rocpd_summary.py $(find /tmp/prof_$tag -name "*.db" | head -1) > gpurun_out/${tag}_kernel_stats.txt
head -24 gpurun_out/${tag}_kernel_stats.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
    low=$(echo $ctr | cut -d_ -f1 | tr A-Z a-z)
    rm -rf /tmp/pmc_${tag}_$low
    timeout -s KILL 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$low -o $tag -- python3 bench.py "$@" --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/${tag}_${low}_pmc.log 2>&1
    python3 tools/pmc_any.py $(find /tmp/pmc_${tag}_$low -name "*counter_collection.csv" | head -1) > gpurun_out/${tag}_${low}_pmc.txt
done
cat gpurun_out/${tag}_fetch_pmc.txt gpurun_out/${tag}_write_pmc.txt | grep "kernel \|k_pt_\|k_hp_\|k_pack\|k_mark\|k_reduce\|k_decide\|k_count"
