#!/bin/bash
# usage (GPU box, repo root): tools/ab.sh "<quick_bench args>" lib1.so lib2.so ...   - the same quick bench with each library, two rounds, interleaved
args=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib (round $round)"
    NKML_LIBRARY=$lib timeout -k 10 200 python3 tools/quick_bench.py $args 2>&1 | grep -v "^generated\|amdgpu.ids" | tail -${TAILN:-4} || exit 1
  done
done
