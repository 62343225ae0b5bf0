#!/bin/bash
# usage (GPU box, repo root): tools/exchange_world1.sh <out file>  - one rank's step of the multi-GPU exchange modes with a world of one
# (nothing crosses a link: what is timed is the work a rank does per 2 M-pair step besides waiting for its peers), k=15
out=$1
{
echo "# tools/exchange_world1.sh on one MI355X: python tools/quick_sharded.py 16000000 2000000 15 per mode (world of one, no look-ahead);"
echo "# per step: host clock around the whole step; phases: HIP events on the launch stream (ms per step, averaged over 8 steps)"
for m in "exchange 0 0" "exchange 1 0" "bucketed 1 0" "bucketed 1 1"; do
  set -- $m
  echo
  echo "== mode $1, ranked routing $2, saturated k-mers stay home $3"
  MODE=$1 NKML_EXCHANGE_RANKED=$2 NKML_EXCHANGE_STAY_HOME=$3 AHEAD=0 timeout -k 10 300 python3 -u tools/quick_sharded.py 16000000 2000000 15 2>&1 | grep "^step\|ranked\|\"" | cut -c1-330 || exit 1
done
} > $out
