#!/bin/bash
# usage (GPU box, repo root): tools/e2e_cli.sh <n_pairs> [copies]  - file-to-file rate of the drop-in command line on synthetic FASTQ
# (n_pairs generated once, the files then repeated `copies` times: page-cached input of n_pairs x copies pairs)
n=${1:-1000000}
copies=${2:-1}
d=/tmp/nkml_e2e; rm -rf $d; mkdir -p $d
python3 - <<PY
import sys, time
sys.path.insert(0, ".")
from nomalise_kmers_multi_large_amd import synth
t = time.time(); synth.write_fastq("$d/one", $n, seed=1, transcripts=max(50, $n // 1000)); print("wrote $n pairs in %.1f s" % (time.time() - t))
PY
for i in $(seq $copies); do cat $d/one_1.fastq >> $d/syn_1.fastq; cat $d/one_2.fastq >> $d/syn_2.fastq; done
rm $d/one_1.fastq $d/one_2.fastq
total=$((n * copies))
ls -la $d | tail -2
cd $d
for rep in 1 2; do
  export NKML_TIMING=1; t0=$(date +%s.%N)
  $GRAFT_REPO_ROOT/nomalise_kmers_multi_large_amd/nkml_amd -f syn_1.fastq -r syn_2.fastq ${NKML_E2E_ARGS:--k 15 -d 100 -g 0.9} > run$rep.log 2> err$rep.log; grep "host phases" err$rep.log
  t1=$(date +%s.%N)
  grep -i "rate\|Printed Records\|Skipped Records\|unique\|runtime" run$rep.log | tail -8
  python3 -c "print('nkml_amd wall %.2f s -> %.2f M pairs/s file to file' % ($t1 - $t0, $total / ($t1 - $t0) / 1e6))"
done
md5sum output_forward*.fastq | head -2
