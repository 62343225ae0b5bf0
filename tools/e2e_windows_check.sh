#!/bin/bash
# the same input through one window and through many: identical output bytes (GPU box, repo root)
d=/tmp/nkml_e2e_w; rm -rf $d; mkdir -p $d
python3 - <<PY
import sys
sys.path.insert(0, ".")
from nomalise_kmers_multi_large_amd import synth
synth.write_fastq("$d/syn", 2000000, seed=1, transcripts=2000)
PY
cd $d
for br in 4194304 700000 123456; do
  NKML_BATCH_READS=$br $GRAFT_REPO_ROOT/nomalise_kmers_multi_large_amd/nkml_amd -f syn_1.fastq -r syn_2.fastq -k 15 -d 100 -g 0.9 > run.log 2>&1
  echo "batch_reads $br: $(grep 'Printed Records' run.log) $(md5sum < output_forward.k15_norm100_thread0.fastq | cut -c1-32) $(md5sum < output_reverse.k15_norm100_thread0.fastq | cut -c1-32)"
done
