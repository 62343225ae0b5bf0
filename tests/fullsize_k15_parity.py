"""The k=15 configuration at its full size (10 M synthetic 150 bp pairs, depth 100) against the CPU oracle, read by read
(a checker run for the GPU box, not collected by pytest: the oracle takes minutes on one core):
every batch of 2 M pairs goes through the product (device path, partitioned counting) and through oracle/nkml_oracle.so,
per-read high and total counters and the distinct k-mers must be identical.
python tests/fullsize_k15_parity.py [pairs] [batch_pairs]   ->  one JSON line"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import nomalise_kmers_multi_large_amd as nk
from nomalise_kmers_multi_large_amd import synth
from tests.helpers import Oracle

PAIRS = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
K, DEPTH, COV = 15, 100, 0.9
o = Oracle(K, DEPTH, False)
t_gpu = t_cpu = 0.0
mismatched_reads = kept = 0
with nk.KmerNormaliser(K, DEPTH, COV, False, max_batch_reads=2 * BATCH, max_batch_bases=300 * BATCH) as g:
    for i, (c1, c2) in enumerate(synth.pair_blocks(PAIRS, seed=1, transcripts=max(50, PAIRS // 1000), block=BATCH)):
        seq, off = synth.interleave(c1, c2)
        t = time.time()
        h, tot = g.process_batch(seq, off)
        t_gpu += time.time() - t
        t = time.time()
        wh, wt = o.batch(seq, off)
        t_cpu += time.time() - t
        bad = int(np.count_nonzero((h != wh) | (tot != wt)))
        mismatched_reads += bad
        rf = np.where(tot[0::2] > 0, h[0::2].astype(np.float32) / np.maximum(tot[0::2], 1).astype(np.float32), np.float32(0))
        rr = np.where(tot[1::2] > 0, h[1::2].astype(np.float32) / np.maximum(tot[1::2], 1).astype(np.float32), np.float32(0))
        kept += int(np.count_nonzero((rf < np.float32(COV)) & (rr < np.float32(COV))))
        print("batch %d: %d reads, %d differ from the oracle (oracle %.0f s so far)" % (i, len(h), bad, t_cpu), flush=True)
    unique = g.unique_kmers
    part = g.stats().get("partitioned")
print(json.dumps({"config": "%d synthetic 150 bp pairs, k=15 depth=100 coverage=0.9, batches of %d pairs, one MI355X vs oracle/nkml_oracle.so on one host core" % (PAIRS, BATCH),
                  "reads": 2 * PAIRS, "reads_differing_from_oracle": mismatched_reads, "kept_pairs": kept, "unique_kmers_gpu": int(unique),
                  "unique_kmers_oracle": int(o.unique), "partitioned": part, "oracle_seconds": round(t_cpu, 1), "gpu_seconds_incl_pcie": round(t_gpu, 2)}))
