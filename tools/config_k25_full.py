"""The k=25 --canonical configuration at its full size (200 M reads = 100 M pairs of 150 bp) on one GPU, checked through
properties that do not need a CPU run of that size: the stream is processed twice, in batches of 2 M pairs and in ragged
batches of 0.7 M pairs, and kept pairs per block, total counters and distinct k-mers must be identical; the hashed table
has to grow several times on the way.  Reads are made on the GPU (random transcriptome, paired ends of random fragments,
0.1 % substitutions, 0.05 % N), block by block from a seed, so both runs see the same stream.
python tools/config_k25_full.py [pairs] [k]   ->  one JSON line"""
import json, sys, time
import torch
sys.path.insert(0, ".")
from nomalise_kmers_multi_large_amd.normaliser import KmerNormaliser

PAIRS = int(sys.argv[1]) if len(sys.argv) > 1 else 100000000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 25
BLOCK = 2000000
dev = torch.device("cuda", 0)
from nomalise_kmers_multi_large_amd.synth import GpuGenomeReads
READS = GpuGenomeReads(device=dev)
block = READS.block


def run(batch_pairs):
    nk = KmerNormaliser(K, 100, 0.9, True, max_batch_reads=2 * batch_pairs, max_batch_bases=300 * batch_pairs)
    nk.set_stream(torch.cuda.current_stream().cuda_stream)
    off = torch.arange(0, 2 * batch_pairs + 1, dtype=torch.int64, device=dev) * 150
    high = torch.zeros(2 * batch_pairs, dtype=torch.int32, device=dev)
    total = torch.zeros_like(high)
    keep = torch.zeros(batch_pairs, dtype=torch.uint8, device=dev)
    kept, sums, busy = [], [0, 0], 0.0
    for i in range(PAIRS // BLOCK):
        seq = block(i, BLOCK)
        torch.cuda.synchronize()
        t0 = time.time()
        k_block = 0
        for at in range(0, BLOCK, batch_pairs):
            n = min(batch_pairs, BLOCK - at)
            nk.process_device(seq[at * 300:].data_ptr(), off.data_ptr(), 2 * n, 300 * n, high.data_ptr(), total.data_ptr())
            nk.decide_device(high.data_ptr(), total.data_ptr(), n, True, keep.data_ptr())
            k_block += int(keep[:n].sum())
            sums[0] += int(high[: 2 * n].sum(dtype=torch.int64))
            sums[1] += int(total[: 2 * n].sum(dtype=torch.int64))
        busy += time.time() - t0
        kept.append(k_block)
        if i % 10 == 9:
            print("batch %d pairs: block %d, kept so far %d, %.1f M pairs/s" % (batch_pairs, i, sum(kept), (i + 1) * BLOCK / busy / 1e6), file=sys.stderr, flush=True)
    st = nk.stats()
    nk.close()
    return kept, sums, st, busy


a = run(2000000)
b = run(700000)
same = a[0] == b[0] and a[1] == b[1] and a[2]["unique_kmers"] == b[2]["unique_kmers"]
print(json.dumps({"config": "%d synthetic 150 bp pairs, k=%d --canonical depth=100 coverage=0.9, one MI355X, hashed table from the default size" % (PAIRS, K),
                  "identical_in_2M_and_0.7M_batches": same, "kept_pairs": sum(a[0]), "high_sum": a[1][0], "total_sum": a[1][1],
                  "unique_kmers": a[2]["unique_kmers"], "table_slots": a[2]["capacity"], "table_bytes": a[2]["table_bytes"], "resizes": a[2]["resizes"],
                  "pairs_per_s_2M_batches": PAIRS / a[3], "pairs_per_s_0.7M_batches": PAIRS / b[3]}))
sys.exit(0 if same else 1)
