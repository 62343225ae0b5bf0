"""The single-GPU configurations at their full sizes (GPU box only).

k=15: 10 M synthetic 150 bp pairs, depth 100, five batches of 2 M pairs: every read's (high, total) against the CPU
oracle (about a minute of oracle time on one core), plus the distinct k-mers.

k=25 --canonical: 200 M reads = 100 M pairs made on the GPU; no CPU run of that size fits a test, so the check is the
batch-cut invariance of the result (2 M-pair batches and ragged 0.7 M-pair batches must give identical kept pairs per
block, counter sums and distinct k-mers, while the hashed table grows from its default size to 2^30 slots) and the
totals recorded when the two counting paths were first cross-checked on this stream (22 466 450 kept pairs,
683 618 685 distinct 25-mers)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_k15_config_at_full_size_read_by_read_against_the_oracle(oracle_lib):
    import nomalise_kmers_multi_large_amd as nk
    from nomalise_kmers_multi_large_amd import synth
    PAIRS, BATCH, K, DEPTH, COV = 10000000, 2000000, 15, 100, 0.9
    o = oracle_lib(K, DEPTH, False)
    differing = kept = 0
    with nk.KmerNormaliser(K, DEPTH, COV, False, max_batch_reads=2 * BATCH, max_batch_bases=300 * BATCH) as g:
        for c1, c2 in synth.pair_blocks(PAIRS, seed=1, transcripts=PAIRS // 1000, block=BATCH):
            seq, off = synth.interleave(c1, c2)
            h, t = g.process_batch(seq, off)
            wh, wt = o.batch(seq, off)
            differing += int(np.count_nonzero((h != wh) | (t != wt)))
            rf = np.where(t[0::2] > 0, h[0::2].astype(np.float32) / np.maximum(t[0::2], 1).astype(np.float32), np.float32(0))
            rr = np.where(t[1::2] > 0, h[1::2].astype(np.float32) / np.maximum(t[1::2], 1).astype(np.float32), np.float32(0))
            kept += int(np.count_nonzero((rf < np.float32(COV)) & (rr < np.float32(COV))))
        assert g.stats()["partitioned"] == 1
        assert differing == 0
        assert g.unique_kmers == o.unique == 141458377
    assert kept == 5174728


@pytest.mark.timeout(900)
def test_k25_canonical_config_at_full_size_does_not_depend_on_the_batch_cuts():
    import torch
    from nomalise_kmers_multi_large_amd.normaliser import KmerNormaliser
    from nomalise_kmers_multi_large_amd.synth import GpuGenomeReads
    PAIRS, BLOCK, K = 100000000, 2000000, 25
    dev = torch.device("cuda", 0)
    reads = GpuGenomeReads(device=dev)

    def run(batch_pairs):
        g = KmerNormaliser(K, 100, 0.9, True, max_batch_reads=2 * batch_pairs, max_batch_bases=300 * batch_pairs)
        try:
            g.set_stream(torch.cuda.current_stream().cuda_stream)
            off = torch.arange(0, 2 * batch_pairs + 1, dtype=torch.int64, device=dev) * 150
            high = torch.zeros(2 * batch_pairs, dtype=torch.int32, device=dev)
            total = torch.zeros_like(high)
            keep = torch.zeros(batch_pairs, dtype=torch.uint8, device=dev)
            kept = torch.zeros(PAIRS // BLOCK, dtype=torch.int64, device=dev)
            sums = torch.zeros(2, dtype=torch.int64, device=dev)
            for i in range(PAIRS // BLOCK):
                seq = reads.block(i, BLOCK)
                for at in range(0, BLOCK, batch_pairs):
                    n = min(batch_pairs, BLOCK - at)
                    g.process_device(seq[at * 300:].data_ptr(), off.data_ptr(), 2 * n, 300 * n, high.data_ptr(), total.data_ptr())
                    g.decide_device(high.data_ptr(), total.data_ptr(), n, True, keep.data_ptr())
                    kept[i] += keep[:n].sum(dtype=torch.int64)
                    sums[0] += high[: 2 * n].sum(dtype=torch.int64)
                    sums[1] += total[: 2 * n].sum(dtype=torch.int64)
            st = g.stats()
            return kept.cpu().tolist(), sums.cpu().tolist(), st
        finally:
            g.close()

    a = run(2000000)
    b = run(700000)
    assert a[0] == b[0], "kept pairs per block differ between the two batchings"
    assert a[1] == b[1]
    assert a[2]["unique_kmers"] == b[2]["unique_kmers"] == 683618685
    assert sum(a[0]) == 22466450
    assert a[2]["resizes"] >= 3 and a[2]["capacity"] >= 1 << 30
