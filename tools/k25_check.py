"""k=25 canonical known answers on the synthetic dataset, several table sizes / batch sizes (GPU)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import nomalise_kmers_multi_large_amd as nk
from nomalise_kmers_multi_large_amd import synth

def run(n, batch_pairs, memory_gb, chunk=0):
    blocks = list(synth.pair_blocks(n, seed=1, transcripts=2000))
    a1, a2 = np.concatenate([b[0] for b in blocks]), np.concatenate([b[1] for b in blocks])
    kept = 0
    with nk.KmerNormaliser(25, 100, 0.9, True, memory_gb=memory_gb, max_batch_reads=2 * batch_pairs, max_batch_bases=2 * batch_pairs * 150, chunk_bases=chunk) as g:
        h = torch.zeros(2 * batch_pairs, dtype=torch.int32, device="cuda"); t = torch.zeros_like(h)
        keep = torch.zeros(batch_pairs, dtype=torch.uint8, device="cuda")
        for at in range(0, n, batch_pairs):
            seq, off = synth.interleave(a1[at:at + batch_pairs], a2[at:at + batch_pairs])
            d_seq, d_off = torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
            nr = len(off) - 1
            g.process_device(d_seq.data_ptr(), d_off.data_ptr(), nr, len(seq), h.data_ptr(), t.data_ptr())
            g.decide_device(h.data_ptr(), t.data_ptr(), nr // 2, True, keep.data_ptr())
            g.synchronize()
            kept += int(keep[: nr // 2].sum())
        st = g.stats()
    print("n %d batch %d mem %d chunk %d: kept %d unique %d resizes %d crossing %d/%d" % (n, batch_pairs, memory_gb, chunk, kept, st["unique_kmers"], st["resizes"], st["crossing_kmers"], st["crossing_occurrences"]), flush=True)

run(200000, 200000, 0)        # oracle: kept 147624 unique 7134967
run(200000, 50000, 0)
run(2000000, 1000000, 16)     # oracle: kept 883957 unique 33851135
run(2000000, 1000000, 0)
run(2000000, 250000, 16)
