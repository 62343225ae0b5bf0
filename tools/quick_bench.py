"""Throw-away timing of the device path on synthetic pairs: python tools/quick_bench.py [n_pairs] [batch_pairs] [k]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import nomalise_kmers_multi_large_amd as nk
from nomalise_kmers_multi_large_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
bp = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 15
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
t0 = time.time()
batches = []
for c1, c2 in synth.pair_blocks(n, seed=1, transcripts=max(50, n // 1000), block=bp):
    seq, off = synth.interleave(c1, c2)
    batches.append((torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), len(off) - 1, len(seq)))
print("generated %d pairs in %.1fs" % (n, time.time() - t0), flush=True)
g = nk.KmerNormaliser(k, 100, 0.9, canonical=(k > 16), max_batch_reads=2 * bp, max_batch_bases=2 * bp * 150, profile=True, chunk_bases=chunk)
d_h = torch.zeros(2 * bp, dtype=torch.int32, device="cuda"); d_t = torch.zeros_like(d_h)
keep = torch.zeros(bp, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
kept = 0
T0 = time.time()
for i, (s, o, nr, nb) in enumerate(batches):
    t = time.time()
    g.process_device(s.data_ptr(), o.data_ptr(), nr, nb, d_h.data_ptr(), d_t.data_ptr())
    g.decide_device(d_h.data_ptr(), d_t.data_ptr(), nr // 2, True, keep.data_ptr())
    g.synchronize()
    dt = time.time() - t
    kept += int(keep[: nr // 2].sum())
    print("batch %d: %d pairs %.2f ms -> %.2f Mpairs/s" % (i, nr // 2, dt * 1e3, nr / 2 / dt / 1e6), flush=True)
T = time.time() - T0
st = g.stats()
print("total %.1f ms, %.2f Mpairs/s, kept %d of %d" % (T * 1e3, n / T / 1e6, kept, n))
print(st)
print("count kernel: %.2f ms over %d launches, %.2f Gkeys/s" % (st["count_kernel_ms"], st["count_kernel_launches"], st["count_kernel_keys"] / st["count_kernel_ms"] / 1e6))
