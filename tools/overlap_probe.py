"""Throw-away: does the chip have room left beside one context's kernels?  Two contexts (own tables, own streams) fed by two
host threads against one context doing the same total work.  python tools/overlap_probe.py [k] [batches] [batch_pairs]"""
import sys, time, threading
import torch
sys.path.insert(0, ".")
import nomalise_kmers_multi_large_amd as nk
from nomalise_kmers_multi_large_amd import synth

k = int(sys.argv[1]) if len(sys.argv) > 1 else 25
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 6
bp = int(sys.argv[3]) if len(sys.argv) > 3 else 2000000
dev = torch.device("cuda", 0)
steps = synth.gpu_pair_steps(nb * bp, bp, seed=1, transcripts=max(50, nb * bp // 1000), device=dev)
off = torch.arange(0, 2 * bp + 1, dtype=torch.int64, device=dev) * 150
batches = [(s, off, 2 * m, int(s.numel())) for s, m in steps]
torch.cuda.synchronize()


def worker(g, outs, rounds):
    d_h, d_t = outs
    for _ in range(rounds):
        for s, o, nr, nbases in batches:
            g.process_device(s.data_ptr(), o.data_ptr(), nr, nbases, d_h.data_ptr(), d_t.data_ptr())
    g.synchronize()


def run(n_ctx, rounds):
    gs = [nk.KmerNormaliser(k, 100, 0.9, canonical=(k > 16), max_batch_reads=2 * bp, max_batch_bases=2 * bp * 150) for _ in range(n_ctx)]
    outs = [(torch.zeros(2 * bp, dtype=torch.int32, device=dev), torch.zeros(2 * bp, dtype=torch.int32, device=dev)) for _ in range(n_ctx)]
    for g, o in zip(gs, outs):
        worker(g, o, 1)   # warm: the table has its k-mers, later rounds find them
    torch.cuda.synchronize()
    t0 = time.time()
    th = [threading.Thread(target=worker, args=(g, o, rounds)) for g, o in zip(gs, outs)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.time() - t0
    for g in gs:
        g.close()
    return n_ctx * rounds * nb * bp / dt / 1e6


print("one context : %.1f M pairs/s" % run(1, 4), flush=True)
print("two contexts: %.1f M pairs/s in sum" % run(2, 2), flush=True)
print("one context : %.1f M pairs/s" % run(1, 4), flush=True)
