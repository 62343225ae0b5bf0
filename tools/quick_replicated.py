"""Throw-away timing of the replicated-reads mode's per-rank work on one GPU: this process plays rank 0 of WORLD
ranks; the other ranks' packed reads are made locally instead of arriving by all-gather, so the time is the compute a
real rank does per step (pack, counting against its shard on WORLD x the reads, per-read reduction) without the links.
python tools/quick_replicated.py [world] [steps] [pairs_per_rank_step]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from nomalise_kmers_multi_large_amd import synth
from nomalise_kmers_multi_large_amd import replicated as R

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bp = int(sys.argv[3]) if len(sys.argv) > 3 else 2000000
torch.cuda.set_device(0)
eng = R.HipReplicaEngine(15, 100, 0.9, False, 0, max_reads=2 * bp, max_bases=2 * bp * 150, world=world, rank=0, profile=True)
data = []
for r in range(world):
    blocks = []
    for c1, c2 in synth.pair_blocks(steps * bp, seed=1 + 1000 * r, transcripts=max(50, steps * bp // 1000), block=bp):
        seq, off = synth.interleave(c1, c2)
        blocks.append((torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), len(off) - 1, len(seq)))
    data.append(blocks)
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.time()
    reads = [data[r][i][2] for r in range(world)]
    bases = [data[r][i][3] for r in range(world)]
    region, fillers, first, total_reads = R.step_layout(reads, bases)
    words_all = eng.gather_buffer(world * region)
    for r in range(world):   # rank 0's pack is real work of the step; the others' stand in for the all-gather
        words_all[r * region:(r + 1) * region].copy_(eng.pack(data[r][i][0], bases[r], region))
    torch.cuda.synchronize(); t1 = time.time()
    goff = R.global_offsets([data[r][i][1] for r in range(world)], reads, bases, region, fillers)
    torch.cuda.synchronize(); t2 = time.time()
    counters = eng.process_packed(words_all, goff, total_reads, world * region * 16)
    eng.synchronize(); torch.cuda.synchronize(); t3 = time.time()
    print("step %d: pack x%d %.2f ms, offsets %.2f ms, count+reduce %.2f ms; shard total %d of %d positions" % (
        i, world, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, int(counters[1].sum()), world * region * 16), flush=True)
st = eng.stats()
print({k: st[k] for k in ("count_stage_ms", "count_kernel_ms", "count_launches", "unique_kmers", "partitioned") if k in st})
