"""Throw-away timing of the routing stage of the key-exchange mode on one GPU for several shard counts:
python tools/quick_route.py [pairs] [shard counts, comma separated]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from nomalise_kmers_multi_large_amd import synth
from nomalise_kmers_multi_large_amd.sharded import HipShardEngine

bp = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
c1, c2 = next(synth.pair_blocks(bp, seed=1, transcripts=2000, block=bp))
seq, off = synth.interleave(c1, c2)
d_seq, d_off = torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
eng = HipShardEngine(15, 100, 0.9, False, 0, max_reads=2 * bp, max_bases=2 * bp * 150)
for n in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]:
    for rep in range(3):
        torch.cuda.synchronize(); t = time.time()
        keys, src, counts, npos = eng.kmerise_route(d_seq, d_off, len(off) - 1, len(seq), n)
        torch.cuda.synchronize(); dt = time.time() - t
    print("shards %d: route %.2f ms, counts %s" % (n, dt * 1e3, counts), flush=True)
