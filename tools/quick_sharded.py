"""Throw-away timing of the sharded pipeline's stages on one GPU (world size 1: the all-to-all is a local copy):
python tools/quick_sharded.py [n_pairs] [batch_pairs]"""
import os, sys, time
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, ".")
from nomalise_kmers_multi_large_amd import synth
from nomalise_kmers_multi_large_amd.sharded import HipShardEngine, ShardedNormaliser

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
bp = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
K = int(sys.argv[3]) if len(sys.argv) > 3 else 15
off = torch.arange(0, 2 * bp + 1, dtype=torch.int64, device="cuda") * 150
batches = [(s, off, 2 * m, int(s.numel())) for s, m in synth.gpu_pair_steps(n, bp, seed=1, transcripts=max(50, n // 1000), device="cuda")]
if os.environ.get("MODE") == "bucketed":   # the bucket-ordered exchange (k <= 15)
    from nomalise_kmers_multi_large_amd.bucketed import HipBucketEngine, BucketedNormaliser
    eng = HipBucketEngine(K, 100, 0.9, False, 0, 2 * bp, 2 * bp * 150, 1, 0, profile=(os.environ.get("PROF", "1") == "1"))
    eng.ranked = "-"
    sn = BucketedNormaliser(eng, time_phases=True)
else:
    eng = HipShardEngine(K, 100, 0.9, K > 16, 0, max_reads=2 * bp, max_bases=2 * bp * 150, profile=(os.environ.get("PROF", "1") == "1"))
    sn = ShardedNormaliser(eng, time_phases=True)
keep = torch.zeros(bp, dtype=torch.uint8, device="cuda")
AHEAD = os.environ.get("AHEAD", "1") == "1"   # send the next step's keys while this step's are counted
for i, (s, o, nr, nb) in enumerate(batches):
    torch.cuda.synchronize(); t = time.time()
    high, total, gathered = sn.process_step(s, o, nr, nb, then=batches[i + 1] if AHEAD and i + 1 < len(batches) else None)
    eng.decide(gathered[0][0], gathered[0][1], nr // 2, True, keep)
    sn.synchronize(); torch.cuda.synchronize()
    dt = time.time() - t
    print("step %d: %d pairs %.2f ms -> %.2f Mpairs/s kept %d; counted so far %d, total sum %d" % (i, nr // 2, dt * 1e3, nr / 2 / dt / 1e6, int(keep[: nr // 2].sum()), eng.stats()["kmers_counted"], int(total.sum())), flush=True)
print(eng.stats(), 'ranked', eng.ranked, 'stay_home', eng.stay_home, 'home positions', eng.home_total)
import json
print(json.dumps(sn.phase_report(), indent=1))
dist.destroy_process_group()
