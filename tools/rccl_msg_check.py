"""Largest all_to_all_single message that arrives whole (RCCL, world size 1): python tools/rccl_msg_check.py"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for dtype in (torch.int32, torch.uint8):
    for mib in (256, 512, 768, 1000, 1023, 1024, 1025, 1100, 2047, 2049):
        n = (mib << 20) // torch.empty(0, dtype=dtype).element_size()
        a = (torch.arange(n, device="cuda", dtype=torch.int64) * 2654435761 % 251).to(dtype)
        b = torch.zeros_like(a)
        dist.all_to_all_single(b, a, [n], [n])
        torch.cuda.synchronize()
        bad = int((a != b).sum())
        print(dtype, "%d MiB: %s (%d elements differ)" % (mib, "ok" if bad == 0 else "CUT", bad), flush=True)
        del a, b
dist.destroy_process_group()
