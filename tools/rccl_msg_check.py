"""Which collectives deliver a message of more than 1 GiB whole (RCCL, world size 1), and what a cut message looks like:
python tools/rccl_msg_check.py"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))


def describe(a, b):
    bad = (a != b)
    n_bad = int(bad.sum())
    if n_bad == 0:
        return "ok"
    idx = torch.nonzero(bad.view(-1)[:: max(1, bad.numel() // (1 << 26))]).view(-1)
    first = int(torch.nonzero(bad.view(-1)[: 1 << 30] if bad.numel() > (1 << 30) else bad.view(-1))[0]) if n_bad else -1
    tail_zero = int((b.view(-1)[first:] == 0).all())
    return "CUT: %d of %d elements differ, first at element %d (byte %d), everything from there on is zero: %s" % (
        n_bad, a.numel(), first, first * a.element_size(), bool(tail_zero))


for dtype in (torch.int32, torch.uint8, torch.int64):
    for mib in (1024, 1025, 2049, 3000):
        n = (mib << 20) // torch.empty(0, dtype=dtype).element_size()
        a = (torch.arange(n, device="cuda", dtype=torch.int64) * 2654435761 % 251 + 1).to(dtype)
        for name in ("all_to_all_single", "all_gather_into_tensor", "all_reduce", "broadcast"):
            b = torch.zeros_like(a)
            if name == "all_to_all_single":
                dist.all_to_all_single(b, a, [n], [n])
            elif name == "all_gather_into_tensor":
                dist.all_gather_into_tensor(b, a)
            elif name == "all_reduce":
                b.copy_(a)
                dist.all_reduce(b)
            else:
                b.copy_(a)
                dist.broadcast(b, 0)
            torch.cuda.synchronize()
            print("%-12s %5d MiB %-24s %s" % (str(dtype), mib, name, describe(a, b)), flush=True)
            del b
        del a
dist.destroy_process_group()
