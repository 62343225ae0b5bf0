"""Does a single hipMemsetAsync clear buffers of 2/4/8 GiB?  (torch used for allocation and checking only)"""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
for gib in (2, 4, 6, 8):
    n = gib << 30
    a = torch.full((n,), 0xAB, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    rc = hip.hipMemsetAsync(ctypes.c_void_p(a.data_ptr()), 0, ctypes.c_size_t(n), ctypes.c_void_p(0))
    torch.cuda.synchronize()
    left = int((a != 0).sum())
    print("%d GiB: rc %d, bytes not cleared %d" % (gib, rc, left), flush=True)
    del a
