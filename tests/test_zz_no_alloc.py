"""nkml_route_keys_static (the entry point that allocates nothing: its scratch is a fixed array in device memory)
against nkml_route_keys_device, through the C ABI, on two calls in a row - the second with a smaller input, so that
whatever the first call left in the shared scratch array must not matter."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _route(g, fn, keys_t, n_shards, torch):
    n = keys_t.numel()
    send_keys = torch.full((n,), 0x5EADBEEF if keys_t.dtype == torch.int32 else 0x0DEADBEEFDEADBEE, dtype=keys_t.dtype, device="cuda")
    send_src = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    counts = fn(keys_t.data_ptr(), n, n_shards, send_keys.data_ptr(), send_src.data_ptr())
    g.synchronize()
    torch.cuda.synchronize()
    total = sum(counts)
    return counts, send_keys[:total].cpu().numpy(), send_src[:total].cpu().numpy()


@pytest.mark.parametrize("k,n_shards", [(15, 4), (25, 3)])
def test_route_keys_static_matches_allocating_entry_point(k, n_shards):
    import torch
    import nomalise_kmers_multi_large_amd as nk
    rng = np.random.default_rng(7)
    with nk.KmerNormaliser(k, 10, 0.9, False, device=0, max_batch_reads=1 << 16, max_batch_bases=1 << 22) as g:
        wide = g.key_bytes == 8
        dt_np, dt_t = (np.int64, torch.int64) if wide else (np.int32, torch.int32)
        hi = (1 << (2 * k)) - 1 if wide else (1 << 30) - 1
        for n in (1 << 20, 5000 * 16):  # the second call is smaller than the first
            keys = rng.integers(0, hi, n, dtype=np.int64)
            keys[rng.random(n) < 0.1] = 0  # positions without a k-mer are not sent
            keys_t = torch.from_numpy(keys.astype(dt_np)).cuda()
            want = _route(g, g.route_keys_device, keys_t, n_shards, torch)
            got = _route(g, g.route_keys_static, keys_t, n_shards, torch)
            assert got[0] == want[0]
            assert sum(got[0]) == int((keys != 0).sum())
            assert np.array_equal(got[1], want[1])
            assert np.array_equal(got[2], want[2])
            # and both are the stable partition by owner
            owners = np.array([g.lib.nkml_owner_of(ctypes.c_uint64(int(x)), n_shards) for x in keys[:2000]])
            sent_first = keys[:2000][keys[:2000] != 0]
            o_first = owners[keys[:2000] != 0]
            pos = np.nonzero(keys[:2000])[0]
            start = 0
            for s in range(n_shards):
                seg_src = got[2][start:start + got[0][s]]
                head = seg_src[seg_src < 2000]
                assert np.array_equal(head, pos[o_first == s])
                start += got[0][s]


def test_route_keys_static_rejects_oversize():
    import torch
    import nomalise_kmers_multi_large_amd as nk
    with nk.KmerNormaliser(15, 10, 0.9, False, device=0, max_batch_reads=1024, max_batch_bases=1 << 20) as g:
        dummy = torch.zeros(16, dtype=torch.int32, device="cuda")
        # (2^19 + 1) tiles of 1024 keys for one shard: one entry more than the fixed scratch holds; rejected before any
        # device work, so the pointers are never touched
        n_keys = ((1 << 19) + 1) * 1024
        with pytest.raises(nk.NkmlError) as e:
            g.route_keys_static(dummy.data_ptr(), n_keys, 1, dummy.data_ptr(), dummy.data_ptr())
        from nomalise_kmers_multi_large_amd import binding as B
        assert e.value.code == B.NKML_E_ARG and "scratch" in str(e.value)


def test_route_keys_static_allocates_no_device_memory():
    """The property the entry point is named for: on a fresh context (no routing buffers yet) a call leaves the device's
    free memory exactly where it was, while the allocating entry point, called next with the same input, takes some."""
    import torch
    import nomalise_kmers_multi_large_amd as nk
    rng = np.random.default_rng(11)
    n, n_shards = 1 << 26, 4   # 65536 tiles x 4 shards: the allocating variant needs a few MiB of scratch for this
    with nk.KmerNormaliser(15, 10, 0.9, False, device=0, max_batch_reads=1024, max_batch_bases=1 << 20) as g:
        keys_t = torch.from_numpy(rng.integers(0, (1 << 30) - 1, n, dtype=np.int64).astype(np.int32)).cuda()
        send_keys = torch.empty(n, dtype=torch.int32, device="cuda")
        send_src = torch.empty(n, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        c_static = g.route_keys_static(keys_t.data_ptr(), n, n_shards, send_keys.data_ptr(), send_src.data_ptr())
        g.synchronize()
        free1 = torch.cuda.mem_get_info()[0]
        assert free1 == free0, "nkml_route_keys_static changed the device's free memory by %d bytes" % (free0 - free1)
        c_alloc = g.route_keys_device(keys_t.data_ptr(), n, n_shards, send_keys.data_ptr(), send_src.data_ptr())
        g.synchronize()
        free2 = torch.cuda.mem_get_info()[0]
        assert c_alloc == c_static
        assert free2 < free1, "the probe saw no allocation by nkml_route_keys_device either: it proves nothing"
