"""ctypes face of libnkml_hip.so (the C ABI declared in include/nkml_hip.h).

There is no CPU fallback: if the shared library has not been built (make -C nomalise_kmers_multi_large_amd/csrc,
or __graft_entry__.build()) loading fails loudly, and nkml_create fails with NKML_E_NODEVICE without a gfx950 GPU."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NKML_OK = 0
NKML_E_ARG, NKML_E_HIP, NKML_E_NOMEM, NKML_E_BADBASE, NKML_E_STATE, NKML_E_NODEVICE = -1, -2, -3, -4, -5, -6
NKML_F_PROFILE = 1
NKML_F_EXACT_COUNTS = 2
NKML_F_COUNT_ATOMIC = 4
NKML_F_COUNT_PARTITIONED = 8
NKML_F_VERIFY_RANKS = 16
NKML_F_NO_TABLE = 32
FLAG_NONE, FLAG_HIGH, FLAG_LOW = 0, 1, 2

# every symbol include/nkml_hip.h declares
EXPORTS = [
    "nkml_abi_version", "nkml_strerror", "nkml_last_error", "nkml_create", "nkml_destroy", "nkml_reset", "nkml_host_staging", "nkml_wait_staged",
    "nkml_set_stream", "nkml_synchronize", "nkml_get_stats", "nkml_submit", "nkml_collect", "nkml_process_device",
    "nkml_seed", "nkml_keep_pair", "nkml_keep_single", "nkml_decide_device", "nkml_key_bytes", "nkml_kmerise_device",
    "nkml_count_keys_device", "nkml_seed_keys_device", "nkml_reduce_reads_device", "nkml_dump_table", "nkml_owner_of", "nkml_route_keys_device", "nkml_route_keys_static", "nkml_route_reads_device",
    "nkml_unroute_flags_device", "nkml_route_reads_ranked", "nkml_unroute_ranked", "nkml_sat_enable", "nkml_sat_collect", "nkml_sat_apply", "nkml_xb_enable", "nkml_xb_front", "nkml_xb_count", "nkml_xb_finish", "nkml_set_shard", "nkml_shard_of", "nkml_pack_device", "nkml_process_packed_device",
    "nkml_multi_create", "nkml_multi_destroy", "nkml_multi_last_error", "nkml_multi_device_count", "nkml_multi_reset",
    "nkml_multi_host_staging", "nkml_multi_wait_staged", "nkml_multi_submit", "nkml_multi_collect", "nkml_multi_seed",
    "nkml_multi_get_stats", "nkml_multi_dump_table",
]


class NkmlError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("nkml error %d: %s" % (code, message))
        self.code = code


class Config(ctypes.Structure):
    _fields_ = [("ksize", ctypes.c_int32), ("depth", ctypes.c_int32), ("canonical", ctypes.c_int32),
                ("device", ctypes.c_int32), ("memory_gb", ctypes.c_int32), ("max_batch_reads", ctypes.c_uint32),
                ("max_batch_bases", ctypes.c_uint64), ("chunk_bases", ctypes.c_uint32), ("flags", ctypes.c_uint32)]


class Stats(ctypes.Structure):
    _fields_ = [("unique_kmers", ctypes.c_uint64), ("capacity", ctypes.c_uint64), ("table_bytes", ctypes.c_uint64),
                ("kmers_counted", ctypes.c_uint64), ("kmers_saturated", ctypes.c_uint64),
                ("crossing_kmers", ctypes.c_uint64), ("crossing_occurrences", ctypes.c_uint64),
                ("resizes", ctypes.c_uint64), ("count_kernel_ms", ctypes.c_double),
                ("count_kernel_launches", ctypes.c_uint64), ("count_kernel_keys", ctypes.c_uint64),
                ("count_stage_ms", ctypes.c_double), ("partitioned", ctypes.c_uint64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


def library_path():
    # NKML_LIBRARY: another build of the same library (kernel experiments); the in-tree one otherwise
    return os.environ.get("NKML_LIBRARY") or os.path.join(_HERE, "libnkml_hip.so")


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    # PyTorch-ROCm ships its own HIP runtime; when torch is going to be used in this process it has to be the
    # one that initialises the GPU, so it is imported before the library pulls in libamdhip64.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise ImportError("%s is missing: build it with `make -C nomalise_kmers_multi_large_amd/csrc` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    lib = ctypes.CDLL(path)
    vp, i32, u32, u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint64
    lib.nkml_abi_version.restype = ctypes.c_int
    lib.nkml_strerror.restype = ctypes.c_char_p
    lib.nkml_strerror.argtypes = [ctypes.c_int]
    lib.nkml_last_error.restype = ctypes.c_char_p
    lib.nkml_last_error.argtypes = [vp]
    lib.nkml_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(vp)]
    lib.nkml_destroy.argtypes = [vp]
    lib.nkml_destroy.restype = None
    lib.nkml_reset.argtypes = [vp]
    lib.nkml_set_stream.argtypes = [vp, vp]
    lib.nkml_synchronize.argtypes = [vp]
    lib.nkml_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    lib.nkml_submit.argtypes = [vp, vp, vp, u32]
    lib.nkml_collect.argtypes = [vp, vp, vp]
    lib.nkml_process_device.argtypes = [vp, vp, vp, u32, u64, vp, vp]
    lib.nkml_seed.argtypes = [vp, vp, vp, u32]
    lib.nkml_keep_pair.argtypes = [i32, i32, i32, i32, ctypes.c_float]
    lib.nkml_keep_single.argtypes = [i32, i32, ctypes.c_float]
    lib.nkml_decide_device.argtypes = [vp, vp, vp, u32, ctypes.c_int, ctypes.c_float, vp]
    lib.nkml_key_bytes.argtypes = [vp]
    lib.nkml_kmerise_device.argtypes = [vp, vp, vp, u32, u64, vp]
    lib.nkml_count_keys_device.argtypes = [vp, vp, u64, vp]
    lib.nkml_seed_keys_device.argtypes = [vp, vp, u64]
    lib.nkml_reduce_reads_device.argtypes = [vp, vp, vp, vp, u32, vp, vp]
    lib.nkml_dump_table.argtypes = [vp, vp, vp, u64, ctypes.POINTER(u64)]
    lib.nkml_owner_of.argtypes = [u64, u32]
    lib.nkml_route_keys_device.argtypes = [vp, vp, u64, u32, vp, vp, ctypes.POINTER(u64)]
    lib.nkml_route_keys_static.argtypes = [vp, vp, u64, u32, vp, vp, ctypes.POINTER(u64)]
    lib.nkml_route_reads_device.argtypes = [vp, vp, vp, u32, u64, u32, vp, vp, ctypes.POINTER(u64)]
    lib.nkml_unroute_flags_device.argtypes = [vp, vp, vp, u64, vp, u64]
    lib.nkml_route_reads_ranked.argtypes = [vp, vp, vp, u32, u64, u32, vp, vp, vp, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    lib.nkml_unroute_ranked.argtypes = [vp, vp, vp, vp, u64, vp]
    lib.nkml_sat_enable.argtypes = [vp]
    lib.nkml_sat_collect.argtypes = [vp, vp, u64, ctypes.POINTER(u64)]
    lib.nkml_sat_apply.argtypes = [vp, vp, u64]
    lib.nkml_xb_enable.argtypes = [vp, u32, u32, u64]
    lib.nkml_xb_front.argtypes = [vp, vp, vp, u32, u64, vp, vp, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    lib.nkml_xb_count.argtypes = [vp, vp, vp, u64, vp]
    lib.nkml_xb_finish.argtypes = [vp, vp, vp, u32, vp, vp]
    lib.nkml_set_shard.argtypes = [vp, u32, u32]
    lib.nkml_shard_of.argtypes = [u64, u32]
    lib.nkml_pack_device.argtypes = [vp, vp, u64, vp]
    lib.nkml_process_packed_device.argtypes = [vp, vp, vp, u32, u64, vp, vp]
    lib.nkml_multi_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(i32), u32, ctypes.POINTER(vp)]
    lib.nkml_multi_destroy.argtypes = [vp]
    lib.nkml_multi_destroy.restype = None
    lib.nkml_multi_last_error.argtypes = [vp]
    lib.nkml_multi_last_error.restype = ctypes.c_char_p
    lib.nkml_multi_device_count.argtypes = [vp]
    lib.nkml_multi_device_count.restype = u32
    lib.nkml_multi_reset.argtypes = [vp]
    lib.nkml_multi_host_staging.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp)]
    lib.nkml_multi_wait_staged.argtypes = [vp]
    lib.nkml_multi_submit.argtypes = [vp, vp, vp, u32]
    lib.nkml_multi_collect.argtypes = [vp, vp, vp]
    lib.nkml_multi_seed.argtypes = [vp, vp, vp, u32]
    lib.nkml_multi_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    lib.nkml_multi_dump_table.argtypes = [vp, vp, vp, u64, ctypes.POINTER(u64)]
    _LIB = lib
    return lib
