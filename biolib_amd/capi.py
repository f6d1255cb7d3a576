"""ctypes binding of the C ABI in include/biolib_amd.h (biolib_amd/lib/libbiolib_amd.so).

There is no CPU fallback: if the HIP library is missing or no gfx950 device is visible,
loading / context creation raises.  (Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C biolib_amd/csrc`.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BIOLIB_AMD_LIB", os.path.join(_HERE, "lib", "libbiolib_amd.so"))  # override: A/B builds only

BL_OK, BL_ERR_INVALID, BL_ERR_HIP, BL_ERR_OOM, BL_ERR_CAPACITY, BL_ERR_NO_DEVICE, BL_ERR_INTERNAL = 0, -1, -2, -3, -4, -5, -6
FLAG_CANONICAL, FLAG_DROP_LAST, FLAG_SYNC = 1, 2, 4

# every symbol include/biolib_amd.h declares (tests check the library exports exactly these)
SYMBOLS = [
    "bl_last_error", "bl_version", "bl_device_count", "bl_ctx_create", "bl_ctx_destroy", "bl_ctx_set_stream", "bl_ctx_use_own_streams", "bl_ctx_sync",
    "bl_batch_upload", "bl_batch_upload_reads", "bl_batch_from_device", "bl_batch_synth", "bl_batch_destroy", "bl_batch_n_bases", "bl_batch_n_seqs",
    "bl_batch_device_bases", "bl_batch_download", "bl_batch_set_origin", "bl_batch_origin", "bl_scan_kmers", "bl_scan_minimizers", "bl_scan_hash_sample", "bl_scan_super_kmers", "bl_scan_super_kmer_records", "bl_scan_syncmers", "bl_sort_unique_u64", "bl_jaccard_sorted_u64", "bl_partition_u64", "bl_sort_u64", "bl_count_sorted_u64", "bl_probe_hbm", "bl_clock_probe_start", "bl_clock_probe_finish", "bl_ctx_set_lanes", "bl_pack_super_kmers", "bl_count_super_kmers", "bl_partition_records", "bl_expand_super_kmers",
    "bl_ctx_last_scan_ms", "bl_ctx_set_exact_windows", "bl_ctx_set_option", "bl_ctx_kernel_timing", "bl_ctx_kernel_time", "bl_ctx_mark", "bl_ctx_mark_times", "bl_reader_open", "bl_reader_open_threads", "bl_reader_kind", "bl_reader_next_text", "bl_reader_next_batch_device", "bl_reader_close", "bl_reader_next_record",
    "bl_reader_next_batch", "bl_reader_last_batch", "bl_reader_last_name", "bl_batch_from_text", "bl_run_file_name", "bl_write_run_u64", "bl_write_vector_u64", "bl_file_count_u64", "bl_read_file_u64_host", "bl_read_file_u64", "bl_merge_runs_u64", "bl_count_allreduce",
    "bl_device_alloc", "bl_device_free", "bl_copy_to_host", "bl_copy_to_device", "bl_hash64_u64", "bl_bgzf_walk", "bl_bgzf_inflate", "bl_host_alloc", "bl_host_free", "bl_reader_open_shard", "bl_reader_shard_range",
]


class BiolibError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"biolib_amd error {code}: {msg}")
        self.code = code


class Result(C.Structure):
    _fields_ = [("count", C.c_uint64), ("xor_value", C.c_uint64), ("xor_hash", C.c_uint64), ("xor_pos", C.c_uint64),
                ("aux", C.c_uint64), ("status", C.c_int32), ("redone", C.c_int32)]

    def as_dict(self):
        return dict(count=int(self.count), xor_value=int(self.xor_value), xor_hash=int(self.xor_hash), xor_pos=int(self.xor_pos),
                    aux=int(self.aux), status=int(self.status))


_lib = None


def lib():
    """Load the HIP library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build the HIP extension first (__graft_entry__.build()); "
                          "biolib_amd has no CPU fallback")
    # torch's ROCm wheel bundles its own libamdhip64/libhsa-runtime64 (same SONAMEs as /opt/rocm's).
    # Import torch first so that ONE HIP runtime serves both torch (device memory, streams, RCCL) and
    # this library; loading ours first brings in a second runtime and device enumeration then fails.
    import torch  # noqa: F401

    L = C.CDLL(LIB_PATH)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    L.bl_last_error.restype = C.c_char_p
    L.bl_version.restype = C.c_int
    L.bl_device_count.argtypes = [C.POINTER(C.c_int)]
    L.bl_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.bl_ctx_destroy.argtypes = [vp]
    L.bl_ctx_set_stream.argtypes = [vp, vp]
    L.bl_ctx_use_own_streams.argtypes = [vp]
    L.bl_ctx_sync.argtypes = [vp]
    L.bl_batch_upload.argtypes = [vp, vp, u64, vp, u64, C.POINTER(vp)]
    L.bl_batch_from_device.argtypes = [vp, vp, u64, vp, u64, u64, C.POINTER(vp)]
    L.bl_batch_synth.argtypes = [vp, u64, u64, u64, C.POINTER(vp)]
    L.bl_batch_destroy.argtypes = [vp]
    L.bl_batch_n_bases.restype = u64
    L.bl_batch_n_bases.argtypes = [vp]
    L.bl_batch_n_seqs.restype = u64
    L.bl_batch_n_seqs.argtypes = [vp]
    L.bl_batch_device_bases.restype = vp
    L.bl_batch_device_bases.argtypes = [vp]
    L.bl_batch_download.argtypes = [vp, u64, u64, vp]
    L.bl_batch_set_origin.argtypes = [vp, u64]
    if "BIOLIB_AMD_LIB" not in os.environ or hasattr(L, "bl_batch_origin"):  # (an A/B build of an older revision may lack it)
        L.bl_batch_origin.argtypes = [vp]
        L.bl_batch_origin.restype = u64
    L.bl_batch_upload_reads.argtypes = [vp, vp, u64, u64, C.POINTER(vp)]
    L.bl_scan_kmers.argtypes = [vp, vp, u64, u64, u32, u64, u32, vp, vp, vp, C.POINTER(Result)]
    L.bl_scan_minimizers.argtypes = [vp, vp, u64, u64, u32, u32, u64, u32, vp, vp, vp, u64, C.POINTER(Result)]
    L.bl_scan_hash_sample.argtypes = [vp, vp, u64, u64, u32, u64, u64, u32, vp, vp, vp, u64, C.POINTER(Result)]
    L.bl_sort_unique_u64.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.bl_jaccard_sorted_u64.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64), C.POINTER(u64)]
    L.bl_pack_super_kmers.argtypes = [vp, vp, vp, vp, vp, u64, u32, u32, vp]
    L.bl_count_super_kmers.argtypes = [vp, vp, u64, u32, u32, u64, u32, vp, vp, u64, C.POINTER(u64)]
    L.bl_partition_records.argtypes = [vp, vp, vp, u64, u32, vp, vp]
    L.bl_expand_super_kmers.argtypes = [vp, vp, u64, u32, u32, vp, u64, C.POINTER(u64)]
    L.bl_ctx_set_lanes.argtypes = [vp, C.c_int]
    L.bl_probe_hbm.argtypes = [vp, u64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.bl_clock_probe_start.argtypes = [vp, u32, C.POINTER(vp)]
    L.bl_clock_probe_finish.argtypes = [vp, C.POINTER(C.c_double)]
    L.bl_partition_u64.argtypes = [vp, vp, u64, u32, u64, vp, vp]
    L.bl_sort_u64.argtypes = [vp, vp, u64]
    L.bl_count_sorted_u64.argtypes = [vp, vp, u64, vp, vp, C.POINTER(u64)]
    L.bl_scan_super_kmers.argtypes = [vp, vp, u64, u64, u32, u32, u64, u32, vp, vp, vp, vp, vp, u64, C.POINTER(Result)]
    L.bl_scan_super_kmer_records.argtypes = [vp, vp, u64, u64, u32, u32, u64, u32, vp, vp, u64, C.POINTER(Result)]
    L.bl_scan_syncmers.argtypes = [vp, vp, u64, u64, u32, u32, u32, u32, u64, u32, vp, u64, C.POINTER(Result)]
    L.bl_ctx_last_scan_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.bl_ctx_set_exact_windows.argtypes = [vp, C.c_int]
    if "BIOLIB_AMD_LIB" not in os.environ or hasattr(L, "bl_ctx_set_option"):
        L.bl_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.bl_ctx_kernel_timing.argtypes = [vp, C.c_int]
    L.bl_ctx_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u64)]
    L.bl_ctx_mark.argtypes = [vp]
    L.bl_ctx_mark_times.argtypes = [vp, C.POINTER(C.c_double), u32, C.POINTER(u32)]
    L.bl_reader_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.bl_reader_close.argtypes = [vp]
    L.bl_reader_open_threads.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.bl_reader_kind.restype = C.c_char_p
    L.bl_reader_kind.argtypes = [vp]
    L.bl_reader_next_text.argtypes = [vp, u64, C.POINTER(vp), C.POINTER(u64)]
    L.bl_reader_next_batch_device.argtypes = [vp, vp, u64, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    L.bl_file_count_u64.argtypes = [C.c_char_p, C.c_int, C.POINTER(u64)]
    L.bl_read_file_u64_host.argtypes = [C.c_char_p, C.c_int, vp, u64, C.POINTER(u64)]
    L.bl_read_file_u64.argtypes = [vp, C.c_char_p, C.c_int, vp, u64, C.POINTER(u64)]
    L.bl_merge_runs_u64.argtypes = [vp, C.POINTER(C.c_char_p), u32, vp, u64, C.POINTER(u64)]
    L.bl_count_allreduce.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(u64), C.c_int]
    L.bl_reader_next_record.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(vp), C.POINTER(u64)]
    L.bl_reader_next_batch.argtypes = [vp, vp, u64, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    L.bl_reader_last_batch.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
    L.bl_reader_last_name.restype = C.c_char_p
    L.bl_reader_last_name.argtypes = [vp, u64]
    L.bl_batch_from_text.argtypes = [vp, vp, u64, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    L.bl_run_file_name.argtypes = [C.c_char_p, C.c_char_p, u64, C.c_char_p, u64]
    L.bl_write_run_u64.argtypes = [vp, vp, u64, C.c_char_p]
    L.bl_write_vector_u64.argtypes = [vp, vp, u64, C.c_char_p]
    L.bl_device_alloc.argtypes = [vp, u64, C.POINTER(vp)]
    L.bl_device_free.argtypes = [vp, vp]
    L.bl_copy_to_host.argtypes = [vp, vp, vp, u64]
    L.bl_copy_to_device.argtypes = [vp, vp, vp, u64]
    L.bl_reader_open_shard.argtypes = [C.c_char_p, u32, u32, C.POINTER(vp)]
    L.bl_reader_shard_range.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.bl_host_alloc.argtypes = [vp, u64, C.POINTER(vp)]
    L.bl_host_free.argtypes = [vp, vp]
    L.bl_bgzf_walk.argtypes = [vp, u64, u64, u64, vp, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.bl_bgzf_inflate.argtypes = [vp, vp, u64, vp, u64, vp, u64, vp]
    L.bl_hash64_u64.restype = u64
    L.bl_hash64_u64.argtypes = [u64, u64]
    _lib = L
    return L


def check(rc):
    if rc != BL_OK:
        raise BiolibError(rc, lib().bl_last_error().decode(errors="replace"))
    return rc
