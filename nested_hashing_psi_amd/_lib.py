"""ctypes loader for libpiehip.so.  Fails loudly when the HIP library is missing -- the product
has no CPU path."""
import ctypes as C
import os

from .build import LIB_PATH

u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
f64p = C.POINTER(C.c_double)

NKERNELS = 13

# every symbol include/piehip.h declares: (restype, argtypes)
SYMBOLS = {
    "piehip_version": (C.c_int, []),
    "piehip_last_error": (C.c_char_p, []),
    "piehip_kernel_name": (C.c_char_p, [C.c_int]),
    "piehip_default_moduli": (C.c_int, [C.c_uint32, C.c_uint32, u64p, u64p]),
    "piehip_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_uint64, u64p, u64p, C.c_int, C.c_void_p]),
    "piehip_destroy": (C.c_int, [C.c_void_p]),
    "piehip_get_moduli": (C.c_int, [C.c_void_p, u64p]),
    "piehip_get_root": (C.c_int, [C.c_void_p, C.c_uint32, u64p]),
    "piehip_get_twiddles": (C.c_int, [C.c_void_p, C.c_uint32, u64p, u64p]),
    "piehip_get_slot_positions": (C.c_int, [C.c_void_p, u32p]),
    "piehip_load_relin_key": (C.c_int, [C.c_void_p, u64p]),
    "piehip_load_db": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, u64p, u64p]),
    "piehip_load_db_slots": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, i64p, i64p]),
    "piehip_build_db": (C.c_int, [C.c_void_p, u64p, C.c_size_t] + [C.c_uint32] * 5 + [C.c_uint64] * 4),
    "piehip_load_db_table": (C.c_int, [C.c_void_p, u64p] + [C.c_uint32] * 5 + [C.c_uint64] * 2),
    "piehip_build_db_bins": (C.c_int, [C.c_void_p, u64p, C.c_size_t] + [C.c_uint32] * 5 + [C.c_uint64] * 4 + [C.c_uint32] * 2),
    "piehip_load_db_table_bins": (C.c_int, [C.c_void_p, u64p] + [C.c_uint32] * 5 + [C.c_uint64] * 2 + [C.c_uint32] * 2),
    "piehip_reserve": (C.c_int, [C.c_void_p, C.c_size_t] + [C.c_uint32] * 7),
    "piehip_get_hash_table": (C.c_int, [C.c_void_p, u64p]),
    "piehip_tabulation_hash": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, u64p, C.c_size_t, u64p]),
    "piehip_client_cuckoo_table": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, u64p, C.c_size_t, u64p]),
    "piehip_set_index": (C.c_int, [C.c_void_p, u64p]),
    "piehip_set_minus": (C.c_int, [C.c_void_p, u64p]),
    "piehip_set_index_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "piehip_set_minus_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "piehip_run": (C.c_int, [C.c_void_p]),
    "piehip_sync": (C.c_int, [C.c_void_p]),
    "piehip_join": (C.c_int, [C.c_void_p]),
    "piehip_run_into": (C.c_int, [C.c_void_p, C.c_void_p]),
    "piehip_run_host": (C.c_int, [C.c_void_p, u64p, u64p, u64p]),
    "piehip_run_host_async": (C.c_int, [C.c_void_p, u64p, u64p, u64p]),
    "piehip_run_host_wait": (C.c_int, [C.c_void_p]),
    "piehip_stage_minus": (C.c_int, [C.c_void_p, u64p]),
    "piehip_stage_index_row": (C.c_int, [C.c_void_p, C.c_uint32, u64p]),
    "piehip_run_staged": (C.c_int, [C.c_void_p, u64p]),
    "piehip_stage_minus_q": (C.c_int, [C.c_void_p, C.c_uint32, u64p]),
    "piehip_stage_index_row_q": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, u64p]),
    "piehip_stage_index_ct_q": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, u64p]),
    "piehip_stage_reset": (C.c_int, [C.c_void_p]),
    "piehip_host_buffers_q": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(u64p), C.POINTER(u64p), C.POINTER(u64p)]),
    "piehip_load_relin_key_q": (C.c_int, [C.c_void_p, C.c_uint32, u64p]),
    "piehip_host_buffers": (C.c_int, [C.c_void_p, C.POINTER(u64p), C.POINTER(u64p), C.POINTER(u64p)]),
    "piehip_set_run_streams": (C.c_int, [C.c_void_p, C.c_uint32]),
    "piehip_set_graph": (C.c_int, [C.c_void_p, C.c_int]),
    "piehip_attach_database": (C.c_int, [C.c_void_p, C.c_void_p]),
    "piehip_set_query_batch": (C.c_int, [C.c_void_p, C.c_uint32]),
    "piehip_get_query_batch": (C.c_int, [C.c_void_p, u32p]),
    "piehip_set_index_q": (C.c_int, [C.c_void_p, C.c_uint32, u64p]),
    "piehip_set_minus_q": (C.c_int, [C.c_void_p, C.c_uint32, u64p]),
    "piehip_set_index_device_q": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "piehip_set_minus_device_q": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "piehip_get_results": (C.c_int, [C.c_void_p, u64p]),
    "piehip_results_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "piehip_copy_results_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "piehip_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "piehip_rccl_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "piehip_rccl_attach": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "piehip_rccl_destroy": (C.c_int, [C.c_void_p]),
    "piehip_rccl_bin_slice": (C.c_int, [C.c_uint32, C.c_int, C.c_int, u32p, u32p]),
    "piehip_rccl_broadcast": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "piehip_rccl_broadcast_query": (C.c_int, [C.c_void_p, C.c_int]),
    "piehip_gather_results": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]),
    "piehip_gather_results_host": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(u64p)]),
    "piehip_ntt": (C.c_int, [C.c_void_p, u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]),
    "piehip_eval_add": (C.c_int, [C.c_void_p, u64p, u64p, u64p]),
    "piehip_eval_mult_plain": (C.c_int, [C.c_void_p, u64p, u64p, u64p]),
    "piehip_eval_mult": (C.c_int, [C.c_void_p, u64p, u64p, C.c_uint32, C.c_int, u64p]),
    "piehip_eval_automorph": (C.c_int, [C.c_void_p, u64p, C.c_uint32, u64p, u64p]),
    "piehip_encode": (C.c_int, [C.c_void_p, i64p, C.c_uint32, C.c_uint32, u64p]),
    "piehip_base_convert": (C.c_int, [C.c_void_p, C.c_int, u64p, C.c_uint32, u64p]),
    "piehip_bench_ntt": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, f64p]),
    "piehip_rotation_galois": (C.c_int, [C.c_void_p, C.c_int32, u32p]),
    "piehip_load_rotation_keys": (C.c_int, [C.c_void_p, C.c_uint32, i32p, u64p]),
    "piehip_fhepie_load_table": (C.c_int, [C.c_void_p] + [C.c_uint32] * 4 + [i64p, i64p]),
    "piehip_fhepie_set_index": (C.c_int, [C.c_void_p, u64p]),
    "piehip_fhepie_run": (C.c_int, [C.c_void_p]),
    "piehip_fhepie_get_results": (C.c_int, [C.c_void_p, u64p]),
    "piehip_client_rot_keygen": (C.c_int, [C.c_void_p, u64p, C.c_int32, C.c_uint64, u64p]),
    "piehip_client_keygen": (C.c_int, [C.c_void_p, C.c_uint64, u64p]),
    "piehip_client_relin_keygen": (C.c_int, [C.c_void_p, u64p, C.c_uint64, u64p]),
    "piehip_client_encrypt": (C.c_int, [C.c_void_p, u64p, i64p, C.c_uint32, C.c_uint32, u64p, u64p]),
    "piehip_client_decrypt": (C.c_int, [C.c_void_p, u64p, u64p, C.c_uint32, C.c_uint32, i64p]),
    "piehip_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "piehip_profile_read": (C.c_int, [C.c_void_p, u32p, f64p, f64p]),
    "piehip_profile_read_n": (C.c_int, [C.c_void_p, C.c_uint32, u32p, f64p, f64p]),
    "piehip_set_transform_slots": (C.c_int, [C.c_void_p, C.c_uint32]),
    "piehip_get_transform_slots": (C.c_int, [C.c_void_p, u32p, u32p]),
    "piehip_upload_turn_wait": (C.c_int, [C.c_void_p, f64p, f64p, u64p]),
    "piehip_set_host_path_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "piehip_host_path_times": (C.c_int, [C.c_void_p, f64p, f64p]),
    "piehip_rccl_wait": (C.c_int, [C.c_void_p, C.c_uint32]),
    "piehip_rccl_abort": (C.c_int, [C.c_void_p]),
    "piehip_rccl_agree": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_uint32]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                "libpiehip.so is not built (%s). Run nested_hashing_psi_amd.build(); "
                "there is no CPU fallback for the PIE hot path." % path)
        # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64 (same soname as
        # /opt/rocm's).  If libpiehip.so were loaded first it would bind the system copy and a later
        # `import torch` would bring a second runtime into the process, which then finds no device.
        # Importing torch first makes the loader resolve libpiehip's libamdhip64.so.7 to the copy
        # torch already mapped (torch is only plumbing here: streams, device tensors, RCCL).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)  # AttributeError if the library does not export it
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib
