"""ctypes binding of libcatchhip.so (include/catchhip.h).

The product path has no CPU fallback: if the HIP library is missing or no
GPU is visible, every entry point raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CATCHHIP_LIB: another build of the same ABI (A/B runs of two kernel versions)
LIB_PATH = os.environ.get("CATCHHIP_LIB") or os.path.join(_HERE, "libcatchhip.so")



def test_env(name, default=None):
    """A TEST HOOK of the Python side (a switch that forces one of several exact code paths so that tests can
    compare them): only honoured under CATCHHIP_TEST_HOOKS=1, like the C library's (csrc/internal.h).  The
    supported settings are read with os.environ directly and listed in README.md."""
    if os.environ.get("CATCHHIP_TEST_HOOKS", "0") in ("", "0"):
        return default
    return os.environ.get(name, default)


c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_vp = ctypes.c_void_p
c_u16p = ctypes.POINTER(ctypes.c_uint16)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_vpp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); must list every symbol declared in catchhip.h
PROTOTYPES = {
    "catchhip_abi_version": (ctypes.c_int, []),
    "catchhip_last_error": (ctypes.c_char_p, []),
    "catchhip_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "catchhip_ctx_create": (ctypes.c_int, [ctypes.c_int, c_vpp]),
    "catchhip_ctx_destroy": (ctypes.c_int, [c_vp]),
    "catchhip_ctx_sync": (ctypes.c_int, [c_vp]),
    "catchhip_pool_stats": (ctypes.c_int, [c_i64p]),
    "catchhip_pool_trim": (ctypes.c_int, []),
    "catchhip_ctx_last_kernel_ms": (ctypes.c_int, [
        c_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double), c_i64p]),
    "catchhip_ctx_last_counters": (ctypes.c_int, [c_vp, c_i64p]),
    "catchhip_ctx_last_seeds_dropped": (ctypes.c_int, [c_vp, c_i64p]),
    "catchhip_ctx_last_join_counters": (ctypes.c_int, [c_vp, c_i64p]),
    "catchhip_ctx_last_solver_counters": (ctypes.c_int, [c_vp, c_i64p]),
    "catchhip_ctx_last_ndf_counters": (ctypes.c_int, [c_vp, c_i64p]),
    "catchhip_targets_create": (ctypes.c_int, [
        c_vp, c_u8p, c_i64p, c_i32p, ctypes.c_int64, ctypes.c_int32, c_vpp]),
    "catchhip_targets_create_ptrs": (ctypes.c_int, [
        c_vp, ctypes.POINTER(ctypes.c_void_p), c_i64p, c_i32p, ctypes.c_int64,
        ctypes.c_int32, c_vpp]),
    "catchhip_pyset_order": (ctypes.c_int, [c_i64p, ctypes.c_int64, c_i64p]),
    "catchhip_pyset_order_strs": (ctypes.c_int, [c_u8p, c_i64p, ctypes.c_int64, c_i64p]),
    "catchhip_pyset_order_device": (ctypes.c_int, [c_vp, c_i64p, ctypes.c_int64, c_i64p]),
    "catchhip_targets_destroy": (ctypes.c_int, [c_vp]),
    "catchhip_targets_rebind": (ctypes.c_int, [c_vp, c_vp]),
    "catchhip_probes_rebind": (ctypes.c_int, [c_vp, c_vp]),
    "catchhip_candidates_rebind": (ctypes.c_int, [c_vp, c_vp]),
    "catchhip_probes_create": (ctypes.c_int, [
        c_vp, c_u8p, c_i64p, ctypes.c_int64, c_i32p, c_i32p, c_i32p,
        ctypes.c_int64, ctypes.c_int32, c_vpp]),
    "catchhip_probes_destroy": (ctypes.c_int, [c_vp]),
    "catchhip_cover_scan": (ctypes.c_int, [
        c_vp, c_vp, c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_int32, ctypes.c_int32, c_vpp, c_i64p]),
    "catchhip_cover_ranges": (ctypes.c_int, [
        c_vp, c_vp, c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_int32, ctypes.c_int32, c_vpp, c_i64p]),
    "catchhip_rows_fetch": (ctypes.c_int, [
        c_vp, c_vp, c_i32p, c_i32p, c_i64p, c_i64p]),
    "catchhip_rows_from_host": (ctypes.c_int, [
        c_vp, c_i32p, c_i32p, c_i64p, c_i64p, ctypes.c_int64, c_i64p,
        ctypes.c_int32, c_vpp]),
    "catchhip_rows_destroy": (ctypes.c_int, [c_vp]),
    "catchhip_tolerant_bp": (ctypes.c_int, [
        c_vp, c_vp, c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
        c_i64p]),
    "catchhip_setcover_greedy": (ctypes.c_int, [
        c_vp, c_vp, ctypes.c_int64, c_i64p, c_f64p, c_i64p, c_i64p]),
    "catchhip_ndf_minhash": (ctypes.c_int, [
        c_vp, c_u8p, c_i64p, ctypes.c_int64, ctypes.c_int32, c_i64p,
        ctypes.c_int32, ctypes.c_int32, ctypes.c_double, c_u8p]),
    "catchhip_ndf_minhash_many": (ctypes.c_int, [
        c_vp, c_u8p, c_i64p, ctypes.c_int64, c_i64p, ctypes.c_int64,
        ctypes.c_int32, c_i64p, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_double, c_u8p]),
    "catchhip_setcover_filter": (ctypes.c_int, [
        c_vp, c_vp, c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, c_i64p, c_f64p,
        c_i64p, c_i64p, c_i64p]),
    "catchhip_setcover_filter_many": (ctypes.c_int, [
        ctypes.c_int32, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
        ctypes.POINTER(c_vp), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_int32, ctypes.c_int32, c_i64p, ctypes.POINTER(c_i64p),
        ctypes.POINTER(c_f64p), ctypes.POINTER(c_i64p), c_i64p, c_i64p]),
    "catchhip_comm_unique_id": (ctypes.c_int, [c_u8p]),
    "catchhip_comm_init": (ctypes.c_int, [
        c_vp, c_u8p, ctypes.c_int32, ctypes.c_int32]),
    "catchhip_comm_destroy": (ctypes.c_int, [c_vp]),
    "catchhip_comm_selftest": (ctypes.c_int, [c_vp, ctypes.c_int64]),
    "catchhip_comm_info": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int64]),
    "catchhip_shard_create": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_i64p, c_vpp]),
    "catchhip_shard_destroy": (ctypes.c_int, [c_vp]),
    "catchhip_shard_count": (ctypes.c_int, [c_vp]),
    "catchhip_shard_create_p": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_i64p, ctypes.POINTER(ctypes.c_double), c_vpp]),
    "catchhip_shard_create_pi": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_i64p, ctypes.POINTER(ctypes.c_double), ctypes.c_int, c_vpp]),
    "catchhip_shard_verdict": (ctypes.c_int, [c_vp]),
    "catchhip_shard_claim_check": (ctypes.c_int, [c_vp]),
    "catchhip_shard_apply": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_int32)]),
    "catchhip_shard_buffers": (ctypes.c_int, [c_vp, c_vpp, c_i64p, c_vpp, c_i64p]),
    "catchhip_shard_picks": (ctypes.c_int, [c_vp, c_i64p, c_i64p]),
    "catchhip_shard_info": (ctypes.c_int, [c_vp, c_i64p]),
    "catchhip_shard_allreduce": (ctypes.c_int, [c_vp, ctypes.c_int32]),
    "catchhip_shard_buffer_copy": (ctypes.c_int, [c_vp, ctypes.c_int32, c_vp, ctypes.c_int32]),
    "catchhip_shard_allreduce_local": (ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(c_vp), ctypes.c_int32]),
    "catchhip_shard_solve": (ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32, ctypes.c_int32, c_i32p]),
    "catchhip_ndf_hamming": (ctypes.c_int, [
        c_vp, c_u8p, ctypes.c_int64, ctypes.c_int32, c_i32p, ctypes.c_int32,
        ctypes.c_int32, ctypes.c_int32, c_u8p]),
    "catchhip_sigs_create": (ctypes.c_int, [
        c_vp, c_u8p, c_u64p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_uint32,
        ctypes.c_uint32, ctypes.c_uint32, c_vpp]),
    "catchhip_sigs_create_ptrs": (ctypes.c_int, [
        c_vp, ctypes.c_void_p, c_i64p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_uint32,
        ctypes.c_uint32, ctypes.c_uint32, c_vpp]),
    "catchhip_sigs_destroy": (None, [c_vp]),
    "catchhip_sigs_fetch": (ctypes.c_int, [c_vp, c_vp, c_u32p]),
    "catchhip_sigs_common_row": (ctypes.c_int, [
        c_vp, c_vp, ctypes.c_uint32, c_u16p]),
    "catchhip_sigs_condensed": (ctypes.c_int, [c_vp, c_vp, c_f32p, c_f32p]),
    "catchhip_sigs_neighbors": (ctypes.c_int, [
        c_vp, c_vp, ctypes.c_uint32, ctypes.c_uint32, c_u64p, ctypes.c_int64, c_i64p]),
    "catchhip_sigs_neighbors_many": (ctypes.c_int, [
        c_vp, c_vp, c_u32p, ctypes.c_int64, ctypes.c_uint32, c_u64p, ctypes.c_int64, c_i64p]),
    "catchhip_sigs_graph": (ctypes.c_int, [c_vp, c_vp, ctypes.c_uint32, ctypes.c_int64, c_i64p]),
    "catchhip_sigs_graph_fetch": (ctypes.c_int, [c_vp, c_vp, c_i64p, c_u32p, c_u32p]),
    "catchhip_dfs_create": (ctypes.c_int, [ctypes.c_uint32, c_i64p, c_u32p, c_u32p, ctypes.c_uint32, c_vpp]),
    "catchhip_dfs_destroy": (None, [c_vp]),
    "catchhip_dfs_run": (ctypes.c_int, [c_vp, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32), c_i64p]),
    "catchhip_dfs_seen": (ctypes.c_int, [c_vp, ctypes.POINTER(c_u32p), c_i64p]),
    "catchhip_dfs_new_queued": (ctypes.c_int, [c_vp, ctypes.POINTER(c_u32p), c_i64p]),
    "catchhip_dfs_set_copy_rank": (ctypes.c_int, [c_vp, c_i64p]),
    "catchhip_dfs_set_copy_members": (ctypes.c_int, [c_vp, c_i64p, ctypes.c_int64]),
    "catchhip_dfs_push": (ctypes.c_int, [c_vp, c_i64p, c_u8p, ctypes.c_int64]),
    "catchhip_dfs_counts": (ctypes.c_int, [c_vp, c_i64p]),
    "catchhip_dfs_run_all": (ctypes.c_int, [c_vp, c_u32p, c_i64p, c_i64p, c_i64p]),
    "catchhip_pyintset_create": (ctypes.c_int, [ctypes.c_uint32, c_vpp]),
    "catchhip_pyintset_destroy": (None, [c_vp]),
    "catchhip_pyintset_isub": (ctypes.c_int, [c_vp, c_u32p, ctypes.c_int64]),
    "catchhip_pyintset_list": (ctypes.c_int, [c_vp, ctypes.c_int32, c_u32p, ctypes.c_int64, ctypes.POINTER(c_u32p), c_i64p]),
    "catchhip_cover_scan_first_seen": (ctypes.c_int, [
        c_vp, c_vp, c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_int32, ctypes.c_int32, c_u32p, c_vpp, c_i64p]),
    "catchhip_rows_fetch_first_seen": (ctypes.c_int, [c_vp, c_vp, c_u64p]),
    "catchhip_candidates_create": (ctypes.c_int, [
        c_vp, c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, c_vpp,
        c_i64p, c_i64p]),
    "catchhip_candidates_destroy": (None, [c_vp]),
    "catchhip_candidates_fetch": (ctypes.c_int, [
        c_vp, c_vp, c_i64p, ctypes.c_int64, c_i64p]),
    "catchhip_candidates_ndf_hamming": (ctypes.c_int, [
        c_vp, c_vp, c_i32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
        c_i64p]),
    "catchhip_candidates_ndf_minhash": (ctypes.c_int, [
        c_vp, c_vp, ctypes.c_int32, c_i64p, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_double, c_i64p]),
    "catchhip_candidates_ndf_hamming_many": (ctypes.c_int, [
        c_vp, c_vp, c_i32p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_int32, c_i64p]),
    "catchhip_candidates_ndf_minhash_many": (ctypes.c_int, [
        c_vp, c_vp, ctypes.c_int32, c_i64p, ctypes.c_int64, ctypes.c_int32,
        ctypes.c_int32, ctypes.c_double, c_i64p]),
    "catchhip_candidates_groups": (ctypes.c_int, [c_vp, c_vp, c_i32p]),
    "catchhip_probes_from_candidates": (ctypes.c_int, [
        c_vp, c_vp, c_i32p, c_i32p, ctypes.c_int64, ctypes.c_int32, c_vpp]),
    "catchhip_probes_from_candidates_draws": (ctypes.c_int, [
        c_vp, c_vp, c_u8p, ctypes.c_int32, ctypes.c_int32, c_vpp]),
    "catchhip_adapter_votes": (ctypes.c_int, [
        c_vp, c_vp, ctypes.c_int64, c_i64p, c_i64p, c_i64p]),
    "catchhip_rows_stats": (ctypes.c_int, [
        c_vp, c_vp, c_i64p, c_i64p, ctypes.c_int64, c_i64p]),
    "catchhip_rows_cover_check": (ctypes.c_int, [
        c_vp, c_vp, ctypes.c_int64, c_i64p, ctypes.c_int64, c_f64p, c_i64p]),
    "catchhip_probes_set_groups": (ctypes.c_int, [c_vp, c_vp, c_i32p]),
    "catchhip_targets_set_groups": (ctypes.c_int, [c_vp, c_vp, c_i32p]),
}

_lib = None


class CatchHipError(RuntimeError):
    pass


def lib():
    """Load libcatchhip.so (raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CatchHipError(
                "libcatchhip.so is not built (%s); run "
                "`python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C catch_amd/csrc`.  There is no CPU fallback."
                % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().catchhip_last_error()
        msg = msg.decode("utf-8", "replace") if msg else ""
        if rc == -1:
            raise ValueError("catchhip: " + msg)
        if rc == -4:
            raise IndexError("catchhip: " + msg)
        raise CatchHipError("catchhip error %d: %s" % (rc, msg))
