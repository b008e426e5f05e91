"""ctypes loader for libqexhip.so (the C ABI of include/qexhip.h).

There is no CPU fallback: if the shared library is missing, or no GPU is visible when a
context is created, this raises.  The library is built in-tree by `make -C qex_amd`
(see __graft_entry__.build()).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QEXHIP_LIB", os.path.join(_HERE, "libqexhip.so"))  # override: A/B builds
_lib = None

# every symbol include/qexhip.h declares: (name, restype, argtypes)
_vp, _ci, _cd = C.c_void_p, C.c_int, C.c_double
_pi, _pd = C.POINTER(C.c_int), C.POINTER(C.c_double)
SYMBOLS = [
    ("qexhip_init", _ci, [C.POINTER(_vp), _ci, _pi, _pi, _pi]),
    ("qexhip_finalize", _ci, [_vp]),
    ("qexhip_device_count", _ci, [_pi]),
    ("qexhip_last_error", C.c_char_p, []),
    ("qexhip_sync", _ci, [_vp]),
    ("qexhip_device_info", _ci, [_vp, C.c_char_p, _ci]),
    ("qexhip_layout_default_inner", _ci, [_pi, _ci, _pi]),
    ("qexhip_layout_simd_map", _ci, [_pi, _pi, _pi]),
    ("qexhip_layout_vec_simd_to_v1", _ci, [_pi, _pi, _vp, _vp]),
    ("qexhip_layout_vec_v1_to_simd", _ci, [_pi, _pi, _vp, _vp]),
    ("qexhip_layout_gauge_simd_to_v1", _ci, [_pi, _pi, C.POINTER(C.c_void_p), _vp]),
    ("qexhip_layout_gauge_v1_to_simd", _ci, [_pi, _pi, _vp, C.POINTER(C.c_void_p)]),
    ("qexhip_comm_unique_id", _ci, [C.c_char_p]),
    ("qexhip_comm_init", _ci, [_vp, C.c_char_p, _ci, _ci]),
    ("qexhip_comm_force_halo", _ci, [_vp, _ci]),
    ("qexhip_comm_info", _ci, [_vp, _pi, _pi, _pi, C.c_char_p, _ci]),
    ("qexhip_comm_count", _ci, [_vp, _pi]),
    ("qexhip_comm_transport", _ci, [_vp, C.c_char_p, _ci, C.POINTER(C.c_long)]),
    ("qexhip_stag_sweep_info", _ci, [_vp, _pi]),
    ("qexhip_stag_sweep_tuning", _ci, [_vp, _pd]),
    ("qexhip_dot", _ci, [_vp, _vp, _vp, _ci, _pd]),
    ("qexhip_dev_dot", _ci, [_vp, _ci, _ci, _ci, _pd]),
    ("qexhip_stag_set_links", _ci, [_vp, _vp, _vp]),
    ("qexhip_stag_dslash", _ci, [_vp, _vp, _vp, _ci, _cd, _cd]),
    ("qexhip_stag_D", _ci, [_vp, _vp, _vp, _cd, _cd]),
    ("qexhip_stag_D_acc", _ci, [_vp, _vp, _vp, _cd, _cd, _cd]),
    ("qexhip_stag_op_xx", _ci, [_vp, _vp, _vp, _cd, _ci]),
    ("qexhip_stag_eo_reconstruct", _ci, [_vp, _vp, _vp, _cd]),
    ("qexhip_stag_eo_reduce", _ci, [_vp, _vp, _vp, _cd]),
    ("qexhip_stag_stagD", _ci, [_vp, _vp, _vp, _ci, _cd, _cd, _cd]),
    ("qexhip_stag_outer", _ci, [_vp, _vp, _vp, _cd, _cd, _ci]),
    ("qexhip_stag_solve_xx", _ci, [_vp, _vp, _vp, _cd, _cd, _ci, _ci, _pi, _pd, _vp, _ci]),
    ("qexhip_stag_solve", _ci, [_vp, _vp, _vp, _cd, _cd, _ci, _pi, _pd]),
    ("qexhip_stag_solve_prev", _ci, [_vp, _vp, _vp, _cd, _cd, _ci, _ci, _pi, _pd]),
    ("qexhip_stag_solve_xx_multi", _ci, [_vp, _vp, _vp, _vp, _ci, _cd, _ci, _ci, _pi, _vp, _ci]),
    ("qexhip_stag_solve_multi", _ci, [_vp, _vp, _vp, _vp, _ci, _cd, _ci, _pi, _pd]),
    ("qexhip_norm2", _ci, [_vp, _vp, _ci, _pd]),
    ("qexhip_redot", _ci, [_vp, _vp, _vp, _ci, _pd]),
    ("qexhip_axpy", _ci, [_vp, _cd, _vp, _vp, _ci]),
    ("qexhip_xpay", _ci, [_vp, _vp, _cd, _vp, _ci]),
    ("qexhip_field_new", _ci, [_vp, _pi]),
    ("qexhip_field_free", _ci, [_vp, _ci]),
    ("qexhip_field_upload", _ci, [_vp, _ci, _vp]),
    ("qexhip_field_download", _ci, [_vp, _ci, _vp]),
    ("qexhip_field_zero", _ci, [_vp, _ci]),
    ("qexhip_dev_dslash", _ci, [_vp, _ci, _ci, _ci, _cd, _cd]),
    ("qexhip_dev_op_xx", _ci, [_vp, _ci, _ci, _cd, _ci]),
    ("qexhip_dev_solve_xx", _ci, [_vp, _ci, _ci, _cd, _cd, _ci, _ci, _pi, _pd, _vp, _ci]),
    ("qexhip_dev_solve_xx_continue", _ci, [_vp, _ci, _cd, _ci, _pi, _pd, _vp, _ci]),
    ("qexhip_dev_solve_xx_multi", _ci, [_vp, _pi, _ci, _pd, _ci, _cd, _ci, _ci, _pi, _vp, _ci]),
    ("qexhip_release_workspace", _ci, [_vp]),
    ("qexhip_dev_norm2", _ci, [_vp, _ci, _ci, _pd]),
    ("qexhip_dev_redot", _ci, [_vp, _ci, _ci, _ci, _pd]),
    ("qexhip_dev_D", _ci, [_vp, _ci, _ci, _cd, _cd]),
    ("qexhip_gauge_set", _ci, [_vp, _vp]),
    ("qexhip_gauge_get", _ci, [_vp, _vp]),
    ("qexhip_plaq", _ci, [_vp, _vp]),
    ("qexhip_gauge_force", _ci, [_vp, _vp, _cd]),
    ("qexhip_wflow", _ci, [_vp, _ci, _cd]),
    ("qexhip_flow_EQ", _ci, [_vp, _ci, _vp]),
    ("qexhip_flow_measure", _ci, [_vp, _vp, _vp]),
    ("qexhip_gauge_force_general", _ci, [_vp, _vp, _cd, _cd, _ci]),
    ("qexhip_wflow_general", _ci, [_vp, _ci, _cd, _cd, _cd, _ci]),
    ("qexhip_fat7", _ci, [_vp, _vp, _pd, _vp, _vp, _cd]),
    ("qexhip_hisq_smear", _ci, [_vp, _vp, _vp, _vp]),
    ("qexhip_nhyp_smear", _ci, [_vp, _vp, _vp, _cd, _cd, _cd]),
    ("qexhip_stag_solve_xx_batch", _ci, [_vp, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _pi, _vp]),
    ("qexhip_stag_solve_batch", _ci, [_vp, _ci, _vp, _vp, _vp, _vp, _ci, _pi, _vp]),
    ("qexhip_stag_links_info", _ci, [_vp, _pi, _pi, _vp]),
    ("qexhip_set_option", _ci, [_vp, C.c_char_p, _ci]),
    ("qexhip_fat7_deriv", _ci, [_vp, _vp, _vp, _vp, _vp, _cd, _vp]),
    ("qexhip_hisq_force", _ci, [_vp, _vp, _vp, _vp, _vp]),
    ("qexhip_hisq_prepare", _ci, [_vp, _vp, _vp, _vp]),
    ("qexhip_hisq_closure_force", _ci, [_vp, _vp, _vp, _vp]),
    ("qexhip_hisq_release", _ci, [_vp]),
    ("qexhip_hisq_fermion_force", _ci, [_vp, _vp, _vp, _vp, _ci]),
    ("qexhip_stag_set_links_hisq", _ci, [_vp, _vp]),
    ("qexhip_stag_set_links_nhyp", _ci, [_vp, _vp, _cd, _cd, _cd, _pi, _pi]),
    ("qexhip_nhyp_prepare", _ci, [_vp, _vp, _cd, _cd, _cd, _vp]),
    ("qexhip_nhyp_force", _ci, [_vp, _vp, _vp]),
    ("qexhip_nhyp_release", _ci, [_vp]),
    ("qexhip_nhyp_fforce", _ci, [_vp, _vp, _ci, _vp, _vp, _vp, _vp, _ci, _pi, _pi, _pi]),
    ("qexhip_nhyp_gauge_force", _ci, [_vp, _vp, _cd, _cd, _cd]),
    ("qexhip_nhyp_fermion_force", _ci, [_vp, _vp, _vp, _vp, _ci, _pi, _pi]),
    ("qexhip_gauge_action", _ci, [_vp, _cd, _cd, _cd, _vp]),
    ("qexhip_gauge_update", _ci, [_vp, _vp, _cd]),
    ("qexhip_gauge_reunit", _ci, [_vp]),
    ("qexhip_wline", _ci, [_vp, _pi, _ci, _vp]),
    ("qexhip_polyakov_loops", _ci, [_vp, _vp]),
    ("qexhip_plaq_s4", _ci, [_vp, _vp]),
    ("qexhip_rng_new", _ci, [_vp, _ci, C.c_ulonglong, _pi, _pi, _ci]),
    ("qexhip_rng_free", _ci, [_vp]),
    ("qexhip_rng_uniform", _ci, [_vp, _ci, _vp]),
    ("qexhip_rng_gaussian_vector", _ci, [_vp, _vp]),
    ("qexhip_rng_u1_vector", _ci, [_vp, _vp]),
    ("qexhip_rng_random_tah", _ci, [_vp, _vp]),
    ("qexhip_rng_gauge_random", _ci, [_vp, _vp]),
    ("qexhip_rng_gauge_warm", _ci, [_vp, _cd, _vp]),
    ("qexhip_rng_dev_gaussian_vector", _ci, [_vp, _vp, _ci]),
    ("qexhip_rng_dev_u1_vector", _ci, [_vp, _vp, _ci]),
    ("qexhip_md_refresh_momenta", _ci, [_vp, _vp]),
    ("qexhip_dev_zero", _ci, [_vp, _ci, _ci]),
    ("qexhip_dev_solve_batch", _ci, [_vp, _ci, _pi, _pi, _pd, _pd, _ci, _pi, _pd]),
    ("qexhip_nhyp_fforce_dev", _ci, [_vp, _vp, _ci, _pi, _pd, _pd, _pd, _ci, _pi, _pi, _pi]),
    ("qexhip_rng_state_words", _ci, [_vp]),
    ("qexhip_rng_get_state", _ci, [_vp, _vp]),
    ("qexhip_rng_set_state", _ci, [_vp, _vp]),
    ("qexhip_io_write_field", _ci, [C.c_char_p, _pi, _vp, _ci, _ci, C.c_char_p, C.c_char, _ci, _ci, C.c_char_p, C.c_char_p]),
    ("qexhip_io_read_field", _ci, [C.c_char_p, _pi, _vp, _ci, _ci, C.c_char_p]),
    ("qexhip_md_begin", _ci, [_vp, _vp, _vp]),
    ("qexhip_md_end", _ci, [_vp, _vp, _vp]),
    ("qexhip_md_momentum_norm2", _ci, [_vp, _pd]),
    ("qexhip_md_update_links", _ci, [_vp, _cd]),
    ("qexhip_md_gauge_force", _ci, [_vp, _cd, _cd, _cd]),
    ("qexhip_md_kick", _ci, [_vp, _ci, _cd]),
    ("qexhip_md_shift_links", _ci, [_vp, _ci, _cd]),
    ("qexhip_md_save_links", _ci, [_vp]),
    ("qexhip_md_restore_links", _ci, [_vp]),
    ("qexhip_io_metadata", _ci, [C.c_char_p, C.c_char_p, _ci, C.c_char_p, _ci, _pi, _pi]),
    ("qexhip_io_crc32", _ci, [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint)]),   # data pointer, not a handle
    ("qexhip_io_gauge_info", _ci, [C.c_char_p, _pi, C.c_char_p, _pi]),
    ("qexhip_io_read_gauge", _ci, [C.c_char_p, _pi, _vp, _vp, _vp]),
    ("qexhip_io_read_gauge_slab", _ci, [C.c_char_p, _pi, _ci, _ci, _vp]),
    ("qexhip_io_write_gauge", _ci, [C.c_char_p, _pi, _vp, C.c_char, C.c_char_p, C.c_char_p]),
    ("qexhip_timers_enable", _ci, [_vp, _ci]),
    ("qexhip_timers_reset", _ci, [_vp]),
    ("qexhip_timers_get", _ci, [_vp, C.c_char_p, C.POINTER(C.c_long), _pd]),
    ("qexhip_debug_geom", _ci, [_pi, _ci, _ci, _pi]),
    ("qexhip_debug_nbr_pos", _ci, [_pi, _ci, _ci, _ci, _ci, _ci, _ci]),
    ("qexhip_debug_site_coord", _ci, [_pi, _ci, _ci, _pi]),
    ("qexhip_debug_tile_order", _ci, [_pi, _ci, _ci, _pi, _ci]),
]


class QexHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QexHipError(
                f"{LIB_PATH} not found: build it with `make -C {_HERE}` "
                "(__graft_entry__.build()).  qex_amd has no CPU fallback."
            )
        # RTLD_LOCAL | RTLD_DEEPBIND: libqexhip binds to the ROCm runtime it was linked against
        # (/opt/rocm) and keeps it out of the global namespace.  PyTorch wheels bundle their own
        # libamdhip64/librccl; mixing the two through symbol interposition (RTLD_GLOBAL) corrupts
        # the heap at exit ("free(): invalid pointer").
        L = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL | getattr(os, "RTLD_DEEPBIND", 0))
        for name, res, args in SYMBOLS:
            f = getattr(L, name)  # AttributeError if the symbol is missing
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


_tune = None


def tune_lib():
    """libqexhip_tune.so (include/qexhip_tune.h): measurement scaffolding for scratch/ and profiles/ -- A/B sweep variants,
    streaming / fp64 calibration kernels.  Not part of the product library; nothing under qex_amd/ calls it."""
    global _tune
    if _tune is None:
        lib()                                   # libqexhip.so first: the tune library resolves its internals from it
        path = os.path.join(os.path.dirname(LIB_PATH), "libqexhip_tune.so")
        if not os.path.exists(path):
            raise QexHipError(f"{path} not found: build it with `make -C {_HERE} libqexhip_tune.so`")
        T = C.CDLL(path, mode=C.RTLD_LOCAL | getattr(os, "RTLD_DEEPBIND", 0))
        pd = C.POINTER(C.c_double)
        T.qexhip_tune_dslash.argtypes = [_vp, _ci, _ci, _ci, pd]
        T.qexhip_tune_dslash_norm2.argtypes = [_vp, pd]
        T.qexhip_tune_stream.argtypes = [_vp, _ci, C.c_size_t, _ci, _ci, pd]
        T.qexhip_tune_fma64.argtypes = [_vp, _ci, _ci, _ci, _ci, pd]
        T.qexhip_tune_gather.argtypes = [_vp, _ci, _ci, _ci, _ci, pd]
        T.qexhip_tune_gather_rows.argtypes = [_vp, _ci, _ci, _ci, _ci, _ci, pd]
        _tune = T
    return _tune


def check(rc):
    if rc != 0:
        msg = lib().qexhip_last_error()
        raise QexHipError(f"libqexhip error {rc}: {msg.decode() if msg else ''}")
