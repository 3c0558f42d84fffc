"""Loads libpsacx.so (the HIP engine) and declares the C-ABI of include/psacx.h.

There is no CPU fallback: if the library is missing or no GPU is present the
calls raise.  When PyTorch is used in the same process it must own the HIP
runtime, so torch is imported before the library whenever it is installed
(both resolve libamdhip64.so.7 to one copy).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpsacx.so")

PSACX_LCP = 1
PSACX_NO_FAST = 2
PSACX_PROFILE = 4
PSACX_MAX_ROUNDS = 72

# psacx_configure options (include/psacx.h); Context.configure(force_diet=1, ...) takes them by name
OPTIONS = {"reset": 0, "force_diet": 1, "diet_cap": 2, "one_stage": 3, "ties_radix": 4, "no_one_word": 5, "one_word_always": 6, "one_word_min": 7,
           "widen_last": 8, "no_digit_bytes": 9, "no_bucket_sort": 10, "isa_update": 11, "gather": 12, "no_heavy": 13, "no_whole": 14, "no_lazy_ranks": 15, "no_early_out": 16}
MULTI_OPTIONS = {"layout": 1, "slab": 2, "output_slack": 3, "trace": 4, "wire_piece": 5, "pieces": 6, "check_chunks": 7, "global_refine_sort": 8,
                 "one_stage": 9, "two_word": 10, "one_word": 11, "no_slices": 12, "slice_wide": 13, "slice_shape": 14}
MULTI_FORCE_WIRE, MULTI_NO_RCCL, MULTI_SHM = 1, 2, 4
# The library never reads the environment.  With ENV_KNOBS set (the test suite and the tools/ scripts do: PSACX_ENV_KNOBS=1, tests/conftest.py)
# the Python wrappers call the debug shims psacx_configure_from_env / psacx_multi_configure_from_env before every call that runs the engine,
# so that PSACX_* variables select the forms of single stages, and read the transport variables when a multi-GPU context is made.
ENV_KNOBS = bool(os.environ.get("PSACX_ENV_KNOBS"))
if ENV_KNOBS and os.environ.get("PSACX_LIB"):          # (tools/experiments: a variant build of the library, e.g. with parts of a kernel left out to time them)
    LIB_PATH = os.environ["PSACX_LIB"]

EXPORTS = [
    "psacx_create", "psacx_destroy", "psacx_strerror", "psacx_last_hip_error", "psacx_trim", "psacx_configure", "psacx_configure_from_env", "psacx_debug_env",
    "psacx_construct_u32", "psacx_construct_u64", "psacx_construct_dev_u32", "psacx_construct_dev_u64",
    "psacx_construct_gsa_u32", "psacx_construct_gsa_u64", "psacx_construct_gsa_dev_u32", "psacx_construct_gsa_dev_u64",
    "psacx_construct_lc_u32", "psacx_construct_lc_u64", "psacx_construct_lc_dev_u32", "psacx_construct_lc_dev_u64",
    "psacx_get_stats", "psacx_profile", "psacx_check_dev_u32", "psacx_check_dev_u64", "psacx_pair_sort_dev_u32", "psacx_pair_sort_dev_u64", "psacx_ansv_u32",
    "psacx_ansv_u64", "psacx_ansv_dev_u32", "psacx_ansv_dev_u64", "psacx_suffix_tree_u32", "psacx_suffix_tree_u64", "psacx_dev_alloc", "psacx_dev_free", "psacx_copy_h2d", "psacx_copy_d2h", "psacx_sync",
    "psacx_rand_dna", "psacx_synth_text_dev",
    "psacx_multi_create", "psacx_multi_unique_id", "psacx_multi_create_rank", "psacx_multi_destroy", "psacx_multi_nranks",
    "psacx_multi_nlocal", "psacx_multi_uses_rccl", "psacx_multi_last_error", "psacx_multi_ctx", "psacx_multi_construct_dev_u32",
    "psacx_multi_construct_dev_u64", "psacx_multi_construct_u32", "psacx_multi_construct_u64", "psacx_multi_get_stats",
    "psacx_multi_check_dev_u32", "psacx_multi_check_dev_u64", "psacx_multi_ansv_dev_u32", "psacx_multi_ansv_dev_u64",
    "psacx_multi_construct_gsa_dev_u32", "psacx_multi_construct_gsa_dev_u64", "psacx_multi_construct_gsa_u32", "psacx_multi_construct_gsa_u64",
    "psacx_multi_suffix_tree_dev_u32", "psacx_multi_suffix_tree_dev_u64", "psacx_multi_suffix_tree_u32", "psacx_multi_suffix_tree_u64",
    "psacx_multi_left_chars_dev_u32", "psacx_multi_left_chars_dev_u64", "psacx_multi_construct_lc_u32", "psacx_multi_construct_lc_u64",
    "psacx_multi_configure", "psacx_multi_configure_from_env", "psacx_multi_create_ex", "psacx_multi_create_rank_ex", "psacx_multi_get_memory", "psacx_multi_transport", "psacx_multi_get_wire", "psacx_multi_get_phases", "psacx_multi_last_form",
]


class Round(C.Structure):
    _fields_ = [("h", C.c_uint64), ("active", C.c_uint64), ("unfinished_buckets", C.c_uint64),
                ("unfinished_elements", C.c_uint64), ("sort_passes", C.c_uint32),
                ("sort_passes_skipped", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("sigma", C.c_uint32), ("bits_per_char", C.c_uint32), ("k", C.c_uint32),
                ("n_rounds", C.c_uint32), ("rounds", Round * PSACX_MAX_ROUNDS),
                ("ms_total", C.c_double), ("ms_alphabet", C.c_double), ("ms_kmer", C.c_double),
                ("ms_sort_hist", C.c_double), ("ms_sort_scatter", C.c_double), ("ms_rebucket", C.c_double),
                ("ms_isa_scatter", C.c_double), ("ms_gather", C.c_double), ("ms_compact", C.c_double),
                ("ms_rmq_build", C.c_double), ("ms_finalize", C.c_double),
                ("ms_sort_scatter3", C.c_double), ("ms_sort_tilehist", C.c_double), ("ms_sort_scatter2", C.c_double),
                ("scatter_launches", C.c_uint64 * 3), ("scatter_records", C.c_uint64 * 3),
                ("scatter_bytes", C.c_uint64 * 3), ("hist_bytes", C.c_uint64), ("workspace_bytes", C.c_uint64), ("onew_passes", C.c_uint64),
                ("heavy_rounds", C.c_uint64), ("heavy_records", C.c_uint64), ("light_records", C.c_uint64), ("level_gathers", C.c_uint64),
                ("ms_host", C.c_double * 9)]


class PsacxError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "psacx error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Returns the ctypes handle of libpsacx.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libpsacx.so not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`"
                          % LIB_PATH)
    try:                       # let torch's bundled HIP runtime load first if torch is around
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    lib.psacx_create.argtypes = [C.POINTER(vp), i32, vp]
    lib.psacx_destroy.argtypes = [vp]
    lib.psacx_destroy.restype = None
    lib.psacx_strerror.argtypes = [i32]
    lib.psacx_strerror.restype = C.c_char_p
    lib.psacx_last_hip_error.argtypes = [vp]
    lib.psacx_last_hip_error.restype = C.c_char_p
    lib.psacx_trim.argtypes = [vp]
    lib.psacx_configure.argtypes = [vp, i32, u64]
    lib.psacx_configure_from_env.argtypes = [vp]
    lib.psacx_debug_env.argtypes = [C.c_char_p]
    lib.psacx_debug_env.restype = C.c_char_p
    for suf in ("u32", "u64"):
        for name in ("psacx_construct_", "psacx_construct_dev_"):
            getattr(lib, name + suf).argtypes = [vp, vp, u64, u32, u32, vp, vp, vp]
        for name in ("psacx_construct_gsa_", "psacx_construct_gsa_dev_"):
            getattr(lib, name + suf).argtypes = [vp, vp, u64, vp, u64, u32, u32, vp, vp, vp]
        for name in ("psacx_construct_lc_", "psacx_construct_lc_dev_"):
            getattr(lib, name + suf).argtypes = [vp, vp, u64, u32, u32, vp, vp, vp, vp]
        getattr(lib, "psacx_pair_sort_dev_" + suf).argtypes = [vp, vp, vp, vp, u64, u32]
        getattr(lib, "psacx_ansv_" + suf).argtypes = [vp, vp, u64, i32, i32, u64, vp, vp]
        getattr(lib, "psacx_ansv_dev_" + suf).argtypes = [vp, vp, u64, i32, i32, u64, vp, vp]
    lib.psacx_get_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.psacx_profile.argtypes = [vp, i32]
    lib.psacx_suffix_tree_u32.argtypes = [vp, vp, u64, vp, vp, vp, C.POINTER(C.c_uint32)]
    lib.psacx_suffix_tree_u64.argtypes = [vp, vp, u64, vp, vp, vp, C.POINTER(C.c_uint32)]
    lib.psacx_check_dev_u32.argtypes = [vp, vp, u64, vp, vp, vp, C.POINTER(C.c_uint64)]
    lib.psacx_check_dev_u64.argtypes = [vp, vp, u64, vp, vp, vp, C.POINTER(C.c_uint64)]
    lib.psacx_dev_alloc.argtypes = [vp, C.POINTER(vp), u64]
    lib.psacx_dev_free.argtypes = [vp, vp]
    lib.psacx_copy_h2d.argtypes = [vp, vp, vp, u64]
    lib.psacx_copy_d2h.argtypes = [vp, vp, vp, u64]
    lib.psacx_sync.argtypes = [vp]
    lib.psacx_rand_dna.argtypes = [vp, u64, i32]
    lib.psacx_synth_text_dev.argtypes = [vp, vp, u64, u64, i32, u64, u64]
    # several GPUs behind the same boundary (psacx_multi_*)
    lib.psacx_multi_create.argtypes = [C.POINTER(vp), i32, C.POINTER(C.c_int)]
    lib.psacx_multi_unique_id.argtypes = [vp]
    lib.psacx_multi_create_rank.argtypes = [C.POINTER(vp), i32, i32, i32, vp]
    lib.psacx_multi_destroy.argtypes = [vp]
    lib.psacx_multi_destroy.restype = None
    lib.psacx_multi_get_wire.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_double)]
    lib.psacx_multi_get_phases.argtypes = [vp, C.c_char_p, u64]
    for nm in ("psacx_multi_nranks", "psacx_multi_nlocal", "psacx_multi_uses_rccl", "psacx_multi_transport", "psacx_multi_last_form"):
        getattr(lib, nm).argtypes = [vp]
    lib.psacx_multi_last_error.argtypes = [vp]
    lib.psacx_multi_last_error.restype = C.c_char_p
    lib.psacx_multi_ctx.argtypes = [vp, i32]
    lib.psacx_multi_ctx.restype = vp
    for suf in ("u32", "u64"):
        getattr(lib, "psacx_multi_construct_dev_" + suf).argtypes = [vp, vp, vp, u32, u32, vp, vp, vp]
        getattr(lib, "psacx_multi_construct_" + suf).argtypes = [vp, vp, u64, u32, u32, vp, vp, vp]
        getattr(lib, "psacx_multi_check_dev_" + suf).argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(C.c_uint64)]
        getattr(lib, "psacx_multi_left_chars_dev_" + suf).argtypes = [vp, vp, vp, vp, vp, vp]
        getattr(lib, "psacx_multi_suffix_tree_dev_" + suf).argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(u32)]
        getattr(lib, "psacx_multi_suffix_tree_" + suf).argtypes = [vp, vp, u64, vp, vp, vp, C.POINTER(u32)]
        getattr(lib, "psacx_multi_construct_gsa_dev_" + suf).argtypes = [vp, vp, vp, vp, u64, u32, u32, vp, vp, vp]
        getattr(lib, "psacx_multi_construct_gsa_" + suf).argtypes = [vp, vp, u64, vp, u64, u32, u32, vp, vp, vp]
        getattr(lib, "psacx_multi_construct_lc_" + suf).argtypes = [vp, vp, u64, u32, u32, vp, vp, vp, vp]
        getattr(lib, "psacx_multi_ansv_dev_" + suf).argtypes = [vp, vp, vp, i32, i32, u64, vp, vp]
    lib.psacx_multi_get_stats.argtypes = [vp, C.POINTER(Stats), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    lib.psacx_multi_configure.argtypes = [vp, i32, u64]
    lib.psacx_multi_configure_from_env.argtypes = [vp]
    lib.psacx_multi_create_ex.argtypes = [C.POINTER(vp), i32, C.POINTER(C.c_int), u32]
    lib.psacx_multi_create_rank_ex.argtypes = [C.POINTER(vp), i32, i32, i32, vp, u32, u64]
    lib.psacx_multi_get_memory.argtypes = [vp, vp, C.POINTER(i32), C.POINTER(u32)]
    # step-level ops of the distributed path (include/psacx_ops.h)
    i64, u16p = C.c_int64, C.POINTER(C.c_uint16)
    u64p = C.POINTER(C.c_uint64)
    lib.psacx_op_char_hist.argtypes = [vp, vp, u64, vp]
    sig = {
        "make_keys": [vp, vp, u64, u64, u16p, u32, u32, u32, vp, vp],
        "iota": [vp, vp, u64, u64],
        "pair_sort": [vp, vp, vp, vp, vp, vp, vp, u64, u32, u32, C.POINTER(C.c_int32)],
        "put_perm": [vp, vp, vp, u64, u64, vp, vp, vp, vp, vp],
        "split_by": [vp, vp, vp, vp, u64, u64p, u64p, u64p, u64p, u32, u64, vp, vp, vp, u64p],
        "pair_bounds": [vp, vp, vp, u64, u64p, u64p, u32, i32, u64p, u64p],
        "owners": [vp, vp, u64, u64, u32, vp],
        "take": [vp, vp, vp, u64, u64, u64, vp],
        "put": [vp, vp, vp, u64, u64, vp, i64],
        "add_scalar": [vp, vp, u64, u64, u64, vp],
        "finish_b2": [vp, vp, vp, u64, u64, vp],
        "last_head": [vp, i32, vp, vp, vp, u64, u64, u32, u32, u32, vp, u64p],
        "rebucket_first": [vp, vp, vp, vp, u64, u64, u32, u32, u32, vp, vp, vp, u64p, u64p],
        "rebucket_refine": [vp, vp, vp, vp, vp, u64, u64, u64, vp, vp, vp, vp, vp, vp, vp, vp, u64p, u64p, u64p],
        "compact": [vp, vp, vp, u64, u64, u64, u64, vp, u64p],
        "block_min": [vp, vp, u64, u64p],
        "range_min": [vp, vp, u64, vp, vp, u64, u64, vp],
        "rmq_split": [vp, vp, vp, u64, u64, u32, vp, vp, vp, vp, vp, vp, vp, vp],
        "rmq_combine": [vp, vp, vp, vp, vp, u64, u64p, u32, vp],
        "lcp_apply": [vp, vp, vp, u64, u64, vp, u64],
        "nsv_from": [vp, vp, u64, u64, vp, vp, u64, i32, i32, vp, vp],
    }
    for name, args in sig.items():
        for suf in ("u32", "u64"):
            getattr(lib, "psacx_op_%s_%s" % (name, suf)).argtypes = args
    _lib = lib
    return lib
