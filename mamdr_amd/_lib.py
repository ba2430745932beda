"""ctypes binding of libmamdr_hip.so (C ABI: include/mamdr_hip.h).

There is no CPU fallback: if the shared library is missing or a symbol is absent
the import of the product path fails loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MAMDR_LIB_PATH") or os.path.join(HERE, "libmamdr_hip.so")      # (MAMDR_LIB_PATH: A/B of diagnostic builds, tools/build_variant.sh)

ABI_VERSION = 18
OK, EINVAL, ESTATE, EHIP, ENOTBUILT = 0, -1, -2, -3, -4
TOWER_MLP, TOWER_DEEPFM, TOWER_STAR, TOWER_WDL, TOWER_PNN, TOWER_NFM = 0, 1, 2, 3, 4, 5
SPLIT_TRAIN, SPLIT_VAL, SPLIT_TEST = 0, 1, 2
OPT_ADAM, OPT_SGD, OPT_ACCUMULATE = 0, 1, 2
MERGE_PLUS, MERGE_TIMES = 0, 1
(SEG_USER_EMB, SEG_ITEM_EMB, SEG_DOMAIN_EMB, SEG_W0, SEG_W1, SEG_W2, SEG_B0, SEG_B1, SEG_B2, SEG_WO,
 SEG_GB, SEG_LIN_USER, SEG_LIN_ITEM, SEG_LIN_DOMAIN) = range(14)
SEG_NAMES = ("user_emb", "item_emb", "domain_emb", "W0", "W1", "W2", "b0", "b1", "b2", "wo", "gb",
             "lin_user", "lin_item", "lin_domain",
             # Star tower
             "Ws0", "Ws1", "Ws2", "bs0", "bs1", "bs2", "pn_gamma_shared", "pn_beta_shared", "pn_gamma_spec",
             "pn_beta_spec", "Wd0", "Wd1", "Wd2", "bd0", "bd1", "bd2",
             "log_var",
             "W0x")          # PNN: the inner products' three rows of the first kernel
KERNEL_FWD_BWD, KERNEL_WGRAD, KERNEL_UPDATE, KERNEL_EVAL, KERNEL_GATHER, KERNEL_EMB_SWEEP, KERNEL_AUX, KERNEL_FLUSH = range(8)
KERNEL_NAMES = ("k_tower<train>", "k_wgrad", "k_update", "k_tower<eval>", "k_gather", "k_emb_sweep",
                "other launches of a step (include/mamdr_hip.h: MAMDR_KERNEL_AUX)", "k_emb_flush")


class MamdrError(RuntimeError):
    def __init__(self, code, text):
        RuntimeError.__init__(self, "libmamdr_hip error %d: %s" % (code, text))
        self.code = code


class NotBuiltError(MamdrError, NotImplementedError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("tower", C.c_int32), ("n_user", C.c_int32), ("n_item", C.c_int32),
        ("n_domain", C.c_int32), ("emb_dim", C.c_int32), ("hidden", C.c_int32 * 3), ("max_batch", C.c_int32),
        ("emb_trainable", C.c_int32), ("dropout", C.c_float), ("l2_emb", C.c_float), ("l2_linear", C.c_float),
        ("adam_beta1", C.c_float), ("adam_beta2", C.c_float), ("adam_eps", C.c_float),
        ("uncertainty_weight", C.c_int32),
    ]


GRAPH_SHARED_BOTTOM, GRAPH_MMOE, GRAPH_PLE, GRAPH_NFM, GRAPH_PNN, GRAPH_CCPM, GRAPH_AUTOINT = 0, 1, 2, 3, 4, 5, 6
GRAPH_MLP, GRAPH_WDL, GRAPH_DEEPFM = 7, 8, 9       # the step kernels' towers with any hidden_dim of 1..4 layers


class GraphConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("kind", C.c_int32), ("n_user", C.c_int32), ("n_item", C.c_int32),
        ("n_domain", C.c_int32), ("emb_dim", C.c_int32), ("max_batch", C.c_int32), ("emb_trainable", C.c_int32),
        ("n_expert_hidden", C.c_int32), ("expert_hidden", C.c_int32 * 4),
        ("n_tower_hidden", C.c_int32), ("tower_hidden", C.c_int32 * 4),
        ("n_gate_hidden", C.c_int32), ("gate_hidden", C.c_int32 * 4),
        ("num_experts", C.c_int32), ("shared_expert_num", C.c_int32), ("specific_expert_num", C.c_int32),
        ("dropout", C.c_float), ("l2_emb", C.c_float), ("adam_beta1", C.c_float), ("adam_beta2", C.c_float),
        ("adam_eps", C.c_float), ("l2_linear", C.c_float), ("uncertainty_weight", C.c_int32),
    ]


_VP, _I32, _I64, _U32, _U64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float

# name -> (restype, argtypes); every symbol include/mamdr_hip.h declares
SIGNATURES = {
    "mamdr_last_error": (C.c_char_p, []),
    "mamdr_abi_version": (C.c_int, []),
    "mamdr_env_switches": (C.c_char_p, []),
    "mamdr_env_unknown": (C.c_int, []),
    "mamdr_create": (C.c_int, [C.POINTER(Config), _VP, C.POINTER(_VP)]),
    "mamdr_destroy": (C.c_int, [_VP]),
    "mamdr_param_count": (_I64, [_VP]),
    "mamdr_param_segment": (C.c_int, [_VP, C.c_int, C.POINTER(_I64), C.POINTER(_I64)]),
    "mamdr_meta_count": (_I64, [_VP]),
    "mamdr_aux_count": (_I64, [_VP]),
    "mamdr_bind_aux": (C.c_int, [_VP, _VP]),
    "mamdr_bind_state": (C.c_int, [_VP, _VP, _VP, _VP]),
    "mamdr_optimizer_reset": (C.c_int, [_VP]),
    "mamdr_optimizer_steps": (_I64, [_VP]),
    "mamdr_set_counters": (C.c_int, [_VP, _I64, _I64]),
    "mamdr_table_flushes": (_I64, [_VP, _I32]),
    "mamdr_pregather_passes": (_I32, [_VP, _I32, _VP, _VP, _VP, _I32]),
    "mamdr_pregather_hits": (_I64, [_VP]),
    "mamdr_pregather_launches": (_I64, [_VP]),
    "mamdr_sync_tables": (C.c_int, [_VP]),
    "mamdr_bind_accumulator": (C.c_int, [_VP, _VP]),
    "mamdr_bind_table": (C.c_int, [_VP, C.c_int, _VP, _I64]),
    "mamdr_bind_domain_data": (C.c_int, [_VP, C.c_int, C.c_int, _VP, _VP, _VP, _VP, _I64]),
    "mamdr_train_steps": (C.c_int, [_VP, C.c_int, _VP, _I64, _I64, _I32, _U32, _I32, _F, _VP]),
    "mamdr_train_steps_n": (C.c_int, [_VP, C.c_int, _VP, _I64, _I64, _I64, _I32, _U32, _I32, _F, _VP]),
    "mamdr_eval_domain": (C.c_int, [_VP, C.c_int, C.c_int, _I32, _VP, _VP, _VP]),
    "mamdr_gather_rows": (C.c_int, [_VP, C.c_int, C.c_int, _VP, _I64, _I64, _VP]),
    "mamdr_interp": (C.c_int, [_VP, _VP, _VP, _F, _I64, _VP]),
    "mamdr_moving_average": (C.c_int, [_VP, _VP, _VP, _F, _F, _I64, _VP]),
    "mamdr_merge": (C.c_int, [_VP, _VP, _VP, _I32, _I64, _VP]),
    "mamdr_dr_advance": (C.c_int, [_VP, _VP, _VP, _VP, _F, _I32, _I32, _I64, _VP]),
    "mamdr_dr_advance_live": (C.c_int, [_VP, _VP, _VP, _VP, _F, _I32, _I32, _I64, _I64]),
    "mamdr_sub": (C.c_int, [_VP, _VP, _VP, _I64, _VP]),
    "mamdr_accumulate": (C.c_int, [_VP, _VP, _VP, _VP, _F, _I64, _VP]),
    "mamdr_apply_accumulated": (C.c_int, [_VP, _VP, _F, _F, _I64, _VP]),
    "mamdr_adam_apply": (C.c_int, [_VP, _VP, _VP, _VP, _F, _F, _F, _F, _F, _F, _F, _I64, _VP]),
    "mamdr_copy": (C.c_int, [_VP, _VP, _I64, _VP]),
    "mamdr_pcgrad_project": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _I32, _VP]),
    "mamdr_shuffle_perm": (C.c_int, [_I64, _I64, _U64, _VP]),
    "mamdr_shuffle_perms": (C.c_int, [_I32, _VP, _I64, _VP, _VP]),
    "mamdr_step_path": (C.c_int, [_VP, _I32]),
    "mamdr_dropout_steps": (_I64, [_VP]),
    "mamdr_set_tower_tile": (C.c_int, [_VP, _I32]),
    "mamdr_tower_tile": (C.c_int, [_VP, _I32]),
    "mamdr_profile_enable": (C.c_int, [_VP, _I32]),
    "mamdr_profile_reset": (C.c_int, [_VP]),
    "mamdr_profile_read": (C.c_int, [_VP, _I32, C.POINTER(C.c_double), C.POINTER(_I64)]),
    # towers built from generic dense layers (shared_bottom / mmoe / ple)
    "mamdr_graph_last_error": (C.c_char_p, []),
    "mamdr_graph_create": (C.c_int, [C.POINTER(GraphConfig), _VP, C.POINTER(_VP)]),
    "mamdr_graph_destroy": (C.c_int, [_VP]),
    "mamdr_graph_param_count": (_I64, [_VP]),
    "mamdr_graph_tensor_count": (_I32, [_VP]),
    "mamdr_graph_tensor_info": (C.c_int, [_VP, _I32, C.c_char_p, _I32, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64)]),
    "mamdr_graph_task_ranges": (C.c_int, [_VP, C.c_int, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64)]),
    "mamdr_graph_bind_state": (C.c_int, [_VP, _VP, _VP, _VP]),
    "mamdr_graph_optimizer_reset": (C.c_int, [_VP]),
    "mamdr_graph_set_adam_eps": (C.c_int, [_VP, _F]),
    "mamdr_graph_launch_count": (_I64, []),
    "mamdr_graph_optimizer_steps": (_I64, [_VP]),
    "mamdr_graph_dropout_steps": (_I64, [_VP]),
    "mamdr_graph_set_counters": (C.c_int, [_VP, _I64, _I64]),
    "mamdr_graph_bind_table": (C.c_int, [_VP, C.c_int, _VP, _I64]),
    "mamdr_graph_bind_domain_data": (C.c_int, [_VP, C.c_int, C.c_int, _VP, _VP, _VP, _VP, _I64]),
    "mamdr_graph_train_steps": (C.c_int, [_VP, C.c_int, _VP, _I64, _I64, _I32, _U32, _I32, _F, _VP]),
    "mamdr_graph_train_steps_n": (C.c_int, [_VP, C.c_int, _VP, _I64, _I64, _I64, _I32, _U32, _I32, _F, _VP]),
    "mamdr_graph_bind_accumulator": (C.c_int, [_VP, _VP]),
    "mamdr_graph_eval_domain": (C.c_int, [_VP, C.c_int, C.c_int, _I32, _VP, _VP, _VP]),
}

_lib = None


def load():
    """dlopen the library and type every entry point. Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it carries its own libamdhip64; loading ours afterwards binds to that
    # same runtime instance, which sharing streams and device pointers with torch requires
    # (two HIP runtimes in one process -> "no ROCm-capable device is detected").
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the MAMDR hot path)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.mamdr_abi_version() != ABI_VERSION:
        raise ImportError("libmamdr_hip.so ABI %d != binding ABI %d" % (lib.mamdr_abi_version(), ABI_VERSION))
    lib.mamdr_env_unknown()          # a MAMDR_* name nobody reads is reported on stderr (once per process)
    _lib = lib
    return lib


def env_switches():
    """[(name, who reads it, effect)] -- the library's table of environment switches (csrc/env_registry.h)."""
    rows = load().mamdr_env_switches().decode("utf-8").strip().split("\n")
    return [tuple(r.split("\t")) for r in rows]


def check(code, graph=False):
    if code != OK:
        text = (load().mamdr_graph_last_error() if graph else load().mamdr_last_error()).decode("utf-8", "replace")
        if code == ENOTBUILT:
            raise NotBuiltError(code, text)
        raise MamdrError(code, text)
    return code
