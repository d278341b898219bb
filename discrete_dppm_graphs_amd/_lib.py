"""ctypes binding of libdppm_hip.so (the C-ABI of include/dppm_hip.h, plus the measurement / test hooks of include/dppm_hip_debug.h).

The library is built in-tree by `make -C discrete_dppm_graphs_amd/csrc` (or `__graft_entry__.build()`).
There is NO fallback: if the shared object is missing or a symbol cannot be resolved, importing this module
raises, and every op in `discrete_dppm_graphs_amd.ops` raises RuntimeError on a non-zero status.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdppm_hip.so")

ABI_VERSION = 5

# DPPM_PREC_* of include/dppm_hip.h
PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16": 2, "f32x6": 3}

c_f32p = C.c_void_p     # device pointers travel as integers
c_f64p = C.c_void_p
c_stream = C.c_void_p


class PpgnLayer(C.Structure):
    """mirror of dppm_ppgn_layer"""
    _fields_ = [(n, C.c_void_p) for n in
                ("m1_w0", "m1_b0", "m1_w1", "m1_b1", "m2_w0", "m2_b0", "m2_w1", "m2_b1", "m4_w", "m4_b")]


class PpgnParams(C.Structure):
    """mirror of dppm_ppgn_params"""
    _fields_ = [("num_layers", C.c_int32), ("hidden", C.c_int32),
                ("in_w", C.c_void_p), ("in_b", C.c_void_p),
                ("cat_w", C.c_void_p), ("cat_b", C.c_void_p),
                ("after_w", C.c_void_p), ("after_b", C.c_void_p),
                ("final_w", C.c_void_p), ("final_b", C.c_void_p),
                ("layers_host", C.POINTER(PpgnLayer)), ("layer_cond", C.c_void_p)]


_RNG = [c_f32p, C.c_uint64, C.c_uint32, C.c_uint32]   # U, seed, graph_id0, offset

SIGNATURES = {
    "dppm_last_error": (C.c_char_p, []),
    "dppm_abi_version": (C.c_int, []),
    "dppm_generate_mask": (C.c_int, [c_f32p, C.c_int, C.c_int, c_f32p, c_stream]),
    "dppm_edgeflip_fwd": (C.c_int, [c_f32p, c_f32p, c_f64p, C.c_double] + _RNG + [C.c_int, C.c_int, c_f32p, c_f32p, c_stream]),
    "dppm_init_xT": (C.c_int, [c_f32p] + _RNG + [C.c_int, C.c_int, c_f32p, c_stream]),
    "dppm_reverse_step_simple": (C.c_int, [c_f32p, c_f32p, C.c_int, c_f32p] + _RNG + [C.c_int, C.c_int, c_f32p, c_stream]),
    "dppm_reverse_step_vlb": (C.c_int, [c_f32p, c_f32p, C.c_int, c_f32p, C.c_double, C.c_double, C.c_double] + _RNG
                              + [C.c_int, C.c_int, c_f32p, c_stream]),
    "dppm_loss_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "dppm_loss_bce_fwd_bwd": (C.c_int, [c_f32p, c_f32p, c_f64p, c_f32p, C.c_int, C.c_int, C.c_int, c_f64p, c_f32p,
                                        C.c_void_p, C.c_size_t, c_stream]),
    "dppm_loss_bce_fwd_bwd_noalloc": (C.c_int, [c_f32p, c_f32p, c_f64p, c_f32p, C.c_int, C.c_int, C.c_int, c_f64p, c_f32p, c_stream]),
    "dppm_loss_vlb_fwd_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f64p, c_f32p, C.c_int, C.c_int, c_f32p, c_f32p,
                                        C.c_void_p, C.c_size_t, c_stream]),
    "dppm_ppgn_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "dppm_ppgn_forward": (C.c_int, [C.POINTER(PpgnParams), c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, c_f32p, C.c_void_p,
                                    C.c_size_t, C.c_int, C.c_int, c_stream]),
    "dppm_ppgn_backward": (C.c_int, [C.POINTER(PpgnParams), C.POINTER(PpgnParams), c_f32p, c_f32p, c_f32p, C.c_int, C.c_int,
                                     c_f32p, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "dppm_ppgn_chain_supported": (C.c_int, [C.POINTER(PpgnParams), C.c_int]),
    "dppm_ppgn_chain_simple": (C.c_int, [C.POINTER(PpgnParams), c_f32p, c_f32p, c_f32p, C.c_int, C.c_uint64, C.c_uint32, C.c_int, C.c_int,
                                         c_f32p, C.c_int, c_stream]),
    "dppm_ppgn_chain_vlb": (C.c_int, [C.POINTER(PpgnParams), c_f32p, c_f32p, c_f32p, C.c_int, C.c_uint64, C.c_uint32, C.c_int, C.c_int,
                                      c_f32p, C.c_int, c_stream]),
    "dppm_plane_matmul": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p,
                                    C.c_int, C.c_int, c_stream]),
    "dppm_linear_fwd": (C.c_int, [c_f32p, C.c_int, C.c_int, c_f32p, C.c_long, C.c_int, c_f32p, C.c_int, c_f32p, c_f32p, C.c_long,
                                  C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "dppm_mlp_pair_fwd": (C.c_int, [c_f32p, C.c_long, C.POINTER(PpgnLayer), C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int,
                                    C.c_int, C.c_int, c_stream]),
    "dppm_linear_wgrad_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "dppm_linear_wgrad": (C.c_int, [c_f32p, C.c_long, c_f32p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int,
                                    c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "dppm_inorm_fwd": (C.c_int, [c_f32p, C.c_long, c_f32p, c_f32p, C.c_long, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_stream]),
    "dppm_graph_stats": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, c_f64p, c_f32p, c_f32p, c_stream]),
    "dppm_mmd_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "dppm_mmd_gaussian_emd": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, C.c_int, C.c_double, C.c_double, c_f64p, C.c_void_p,
                                        C.c_size_t, c_stream]),
    "dppm_test_fill_lds": (C.c_int, [C.c_uint32, c_stream]),
    "dppm_prof_begin": (C.c_int, [C.c_int]),
    "dppm_prof_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_long), C.c_int, C.c_int]),
    "dppm_debug_set": (C.c_int, [C.c_char_p, C.c_int]),
    "dppm_debug_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "dppm_adam_step": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_float, C.c_int, C.c_float, c_stream]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(`make -C discrete_dppm_graphs_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU or PyTorch fallback for the DDPM hot path.")
    # torch first: it ships its own HIP runtime (torch/lib/libamdhip64.so); loading ours against /opt/rocm's copy before torch's is in
    # the process leaves two runtimes side by side, and the second one reports "no ROCm-capable device"
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.dppm_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libdppm_hip.so ABI {got} != expected {ABI_VERSION}; rebuild the extension")
    return lib


lib = _load()


def check(status, what):
    if status != 0:
        msg = lib.dppm_last_error()
        raise RuntimeError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")
