"""ctypes binding of libglowtts_hip.so (include/glowtts_hip.h) — the only way the Python host reaches the GPU kernels.

PyTorch is used for device memory (``tensor.data_ptr()``) and streams (``torch.cuda.current_stream()``); the
kernels themselves are the hand-written HIP in ``glow-tts-train_amd/csrc``.  There is NO fallback: if the library
is missing, or a tensor handed to an operator is not a contiguous fp32 CUDA/HIP tensor, the call raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get(
    "GLOWTTS_HIP_LIB", os.path.join(os.path.dirname(_PKG_DIR), "lib", "libglowtts_hip.so")
)

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_int64
_F = ctypes.c_float

# name -> argument ctypes (every function returns int; the trailing stream argument is appended automatically)
_SIGNATURES = {
    "glowtts_mas_path": [_P, _P, _P, _P, _I, _I, _I],
    "glowtts_mas_path_spans": [_P, _P, _P, _P, _P, _P, _I, _I, _I],
    "glowtts_mas_path_from_spans": [_P, _P, _I, _I, _I],
    "glowtts_align_logp": [_P, _P, _P, _P, _I, _I, _I, _I],
    "glowtts_align_expand_fwd": [_P, _P, _P, _I, _I, _I, _I],
    "glowtts_align_expand_bwd": [_P, _P, _P, _I, _I, _I, _I],
    "glowtts_mask_len": [_P, _P, _I, _I],
    "glowtts_keep_mask": [_P, _L, _L, _F],
    "glowtts_actnorm_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I],
    "glowtts_actnorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I],
    "glowtts_actnorm_stats": [_P, _P, _P, _P, _I, _I, _I],
    "glowtts_invconv_prepare": [_P, _P, _P, _I],
    "glowtts_invconv_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I],
    "glowtts_invconv_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I],
    "glowtts_actnorm_invconv_fwd": [_P] * 9 + [_I] * 4,
    "glowtts_actnorm_invconv_bwd": [_P] * 13 + [_I] * 4,
    "glowtts_invconv_prepare_multi": [_P, _P, _L, _I, _I],
    "glowtts_flow_boundary_fwd": [_P] * 17 + [_I] * 6,
    "glowtts_flow_boundary_bwd": [_P] * 15 + [_I] * 7,
    "glowtts_flow_boundary_bwd_reduce": [_P] * 7 + [_I] * 4,
    "glowtts_coupling_actnorm_invconv_fwd": [_P] * 11 + [_I] * 5,
    "glowtts_coupling_actnorm_invconv_bwd": [_P] * 15 + [_I] * 5,
    "glowtts_coupling_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I],
    "glowtts_coupling_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I],
    "glowtts_gate_fwd": [_P, _P, _P, _I, _I, _I],
    "glowtts_gate_bwd": [_P, _P, _P, _P, _I, _I, _I],
    "glowtts_res_skip_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I],
    "glowtts_res_skip_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I],
    "glowtts_conv_fwd": [_P, _L, _P, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I],
    "glowtts_conv_gate_fwd": [_P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _I, _I],
    "glowtts_conv_res_skip_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I],
    "glowtts_conv_gate_bwd": [_P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _I],
    "glowtts_conv_wrw2": [_P, _L, _P, _L, _P, _L, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I],
    "glowtts_conv_wrw": [_P, _L, _P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I],
    "glowtts_conv_wrw_batch": [_I, _P, _L, _P, _L, _P, _L, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I],
    "glowtts_conv_wrw1_multi": [_I, _P, _I, _I],
    "glowtts_chan_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F],
    "glowtts_chan_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I],
    "glowtts_pack_weight": [_P, _P, _P, _P, _P, _I, _I, _I],
    "glowtts_conv_split_weights": [_P, _L, _P],
    "glowtts_wino_weights": [_P, _L, _P, _I, _P, _L],
    "glowtts_split_planes": [_P, _L, _P, _I],
    "glowtts_conv_wrw_planes": [_P, _L, _L, _P, _L, _L, _P, _P, _I, _I, _I, _I, _I, _I],
    "glowtts_unpack_weight_grad": [_P, _P, _P, _P, _P, _P, _I, _I, _I],
    "glowtts_rowsum": [_P, _L, _P, _P, _I, _I, _I],
    "glowtts_pack_weight_multi": [_P, _P, _I, _I],
    "glowtts_pack_weight_planes_multi": [_P, _P, _I, _I, _P, _L, _P],
    "glowtts_unpack_weight_grad_multi": [_P, _P, _I, _I],
    "glowtts_gate_bwd_ts": [_P, _P, _P, _F, _P, _I, _I, _I],
    "glowtts_rel_attn_fwd": [_P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _I, _I, _I],
    "glowtts_rel_attn_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I],
    "glowtts_rel_attn_fwd_ex": [_P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I],
    "glowtts_rel_attn_bwd_ex": [_P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I],
    "glowtts_squeeze": [_P, _P, _P, _P, _I, _I, _I, _I],
    "glowtts_unsqueeze": [_P, _P, _P, _P, _I, _I, _I, _I],
    "glowtts_mle_fwd": [_P, _P, _P, _P, _P, _I, _I, _I],
    "glowtts_mle_bwd": [_P, _P, _P, _P, _P, _P, _P, _L],
    "glowtts_mle_loss_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I],
    "glowtts_mle_loss_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _L],
    "glowtts_duration_loss_fwd": [_P, _P, _P, _P, _I, _L],
    "glowtts_duration_loss_bwd": [_P, _P, _P, _P, _P, _L],
    "glowtts_span_logw": [_P, _P, _P, _I, _I],
    "glowtts_clip_grad_value": [_P, _L, _F, _P],
    "glowtts_adam_noam": [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _F, _F],
    "glowtts_adam_advance": [_P, _F, _F, _F],
    # whole WN stack per call (csrc/wn_stack.hip); the first argument is a HOST array of WnLayer
    "glowtts_wn_fwd": [_P, _I, _P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _I, _I],
    "glowtts_wn_bwd": [_P, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    # a whole flow block per call (csrc/wn_stack.hip); the first argument is a HOST struct glowtts_flow_block
    "glowtts_flow_block_fwd": [_P, _P, _P, _P, _P, _F] + [_P] * 9 + [_I] * 8,
    "glowtts_flow_block_bwd": [_P, _P, _P, _P, _P, _F] + [_P] * 16 + [_I] * 9 + [_P],
    # text-encoder neighbours in kernel epilogues and a whole transformer layer per call
    "glowtts_conv_fwd_act": [_P, _L, _P, _P, _P, _P, _L, _P, _L] + [_I] * 11 + [_P, _F, _P, _F],
    "glowtts_chan_layernorm_fwd_ex": [_P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _F],
    "glowtts_chan_layernorm_bwd_ex": [_P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I],
    "glowtts_chan_layernorm_fwd_act": [_P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _P, _F, _I, _I, _I, _F],
    "glowtts_chan_layernorm_bwd_act": [_P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _P, _F, _P, _P, _P, _P, _I, _I, _I],
    "glowtts_embed_fwd": [_P, _P, _F, _P, _I, _I, _I, _I],
    "glowtts_embed_bwd": [_P, _P, _F, _P, _I, _I, _I, _I],
    "glowtts_encoder_layer_fwd": [_P] * 7 + [_F] + [_P] * 12 + [_I] * 9 + [_F],
    "glowtts_encoder_layer_bwd": [_P] * 7 + [_F] + [_P] * 24 + [_I] * 9 + [_P],
    # `_io` forms (bf16 activation tensors in HBM: BASELINE configs[2]); the trailing int before the stream(s) is the io flag
    "glowtts_flow_block_fwd_io": [_P, _P, _P, _P, _P, _P, _F] + [_P] * 10 + [_I] * 9,
    "glowtts_flow_block_bwd_io": [_P, _P, _P, _P, _P, _F] + [_P] * 18 + [_I] * 10 + [_P],
    "glowtts_squeeze_io": [_P, _P, _P, _P, _I, _I, _I, _I, _I],
    "glowtts_unsqueeze_io": [_P, _P, _P, _P, _I, _I, _I, _I, _I],
    "glowtts_conv_fwd_io": [_P, _L, _P, _P, _P, _P, _L, _P, _L] + [_I] * 12,
    "glowtts_conv_gate_fwd_io": [_P, _P, _P, _P, _P, _F, _P, _P] + [_I] * 7,
    "glowtts_conv_res_skip_fwd_io": [_P] * 8 + [_I] * 5,
    "glowtts_conv_gate_bwd_io": [_P, _P, _P, _P, _P, _F, _P, _P] + [_I] * 5,
    "glowtts_res_skip_bwd_io": [_P] * 5 + [_I] * 5,
    "glowtts_actnorm_invconv_fwd_io": [_P] * 10 + [_I] * 5,
    "glowtts_actnorm_invconv_bwd_io": [_P] * 13 + [_I] * 5,
    "glowtts_coupling_fwd_io": [_P, _P, _P, _P, _P] + [_I] * 6,
    "glowtts_coupling_bwd_io": [_P] * 7 + [_I] * 6,
    "glowtts_wn_fwd_io": [_P, _I, _P, _P, _P, _P, _F, _P, _P, _P, _P] + [_I] * 6,
    "glowtts_wn_bwd_io": [_P, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P] + [_I] * 10 + [_P],
}


class WnLayer(ctypes.Structure):
    """struct glowtts_wn_layer (include/glowtts_hip.h): device pointers of one WN layer's packed weights and gradients."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("wf_in", "wb_in", "b_in", "wf_rs", "wb_rs", "b_rs", "dwp_in", "dwp_rs",
                                                 "db_in", "db_rs")]


class Wrw1Problem(ctypes.Structure):
    """struct glowtts_wrw1_problem (include/glowtts_hip.h): one 1x1 weight gradient of a multi-problem launch."""
    _fields_ = ([(n, ctypes.c_void_p) for n in ("x", "d", "d2", "mask_d", "mask_x", "dwp", "dbias")]
                + [(n, ctypes.c_long) for n in ("x_bs", "d_bs", "d2_bs")]
                + [(n, ctypes.c_int) for n in ("Cin", "M", "d_split", "reserved")])


class FlowBlock(ctypes.Structure):
    """struct glowtts_flow_block (include/glowtts_hip.h): one [ActNorm, InvConvNear, CouplingBlock] block's device pointers."""
    _fields_ = ([(n, ctypes.c_void_p) for n in (
        "logs", "bias", "w", "w_inv", "logdet_w", "wf_start", "wb_start", "b_start", "wf_end", "wb_end", "b_end",
        "dwp_start", "dwp_end", "db_start", "db_end", "dlogs", "dbias", "dw", "layers", "pack_desc", "unpack_desc",
        "pack_prefix", "dwp_all")]
                + [("dwp_floats", ctypes.c_longlong), ("n_layers", ctypes.c_int), ("n_conv", ctypes.c_int),
                   ("total_rows", ctypes.c_int), ("reserved", ctypes.c_int)])


class EncLayer(ctypes.Structure):
    """struct glowtts_enc_layer (include/glowtts_hip.h): one transformer layer's device pointers."""
    _fields_ = ([(n, ctypes.c_void_p) for n in (
        "wf_q", "wb_q", "b_q", "wf_k", "wb_k", "b_k", "wf_v", "wb_v", "b_v", "wf_o", "wb_o", "b_o",
        "wf_1", "wb_1", "b_1", "wf_2", "wb_2", "b_2", "emb_k", "emb_v", "gamma1", "beta1", "gamma2", "beta2",
        "dwp_q", "dwp_k", "dwp_v", "dwp_o", "dwp_1", "dwp_2", "db_q", "db_k", "db_v", "db_o", "db_1", "db_2",
        "demb_k", "demb_v", "dgamma1", "dbeta1", "dgamma2", "dbeta2", "pack_desc", "unpack_desc", "pack_prefix", "dwp_all")]
                + [("dwp_floats", ctypes.c_longlong), ("n_conv", ctypes.c_int), ("total_rows", ctypes.c_int),
                   ("attn_bf16", ctypes.c_int)])


EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ["glowtts_last_error", "glowtts_abi_version", "glowtts_conv_math",
                           "glowtts_conv_bind_planes", "glowtts_conv_bind_planes_ns", "glowtts_wn_fused",
                           "glowtts_set_knob", "glowtts_get_knob", "glowtts_mas_spans_supported",
                           "glowtts_conv_bind_wino", "glowtts_wino_plane_elems", "glowtts_wino_launches"])

_lib: Optional[ctypes.CDLL] = None
_fn_cache: dict = {}


class HipLibraryMissing(RuntimeError):
    pass


def library_path() -> str:
    return _LIB_PATH


def load() -> ctypes.CDLL:
    """Load (once) and type the C-ABI library.  Raises HipLibraryMissing — never falls back to another path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipLibraryMissing(
            f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  This package has no CPU or eager fallback."
        )
    lib = ctypes.CDLL(_LIB_PATH)
    lib.glowtts_last_error.restype = ctypes.c_char_p
    lib.glowtts_last_error.argtypes = []
    lib.glowtts_abi_version.restype = _I
    lib.glowtts_abi_version.argtypes = []
    lib.glowtts_conv_math.restype = _I
    lib.glowtts_conv_math.argtypes = [_I]
    lib.glowtts_wn_fused.restype = _I
    lib.glowtts_wn_fused.argtypes = [_I]
    lib.glowtts_mas_spans_supported.restype = _I
    lib.glowtts_mas_spans_supported.argtypes = [_I, _I]
    lib.glowtts_set_knob.restype = _I
    lib.glowtts_set_knob.argtypes = [ctypes.c_char_p, _I]
    lib.glowtts_get_knob.restype = _I
    lib.glowtts_get_knob.argtypes = [ctypes.c_char_p, ctypes.POINTER(_I)]
    lib.glowtts_conv_bind_planes.restype = _I
    lib.glowtts_conv_bind_planes.argtypes = [_P, _L, _P]
    lib.glowtts_conv_bind_planes_ns.restype = _I
    lib.glowtts_conv_bind_planes_ns.argtypes = [_P, _L, _P, _I]
    lib.glowtts_conv_bind_wino.restype = _I
    lib.glowtts_conv_bind_wino.argtypes = [_P, _L, _P, _L]
    lib.glowtts_wino_plane_elems.restype = _L
    lib.glowtts_wino_plane_elems.argtypes = [_L]
    lib.glowtts_wino_launches.restype = _L
    lib.glowtts_wino_launches.argtypes = []
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = _I
        fn.argtypes = list(args) + [_P]
    _lib = lib
    _bind_fastcall(lib)
    return lib


def _bind_fastcall(lib) -> None:
    """`call()` goes through `_glowtts_fastcall` when it is there (csrc/gen_fastcall.py: METH_FASTCALL wrappers around the
    same entry points, ~3 us less host time per call than ctypes' argument conversion, ~1 000 calls per step).  It is
    only a binding: its function pointers are taken from the ctypes handle opened above, and without it every call goes
    through ctypes.  GLOWTTS_FASTCALL=0 disables it."""
    global FASTCALL
    FASTCALL = False
    if os.environ.get("GLOWTTS_FASTCALL", "1") == "0":
        return
    lib_dir = os.path.join(os.path.dirname(_PKG_DIR), "lib")
    try:
        import importlib.machinery
        import importlib.util
        import sysconfig

        path = os.path.join(lib_dir, "_glowtts_fastcall" + sysconfig.get_config_var("EXT_SUFFIX"))
        if not os.path.exists(path):
            return
        loader = importlib.machinery.ExtensionFileLoader("_glowtts_fastcall", path)
        mod = importlib.util.module_from_spec(importlib.util.spec_from_loader("_glowtts_fastcall", loader))
        loader.exec_module(mod)
        if sorted(mod.NAMES) != sorted(_SIGNATURES):          # built from another _hip.py: ignore it
            return
        for i, name in enumerate(mod.NAMES):
            mod._bind(i, ctypes.cast(getattr(lib, name), ctypes.c_void_p).value)
        for name in mod.NAMES:
            _fn_cache[name] = getattr(mod, name)
        FASTCALL = True
    except Exception:                                         # a binding convenience must never stop the library loading
        _fn_cache.clear()


FASTCALL = False


def direct_apply(cls):
    """`cls.apply` without the Python wrapper of torch.autograd.Function.apply (default-argument binding for setup_context
    and functorch unwrapping, neither used by these operators): the engine's own entry point, ~4 us less per call on a path
    that makes ~130 such calls per step and is within 20 % of being host-bound."""
    return super(torch.autograd.Function, cls).apply


CONV_MATH_MODES = {"fp32": 0, "bf16": 1, "bf16x3": 2, "bf16x6": 3}


def conv_math(mode=None) -> int:
    """Arithmetic of the WN convolutions (include/glowtts_hip.h: glowtts_conv_math): "fp32" (native MFMA, default),
    "bf16", "bf16x3", "bf16x6" or the numeric code; None only queries.  Returns the mode in force BEFORE the call."""
    lib = load()
    before = lib.glowtts_conv_math(-1)
    if mode is not None:
        if isinstance(mode, str):            # "bf16x6" or "bf16x6+wrw" (the weight-gradient kernel as well)
            base, _, wrw = mode.partition("+")
            if base not in CONV_MATH_MODES or wrw not in ("", "wrw"):
                raise ValueError(f"conv math mode {mode!r}: expected one of {sorted(CONV_MATH_MODES)}, optionally with '+wrw'")
            code = CONV_MATH_MODES[base] * (5 if wrw == "wrw" else 1)
        else:
            code = int(mode)
        if lib.glowtts_conv_math(code) != 0:
            raise RuntimeError(lib.glowtts_last_error().decode())
    return before


def wn_fused(enable: Optional[bool] = None) -> bool:
    """The layer-resident WN forward kernel (include/glowtts_hip.h: glowtts_wn_fused): True / False switches it, None only
    queries.  Returns the setting in force BEFORE the call."""
    return bool(load().glowtts_wn_fused(-1 if enable is None else int(bool(enable))))


def wn_fused_launches() -> int:
    """Launches of the layer-resident WN forward kernel so far in this process."""
    return int(load().glowtts_wn_fused(-2))


def set_knob(name: str, value: int) -> None:
    """Set one of the library's tuning switches (include/glowtts_hip.h, conventions block) for the launches queued from now
    on; an unknown name raises."""
    lib = load()
    if lib.glowtts_set_knob(name.encode(), int(value)) != 0:
        raise RuntimeError(lib.glowtts_last_error().decode())


def get_knob(name: str) -> int:
    lib = load()
    out = _I(0)
    if lib.glowtts_get_knob(name.encode(), ctypes.byref(out)) != 0:
        raise RuntimeError(lib.glowtts_last_error().decode())
    return int(out.value)


def conv_bind_planes(wp: Optional[torch.Tensor], planes: Optional[torch.Tensor] = None):
    """Bind (or, with None, unbind) a packed-weight buffer's bf16 planes for this thread's next convolution launches."""
    lib = load()
    if wp is None:
        lib.glowtts_conv_bind_planes(None, 0, None)
    elif lib.glowtts_conv_bind_planes(wp.data_ptr(), wp.numel(), planes.data_ptr()) != 0:
        raise RuntimeError(lib.glowtts_last_error().decode())


def wino_launches() -> int:
    """Launches of the Winograd form of the gated in-conv (csrc/convwino.hip) by this process so far."""
    return int(load().glowtts_wino_launches())


def wino_plane_elems(n: int) -> int:
    """bf16 elements per plane of the Winograd-domain weights (csrc/convwino.hip) of a packed buffer of `n` floats."""
    return int(load().glowtts_wino_plane_elems(int(n)))


def conv_bind_wino(wp: Optional[torch.Tensor], planes: Optional[torch.Tensor] = None):
    """Bind (None: unbind) the Winograd-domain planes (3 x wino_plane_elems(n) bf16, written by glowtts_wino_weights) of a packed-weight
    buffer for this thread's next gated in-conv launches (used when the switch GLOWTTS_WINO is on)."""
    lib = load()
    if wp is None:
        lib.glowtts_conv_bind_wino(None, 0, None, 0)
    elif lib.glowtts_conv_bind_wino(wp.data_ptr(), wp.numel(), planes.data_ptr(), planes.numel() // 3) != 0:
        raise RuntimeError(lib.glowtts_last_error().decode())


def conv_bind_planes_ns(wp: Optional[torch.Tensor], planes: Optional[torch.Tensor] = None, n_planes: int = 1):
    """Bind (None: unbind) `n_planes` bf16 planes of a packed-weight buffer for this thread's next convolution launches,
    whatever glowtts_conv_math says (bf16-tensor flow blocks bind ONE plane: the weights rounded to bf16)."""
    lib = load()
    if wp is None:
        lib.glowtts_conv_bind_planes_ns(None, 0, None, 0)
    elif lib.glowtts_conv_bind_planes_ns(wp.data_ptr(), wp.numel(), planes.data_ptr(), int(n_planes)) != 0:
        raise RuntimeError(lib.glowtts_last_error().decode())


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(
            "glow_tts_train (MI355X build): operator received a CPU tensor; this package runs only on HIP devices"
        )
    if not t.is_contiguous():
        raise RuntimeError("glow_tts_train: non-contiguous tensor passed to a HIP kernel")
    return t.data_ptr()


def f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise RuntimeError(f"glow_tts_train: HIP kernels are fp32, got {t.dtype}")
    return t


# Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).  Off by
# default: the product path records nothing.
_timing: Optional[dict] = None


def enable_timing() -> None:
    global _timing
    _timing = {}


def disable_timing() -> dict:
    """Stop timing and return {name: [milliseconds per launch, ...]} (synchronises once to read the events)."""
    global _timing
    t, _timing = _timing or {}, None
    torch.cuda.synchronize()
    return {k: [a.elapsed_time(b) for a, b in v] for k, v in t.items()}


_side_streams: dict = {}


def side_stream_enabled() -> bool:
    """The text encoder may run on a second stream next to the flow decoder (models.FlowGenerator.forward), but only
    inside a training step bracketed by `zero_scope` (train.train_batch joins the stream again after backward — gradients
    the operators write straight into `.grad` never pass an AccumulateGrad node, so autograd's own end-of-backward stream
    sync does not cover them).  Off while per-launch timings are collected (overlapped kernels would time each other)
    and when GLOWTTS_SIDE_STREAM=0."""
    return _arena.active and _timing is None and os.environ.get("GLOWTTS_SIDE_STREAM", "1") != "0"


def timing_off() -> bool:
    """False while per-launch timings are collected: multi-launch native executors then yield to the per-kernel path."""
    return _timing is None


def join_side_streams() -> None:
    """Make the current stream wait for everything queued on the side streams (after backward, before the optimizer)."""
    for (_, role), s in _side_streams.items():
        if not role.startswith("fwd"):
            torch.cuda.current_stream(s.device).wait_stream(s)


def side_stream(device, role: str = "encoder", priority: int = 0) -> "torch.cuda.Stream":
    """One extra stream per (device, role): "encoder" (the text-encoder branch), "wgrad" (weight-gradient kernels) and
    "comm" (the DP reducer's collectives)."""
    if role == "fwd2" and _FWD2_ON_WGRAD:            # the decoder's second forward chain runs on the stream the weight gradients use
        role = "wgrad"                               # in the backward (idle during the forward): one stream fewer — DESIGN.md lesson 38
    key = (torch.device(device).index, role)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device, priority=priority)
    return _side_streams[key]


_FWD2_ON_WGRAD = os.environ.get("GLOWTTS_FWD2_ON_WGRAD", "1") != "0"


def all_side_streams(device):
    """The side streams gradients may come from (the forward-only chains "fwd2.." are joined inside the forward: not among them)."""
    idx = torch.device(device).index
    return [s for (d, role), s in _side_streams.items() if d == idx and not role.startswith("fwd")]


def call(name: str, *args, tag=None) -> None:
    """Launch `name` on PyTorch's current stream; raise RuntimeError with the library's message on failure.
    `tag` only labels the launch for the optional timing table: a string, or a (format, *values) tuple that is formatted
    only when timings are being collected.  This function runs ~1 100 times per training step, so the untimed path does
    nothing but look up the cached ctypes entry and the raw stream handle."""
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(load(), name)
    if _timing is None:
        rc = fn(*args, torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    else:
        cur = torch.cuda.current_stream()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(cur)
        rc = fn(*args, cur.cuda_stream)
        b.record(cur)
        if isinstance(tag, tuple):
            tag = tag[0] % tag[1:]
        _timing.setdefault(name if tag is None else f"{name}[{tag}]", []).append((a, b))
    if rc != 0:
        msg = load().glowtts_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{name} failed (code {rc}): {msg}")


def call_on(stream_ptr: int, name: str, *args) -> None:
    """`call` on an explicit raw HIP stream (no stream context switch on the host: the stack node's two forward chains)."""
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(load(), name)
    rc = fn(*args, stream_ptr)
    if rc != 0:
        msg = load().glowtts_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{name} failed (code {rc}): {msg}")


# ----------------------------------------------------------------------------------------------------------------
class _ZeroArena:
    """Zero-initialised scratch for ONE training step: accumulators the kernels add into with atomics (packed weight
    gradients, log-determinants, loss sums) are carved out of one buffer that a step clears with ONE fill instead of
    ~100 `torch.zeros` launches.  Only temporaries that die inside the step live here — never a tensor handed to
    autograd as a parameter gradient (AccumulateGrad may keep such a tensor as `param.grad`)."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None
        self.off = 0          # floats handed out in the current step
        self.want = 0         # floats requested in the current step (sizes the buffer of the next one)
        self.active = False

    def begin(self, device) -> None:
        if self.want and (self.buf is None or self.buf.device != torch.device(device) or self.want > self.buf.numel()):
            if not torch.cuda.is_current_stream_capturing():
                self.buf = torch.empty(int(self.want * 1.25) + 1024, device=device, dtype=torch.float32)
        if self.buf is not None:
            self.buf.zero_()
        self.off, self.want, self.active = 0, 0, True

    def end(self) -> None:
        self.active = False

    def zeros(self, shape, device) -> torch.Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        span = (n + 63) // 64 * 64                      # 256-byte granules keep every carve-out 16-byte aligned
        if self.active:
            self.want += span
            if self.buf is not None and self.buf.device == torch.device(device) and self.off + span <= self.buf.numel():
                out = self.buf[self.off: self.off + n].view(shape)
                self.off += span
                return out
        return torch.zeros(shape, device=device, dtype=torch.float32)


_arena = _ZeroArena()


class zero_scope:
    """`with zero_scope(device):` brackets one training step (train.train_batch); inside it `scratch_zeros` carves from
    the arena, outside it is plain `torch.zeros`."""

    def __init__(self, device):
        self.device = device

    def __enter__(self):
        _arena.begin(self.device)
        return self

    def __exit__(self, *exc):
        _arena.end()
        return False


def scratch_zeros(shape, device) -> torch.Tensor:
    return _arena.zeros(tuple(shape) if not isinstance(shape, int) else (shape,), device)
