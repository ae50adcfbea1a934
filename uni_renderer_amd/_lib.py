"""ctypes binding of ``liburhip.so`` (the C ABI declared in ``include/ur_kernels.h``).

There is deliberately no fallback: if the HIP library is missing or its ABI does not match, importing
the compute path raises.  ``import torch`` must happen before the library is loaded so that the
library's ``libamdhip64.so.7`` dependency resolves to the HIP runtime PyTorch already loaded (same
streams, same device context, same graph capture).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported first, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UR_LIB_PATH", os.path.join(_HERE, "liburhip.so"))  # override = kernel experiments only
ABI_VERSION = 12

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class IGemmDesc(C.Structure):
    """Mirror of ``ur_igemm_desc`` (include/ur_kernels.h) -- field order and types must match."""

    _fields_ = [
        ("x0", vp), ("x1", vp), ("w", vp), ("bias", vp), ("rowadd", vp), ("res", vp), ("out", vp),
        ("partial", vp), ("zero_page", vp),
        ("ldx0", i64), ("ldx1", i64), ("ldw", i64), ("ldres", i64), ("ldc", i64),
        ("zx", i64), ("zw", i64), ("zout", i64), ("zx1", i64), ("zbias", i64), ("zrow", i64), ("zres", i64),
        ("ldp", i64),
        ("c0", i32), ("c1", i32),
        ("B", i32), ("Hin", i32), ("Win", i32), ("Hout", i32), ("Wout", i32),
        ("taps", i32), ("stride", i32), ("ups", i32),
        ("M", i32), ("N", i32), ("K", i32),
        ("n_store", i32), ("ld_rowadd", i32), ("rows_per_b", i32),
        ("act", i32), ("out_scale", f32),
        ("zbatch", i32), ("splitk", i32), ("zx_div", i32), ("tile", i32), ("dtype", i32),
        ("res_lo", vp), ("out_lo", vp),
        ("cblock", i32),
        ("t0", vp), ("t1", vp), ("ldt0", i64), ("ldt1", i64), ("zt0", i64), ("zt1", i64), ("ct0", i32), ("ct1", i32),
        ("pad", i32),
        ("out_vt", vp), ("ldvt", i64), ("vt_bstride", i64), ("zvt", i64), ("vt_n0", i32), ("vt_rows", i32),
        ("zero_page_bytes", i32),
    ]


class AttnDesc(C.Structure):
    """Mirror of ``ur_attn_desc``."""

    _fields_ = [
        ("q", vp), ("k", vp), ("vt", vp), ("o", vp), ("zero_page", vp),
        ("ldq", i64), ("ldk", i64), ("ldvt", i64), ("ldo", i64), ("vt_bstride", i64),
        ("q_hstride", i64), ("k_hstride", i64),
        ("q_off", i32), ("k_off", i32),
        ("B", i32), ("H", i32), ("Tq", i32), ("Tk", i32), ("d", i32),
        ("scale", f32), ("dtype", i32),
        ("lse", vp),
    ]


class AttnBwdDesc(C.Structure):
    """Mirror of ``ur_attn_bwd_desc``."""

    _fields_ = [
        ("q", vp), ("k", vp), ("v", vp), ("o", vp), ("dout", vp), ("qt", vp), ("kt", vp), ("dot", vp),
        ("ldq", i64), ("ldk", i64), ("ldv", i64), ("ldo", i64), ("lddo", i64), ("ldqt", i64), ("ldkt", i64), ("lddot", i64),
        ("stats", vp), ("dq", vp), ("dk", vp), ("dv", vp),
        ("lddq", i64), ("lddk", i64), ("lddv", i64),
        ("part", vp),
        ("B", i32), ("H", i32), ("d", i32), ("Tq", i32), ("Tk", i32), ("Tk_rows", i32), ("has_lse", i32),
        ("scale", f32), ("dtype", i32),
    ]


class TChainDesc(C.Structure):
    """Mirror of ``ur_tchain_desc``."""

    _fields_ = [
        ("a0", vp), ("res", vp), ("res_lo", vp), ("blk", vp), ("blk_lo", vp), ("y_out", vp), ("y_out_lo", vp),
        ("out", vp), ("out_lo", vp), ("out2", vp), ("out3", vp), ("wstream", vp), ("consts", vp),
        ("z_wstream", i64), ("z_consts", i64),
        ("M", i32), ("zbatch", i32), ("mode", i32), ("dtype", i32), ("channels", i32),
        ("rows_per_b", i32), ("ld_vt", i32), ("qk_heads", i32),
        ("eps", f32),
        ("profile", vp),
    ]


# name -> (restype, argtypes): every symbol include/ur_kernels.h declares
SYMBOLS = {
    "ur_igemm": (C.c_int, [C.POINTER(IGemmDesc), vp]),
    "ur_igemm_partial_floats": (C.c_int64, [C.POINTER(IGemmDesc)]),
    "ur_groupnorm_stats": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "ur_groupnorm_apply": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp,
                                     vp, C.c_float, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "ur_groupnorm_fused": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_float,
                                     C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "ur_layernorm": (C.c_int, [vp, vp, vp, vp, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "ur_attention": (C.c_int, [C.POINTER(AttnDesc), vp]),
    "ur_add": (C.c_int, [vp, vp, C.c_float, vp, C.c_int64, C.c_int, vp]),
    "ur_add_hilo": (C.c_int, [vp, vp, vp, vp, C.c_float, vp, vp, C.c_int64, C.c_int, vp]),
    "ur_add_hilo_multi": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "ur_sizeof_add_item": (C.c_int, []),
    "ur_timestep_embedding": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, C.c_int, vp]),
    "ur_resize_nearest": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ur_nchw_to_nhwc": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]),
    "ur_nhwc_to_nchw": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "ur_ddim_update": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp,
                                 C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, vp]),
    "ur_sampler_advance": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, vp]),
    "ur_select_step_rows": (C.c_int, [vp, vp, vp, C.c_int, vp, C.c_int, vp]),
    "ur_prefetch": (C.c_int, [vp, C.c_int64, C.c_int, vp]),
    "ur_pack_conv_weight": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ur_unpack_conv_weight_grad": (C.c_int, [vp, C.c_int64, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ur_unpack_conv_weight_grad_blocks": (C.c_int, [C.c_int, C.c_int]),
    "ur_unpack_conv_weight_grad_sumsq": (C.c_int, [vp, C.c_int64, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "ur_unipc_update": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp,
                                  vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, vp]),
    "ur_transpose2d": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ur_im2col3x3_t": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int64, C.c_int, vp]),
    "ur_colsum_workspace_floats": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "ur_colsum": (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp]),
    "ur_colsum_counters": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "ur_colsum_fused": (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp]),
    "ur_pairsum_rows": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    "ur_silu_backward": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, vp]),
    "ur_geglu_forward": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, vp]),
    "ur_geglu_backward": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, C.c_int, vp]),
    "ur_groupnorm_backward": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_float,
                                        C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_int, vp]),
    "ur_groupnorm_backward_fused": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_float, C.c_int, vp, vp,
                                              C.c_int, vp]),
    "ur_layernorm_backward": (C.c_int, [vp, vp, vp, C.c_float, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp]),
    "ur_layernorm_backward_skip": (C.c_int, [vp, vp, vp, C.c_float, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp]),
    "ur_split_heads": (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    "ur_merge_heads": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int64, C.c_int, C.c_int, vp]),
    "ur_split_heads_multi": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ur_merge_heads_multi": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ur_sizeof_heads_desc": (C.c_int, []),
    "ur_colsum_multi": (C.c_int, [vp, C.c_int, vp]),
    "ur_sizeof_colsum_item": (C.c_int, []),
    "ur_softmax_rows": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int, C.c_int, vp]),
    "ur_softmax_backward_rows": (C.c_int, [vp, vp, C.c_int64, C.c_int64, C.c_int, C.c_float, C.c_int, vp]),
    "ur_attention_backward": (C.c_int, [C.POINTER(AttnBwdDesc), vp]),
    "ur_attention_backward_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "ur_attention_backward_splits": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ur_attention_backward_needs_transposes": (C.c_int, []),
    "ur_transpose2d_multi": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "ur_cast_multi": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp]),
    "ur_cast_multi_blocks": (C.c_int64, [vp, C.c_int]),
    "ur_cast_multi_sumsq": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "ur_adamw_multi": (C.c_int, [vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp, vp, vp, vp]),
    "ur_silu_forward": (C.c_int, [vp, vp, C.c_int64, C.c_int, vp]),
    "ur_resample2x": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ur_tchain": (C.c_int, [C.POINTER(TChainDesc), vp]),
    "ur_tchain_stream_bytes": (C.c_int64, [C.c_int]),
    "ur_tchain_const_floats": (C.c_int, [C.c_int]),
    "ur_sizeof_tchain_desc": (C.c_int, []),
    "ur_sizeof_transpose_desc": (C.c_int, []),
    "ur_sizeof_wgrad_desc": (C.c_int, []),
    "ur_wgrad": (C.c_int, [vp, vp]),
    "ur_wgrad_plan": (C.c_int, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "ur_wgrad_partial_floats": (C.c_int64, [vp]),
    "ur_wgrad_group": (C.c_int, [vp, vp, C.c_int, vp]),
    "ur_wgrad_group_plan": (C.c_int, [vp, vp, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "ur_abi_version": (C.c_int, []),
    "ur_build_info": (C.c_char_p, []),
    "ur_sizeof_igemm_desc": (C.c_int, []),
    "ur_has_wsconv": (C.c_int, []),
    "ur_has_pp": (C.c_int, []),
    "ur_igemm_uses_dxs": (C.c_int, [vp]),
    "ur_igemm_splitk_gn": (C.c_int, [vp, vp, vp, i64, C.c_float, C.c_int, C.c_int, vp]),
    "ur_sizeof_attn_desc": (C.c_int, []),
    "ur_sizeof_attn_bwd_desc": (C.c_int, []),
}

_lib = None


class UrLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load (once) and type the library; raise ``UrLibraryError`` loudly if it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UrLibraryError(
            f"{LIB_PATH} not found: the HIP kernels are not built. Run `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the compute path."
        )
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover - depends on the host
        raise UrLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise UrLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ur_abi_version() != ABI_VERSION:
        raise UrLibraryError(f"ABI version mismatch: library {lib.ur_abi_version()} vs binding {ABI_VERSION}")
    if (lib.ur_sizeof_igemm_desc() != C.sizeof(IGemmDesc) or lib.ur_sizeof_attn_desc() != C.sizeof(AttnDesc)
            or lib.ur_sizeof_attn_bwd_desc() != C.sizeof(AttnBwdDesc) or lib.ur_sizeof_tchain_desc() != C.sizeof(TChainDesc)):
        raise UrLibraryError("descriptor layout mismatch between include/ur_kernels.h and _lib.py")
    for fn_name, ctype in LAYOUT_CHECKS:  # mirrors declared next to their users (backward.py: ur_transpose_desc)
        _check_layout(lib, fn_name, ctype)
    _lib = lib
    return lib


LAYOUT_CHECKS: list = []


def _check_layout(lib, fn_name: str, ctype) -> None:
    if getattr(lib, fn_name)() != C.sizeof(ctype):
        raise UrLibraryError(f"descriptor layout mismatch: {fn_name}() = {getattr(lib, fn_name)()} vs sizeof({ctype.__name__}) = "
                             f"{C.sizeof(ctype)} (include/ur_kernels.h vs the ctypes mirror)")


def register_layout(fn_name: str, ctype) -> None:
    """A ctypes mirror of a descriptor struct that lives outside this module: checked against the library's sizeof when
    the library is loaded (immediately if it already is) -- the same loud failure as for the descriptors above."""
    LAYOUT_CHECKS.append((fn_name, ctype))
    if _lib is not None:
        _check_layout(_lib, fn_name, ctype)


def check(rc: int, what: str) -> None:
    if rc != 0:
        if rc == -1001:
            msg = "UR_E_BADARG (inconsistent descriptor)"
        elif rc == -1002:
            msg = "UR_E_UNSUPPORTED (shape not instantiated)"
        else:
            msg = f"hipError {-rc}"
        raise RuntimeError(f"{what} failed: {msg}")
