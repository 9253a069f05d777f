"""ctypes binding of libvdr.so (include/vdr.h).  No fallback: if the HIP library is missing or a
symbol is absent this module raises — there is no CPU path behind the boundary."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvdr.so")

VDR_F32, VDR_BF16, VDR_F64, VDR_I16, VDR_U8 = 0, 1, 2, 3, 4
ACT_GELU, ACT_SWIGLU = 0, 1
OUT_CLS, OUT_DENSE, OUT_PATCH_EMBED, OUT_TOKENS, OUT_ENCODER = 0, 1, 2, 3, 4
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_SWIGLU = 0, 1, 2, 3
K_COUNT = 11


class VdrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libvdr error {code}: {msg}")
        self.code = code


class vdr_config(C.Structure):
    _fields_ = [("img", C.c_int32), ("patch", C.c_int32), ("in_chans", C.c_int32), ("dim", C.c_int32),
                ("heads", C.c_int32), ("layers", C.c_int32), ("mlp_hidden", C.c_int32), ("act", C.c_int32),
                ("pre_ln", C.c_int32), ("layerscale", C.c_int32), ("has_cls", C.c_int32), ("has_pos", C.c_int32),
                ("input_ln", C.c_int32), ("ln_eps", C.c_float), ("micro_batch", C.c_int32),
                ("streams", C.c_int32), ("window", C.c_int32),
                ("global_mask", C.c_int32), ("neck_chans", C.c_int32), ("fp8", C.c_int32), ("no_ln_fold", C.c_int32),
                ("full_last_block", C.c_int32), ("fp8_cls_bf16", C.c_int32), ("resid_fp32", C.c_int32),
                ("ln_fin_fused", C.c_int32)]


# every symbol include/vdr.h declares: name -> (restype, argtypes)
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
SYMBOLS = {
    "vdr_abi_version": (_I, []),
    "vdr_tuning_build": (_I, []),
    "vdr_device_count": (_I, []),
    "vdr_create": (_I, [C.POINTER(vdr_config), _I, C.POINTER(_P)]),
    "vdr_destroy": (None, [_P]),
    "vdr_last_error": (C.c_char_p, [_P]),
    "vdr_set_weight": (_I, [_P, C.c_char_p, _P, C.POINTER(_L), _I]),
    "vdr_finalize": (_I, [_P]),
    "vdr_num_weights": (_I, [_P]),
    "vdr_weight_name": (C.c_char_p, [_P, _I]),
    "vdr_workspace_bytes": (_I, [_P, _I, _I, C.POINTER(C.c_size_t)]),
    "vdr_forward": (_I, [_P, _P, _I, _I, _P, _I, _I, _P, C.c_size_t, _P]),
    "vdr_forward_tokens": (_I, [_P, _P, _I, _I, _I, _P, _I, _I, _P, C.c_size_t, _P]),
    "vdr_forward_tokens_varlen": (_I, [_P, _P, _I, _I, _I, _P, _P, _I, _I, _P, C.c_size_t, _P]),
    "vdr_op_layernorm": (_I, [_P, _I, _P, _I, _P, _P, _L, _I, _F, _P]),
    "vdr_op_linear": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P]),
    "vdr_op_pack_linear_weight": (_I, [_P, _I, _I, _P, _P]),
    "vdr_op_linear_packed": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P]),
    "vdr_prepare_scratch_bytes": (C.c_size_t, [_I, _I, _I, _I, _I]),
    "vdr_op_prepare_image": (_I, [_P, _I, _I, _I, _I, _I, _L, _L, _L, _L, _I, _I, _P, _I, _P, _P]),
    "vdr_op_window_ct": (_I, [_P, _I, _L, C.c_double, C.c_double, _P, _P]),
    "vdr_op_hu_to_rgb": (_I, [_P, _I, _L, _P, _P]),
    "vdr_op_crop_hwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "vdr_op_voxel_sequence": (_I, [_P, _P, _P, _P, _L, _I, _P, _I, _P]),
    "vdr_affine_cubic_scratch_bytes": (C.c_size_t, [_I, _I, _L]),
    "vdr_op_affine_cubic": (_I, [_P, _I, _I, _I, _L, C.POINTER(C.c_double), C.POINTER(C.c_double), _P, _I, _P, _P]),
    "vdr_mx_scale_bytes": (C.c_size_t, [_L, _I]),
    "vdr_op_mx_quantize": (_I, [_P, _L, _I, _P, _P, _P]),
    "vdr_op_mx_dequantize": (_I, [_P, _P, _L, _I, _P, _P]),
    "vdr_op_layernorm_mx": (_I, [_P, _P, _P, _F, _L, _I, _P, _P, _P]),
    "vdr_op_linear_mx": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P]),
    "vdr_op_attention": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "vdr_op_attention_relpos": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vdr_op_patch_embed": (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "vdr_profile_enable": (_I, [_P, _I]),
    "vdr_profile_mask": (_I, [_P, C.c_uint32]),
    "vdr_profile_read": (_I, [_P, C.POINTER(C.c_double), C.POINTER(_L), C.POINTER(C.c_double),
                              C.POINTER(C.c_double), _I]),
    "vdr_kernel_class_name": (C.c_char_p, [_I]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen libvdr.so and bind every declared symbol.  torch is imported first so that the HIP
    runtime already mapped by PyTorch-ROCm (same SONAME libamdhip64.so.7) is the one libvdr uses:
    streams and device pointers are then directly exchangeable."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (maps libamdhip64 before libvdr asks for it)
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C vit-deep-radiomics_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.vdr_abi_version() != 8:
        raise ImportError("libvdr ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, handle=None):
    if rc != 0:
        msg = load().vdr_last_error(handle)
        raise VdrError(rc, msg.decode() if msg else "?")


def source_id() -> str:
    """sha256 (first 16 hex digits) over the kernel sources under csrc/ (*.hip, *.h, Makefile, sorted by name): identifies
    the kernels a committed profile was taken with, independent of when or where libvdr.so was built."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(csrc, "Makefile")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
