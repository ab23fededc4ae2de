"""ctypes binding of libdiffsci_hip.so (the C ABI in include/diffsci_hip.h).

There is deliberately no fallback: if the library cannot be loaded, every compute entry
point of diffsci_amd raises.  Build it with ``python build.py`` (hipcc, gfx950).
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_longlong, c_size_t, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIFFSCI_HIP_LIB") or os.path.join(_HERE, "_lib", "libdiffsci_hip.so")

ABI_VERSION = 4
DS_IN_NETWORK, DS_IN_SCORE, DS_IN_DRIFT, DS_IN_FLOW = 0, 1, 2, 3
DS_LOAD_PLAIN, DS_LOAD_MAXPOOL2, DS_LOAD_UPSAMPLE2, DS_LOAD_AVGPOOL2 = 0, 1, 2, 3
DS_PAD_CIRCULAR = 16
DS_RES1_UPSAMPLED = 32


class EvalCoef(Structure):
    """struct ds_eval_coef."""
    _fields_ = [("c_out", c_float), ("c_skip", c_float), ("sigma_sq", c_float),
                ("neg_mult", c_float), ("neg_lang", c_float), ("guidance", c_float),
                ("one_minus_guidance", c_float), ("input_kind", c_int), ("stochastic", c_int),
                ("scaled", c_int), ("scale", c_float), ("scale_mult", c_float), ("next_scale", c_float), ("xin_copies", c_int),
                ("nonfinite", c_void_p)]


class NativeLibraryError(RuntimeError):
    pass


_P = c_void_p
_PROTOS = {
    "ds_version": (c_int, []),
    "ds_last_error": (c_char_p, []),
    "ds_device_info": (c_int, [POINTER(c_int), POINTER(c_int), ctypes.c_char_p, c_int]),
    "ds_karras_scale": (c_int, [_P, _P, c_float, c_size_t, _P]),
    "ds_karras_drift": (c_int, [_P, _P, _P, _P, POINTER(EvalCoef), c_size_t, _P]),
    "ds_karras_score": (c_int, [_P, _P, _P, _P, POINTER(EvalCoef), c_size_t, _P]),
    "ds_philox_normal": (c_int, [_P, _P, c_uint64, c_size_t, _P]),
    "ds_karras_euler": (c_int, [_P, _P, _P, _P, _P, POINTER(EvalCoef), c_float, c_float,
                                _P, _P, c_uint64, c_float, c_float, c_size_t, _P]),
    "ds_karras_heun": (c_int, [_P, _P, _P, _P, _P, POINTER(EvalCoef), _P, _P, POINTER(EvalCoef),
                               c_float, c_float, c_size_t, _P]),
    "ds_karras_churn": (c_int, [_P, _P, _P, _P, _P, c_uint64, c_float, c_float, c_float, c_float, c_int, c_size_t, _P]),
    "ds_karras_denoiser": (c_int, [_P, _P, _P, _P, c_float, c_float, _P, _P, c_int, c_size_t, _P]),
    "ds_inorm_silu": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, c_int, _P]),
    "ds_conv2d_packed_floats": (c_size_t, [c_int, c_int, c_int]),
    "ds_conv2d_pack_weights": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "ds_conv2d": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int,
                          c_int, c_int, _P]),
    "ds_conv2d_x6_packed_bytes": (c_size_t, [c_int, c_int]),
    "ds_conv2d_x6_pack_weights": (c_int, [_P, _P, c_int, c_int, _P]),
    "ds_conv2d_x6": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "ds_conv2d_h3_packed_bytes": (c_size_t, [c_int, c_int]),
    "ds_conv2d_h3_pack_weights": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "ds_conv2d_h3": (c_int, [_P, _P, _P, c_int, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "ds_absmax_rows": (c_int, [_P, _P, c_int, c_size_t, c_size_t, _P]),
    "ds_amax_merge": (c_int, [_P, _P, _P, c_int, _P]),
    "ds_absmax_channels": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_size_t, c_int, _P]),
    "ds_input_amax": (c_int, [_P, c_int, c_int, _P, _P, _P, c_int, c_int, c_size_t, c_int, _P]),
    "ds_fill_u32": (c_int, [_P, c_uint32, c_size_t, _P]),
    "ds_conv2d_h3_up_supported": (c_int, [c_int, c_int]),
    "ds_conv2d_h3_up_packed_bytes": (c_size_t, [c_int, c_int]),
    "ds_conv2d_h3_up_pack_weights": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "ds_conv2d_h3_up": (c_int, [_P, _P, _P, c_int, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "ds_table_apply_images": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "ds_conv2d_h3_up_img": (c_int, [_P, _P, _P, c_int, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "ds_conv_tile_count": (c_int, [c_int, c_int]),
    "ds_inorm_table": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    "ds_gnorm1_table": (c_int, [_P, _P, c_int, c_int, _P, c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_longlong,
                                c_float, c_int, _P]),
    "ds_gnorm1_stats_tiles": (c_int, [_P, _P, c_int, c_int, _P, c_int, c_int, c_int, c_longlong, c_float, c_int, _P]),
    "ds_conv2d_direct": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "ds_conv3d_direct": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "ds_volume_to_slices": (c_int, [_P, _P, c_int, c_int, c_int, c_size_t, c_int, c_int, c_int, _P]),
    "ds_slices_to_volume": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_size_t, c_int, _P]),
    "ds_conv_images_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ds_inorm_silu_images_supported": (c_int, [c_int, c_int]),
    "ds_inorm_silu_images": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    "ds_conv2d_h3_img": (c_int, [_P, _P, _P, c_int, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "ds_gnorm1_apply_images": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "ds_volume_stat_tiles": (c_int, [c_int, c_size_t]),
    "ds_volume_to_slices_act": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_size_t, c_int, _P]),
    "ds_wrap_pad_slices": (c_int, [_P, c_int, c_int, c_int, c_size_t, _P]),
    "ds_slices_to_volume_stats": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_size_t, c_int, _P]),
    "ds_slice_tables": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_longlong, c_float, c_int, c_int, _P]),
    "ds_avgpool3d": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "ds_upsample3d": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "ds_conv1x1_h3_packed_bytes": (c_size_t, [c_int, c_int]),
    "ds_conv1x1_h3_pack_weights": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "ds_conv1x1_h3": (c_int, [_P, _P, _P, c_int, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P]),
    "ds_attention": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "ds_attention_generic": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "ds_attention_h3": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "ds_attention_h3_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ds_attention_h3_ws": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "ds_token_l2_normalize": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, _P]),
    "ds_linear": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "ds_fourier_features": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "ds_fourier_channels": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_size_t, _P]),
    "ds_gnorm1_workspace_bytes": (c_size_t, [c_int]),
    "ds_gnorm1_stats": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, c_int, _P]),
    "ds_gnorm1_apply": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "ds_concat2": (c_int, [_P, _P, _P, c_int, c_size_t, c_size_t, _P]),
    "ds_add_act": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "ds_mask_blend": (c_int, [_P, _P, _P, _P, c_size_t, c_int, _P]),
    "ds_axpby": (c_int, [_P, _P, c_float, _P, c_float, c_size_t, _P]),
    "ds_div_scalar": (c_int, [_P, _P, c_float, c_size_t, _P]),
    "ds_batchnorm_eval": (c_int, [_P, _P, _P, _P, _P, _P, c_float, c_float, c_int, c_int, c_int, c_int, c_size_t, _P]),
    "ds_lerp_stack": (c_int, [_P, _P, _P, c_int, c_size_t, _P]),
    "ds_add": (c_int, [_P, _P, _P, c_size_t, _P]),
    "ds_graph_begin_capture": (c_int, [_P]),
    "ds_graph_end_capture": (c_int, [_P, POINTER(_P), POINTER(c_int)]),
    "ds_graph_launch": (c_int, [_P, _P]),
    "ds_graph_destroy": (c_int, [_P]),
}

_lib = None


def exported_symbols():
    """Names this binding expects; tests compare them with include/diffsci_hip.h."""
    return sorted(_PROTOS)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: diffsci_amd has no CPU or PyTorch fallback. "
            "Run `python build.py` (needs hipcc; cross-compiles for gfx950 without a GPU).")
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _PROTOS.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}; rebuild with build.py") from e
        fn.restype = res
        fn.argtypes = args
    if L.ds_version() != ABI_VERSION:
        raise NativeLibraryError(f"ABI version {L.ds_version()} != {ABI_VERSION}; rebuild with build.py")
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = lib().ds_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (status {rc}): {msg}")
