"""ctypes binding of libwhisper_amd.so (include/whisper_amd.h).  There is no CPU fallback: if the library is
missing or fails to load, importing the product path raises."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwhisper_amd.so")

c_i32, c_i64, c_f32p, c_i32p, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p


class WxConfig(ctypes.Structure):
    _fields_ = [(n, c_i32) for n in (
        "n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
        "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer",
        "precision", "max_batch", "device",
        "tok_eot", "tok_sot", "tok_translate", "tok_transcribe", "tok_sot_prev", "tok_sot_lm",
        "tok_no_speech", "tok_no_timestamps", "tok_timestamp_begin", "tok_blank", "n_lanes")]


class WxDecodeOpts(ctypes.Structure):
    _fields_ = [(n, c_i32) for n in (
        "n_prompt", "sot_index", "max_new_tokens", "without_timestamps", "suppress_blank",
        "max_initial_timestamp_index", "use_graph", "check_interval", "profile_stride")]


# name -> (restype, argtypes); exactly the declarations of include/whisper_amd.h
PROTOTYPES = {
    "wx_last_error": (ctypes.c_char_p, []),
    "wx_version": (ctypes.c_char_p, []),
    "wx_create": (c_i32, [ctypes.POINTER(WxConfig), ctypes.POINTER(c_vp)]),
    "wx_destroy": (None, [c_vp]),
    "wx_set_weight": (c_i32, [c_vp, ctypes.c_char_p, c_vp, ctypes.POINTER(c_i64), c_i32]),
    "wx_set_mel_filters": (c_i32, [c_vp, c_vp]),
    "wx_set_suppress_tokens": (c_i32, [c_vp, c_vp, c_i32]),
    "wx_finalize": (c_i32, [c_vp]),
    "wx_create_shared": (c_i32, [c_vp, ctypes.POINTER(WxConfig), ctypes.POINTER(c_vp)]),
    "wx_pcm16_to_f32": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "wx_resample_pcm16": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "wx_mel_reset_max": (c_i32, [c_vp, c_vp, c_vp]),
    "wx_mel_stage1": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "wx_mel_stage2": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "wx_log_mel_static": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp]),
    "wx_encode": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "wx_get_encoder_output": (c_i32, [c_vp, c_i32, c_vp, c_vp]),
    "wx_reserve_cross_kv": (c_i32, [c_vp]),
    "wx_compact_windows": (c_i32, [c_vp, c_vp, c_i32, c_vp]),
    "wx_decode_greedy": (c_i32, [c_vp, c_i32, c_vp, ctypes.POINTER(WxDecodeOpts), c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(c_i32), c_vp]),
    "wx_decode_rows": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, ctypes.POINTER(WxDecodeOpts), ctypes.c_float, ctypes.c_uint32, c_vp, c_vp, c_vp,
                               c_vp, ctypes.POINTER(c_i32), c_vp]),
    "wx_decode_beam": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, ctypes.POINTER(WxDecodeOpts), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                               ctypes.POINTER(c_i32), c_vp]),
    "wx_decoder_logits": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "wx_get_profile": (c_i32, [c_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64)]),
    "wx_weight_bytes": (c_i64, [c_vp]),
    "wx_workspace_bytes": (c_i64, [c_vp]),
    "wx_cross_attention_form": (c_i32, [c_vp]),
    "wx_cross_key_splits": (c_i32, [c_vp]),
    "wx_stream_spin": (c_i32, [c_vp, c_i32]),
    "wx_stream_create_exclusive": (c_i32, [c_i32, ctypes.POINTER(c_vp)]),
    "wx_stream_destroy": (c_i32, [c_vp]),
}

_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built (run `python -m whisper_amd.build`). "
            "whisper_amd has no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own libamdhip64; this library links the system one.  Whichever HIP runtime is
    # loaded first owns the device, and the other then reports "no ROCm-capable device" -- load torch's first, always.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError here means header and library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class WxError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().wx_last_error().decode("utf-8", "replace")
        raise WxError(f"{what} failed with status {rc}: {msg}")
