"""ctypes binding of libsedhip.so (the C ABI declared in include/sed_hip.h).

There is no fallback: if the shared library is missing the import fails, and every compute entry point
needs a ROCm device.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SED_HIP_LIB") or os.path.join(_HERE, "libsedhip.so")  # override = diagnostic builds

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C sound-event-detection_amd/csrc` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")

lib = C.CDLL(LIB_PATH)

SED_OK = 0
SED_ERR_BAD_ARG, SED_ERR_UNSUPPORTED_NFFT, SED_ERR_CAPACITY, SED_ERR_SHORT_INPUT, SED_ERR_WRONG_DEVICE = -1, -2, -3, -4, -5
PAD_NONE, PAD_CONSTANT, PAD_REFLECT = 0, 1, 2
SPEC_MAGNITUDE, SPEC_POWER = 1, 2
MEL_SLANEY, MEL_HTK_TF, MEL_KALDI = 0, 1, 2
LOG_NONE, LOG_DB10, LOG_LN, LOG_LN_FLOOR = 0, 1, 2, 3
PCM_F32, PCM_I16 = 0, 1
WIN_CENTER, WIN_LEFT = 0, 1
WINDOW_HANN_PERIODIC, WINDOW_HANN_SYMMETRIC = 0, 1
SMOOTH_NONE, SMOOTH_MEDIAN3, SMOOTH_GAUSS25 = 0, 1, 2


class FrontendCfg(C.Structure):
    """sed_frontend_cfg."""
    _fields_ = [
        ("sample_rate", C.c_int32), ("n_fft", C.c_int32), ("win_length", C.c_int32), ("hop", C.c_int32),
        ("center", C.c_int32), ("pad_mode", C.c_int32), ("win_align", C.c_int32),
        ("tail_min_samples", C.c_int32), ("tail_hop_samples", C.c_int32),
        ("spectrum", C.c_int32), ("mel_kind", C.c_int32), ("n_mels", C.c_int32),
        ("fmin", C.c_double), ("fmax", C.c_double),
        ("log_kind", C.c_int32), ("amin", C.c_float), ("ref", C.c_float), ("ln_offset", C.c_float),
        ("pcm_kind", C.c_int32), ("pcm_divisor", C.c_float),
        ("window_kind", C.c_int32), ("remove_dc", C.c_int32), ("preemphasis", C.c_float),
    ]


_vp, _i32, _i64, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double

# name -> (restype, argtypes); every symbol declared in include/sed_hip.h
SIGNATURES = {
    "sed_version": (C.c_char_p, []),
    "sed_status_string": (C.c_char_p, [C.c_int]),
    "sed_mel_filterbank": (C.c_int, [_i32, _i32, _i32, _i32, _f64, _f64, _vp]),
    "sed_hann_window": (C.c_int, [_i32, _i32, _i32, _vp]),
    "sed_window": (C.c_int, [_i32, _i32, _i32, _i32, _vp]),
    "sed_plan_create": (C.c_int, [C.POINTER(FrontendCfg), C.POINTER(_vp)]),
    "sed_plan_destroy": (None, [_vp]),
    "sed_plan_num_frames": (_i64, [_vp, _i64]),
    "sed_plan_get_mel_weights": (C.c_int, [_vp, _vp]),
    "sed_plan_get_window": (C.c_int, [_vp, _vp]),
    "sed_plan_set_mel_weights": (C.c_int, [_vp, _vp]),
    "sed_plan_set_window": (C.c_int, [_vp, _vp]),
    "sed_plan_set_affine": (C.c_int, [_vp, _vp, _vp]),
    "sed_logmel_forward": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "sed_logmel_forward_segments": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp]),
    "sed_logmel_forward_segments_own": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp]),
    "sed_spectrogram_forward": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "sed_mel_from_spectrogram": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "sed_db_normalize": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _vp]),
    "sed_db_normalize_static": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _vp]),
    "sed_threshold_runs": (C.c_int, [_vp, _i64, _i64, _i64, _f32, _i32, _vp, _vp, _i32, _vp]),
    "sed_activity_edges": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _vp]),
    "sed_argmax_runs": (C.c_int, [_vp, _i64, _i64, _i64, _i32, _i32, _vp, _vp, _i32, _vp]),
    "sed_argmax_histogram": (C.c_int, [_vp, _i64, _i64, _i64, _i32, _vp, _vp]),
    "sed_merge_events": (C.c_int, [_vp, _vp, _vp, _i64, _f64, _vp, _vp, _vp, _vp]),
    "sed_merge_runs": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i32, _f64, _vp, _vp, _vp]),
    "sed_threshold_merge_runs": (C.c_int, [_vp, _i64, _i64, _i64, _f32, _i32, _vp, _i32, _f64, _vp, _i32, _vp]),
    "sed_threshold_merge_records": (C.c_int, [_vp, _i64, _i64, _i64, _f32, _i32, _vp, _i32, _f64, _i32, _vp, _i32, _vp]),
    "sed_pack_events": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i32, _vp]),
    "sed_htsat_wav2img": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp]),
    "sed_htsat_wav2img_ragged": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp]),
    "sed_htsat_repeat2img": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp, _vp]),
    "sed_framewise_expand": (C.c_int, [_vp, _i64, _i64, _i64, _i32, _i64, _vp, _vp]),
    "sed_resampler_create": (C.c_int, [_i32, _i32, C.POINTER(_vp)]),
    "sed_resampler_destroy": (None, [_vp]),
    "sed_resampler_out_len": (_i64, [_vp, _i64]),
    "sed_resampler_get_filter": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), _vp]),
    "sed_resampler_forward": (C.c_int, [_vp, _vp, _i32, _f32, _i64, _i64, _i64, _vp, _i64, _vp]),
    "sed_cut_events": (C.c_int, [_vp, _i32, _f32, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


def status_string(status: int) -> str:
    return lib.sed_status_string(int(status)).decode()


def check(status: int, what: str = "") -> None:
    """Turn a negative sed_status into the exception the reference interface would raise."""
    if status >= 0:
        return
    msg = f"{what}: {status_string(status)} (sed_status {status})" if what else f"{status_string(status)} ({status})"
    if status in (SED_ERR_BAD_ARG, SED_ERR_UNSUPPORTED_NFFT, SED_ERR_SHORT_INPUT, SED_ERR_CAPACITY):
        raise ValueError(msg)
    raise RuntimeError(msg)


def stream_ptr(device) -> int:
    """The HIP stream torch is currently recording on for `device`, as an integer handle."""
    import torch
    return int(torch.cuda.current_stream(device).cuda_stream)
