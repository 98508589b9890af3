"""sed_amd: MI355X-native log-mel front-end and frame-level event decoding (HIP kernels behind a C ABI)."""
from . import _lib
from .frontend import (FrontendConfig, LogMelFrontend, LogmelFilterBank, Spectrogram, extract_mbe, get_mel_spec_db,
                       hann_window, infer_mode_image, install_fused_frontend, kaldi_fbank, long_input_images,
                       mel_filterbank, repeat_mode_images, repeat_wat2img, reshape_wav2img, wav2fbank, yamnet_log_mel)
from .decode import (decode_predictions, decode_probabilities, detect_onsets_offsets, detect_onsets_offsets_batch,
                     get_classification_events, get_classification_label, get_events, get_events_from_batch, interpolate, interpolate_and_pad,
                     merge_close_events, merge_close_events_batch, pad_framewise_output, predict_events, predict_events_batch, process_json_file)
from .handoff import cut_events, cut_events_from_audio, fbank_of_events, logmel_of_events
from .metrics import EventBasedMetrics, SegmentBasedMetrics, calculate_metrics
from .resample import Resampler, resample
from . import sharding

__version__ = _lib.lib.sed_version().decode()
