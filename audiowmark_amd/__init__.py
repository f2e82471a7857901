"""audiowmark_amd -- MI355X-native spectral watermark path of audiowmark.

The product is the C-ABI library ``libawm_hip.so`` (``include/awm_hip.h``: hand-written gfx950
kernels + the host pipeline).  This package is only the ctypes binding used by the tests, the
benchmark and multi-GPU sharding; PyTorch provides device memory, streams and
``torch.distributed`` -- plumbing, not compute.

There is no CPU fallback: importing works without a GPU (table helpers are pure host code),
every compute entry point raises ``AwmError`` when no gfx950 device is usable or when the
library is missing.
"""
from .binding import (AwmError, Context, Pattern, lib, library_path, tab_up_down, tab_bit_pos, tab_mix_entries,
                      tab_bit_order, tab_frame_mod, tab_sync_bits, tab_window, tab_synth_window, conv_encode,
                      set_params, set_speed_params, key_bytes, test_key, plan_chunks, merge_patterns, merge_patterns_raw, patterns_to_dicts,
                      PATTERN_DTYPE)
from . import binding

__all__ = ["AwmError", "Context", "Pattern", "lib", "library_path", "tab_up_down", "tab_bit_pos", "tab_mix_entries",
           "tab_bit_order", "tab_frame_mod", "tab_sync_bits", "tab_window", "tab_synth_window", "conv_encode",
           "set_params", "set_speed_params", "key_bytes", "test_key", "plan_chunks", "merge_patterns", "merge_patterns_raw", "patterns_to_dicts",
           "PATTERN_DTYPE", "binding"]
