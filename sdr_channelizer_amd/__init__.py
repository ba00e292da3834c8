"""MI355X-native polyphase-filterbank channelizer: a drop-in for the channelizer
path of cwozny/sdr_channelizer (see DESIGN.md / INTEGRATION.md).

The arithmetic lives in libpfb_channelizer.so (HIP, gfx950) behind the C ABI of
include/pfb_channelizer.h; this package is the thin host side.
"""
from ._lib import LIB_PATH, PfbError  # noqa: F401
from .channelizer import Channelizer, center_frequencies, design_prototype, pinned_empty  # noqa: F401

__all__ = ["Channelizer", "center_frequencies", "design_prototype", "pinned_empty", "PfbError", "LIB_PATH"]
