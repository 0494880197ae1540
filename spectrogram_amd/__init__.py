"""spectrogram_amd: MI355X-native engine for SonoRadar's spectrogram hot path.

The computation lives in ``libsonospec.so`` (hand-written HIP for gfx950 behind the C ABI of
``include/sonospec.h``).  This package is the Python host side: a ctypes binding (``_ffi``), a
numpy-friendly context wrapper (``engine``) and mirrors of the two reference interfaces --
the wavesurfer Spectrogram plugin (``plugin``) and the spectrogram Web Worker (``worker``).
"""
from . import _ffi  # noqa: F401
from .engine import Engine  # noqa: F401
from .plugin import AudioBuffer, Spectrogram, createSpectrogramPlugin, noverlap_from_percent  # noqa: F401
from .worker import SpectrogramWorker  # noqa: F401

__all__ = ["Engine", "Spectrogram", "AudioBuffer", "SpectrogramWorker", "createSpectrogramPlugin",
           "noverlap_from_percent"]
