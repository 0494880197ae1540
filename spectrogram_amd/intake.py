"""File intake with the reference's semantics, over the C ABI (sono_wav_* / sono_guano_fields):
RIFF/WAVE chunk walk (modules/fileLoader.js:7-63), GUANO metadata (modules/guanoReader.js:3-80),
crop by time (modules/cropAudio.js:1-59), the load filters (fileLoader.js:179-181), and PCM -> device
float32 without a host-side float copy (sono_pcm_to_f32).
"""
import ctypes as C

import numpy as np

from . import _ffi as F

NO_GUANO = "(No GUANO metadata found in file)"


def _buf(data):
    b = np.frombuffer(bytes(data) if not isinstance(data, (bytes, bytearray, memoryview, np.ndarray)) else data, np.uint8)
    return np.ascontiguousarray(b)


def wav_info(data):
    """All three chunk walks of the reference in one pass; raises SonoError(ERR_FORMAT) where its reads throw."""
    b = _buf(data)
    w = F.SonoWavInfo()
    rc = F.lib().sono_wav_parse(F.ptr(b), b.size, C.byref(w))
    if rc != F.OK:
        raise F.SonoError(rc, "not a readable RIFF/WAVE file")
    return w


def get_wav_sample_rate(data):
    """getWavSampleRate (fileLoader.js:7-27): 256000 when there is no file or no fmt chunk."""
    if data is None:
        return 256000
    return int(wav_info(data).sample_rate)


def get_wav_duration(data):
    """getWavDuration (fileLoader.js:29-63)."""
    if data is None:
        return 0
    return float(wav_info(data).duration_s)


def extract_guano_metadata(data):
    """extractGuanoMetadata (guanoReader.js:3-34): the text of the first chunk that holds 'GUANO|Version:'."""
    if data is None:
        return "(no file selected)"
    b = _buf(data)
    off, ln = C.c_uint64(0), C.c_uint64(0)
    rc = F.lib().sono_wav_guano(F.ptr(b), b.size, C.byref(off), C.byref(ln))
    if rc != F.OK:
        raise F.SonoError(rc, "truncated chunk")     # the reference throws a RangeError
    if ln.value == 0:
        return NO_GUANO
    return bytes(b[off.value:off.value + ln.value]).decode("utf-8", errors="replace")


def parse_guano_metadata(text):
    """parseGuanoMetadata (guanoReader.js:36-80): {} or {date, time, latitude, longitude}."""
    raw = (text or "").encode("utf-8")
    g = F.SonoGuanoFields()
    rc = F.lib().sono_guano_fields(raw, len(raw), C.byref(g))
    if rc != F.OK:
        raise F.SonoError(rc, "sono_guano_fields")
    if not g.present:
        return {}
    return {k: getattr(g, k).decode("utf-8", errors="replace") for k in ("date", "time", "latitude", "longitude")}


def crop_wav(data, start_time, end_time):
    """cropWavBlob (cropAudio.js:1-59): bytes of the cropped file, or None where the reference returns null."""
    if data is None:
        return None
    b = _buf(data)
    L = F.lib()
    need = L.sono_wav_crop(F.ptr(b), b.size, float(start_time), float(end_time), None, 0)
    if need < 0:
        raise F.SonoError(int(need), "sono_wav_crop")
    if need == 0:
        return None
    out = np.empty(need, np.uint8)
    got = L.sono_wav_crop(F.ptr(b), b.size, float(start_time), float(end_time), F.ptr(out), out.size)
    if got != need:
        raise F.SonoError(int(got), "sono_wav_crop")
    return out.tobytes()


def accept_file(file_bytes, duration_s, time_expansion=False):
    """fileLoader.js:179-181: 'ok', 'small' (< 200 KiB) or 'long' (> 20 s unless time expansion)."""
    return ("ok", "small", "long")[F.lib().sono_wav_accept(int(file_bytes), float(duration_s), 1 if time_expansion else 0)]


def pcm_samples(data, info=None):
    """The data chunk's bytes as a numpy view (no copy) and the layout to hand to Engine.pcm_to_device."""
    b = _buf(data)
    w = info or wav_info(b)
    if not (w.has_fmt and w.has_data):
        raise F.SonoError(F.ERR_FORMAT, "fmt or data chunk missing")
    n = int(w.frames) * int(w.block_align)
    return b[int(w.data_offset):int(w.data_offset) + n], w
