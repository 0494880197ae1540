#!/usr/bin/env python3
"""Per-file batch benchmark on one GPU: many 5 s x 500 kHz int16 recordings
from host memory to host memory, worker render (RGBA) or colour indices (u8), through the overlapped
FilePipeline and, for comparison, through synchronous sono_stft calls on one context.  PCIe-inclusive.

  python tools/bench_files.py [--files 200] [--seconds 5] [--rate 500000] [--out rgba|u8] [--slots 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrogram_amd import _ffi as F          # noqa: E402
from spectrogram_amd import engine as E        # noqa: E402
from spectrogram_amd.sharding import FilePipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=200)
    ap.add_argument("--seconds", type=float, default=5.0)
    ap.add_argument("--rate", type=int, default=500000)
    ap.add_argument("--out", choices=["rgba", "u8"], default="rgba")
    ap.add_argument("--slots", type=int, default=3)
    a = ap.parse_args()
    L = int(a.seconds * a.rate)
    rng = np.random.default_rng(1)
    base = [np.round(np.clip(rng.standard_normal(L) * 0.1, -1, 1) * 32767).astype(np.int16) for _ in range(4)]
    kw = dict(fft_n=1024, hop=256, window_id=F.WIN_WORKER_HANN, scale_mode=F.SCALE_WORKER_LOG5, frame_rule=F.FRAMES_WORKER_COMPAT,
              twiddle_mode=F.TW_ACCURATE, arith=F.ARITH_FAST_F32, in_kind=F.IN_I16,
              out_kind=F.OUT_RGBA8 if a.out == "rgba" else F.OUT_U8_INDEX)
    files = [(i, base[i % 4]) for i in range(a.files)]
    res = {"files": a.files, "samples_per_file": L, "out": a.out, "slots": a.slots}
    with FilePipeline(L, slots=a.slots, copy=False, **kw) as pipe:
        for _ in pipe.run(files[:8]):
            pass
        t0 = time.perf_counter()
        chk = 0
        for _, out in pipe.run(files):
            chk += int(out[out.size // 2])
        dt = time.perf_counter() - t0
    res["pipeline"] = {"seconds": round(dt, 4), "msamples_per_s": round(a.files * L / dt / 1e6, 1), "files_per_s": round(a.files / dt, 1),
                       "host_gbps_in_plus_out": round(a.files * (L * 2 + out.size) / dt / 1e9, 2)}
    with E.Engine(**kw) as eng:
        eng.stft(base[0])
        n = max(8, a.files // 8)
        t0 = time.perf_counter()
        for i in range(n):
            eng.stft(base[i % 4])
        dt = time.perf_counter() - t0
    res["synchronous"] = {"seconds": round(dt, 4), "msamples_per_s": round(n * L / dt / 1e6, 1), "files_per_s": round(n / dt, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
