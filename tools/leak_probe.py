import numpy as np, sys
sys.path.insert(0, '/root/repo')
from spectrogram_amd import _ffi as F, engine as E
rng = np.random.default_rng(5)
for n, hop in ((1024, 1024), (512, 512), (2048, 2048), (4096, 4096)):
    worst = 0.0
    for trial in range(6):
        frames = 64
        x = np.zeros(n * frames, np.float32)
        t = np.arange(n)
        for f in range(1, frames, 2):          # odd frames loud, even frames silent: pairs (silent, loud)
            kind = trial % 3
            if kind == 0: seg = 0.9 * np.sin(2 * np.pi * rng.uniform(0.01, 0.49) * t + rng.uniform(0, 6))
            elif kind == 1: seg = rng.standard_normal(n) * 0.3
            else: seg = 0.5 * np.sin(2 * np.pi * (0.05 * t + 0.2 * t * t / n)) + rng.standard_normal(n) * 0.01
            x[f * n:(f + 1) * n] = np.clip(seg, -1, 1)
        for tw, sm, fr, win in ((F.TW_ACCURATE, F.SCALE_PLUGIN_DB, F.FRAMES_PLUGIN_STRICT, F.WINDOW_IDS["hann"]),):
            with E.Engine(fft_n=n, hop=hop, window_id=win, scale_mode=sm, frame_rule=fr, twiddle_mode=tw, arith=F.ARITH_FAST_F32, out_kind=F.OUT_F32_MAG) as e:
                m = e.stft(x); route = e.kernel_route()
            nf = m.shape[0]
            quiet = m[0:nf:2].max(axis=1); loud = m[1:nf:2].max(axis=1)
            k = min(len(quiet), len(loud))
            r = (quiet[:k] / loud[:k]).max()
            worst = max(worst, r)
    print(n, route, "max |X_silent| / |X_loud|max = %.3g" % worst)
