"""How far beyond full scale the float32 kernels keep their 1-LSB promise (they state it for |x| <= 1).

    python3 tools/amplitude_probe.py QUIET [tone]

Noise of standard deviation QUIET plus bursts (or, with `tone`, one long tone) of amplitude A = 1 ... 1e4; worker and
plugin colour indices against the CPU checker; prints the largest index difference and the number of bins off by more
than one.  Test infrastructure: needs a GPU and the checker in oracle/.
"""
import numpy as np, sys
sys.path.insert(0, '.')
import oracle
from spectrogram_amd import _ffi as F, engine as E
def cyc(a,b):
    d=(a.astype(np.int16)-b.astype(np.int16))%256
    return np.minimum(d,256-d)
rng=np.random.default_rng(5)
QUIET=float(sys.argv[1]) if len(sys.argv)>1 else 0.02
for n in (512,1024,2048,4096):
    hop=n//4
    L=n+hop*400
    for A in (1.0, 3.0, 10.0, 30.0, 100.0, 1000.0, 1e4):
        x=(rng.standard_normal(L)*QUIET).astype(np.float32)
        t=np.arange(L)
        if len(sys.argv)>2:
            seg=slice(20*hop, 20*hop+40*n)
            x[seg]+=(A*np.sin(2*np.pi*0.125*t[seg])).astype(np.float32)
        else:
          for s in range(5*hop, L-2*n, 37*hop):
            x[s:s+n//2]+= (A*np.sin(0.7*t[:n//2])).astype(np.float32)
        _, ref = oracle.worker_render(x, n, 75.0, None, want_index=True)
        with E.Engine(fft_n=n, hop=hop, window_id=F.WIN_WORKER_HANN, scale_mode=F.SCALE_WORKER_LOG5, frame_rule=F.FRAMES_WORKER_COMPAT,
                      twiddle_mode=F.TW_ACCURATE, arith=F.ARITH_FAST_F32, out_kind=F.OUT_U8_INDEX) as eng:
            got=eng.stft(x); route=eng.kernel_route() if hasattr(eng,'kernel_route') else ''
        m=min(len(got),len(ref))
        d=np.abs(got[:m].astype(int)-ref[:m].astype(int))
        refp=oracle.plugin_get_frequencies(x, n, n-hop)
        with E.Engine(fft_n=n, hop=hop, window_id=F.WINDOW_IDS['hann'], gain_db=20.0, range_db=80.0, scale_mode=F.SCALE_PLUGIN_DB,
                      frame_rule=F.FRAMES_PLUGIN_STRICT, twiddle_mode=F.TW_PLUGIN_RECURRENCE, arith=F.ARITH_FAST_F32, out_kind=F.OUT_U8_INDEX) as eng:
            gp=eng.stft(x)
        dp=cyc(gp,refp)
        print(n, A, 'worker max', d.max(), 'n>1', int((d>1).sum()), '| plugin max', dp.max(), 'n>1', int((dp>1).sum()), flush=True)
