#!/usr/bin/env python3
"""Headline benchmark: STFT Msamples/s (node) at FFT 1024, Hann, 75 % overlap (hop 256) on MI355X.

Workload: one hour of synthetic 384 kHz mono bat-call PCM per GPU,
resident in HBM before the timed region; one step = one pass of the worker-path transform
(spectrogramWorker.js semantics: worker Hann, accurate twiddles, log10|X|/5 quantiser, colour LUT,
RGBA ImageData layout) over the whole recording.  `--out u8` stops at the colour indices
([frame][bin] uint8, what the plugin's getFrequencies returns); it is reported under `variants`.  Ranks are independent (one recording each, no collective in the data path).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--out u8|rgba] [--seconds S] [--files F]

The same line also carries (`files`) the per-file batch of the baseline's third configuration -- 5 s x 500 kHz int16
recordings from host memory to host memory, sharded over the ranks by `sharding.assign_files`, each rank's share
through the overlapped `FilePipeline`, per-file checksums gathered on rank 0 -- and (`variants.node_host_*`) the
headline workload driven through the Node.js addon (node/bench.js).

Prints ONE JSON line (rank 0).  `value` is whole-job Msamples/s.  `roofline` prices the dominant
kernel against HBM with algorithmic bytes (each input sample once + each output byte once) over
its average launch time measured with HIP events on the launch stream; `traffic` is the fabric-side
byte count of one launch from the committed counter profile of the same workload (profiles/), if any.  `cpu_baseline` is the CPU
checker in oracle/ (a C restatement of the reference's worker) timed on the host cores on a bounded
sample of the same signal.
"""
import argparse
import json
import os
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SAMPLE_RATE = 384000
FFT_N = 1024
HOP = 256
HBM_PEAK_GBS = 8000.0


def synth_chunk(torch, dev, start, count, seed):
    """Bat-call-like PCM on the device: white noise 0.01 + 5 ms FM down-sweeps (80->35 kHz, Hann
    envelope, amplitude 0.3) every 80 ms + a 110 kHz CF tone (0.05) in the second half of each
    10 s block.  float32 in [-1, 1]."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    i = torch.arange(start, start + count, device=dev, dtype=torch.float64)
    x = 0.01 * (2 * torch.rand(count, device=dev, generator=g, dtype=torch.float32) - 1)
    period, plen = int(0.080 * SAMPLE_RATE), int(0.005 * SAMPLE_RATE)
    p = torch.remainder(i, period)
    t = p / SAMPLE_RATE
    dur = plen / SAMPLE_RATE
    phase = 2 * np.pi * (80000.0 * t + 0.5 * (35000.0 - 80000.0) * t * t / dur)
    env = 0.5 * (1 - torch.cos(2 * np.pi * p / plen))
    pulse = (0.3 * env * torch.sin(phase)).to(torch.float32)
    x += torch.where(p < plen, pulse, torch.zeros_like(pulse))
    block = torch.remainder(i, 10 * SAMPLE_RATE)
    cf = (0.05 * torch.sin(2 * np.pi * 110000.0 * i / SAMPLE_RATE)).to(torch.float32)
    x += torch.where(block >= 5 * SAMPLE_RATE, cf, torch.zeros_like(cf))
    return x


def profiled_traffic(out_name, samples):
    """HBM-side bytes of one launch of the dominant kernel, from the PMC passes committed under
    profiles/ (tools/profile.sh writes traffic.json); None when there is no profile of this workload."""
    best = None
    for d in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        f = os.path.join(ROOT, "profiles", d, "traffic.json")
        if os.path.isfile(f):
            try:
                t = json.load(open(f))
            except ValueError:
                continue
            e = t.get(out_name)
            if e and int(e.get("samples", 0)) == int(samples):
                best = int(e["bytes_per_launch"])
    return best


def profiled_counter(out_name, samples, counter):
    """One counter of one launch of the dominant kernel from the newest profiles/*/summary.json of this workload."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for d in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, d, "summary.json")
        if os.path.isfile(f):
            try:
                t = json.load(open(f))
                if int(t.get("samples", 0)) == int(samples):
                    best = int(t["kinds"][out_name]["per_dispatch"][counter])
            except (ValueError, KeyError, TypeError):
                continue
    return best


def lds_entry(out_name, samples, cus):
    """LDS busy fraction of one launch of the dominant kernel: SQ_LDS_IDX_ACTIVE over CU-cycles (GRBM_GUI_ACTIVE is
    summed over the 8 XCDs), from the committed counter profile of this workload."""
    idx = profiled_counter(out_name, samples, "SQ_LDS_IDX_ACTIVE")
    gui = profiled_counter(out_name, samples, "GRBM_GUI_ACTIVE")
    conf = profiled_counter(out_name, samples, "SQ_LDS_BANK_CONFLICT")
    return {"idx_active_cycles_per_launch": idx, "cu_cycles_per_launch": int(cus * gui / 8) if gui else None,
            "bank_conflict_cycles_per_launch": conf, "source": "profiles/*/summary.json",
            "frac": round(idx / (cus * gui / 8.0), 4) if idx and gui else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--out", choices=["u8", "rgba"], default="rgba")
    ap.add_argument("--seconds", type=int, default=3600, help="recording length per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--files", type=int, default=-1,
                    help="per-file batch: number of 5 s x 500 kHz int16 files over all ranks (-1: 1250 per rank, which on "
                         "8 GPUs is the baseline's 10000; 0: skip)")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank dry run on a box with one GPU: every rank uses cuda:0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus), file=sys.stderr)
            sys.exit(2)

    import torch
    import torch.distributed as dist

    from spectrogram_amd import _ffi as F
    from spectrogram_amd import engine as E

    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # The ranks exchange two scalars (the slowest rank's wall time) and meet at barriers; no tensor ever crosses
        # GPUs (files and spans are independent), so the group is gloo over the loopback interface and RCCL is not
        # initialised at all: nothing about the interconnect can fail the run.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    L = args.seconds * SAMPLE_RATE
    # ---- synthetic recording, generated on the device in 300 s blocks
    x = torch.empty(L, dtype=torch.float32, device=dev)
    blk = 300 * SAMPLE_RATE
    for s in range(0, L, blk):
        c = min(blk, L - s)
        x[s:s + c] = synth_chunk(torch, dev, s, c, 12345 + rank * 1000 + s // blk)
    torch.cuda.synchronize()

    OUT_KINDS = {"u8": F.OUT_U8_INDEX, "rgba": F.OUT_RGBA8, "features": F.OUT_FEATURES}

    def make_engine(out_kind, path="worker", fft_n=FFT_N, hop=HOP, **kw):
        if path == "worker":      # spectrogramWorker.js semantics
            cfg = dict(window_id=F.WIN_WORKER_HANN, scale_mode=F.SCALE_WORKER_LOG5, frame_rule=F.FRAMES_WORKER_COMPAT,
                       twiddle_mode=F.TW_ACCURATE)
        else:                     # spectrogram.esm.js semantics (hann, recurrence twiddles, gain 20 / range 80)
            cfg = dict(window_id=E.window_id("hann"), scale_mode=F.SCALE_PLUGIN_DB, frame_rule=F.FRAMES_PLUGIN_STRICT,
                       twiddle_mode=F.TW_PLUGIN_RECURRENCE, gain_db=20.0, range_db=80.0)
        arith = F.ARITH_EXACT_F64 if kw.pop("exact", False) else F.ARITH_FAST_F32
        freq_scale = kw.pop("freq_scale", None)
        eng = E.Engine(fft_n=fft_n, hop=hop, arith=arith, out_kind=out_kind, device=local_rank, **cfg, **kw)
        if freq_scale:            # the plugin's option scale: 'mel' | 'logarithmic' | 'bark' | 'erb' (spectrogram.esm.js:389-411)
            eng.set_filterbank(*E.filterbank_table(freq_scale, fft_n, SAMPLE_RATE))
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        return eng

    def run(out_name, steps, warmup, samples=None, pcm=None, **eng_kw):
        """`pcm`: a device tensor of int16 samples instead of the float32 recording (the engine then gets in_kind = int16)."""
        src = x if pcm is None else pcm
        if pcm is not None:
            eng_kw["in_kind"] = F.IN_I16
        eng = make_engine(OUT_KINDS[out_name], **eng_kw)
        n_in = L if samples is None else min(L, samples)
        nbytes = eng.out_bytes(n_in)
        out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        g = eng.geometry(n_in)
        for _ in range(warmup):
            eng.stft_device(src.data_ptr(), n_in, out.data_ptr(), nbytes, wait=False)
        barrier()
        t0 = time.perf_counter()
        eng.timer_start()
        for _ in range(steps):
            eng.stft_device(src.data_ptr(), n_in, out.data_ptr(), nbytes, wait=False)
        ev_ms = eng.timer_stop_ms()          # HIP events on the launch stream
        barrier()
        wall = time.perf_counter() - t0
        checksum = int(out[: 1 << 20].to(torch.int64).sum().item())
        eng.close()
        del out
        return wall, ev_ms / steps, nbytes, g, checksum

    wall, kern_ms, out_bytes, geo, checksum = run(args.out, args.steps, args.warmup)
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    value = world * L * args.steps / wall / 1e6

    # ---- per-file batch (the baseline's third configuration): host memory -> host memory, files sharded over the ranks
    files_res = None
    n_files = 1250 * world if args.files < 0 else args.files
    if n_files > 0:
        from spectrogram_amd import sharding
        FL, FRATE = 2_500_000, 500000                       # 5 s x 500 kHz
        rng = np.random.default_rng(7)
        bases = []
        for b in range(4):                                   # four base recordings ...
            v = rng.standard_normal(FL) * 0.01
            for p0 in range(10000 * (b + 1), FL - 3000, 40000):   # 5 ms pulses every 80 ms
                tt = np.arange(2500, dtype=np.float64)
                v[p0:p0 + 2500] += 0.3 * np.hanning(2500) * np.sin(2 * np.pi * (80e3 * tt / FRATE - 0.5 * 45e3 / 2500 * tt * tt / FRATE))
            bases.append(np.round(np.clip(v, -1, 1) * 32767).astype(np.int16))

        def file_content(i):
            """File i: every file differs from every other one (a base recording rotated by a file-specific number of
            samples, with the file index written into its first two samples)."""
            v = np.roll(bases[i % 4], 4099 * (i // 4 + 1) % FL)
            v[0], v[1] = np.int16(i & 0x7fff), np.int16((i >> 15) & 0x7fff)
            return v

        sizes = [FL] * n_files
        mine = sharding.assign_files(sizes, world)[rank]
        content = {i: file_content(i) for i in mine}        # this rank's files, in host memory before the clock starts
        digest = lambda o: zlib.crc32(o[::521].tobytes())               # a CRC over every 521st byte of every output
        fkw = dict(fft_n=FFT_N, hop=HOP, window_id=F.WIN_WORKER_HANN, scale_mode=F.SCALE_WORKER_LOG5,
                   frame_rule=F.FRAMES_WORKER_COMPAT, twiddle_mode=F.TW_ACCURATE, arith=F.ARITH_FAST_F32, in_kind=F.IN_I16,
                   out_kind=F.OUT_RGBA8 if args.out == "rgba" else F.OUT_U8_INDEX, device=local_rank)
        # (more than one rank: each binds to the CPUs next to its GPU before its page-locked buffers are allocated)
        numa = sharding.bind_to_gpu_numa(local_rank) if world > 1 and not args.rehearse else None
        with sharding.FilePipeline(FL, slots=3, copy=False, **fkw) as pipe:
            for _ in pipe.run([(i, content[i]) for i in mine[:6]]):
                pass
            barrier()
            t0 = time.perf_counter()
            sums, my_s = sharding.run_file_shard(sizes, rank, world, lambda i: content[i], pipe, digest)
            barrier()
            wall_f = time.perf_counter() - t0
        # each rank renders the first and the last file of its share once more, one synchronous call each through a
        # fresh context, and compares with what its pipeline delivered
        recheck_ok = True
        if mine:
            with E.Engine(**fkw) as chk:
                for i in sorted({mine[0], mine[-1]}):
                    recheck_ok = recheck_ok and digest(chk.stft(content[i]).reshape(-1).view(np.uint8)) == sums[i]
        del content
        if world > 1:
            tf = torch.tensor([wall_f], dtype=torch.float64)
            dist.all_reduce(tf, op=dist.ReduceOp.MAX)
            wall_f = float(tf.item())
        gathered = sharding.gather_results((rank, len(mine), my_s, sums, numa, recheck_ok))
        if rank == 0:
            allsums = {}
            for g_ in gathered:
                allsums.update(g_[3])
            files_res = {
                "workload": "%d files x %d int16 samples (5 s x 500 kHz), every file different, host memory to host memory, worker "
                            "render to %s, files sharded over %d rank(s) by size, 3 overlapped contexts per rank" % (
                                n_files, FL, "RGBA8" if args.out == "rgba" else "uint8 indices", world),
                "files": n_files, "seconds": round(wall_f, 4), "files_per_s": round(n_files / wall_f, 1),
                "msamples_per_s": round(n_files * FL / wall_f / 1e6, 1),
                "per_rank": [{"rank": r, "files": nf, "seconds": round(s_, 4), "msamples_per_s": round(nf * FL / s_ / 1e6, 1) if s_ > 0 else None,
                              "numa_node": nn, "recheck_ok": ok}
                             for r, nf, s_, _, nn, ok in gathered],
                # every file delivered exactly once, and every rank's re-rendered files came out the same
                "outputs_consistent": len(allsums) == n_files and all(g_[5] for g_ in gathered),
                "distinct_outputs": len(set(allsums.values())),
            }

    result = None
    if rank == 0:
        algo_bytes = L * 4 + out_bytes
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
        result = {
            "metric": "STFT Msamples/s (node) at FFT 1024 Hann 75% overlap; % of HBM roofline",
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%d s x 384 kHz mono synthetic bat-call PCM per GPU, FFT 1024, worker Hann, hop 256 (75%% overlap), "
                            "worker log10|X|/5 quantiser, output %s" % (args.seconds, "uint8 index [frame][bin]" if args.out == "u8"
                                                                        else "RGBA8 image [bin][frame]"),
                "samples_per_gpu": L, "frames_per_gpu": int(geo.frames), "fft": FFT_N, "hop": HOP,
                "sharding": "one recording per GPU, no collective",
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": profiled_traffic(args.out, L),
                "kernel_ms": round(kern_ms, 4), "algorithmic_bytes_per_launch": int(algo_bytes),
                "bytes_per_sample": round(algo_bytes / L, 3),
            },
            "checksum": checksum,
            "notes": "host of this line: Python ctypes over the C ABI (the same C entry points the Node addon calls; "
                     "variants.node_host_* is the same workload through node/bench.js); colour indices within 1 LSB of the reference; "
                     "float32 dB within 0.01 dB for bins within 80 dB of the frame peak and within 0.04 dB (measured 0.026) for the "
                     "rest of the display range down to the -100 dB floor (tests/); feature records are bit-exact",
        }
        if files_res:
            result["files"] = files_res

    if rank == 0 and world == 1:
        # what this box's memory system delivers to the simplest possible kernels, next to the nominal 8 TB/s:
        # a device-to-device copy (reads + writes) and a fill (writes only) of the recording-sized buffer
        y = torch.empty_like(x)
        for fn, key, nbytes in ((lambda: y.copy_(x), "measured_copy_gbps", 2 * x.numel() * 4),
                                (lambda: y.fill_(1.0), "measured_fill_gbps", x.numel() * 4)):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            result["roofline"][key] = round(nbytes * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del y
        # the kernel's algorithmic rate against what this box's memory system does for a linear read + write mix
        result["roofline"]["frac_of_measured_copy"] = round(result["roofline"]["achieved"] / result["roofline"]["measured_copy_gbps"], 4)

    if rank == 0 and world == 1 and args.out == "rgba":
        # The floor the memory system sets for this kernel's access pattern, measured now: tools/microbench_mix is the
        # kernel with every instruction but its loads and stores removed (same grid, same rounds of 128 frames, the
        # same whole-sector row runs of the same image, the same linear sample reads), as a child process of its own.
        exe = os.path.join(ROOT, "tools", "microbench_mix")
        entry = None
        if os.path.exists(exe):
            try:
                r = subprocess.run([exe, "floor", str(int(geo.frames))], capture_output=True, text=True, timeout=120)
                m = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else None
            except (subprocess.TimeoutExpired, ValueError):
                m = None
            if m:
                entry = {"ms": round(m["both_ms"], 4), "writes_alone_ms": round(m["writes_ms"], 4), "reads_alone_ms": round(m["reads_ms"], 4),
                         "frac_of_peak": round((m["image_bytes"] + m["sample_bytes"]) / (m["both_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "kernel_over_model": round(kern_ms / m["both_ms"], 4),
                         "what": "tools/microbench_mix.hip: the kernel's loads and stores with no arithmetic"}
        result["roofline"]["memory_only_model"] = entry

    if rank == 0 and world == 1:
        # The vector-ALU roofline beside the HBM one: at hop = N/4 the transform is ~200 flop per sample over 6-12
        # bytes, i.e. at or above the machine balance, so instruction issue -- not bytes -- fills most of the time.
        # insts_per_launch: SQ_INSTS_VALU of one launch from the committed counter profile of this workload;
        # issue_peak_measured: this GPU's v_fma_f32 issue rate at the kernels' occupancy, measured now.
        probe = make_engine(OUT_KINDS[args.out])
        peak = probe.measure_valu_peak()
        probe.close()
        insts = profiled_counter(args.out, L, "SQ_INSTS_VALU")
        result["roofline"]["valu"] = {
            "insts_per_launch": insts, "issue_peak_measured": round(peak, 1), "unit": "G wavefront-instructions/s",
            "frac": round(insts / (kern_ms * 1e-3) / 1e9 / peak, 4) if insts and peak > 0 else None,
        }
        # ... and the LDS pipe: cycles its index was active over the CU-cycles of the launch, both from that profile
        cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
        result["roofline"]["lds"] = lds_entry(args.out, L, cus)

    if rank == 0 and world == 1 and not args.no_variants:
        # other entry points of the same path on the same recording (not the headline; same timing method)
        k = max(3, args.steps // 4)
        c4 = min(L, 60 * SAMPLE_RATE)
        specs = {
            ("rgba" if args.out == "u8" else "u8") + "_worker_1024": dict(out=("rgba" if args.out == "u8" else "u8")),
            "u8_plugin_getFrequencies_1024": dict(out="u8", path="plugin"),
            "features_plugin_1024": dict(out="features", path="plugin"),
            "u8_plugin_mel_1024": dict(out="u8", path="plugin", freq_scale="mel"),
            "u8_worker_512_hop128": dict(out="u8", fft_n=512, hop=128),
            "rgba_worker_512_hop128": dict(out="rgba", fft_n=512, hop=128),      # the page's default FFT size (main.js:886-889)
            "features_plugin_512_hop128": dict(out="features", path="plugin", fft_n=512, hop=128),
            "u8_worker_2048_hop512": dict(out="u8", fft_n=2048, hop=512),
            "rgba_worker_2048_hop512": dict(out="rgba", fft_n=2048, hop=512),
            "u8_plugin_2048_hop512": dict(out="u8", path="plugin", fft_n=2048, hop=512),
            "rgba_plugin_4096_hop256_bins160-1280_60s": dict(out="rgba", path="plugin", fft_n=4096, hop=256, bin_lo=160,
                                                            bin_hi=1280, samples=c4),
            # bit-identical arithmetic (double butterflies over float32 storage), 60 s of the recording
            "rgba_worker_1024_exact_60s": dict(out="rgba", exact=True, samples=c4),
        }
        result["variants"] = {}
        # the same recording as 16-bit PCM (what a WAV file holds; value of a sample = s / 32768): half the input bytes
        specs[args.out + "_worker_1024_int16_pcm"] = dict(out=args.out, pcm=True)
        x16 = None
        for name, sp in specs.items():
            sp = dict(sp)
            o = sp.pop("out")
            n_in = sp.get("samples", L)
            in_bytes = 4
            if sp.pop("pcm", False):
                if x16 is None:
                    x16 = torch.empty(L, dtype=torch.int16, device=dev)
                    for s0 in range(0, L, blk):
                        x16[s0:s0 + blk] = torch.clamp(torch.round(x[s0:s0 + blk] * 32767.0), -32768, 32767).to(torch.int16)
                sp["pcm"] = x16
                in_bytes = 2
            w2, k2, ob2, _, _ = run(o, k, 1, **sp)
            result["variants"][name] = {"value": round(n_in * k / w2 / 1e6, 1), "kernel_ms": round(k2, 4),
                                        "roofline_frac": round((n_in * in_bytes + ob2) / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            if name.endswith("_worker_1024") and n_in == L:     # the other output of the headline workload: its issue and LDS fractions too
                vi = profiled_counter(o, L, "SQ_INSTS_VALU")
                result["variants"][name]["valu_frac"] = round(vi / (k2 * 1e-3) / 1e9 / peak, 4) if vi and peak > 0 else None
                result["variants"][name]["lds"] = lds_entry(o, L, cus)
        del x16
        # the headline workload through the JavaScript host (N-API addon): must agree with the figure above
        import shutil
        node = shutil.which("node")
        if node and os.path.isfile(os.path.join(ROOT, "node", "build", "sonospec.node")):
            try:
                r = subprocess.run([node, os.path.join(ROOT, "node", "bench.js"), "--seconds", str(args.seconds), "--steps",
                                    str(args.steps), "--warmup", str(args.warmup), "--out", args.out],
                                   capture_output=True, text=True, timeout=600)
                j = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-300:]}
            except (subprocess.SubprocessError, ValueError, IndexError) as e:
                j = {"error": str(e)[:300]}
            if "value" in j:
                j["vs_python_host"] = round(j["value"] / value, 4)
            result["variants"]["node_host_%s_1024" % args.out] = j

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import shutil
        import tempfile
        import oracle
        # Every core this process may run on -- unless the box states its CPU share for one GPU's job in OMP_NUM_THREADS
        # (on a shared multi-GPU host the affinity mask shows the whole machine, and a pool of that size is not this
        # job's to use); both numbers are reported.
        cores_available = len(os.sched_getaffinity(0))
        share = os.environ.get("OMP_NUM_THREADS", "")
        cores = min(cores_available, int(share)) if share.isdigit() and int(share) > 0 else cores_available
        cores_note = ("OMP_NUM_THREADS=%s (this job's CPU share on the box)" % share) if cores < cores_available else "the whole affinity mask"
        # (1) the C restatement of the worker render on `cores` POSIX threads
        probe = x[: 2 * SAMPLE_RATE].cpu().numpy()
        t0 = time.perf_counter()
        oracle.worker_render(probe, FFT_N, 75.0, None, threads=cores)
        rate = len(probe) / (time.perf_counter() - t0)            # samples/s on `cores` threads
        secs = int(max(4, min(args.seconds, rate * 15 / SAMPLE_RATE)))  # about 15 s of CPU work
        sample = x[: secs * SAMPLE_RATE].cpu().numpy()
        t0 = time.perf_counter()
        oracle.worker_render(sample, FFT_N, 75.0, None, threads=cores)
        dt = time.perf_counter() - t0
        c_port = {"value": round(len(sample) / dt / 1e6, 2), "unit": "Msamples/s", "cores": cores, "kind": "port",
                  "sample": "first %d s of the same recording (%d samples), RGBA render by the C restatement of "
                            "spectrogramWorker.js (oracle/sono_oracle.c), %d POSIX threads" % (secs, len(sample), cores),
                  "seconds": round(dt, 2), "cores_available": cores_available, "cores_used": cores, "cores_limit": cores_note}
        result["cpu_baseline"] = c_port
        # (2) the same render in the reference's own language and runtime: the JavaScript restatement under
        #     Node worker_threads, one worker per core, hop-aligned 10 s chunks with a 768-sample halo
        node = shutil.which("node")
        js = os.path.join(ROOT, "oracle", "worker_restatement.js")
        if node and os.path.isfile(js):
            jsecs, jpasses = 600, 8                                               # about 15 s of CPU work on the GPU box's host cores
            jsample = x[: min(L, jsecs * SAMPLE_RATE)].cpu().numpy()
            with tempfile.TemporaryDirectory() as td:
                f = os.path.join(td, "pcm.f32")
                jsample.tofile(f)
                try:
                    r = subprocess.run([node, js, "bench", f, str(FFT_N), "75", str(cores), "10", str(SAMPLE_RATE), str(jpasses)],
                                       capture_output=True, text=True, timeout=600)
                    j = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
                except (subprocess.SubprocessError, ValueError, IndexError):
                    j = None
            if j:
                result["cpu_baseline"] = {
                    "value": round(j["msamples_per_s"], 2), "unit": "Msamples/s", "cores": cores, "kind": "port",
                    "sample": "first %d s of the same recording rendered %d times over (%d samples), RGBA render by the JavaScript restatement of "
                              "spectrogramWorker.js (oracle/worker_restatement.js, byte-identical to the reference on its golden "
                              "cases) under Node %s worker_threads, %d workers, 10 s chunks" % (
                                  len(jsample) // SAMPLE_RATE, j["passes"], j["samples"], j["node"], cores),
                    "seconds": round(j["wall_s"], 2), "cores_available": cores_available, "cores_used": cores,
                    "cores_limit": cores_note, "c_port": c_port,
                }
                # how the restatement relates to the reference itself (which cannot travel to this box): both were
                # timed in the build container on one thread and the same 10 s signal by tools/time_reference.js, like
                # for like (both in Node's main context with the real globals; the ratio inside a `vm` sandbox is stored
                # beside it)
                rf = os.path.join(ROOT, "profiles", "cpu_baseline_ratio.json")
                if os.path.isfile(rf):
                    ratio = json.load(open(rf))
                    result["cpu_baseline"]["restatement_over_reference"] = ratio["restatement_over_reference"]
                    result["cpu_baseline"]["reference_equivalent"] = round(j["msamples_per_s"] / ratio["restatement_over_reference"], 2)
                    result["cpu_baseline"]["ratio_measured"] = "%s, %s, both in the main context: reference %.2f, restatement %.2f Msamples/s" % (
                        ratio["node"], ratio["signal"], ratio["reference_msamples_per_s"], ratio["restatement_msamples_per_s"])
                    if "vm_sandbox" in ratio:
                        result["cpu_baseline"]["restatement_over_reference_in_vm_sandbox"] = ratio["vm_sandbox"]["restatement_over_reference"]

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
