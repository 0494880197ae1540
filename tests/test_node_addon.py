"""The JavaScript host side (node/: N-API addon, Spectrogram plugin mirror, worker contract).

CPU: the addon loads and its host rules / option validation reproduce the reference's outputs.
GPU: node/test_parity.js renders every golden case through the addon (exact: bit-identical,
fast: within 1 LSB) and node/test_worker_threads.js drives the message contract over a real
worker_threads Worker with transferred buffers.
"""
import json
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

NODE = shutil.which("node")
ADDON = os.path.join(ROOT, "node", "build", "sonospec.node")
needs_node = pytest.mark.skipif(NODE is None or not os.path.exists(ADDON), reason="node or the built addon is missing")


def run_node(script, *args, timeout=300):
    p = subprocess.run([NODE, os.path.join(ROOT, "node", script), *args], capture_output=True, text=True, timeout=timeout)
    last = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    return p, (json.loads(last[-1]) if last else None)


@needs_node
def test_addon_loads_and_host_rules_match_reference():
    p, res = run_node("test_parity.js", "cpu")
    assert p.returncode == 0, p.stdout + p.stderr
    assert res["failures"] == 0 and res["checks"] > 200


@needs_node
@pytest.mark.gpu
def test_node_parity_on_gpu():
    p, res = run_node("test_parity.js")
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert res["gpu"] is True and res["failures"] == 0 and res["checks"] > 350


@needs_node
@pytest.mark.gpu
def test_worker_threads_contract_on_gpu():
    p, res = run_node("test_worker_threads.js")
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert res == {"rendered": True, "width": 41, "height": 512, "identical": True, "input_detached": True}


@needs_node
@pytest.mark.gpu
def test_node_file_batch_on_gpu():
    """node/batch.js: .wav files from disk through the overlapped per-file pipeline equal synchronous transforms."""
    p, res = run_node("test_batch.js")
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert res["failures"] == 0 and res["checks"] >= 40


@needs_node
@pytest.mark.gpu
def test_node_tile_cache_and_bridge_on_gpu():
    """node/tiles.js + node/bridge.js: canvas tiles from the cached index plane, directly and over HTTP on 127.0.0.1."""
    p, res = run_node("test_bridge.js")
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert res["failures"] == 0 and res["checks"] >= 20


@needs_node
@pytest.mark.gpu
def test_node_and_python_hosts_agree_on_random_option_sets(tmp_path):
    """The same random plugin option sets (size, overlap given or derived, window, alpha, gain, range, frequency range,
    scale, colour map, height, canvas width, arithmetic) through the Node host and through the Python host: both call
    the same C ABI, so getFrequencies' bytes, the colour-mapped image with its crop window and the canvas image must be
    identical -- what differs can only be how a host marshals an option.  (The Python host is held to the checker on
    such sets in test_gpu_random_configs.py.)"""
    import hashlib

    import numpy as np

    from spectrogram_amd.plugin import AudioBuffer, Spectrogram
    rng = np.random.default_rng(31)
    cases, mine = [], []
    for i in range(16):
        n = int(rng.choice([256, 512, 1024, 2048, 4096]))
        sr = int(rng.choice([44100, 256000, 384000, 500000]))
        L = int(rng.integers(n + 50, 50000))
        x = (rng.standard_normal(L) * 0.05 + 0.4 * np.sin(np.arange(L) * rng.uniform(0.05, 2.5))).astype(np.float32)
        width = int(rng.integers(20, 700))
        opts = {"fftSamples": n, "colorMap": rng.random((256, 4)).round(6).tolist(),
                "scale": str(rng.choice(["linear", "linear", "mel", "logarithmic", "bark", "erb"])),
                "windowFunc": str(rng.choice(["bartlett", "bartlettHann", "blackman", "cosine", "gauss", "hamming", "hann",
                                              "lanczoz", "rectangular", "triangular"])),
                "gainDB": float(rng.choice([20, 0, 40])), "rangeDB": float(rng.choice([80, 100, 60])),
                "frequencyMin": float(rng.choice([0.0, round(rng.uniform(0, 0.2) * sr)])),
                "frequencyMax": float(rng.choice([0.0, round(rng.uniform(0.25, 0.5) * sr), round(0.7 * sr)])),
                "height": int(rng.integers(16, 400)), "arith": str(rng.choice(["fast", "exact"]))}
        if rng.random() < 0.6:
            opts["noverlap"] = int(rng.choice([n // 2, 3 * n // 4, int(rng.integers(1, n))]))
        if opts["windowFunc"] in ("blackman", "gauss") and rng.random() < 0.5:
            opts["alpha"] = float(rng.choice([0.16, 0.25, 0.4]))
        name = "x%d.f32" % i
        x.tofile(tmp_path / name)
        cases.append({"id": i, "pcm": name, "sampleRate": sr, "width": width, "options": opts})
        p = Spectrogram.create(dict(opts))
        try:
            buf = AudioBuffer(x, sr)
            frames = p.getFrequencies(buf, canvas_width=width)[0]
            rgba, (sy, sh) = p.drawSpectrogram(frames, width)
            canvas = p.drawCanvas(frames, width)
        finally:
            p.destroy()
        sha = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()
        mine.append({"id": i, "frames": int(frames.shape[0]), "freq": sha(frames), "image": sha(rgba), "crop": [int(sy), int(sh)],
                     "canvas": sha(canvas), "canvasHeight": int(canvas.shape[0])})
    (tmp_path / "cases.json").write_text(json.dumps(cases))
    p, res = run_node("test_random.js", str(tmp_path))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    for a, b, c in zip(mine, res["results"], cases):
        assert a == b, (a, b, {k: v for k, v in c["options"].items() if k != "colorMap"}, c["width"], c["sampleRate"])
