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
