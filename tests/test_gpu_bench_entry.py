"""bench.py's multi-rank entry, as the driver launches it: `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...`, one process per rank, here two ranks rehearsed on the box's one GPU (`--rehearse`: every rank uses cuda:0).

The ranks meet over gloo (two scalars and barriers; no tensor crosses GPUs), each takes its share of the per-file
batch (modules/fileLoader.js:110-157 loads one file at a time; the batch shards them by size), and rank 0 prints the
one JSON line.  Run in a fresh child process, never in the test process itself.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_bench_line():
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--seconds", "60",
           "--steps", "3", "--warmup", "1", "--files", "16", "--no-cpu-baseline", "--no-variants"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 alone prints
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["scaling"] == "weak"
    assert r["value"] > 0 and r["roofline"]["frac"] > 0
    f = r["files"]
    assert f["files"] == 16 and f["outputs_consistent"] is True and f["distinct_outputs"] == 16
    assert len(f["per_rank"]) == 2 and sorted(e["rank"] for e in f["per_rank"]) == [0, 1]
    assert sum(e["files"] for e in f["per_rank"]) == 16 and all(e["recheck_ok"] for e in f["per_rank"])
