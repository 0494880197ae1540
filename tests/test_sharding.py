"""Multi-GPU partitioning (CPU): files across ranks, spans of one recording, call stitching, and a
world_size-2 gloo run whose gathered result equals the single-process one."""
import os
import socket
import sys

import numpy as np
import pytest

import oracle
from conftest import ROOT, golden_store
from spectrogram_amd import _ffi as F
from spectrogram_amd import engine as E
from spectrogram_amd import sharding

G = golden_store()


def test_assign_files_balances_and_is_complete():
    sizes = [5, 9, 2, 7, 7, 1, 3, 8, 4, 6]
    parts = sharding.assign_files(sizes, 4)
    assert sorted(i for p in parts for i in p) == list(range(10))
    loads = [sum(sizes[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(sizes)
    assert sharding.assign_files(sizes, 4) == parts
    assert sharding.assign_files([], 3) == [[], [], []]
    assert sharding.assign_files([4, 4], 3) == [[0], [1], []]


@pytest.mark.parametrize("rule", [F.FRAMES_PLUGIN_STRICT, F.FRAMES_WORKER_COMPAT])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_split_recording_covers_every_frame_once(rule, world):
    for L in (0, 700, 1024, 1025, 11520, 11433, 1_000_003):
        n, hop = 1024, 256
        g = E.geometry_for(n, hop, rule, L)
        spans = sharding.split_recording(L, n, hop, world, rule, align=8)
        assert len(spans) == world
        assert sum(s["frames"] for s in spans) == g.frames
        nxt = 0
        for s in spans:
            if not s["frames"]:
                continue
            assert s["first_frame"] == nxt
            nxt += s["frames"]
            assert s["sample_start"] == s["first_frame"] * hop
            assert s["sample_start"] + s["sample_count"] <= L
            # the span, transformed on its own, has exactly its frames
            gs = E.geometry_for(n, hop, rule, s["sample_count"])
            assert gs.frames == s["frames"], (L, s, gs.frames)


def test_spans_reproduce_the_whole_transform():
    x = G.signal("bat_ragged")
    whole = oracle.plugin_get_frequencies(x, 1024, 768)
    spans = sharding.split_recording(len(x), 1024, 256, 3, align=4)
    parts = [oracle.plugin_get_frequencies(x[s["sample_start"]:s["sample_start"] + s["sample_count"]], 1024, 768)
             for s in spans if s["frames"]]
    assert np.array_equal(np.concatenate(parts), whole)


def test_merge_calls_stitches_across_spans():
    f = np.zeros(64, F.FEATURE_DTYPE)
    f["peak_bin"] = np.arange(64) + 100
    f["peak_idx"][10:20] = 200
    f["peak_idx"][30:36] = 180          # crosses the span boundary at frame 32
    f["peak_idx"][33] = 220
    whole = E.segment_calls(f, 100, 1)
    a = E.segment_calls(f[:32], 100, 1, first_frame=0)
    b = E.segment_calls(f[32:], 100, 1, first_frame=32)
    merged = sharding.merge_calls([a, b])
    assert len(whole) == 2 and len(a) == 2 and len(b) == 1
    assert merged.tobytes() == whole.tobytes()


def test_merge_calls_applies_min_frames_after_stitching():
    """A 4-frame run crossing the boundary at frame 32 as 2 + 2 survives min_frames=3 (as in the whole-recording
    pass); a 2-frame run inside a span and a 1 + 1 run across the boundary do not."""
    f = np.zeros(96, F.FEATURE_DTYPE)
    f["peak_bin"] = 200 - np.arange(96)
    f["peak_idx"][30:34] = 190          # 2 + 2 across frame 32
    f["peak_idx"][10:12] = 190          # short run inside span 0
    f["peak_idx"][63:65] = 190          # 1 + 1 across frame 64
    f["peak_idx"][70:80] = 210
    whole = E.segment_calls(f, 100, 3)
    assert [(int(c["start_frame"]), int(c["end_frame"])) for c in whole] == [(30, 33), (70, 79)]
    parts = [E.segment_calls(f[a:b], 100, 1, first_frame=a) for a, b in ((0, 32), (32, 64), (64, 96))]
    assert sharding.merge_calls(parts, min_frames=3).tobytes() == whole.tobytes()
    # the per-span minimum loses the boundary call
    lossy = [E.segment_calls(f[a:b], 100, 3, first_frame=a) for a, b in ((0, 32), (32, 64), (64, 96))]
    assert len(sharding.merge_calls(lossy)) == 1


@pytest.mark.parametrize("seed", range(40))
def test_random_spans_reproduce_the_whole_segmentation(seed):
    """Any cut of a recording into contiguous spans, segmented span by span (minimum run length 1) and merged with the
    real minimum, gives the whole-recording segmentation byte for byte: runs of every length across every kind of
    boundary, spans of one frame, empty spans, thresholds that select nothing or everything."""
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.integers(1, 400))
    f = np.zeros(n, F.FEATURE_DTYPE)
    f["peak_bin"] = rng.integers(0, 512, n)
    # runs of loud frames of random length between gaps of random length
    level = np.zeros(n, np.int64)
    i = 0
    while i < n:
        gap, run = int(rng.integers(0, 12)), int(rng.integers(1, 15))
        level[i + gap:i + gap + run] = rng.integers(90, 256, min(run, max(0, n - i - gap)))
        i += gap + run
    f["peak_idx"] = level
    threshold = int(rng.choice([0, 1, 100, 150, 255]))
    min_frames = int(rng.integers(1, 7))
    whole = E.segment_calls(f, threshold, min_frames)
    cuts = sorted(set(int(c) for c in rng.integers(0, n + 1, int(rng.integers(0, 9)))) | {0, n})
    if rng.random() < 0.3:
        cuts = sorted(cuts + [cuts[len(cuts) // 2]])          # an empty span
    parts = [E.segment_calls(f[a:b], threshold, 1, first_frame=a) for a, b in zip(cuts[:-1], cuts[1:])]
    merged = sharding.merge_calls(parts, min_frames=min_frames)
    assert merged.tobytes() == whole.tobytes(), (n, threshold, min_frames, cuts)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = golden_store()
    names = ["noise", "tone", "bat", "bat_ragged", "fullscale"]
    sizes = [g.manifest["signals"][n]["length"] for n in names]
    mine = sharding.assign_files(sizes, world)[rank]
    # (the CPU checker stands in for the GPU transform: this test is about partition + gather)
    local = {i: oracle.worker_render(g.signal(names[i]), 1024, 75.0, None) for i in mine}
    # one long recording, frame spans + call stitching
    x = g.signal("bat")
    span = sharding.split_recording(len(x), 1024, 256, world, align=4)[rank]
    idx, mags, _, _ = oracle.plugin_get_frequencies(x[span["sample_start"]:span["sample_start"] + span["sample_count"]],
                                                    1024, 768, want_mags=True)
    ft = np.zeros(len(idx), F.FEATURE_DTYPE)
    ft["peak_bin"] = oracle.frame_peaks_f32(mags)
    ft["peak_idx"] = idx[np.arange(len(idx)), ft["peak_bin"]]
    calls = E.segment_calls(ft, 150, 1, first_frame=span["first_frame"])
    gathered = sharding.gather_results((local, idx, calls))
    dist.barrier()
    if rank == 0:
        q.put(gathered)
    dist.destroy_process_group()


def test_world_size_2_gloo_gather_matches_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    names = ["noise", "tone", "bat", "bat_ragged", "fullscale"]
    images = {}
    for local, _, _ in gathered:
        images.update(local)
    assert sorted(images) == list(range(len(names)))
    for i, n in enumerate(names):
        assert np.array_equal(images[i], oracle.worker_render(G.signal(n), 1024, 75.0, None))
    x = G.signal("bat")
    whole_idx, mags, _, _ = oracle.plugin_get_frequencies(x, 1024, 768, want_mags=True)
    assert np.array_equal(np.concatenate([g[1] for g in gathered]), whole_idx)
    ft = np.zeros(len(whole_idx), F.FEATURE_DTYPE)
    ft["peak_bin"] = oracle.frame_peaks_f32(mags)
    ft["peak_idx"] = whole_idx[np.arange(len(whole_idx)), ft["peak_bin"]]
    assert sharding.merge_calls([g[2] for g in gathered]).tobytes() == E.segment_calls(ft, 150, 1).tobytes()


def _batch_rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import zlib
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    files, sizes = _batch_files()

    class CheckerPipeline:        # the CPU checker stands in for the GPU pipeline: this test is about the orchestration
        def run(self, items):
            for tag, x in items:
                yield tag, oracle.worker_render(x, 1024, 75.0, None)

    kept, secs = sharding.run_file_shard(sizes, rank, world, lambda i: files[i], CheckerPipeline(),
                                         lambda img: zlib.crc32(img.tobytes()))
    gathered = sharding.gather_results((rank, kept, secs))
    dist.barrier()
    if rank == 0:
        q.put(gathered)
    dist.destroy_process_group()


def _batch_files():
    rng = np.random.default_rng(99)
    sizes = [1024 + 256 * int(k) for k in rng.integers(4, 60, 11)]
    files = [(rng.standard_normal(n) * 0.2).astype(np.float32) for n in sizes]
    return files, sizes


def test_world_size_2_file_batch_orchestration():
    """run_file_shard + gather_results over two gloo ranks: every file is transformed exactly once, on the rank
    assign_files gave it to, and the gathered digests equal a single process's."""
    import zlib
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    files, sizes = _batch_files()
    parts = sharding.assign_files(sizes, 2)
    assert [r for r, _, _ in gathered] == [0, 1]
    merged = {}
    for r, kept, secs in gathered:
        assert sorted(kept) == parts[r] and secs > 0
        merged.update(kept)
    assert sorted(merged) == list(range(len(files)))
    for i, x in enumerate(files):
        assert merged[i] == zlib.crc32(oracle.worker_render(x, 1024, 75.0, None).tobytes())


def test_bind_to_gpu_numa_uses_the_nodes_cpus(tmp_path):
    """The rank of a per-file batch moves to the CPUs beside its GPU before it allocates page-locked buffers."""
    assert sharding.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert sharding.parse_cpulist("") == set()
    before = os.sched_getaffinity(0)
    try:
        some = sorted(before)[: max(1, len(before) // 2)]
        (tmp_path / "node3").mkdir()
        (tmp_path / "node3" / "cpulist").write_text(",".join(str(c) for c in some) + "\n")
        assert sharding.bind_to_gpu_numa(0, sysfs=str(tmp_path), node=3) == 3
        assert os.sched_getaffinity(0) == set(some)
        # a node whose CPUs this process may not use, a node without a list, no node: nothing changes
        (tmp_path / "node4").mkdir()
        (tmp_path / "node4" / "cpulist").write_text("100000-100003\n")
        assert sharding.bind_to_gpu_numa(0, sysfs=str(tmp_path), node=4) is None
        assert sharding.bind_to_gpu_numa(0, sysfs=str(tmp_path), node=5) is None
        assert sharding.bind_to_gpu_numa(0, sysfs=str(tmp_path), node=-1) is None
        assert os.sched_getaffinity(0) == set(some)
    finally:
        os.sched_setaffinity(0, before)
