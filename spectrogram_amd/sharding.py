"""Spreading the hot path over the GPUs of one node: no collective in the data path.

Recordings are independent, and so are the frames of one recording, so multi-GPU operation is pure
partitioning plus a host-side gather:

* many files (the reference loads one file at a time, modules/fileLoader.js:110-157): each rank takes
  whole files, balanced by size (`assign_files`);
* one long recording: contiguous spans of frames, each span reading `fft_n - hop` samples past its
  last hop so that every frame is computed exactly once (`split_recording`); a call that touches the
  last frame of a span is merged with one starting at the first frame of the next (`merge_calls`).

`gather_results` collects per-rank results on rank 0 with torch.distributed's object gather (any
backend; gloo on CPU, RCCL on GPU) -- the only communication there is.
"""
from . import _ffi as F
from . import engine as E


def assign_files(sizes, world_size):
    """Longest-processing-time-first assignment of files to ranks.  Returns, per rank, the list of
    file indices (each ascending); deterministic for equal sizes."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += int(sizes[i])
    return [sorted(v) for v in out]


def split_recording(n_samples, fft_n, hop, world_size, frame_rule=F.FRAMES_PLUGIN_STRICT, align=64):
    """Contiguous frame spans of one recording, one per rank.

    Returns a list of dicts {rank, first_frame, frames, sample_start, sample_count}.  Span k covers
    frames [first_frame, first_frame + frames); it needs samples [first_frame*hop, (last_frame)*hop + fft_n)
    -- i.e. fft_n - hop samples of halo past its last hop.  Under SONO_FRAMES_PLUGIN_STRICT a frame
    exists only if one more sample follows it, so `sample_count` includes that sample for the last
    frame of the recording.  Span boundaries are multiples of `align` frames (image columns of one
    span then start on aligned addresses)."""
    g = E.geometry_for(fft_n, hop, frame_rule, n_samples)
    total = int(g.frames)
    per = -(-total // world_size)
    per = -(-per // align) * align
    spans = []
    for r in range(world_size):
        f0 = min(total, r * per)
        f1 = min(total, (r + 1) * per)
        n = f1 - f0
        if n > 0:
            s0 = f0 * hop
            s1 = (f1 - 1) * hop + fft_n
            if frame_rule == F.FRAMES_PLUGIN_STRICT:
                s1 += 1                      # a + N < L must hold for the span's last frame too
            s1 = min(s1, n_samples)
            spans.append(dict(rank=r, first_frame=f0, frames=n, sample_start=s0, sample_count=s1 - s0))
        else:
            spans.append(dict(rank=r, first_frame=f0, frames=0, sample_start=0, sample_count=0))
    return spans


def merge_calls(call_lists, min_frames=1):
    """Concatenate per-span call records (absolute frame numbers, spans in order) and stitch a run that
    ends on the last frame of one span to a run that starts on the first frame of the next.

    Segment every span with min_frames=1 and give the minimum run length HERE: it applies to the stitched
    runs, so a call that crosses a span boundary as two short pieces is kept or dropped as the whole-recording
    pass would keep or drop it (a per-span minimum would drop the pieces before they meet)."""
    import numpy as np
    merged = []
    for calls in call_lists:
        for c in calls:
            c = c.copy()
            if merged and int(merged[-1]["end_frame"]) + 1 == int(c["start_frame"]):
                p = merged[-1]
                p["end_frame"] = c["end_frame"]
                p["end_bin"] = c["end_bin"]
                p["high_bin"] = max(int(p["high_bin"]), int(c["high_bin"]))
                p["low_bin"] = min(int(p["low_bin"]), int(c["low_bin"]))
                if int(c["peak_idx"]) > int(p["peak_idx"]):
                    p["peak_idx"], p["peak_bin"], p["peak_frame"] = c["peak_idx"], c["peak_bin"], c["peak_frame"]
            else:
                merged.append(c)
    merged = [c for c in merged if int(c["end_frame"]) - int(c["start_frame"]) + 1 >= int(min_frames)]
    out = np.zeros(len(merged), F.CALL_DTYPE)
    for i, c in enumerate(merged):
        out[i] = c
    return out


def parse_cpulist(text):
    """A sysfs CPU list ("0-15,64-79") as a set of CPU numbers."""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_gpu_numa(device, sysfs="/sys/devices/system/node", node=None):
    """Restricts the calling process to the CPUs of the NUMA node GPU `device` hangs off, so that the page-locked
    buffers a `FilePipeline` allocates next (first touch) and the threads that fill them sit on the GPU's side of
    the machine: with eight ranks each moving tens of GB/s host <-> device, copies that cross sockets are what
    limits the per-file batch.  Best effort: returns the node, or None when the platform does not say, the node
    has no CPU this process may use, or the affinity call is refused (the process then stays where it was)."""
    import os
    if node is None:
        node = F.lib().sono_device_numa_node(int(device))
    if node is None or node < 0:
        return None
    try:
        with open(os.path.join(sysfs, "node%d" % node, "cpulist")) as fh:
            cpus = parse_cpulist(fh.read())
        allowed = cpus & os.sched_getaffinity(0)
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
    except (OSError, ValueError, AttributeError):
        return None
    return node


def run_file_shard(sizes, rank, world_size, load, pipeline, digest):
    """The per-file batch on one rank (the reference loads one file after the other, modules/fileLoader.js:110-157):
    this rank's share of the files (`assign_files` over `sizes`) goes through `pipeline` -- a FilePipeline, or anything
    with run(iterable of (tag, samples)) -> iterable of (tag, output) -- in ascending file order.
    `load(i)` returns file i's samples, `digest(output)` what to keep of its result (the outputs themselves stay on
    the rank that made them).  Returns ({file index: digest}, seconds); rank 0 collects the dicts with gather_results."""
    import time
    mine = assign_files(sizes, world_size)[rank]
    t0 = time.perf_counter()
    kept = {}
    for i, out in pipeline.run((i, load(i)) for i in mine):
        kept[i] = digest(out)
    return kept, time.perf_counter() - t0


def gather_results(local, dst=0):
    """Host-side gather of arbitrary per-rank results on rank `dst` (list indexed by rank, else None)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local, out, dst=dst)
    return out


class FilePipeline:
    """Many short recordings through one GPU with the copies overlapped: `slots` contexts, each with its
    own stream, device staging and a pair of page-locked host buffers; file i goes to slot i % slots, so
    while one file is being transformed the next one is on its way in and the previous result on its
    way out (PCIe is full duplex).  This is the per-file batch path (the reference handles one file at
    a time, modules/fileLoader.js:110-157; its worker receives a transferred buffer, main.js:1417).

    All files of a pipeline share one configuration and must not exceed `max_samples`.
    """

    def __init__(self, max_samples, slots=3, copy=True, **engine_kw):
        self._copy = copy   # False: yield views of the page-locked output buffers (valid until the slot is reused)
        import ctypes as C
        import numpy as np
        self._np, self._C = np, C
        self._lib = F.lib()
        self.engines = [E.Engine(**engine_kw) for _ in range(slots)]
        e0 = self.engines[0]
        self._in_dt = np.int16 if e0.cfg.in_kind == F.IN_I16 else np.float32
        self._in_bytes = int(max_samples) * np.dtype(self._in_dt).itemsize
        self._out_bytes = int(e0.out_bytes(int(max_samples)))
        self.max_samples = int(max_samples)
        self._pin_in, self._pin_out, self._busy = [], [], [None] * slots
        for e in self.engines:
            e.reserve(self.max_samples)      # staging sized now: no allocation (= device-wide wait) between files
        for _ in range(slots):
            pi, po = self._lib.sono_host_malloc(max(4, self._in_bytes)), self._lib.sono_host_malloc(max(4, self._out_bytes))
            if not pi or not po:
                raise F.SonoError(F.ERR_DEVICE, "page-locked host allocation failed")
            self._pin_in.append(pi)
            self._pin_out.append(po)

    def _view(self, ptr, nbytes, dtype):
        buf = (self._C.c_uint8 * nbytes).from_address(ptr)
        return self._np.frombuffer(buf, dtype=dtype)

    def _collect(self, s):
        tag, n, shape_of = self._busy[s]
        self.engines[s].wait()
        out = self.engines[s]._out_array(n) if self._copy else None
        if out is not None:
            out.reshape(-1).view(self._np.uint8)[:] = self._view(self._pin_out[s], out.nbytes, self._np.uint8)
        else:
            out = self._view(self._pin_out[s], int(self.engines[s].out_bytes(n)), self._np.uint8)
        self._busy[s] = None
        return tag, out

    def run(self, files):
        """files: iterable of (tag, samples array).  Yields (tag, output) in submission order."""
        np = self._np
        i = 0
        for tag, x in files:
            s = i % len(self.engines)
            if self._busy[s] is not None:
                yield self._collect(s)
            x = np.ascontiguousarray(x, self._in_dt).reshape(-1)
            if x.shape[0] > self.max_samples:
                raise ValueError("file longer than the pipeline's max_samples")
            self._view(self._pin_in[s], x.nbytes, self._in_dt)[:] = x
            eng = self.engines[s]
            need = int(eng.out_bytes(x.shape[0]))
            eng._check(self._lib.sono_stft_host_async(eng._ctx, self._C.c_void_p(self._pin_in[s]), x.shape[0],
                                                      self._C.c_void_p(self._pin_out[s]), need))
            self._busy[s] = (tag, x.shape[0], None)
            i += 1
        # drain in submission order
        n = len(self.engines)
        for k in range(n):
            s = (i + k) % n
            if self._busy[s] is not None:
                yield self._collect(s)

    def close(self):
        for e in self.engines:
            e.close()
        for p in self._pin_in + self._pin_out:
            self._lib.sono_host_free(self._C.c_void_p(p))
        self._pin_in, self._pin_out, self.engines = [], [], []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
