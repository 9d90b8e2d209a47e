"""Region-feature input side of the hot path: what sits between the bottom-up-attention
`.npz` files and ``ObjectRelation.forward``.

Reference behaviour mirrored here (file:line are into the reference repo):
  * frame choice per video          base/base_dataset.py:82-101 (`_sample_objects`: 'rand' for training, 'uniform' otherwise)
  * per-frame `.npz` schema         data_loader/WebVid_dataset.py:243-256 (`x`, `bbox`, `info{objects_conf, objects_id, image_w, image_h}`)
  * top-R selection / padding / box geometry / mask
                                    data_loader/WebVid_dataset.py:134-283 -- done ON THE DEVICE by `dvlp_region_select`
                                    (bit-exact indices, tests/golden/g1_region_select.npz) instead of in DataLoader workers
  * per-rank sharding               base/base_data_loader.py:23-28 (`DistributedSampler(shuffle, drop_last=True)`, `set_epoch`)
  * alternating multi-loader epoch  trainer/trainer_dist.py:123-129 (`zip(*loaders)`, one batch of each in turn)

Host work is limited to reading files into pinned staging buffers; the 2.37 MB/sample of fp32 features cross PCIe once,
raw, and are selected / padded / concatenated to `[B, F, R, 2054]` in HBM.  Dataset metadata (csv/json), tokenisation and
the torch DataLoader worker pool stay the reference's Python and are out of scope.
"""
from __future__ import annotations

import os
import random
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

FEAT_DIM = 2048


def sample_frame_indices(num_segments: int, num_files: int, mode: str = "rand", rng: Optional[random.Random] = None) -> List[int]:
    """Frame files to read for one video (base/base_dataset.py:82-101 + WebVid_dataset.py:96-110).  'rand': one random frame per
    equal interval, sorted; 'uniform': the interval midpoints.  All files in order when there are exactly `num_segments`."""
    if num_segments == num_files:
        return list(range(num_segments))
    acc = min(num_segments, num_files)
    intervals = np.linspace(start=0, stop=num_files, num=acc + 1).astype(int)
    ranges = [(int(a), int(b) - 1) for a, b in zip(intervals[:-1], intervals[1:])]
    if mode == "rand":
        rng = rng or random
        return sorted(rng.choice(range(lo, hi)) for lo, hi in ranges)       # `range(lo, hi)`: the reference never picks `hi`
    if mode == "uniform":
        return [(lo + hi) // 2 for lo, hi in ranges]
    raise NotImplementedError(mode)


def read_frame_npz(path: str):
    """One frame file -> (x [N,2048] f32, bbox [N,4] f32, conf [N] f32, (image_w, image_h))."""
    with np.load(path, allow_pickle=True) as z:
        info = z["info"].item()
        return (np.ascontiguousarray(z["x"], np.float32), np.ascontiguousarray(z["bbox"], np.float32),
                np.ascontiguousarray(info["objects_conf"], np.float32), (float(info["image_w"]), float(info["image_h"])))


def shard_indices(n: int, world: int, rank: int, epoch: int = 0, shuffle: bool = True, seed: int = 0) -> np.ndarray:
    """The index list `DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=shuffle, drop_last=True)` yields after
    `set_epoch(epoch)` (base/base_data_loader.py:23-28; torch/utils/data/distributed.py) -- same permutation, same tail drop."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).numpy()
    else:
        idx = np.arange(n)
    per_rank = (n - world + world - 1) // world if n % world != 0 else n // world      # ceil((n - world) / world) when ragged
    per_rank = max(per_rank, 0)
    total = per_rank * world
    return idx[:total][rank:total:world]


def alternate(loaders: Sequence[Iterable]) -> Iterator[Tuple[int, object]]:
    """(loader index, batch) in the order the reference trainer consumes its loaders: one batch of each per round, stopping
    with the shortest (trainer/trainer_dist.py:123-129)."""
    for batches in zip(*loaders):
        for i, b in enumerate(batches):
            yield i, b


class _Staging:
    """One set of pinned host buffers + the event that marks the end of the last host->device copy issued from them."""

    def __init__(self, batch, frames, max_regions, pin):
        mk = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt).pin_memory() if pin else torch.zeros(*s, dtype=dt)  # noqa: E731
        self.feat, self.box = mk(batch, frames, max_regions, FEAT_DIM), mk(batch, frames, max_regions, 4)
        self.conf, self.wh = mk(batch, frames, max_regions), mk(batch, frames, 2)
        self.n = mk(batch, frames, dt=torch.int32)
        # numpy views of the same (pinned) memory: plain memcpys that release the GIL, so frames can be staged from several threads
        self.nfeat, self.nbox, self.nconf, self.nwh, self.nn = (t.numpy() for t in (self.feat, self.box, self.conf, self.wh, self.n))
        self.copied = torch.cuda.Event() if pin else None
        self.in_flight = False

    def wait_reusable(self):
        """Block the host until the DMA that last read these buffers has finished (they are about to be overwritten)."""
        if self.in_flight:
            self.copied.synchronize()
            self.in_flight = False

    def tensors(self):
        return self.feat, self.box, self.conf, self.wh, self.n


class RegionBatcher:
    """Raw frames of a batch -> `object [B,F,R,2054]` fp32 + `object_mask [B,F,R]` on the device.

    Frames may have different region counts (20-100 in the released features): they are packed into pinned host buffers of
    `max_regions` rows with a per-frame valid count, copied asynchronously on `copy_stream`, and selected on the device.
    The pinned buffers are double-buffered (``nbuf``): the host fills set k+1 while the DMA of set k is still running, and a
    set is only overwritten after the event recorded behind its last copy has completed."""

    def __init__(self, batch: int, frames: int, regions: int, max_regions: int = 100, device: str | torch.device = "cuda", nbuf: int = 2):
        self.B, self.F, self.R, self.M = batch, frames, regions, max_regions
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None and torch.cuda.is_available():
            # resolve to an INDEXED device now, on the constructing thread: a worker thread (prefetching()) starts with torch's
            # current device 0 whatever set_device(LOCAL_RANK) the main thread did
            self.device = torch.device("cuda", torch.cuda.current_device())
        pin = self.device.type == "cuda"
        self.bufs = [_Staging(batch, frames, max_regions, pin) for _ in range(max(1, nbuf))]
        self.cur = 0
        self.copy_stream = torch.cuda.Stream(device=self.device) if pin else None
        self.bytes_staged = 0
        self._pool, self._pool_n = None, 0

    # the staging set being filled (kept as attributes for callers that peek at the buffers)
    @property
    def h_feat(self):
        return self.bufs[self.cur].feat

    @property
    def h_conf(self):
        return self.bufs[self.cur].conf

    def stage(self, b: int, f: int, x: np.ndarray, bbox: np.ndarray, conf: np.ndarray, wh: Tuple[float, float]) -> None:
        n = x.shape[0]
        if n > self.M:
            raise ValueError(f"frame has {n} regions, staging buffers hold {self.M}")
        s = self.bufs[self.cur]
        s.wait_reusable()
        s.nfeat[b, f, :n] = x
        s.nbox[b, f, :n] = bbox
        s.nconf[b, f, :n] = conf
        s.nconf[b, f, n:] = -1.0                                  # never selected: real confidences are positive
        s.nwh[b, f, 0], s.nwh[b, f, 1] = wh
        s.nn[b, f] = n

    def stage_frames(self, items, workers: int = 8) -> None:
        """Stage many frames ``(b, f, x, bbox, conf, wh)`` from a thread pool (the memcpys into pinned memory release the GIL): the
        host-side counterpart of the reference's DataLoader worker processes (base/base_data_loader.py:23-38)."""
        from concurrent.futures import ThreadPoolExecutor
        self.bufs[self.cur].wait_reusable()
        if self._pool is None or self._pool_n != workers:
            self._pool, self._pool_n = ThreadPoolExecutor(max_workers=workers), workers
        list(self._pool.map(lambda it: self.stage(*it), items, chunksize=max(1, len(items) // (4 * workers)) if hasattr(items, "__len__") else 1))

    def stage_video(self, b: int, frame_dir: str, frame_idxs: Sequence[int]) -> None:
        for f, idx in enumerate(frame_idxs):
            self.stage(b, f, *read_frame_npz(os.path.join(frame_dir, f"{idx}.npz")))

    def to_device(self, out=None):
        """-> (object [B,F,R,2054] f32, object_mask [B,F,R] f32, object_len [B,F] int32), all on the device.  Returns as soon as
        the copies and the selection kernel are enqueued; the next batch may be staged immediately (into the other buffer set).
        ``out``: a dict with 'object' / 'object_mask' tensors (``GraphedTrainStep.inputs``) the selection writes into -- from the
        thread that runs the step only (see there)."""
        from . import ops
        if self.device.type != "cuda":
            raise ops._lib.DemoVLPHipError("RegionBatcher.to_device needs a ROCm device: there is no CPU fallback")
        s = self.bufs[self.cur]
        if torch.cuda.current_device() != self.device.index:
            torch.cuda.set_device(self.device)                  # worker threads start on device 0 (see __init__)
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.copy_stream):
            d = [t.to(self.device, non_blocking=True) for t in s.tensors()]
            s.copied.record(self.copy_stream)
        s.in_flight = True
        self.bytes_staged += sum(t.numel() * t.element_size() for t in s.tensors())
        cur.wait_stream(self.copy_stream)
        for t in d:
            t.record_stream(cur)
        self.cur = (self.cur + 1) % len(self.bufs)
        obj, mask, _order, lens = ops.region_select(d[0], d[1], d[2], d[3], self.R, nvalid=d[4],
                                                    out=None if out is None else (out["object"], out["object_mask"]))
        return obj, mask, lens


def prefetching(batches: Iterable, depth: int = 2, device=None) -> Iterator:
    """Run a batch iterator (file reads + staging + enqueueing of copies) on a background thread, ``depth`` batches ahead of the
    consumer: host file I/O for batch k+1 overlaps the device step of batch k (the reference gets this from DataLoader workers,
    base/base_data_loader.py:23-38).

    The worker thread adopts the CALLER's current device (or ``device``): a fresh thread starts on device 0 regardless of the
    ``torch.cuda.set_device(LOCAL_RANK)`` the main thread did.  The worker allocates, copies and launches, which would invalidate a
    hipGraph capture running in 'global' error mode on another thread: ``GraphedTrainStep`` therefore captures with
    ``capture_error_mode='thread_local'`` (its captures come whenever a new batch shape has been seen three times, mid-epoch included)."""
    import queue
    import threading
    q: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
    END = object()
    dev = None
    if torch.cuda.is_available():
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if dev.type != "cuda":
            dev = None

    def run():
        try:
            if dev is not None:
                torch.cuda.set_device(dev)
            for b in batches:
                q.put(b)
            q.put(END)
        except BaseException as e:  # noqa: BLE001
            q.put(e)

    t = threading.Thread(target=run, daemon=True)
    t.start()
    while True:
        item = q.get()
        if item is END:
            return
        if isinstance(item, BaseException):
            raise item
        yield item
