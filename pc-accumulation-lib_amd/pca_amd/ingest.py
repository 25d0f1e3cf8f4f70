"""Ingest pipeline: overlaps disk reads, label remapping and host->device copies with the GPU work
of earlier frames.  Wraps any of the observation loaders; yields the same `[obs]` batches, but with the arrays already
resident in HBM (the accumulators accept cuda tensors in place of numpy / PIL inputs).

    loader = PrefetchingLoader(Kitti360Dataloader(...), depth=4)
    for observations in loader:
        acc.integrate(observations)

A small pool of reader threads (PCA_INGEST_THREADS, default 8; the decoders release the GIL) does the part that has
nothing to do with the GPU -- file reads, PNG decode, label remapping -- up to `depth` batches ahead, delivered in
order.  Everything that talks to HIP stays on the consumer's thread (measured: HIP calls from a second
Python thread, even a lone event wait, cost the first one milliseconds per step): when batch k is handed out, batch k+1
is copied into a ring of reused pinned buffers and its H2D copies are enqueued on a side stream, so they travel while
the GPU works on batch k; the consumer's stream waits on the copy event, never on the host."""
import collections
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np


class DeviceImage:
    """An image that lives on the device for the kernels and on the host (numpy) for viz / PIL users.  Built from a decoded
    array with its device copy already on the way, or -- images not wanted ahead of time (PCA_INGEST_IMAGES=0: the
    ground-truth-semantics flow never looks at them) -- from the lazily opened file: decoded / uploaded on first use."""

    def __init__(self, host, dev, shape=None):
        self._host, self._dev = host, dev
        self.shape = tuple(shape) if shape is not None else tuple(host.shape)

    @property
    def host(self):
        if not isinstance(self._host, np.ndarray):
            self._host = np.asarray(self._host, dtype=np.uint8)
        return self._host

    @property
    def dev(self):
        if self._dev is None:
            import torch
            self._dev = torch.from_numpy(np.ascontiguousarray(self.host)).cuda()
        return self._dev

    def __array__(self, dtype=None, copy=None):
        return self.host if dtype is None else self.host.astype(dtype)


def _torch_dtype(np_dtype):
    import torch
    return torch.from_numpy(np.empty(0, dtype=np_dtype)).dtype


class PinnedUploader:
    """Host array -> device tensor without a pageable copy: numpy copies the array into a reused pinned block (25 us for
    2 MB) and the H2D copy is asynchronous.  torch's `.to(device)` of a pageable array is synchronous -- the host sits
    in it until the stream's earlier kernels AND the copy are done, 0.2 ms of the unchanged driver's 0.42 ms step."""

    DEPTH = 4

    def __init__(self, device):
        self.device = device
        self._slots = {}                        # kind -> [next, [(pinned tensor, event)] * DEPTH]

    def __call__(self, kind, a):
        return self.upload_many([(kind, a)])[0]

    _pool = None

    @classmethod
    def _copy_pool(cls):
        if cls._pool is None:
            cls._pool = ThreadPoolExecutor(max_workers=int(os.environ.get('PCA_STAGING_THREADS', '3')),
                                           thread_name_prefix='pca-staging')
        return cls._pool

    def upload_many(self, items):
        """items: [(kind, host array)].  The host-side copies into the pinned blocks run side by side on a small pool
        (numpy releases the GIL for them; three arrays of one KITTI observation: 0.15 ms one after the other on the
        consumer's thread), then the H2D copies are enqueued -- by THIS thread, the only one that talks to HIP -- with one
        event for all of them.  Returns the device tensors in order."""
        import torch
        arrays = [np.ascontiguousarray(a) for _, a in items]
        pins, slots = [], []
        for (kind, _), a in zip(items, arrays):
            ring = self._slots.setdefault(kind, [0, [None] * self.DEPTH])
            i = ring[0] % self.DEPTH
            ring[0] += 1
            slot = ring[1][i]
            tdtype = _torch_dtype(a.dtype)
            if slot is None or slot[0].numel() < a.size or slot[0].dtype != tdtype:
                slot = (torch.empty(max(a.size, 1), dtype=tdtype, pin_memory=True), None)
            elif slot[1] is not None:
                slot[1].synchronize()            # the copy out of this block, DEPTH uploads ago: long done
            pins.append(slot[0][:a.size].view(*a.shape))
            slots.append((ring, i, slot[0]))
        big = [k for k, a in enumerate(arrays) if a.nbytes >= (1 << 18)]
        jobs = []
        if len(big) > 1:                         # the largest stays on this thread, the others go to the pool
            big.sort(key=lambda k: -arrays[k].nbytes)
            pool = self._copy_pool()
            jobs = [pool.submit(np.copyto, pins[k].numpy(), arrays[k]) for k in big[1:]]
        mine = [k for k in range(len(arrays)) if k not in big[1:]] if jobs else range(len(arrays))
        for k in mine:
            np.copyto(pins[k].numpy(), arrays[k])
        for j in jobs:
            j.result()
        out = []
        for pin, a in zip(pins, arrays):
            dev = torch.empty(a.shape, dtype=pin.dtype, device=self.device)
            dev.copy_(pin, non_blocking=True)
            out.append(dev)
        ev = torch.cuda.Event()
        ev.record()
        for ring, i, block in slots:
            ring[1][i] = (block, ev)
        return out


def compose_label_lut(idx2idx, lo=-1, hi=255):
    """The reference remaps labels with a SEQUENCE of in-place masked assignments (conv_semantic_ids): a later pair
    sees the result of an earlier one.  Returns the composed table as an array indexed by (label - lo)."""
    ids = np.arange(lo, hi + 1, dtype=np.int64)
    out = ids.copy()
    for old, new in idx2idx.items():
        out[out == old] = new
    return out, lo


class PrefetchingLoader:
    """Ring of slots, each with pinned host staging AND device buffers allocated once and reused (a fresh device / pinned
    allocation costs milliseconds when the allocator cache has no free block).  A slot's device buffers are overwritten
    by the copy of a later batch only after an event recorded on the consumer's stream when it came back for the next
    batch: everything it did with the slot has been enqueued by then (a GPU-side wait, nobody blocks on the host)."""

    RING = 4                                 # batches: being used, staged ahead, and two of slack

    def __init__(self, loader, depth=4, device=None):
        import torch
        from . import _lib
        _lib.Context.get(device)            # no GPU / no library: fail here, loudly (there is no CPU fallback)
        self.loader = loader
        self.depth = depth
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        # PCA_INGEST_SIDE_STREAM=1: copies on a stream of their own (overlap with kernels); default: the consumer's stream
        self.stream = torch.cuda.Stream(self.device) if os.environ.get('PCA_INGEST_SIDE_STREAM') else None
        # PCA_INGEST_IMAGES=0: do not decode / upload images ahead (the ground-truth-semantics flow never reads them)
        self.images = os.environ.get('PCA_INGEST_IMAGES', '1') != '0'
        self.lut = None
        if hasattr(loader, 'idx2idx'):
            self.lut = compose_label_lut(loader.idx2idx)
        bs = max(int(getattr(loader, 'batch_size', 1)), 1)
        self._ring = [dict(pin={}, dev={}, copied=None, released=None) for _ in range(self.RING * bs)]
        self._ring_next = 0

    def __len__(self):
        return len(self.loader)

    def _buffers(self, slot, name, shape, dtype):
        """(pinned, device) buffers `name` of a slot, grown on demand and reused from then on."""
        import torch
        n = int(np.prod(shape))
        pin, dev = slot['pin'].get(name), slot['dev'].get(name)
        if pin is None or pin.numel() < n or pin.dtype != dtype:
            pin = slot['pin'][name] = torch.empty(max(n, 1), dtype=dtype).pin_memory()
            dev = slot['dev'][name] = torch.empty(max(n, 1), dtype=dtype, device=self.device)
        return pin[:n].view(*shape), dev[:n].view(*shape)

    def _to_device(self, obs):
        """Consumer thread: stage one observation (host arrays) and enqueue its H2D copies on the side stream."""
        import torch
        img, pc, sem_gt = obs
        slot = self._ring[self._ring_next % len(self._ring)]
        self._ring_next += 1
        if slot['copied'] is not None:
            slot['copied'].synchronize()                                # long done: RING batches ago
        lazy_img = not isinstance(img, np.ndarray)           # images are not staged ahead: see DeviceImage
        src = [('pc', np.ascontiguousarray(pc, dtype=np.float32)),
               ('sem', np.ascontiguousarray(np.asarray(sem_gt)[:, -1]).astype(np.uint8))]
        if not lazy_img:
            src.append(('img', np.ascontiguousarray(img, dtype=np.uint8)))
        dev = {}
        stream = self.stream if self.stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.stream(stream):
            if slot['released'] is not None and self.stream is not None:
                stream.wait_event(slot['released'])                     # kernels that read the device buffers are done
            for name, a in src:
                pin, d = self._buffers(slot, name, a.shape, _torch_dtype(a.dtype))
                # numpy does the host copy: torch's CPU copy_ fans a 2 MB copy out over every core of the box and
                # took 1-3 ms per call here
                np.copyto(pin.numpy(), a)
                d.copy_(pin, non_blocking=True)
                dev[name] = d
            ev = torch.cuda.Event()
            ev.record(stream)
        slot['copied'] = ev
        image = DeviceImage(img, None, shape=(img.size[1], img.size[0], 3)) if lazy_img else DeviceImage(src[2][1], dev['img'])
        return (image, dev['pc'], dev['sem']), ev, slot

    def _read(self, idx):
        """read_obs with the label remap done through the composed table (identical result, one pass)."""
        ld = self.loader
        if self.lut is None or not hasattr(ld, 'pc_paths'):
            obs = ld.read_obs(idx)
            if self.images and not isinstance(obs[0], np.ndarray):
                obs = (np.asarray(obs[0], dtype=np.uint8), ) + tuple(obs[1:])   # decode on the reader thread
            return obs
        import os

        import PIL.Image as Image
        from datasets.kitti360_utils import read_pc_bin_file, read_sem_gt_bin_file
        pc = read_pc_bin_file(os.path.join(ld.root_path, ld.pc_paths[idx]))
        img = Image.open(os.path.join(ld.root_path, ld.img_paths[idx]))
        if self.images:
            img = np.asarray(img, dtype=np.uint8)    # decode HERE, on the reader thread (Image.open is lazy)
        sem = read_sem_gt_bin_file(os.path.join(ld.root_path, ld.sem_gt_paths[idx]))
        if sem is None:
            sem = np.zeros((pc.shape[0], 1))
        table, lo = self.lut
        k = sem[:, 0].astype(np.int64) - lo
        ok = (k >= 0) & (k < table.size)
        out = sem[:, 0].astype(np.int64)
        out[ok] = table[k[ok]]
        return (img, pc, out[:, None])

    def __iter__(self):
        import torch
        n, bs = len(self.loader), self.loader.batch_size
        torch.cuda.set_device(self.device)
        # host-only work on the pool's threads: no HIP call there.  Batches are submitted in order and collected in order.
        n_threads = int(os.environ.get('PCA_INGEST_THREADS', '8')) if os.environ.get('PCA_INGEST_THREAD', '1') != '0' else 0
        pool = ThreadPoolExecutor(max_workers=n_threads) if n_threads > 0 else None
        pending = collections.deque()
        state = {'idx': 0}

        def read_batch(i0):
            return [self._read(i0 + k) for k in range(bs)]

        def stage():
            while state['idx'] + bs <= n and len(pending) < max(self.depth, 1):
                i0 = state['idx']
                state['idx'] += bs
                pending.append(pool.submit(read_batch, i0) if pool is not None else i0)
            if not pending:
                if pool is not None:
                    pool.shutdown(wait=False)
                return None
            head = pending.popleft()
            try:
                item = head.result() if pool is not None else read_batch(head)
            except BaseException as e:           # surfaced in the consumer
                return e
            return [self._to_device(obs) for obs in item]

        ahead = stage()
        held = []
        while True:
            item = ahead
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            cur = torch.cuda.current_stream(self.device)
            for slot in held:                    # the previous batch: all its uses are on the consumer's stream by now
                rel = torch.cuda.Event()
                rel.record(cur)
                slot['released'] = rel
            held = [slot for _, _, slot in item]
            if self.stream is not None:
                for _, ev, _ in item:
                    cur.wait_event(ev)
            ahead = stage()                      # the next batch's copies travel while this one is being integrated
            yield [obs for obs, _, _ in item]
