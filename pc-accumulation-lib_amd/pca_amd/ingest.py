"""Ingest pipeline: overlaps disk reads, label remapping and host->device copies with the GPU work
of earlier frames.  Wraps any of the observation loaders; yields the same `[obs]` batches, but with the arrays already
resident in HBM (the accumulators accept cuda tensors in place of numpy / PIL inputs).

    loader = PrefetchingLoader(Kitti360Dataloader(...), depth=4)
    for observations in loader:
        acc.integrate(observations)

A reader thread fills a ring of REUSED pinned host buffers and enqueues the copies on a side stream; the consumer's
stream waits on the copy event, never on the host, and every handed-out device tensor is registered with the consumer's
stream (`record_stream`), so the caching allocator cannot recycle its block for a later copy while kernels that read it
are still queued."""
import queue
import threading

import numpy as np


class DeviceImage:
    """An image that lives on the device for the kernels and on the host (numpy, pinned) for viz / PIL users."""

    def __init__(self, host, dev):
        self.host, self.dev = host, dev
        self.shape = host.shape

    def __array__(self, dtype=None, copy=None):
        return self.host if dtype is None else self.host.astype(dtype)


def compose_label_lut(idx2idx, lo=-1, hi=255):
    """The reference remaps labels with a SEQUENCE of in-place masked assignments (conv_semantic_ids): a later pair
    sees the result of an earlier one.  Returns the composed table as an array indexed by (label - lo)."""
    ids = np.arange(lo, hi + 1, dtype=np.int64)
    out = ids.copy()
    for old, new in idx2idx.items():
        out[out == old] = new
    return out, lo


class PrefetchingLoader:

    def __init__(self, loader, depth=4, device=None):
        import torch
        self.loader = loader
        self.depth = depth
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self.lut = None
        if hasattr(loader, 'idx2idx'):
            self.lut = compose_label_lut(loader.idx2idx)

    def __len__(self):
        return len(self.loader)

    def _staging(self, name, shape, dtype):
        """Pinned buffer `name` of the current ring slot, grown on demand and reused from then on."""
        import torch
        slot = self._slot
        n = int(np.prod(shape))
        buf = slot.get(name)
        if buf is None or buf.numel() < n or buf.dtype != dtype:
            buf = slot[name] = torch.empty(max(n, 1), dtype=dtype).pin_memory()
        return buf[:n].view(*shape)

    def _to_device(self, obs):
        import torch
        img, pc, sem_gt = obs
        if not hasattr(self, '_ring'):
            self._ring = [dict() for _ in range(self.depth + 2)]      # one more than can be in flight + being filled
            self._ring_next = 0
        self._slot = self._ring[self._ring_next % len(self._ring)]
        self._ring_next += 1
        if 'event' in self._slot:
            self._slot['event'].synchronize()                           # the copy that last read this slot is done
        host_img = np.array(np.asarray(img), dtype=np.uint8)            # own copy: kept by the accumulator (rgbs)
        src = [np.ascontiguousarray(pc, dtype=np.float32), host_img,
               np.ascontiguousarray(np.asarray(sem_gt)[:, -1]).astype(np.uint8)]
        pin = []
        for name, a in zip(('pc', 'img', 'sem'), src):
            buf = self._staging(name, a.shape, torch.from_numpy(a).dtype)
            buf.copy_(torch.from_numpy(a))
            pin.append(buf)
        with torch.cuda.stream(self.stream):
            dev = [t.to(self.device, non_blocking=True) for t in pin]
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._slot['event'] = ev
        return (DeviceImage(host_img, dev[1]), dev[0], dev[2]), ev, dev

    def _read(self, idx):
        """read_obs with the label remap done through the composed table (identical result, one pass)."""
        ld = self.loader
        if self.lut is None or not hasattr(ld, 'pc_paths'):
            return ld.read_obs(idx)
        import os

        import PIL.Image as Image
        from datasets.kitti360_utils import read_pc_bin_file, read_sem_gt_bin_file
        pc = read_pc_bin_file(os.path.join(ld.root_path, ld.pc_paths[idx]))
        img = Image.open(os.path.join(ld.root_path, ld.img_paths[idx]))
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
        q = queue.Queue(maxsize=self.depth)
        n, bs = len(self.loader), self.loader.batch_size
        torch.cuda.set_device(self.device)

        def worker():
            torch.cuda.set_device(self.device)
            try:
                idx = 0
                while idx + bs <= n:
                    batch = [self._to_device(self._read(idx + k)) for k in range(bs)]
                    idx += bs
                    q.put(batch)
            except BaseException as e:           # surfaced in the consumer
                q.put(e)
            q.put(None)

        threading.Thread(target=worker, daemon=True).start()
        while True:
            item = q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            cur = torch.cuda.current_stream(self.device)
            for _, ev, dev in item:
                cur.wait_event(ev)
                for t in dev:                    # allocated on the copy stream, used on the consumer's from here on
                    t.record_stream(cur)
            yield [obs for obs, _, _ in item]
