"""Ingest pipeline: overlaps disk reads, label remapping and host->device copies with the GPU work
of earlier frames.  Wraps any of the observation loaders; yields the same `[obs]` batches, but with the arrays already
resident in HBM (the accumulators accept cuda tensors in place of numpy / PIL inputs).

    loader = PrefetchingLoader(Kitti360Dataloader(...), depth=4)
    for observations in loader:
        acc.integrate(observations)

A reader thread fills pinned host buffers and enqueues the copies on a side stream; the consumer's stream waits on
the copy event, never on the host."""
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

    def _to_device(self, obs):
        import torch
        img, pc, sem_gt = obs
        host_img = np.ascontiguousarray(np.asarray(img), dtype=np.uint8)
        pin = [torch.from_numpy(np.ascontiguousarray(pc, dtype=np.float32)).pin_memory(),
               torch.from_numpy(host_img.copy()).pin_memory(),
               torch.from_numpy(np.ascontiguousarray(np.asarray(sem_gt)[:, -1]).astype(np.uint8)).pin_memory()]
        with torch.cuda.stream(self.stream):
            dev = [t.to(self.device, non_blocking=True) for t in pin]
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return (DeviceImage(host_img, dev[1]), dev[0], dev[2]), ev, pin

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
            for _, ev, _ in item:
                torch.cuda.current_stream(self.device).wait_event(ev)
            self._keepalive = item               # pinned staging stays alive until the next batch is handed out
            yield [obs for obs, _, _ in item]
