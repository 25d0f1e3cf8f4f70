"""Output writer: the reference writes each BEV sample as gzip(pickle(dict)) synchronously
(sem_pc_accum.py:280-294); once a sample costs < 1 ms on the GPU that dominates.  AsyncBevWriter does the device->host
copy, pickling and compression on worker threads.  Same container, same dict schema (README.md:60-99 of the reference)."""
import atexit
import gzip
import os
import pickle
import queue
import threading

import numpy as np


class AsyncBevWriter:

    def __init__(self, n_threads=4, compresslevel=9, max_pending=64):
        self.q = queue.Queue(maxsize=max_pending)
        self.compresslevel = compresslevel
        self.errors = []
        self.threads = [threading.Thread(target=self._work, daemon=True) for _ in range(n_threads)]
        for t in self.threads:
            t.start()

    @staticmethod
    def _to_host(bev):
        """Accepts the reference dict (host arrays) or the device_only form {'planes_f16': cuda [21,px,px], trajs}."""
        if 'planes_f16' not in bev:
            return bev
        from bev_generator.sem_bev import SemBEVGenerator
        planes = bev['planes_f16'].cpu().numpy()
        out = SemBEVGenerator.pack_bev(planes, bev['trajs_present'], bev['trajs_future'], bev['trajs_full'],
                                       bev.get('gt_lanes'))
        for k, v in bev.items():
            if k not in out and k != 'planes_f16':
                out[k] = v
        return out

    @staticmethod
    def _plain(bev):
        return dict(bev) if type(bev) is not dict else bev          # LazyBev: waits for its copy here, off the main thread

    def _work(self):
        while True:
            job = self.q.get()
            if job is None:
                return
            bev, filename, write_dir = job
            try:
                os.makedirs(write_dir, exist_ok=True)
                blob = pickle.dumps(self._plain(self._to_host(bev)))
                with gzip.open(os.path.join(write_dir, f'{filename}.gz'), 'wb', compresslevel=self.compresslevel) as f:
                    f.write(blob)
            except Exception as e:              # reported by close(); the reference prints IOErrors and goes on
                self.errors.append(e)
            finally:
                self.q.task_done()

    def submit(self, bev, filename, write_dir):
        """Same arguments as SemanticPointCloudAccumulator.write_compressed_pickle.  A sample whose planes are still on
        their way from the device (LazyBev) is parked; it is filled in -- on THIS thread, the one that talks to HIP -- and
        queued when the next sample arrives (its copy has long finished by then) or at flush / close.  The worker threads
        only pickle and compress."""
        self._release_parked()
        if getattr(bev, '_pending', None) is not None:
            self._parked = (bev, filename, write_dir)
            return
        if type(bev) is dict and 'planes_f16' in bev:       # snapshot: the caller may reuse the device buffer
            bev = dict(bev, planes_f16=bev['planes_f16'].clone())
        self.q.put((bev, filename, write_dir))

    def _release_parked(self):
        parked, self._parked = getattr(self, '_parked', None), None
        if parked is not None:
            self.q.put((dict(parked[0]), parked[1], parked[2]))

    def flush(self):
        self._release_parked()
        self.q.join()

    def close(self):
        self.flush()
        for _ in self.threads:
            self.q.put(None)
        for t in self.threads:
            t.join()
        if self.errors:
            raise self.errors[0]


_shared = None


def shared_writer():
    """Process-wide writer used by SemanticPointCloudAccumulator.write_compressed_pickle for in-flight BEV samples;
    joined at interpreter exit, so every file is on disk when the unchanged driver returns."""
    global _shared
    if _shared is None:
        _shared = AsyncBevWriter(n_threads=int(os.environ.get('PCA_WRITER_THREADS', '4')))
        atexit.register(_shared.close)
    return _shared


def flush_shared():
    """Blocks until everything submitted to the shared writer is on disk (no-op if it was never used)."""
    if _shared is not None:
        _shared.flush()
        if _shared.errors:
            raise _shared.errors.pop(0)
