"""Semantic-segmentation model wrapper (reference: utils/onnx_utils.py) -- SURVEY.md 8f rank 4.

The CNN is an external ONNX file, not part of this library.  What this wrapper changes against the reference:
  * execution providers: MIGraphX / ROCm (then CPU) instead of the reference's CUDAExecutionProvider;
  * with a GPU provider the class map NEVER visits the host: the image is normalised on the device (the reference's
    ToTensor + Normalize, same f32 arithmetic), the session runs with IOBinding on device pointers, and ``pred`` returns a
    ``DeviceMap`` -- an array-like ``(1, 1, H, W)`` whose ``.dev`` is the cuda tensor the accumulators hand to K1 / K1n
    directly (``np.asarray`` still gives the host copy, lazily, for viz and pickling);
  * without one (or ``keep_on_device=False``) it behaves as the reference: numpy in, numpy out.
"""
import numpy as np

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
GPU_PROVIDERS = ('MIGraphXExecutionProvider', 'ROCMExecutionProvider')


class DeviceMap:
    """Device-resident array with a lazy host view: indexable like the numpy array the reference returns."""

    def __init__(self, dev):
        self.dev = dev
        self._host = None

    @property
    def shape(self):
        return tuple(self.dev.shape)

    @property
    def dtype(self):
        return self.__array__().dtype

    def __getitem__(self, idx):
        return DeviceMap(self.dev[idx])

    def __array__(self, dtype=None, copy=None):
        if self._host is None:
            self._host = self.dev.cpu().numpy()
        return self._host if dtype is None else self._host.astype(dtype)

    def __len__(self):
        return self.dev.shape[0]


class SemSegONNX():

    def __init__(self, sem_onnx_path: str, keep_on_device=None):
        import onnxruntime as ort
        wanted = list(GPU_PROVIDERS) + ['CPUExecutionProvider']
        providers = [p for p in wanted if p in ort.get_available_providers()]
        self.ort_session_semseg = ort.InferenceSession(sem_onnx_path, providers=providers)
        on_gpu = any(p in GPU_PROVIDERS for p in self.ort_session_semseg.get_providers())
        self.keep_on_device = on_gpu if keep_on_device is None else bool(keep_on_device)
        self.accepts_device = self.keep_on_device      # the accumulators may pass a DeviceImage / cuda tensor
        self._bound = {}                               # tensors bound to the session stay alive until the next call

    # ---- reference behaviour (host) ------------------------------------------------------------------------------
    @staticmethod
    def input_preproc(rgb):
        """PIL / (H,W,3) u8 -> (3,H,W) f32, ImageNet-normalised: torchvision's ToTensor + Normalize in numpy
        (x / 255 in f32, then (x - mean) / std in f32)."""
        a = np.asarray(rgb, dtype=np.float32) / np.float32(255.)
        a = (a - np.asarray(MEAN, dtype=np.float32)) / np.asarray(STD, dtype=np.float32)
        return np.ascontiguousarray(np.transpose(a, (2, 0, 1)))

    def _pred_host(self, rgb):
        x = self.input_preproc(np.asarray(rgb))[None]
        name = self.ort_session_semseg.get_inputs()[0].name
        return self.ort_session_semseg.run(None, {name: x})[0]

    # ---- device path ---------------------------------------------------------------------------------------------
    @staticmethod
    def input_preproc_device(rgb_dev):
        """cuda u8 (H,W,3) -> cuda f32 (1,3,H,W), bit for bit the arithmetic of input_preproc (pca_image_to_nchw_f32:
        IEEE f32 divisions -- a torch expression on the GPU rounds its divisions differently)."""
        import ctypes as C

        import torch
        from pca_amd import _lib
        ctx = _lib.Context.get(rgb_dev.device)
        rgb_dev = rgb_dev.contiguous()
        H, W = int(rgb_dev.shape[0]), int(rgb_dev.shape[1])
        out = torch.empty((1, 3, H, W), dtype=torch.float32, device=rgb_dev.device)
        ctx.check(ctx.lib.pca_image_to_nchw_f32(ctx.h, rgb_dev.data_ptr(), H, W, (C.c_float * 3)(*MEAN), (C.c_float * 3)(*STD),
                                                out.data_ptr(), ctx.stream()))
        return out

    def _pred_device(self, rgb):
        import torch
        dev_img = getattr(rgb, 'dev', rgb)
        if not isinstance(dev_img, torch.Tensor):
            dev_img = torch.from_numpy(np.ascontiguousarray(np.asarray(rgb), dtype=np.uint8)).cuda()
        x = self.input_preproc_device(dev_img)
        sess = self.ort_session_semseg
        inp, out = sess.get_inputs()[0], sess.get_outputs()[0]
        H, W = int(x.shape[2]), int(x.shape[3])
        np_type = {'tensor(int64)': np.int64, 'tensor(int32)': np.int32, 'tensor(float)': np.float32,
                   'tensor(uint8)': np.uint8}.get(getattr(out, 'type', 'tensor(int64)'), np.int64)
        y = torch.empty((1, 1, H, W), dtype=getattr(torch, np.dtype(np_type).name), device=x.device)
        bind = sess.io_binding()
        idx = x.device.index or 0
        bind.bind_input(name=inp.name, device_type='cuda', device_id=idx, element_type=np.float32, shape=tuple(x.shape),
                        buffer_ptr=x.data_ptr())
        bind.bind_output(name=out.name, device_type='cuda', device_id=idx, element_type=np_type, shape=tuple(y.shape),
                         buffer_ptr=y.data_ptr())
        self._bound = {'x': x, 'y': y}
        torch.cuda.current_stream(x.device).synchronize()          # the session runs on its own stream
        sess.run_with_iobinding(bind)
        return DeviceMap(y)

    def pred(self, rgb):
        """(1, 1, H, W) class ids: numpy (reference behaviour) or, with a GPU provider, a DeviceMap."""
        if self.keep_on_device:
            return self._pred_device(rgb)
        return self._pred_host(rgb)
