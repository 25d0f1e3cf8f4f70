"""Semantic-segmentation model wrapper (reference: utils/onnx_utils.py).  The CNN is an external ONNX
file and not part of this library; on ROCm the session is opened with the MIGraphX / ROCm execution
providers instead of the reference's CUDAExecutionProvider."""
import numpy as np


class SemSegONNX():

    MEAN = np.array((0.485, 0.456, 0.406), dtype=np.float32)
    STD = np.array((0.229, 0.224, 0.225), dtype=np.float32)

    def __init__(self, sem_onnx_path: str):
        import onnxruntime as ort
        wanted = ['MIGraphXExecutionProvider', 'ROCMExecutionProvider', 'CPUExecutionProvider']
        providers = [p for p in wanted if p in ort.get_available_providers()]
        self.ort_session_semseg = ort.InferenceSession(sem_onnx_path, providers=providers)

    def input_preproc(self, rgb):
        """PIL / (H,W,3) u8 -> (3,H,W) f32, ImageNet-normalised (ToTensor + Normalize)."""
        a = np.asarray(rgb, dtype=np.float32) / 255.
        return np.transpose((a - self.MEAN) / self.STD, (2, 0, 1))

    def pred(self, rgb) -> np.array:
        x = self.input_preproc(rgb)[None]
        name = self.ort_session_semseg.get_inputs()[0].name
        return self.ort_session_semseg.run(None, {name: x})[0]
