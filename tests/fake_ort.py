"""A stand-in for the `onnxruntime` module (absent from this image) with the calls utils/onnx_utils.py makes: enough to
exercise the wrapper's host path and its device path (IOBinding on device pointers) with a deterministic "network".
The fake session finds the bound tensors through `session.owner._bound` (the wrapper keeps them alive there)."""
import types

import numpy as np

STATE = {'providers': ['ROCMExecutionProvider', 'CPUExecutionProvider'], 'runs': 0, 'bound_runs': 0}


def _network_numpy(x):
    """(1,3,H,W) f32 -> (1,1,H,W) int64 class ids in 0..18."""
    return (np.floor(np.abs(x[:, 0] * 5.0 + x[:, 1] * 3.0 + x[:, 2] * 7.0)).astype(np.int64) % 19)[:, None]


def _network_torch(x):
    import torch
    return (torch.floor(torch.abs(x[:, 0] * 5.0 + x[:, 1] * 3.0 + x[:, 2] * 7.0)).to(torch.int64) % 19)[:, None]


class _Binding:
    def __init__(self):
        self.inputs, self.outputs = {}, {}

    def bind_input(self, name, device_type, device_id, element_type, shape, buffer_ptr):
        assert device_type == 'cuda' and element_type == np.float32
        self.inputs[name] = (tuple(shape), buffer_ptr)

    def bind_output(self, name, device_type, device_id, element_type, shape, buffer_ptr):
        assert device_type == 'cuda'
        self.outputs[name] = (tuple(shape), buffer_ptr, element_type)


class InferenceSession:
    def __init__(self, path, providers=None):
        self.path, self._providers = path, list(providers or [])
        self.owner = None

    def get_providers(self):
        return self._providers

    def get_inputs(self):
        return [types.SimpleNamespace(name='input', type='tensor(float)')]

    def get_outputs(self):
        return [types.SimpleNamespace(name='seg', type='tensor(int64)')]

    def run(self, names, feed):
        STATE['runs'] += 1
        return [_network_numpy(feed['input'])]

    def io_binding(self):
        return _Binding()

    def run_with_iobinding(self, b):
        STATE['bound_runs'] += 1
        x, y = self.owner._bound['x'], self.owner._bound['y']
        assert b.inputs['input'] == (tuple(x.shape), x.data_ptr())
        assert b.outputs['seg'][:2] == (tuple(y.shape), y.data_ptr()) and b.outputs['seg'][2] == np.int64
        y.copy_(_network_torch(x))


def get_available_providers():
    return list(STATE['providers'])
