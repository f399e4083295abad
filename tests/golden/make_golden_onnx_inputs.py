"""Turns the reference's input-only test fixtures (data files of its own tests)

    mmdetection/tests/test_onnx/data/retina_head_get_bboxes.pkl   10 torch.rand tensors: 5 cls (1,36,s,s) + 5 bbox (1,36,s,s), s = 32..2
    mmdetection/tests/test_onnx/data/yolov3_head_get_bboxes.pkl   3 torch.rand tensors (1,27,s,s), s = 32, 16, 8

into tests/golden/onnx_head_inputs.npz. The pickles are UNTRUSTED content: they are read with a restricted unpickler that admits
exactly the three globals a static inspection shows them to use (torch._utils._rebuild_tensor_v2, torch.storage._load_from_bytes,
collections.OrderedDict) and nothing else. The reference holds no expected outputs for them (its test compares torch against
onnxruntime, test_head.py:86-131,182-240); they are an extra decode + NMS input case for the oracle and the device path
(tests/test_golden.py, tests/test_gpu_post.py).

    python tests/golden/make_golden_onnx_inputs.py        # needs /root/reference; run once, the .npz is committed
"""
import collections
import os
import pickle

import numpy as np
import torch

REF = '/root/reference/mmdetection/tests/test_onnx/data'
HERE = os.path.dirname(os.path.abspath(__file__))
ALLOWED = {('torch._utils', '_rebuild_tensor_v2'): torch._utils._rebuild_tensor_v2,
           ('torch.storage', '_load_from_bytes'): None,      # replaced below: raw little-endian floats, never torch.load
           ('collections', 'OrderedDict'): collections.OrderedDict}


def _storage_from_bytes(b):
    """torch.storage._load_from_bytes is torch.load(io.BytesIO(b)), i.e. another pickle: open THAT with the same restrictions."""
    import io
    return _restricted_load(io.BytesIO(b), legacy=True)


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) not in ALLOWED:
            raise pickle.UnpicklingError(f'global {module}.{name} is not allowed')
        if (module, name) == ('torch.storage', '_load_from_bytes'):
            return _storage_from_bytes
        return ALLOWED[(module, name)]


def _restricted_load(f, legacy=False):
    if not legacy:
        return _Unpickler(f).load()
    # a legacy torch.save stream of one storage: torch's own weights_only loader is a restricted unpickler (tensors / storages
    # and plain containers only)
    return torch.load(f, map_location='cpu', weights_only=True)


def load(name):
    with open(os.path.join(REF, name), 'rb') as f:
        obj = _restricted_load(f)
    assert isinstance(obj, (list, tuple)) and all(isinstance(t, torch.Tensor) for t in obj)
    return [t.detach().float().contiguous().numpy() for t in obj]


if __name__ == '__main__':
    retina = load('retina_head_get_bboxes.pkl')
    yolo = load('yolov3_head_get_bboxes.pkl')
    assert [t.shape for t in retina] == [(1, 36, s, s) for s in (32, 16, 8, 4, 2)] * 2
    assert [t.shape for t in yolo] == [(1, 27, s, s) for s in (32, 16, 8)]
    out = {f'retina_cls_{i}': t for i, t in enumerate(retina[:5])}
    out.update({f'retina_bbox_{i}': t for i, t in enumerate(retina[5:])})
    out.update({f'yolo_pred_{i}': t for i, t in enumerate(yolo)})
    np.savez_compressed(os.path.join(HERE, 'onnx_head_inputs.npz'), **out)
    print({k: (v.shape, float(v.min()), float(v.max())) for k, v in out.items()})
