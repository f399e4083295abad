"""Checkpoint loading and saving in the reference's format: torch.save({'meta', 'state_dict', 'optimizer'}) with module-path
keys (mmcv CheckpointHook; call sites mmdetection/mmdet/apis/train.py:166-169, mmtracking/mmtrack/models/vid/base.py:22-37).
Remote schemes (torchvision://, open-mmlab://, http://) cannot be fetched here: they resolve only from a local
directory given by $LLV_CHECKPOINT_DIR; otherwise the caller's random-init scheme stays in effect, with a warning."""
import os
import warnings
from collections import OrderedDict

import torch


def _resolve(filename):
    for scheme in ('torchvision://', 'open-mmlab://', 'mmcls://', 'http://', 'https://'):
        if filename.startswith(scheme):
            d = os.environ.get('LLV_CHECKPOINT_DIR')
            name = filename[len(scheme):].rstrip('/').split('/')[-1]
            if d:
                for cand in (name, name + '.pth', name + '.pt'):
                    path = os.path.join(d, cand)
                    if os.path.isfile(path):
                        return path
            return None
    return filename


def load_state_dict_file(filename, map_location='cpu'):
    path = _resolve(filename)
    if path is None:
        return None
    if not os.path.isfile(path):
        raise IOError(f'{path} is not a checkpoint file')
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    if isinstance(ckpt, dict) and 'state_dict' in ckpt:
        sd = ckpt['state_dict']
    elif isinstance(ckpt, dict) and 'model' in ckpt:
        sd = ckpt['model']
    else:
        sd = ckpt
    if list(sd.keys()) and list(sd.keys())[0].startswith('module.'):
        sd = OrderedDict((k[7:], v) for k, v in sd.items())
    return sd


def load_checkpoint(model, filename, map_location='cpu', strict=False, logger=None):
    sd = load_state_dict_file(filename, map_location)
    if sd is None:
        warnings.warn(f'checkpoint {filename} is a remote reference and no local copy was found under '
                      f'$LLV_CHECKPOINT_DIR; keeping the current (random-init) weights')
        return None
    missing, unexpected = model.load_state_dict(sd, strict=strict)
    if (missing or unexpected) and not strict:
        warnings.warn(f'load_checkpoint({filename}): missing keys {list(missing)[:8]}..., unexpected {list(unexpected)[:8]}...')
    return sd


def save_checkpoint(model, filename, optimizer=None, meta=None):
    """mmcv.runner.save_checkpoint (CheckpointHook; call site mmdetection/mmdet/apis/train.py:166-169): torch.save of
    {'meta', 'state_dict', 'optimizer'} with module-path keys on the CPU and a DataParallel 'module.' prefix stripped, so that
    load_checkpoint above -- and the reference's own loader -- read it back."""
    sd = OrderedDict()
    for k, v in model.state_dict().items():
        sd[k[7:] if k.startswith('module.') else k] = v.detach().cpu()
    m = dict(meta or {})
    if getattr(model, 'CLASSES', None) is not None:
        m.setdefault('CLASSES', model.CLASSES)
    ckpt = {'meta': m, 'state_dict': sd}
    if isinstance(optimizer, dict):
        ckpt['optimizer'] = {k: o.state_dict() for k, o in optimizer.items()}
    elif optimizer is not None:
        ckpt['optimizer'] = optimizer.state_dict()
    d = os.path.dirname(os.path.abspath(filename))
    os.makedirs(d, exist_ok=True)
    with open(filename, 'wb') as f:
        torch.save(ckpt, f)
        f.flush()
    return filename
