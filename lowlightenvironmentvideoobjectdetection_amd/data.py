"""Synthetic dark clips and the reference's test-time preprocessing (Brighten -> Normalize -> Pad), on device.

  Brighten      mmtracking/mmtrack/datasets/pipelines/transforms.py:381-403 (gain shared per clip, :421-430)
  Normalize     mmdetection/mmdet/datasets/pipelines/transforms.py:547-575 (mean/std, BGR -> RGB)
  Pad           mmdetection/mmdet/datasets/pipelines/transforms.py:476-524 (zero pad to a multiple of 32)
  noise model   mmtracking/mmtrack/core/utils/add_noise.py:26-52 (Poisson + Gaussian + row noise in a 4-channel
                pseudo-RAW domain), used only to define the synthetic input distribution
These run once, outside the timed region (the reference's benchmark excludes data loading too)."""
import math

import torch


def synth_clean_clip(t, h, w, seed=0, device='cpu', first=0):
    """Smooth content + moving rectangles, uint8-valued BGR frames (t, h, w, 3) as float32. `first`: the frames are frames
    first .. first+t-1 of the clip the seed defines (a rank of a frame-sharded run synthesises only its segment and halo)."""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(1, 3, h // 16 + 2, w // 16 + 2, generator=g)
    bg = torch.nn.functional.interpolate(base, size=(h, w), mode='bicubic', align_corners=False).clamp(0, 1)[0]
    frames = bg.unsqueeze(0).repeat(t, 1, 1, 1) * 200 + 20
    nrect = 12
    pos = torch.rand(nrect, 2, generator=g)
    vel = (torch.rand(nrect, 2, generator=g) - 0.5) * 0.02
    size = torch.rand(nrect, 2, generator=g) * 0.2 + 0.05
    col = torch.rand(nrect, 3, generator=g) * 255
    for i in range(t):
        for r in range(nrect):
            cy, cx = (pos[r] + vel[r] * (first + i)) % 1.0
            y0, x0 = int(cy * h), int(cx * w)
            y1, x1 = min(h, y0 + int(size[r, 0] * h)), min(w, x0 + int(size[r, 1] * w))
            frames[i, :, y0:y1, x0:x1] = col[r].view(3, 1, 1)
    return frames.permute(0, 2, 3, 1).round().clamp(0, 255).contiguous().to(device)


def darken_and_add_noise(frames, mean_level=0.1, ratio=0.5, seed=0, first=None):
    """Scale to a dark exposure (mean ~ mean_level*255) and add the reference's noise model (add_noise.py:26-52):
    per RGGB channel Poisson gain a*ratio, Gaussian variance b*ratio, row noise variance c*ratio^3.
    first = k: the frames are frames k, k+1, .. of a longer clip and each is processed on its own (exposure from its own mean, noise
    from a generator seeded by (seed, frame index)), so that any rank of a frame-sharded run produces the same pixels for a frame."""
    if first is not None:
        return torch.cat([darken_and_add_noise(frames[i:i + 1], mean_level, ratio, seed * 1000003 + first + i) for i in range(frames.shape[0])], 0)
    g = torch.Generator(device='cpu').manual_seed(seed)
    t, h, w, _ = frames.shape
    x = frames.float().cpu()
    x = x * (mean_level * 255.0 / x.mean().clamp(min=1e-6))
    a = torch.tensor([10.43, 9.05, 16.68, 8.92]) * ratio
    b = torch.tensor([92.5, 80.0, 286.0, 80.0]) * ratio
    c = torch.tensor([0.08, 0.035, 0.25, 0.035]) * ratio ** 3
    # pseudo-RAW: B, G, R -> R G G B planes; noise per plane, averaged back for G
    bgr = x
    planes = torch.stack([bgr[..., 2], bgr[..., 1], bgr[..., 1], bgr[..., 0]], -1)          # R G G B
    pois = torch.poisson((planes / a).clamp(min=0), generator=g) * a
    gauss = torch.randn(planes.shape, generator=g) * b.sqrt()
    row = torch.randn(t, h, 1, 4, generator=g) * c.sqrt()
    noisy = (pois + gauss + row).clamp(0, 255)
    out = torch.stack([noisy[..., 3], 0.5 * (noisy[..., 1] + noisy[..., 2]), noisy[..., 0]], -1)   # back to B G R
    return out.round().clamp(0, 255)


def preprocess_clip(frames_bgr_u8, device, m=0.25, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375),
                    size_divisor=32, replicate_ends=2):
    """(T,H,W,3) BGR 0..255 -> (T + 2*replicate_ends, Hp, Wp, 8) fp16 NHWC on `device`:
    Brighten(m) with one gain for the clip, clip to [0,255] uint8 (floor of the cast), BGR->RGB, (x-mean)/std,
    zero pad bottom/right to a multiple of size_divisor, 3 real + 5 zero channels."""
    x = frames_bgr_u8.float().to(device)
    t, h, w, _ = x.shape
    amp = m * (h * w * 3) / (x[0] / 255.0).sum().clamp(min=1e-6)          # transforms.py:421-430: gain from the key frame
    x = (x * amp).clamp(0, 255).floor()                                    # np.clip(...).astype(uint8)
    x = x.flip(-1)                                                         # to_rgb
    x = (x - torch.tensor(mean, device=device)) / torch.tensor(std, device=device)
    hp, wp = (h + size_divisor - 1) // size_divisor * size_divisor, (w + size_divisor - 1) // size_divisor * size_divisor
    out = torch.zeros((t + 2 * replicate_ends, hp, wp, 8), device=device, dtype=torch.float16)
    out[replicate_ends:replicate_ends + t, :h, :w, :3] = x.half()
    for r in range(replicate_ends):
        out[r] = out[replicate_ends]
        out[-1 - r] = out[replicate_ends + t - 1]
    return out, (h, w, 3), (hp, wp, 3)


def preprocess_clip_device(frames_bgr_u8, m=0.25, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), size_divisor=32,
                           replicate_ends=2):
    """preprocess_clip on the device kernel (llv_brighten_normalize_pad_u8): frames (T,H,W,3) uint8 on the GPU ->
    (T + 2*replicate_ends, Hp, Wp, 8) fp16. One launch pair; no host arithmetic, no CPU fallback."""
    import ctypes as C
    from ._lib import LlvError, check, lib
    x = frames_bgr_u8
    if x.device.type != 'cuda' or x.dtype != torch.uint8:
        raise LlvError('preprocess_clip_device takes uint8 frames on the GPU')
    x = x.contiguous()
    t, h, w, _ = x.shape
    hp, wp = (h + size_divisor - 1) // size_divisor * size_divisor, (w + size_divisor - 1) // size_divisor * size_divisor
    out = torch.empty((t + 2 * replicate_ends, hp, wp, 8), device=x.device, dtype=torch.float16)
    ws = torch.zeros(1, dtype=torch.int64, device=x.device)
    mean_c, std_c = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    check(lib().llv_brighten_normalize_pad_u8(x.data_ptr(), t, h, w, hp, wp, float(m), mean_c, std_c, 1, out[replicate_ends:].data_ptr(),
                                              ws.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
          'llv_brighten_normalize_pad_u8')
    for r in range(replicate_ends):
        out[r] = out[replicate_ends]
        out[-1 - r] = out[replicate_ends + t - 1]
    return out, (h, w, 3), (hp, wp, 3)
