"""Training-time noise injection of the authors' video models: general_clean_noise_pairs
(mmtracking/mmtrack/core/utils/add_noise.py:13-75; called by fgfa.py:144 / llvod.py:91 / selsa_fastdvd_darkfarm.py, whose imports of it are
commented out in the reference). Element-wise sampling on whatever device the frames live on (torch.distributions, as the reference);
the reference's plotting / PSNR printing side effects are not reproduced.

Input: un-normalised BGR frames (b, 3, h, w) in 0..255. Pseudo-RAW R G B G planes scaled by `am`; Poisson noise with per-plane gain
a * ratio, Gaussian noise with variance b * ratio (times the gain), a per-row multiplicative factor N(1, c * ratio^3); clamp to
0..255; both outputs normalised with the reference's (BGR-ordered) mean / std (:71-72)."""
import random

import torch
from torch.distributions.normal import Normal
from torch.distributions.poisson import Poisson

POISSON = (10.4304723, 9.05125669, 16.68142166, 8.92170499)
GAUSSIAN = (92.5, 80.0, 286.0, 80.0)
ROW = (0.08, 0.035, 0.25, 0.035)
MEAN, STD = (103.53, 116.28, 123.675), (57.375, 57.12, 58.395)


def normalize(tensor, mean, std):
    """add_noise.py:78-110 (torchvision's Normalize on a batch)."""
    mean = torch.as_tensor(mean, dtype=tensor.dtype, device=tensor.device).view(1, -1, 1, 1)
    std = torch.as_tensor(std, dtype=tensor.dtype, device=tensor.device).view(1, -1, 1, 1)
    return (tensor - mean) / std


def general_clean_noise_pairs(clean, constant=None):
    device = clean.device
    bgr = clean.permute(0, 2, 3, 1).float()
    bs, h, w, _ = bgr.shape
    if isinstance(constant, (tuple, list)) and len(constant) == 2:
        ratio, am = float(constant[0]), float(constant[1])
    else:
        ratio, am = random.uniform(0, 1), random.uniform(0, 1)
    pois = torch.tensor(POISSON, device=device).view(1, 1, 1, 4) * ratio
    gaus = torch.tensor(GAUSSIAN, device=device).view(1, 1, 1, 4) * ratio
    wi = torch.tensor(ROW, device=device).view(1, 1, 4) * ratio ** 3
    b, g, r = torch.split(bgr, [1, 1, 1], 3)
    raw = torch.cat([r, g, b, g], 3) * am
    pnoisy = pois * Poisson(rate=raw / pois).sample()
    gnoisy = pois * Normal(loc=torch.zeros_like(raw), scale=torch.sqrt(gaus).expand_as(raw)).sample()
    k_noisy = Normal(loc=torch.ones(bs, h, 4, device=device), scale=torch.sqrt(wi).expand(bs, h, 4)).sample()
    noisy = (pnoisy + gnoisy) * k_noisy.view(bs, h, 1, 4)
    _r, _g, _b, _ = torch.split(noisy, [1, 1, 1, 1], 3)
    rgb_noise = torch.clamp(torch.cat([_r, _g, _b], 3), 0, 255.0)
    rgb_clean = torch.clamp(torch.cat([r, g, b], 3), 0, 255.0)
    noise = normalize(rgb_noise.permute(0, 3, 1, 2).contiguous(), MEAN, STD)
    clean = normalize(rgb_clean.permute(0, 3, 1, 2).contiguous(), MEAN, STD)
    return noise, clean
