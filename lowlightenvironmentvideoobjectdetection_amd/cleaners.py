"""FastDVDnet enhancement CNN with the reference's structure and parameter names
(mmtracking/mmtrack/models/cleaners/FastDVDnet.py:7-149). The nn.Sequential containers exist so that
state_dict keys match the reference's checkpoints (temp1.inc.convblock.0.weight ...); forward is a HIP plan."""
import torch
import torch.nn as nn

from . import engine, ops
from .planned import PlannedModule
from .registry import CLEANER


class CvBlock(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.convblock = nn.Sequential(nn.Conv2d(in_ch, out_ch, 3, padding=1, bias=False), nn.BatchNorm2d(out_ch),
                                       nn.ReLU(inplace=True),
                                       nn.Conv2d(out_ch, out_ch, 3, padding=1, bias=False), nn.BatchNorm2d(out_ch),
                                       nn.ReLU(inplace=True))


class InputCvBlock(nn.Module):
    def __init__(self, num_in_frames, out_ch):
        super().__init__()
        self.interm_ch = 30
        self.convblock = nn.Sequential(
            nn.Conv2d(num_in_frames * 3, num_in_frames * self.interm_ch, 3, padding=1, groups=num_in_frames, bias=False),
            nn.BatchNorm2d(num_in_frames * self.interm_ch), nn.ReLU(inplace=True),
            nn.Conv2d(num_in_frames * self.interm_ch, out_ch, 3, padding=1, bias=False), nn.BatchNorm2d(out_ch),
            nn.ReLU(inplace=True))


class DownBlock(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.convblock = nn.Sequential(nn.Conv2d(in_ch, out_ch, 3, padding=1, stride=2, bias=False), nn.BatchNorm2d(out_ch),
                                       nn.ReLU(inplace=True), CvBlock(out_ch, out_ch))


class UpBlock(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.convblock = nn.Sequential(CvBlock(in_ch, in_ch), nn.Conv2d(in_ch, out_ch * 4, 3, padding=1, bias=False),
                                       nn.PixelShuffle(2))


class OutputCvBlock(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.convblock = nn.Sequential(nn.Conv2d(in_ch, in_ch, 3, padding=1, bias=False), nn.BatchNorm2d(in_ch),
                                       nn.ReLU(inplace=True), nn.Conv2d(in_ch, out_ch, 3, padding=1, bias=False))


class DenBlock(nn.Module):
    def __init__(self, num_input_frames=3):
        super().__init__()
        self.chs_lyr0, self.chs_lyr1, self.chs_lyr2 = 32, 64, 128
        self.inc = InputCvBlock(num_in_frames=num_input_frames, out_ch=self.chs_lyr0)
        self.downc0 = DownBlock(self.chs_lyr0, self.chs_lyr1)
        self.downc1 = DownBlock(self.chs_lyr1, self.chs_lyr2)
        self.upc2 = UpBlock(self.chs_lyr2, self.chs_lyr1)
        self.upc1 = UpBlock(self.chs_lyr1, self.chs_lyr0)
        self.outc = OutputCvBlock(self.chs_lyr0, 3)


@CLEANER.register_module()
class FastDVDnet(PlannedModule):
    """FastDVDnet.py:119-149. forward(x): x (N, 15, H, W) -- five stacked frames per sample -> (N, 3, H, W)."""

    def __init__(self, num_input_frames=5):
        super().__init__()
        if num_input_frames != 5:
            raise NotImplementedError('FastDVDnet is defined for 5 input frames')
        self.num_input_frames = num_input_frames
        self.temp1 = DenBlock(3)
        self.temp2 = DenBlock(3)
        self.reset_params()

    def reset_params(self):             # FastDVDnet.py:128-136
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, nonlinearity='relu')
        self.invalidate_plans()

    def make_plan(self, batch, height, width, device, frames=None, out=None):
        return engine.FastDVDnetPlan(self.state_dict(), batch, height, width, device, frames=frames, out=out)

    def forward(self, x):
        """Reference call signature: each of the N samples carries its own 5-frame window, so the windows are
        unrelated and every sample is denoised by its own pass (batch-1 plan, 3 + 1 DenBlocks)."""
        self._check_input(x)
        n, c, h, w = x.shape
        assert c == 15, 'FastDVDnet takes 5 stacked RGB frames (15 channels)'
        plan = self._plan((h, w, str(x.device)), lambda: self.make_plan(1, h, w, x.device))
        outs = []
        for i in range(n):
            ops.pack_input(x[i].float().reshape(5, 3, h, w).contiguous(), 8, out=plan.frames)
            plan.plan.run()
            outs.append(ops.unpack_output(plan.out, 3))
        return torch.cat(outs, 0)


from .backbones import ResNet  # noqa: E402


@CLEANER.register_module()
class ResCleaner(ResNet):
    """mmtracking/mmtrack/models/cleaners/resclean.py:6-20: "ResNet cleaner for feature extracting to supervise noise feature" -- mmdet's
    ResNet (depth 50 / 101) under the CLEANER registry, fed the clean frames; the authors' `llvod_l1234*` configs build it as
    `cleaner=dict(type='ResCleaner', depth=50, ...)` and freeze it (`frozen_modules='cleaner'`). Forward = the ResNet launch plan."""

    def __init__(self, depth=50, **kwargs):
        if depth not in (50, 101):
            raise KeyError(f'invalid depth {depth} for resnet')          # resclean.py:13-16: arch_settings holds 50 and 101 only
        super().__init__(depth, **kwargs)
