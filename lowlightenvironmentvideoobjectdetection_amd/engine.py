"""Launch plans: a model with fixed input geometry becomes a flat list of C-ABI launches over
preallocated NHWC fp16 buffers, replayable eagerly or from a captured HIP graph.

Nothing here computes: packing (BN folding, fp16 conversion, channel padding) happens once at build
time on the host, everything per-frame is a libllv.so launch on the current HIP stream.

Reference structure followed (relative to /root/reference):
  ResNet / Bottleneck          mmdetection/mmdet/models/backbones/resnet.py:260-299,622-640
  FPN                          mmdetection/mmdet/models/necks/fpn.py:170-221
  RetinaHead                   mmdetection/mmdet/models/dense_heads/retina_head.py:93-114
  get_bboxes                   mmdetection/mmdet/models/dense_heads/anchor_head.py:495-726
  DenBlock / FastDVDnet        mmtracking/mmtrack/models/cleaners/FastDVDnet.py:77-149
"""
import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import ops
from ._lib import (CONV_ACT_FIRST, CONV_CLSMAX, CONV_LEAKY, CONV_OUT_F32, CONV_PIXSHUF, CONV_RELU, CONV_RES_ADD, CONV_RES_RSUB, CONV_RES_UPS, LlvError, check,
                   lib)

BN_EPS = 1e-5
# direct (halo-in-LDS) 3x3 kernel for the FastDVDnet layers: on by default; LLV_HALO=0 falls back to implicit GEMM
import os as _os
HALO = _os.environ.get('LLV_HALO', '1') != '0'
HALO_MAX_COUT = int(_os.environ.get('LLV_HALO_MAX_COUT', '128'))
PS_EMAJOR = _os.environ.get('LLV_PS_EMAJOR', '1') != '0'
FUSE_INPUTCV = _os.environ.get('LLV_FUSE_INPUTCV', '1') != '0'
FUSE_OUTPUTCV = _os.environ.get('LLV_FUSE_OUTPUTCV', '1') != '0'
ARCH = {18: ('basic', (2, 2, 2, 2)), 34: ('basic', (3, 4, 6, 3)), 50: ('bottleneck', (3, 4, 6, 3)),
        101: ('bottleneck', (3, 4, 23, 3)), 152: ('bottleneck', (3, 8, 36, 3))}


def _bn(sd, p):
    return (sd[p + '.weight'], sd[p + '.bias'], sd[p + '.running_mean'], sd[p + '.running_var'])


class Plan:
    """An ordered list of launches. run() replays it on the current stream; capture() records it into a
    HIP graph (torch.cuda.CUDAGraph drives hipStreamBeginCapture on ROCm) so a frame is one graph launch."""

    def __init__(self, device):
        self.device = device
        self.steps = []          # (name, fn, args, flops, bytes)
        self.keep = []           # tensors the launches point into
        self.kernel_of = {}      # step name -> device kernel it launches (for per-kernel rooflines)
        self.graph = None
        self.epoch = 0           # bumped by every run / replay: which generation of values the buffers hold

    def add(self, name, fn, args, flops=0.0, nbytes=0.0, kernel=None):
        self.steps.append((name, fn, args, flops, nbytes))
        if kernel:
            self.kernel_of[name] = kernel

    def extend(self, other: 'Plan'):
        self.steps += other.steps
        self.keep += other.keep
        self.kernel_of.update(other.kernel_of)

    def run(self):
        self.epoch += 1          # the buffers are about to be rewritten (ops._nhwc_source)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for name, fn, args, _, _ in self.steps:
            rc = fn(*args, stream)
            if rc != 0:
                raise LlvError(f'{name}: {lib().llv_strerror(rc).decode()} ({rc})')

    def capture(self):
        self.run()               # warm-up: one-time attribute calls must not happen inside capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.run()
        self.graph = g
        return g

    def replay(self):
        if self.graph is None:
            self.run()
        else:
            self.epoch += 1
            self.graph.replay()

    def run_timed(self, iters=3):
        """Eager replay with a HIP event pair around every launch (on the stream the launches go to).
        Returns [(name, avg ms, flops, bytes)] per step. Diagnostic: not the throughput path."""
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        evs = []
        for _ in range(iters):
            row = []
            for name, fn, args, _, _ in self.steps:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = fn(*args, stream)
                e1.record()
                if rc != 0:
                    raise LlvError(f'{name}: {lib().llv_strerror(rc).decode()} ({rc})')
                row.append((e0, e1))
            evs.append(row)
        torch.cuda.synchronize()
        out = []
        for i, (name, _, _, fl, nb) in enumerate(self.steps):
            ms = sum(r[i][0].elapsed_time(r[i][1]) for r in evs) / iters
            out.append((name, ms, fl, nb))
        return out

    def check_extents(self):
        """Replay, on the host, the addresses of every convolution launch of the plan (llv_conv2d_extents /
        llv_convpair_extents) against the sizes of the buffers behind its descriptor. Returns [(step, pointer, needed, declared)]
        for every violation: an empty list means no launch reads or writes outside its tensors."""
        bad = []
        for name, fn, args, _, _ in self.steps:
            if getattr(fn, '__name__', '') in ('llv_conv2d_nhwc_fwd', 'llv_inputcvblock_fwd', 'llv_outputcvblock_fwd'):
                for (ptr, need, have) in ops.extents_violations(args[0]._obj):
                    bad.append((name, ptr, need, have))
            elif getattr(fn, '__name__', '') == 'llv_bottleneck_fwd':
                # closed form (bottleneck.hip): in-image pixels x all channels of x and y, whole weight / bias arrays
                d = args[0]._obj
                c, px = 4 * d.mid, d.batch * d.h * d.w
                need = dict(x=((px - 1) * d.x_ld + c) * 2, y=((px - 1) * d.y_ld + c) * 2, w1=d.mid * c * 2, w2=9 * d.mid * d.mid * 2,
                            w3=c * d.mid * 2, b1=d.mid * 4, b2=d.mid * 4, b3=c * 4)
                for k, v in need.items():
                    if v > d._have[k]:
                        bad.append((name, k, v, d._have[k]))
        return bad

    @property
    def flops(self):
        return sum(s[3] for s in self.steps)

    def conv_steps(self):
        return [s for s in self.steps if s[3] > 0]


class Builder:
    def __init__(self, device):
        self.dev = device
        self.plan = Plan(device)
        self._fn_conv = lib().llv_conv2d_nhwc_fwd
        self._ws = None              # one split-K workspace shared by all launches of the plan (they run in order)
        self._ws_descs = []
        self.tape = []               # every conv launch in order (training: the backward pass walks it in reverse)
        # kernel selection as if the launch had this many images (None: its own batch). The planner follows the launch size (tile
        # family, split K on launches that cannot fill the chip), and tile / split decide the order a pixel's sum is accumulated
        # in; two launches of different batch whose results must agree bit for bit select for the same batch (FastDVDnetPlan)
        self.select_batch = None
        self.window_descs = []       # fused-CvBlock descriptors that read the clip window directly (FastDVDnetPlan.bind_clip)

    def buf(self, *shape, dtype=torch.float16):
        t = torch.zeros(shape, device=self.dev, dtype=dtype)
        self.plan.keep.append(t)
        return t

    def conv_desc(self, name, pc: ops.PackedConv, xs, y, batch, segs, x_ld, y_ld, res=None, res_ld=0, flags=0,
                  res_hw=(0, 0), tile=0, y_group_stride=0, act_slope=0.0, ps_out=(0, 0), clsmax=None):
        pc.to(self.dev)
        d = ops.make_conv_desc(pc, xs, y, batch, segs, x_ld, y_ld, res, res_ld, flags, res_hw, tile, 0, y_group_stride,
                               act_slope, ps_out, clsmax)
        if clsmax is not None:
            self.plan.keep += [clsmax[0], clsmax[1]]
        if self.select_batch is not None and self.select_batch != batch and tile == 0:
            d.batch = self.select_batch
            t_, k_ = C.c_int(0), C.c_int(0)
            check(lib().llv_conv2d_plan(C.byref(d), C.byref(t_), C.byref(k_)), 'llv_conv2d_plan')
            d.batch = batch
            if t_.value > 0:          # -1: the register-staged halo kernel, which has no variants
                d.tile, d.k_splits = t_.value, k_.value
        m = sum(batch * s[2] * s[3] for s in segs)
        flops = 2.0 * m * pc.true_macs_per_pixel
        # algorithmic bytes of the launch: input once, weights once, output once in its own type, the residual once where there is one
        # (round 3 left the residual and the fp32 output's second half out: the identity read of a Bottleneck's conv3 made its counter
        # bytes look like 1.9x "over-fetch", which it is not)
        out_b = 4.0 if (flags & CONV_OUT_F32) else 2.0
        nbytes = 2.0 * sum(batch * s[0] * s[1] for s in segs) * pc.cin * pc.groups + out_b * m * pc.cout * pc.groups + pc.w.numel() * 2.0
        if res is not None:
            nbytes += 2.0 * pc.cout * pc.groups * (batch * res_hw[0] * res_hw[1] if (flags & CONV_RES_UPS) else m)
        if clsmax is not None:          # the mandatory part only: maxima + flags; the stored logit blocks depend on the data
            nbytes += m * (4.0 * (pc.cout // clsmax[2]) + (pc.cout + 127) // 128) - out_b * m * pc.cout * pc.groups
        wsb = ops.conv_workspace_bytes(d)
        if wsb:
            if self._ws is None or self._ws.numel() < wsb:
                self._ws = torch.empty((wsb,), device=self.dev, dtype=torch.uint8)
                self.plan.keep.append(self._ws)
                for od in self._ws_descs:
                    od.workspace = self._ws.data_ptr()
                    od._ext.workspace = self._ws.numel()
            d.workspace, d.workspace_bytes = self._ws.data_ptr(), wsb
            d._ext.workspace = self._ws.numel()
            self._ws_descs.append(d)
        self.plan.keep += [pc, pc.w, pc.bias, d] + list(xs) + [y] + ([res] if res is not None else [])
        self.plan.add(name, self._fn_conv, (C.byref(d),), flops, nbytes, kernel=ops.conv_kernel_name(d))
        self.tape.append(dict(name=name, pc=pc, xs=list(xs), y=y, batch=batch, segs=list(segs), x_ld=x_ld, y_ld=y_ld, res=res,
                              res_ld=res_ld, flags=flags, res_hw=res_hw, desc=d, act_slope=act_slope))
        return d

    def conv(self, name, x, pc: ops.PackedConv, flags=0, res=None, out=None, tile=0):
        """x: (N,H,W,C) tensor or list of n_src tensors -> new (N,Ho,Wo,Cout) buffer."""
        xs = list(x) if isinstance(x, (list, tuple)) else [x]
        n, h, w, cx = xs[0].shape
        ho, wo = ops.conv_out_size(h, w, pc)
        ctot = pc.groups * pc.cout
        if out is None:
            if flags & CONV_PIXSHUF:
                out = self.buf(n, 2 * ho, 2 * wo, ctot // 4)
            else:
                out = self.buf(n, ho, wo, ctot, dtype=torch.float32 if flags & CONV_OUT_F32 else torch.float16)
        res_hw = (res.shape[1], res.shape[2]) if (res is not None and flags & CONV_RES_UPS) else (0, 0)
        self.conv_desc(name, pc, xs, out, n, [(h, w, ho, wo, 0, 0)], cx, out.shape[-1], res,
                       res.shape[-1] if res is not None else 0, flags, res_hw, tile)
        return out

    def bottleneck(self, name, x, pc1, pc2, pc3):
        """A stride-1 Bottleneck without downsample as ONE launch (llv_bottleneck_fwd): both narrow intermediates stay in LDS."""
        from ._lib import BottleneckDesc
        n, h, w, c = x.shape
        mid = pc1.cout
        assert pc3.cout == c == 4 * mid and pc2.cin == mid and pc1.k_pad == c and pc2.k_pad == 9 * mid and pc3.k_pad == mid
        out = self.buf(n, h, w, c)
        for pc in (pc1, pc2, pc3):
            pc.to(self.dev)
        d = BottleneckDesc(x=x.data_ptr(), y=out.data_ptr(), w1=pc1.w.data_ptr(), w2=pc2.w.data_ptr(), w3=pc3.w.data_ptr(),
                           b1=pc1.bias.data_ptr(), b2=pc2.bias.data_ptr(), b3=pc3.bias.data_ptr(),
                           batch=n, h=h, w=w, mid=mid, x_ld=c, y_ld=c)
        d._have = dict(x=ops.tensor_extent(x), y=ops.tensor_extent(out), w1=ops.tensor_extent(pc1.w), w2=ops.tensor_extent(pc2.w),
                       w3=ops.tensor_extent(pc3.w), b1=ops.tensor_extent(pc1.bias), b2=ops.tensor_extent(pc2.bias), b3=ops.tensor_extent(pc3.bias))
        px = n * h * w
        flops = 2.0 * px * (pc1.true_macs_per_pixel + pc2.true_macs_per_pixel + pc3.true_macs_per_pixel)
        nbytes = 2.0 * 2 * px * c + 2.0 * (pc1.w.numel() + pc2.w.numel() + pc3.w.numel())
        self.plan.keep += [pc1, pc2, pc3, d, x, out]
        self.plan.add(name, lib().llv_bottleneck_fwd, (C.byref(d),), flops, nbytes, kernel='bottleneck_kernel')
        return out

    def maxpool(self, name, x):
        n, h, w, c = x.shape
        out = self.buf(n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c)
        self.plan.add(name, lib().llv_maxpool3x3s2_nhwc, (C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), n, h, w, c),
                      0.0, 2.0 * (x.numel() + out.numel()))
        # training tape: a parameter-free op between convolutions (train_engine.TrainGraph._finish); only walked when the stem trains
        self.tape.append(dict(name=name, custom=True, xs=[x], y=out, bwd=_maxpool_bwd))
        return out


def _maxpool_bwd(g, op):
    x, y = op['xs'][0], op['y']
    gx, gy = g.gview(x), g.gview(y)
    n, h, w, c = x.shape
    check(lib().llv_maxpool3x3s2_nhwc_bwd(ops._ptr(x), ops._ptr(gy), ops._ptr(gx), n, h, w, c, x.shape[-1], gy.shape[-1], gx.shape[-1],
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'llv_maxpool3x3s2_nhwc_bwd')


# ------------------------------------------------------------------------------------------------
# RetinaNet stages (ResNet, FPN, RetinaHead, decode + NMS)
# ------------------------------------------------------------------------------------------------
def _cpu_sd(sd):
    out = {}
    for k, v in sd.items():
        t = v.detach().float().cpu()
        if t is v:
            t = t.clone()
        t._llv_key = k           # pack_conv records which parameters a launch was packed from (PackedConv.parts)
        out[k] = t
    return out


def build_resnet(b: Builder, sd, x, depth=50, strides=(1, 2, 2, 2), style='pytorch', out_indices=(0, 1, 2, 3), p='',
                 num_stages=4, after_stage=None, dilations=(1, 1, 1, 1), fuse_blocks=False, dark_stages=()):
    """x: (N,H,W,8) fp16 NHWC normalised image (3 real channels). Returns the out_indices stage outputs.
    resnet.py:622-640 with every BatchNorm folded (norm_eval, running stats) and ReLU / identity-add fused
    into the producing conv's epilogue. fuse_blocks (inference plans only: the training tape needs every activation) runs
    the stride-1 blocks of layer1 / layer2 as one launch each (Builder.bottleneck)."""
    kind, blocks = ARCH[depth]
    blocks = blocks[:num_stages]
    # stem: 7x7/s2 conv (Cin 3 -> padded 8) + BN + ReLU, then 3x3/s2 max pool (resnet.py:559-570)
    x = b.conv('stem.conv1', x, ops.pack_conv(sd[p + 'conv1.weight'], None, 2, 3, bn=_bn(sd, p + 'bn1'), eps=BN_EPS), CONV_RELU)
    x = b.maxpool('stem.maxpool', x)
    feats = []
    for i, nb in enumerate(blocks):
        for j in range(nb):
            q = f'{p}layer{i + 1}.{j}.'
            stride = strides[i] if j == 0 else 1
            dil = dilations[i]                               # resnet.py:151-167: conv2 padding = dilation (DC5 stages)
            identity = x
            # conv3 and a stride-1 downsample conv over as many channels as conv2 writes are ONE 1x1 convolution over the two
            # tensors side by side (K = [conv2 out | block input], weights [W3' | Wd'] with both BatchNorms folded, biases added):
            # the (4 x MID)-channel identity tensor is neither written nor read back. Inference plans only (fuse_blocks): the training
            # tape keeps a launch per parameter tensor. R-50's layer1.0 (64 | 64 channels at 184 x 320): 403 -> 160 us at 16 frames.
            kcat = (kind == 'bottleneck' and fuse_blocks and i not in dark_stages and (q + 'downsample.0.weight') in sd and stride == 1 and
                    sd[q + 'downsample.0.weight'].shape[1] == sd[q + 'conv3.weight'].shape[1] and x.shape[-1] == sd[q + 'conv3.weight'].shape[1]
                    and sd[q + 'conv3.weight'].shape[1] % 32 == 0 and _os.environ.get('LLV_KCAT', '1') != '0')
            # the strided first blocks (pytorch style: stride on conv2) do the same over THREE sources of MID channels each: conv2 writes
            # its MID channels into rows of 2 MID, the block input's even pixels (what the stride-2 downsample conv reads) are gathered
            # into a dense 2 MID-channel map by llv_subsample2_nhwc and enter as two halves (equal widths and pitches are what the
            # multi-source launch wants). R-50 layer2.0 / 3.0 / 4.0 at 16 frames: 287 / 200 / 168 us -> ~150 / 110 / 95.
            mid = sd[q + 'conv3.weight'].shape[1] if kind == 'bottleneck' else 0
            kcat3 = (kind == 'bottleneck' and fuse_blocks and i not in dark_stages and (q + 'downsample.0.weight') in sd and stride == 2 and
                     style == 'pytorch' and dil == 1 and x.shape[-1] == 2 * mid and sd[q + 'downsample.0.weight'].shape[1] == 2 * mid and
                     mid % 32 == 0 and _os.environ.get('LLV_KCAT', '1') != '0')
            if (q + 'downsample.0.weight') in sd and not kcat and not kcat3:          # res_layer.py:36-58: conv1x1(stride) + BN
                identity = b.conv(q + 'downsample', x, ops.pack_conv(sd[q + 'downsample.0.weight'], None, stride, 0,
                                                                     bn=_bn(sd, q + 'downsample.1'), eps=BN_EPS))
            if kind == 'bottleneck' and fuse_blocks and i not in dark_stages and identity is x and stride == 1 and dil == 1 and \
                    sd[q + 'conv1.weight'].shape[0] in (64, 128) and x.shape[-1] == 4 * sd[q + 'conv1.weight'].shape[0]:
                x = b.bottleneck(q + 'fused', x,
                                 ops.pack_conv(sd[q + 'conv1.weight'], None, 1, 0, bn=_bn(sd, q + 'bn1'), eps=BN_EPS),
                                 ops.pack_conv(sd[q + 'conv2.weight'], None, 1, 1, bn=_bn(sd, q + 'bn2'), eps=BN_EPS),
                                 ops.pack_conv(sd[q + 'conv3.weight'], None, 1, 0, bn=_bn(sd, q + 'bn3'), eps=BN_EPS))
            elif kind == 'bottleneck' and i in dark_stages:
                x = _denoising_block(b, sd, q, x, identity, stride, dil, style)
            elif kind == 'bottleneck':
                s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)          # resnet.py:151-156
                o = b.conv(q + 'conv1', x, ops.pack_conv(sd[q + 'conv1.weight'], None, s1, 0, bn=_bn(sd, q + 'bn1'),
                                                         eps=BN_EPS), CONV_RELU)
                pc2 = ops.pack_conv(sd[q + 'conv2.weight'], None, s2, dil, bn=_bn(sd, q + 'bn2'), eps=BN_EPS, dilation=dil)
                if kcat3:
                    n_, h_, w_, c_ = o.shape
                    ho_, wo_ = ops.conv_out_size(h_, w_, pc2)
                    t2 = b.buf(n_, ho_, wo_, 2 * mid)                # conv2's MID channels in rows of 2 MID
                    b.conv_desc(q + 'conv2', pc2, [o], t2, n_, [(h_, w_, ho_, wo_, 0, 0)], c_, 2 * mid, flags=CONV_RELU)
                    xs_ = b.buf(n_, ho_, wo_, 2 * mid)               # the block input at the pixels the stride-2 downsample conv reads
                    b.plan.add(q + 'subsample', lib().llv_subsample2_nhwc, (C.c_void_p(x.data_ptr()), C.c_void_p(xs_.data_ptr()), n_, x.shape[1],
                                                                           x.shape[2], 2 * mid, x.shape[-1], 2 * mid), 0.0, 2.0 * 2 * xs_.numel())
                    o = t2
                else:
                    o = b.conv(q + 'conv2', o, pc2, CONV_RELU)
                if kcat or kcat3:
                    def fold(wk, bnk):
                        w = sd[wk].detach().double().cpu()
                        gamma, beta, mean, var = [t.detach().double().cpu() for t in _bn(sd, bnk)]
                        sc = gamma / torch.sqrt(var + BN_EPS)
                        return w * sc.view(-1, 1, 1, 1), beta - mean * sc
                    w3, b3 = fold(q + 'conv3.weight', q + 'bn3')
                    wd, bd = fold(q + 'downsample.0.weight', q + 'downsample.1')
                    pcc = ops.pack_conv(torch.cat([w3, wd], 1).float(), (b3 + bd).float(), 1, 0, n_src=3 if kcat3 else 2)
                    pcc.true_macs_per_pixel = w3.shape[0] * (w3.shape[1] + wd.shape[1])
                    srcs = [o, xs_, xs_[..., mid:]] if kcat3 else [o, x]
                    x = b.conv(q + 'conv3+downsample', srcs, pcc, CONV_RELU)
                else:
                    x = b.conv(q + 'conv3', o, ops.pack_conv(sd[q + 'conv3.weight'], None, 1, 0, bn=_bn(sd, q + 'bn3'),
                                                             eps=BN_EPS), CONV_RELU | CONV_RES_ADD, res=identity)
            else:
                o = b.conv(q + 'conv1', x, ops.pack_conv(sd[q + 'conv1.weight'], None, stride, dil,
                                                         bn=_bn(sd, q + 'bn1'), eps=BN_EPS, dilation=dil), CONV_RELU)
                x = b.conv(q + 'conv2', o, ops.pack_conv(sd[q + 'conv2.weight'], None, 1, 1, bn=_bn(sd, q + 'bn2'),
                                                         eps=BN_EPS), CONV_RELU | CONV_RES_ADD, res=identity)
        if after_stage is not None:             # InsertResNet: a plugin appended to the stage (insert_resnet.py:93-97)
            x = after_stage(i, nb, x)
        for j in out_indices:                   # an index may repeat (insert_resnet.py:410-415, configs use (2, 3, 3))
            if i == j:
                feats.append(x)
    return feats


def _denoising_block(b: Builder, sd, q, x, identity, stride, dil, style):
    """DarkResNet's DenoisingBlock (mmtracking/mmtrack/models/backbones/dark_resnet.py:226-285): conv1 + BN + ReLU over all frames, then a
    ConvLSTM along the BATCH (= the frames of the clip, in order): per frame conv2 over [x_t | h_{t-1}] (a two-source launch, 2P -> 4P
    channels, no BatchNorm in between) and llv_convlstm_gate (c, h; norm2 + ReLU of h as its second output), then conv3 + BN + identity +
    ReLU over all frames. h_0 = c_0 = 0 at every call (:230-231). With the stride on conv2 (style 'pytorch') and more than one frame the
    reference's torch.cat([x, h_cur]) (:233) meets two resolutions and raises; so does this."""
    L = lib()
    s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)
    planes = sd[q + 'conv1.weight'].shape[0]
    if planes % 32:
        raise LlvError('DenoisingBlock: planes must be a multiple of 32 (a 32-channel chunk of the two-source conv lies in one source)')
    t1 = b.conv(q + 'conv1', x, ops.pack_conv(sd[q + 'conv1.weight'], None, s1, 0, bn=_bn(sd, q + 'bn1'), eps=BN_EPS), CONV_RELU)
    n, h1, w1, _ = t1.shape
    if s2 != 1 and n > 1:
        raise LlvError(f'{q}: DenoisingBlock with stride {s2} on conv2 and {n} frames: the hidden state of frame 0 has the strided size, frame '
                       "1's input has not (dark_resnet.py:233 raises 'Sizes of tensors must match' there); use style='caffe' or stride 1")
    pc2 = ops.pack_conv(sd[q + 'conv2.weight'], None, s2, dil, n_src=2, dilation=dil)
    ho, wo = ops.conv_out_size(h1, w1, pc2)
    h_all, act = b.buf(n, ho, wo, planes), b.buf(n, ho, wo, planes)
    h0 = b.buf(1, h1, w1, planes)                                  # zeros, never written
    gates = b.buf(1, ho, wo, 4 * planes)
    c = b.buf(ho * wo, planes, dtype=torch.float32)
    czero = b.buf(ho * wo, planes, dtype=torch.float32)
    gamma, beta, mean, var = [t.detach().double().cpu() for t in _bn(sd, q + 'bn2')]
    sc = gamma / torch.sqrt(var + BN_EPS)
    scale, shift = sc.float().to(b.dev), (beta - mean * sc).float().to(b.dev)
    b.plan.keep += [scale, shift]
    b.plan.add(q + 'lstm.zero_c', L.llv_copy_d2d, (ops._ptr(c), ops._ptr(czero), C.c_size_t(c.numel() * 4)), 0.0, 8.0 * c.numel())
    for t in range(n):
        b.conv(f'{q}conv2.t{t}', [t1[t:t + 1], h0 if t == 0 else h_all[t - 1:t]], pc2, 0, out=gates)
        b.plan.add(f'{q}lstm.t{t}', L.llv_convlstm_gate, (ops._ptr(gates), ops._ptr(c), ops._ptr(scale), ops._ptr(shift), ops._ptr(h_all[t:t + 1]),
                                                        ops._ptr(act[t:t + 1]), ho * wo, planes, 4 * planes, planes, planes),
                   0.0, 2.0 * ho * wo * planes * 8)
    return b.conv(q + 'conv3', act, ops.pack_conv(sd[q + 'conv3.weight'], None, 1, 0, bn=_bn(sd, q + 'bn3'), eps=BN_EPS),
                  CONV_RELU | CONV_RES_ADD, res=identity)


class Pyramid:
    """All FPN levels in ONE pixel-major buffer (total_cells, C): level l holds batch x h_l x w_l pixels
    starting at pixel offset level_off[l], so a shared-weight head conv is a single multi-segment launch."""

    def __init__(self, buf, level_hw, level_off, batch):
        self.buf, self.level_hw, self.level_off, self.batch = buf, level_hw, level_off, batch

    @property
    def total(self):
        return self.buf.shape[0]

    def level(self, i, buf=None):
        buf = self.buf if buf is None else buf
        h, w = self.level_hw[i]
        o = self.level_off[i]
        return buf[o:o + self.batch * h * w].view(self.batch, h, w, buf.shape[-1])

    def segs(self):
        return [(h, w, h, w, self.level_off[i], self.level_off[i]) for i, (h, w) in enumerate(self.level_hw)]


def build_fpn(b: Builder, sd, feats, start_level=0, end_level=-1, num_outs=None, add_extra_convs=False,
              relu_before_extra_convs=False, p='') -> Pyramid:
    """fpn.py:170-221: lateral 1x1 convs run top-down with the nearest-upsample-add fused into the epilogue."""
    batch = feats[0].shape[0]
    backbone_end = len(feats) if end_level == -1 else end_level
    n_lat = backbone_end - start_level
    num_outs = num_outs if num_outs is not None else n_lat
    lat = [None] * n_lat
    for i in range(n_lat - 1, -1, -1):
        pc = ops.pack_conv(sd[f'{p}lateral_convs.{i}.conv.weight'], sd.get(f'{p}lateral_convs.{i}.conv.bias'))
        if i == n_lat - 1:
            lat[i] = b.conv(f'neck.lateral{i}', feats[i + start_level], pc)
        else:
            lat[i] = b.conv(f'neck.lateral{i}', feats[i + start_level], pc, CONV_RES_ADD | CONV_RES_UPS, res=lat[i + 1])
    level_hw = [(t.shape[1], t.shape[2]) for t in lat]
    extra_pcs = []
    if num_outs > n_lat:
        if not add_extra_convs:
            raise LlvError('FPN max-pool extra levels (add_extra_convs=False) are not on the RetinaNet path')
        for i in range(n_lat, num_outs):
            pc = ops.pack_conv(sd[f'{p}fpn_convs.{i}.conv.weight'], sd.get(f'{p}fpn_convs.{i}.conv.bias'), 2, 1)
            if i == n_lat and add_extra_convs == 'on_input':
                h, w = feats[backbone_end - 1].shape[1], feats[backbone_end - 1].shape[2]
            else:
                h, w = level_hw[-1]
            level_hw.append(ops.conv_out_size(h, w, pc))
            extra_pcs.append(pc)
    c_f = lat[0].shape[-1]
    lvl_off, tot = [], 0
    for h, w in level_hw:
        lvl_off.append(tot)
        tot += batch * h * w
    pyr = b.buf(tot, c_f)
    for i in range(n_lat):
        pc = ops.pack_conv(sd[f'{p}fpn_convs.{i}.conv.weight'], sd.get(f'{p}fpn_convs.{i}.conv.bias'), 1, 1)
        h, w = level_hw[i]
        b.conv_desc(f'neck.fpn{i}', pc, [lat[i]], pyr, batch, [(h, w, h, w, 0, lvl_off[i])], c_f, c_f)
    for k, i in enumerate(range(n_lat, num_outs)):
        pc = extra_pcs[k]
        ho, wo = level_hw[i]
        if i == n_lat:
            if add_extra_convs == 'on_input':
                src = feats[backbone_end - 1]
                (h, w), xo = (src.shape[1], src.shape[2]), 0
            elif add_extra_convs == 'on_lateral':
                src, (h, w), xo = lat[-1], level_hw[n_lat - 1], 0
            else:
                src, (h, w), xo = pyr, level_hw[n_lat - 1], lvl_off[n_lat - 1]
        else:
            if relu_before_extra_convs:
                raise LlvError('relu_before_extra_convs=True is not supported')
            src, (h, w), xo = pyr, level_hw[i - 1], lvl_off[i - 1]
        b.conv_desc(f'neck.fpn{i}', pc, [src], pyr, batch, [(h, w, ho, wo, xo, lvl_off[i])], src.shape[-1], c_f)
    return Pyramid(pyr, level_hw, lvl_off, batch)


def fused_cls_eligible(sd, p, num_classes):
    """The class convolution can carry the first step of get_bboxes in its epilogue (LLV_CONV_CLSMAX, include/llv.h): 3x3 conv with
    whole 16-channel blocks per anchor."""
    w = sd[p + 'retina_cls.weight']
    return (_os.environ.get('LLV_FUSED_CLS', '1') != '0' and w.shape[-1] == 3 and w.shape[1] % 32 == 0 and num_classes % 16 == 0
            and w.shape[0] % num_classes == 0)


def build_retina_head(b: Builder, sd, pyr: Pyramid, stacked_convs=4, logits_f32=True, p='', pingpong=True, fused_cls=None):
    """retina_head.py:93-114. The levels share weights -> one multi-segment launch per layer; the cls and
    reg towers run side by side (layer 0 as one 2x-wide conv on the shared input, layers 1.. as groups=2).
    fused_cls = dict(num_classes=C, score_thr=t) (inference plans whose logits go straight to PostProcess): the class convolution keeps
    every anchor's maximum logit and stores only the 128-channel segments that hold a logit above logit(t) - margin; returns
    (cls_buf, reg_buf, (clsmax, seg_written)) then."""
    batch, segs, tot = pyr.batch, pyr.segs(), pyr.total
    c_f = pyr.buf.shape[-1]
    pccls = ops.pack_conv(sd[p + 'retina_cls.weight'], sd[p + 'retina_cls.bias'], 1, 1)
    pcreg = ops.pack_conv(sd[p + 'retina_reg.weight'], sd[p + 'retina_reg.bias'], 1, 1)
    odt = torch.float32 if logits_f32 else torch.float16
    of = CONV_OUT_F32 if logits_f32 else 0
    cls_buf = b.buf(tot, pccls.cout, dtype=odt)
    reg_buf = b.buf(tot, pcreg.cout, dtype=odt)
    cm, cflag = None, 0
    if fused_cls is not None:
        if not logits_f32:
            raise LlvError('build_retina_head: the fused class maximum works on fp32 logits')
        ncls = int(fused_cls['num_classes'])
        thr = float(fused_cls.get('score_thr', 0.0))
        # a segment whose logits are all <= store_thr is never read as a score above score_thr: logit(score_thr) minus a margin that covers
        # the difference between this comparison on the logit and the NMS's on sigmoid(logit)
        import math
        store_thr = (math.log(thr / (1.0 - thr)) - 1e-2) if 0.0 < thr < 1.0 else -3.0e38
        cm = (b.buf(tot, pccls.cout // ncls, dtype=torch.int32), b.buf(tot, (pccls.cout + 127) // 128, dtype=torch.uint8), ncls, store_thr)
        cflag = CONV_CLSMAX
    if stacked_convs > 0:
        feat_ch = sd[p + 'cls_convs.0.conv.weight'].shape[0]
        ta = b.buf(tot, 2 * feat_ch)
        tb = b.buf(tot, 2 * feat_ch) if stacked_convs > 1 else None
        cur, nxt = ta, tb
        for i in range(stacked_convs):
            pcc = ops.pack_conv(sd[f'{p}cls_convs.{i}.conv.weight'], sd[f'{p}cls_convs.{i}.conv.bias'], 1, 1)
            pcr = ops.pack_conv(sd[f'{p}reg_convs.{i}.conv.weight'], sd[f'{p}reg_convs.{i}.conv.bias'], 1, 1)
            if i == 0:
                b.conv_desc('head.tower0', ops.concat_cout([pcc, pcr]), [pyr.buf], cur, batch, segs, c_f, 2 * feat_ch,
                            flags=CONV_RELU)
            else:
                if not pingpong:
                    nxt = b.buf(tot, 2 * feat_ch)
                b.conv_desc(f'head.tower{i}', ops.stack_groups([pcc, pcr]), [cur], nxt, batch, segs, 2 * feat_ch,
                            2 * feat_ch, flags=CONV_RELU)
                cur, nxt = nxt, cur
        b.conv_desc('head.retina_cls', pccls, [cur[:, :feat_ch]], cls_buf, batch, segs, 2 * feat_ch, pccls.cout, flags=of | cflag, clsmax=cm)
        b.conv_desc('head.retina_reg', pcreg, [cur[:, feat_ch:]], reg_buf, batch, segs, 2 * feat_ch, pcreg.cout, flags=of)
    else:
        b.conv_desc('head.retina_cls', pccls, [pyr.buf], cls_buf, batch, segs, c_f, pccls.cout, flags=of | cflag, clsmax=cm)
        b.conv_desc('head.retina_reg', pcreg, [pyr.buf], reg_buf, batch, segs, c_f, pcreg.cout, flags=of)
    if fused_cls is not None:
        return cls_buf, reg_buf, (cm[0], cm[1])
    return cls_buf, reg_buf


class PostProcess:
    """Per image: llv_retina_decode_topk then llv_multiclass_nms (anchor_head.py:636-726), no host sync."""

    def __init__(self, b: Builder, cls_buf, reg_buf, pyr: Pyramid, num_classes, anchor_generator, test_cfg,
                 target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.), img_shapes=None, scale_factors=None,
                 rescale=False, clip_border=True, clsmax=None):
        tc = dict(test_cfg)
        nms_cfg = tc['nms']
        if nms_cfg.get('type', 'nms') != 'nms':
            raise LlvError('only greedy nms is on this path')
        iou_thr = nms_cfg.get('iou_threshold', nms_cfg.get('iou_thr'))
        A = anchor_generator.num_base_anchors[0]
        assert cls_buf.shape[-1] >= A * num_classes and reg_buf.shape[-1] >= A * 4
        self.num_classes, self.batch, self.clip_border = num_classes, pyr.batch, clip_border
        nb_ = pyr.batch
        shapes = list(img_shapes) if (img_shapes is not None and clip_border) else None
        sfs = list(scale_factors) if scale_factors is not None else [(1., 1., 1., 1.)] * nb_
        # ONE batched decode and ONE batched NMS for all images of the step (image = blockIdx.y)
        self.dp = ops.make_decode_plan(cls_buf, reg_buf, pyr.level_hw, anchor_generator.strides, pyr.level_off, pyr.level_off,
                                       cls_buf.shape[-1], reg_buf.shape[-1], num_classes, A, anchor_generator.base_anchors,
                                       tc.get('nms_pre', -1), target_means, target_stds, 16 / 1000, shapes, sfs, rescale, batch=nb_,
                                       clsmax=clsmax)
        self.nb = ops.NmsBuffers(self.dp.n_cand, num_classes, tc['max_per_img'], b.dev, batch=nb_)
        b.plan.keep += [self.dp, self.nb]
        b.plan.add('decode', lib().llv_retina_decode_topk, (C.byref(self.dp.desc),), 0.0,
                   float(cls_buf.element_size() * cls_buf.numel()))
        b.plan.add('nms', lib().llv_multiclass_nms_batched,
                   (C.c_void_p(self.dp.boxes.data_ptr()), C.c_void_p(self.dp.scores.data_ptr()), nb_, self.dp.n_cand, num_classes,
                    num_classes, C.c_float(tc['score_thr']), C.c_float(iou_thr), tc['max_per_img'],
                    C.c_void_p(self.nb.dets.data_ptr()), C.c_void_p(self.nb.labels.data_ptr()), C.c_void_p(self.nb.count.data_ptr()),
                    C.c_void_p(self.nb.keep_flat.data_ptr()), C.c_void_p(self.nb.ws.data_ptr()), C.c_size_t(self.nb.ws_bytes)), 0.0, 0.0)

    def set_metas(self, img_metas, rescale):
        """Per-call image geometry (img_shape clip, scale_factor): a host-side edit of the descriptor."""
        ops.set_decode_image_params(self.dp.desc, [m['img_shape'] for m in img_metas[:self.batch]],
                                    [m.get('scale_factor', 1.0) for m in img_metas[:self.batch]], rescale, self.clip_border)

    def dets(self, n=0):
        return self.nb.dets if self.batch == 1 else self.nb.dets[n]

    def results(self):
        """D2H of the detections: [(dets (k,5), labels (k,)), ...] per image."""
        counts = self.nb.count.cpu().tolist()
        ops.raise_on_guard_trips('PostPlan.results')
        d = self.nb.dets.cpu().reshape(self.batch, -1, 5)
        l = self.nb.labels.cpu().reshape(self.batch, -1)
        return [(d[n, :k], l[n, :k]) for n, k in enumerate(counts)]


DEFAULT_TEST_CFG = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100)


class RetinaNetPlan:
    """Whole-detector inference plan for a fixed (batch, H, W). Input: self.x_in (N,H,W,8) fp16 NHWC
    (3 real channels, normalised image). After run()/replay(): self.post.results()."""

    def __init__(self, sd: Dict[str, torch.Tensor], num_classes: int, batch: int, height: int, width: int, device,
                 depth=50, strides=(1, 2, 2, 2), style='pytorch', out_indices=(0, 1, 2, 3), start_level=1, num_outs=5,
                 add_extra_convs='on_input', relu_before_extra_convs=False, stacked_convs=4, anchor_generator=None,
                 test_cfg=None, target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.), img_shapes=None,
                 scale_factors=None, rescale=False, logits_f32=True, x_in=None, prefix=''):
        from .core.anchor import AnchorGenerator
        b = Builder(device)
        self.builder, self.plan = b, b.plan
        self.num_classes, self.batch = num_classes, batch
        sd = _cpu_sd(sd)
        self.x_in = x_in if x_in is not None else b.buf(batch, height, width, 8)
        self.backbone_feats = build_resnet(b, sd, self.x_in, depth, strides, style, out_indices, p=prefix + 'backbone.',
                                           fuse_blocks=_os.environ.get("LLV_FUSED_BOTTLENECK", "1") != '0')
        self.pyramid = build_fpn(b, sd, self.backbone_feats, start_level, -1, num_outs, add_extra_convs,
                                 relu_before_extra_convs, p=prefix + 'neck.')
        tcfg = dict(test_cfg or DEFAULT_TEST_CFG)
        # the class convolution's epilogue does the first step of get_bboxes (per-anchor class maximum) and stores only the logit segments
        # that can pass score_thr: the dense fp32 logits (941 MB per 16-frame step) are neither written nor read back by the decode
        self.clsmax = None
        if logits_f32 and fused_cls_eligible(sd, prefix + 'bbox_head.', num_classes):
            self.cls_buf, self.reg_buf, self.clsmax = build_retina_head(b, sd, self.pyramid, stacked_convs, logits_f32, p=prefix + 'bbox_head.',
                                                                        fused_cls=dict(num_classes=num_classes, score_thr=tcfg.get('score_thr', 0.0)))
        else:
            self.cls_buf, self.reg_buf = build_retina_head(b, sd, self.pyramid, stacked_convs, logits_f32, p=prefix + 'bbox_head.')
        if isinstance(anchor_generator, AnchorGenerator):
            ag = anchor_generator
        else:
            cfg = dict(anchor_generator or dict(octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0],
                                                strides=[8, 16, 32, 64, 128]))
            cfg.pop('type', None)
            ag = AnchorGenerator(**cfg)
        self.anchor_generator = ag
        if img_shapes is None:
            img_shapes = [(height, width, 3)] * batch
        self.post = PostProcess(b, self.cls_buf, self.reg_buf, self.pyramid, num_classes, ag, tcfg,
                                target_means, target_stds, img_shapes, scale_factors, rescale, clsmax=self.clsmax)

    def results(self):
        return self.post.results()


# ------------------------------------------------------------------------------------------------
# FastDVDnet
# ------------------------------------------------------------------------------------------------
def _denblock(b: Builder, sd, p, srcs, batch, tag, out):
    """One DenBlock over `batch` images whose three input frames are srcs[0..2] (each a view (batch,H,W,8):
    image n of source s is frame n+s of a clip, so consecutive windows share memory). FastDVDnet.py:102-116."""
    f = CONV_RELU
    # inc: Conv2d(9, 90, 3, groups=3) over three separate NHWC8 frames: a grouped launch whose group g reads
    # source tensor g (3 real + 5 zero channels) and writes channels [32g, 32g+30) (+2 zero) of a 96-wide buffer
    wg = sd[p + 'inc.convblock.0.weight']                    # (90, 3, 3, 3)
    ng = 3
    og = wg.shape[0] // ng
    og_pad = ops.round_up(og, 4)
    bnp = _bn(sd, p + 'inc.convblock.1')
    pcs = []
    for g in range(ng):
        sl = slice(og * g, og * (g + 1))
        pcs.append(ops.pack_conv(wg[sl], None, 1, 1, in_map=[0, 1, 2, -1, -1, -1, -1, -1], cout_store=og_pad,
                                 bn=tuple(t[sl] for t in bnp), eps=BN_EPS))
    pc = ops.stack_groups(pcs)
    pc.n_src, pc.src_cvec = ng, 1
    if HALO:
        ops.add_halo_packing(pc)
    n, hh, ww, _ = srcs[0].shape
    w2 = sd[p + 'inc.convblock.3.weight']
    in_map = []
    for g in range(ng):
        in_map += [og * g + j if j < og else -1 for j in range(og_pad)]
    pc2 = ops.pack_conv(w2, None, 1, 1, in_map=in_map, n_src=ng, bn=_bn(sd, p + 'inc.convblock.4'), eps=BN_EPS)
    if HALO:
        ops.add_halo_packing(pc2)
    if HALO and FUSE_INPUTCV and og_pad == 32 and pc2.cout == 32 and pc.w_halo is not None and pc2.w_halo is not None:
        # the whole InputCvBlock as one launch: the 90-channel intermediate never leaves LDS (llv_inputcvblock_fwd)
        from ._lib import InputCvDesc
        x0 = b.buf(n, hh, ww, pc2.cout)
        pc.to(b.dev)
        pc2.to(b.dev)
        d = InputCvDesc()
        cd = ops.make_conv_desc(pc2, list(srcs), x0, n, [(hh, ww, hh, ww, 0, 0)], srcs[0].shape[-1], pc2.cout, flags=f)
        cd._ext.w_pre, cd._ext.bias_pre = ops.tensor_extent(pc.w_halo), ops.tensor_extent(pc.bias)
        d.conv, d._keep, d._output_block = cd, cd, 0      # the copied descriptor points at cd's extents struct
        d.w_pre, d.bias_pre = pc.w_halo.data_ptr(), pc.bias.data_ptr()
        b.window_descs.append((tag, d, 'x'))
        m = n * hh * ww
        b.plan.keep += [pc, pc2, d, x0] + list(srcs)
        b.plan.add(f'{tag}.inputcv', lib().llv_inputcvblock_fwd, (C.byref(d),), 2.0 * m * (pc.true_macs_per_pixel + pc2.true_macs_per_pixel),
                   2.0 * m * (24 + pc2.cout), kernel='convpair2_kernel<8,32>')
    else:
        # each group writes its own (batch, H, W, 32) plane (full cache lines); inc1 reads the three planes as three sources
        planes = b.buf(ng, n, hh, ww, og_pad)
        b.conv_desc(f'{tag}.inc0', pc, list(srcs), planes, n, [(hh, ww, hh, ww, 0, 0)], srcs[0].shape[-1], og_pad, flags=f,
                    y_group_stride=n * hh * ww * og_pad)
        x0 = b.conv(f'{tag}.inc1', [planes[g] for g in range(ng)], pc2, f)

    def cbr(name, x, conv, bn, stride=1, flags=f, res=None):
        pc = ops.pack_conv(sd[conv + '.weight'], None, stride, 1, bn=_bn(sd, bn), eps=BN_EPS)
        if HALO and stride == 1 and pc.cout <= HALO_MAX_COUT and pc.cin % 32:
            ops.add_halo_packing(pc)        # Cin a multiple of 32 goes to the tap-streamed kernel instead
        return b.conv(name, x, pc, flags, res=res)

    def down(x, q, t):                                       # DownBlock, FastDVDnet.py:39-49
        x = cbr(f'{tag}.{t}.0', x, q + 'convblock.0', q + 'convblock.1', 2)
        x = cbr(f'{tag}.{t}.1', x, q + 'convblock.3.convblock.0', q + 'convblock.3.convblock.1')
        return cbr(f'{tag}.{t}.2', x, q + 'convblock.3.convblock.3', q + 'convblock.3.convblock.4')

    def up(x, q, t, skip):                                   # UpBlock, FastDVDnet.py:52-61; the skip add of
        x = cbr(f'{tag}.{t}.0', x, q + 'convblock.0.convblock.0', q + 'convblock.0.convblock.1')   # :111-113 is fused
        x = cbr(f'{tag}.{t}.1', x, q + 'convblock.0.convblock.3', q + 'convblock.0.convblock.4')
        pc = ops.pack_conv(sd[q + 'convblock.1.weight'], None, 1, 1)
        if pc.cin % 32 == 0 and pc.cout % 32 == 0 and PS_EMAJOR:
            # sub-pixel-major output channels: the 64-channel tap kernel then writes whole output pixels
            return b.conv(f'{tag}.{t}.2', x, ops.pixshuf_emajor(pc), CONV_PIXSHUF | CONV_RES_ADD | ops.CONV_PS_EMAJOR, res=skip)
        if HALO and pc.cout <= HALO_MAX_COUT and pc.cin % 32:
            ops.add_halo_packing(pc)
        return b.conv(f'{tag}.{t}.2', x, pc, CONV_PIXSHUF | CONV_RES_ADD, res=skip)

    x1 = down(x0, p + 'downc0.', 'down0')
    x2 = down(x1, p + 'downc1.', 'down1')
    x1s = up(x2, p + 'upc2.', 'up2', x1)                     # = x1 + upc2(x2)
    x0s = up(x1s, p + 'upc1.', 'up1', x0)                    # = x0 + upc1(x1 + x2)
    # OutputCvBlock (FastDVDnet.py:64-74) + the DenBlock residual in1 - x (:115)
    pc0 = ops.pack_conv(sd[p + 'outc.convblock.0.weight'], None, 1, 1, bn=_bn(sd, p + 'outc.convblock.1'), eps=BN_EPS)
    pc1 = ops.pack_conv(sd[p + 'outc.convblock.3.weight'], None, 1, 1, cout_store=8)
    if HALO:
        ops.add_halo_packing(pc0)
        ops.add_halo_packing(pc1)
    res = srcs[1]
    if HALO and FUSE_OUTPUTCV and pc0.cin == 32 and pc0.cout == 32 and pc1.halo_bn == 16 and pc0.halo_bn == 32:
        from ._lib import InputCvDesc
        n, hh, ww, _ = x0s.shape
        pc0.to(b.dev)
        pc1.to(b.dev)
        d = InputCvDesc()
        cd = ops.make_conv_desc(pc1, [x0s], out, n, [(hh, ww, hh, ww, 0, 0)], x0s.shape[-1], out.shape[-1], res, res.shape[-1],
                                CONV_RES_RSUB)
        cd._ext.w_pre, cd._ext.bias_pre = ops.tensor_extent(pc0.w_halo), ops.tensor_extent(pc0.bias)
        d.conv, d._keep, d._output_block = cd, cd, 1
        d.w_pre, d.bias_pre = pc0.w_halo.data_ptr(), pc0.bias.data_ptr()
        b.window_descs.append((tag, d, 'res'))
        m = n * hh * ww
        b.plan.keep += [pc0, pc1, d, x0s, out, res]
        b.plan.add(f'{tag}.outputcv', lib().llv_outputcvblock_fwd, (C.byref(d),), 2.0 * m * (pc0.true_macs_per_pixel + pc1.true_macs_per_pixel),
                   2.0 * m * (32 + 8 + 8), kernel='convpair2_kernel<32,16>')
    else:
        x = b.conv(f'{tag}.out0', x0s, pc0, f)
        b.conv(f'{tag}.out1', x, pc1, CONV_RES_RSUB, res=res, out=out)


class FastDVDnetPlan:
    """Denoise `batch` consecutive frames of a clip in one pass.

    self.frames is (batch + 4, H, W, 8): frames t0-2 .. t0+batch+1 (caller replicates clip ends).
    temp1 runs on the batch+2 distinct triples, temp2 on `batch` triples of temp1 outputs -- every
    temp1 result is computed once per pass (for batch == 1 this is exactly the reference's 3 + 1
    DenBlocks per frame). Output: self.out (batch, H, W, 8), channels 0..2 = denoised frame.
    """

    def __init__(self, sd, batch, height, width, device, frames=None, out=None, prefix=''):
        if height % 4 or width % 4:
            raise LlvError('FastDVDnet needs H, W multiples of 4 (two stride-2 stages)')
        b = Builder(device)
        self.builder, self.plan, self.batch = b, b.plan, batch
        sd = {k: v.detach().float().cpu() for k, v in sd.items()}
        nf = batch + 4
        self.frames = frames if frames is not None else b.buf(nf, height, width, 8)
        n1 = batch + 2
        self.mid = b.buf(n1, height, width, 8)
        self.out = out if out is not None else b.buf(batch, height, width, 8)
        p1, p2 = prefix + 'temp1.', prefix + 'temp2.'
        srcs = [self.frames[s:s + n1] for s in range(3)]
        # the first stage runs on batch + 2 windows here and on `batch` new ones in the streaming step (enable_stream), whose results
        # are carried from step to step next to these: both select their kernels for `batch` images, so a first-stage result has the
        # same bits whichever step computed it and a frame's output does not depend on the frames processed before it
        b.select_batch = batch
        _denblock(b, sd, p1, srcs, n1, 'temp1', self.mid)
        b.select_batch = None
        srcs2 = [self.mid[s:s + batch] for s in range(3)]
        _denblock(b, sd, p2, srcs2, batch, 'temp2', self.out)
        self._sd, self._p1, self.plan_stream = sd, p1, None
        self._window_descs = self._rel_window_descs(b)
        self._stream_window_descs = []
        self._clip = None

    def _rel_window_descs(self, b):
        """(descriptor, 'x' | 'res', byte offsets of its frame pointers from the start of self.frames) of the launches that read frames"""
        out, base = [], self.frames.data_ptr()
        for tag, d, kind in b.window_descs:
            if tag == 'temp1':
                out.append((d, kind, [d.conv.x[i] - base for i in range(3)] if kind == 'x' else [d.conv.res - base]))
        return out

    def bind_clip(self, clip, offset_word):
        """Zero-copy windows: make the first-stage DenBlocks read the window straight out of `clip` ((T+4, H, W, 8) fp16, resident
        in HBM) instead of out of `self.frames`. The launches that touch the frames -- the fused InputCvBlock (its three sources)
        and the fused OutputCvBlock (the residual in1 of `in1 - x`) of temp1 -- are re-pointed at the clip's first window and get
        `offset_word` (a device int64 holding t0 * H*W*8) as their llv_conv_desc.offset_dev, so one captured graph serves every
        window: the caller stores the offset and replays. Needs the fused CvBlock launches (the default)."""
        nf = self.batch + 4
        if clip.dtype != torch.float16 or clip.dim() != 4 or clip.shape[1:] != self.frames.shape[1:] or clip.shape[0] < nf or not clip.is_contiguous():
            raise LlvError(f'bind_clip: clip of shape (>= {nf}, {tuple(self.frames.shape[1:])}) fp16 contiguous expected, got {tuple(clip.shape)}')
        if len(self._window_descs) != 2:
            raise LlvError('bind_clip needs the fused InputCvBlock / OutputCvBlock launches (LLV_FUSE_INPUTCV / LLV_FUSE_OUTPUTCV / LLV_HALO)')
        frame_elems = self.frames[0].numel()
        off_max = (clip.shape[0] - nf) * frame_elems
        total = ops.tensor_extent(clip)
        for d, kind, rels in self._window_descs + self._stream_window_descs:
            cd, ext = d.conv, d._keep._ext
            if kind == 'x':
                for i in range(3):
                    cd.x[i] = clip.data_ptr() + rels[i]
                    ext.x[i] = total - rels[i]
                cd.offset_x = 1
            else:
                cd.res = clip.data_ptr() + rels[0]
                ext.res = total - rels[0]
                cd.offset_res = 1
            cd.offset_dev, cd.offset_max = offset_word.data_ptr(), off_max
        self._clip, self._offset_word, self.frame_elems = clip, offset_word, frame_elems

    def enable_stream(self):
        """Build `plan_stream`, the step for a window that directly FOLLOWS the previous step's window in the same video
        (frames t0 .. t0+batch-1 after t0-batch .. t0-1): the first-stage outputs centred on frames t0-1 and t0 were the last two
        the previous step computed, so they are moved to the front of `mid` and only the `batch` new first-stage DenBlocks run
        (2 DenBlocks per frame instead of 2 + 2/batch; batch = 1: 2 instead of the reference's 4). Same kernels on the same
        pixels: the output is bit-identical to `plan`'s. The first step of a video must use `plan`."""
        if self.plan_stream is not None:
            return self.plan_stream
        batch, dev = self.batch, self.plan.device
        bs = Builder(dev)
        srcs = [self.frames[s + 2:s + 2 + batch] for s in range(3)]
        _denblock(bs, self._sd, self._p1, srcs, batch, 'temp1', self.mid[2:2 + batch])
        ps = Plan(dev)
        img_bytes = self.mid[0].numel() * 2
        if batch >= 2:
            ps.add('temp1.carry', lib().llv_copy_d2d, (C.c_void_p(self.mid.data_ptr()), C.c_void_p(self.mid[batch:].data_ptr()),
                                                      C.c_size_t(2 * img_bytes)), 0.0, 4.0 * img_bytes)
        else:   # batch 1: mid = [c-1, c, c+1]; the two carried images overlap their destination, so they move one at a time
            ps.add('temp1.carry0', lib().llv_copy_d2d, (C.c_void_p(self.mid[0].data_ptr()), C.c_void_p(self.mid[1].data_ptr()),
                                                       C.c_size_t(img_bytes)), 0.0, 2.0 * img_bytes)
            ps.add('temp1.carry1', lib().llv_copy_d2d, (C.c_void_p(self.mid[1].data_ptr()), C.c_void_p(self.mid[2].data_ptr()),
                                                       C.c_size_t(img_bytes)), 0.0, 2.0 * img_bytes)
        ps.extend(bs.plan)
        for st in self.plan.steps:
            if st[0].startswith('temp2.'):
                ps.steps.append(st)
                if st[0] in self.plan.kernel_of:
                    ps.kernel_of[st[0]] = self.plan.kernel_of[st[0]]
        ps.keep += self.plan.keep
        self.plan_stream = ps
        self._stream_window_descs = self._rel_window_descs(bs)
        if self._clip is not None:       # already bound: the new launches read the clip too
            self.bind_clip(self._clip, self._offset_word)
        return ps


# ------------------------------------------------------------------------------------------------
# enhance + detect on a clip
# ------------------------------------------------------------------------------------------------
class EnhanceDetectPlan:
    """FastDVDnet -> RetinaNet over `batch` consecutive frames of a clip resident in HBM.

    The denoiser writes straight into the detector's NHWC8 input buffer (no layout change, no copy).
    clip: (T + 4, H, W, 8) fp16, frames replicated by two at each end (the window of frame t is t-2..t+2,
    clamped at the clip ends as the reference's test-time reference sampler does).
    step(t0): frames t0 .. t0+batch-1.
    """

    def __init__(self, cleaner_sd, detector_sd, num_classes, batch, height, width, device, detector_kwargs=None,
                 use_graph=True):
        self.batch, self.h, self.w, self.dev = batch, height, width, device
        self.den = FastDVDnetPlan(cleaner_sd, batch, height, width, device)
        self.det = RetinaNetPlan(detector_sd, num_classes, batch, height, width, device, x_in=self.den.out,
                                 **(detector_kwargs or {}))
        self.plan = Plan(device)
        self.plan.extend(self.den.plan)
        self.plan.extend(self.det.plan)
        self.use_graph = use_graph
        k = self.det.post.nb.max_num
        # results of the step, device side, gathered into one buffer so the D2H is a single copy
        self.res_dev = torch.zeros((batch, k, 6), device=device, dtype=torch.float32)
        self.cnt_dev = torch.zeros((batch,), device=device, dtype=torch.int32)
        self.res_host = torch.zeros((batch, k, 6), dtype=torch.float32).pin_memory()
        self.cnt_host = torch.zeros((batch,), dtype=torch.int32).pin_memory()
        self.guard_host = torch.zeros((1,), dtype=torch.int32).pin_memory()     # llv_guard_trips_to: the step's last kernel node
        self._graph = None
        self._zero_copy = False

    def _enhance(self, window, carry=False):
        """FastDVDnet half of a step: the clip window -> denoised frames in the detector's input buffer. carry=True: the window
        directly follows the previous step's (FastDVDnetPlan.enable_stream). window=None: the launches read the bound clip
        at the offset the caller stored (zero-copy, prepare(clip, zero_copy=True))."""
        if window is not None:
            self.den.frames.copy_(window)
        (self.den.plan_stream if carry else self.den.plan).run()

    def _detect(self):
        """Detector half of a step: backbone .. NMS on the denoised frames, results into pinned host memory."""
        self.det.plan.run()
        nb = self.det.post.nb
        if _os.environ.get('LLV_GRAPH_MEMNODES', '0') != '0':      # diagnosis only: round 1's copies (memcpy nodes in the captured step)
            self.res_dev[:, :, :5].copy_(nb.dets.reshape(self.batch, -1, 5))
            self.res_dev[:, :, 5].copy_(nb.labels.reshape(self.batch, -1))
            self.cnt_dev.copy_(nb.count)
            self.res_host.copy_(self.res_dev, non_blocking=True)
            self.cnt_host.copy_(self.cnt_dev, non_blocking=True)
            return
        # one kernel writes the <= max_per_img rows of every image straight into the pinned host buffers (device-mapped): the
        # captured step then holds kernel nodes only -- no memcpy / memset node
        rc = lib().llv_pack_results(C.c_void_p(nb.dets.data_ptr()), C.c_void_p(nb.labels.data_ptr()), C.c_void_p(nb.count.data_ptr()),
                                    self.batch, nb.max_num, C.c_void_p(self.res_host.data_ptr()), C.c_void_p(self.cnt_host.data_ptr()),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise LlvError(f'llv_pack_results: {lib().llv_strerror(rc).decode()} ({rc})')
        check(lib().llv_guard_trips_to(C.c_void_p(self.guard_host.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
              'llv_guard_trips_to')

    def _body(self, window, carry=False):
        self._enhance(window, carry)
        self._detect()

    def prepare(self, clip, zero_copy=False):
        """Warm up once (one-time kernel attribute calls) and capture the step into a HIP graph.
        zero_copy=True: `clip` stays resident and every later step(clip, t0) must pass the same tensor; the first-stage launches
        read frames t0 .. t0+batch+3 straight out of it through an offset word (FastDVDnetPlan.bind_clip), so a step moves no
        window at all (round 1 copied it twice: clip -> staging buffer eagerly, staging buffer -> frames inside the graph)."""
        self._zero_copy = bool(zero_copy)
        if self._zero_copy:
            self._off = torch.zeros((1,), device=self.dev, dtype=torch.int64)
            self.den.bind_clip(clip, self._off)
            self._window = None
        else:
            self._window = torch.empty_like(self.den.frames)
            self._window.copy_(clip[:self.batch + 4])
        self._body(self._window)
        torch.cuda.synchronize()
        if self.use_graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._body(self._window)
            self._graph = g

    def prepare_stream(self):
        """Add the streaming step (consecutive windows of one video carry their two shared first-stage results): call after
        prepare(); then step(clip, t0, carry=True) for every window that directly follows the previous one."""
        self.den.enable_stream()
        self._body(self._window, carry=True)
        torch.cuda.synchronize()
        self._graph_stream = None
        if self.use_graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._body(self._window, carry=True)
            self._graph_stream = g

    def step(self, clip, t0, carry=False):
        """Enqueue one step (no host sync). Results land in res_host/cnt_host after the stream drains."""
        if self._zero_copy:
            if clip.data_ptr() != self.den._clip.data_ptr() or t0 < 0 or t0 + self.batch + 4 > clip.shape[0]:
                raise LlvError('zero-copy plan: step() takes the clip given to prepare() and a window inside it')
            self._off.fill_(t0 * self.den.frame_elems)
        else:
            self._window.copy_(clip[t0:t0 + self.batch + 4])
        if carry:
            if self._graph_stream is not None:
                self._graph_stream.replay()
            else:
                self._body(self._window, carry=True)
        elif self._graph is not None:
            self._graph.replay()
        else:
            self._body(self._window)

    def results(self):
        torch.cuda.current_stream().synchronize()
        if int(self.guard_host[0]):              # written by the step's last kernel (no extra device round trip per hand-over)
            ops.raise_on_guard_trips('EnhanceDetectPlan.results')
        out = []
        for n in range(self.batch):
            k = int(self.cnt_host[n])
            r = self.res_host[n, :k]
            out.append((r[:, :5].clone(), r[:, 5].long()))
        return out


class PipelinedEnhanceDetectPlan:
    """EnhanceDetectPlan as a two-stage software pipeline over consecutive steps of a clip.

    A step's two halves depend on each other only through the denoised frames, so step i launches
    FastDVDnet on window i and, on a second HIP stream, the detector (backbone .. NMS .. D2H) on the frames
    window i-1 produced. The detector's small and latency-bound launches (layer4, P6/P7, top-k select, NMS) then
    fill the chip together with FastDVDnet's large ones instead of each draining it. Two complete plans with their
    own buffers alternate (step i enhances into plan i%2, detects from plan (i-1)%2): no copy, no false dependence.
    Both halves of a step are captured into one HIP graph with two branches (fork / join through stream waits).

    Every frame still takes exactly the launches of EnhanceDetectPlan; results of window i are available after
    step i+1 (`results()` returns those of the window detected by the last step)."""

    def __init__(self, cleaner_sd, detector_sd, num_classes, batch, height, width, device, detector_kwargs=None,
                 use_graph=True):
        self.batch, self.dev, self.use_graph = batch, device, use_graph
        self.p = [EnhanceDetectPlan(cleaner_sd, detector_sd, num_classes, batch, height, width, device,
                                    detector_kwargs=detector_kwargs, use_graph=False) for _ in range(2)]
        self.den, self.det, self.plan = self.p[0].den, self.p[0].det, self.p[0].plan    # launch tables (for timing / flops)
        self.side = torch.cuda.Stream(device)
        self._graphs = [None, None]
        self._i = 0

    def _body(self, k):
        """enhance(window k) -> plan k on the current stream  ||  detect(plan 1-k) on the side stream."""
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            self.p[1 - k]._detect()
        self.p[k]._enhance(self._windows[k])
        main.wait_stream(self.side)

    def prepare(self, clip):
        b = self.batch
        self._windows = [torch.empty_like(self.p[0].den.frames) for _ in range(2)]
        for k in range(2):
            self._windows[k].copy_(clip[:b + 4])
        for k in range(2):                     # warm-up outside capture (one-time kernel attribute calls), fills both plans
            self._body(k)
        torch.cuda.synchronize()
        if self.use_graph:
            for k in range(2):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._body(k)
                self._graphs[k] = g
        self._i = 0

    def step(self, clip, t0):
        """Enqueue: enhance window t0 and detect the previous step's window (no host sync)."""
        k = self._i & 1
        self._windows[k].copy_(clip[t0:t0 + self.batch + 4])
        if self._graphs[k] is not None:
            self._graphs[k].replay()
        else:
            self._body(k)
        self._i += 1

    def flush(self):
        """Detect the window the last step enhanced (drains the pipeline); its results are then current."""
        k = self._i & 1
        self.p[1 - k]._detect()
        self._i += 1

    def results(self):
        """Detections of the window most recently detected."""
        return self.p[self._i & 1].results()


class PipelinedStreamingPlan:
    """The per-frame protocol (the reference's benchmark.py:83-109: one frame per step) as a two-stage software pipeline over the
    CONSECUTIVE frames of a video:

        step i:   FastDVDnet streaming step of window i (first-stage results shared with window i-1 are carried: `batch` new
                  first-stage DenBlocks + `batch` second-stage ones)  ->  den.out  ->  copied into detector plan i % 2's input
              ||  detector (backbone .. NMS .. results to pinned memory) of window i-1, on a second HIP stream

    One FastDVDnetPlan (the carried first-stage results live in its `mid` buffer, so it cannot alternate between two copies the
    way PipelinedEnhanceDetectPlan does), two detector plans. At one frame per step both halves leave most of the chip idle (a
    184x320 map is 230 workgroups for 256 CUs; layer3/4 and the pyramid tops far fewer), so running them side by side nearly hides
    the shorter one. Every frame still takes exactly the launches of the sequential streaming step: results are bit-identical, they
    just become available one step later (`results()` after step i+1, or after `flush()`). Windows are read in place from the
    resident clip (bind_clip); both halves of a step are one captured graph with two branches (kernel nodes only)."""

    def __init__(self, cleaner_sd, detector_sd, num_classes, batch, height, width, device, detector_kwargs=None, use_graph=True, stages=2):
        """stages = 3 splits the detector between two more streams: backbone + neck of window i-1 beside head + decode + NMS of window
        i-2 (the pyramid of detector plan k is written in one step and read in the next; the two plans alternate, so the halves never
        touch the same buffers inside a step). Results then leave two steps after their frame entered."""
        assert stages in (2, 3)
        self.batch, self.dev, self.use_graph, self.stages = batch, device, use_graph, stages
        self.den = FastDVDnetPlan(cleaner_sd, batch, height, width, device)
        self.dets = [RetinaNetPlan(detector_sd, num_classes, batch, height, width, device, **(detector_kwargs or {})) for _ in range(2)]
        self.det = self.dets[0]
        names = [st[0] for st in self.det.plan.steps]
        self._head0 = next(i for i, nm in enumerate(names) if nm.startswith('head'))      # first launch that reads the pyramid
        k = self.det.post.nb.max_num
        self.res_host = [torch.zeros((batch, k, 6), dtype=torch.float32).pin_memory() for _ in range(2)]
        self.cnt_host = [torch.zeros((batch,), dtype=torch.int32).pin_memory() for _ in range(2)]
        self.guard_host = [torch.zeros((1,), dtype=torch.int32).pin_memory() for _ in range(2)]
        self.side = torch.cuda.Stream(device)
        self.side2 = torch.cuda.Stream(device) if stages == 3 else None
        self._graphs = [None, None]
        self._i = 0
        self._last = 0                       # parity whose res_host holds the most recent detections
        self._off = torch.zeros((1,), device=device, dtype=torch.int64)
        self._clip = None

    def _enhance(self, k, carry):
        (self.den.plan_stream if carry else self.den.plan).run()
        rc = lib().llv_copy_d2d(C.c_void_p(self.dets[k].x_in.data_ptr()), C.c_void_p(self.den.out.data_ptr()),
                                C.c_size_t(self.den.out.numel() * 2), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise LlvError(f'llv_copy_d2d: {lib().llv_strerror(rc).decode()} ({rc})')

    def _run_range(self, k, lo, hi):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for name, fn, args, _, _ in self.dets[k].plan.steps[lo:hi]:
            rc = fn(*args, stream)
            if rc != 0:
                raise LlvError(f'{name}: {lib().llv_strerror(rc).decode()} ({rc})')

    def _pack(self, k):
        nb = self.dets[k].post.nb
        rc = lib().llv_pack_results(C.c_void_p(nb.dets.data_ptr()), C.c_void_p(nb.labels.data_ptr()), C.c_void_p(nb.count.data_ptr()),
                                    self.batch, nb.max_num, C.c_void_p(self.res_host[k].data_ptr()), C.c_void_p(self.cnt_host[k].data_ptr()),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise LlvError(f'llv_pack_results: {lib().llv_strerror(rc).decode()} ({rc})')
        check(lib().llv_guard_trips_to(C.c_void_p(self.guard_host[k].data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
              'llv_guard_trips_to')

    def _detect(self, k):
        self._run_range(k, 0, len(self.dets[k].plan.steps))
        self._pack(k)

    def _body(self, k, carry, enhance=True, front=True):
        """Step with parity k. Two stages: detector of window i-1 (plan 1-k) beside FastDVDnet of window i. Three stages: backbone +
        neck of window i-1 (plan 1-k), head .. results of window i-2 (plan k), FastDVDnet of window i (into plan k's input, which
        the head half does not read)."""
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)
        if self.stages == 2:
            with torch.cuda.stream(self.side):
                self._detect(1 - k)
        else:
            self.side2.wait_stream(main)
            if front:
                with torch.cuda.stream(self.side):
                    self._run_range(1 - k, 0, self._head0)
            with torch.cuda.stream(self.side2):
                self._run_range(k, self._head0, len(self.dets[k].plan.steps))
                self._pack(k)
            main.wait_stream(self.side2)
        if enhance:
            self._enhance(k, carry)
        main.wait_stream(self.side)

    def prepare(self, clip):
        self.den.bind_clip(clip, self._off)
        self.den.enable_stream()
        self._clip = clip
        self._off.fill_(0)
        for k in (0, 1):                       # warm-up outside capture: full step, then the streaming step, on both parities
            self._body(k, False)
            self._body(k, True)
        torch.cuda.synchronize()
        if self.use_graph:
            for k in (0, 1):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._body(k, True)
                self._graphs[k] = g
        self._i = 0

    def step(self, clip, t0, carry):
        """Enqueue: enhance window t0 (carry=True: it directly follows the previous step's window) and run the detector stage(s) of
        the previous window(s). No host sync."""
        if clip.data_ptr() != self._clip.data_ptr() or t0 < 0 or t0 + self.batch + 4 > clip.shape[0]:
            raise LlvError('PipelinedStreamingPlan.step takes the clip given to prepare() and a window inside it')
        k = self._i & 1
        self._off.fill_(t0 * self.den.frame_elems)
        if carry and self._graphs[k] is not None:
            self._graphs[k].replay()
        else:
            self._body(k, carry)
        self._last = 1 - k if self.stages == 2 else k
        self._i += 1

    def flush(self):
        """Advance the detector stage(s) once without a new window: after `stages - 1` calls the last window's detections are out."""
        k = self._i & 1
        if self.stages == 2:
            self._detect(1 - k)
            self._last = 1 - k
        else:
            self._body(k, False, enhance=False)
            self._last = k
        self._i += 1

    @property
    def lag(self):
        """Steps between a window entering and its detections leaving."""
        return self.stages - 1

    def results(self):
        """Detections of the window most recently detected: [(dets (k, 5), labels (k,))] per image."""
        torch.cuda.synchronize()
        k = self._last
        if int(self.guard_host[k][0]):
            ops.raise_on_guard_trips('PipelinedStreamingPlan.results')
        out = []
        for n in range(self.batch):
            c = int(self.cnt_host[k][n])
            r = self.res_host[k][n, :c]
            out.append((r[:, :5].clone(), r[:, 5].long()))
        return out


# ------------------------------------------------------------------------------------------------
# DenoisingAggregator: residual dense blocks + DCN-based temporal attention fusion
# ------------------------------------------------------------------------------------------------
def build_denoising_aggregator(b: Builder, sd, x, n, num_blocks=1, num_layers=6, emb_nums=3, with_rdb=True, with_taf=True,
                               deform_groups=8, p='', tag='agg'):
    """DenoisingAggregator.forward (mmtracking/mmtrack/models/aggregators/denoising_aggregator.py:186-200) on
    x (n, h, w, C) fp16 NHWC = the feature maps of the n frames of a clip; returns the output buffer (same shape).

    Dense concatenations are channel slices of one buffer (DenseLayer :32-34 writes its growth channels behind its input),
    x + lff(.) is the lff conv's residual epilogue (:66), relu(conv) + skip the ACT_FIRST residual epilogue (:198-199, :151),
    the n x n (reference frame i, frame j) pairs of the fusion loop (:139-149) one batch of n*n images, the offset / mask
    conv writes fp32 and the deformable conv is llv_modulated_deform_im2col followed by the implicit-GEMM kernel as a
    1x1 conv over the 9*C column tensor."""
    L = lib()
    _, h, w, c = x.shape
    cv = lambda q, pad=1: ops.pack_conv(sd[p + q + '.weight'], sd.get(p + q + '.bias'), 1, pad)
    pc1 = cv('conv1')
    m = pc1.cout
    growth = sd[f'{p}rdbs.0.layers.0.conv.weight'].shape[0] if with_rdb else 0
    tot = m + growth * num_layers

    def add(name, a0, ld0, a1, ld1):
        """a0 + a1 over (n*h*w, m) with row strides ld0 / ld1 -> new contiguous buffer"""
        y = b.buf(n, h, w, m)
        b.plan.add(name, L.llv_add_f16, (C.c_void_p(a0.data_ptr()), C.c_void_p(a1.data_ptr()), C.c_void_p(y.data_ptr()),
                                         C.c_int64(n * h * w), m, ld0, ld1, m), 0.0, 6.0 * y.numel())
        b.plan.keep += [a0, a1]
        return y
    # t / t_ld: the running feature map and its row stride (inside a dense block's concat buffer it is a channel slice)
    t = b.buf(n, h, w, tot)
    t_ld = tot
    b.conv_desc(f'{tag}.conv1', pc1, [x], t, n, [(h, w, h, w, 0, 0)], c, t_ld, flags=CONV_RELU)
    x_f, xf_ld = t, t_ld
    if with_rdb:
        for bi in range(num_blocks):
            q = f'rdbs.{bi}'
            cat = t                                     # channels [0, m) hold the block input
            for l in range(num_layers):
                cin = m + growth * l
                b.conv_desc(f'{tag}.{q}.layers.{l}', cv(f'{q}.layers.{l}.conv'), [cat], cat[..., cin:], n, [(h, w, h, w, 0, 0)], tot, tot,
                            flags=CONV_RELU)
            last = bi == num_blocks - 1
            t = b.buf(n, h, w, m if last else tot)      # the next block's concat buffer, or a plain map after the last block
            t_ld = m if last else tot
            b.conv_desc(f'{tag}.{q}.lff', cv(f'{q}.lff', 0), [cat], t, n, [(h, w, h, w, 0, 0)], tot, t_ld, res=cat, res_ld=tot,
                        flags=CONV_RES_ADD)
    t = b.conv(f'{tag}.conv2', add(f'{tag}.skip_f', t, t_ld, x_f, xf_ld), cv('conv2'), CONV_RELU)
    x_m = t
    if with_taf:
        q = 'taf.'
        xt = b.conv(f'{tag}.taf.conv1', t, cv(q + 'conv1'), CONV_RELU)
        mt = xt.shape[-1]
        pairs = b.buf(n * n, h, w, 2 * mt)
        b.plan.add(f'{tag}.taf.pairs', L.llv_taf_pair_concat, (C.c_void_p(xt.data_ptr()), C.c_void_p(pairs.data_ptr()), n, h, w, mt),
                   0.0, 2.0 * (xt.numel() + pairs.numel()))
        x_set = b.conv(f'{tag}.taf.offset_conv', pairs, cv(q + 'offset_conv'))
        om = b.conv(f'{tag}.taf.conv_offset', x_set, cv(q + 'dcn_pack.conv_offset'), CONV_OUT_F32)
        # the deformable conv samples frame j of pair (i, j): the first mt channels of `pairs` (row stride 2*mt)
        col = b.buf(n * n, h, w, 9 * mt)
        b.plan.add(f'{tag}.taf.dcn_im2col', L.llv_modulated_deform_im2col,
                   (C.c_void_p(pairs.data_ptr()), C.c_void_p(om.data_ptr()), C.c_void_p(col.data_ptr()), n * n, h, w, mt, 2 * mt,
                    om.shape[-1], 9 * mt, 3, 3, 1, 1, 1, deform_groups, h, w), 0.0, 2.0 * col.numel() + 4.0 * om.numel())
        x_dcn = b.conv(f'{tag}.taf.dcn_gemm', col, ops.pack_dcn_weight(sd[p + q + 'dcn_pack.weight'], sd.get(p + q + 'dcn_pack.bias')))
        x_cor = b.buf(n * n, h, w, mt)
        b.plan.add(f'{tag}.taf.mul_ref', L.llv_taf_mul_ref, (C.c_void_p(x_dcn.data_ptr()), C.c_void_p(xt.data_ptr()),
                                                              C.c_void_p(x_cor.data_ptr()), n, h, w, mt), 0.0, 4.0 * x_cor.numel())
        for e in range(emb_nums):
            x_cor = b.conv(f'{tag}.taf.emb_conv.{e}', x_cor, cv(f'{q}emb_conv.{e}'))
        fused = b.buf(n, h, w, mt)
        b.plan.add(f'{tag}.taf.softmax_fuse', L.llv_taf_softmax_fuse, (C.c_void_p(x_cor.data_ptr()), C.c_void_p(xt.data_ptr()),
                                                                        C.c_void_p(fused.data_ptr()), n, h, w, mt), 0.0, 2.0 * x_cor.numel())
        b.plan.keep += [x_cor, x_dcn, om, x_set]
        # relu(conv2(out)) + x_m (the skip before conv3, :197-198)
        t = b.conv(f'{tag}.taf.conv2', fused, cv(q + 'conv2'), CONV_RELU | CONV_RES_ADD | CONV_ACT_FIRST, res=x_m)
    else:
        t = add(f'{tag}.skip_m', t, m, x_m, m)
    # relu(conv3(.)) + identity (:198-199)
    return b.conv(f'{tag}.conv3', t, cv('conv3'), CONV_RELU | CONV_RES_ADD | CONV_ACT_FIRST, res=x)


def build_denoising2_aggregator(b: Builder, sd, x_noise, last_all_x, n, in_channel, out_channel, layer_name, rdb_blocks, taf_embs,
                                downsample, with_rdb, with_taf, deform_groups=8, p='layers.', tag='agg2'):
    """Denoising2Aggergator.forward (mmtracking/mmtrack/models/aggregators/denoising2_aggregator.py:214-244) on the NHWC fp16 stage
    features x_noise[i] (n, h_i, w_i, C_i) and last_all_x (n, h, w, out_channel[-1]); returns (x_noise_out buffers, the sum
    all_x[-1] + x_list[-1]). The concatenation in front of conv1 is a two-source launch (in_channel[i] == out_channel[i-1] in every
    config of the reference: equal-width sources), a dense block's concatenations are channel slices of one buffer, the RDB's local
    residual is the lff conv's epilogue, the temporal fusion is the plan of build_denoising_aggregator's."""
    L = lib()
    cv = lambda q, pad=1, stride=1: ops.pack_conv(sd[p + q + '.weight'], sd.get(p + q + '.bias'), stride, pad)

    def add(name, a0, a1):
        nn_, h, w, c = a0.shape
        y = b.buf(nn_, h, w, c)
        b.plan.add(name, L.llv_add_f16, (C.c_void_p(a0.data_ptr()), C.c_void_p(a1.data_ptr()), C.c_void_p(y.data_ptr()),
                                         C.c_int64(nn_ * h * w), c, c, c, c), 0.0, 6.0 * y.numel())
        b.plan.keep += [a0, a1]
        return y
    ns = len(in_channel)
    outs, prev = [], None
    for i in range(ns):
        l = layer_name[i]
        xi = x_noise[i]
        _, h, w, c = xi.shape
        m = in_channel[i]
        growth = sd[f'{p}{l}_rdb.0.layers.0.conv.weight'].shape[0] if with_rdb[i] else 0
        tot = m + growth * 3
        pc1 = cv(f'{l}_conv1')
        if i > 0 and prev.shape[-1] != c:
            raise LlvError('Denoising2Aggergator: in_channel[i] must equal out_channel[i-1] (two equal-width sources of conv1)')
        if i > 0:
            pc1 = ops.pack_conv(sd[f'{p}{l}_conv1.weight'], sd.get(f'{p}{l}_conv1.bias'), 1, 1, n_src=2)
        t = b.buf(n, h, w, tot)
        b.conv_desc(f'{tag}.{l}.conv1', pc1, [xi] if i == 0 else [xi, prev], t, n, [(h, w, h, w, 0, 0)], c, tot)
        t_ld = tot
        if with_rdb[i]:
            for bi in range(rdb_blocks[i]):
                q = f'{l}_rdb.{bi}'
                cat = t
                for k in range(3):
                    cin = m + growth * k
                    b.conv_desc(f'{tag}.{q}.layers.{k}', cv(f'{q}.layers.{k}.conv'), [cat], cat[..., cin:], n, [(h, w, h, w, 0, 0)], tot, tot,
                                flags=CONV_RELU)
                last = bi == rdb_blocks[i] - 1
                t = b.buf(n, h, w, m if last else tot)
                t_ld = m if last else tot
                b.conv_desc(f'{tag}.{q}.lff', cv(f'{q}.lff', 0), [cat], t, n, [(h, w, h, w, 0, 0)], tot, t_ld, res=cat, res_ld=tot,
                            flags=CONV_RES_ADD)
        if with_taf[i]:
            t = _plan_taf(b, sd, t, n, p + f'{l}_taf.', f'{tag}.{l}.taf', taf_embs[i], deform_groups, CONV_RELU, None)
        xa = add(f'{tag}.{l}.add_noise', t, xi)
        outs.append(xa)
        prev = b.conv(f'{tag}.{l}.conv2', t if i == ns - 1 else xa, cv(f'{l}_conv2', 1, 2 if downsample[i] else 1))
    return outs, add(f'{tag}.all_x', last_all_x, prev)


def _plan_taf(b, sd, t, n, p, tag, emb_nums, deform_groups, out_flags, res):
    """TemporalAttentionFusion.forward (denoising_aggregator.py:135-152) on t (n, h, w, c): returns the conv2 launch's output with
    `out_flags` / `res` as its epilogue (relu(conv2(.)) for the bare module)."""
    L = lib()
    _, h, w, c = t.shape
    cv = lambda q, pad=1: ops.pack_conv(sd[p + q + '.weight'], sd.get(p + q + '.bias'), 1, pad)
    xt = b.conv(f'{tag}.conv1', t, cv('conv1'), CONV_RELU)
    mt = xt.shape[-1]
    pairs = b.buf(n * n, h, w, 2 * mt)
    b.plan.add(f'{tag}.pairs', L.llv_taf_pair_concat, (C.c_void_p(xt.data_ptr()), C.c_void_p(pairs.data_ptr()), n, h, w, mt),
               0.0, 2.0 * (xt.numel() + pairs.numel()))
    x_set = b.conv(f'{tag}.offset_conv', pairs, cv('offset_conv'))
    om = b.conv(f'{tag}.conv_offset', x_set, cv('dcn_pack.conv_offset'), CONV_OUT_F32)
    col = b.buf(n * n, h, w, 9 * mt)
    b.plan.add(f'{tag}.dcn_im2col', L.llv_modulated_deform_im2col,
               (C.c_void_p(pairs.data_ptr()), C.c_void_p(om.data_ptr()), C.c_void_p(col.data_ptr()), n * n, h, w, mt, 2 * mt,
                om.shape[-1], 9 * mt, 3, 3, 1, 1, 1, deform_groups, h, w), 0.0, 2.0 * col.numel() + 4.0 * om.numel())
    x_dcn = b.conv(f'{tag}.dcn_gemm', col, ops.pack_dcn_weight(sd[p + 'dcn_pack.weight'], sd.get(p + 'dcn_pack.bias')))
    x_cor = b.buf(n * n, h, w, mt)
    b.plan.add(f'{tag}.mul_ref', L.llv_taf_mul_ref, (C.c_void_p(x_dcn.data_ptr()), C.c_void_p(xt.data_ptr()),
                                                      C.c_void_p(x_cor.data_ptr()), n, h, w, mt), 0.0, 4.0 * x_cor.numel())
    for e in range(emb_nums):
        x_cor = b.conv(f'{tag}.emb_conv.{e}', x_cor, cv(f'emb_conv.{e}'))
    fused = b.buf(n, h, w, mt)
    b.plan.add(f'{tag}.softmax_fuse', L.llv_taf_softmax_fuse, (C.c_void_p(x_cor.data_ptr()), C.c_void_p(xt.data_ptr()),
                                                                C.c_void_p(fused.data_ptr()), n, h, w, mt), 0.0, 2.0 * x_cor.numel())
    b.plan.keep += [x_cor, x_dcn, om, x_set]
    return b.conv(f'{tag}.conv2', fused, cv('conv2'), out_flags, res=res)


# ------------------------------------------------------------------------------------------------
# FlowNetSimple
# ------------------------------------------------------------------------------------------------
class FlowNetPlan:
    """FlowNetSimple.forward (mmtracking/mmtrack/models/motion/flownet_simple.py:192-244) for `batch` image pairs of
    size (h, w): prepare (de-normalise, mask, bilinear x img_scale_factor) -> 9 strided/plain convs with LeakyReLU(0.1)
    -> 4 refinement levels (predict_flow, 4x4/s2 transposed convs as 3x3 conv + PixelShuffle with crop_like, concat by
    writing into channel slices of one buffer) -> predict_flow -> bilinear x(4/img_scale_factor), x flow scale.
    Input: self.pairs (batch, 6, h, w) fp32 NCHW (img, ref stacked on C, as dff.py:211 builds it). Output: self.flow
    (batch, 2, h', w') fp32 NCHW with h' = floor(floor(h*s)/4 ... ) as the reference's arithmetic gives."""
    CONV = dict(inplanes=(6, 64, 128, 256, 512, 512), kernel_size=(7, 5, 5, 3, 3, 3), num_convs=(1, 1, 2, 2, 2, 2))
    DECONV_IN = (386, 770, 1026, 1024)

    def __init__(self, sd, batch, height, width, device, img_scale_factor=0.5, flow_scale_factor=5.0,
                 flow_img_norm_std=(255.0, 255.0, 255.0), flow_img_norm_mean=(0.411, 0.432, 0.450),
                 img_norm_mean=(123.675, 116.28, 103.53), img_norm_std=(58.395, 57.12, 57.375), img_shape=None):
        import math
        b = Builder(device)
        self.builder, self.plan, self.batch = b, b.plan, batch
        sd = _cpu_sd(sd)
        s = float(img_scale_factor)
        ho, wo = int(math.floor(height * s)), int(math.floor(width * s))
        self.pairs = b.buf(batch, 6, height, width, dtype=torch.float32)
        # flow_img = (x*std + mean)/flow_std - flow_mean  ==  x*scale + shift   (flownet_simple.py:178-179)
        sc = [img_norm_std[c % 3] / flow_img_norm_std[c % 3] for c in range(6)]
        sh = [img_norm_mean[c % 3] / flow_img_norm_std[c % 3] - flow_img_norm_mean[c % 3] for c in range(6)]
        self.scale = torch.tensor(sc, device=device, dtype=torch.float32)
        self.shift = torch.tensor(sh, device=device, dtype=torch.float32)
        x0 = b.buf(batch, ho, wo, 8)
        ih, iw = (img_shape[0], img_shape[1]) if img_shape is not None else (height, width)
        self._prep_args = [C.c_void_p(self.pairs.data_ptr()), C.c_void_p(x0.data_ptr()), batch, height, width, ho, wo, ih, iw,
                           C.c_void_p(self.scale.data_ptr()), C.c_void_p(self.shift.data_ptr()), C.c_float(s)]
        b.plan.keep += [self.scale, self.shift]
        b.plan.add('flow.prepare', lib().llv_flow_prepare, tuple(self._prep_args), 0.0, 4.0 * self.pairs.numel())
        self._prep_step = len(b.plan.steps) - 1
        leaky = dict(flags=CONV_LEAKY, act_slope=0.1)
        # ---- encoder; conv2..conv6 outputs are skip connections: each is written straight into the channel slice
        # [0, C) of the concat buffer its refinement level reads (concat = cat(conv_out, deconv_out, upflow))
        inpl, ks, ncv = self.CONV['inplanes'], self.CONV['kernel_size'], self.CONV['num_convs']
        planes_of = [inpl[i + 1] if i < 5 else 2 * inpl[5] for i in range(6)]         # 64,128,256,512,512,1024
        deconv_planes = {2: 64, 3: 128, 4: 256, 5: 512}                                 # deconv{i} output channels
        x, xh, xw, xc = x0, ho, wo, 8
        skips = {}
        for i in range(6):
            pl = planes_of[i]
            for j in range(ncv[i]):
                k = ks[i] if (j == 0 or i != 2) else 3
                k = ks[i] if j == 0 else (3 if i == 2 else ks[i])
                stride = 2 if j == 0 else 1
                wt = sd[f'conv{i + 1}.{j}.conv.weight']
                in_map = None
                if i == 0 and j == 0:
                    in_map = [0, 1, 2, 3, 4, 5, -1, -1]
                pc = ops.pack_conv(wt, sd.get(f'conv{i + 1}.{j}.conv.bias'), stride, (k - 1) // 2, in_map=in_map)
                oh, ow = ops.conv_out_size(xh, xw, pc)
                last = j == ncv[i] - 1
                if last and 1 <= i <= 4:          # conv2..conv5 outputs head a concat buffer: [conv | deconv | upflow(2->8)]
                    tot = pl + deconv_planes[i + 1] + 8
                    out = b.buf(batch, oh, ow, tot)
                else:
                    tot = pl
                    out = b.buf(batch, oh, ow, pl)
                b.conv_desc(f'flow.conv{i + 1}.{j}', pc, [x], out, batch, [(xh, xw, oh, ow, 0, 0)], xc, tot, **leaky)
                x, xh, xw, xc = out, oh, ow, tot
            skips[i + 1] = (x, xh, xw, pl, xc)      # conv{i+1}: buffer, h, w, real channels, ld
        # ---- refinement (flownet_simple.py:217-233): i = 4..1 over conv_outs = (conv2..conv6)
        cat_buf, ch, cw, c_real, cld = skips[6]
        cat_map = list(range(1024))                 # padded channel -> original channel of concat_out
        for lvl in (5, 4, 3, 2):                    # deconv5..deconv2 / predict_flow6..3 / upsample_flow5..2
            tgt, th, tw, tc, tld = skips[lvl]       # conv_outs[i-1]
            dpl = deconv_planes[lvl]
            # flow = predict_flow{lvl+1}(concat_out): 3x3, no bias, no activation -> 8-channel buffer (2 real)
            pcf = ops.pack_conv(sd[f'predict_flow{lvl + 1}.conv.weight'], None, 1, 1, in_map=cat_map, cout_store=8)
            fl = b.buf(batch, ch, cw, 8)
            b.conv_desc(f'flow.predict_flow{lvl + 1}', pcf, [cat_buf], fl, batch, [(ch, cw, ch, cw, 0, 0)], cld, 8)
            # upflow = crop_like(upsample_flow{lvl}(flow)) -> channels [tc+dpl, tc+dpl+2) of the target's concat buffer
            pcu = ops.pack_deconv4x4s2(sd[f'upsample_flow{lvl}.conv.weight'], None, in_map=[0, 1, -1, -1, -1, -1, -1, -1])
            b.conv_desc(f'flow.upsample_flow{lvl}', pcu, [fl], tgt[..., tc + dpl:], batch, [(ch, cw, ch, cw, 0, 0)], 8, tld,
                        flags=CONV_PIXSHUF, ps_out=(th, tw))
            # deconv_out = crop_like(deconv{lvl}(concat_out)) + LeakyReLU -> channels [tc, tc+dpl)
            pcd = ops.pack_deconv4x4s2(sd[f'deconv{lvl}.conv.weight'], None, in_map=cat_map)
            b.conv_desc(f'flow.deconv{lvl}', pcd, [cat_buf], tgt[..., tc:], batch, [(ch, cw, ch, cw, 0, 0)], cld, tld,
                        flags=CONV_PIXSHUF | CONV_LEAKY, act_slope=0.1, ps_out=(th, tw))
            cat_buf, ch, cw, cld = tgt, th, tw, tld
            cat_map = list(range(tc + dpl)) + [tc + dpl, tc + dpl + 1] + [-1] * 6
        pcf = ops.pack_conv(sd['predict_flow.conv.weight'], None, 1, 1, in_map=cat_map, cout_store=8)
        fl = b.buf(batch, ch, cw, 8)
        b.conv_desc('flow.predict_flow', pcf, [cat_buf], fl, batch, [(ch, cw, ch, cw, 0, 0)], cld, 8)
        # ---- NHWC fp16 (2 real channels) -> NCHW fp32, bilinear x(4/s), x(4/s)*flow_scale (flownet_simple.py:235-242)
        fl32 = b.buf(batch, 2, ch, cw, dtype=torch.float32)
        b.plan.add('flow.unpack', lib().llv_unpack_nhwc_f16_to_nchw_f32,
                   (C.c_void_p(fl.data_ptr()), C.c_void_p(fl32.data_ptr()), batch, 2, ch, cw, 8), 0.0, 0.0)
        up = 4.0 / s
        fh, fw = int(math.floor(ch * up)), int(math.floor(cw * up))
        self.flow = b.buf(batch, 2, fh, fw, dtype=torch.float32)
        b.plan.add('flow.upsample', lib().llv_bilinear_resize_f32,
                   (C.c_void_p(fl32.data_ptr()), C.c_void_p(self.flow.data_ptr()), batch * 2, ch, cw, fh, fw, C.c_float(up),
                    C.c_float(up * flow_scale_factor)), 0.0, 0.0)

    def set_img_shape(self, img_shape):
        """The un-padded image extent masks the flow image (flownet_simple.py:180-181): host-side edit of the launch args."""
        a = list(self._prep_args)
        a[7], a[8] = int(img_shape[0]), int(img_shape[1])
        name, fn, _, fl, nb = self.plan.steps[self._prep_step]
        self.plan.steps[self._prep_step] = (name, fn, tuple(a), fl, nb)
