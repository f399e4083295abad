"""FlowNetSimple on the training tape: the flow network of DFF / FGFA trained together with the detector, as
the reference's configs have it (mmtracking/configs/vid/dff, fgfa: `motion` is not among frozen_modules).

  FlowNetSimple.forward        mmtracking/mmtrack/models/motion/flownet_simple.py:192-244
  gradient into it             d(loss)/d(flow) from the feature warp (llv_flow_warp_bwd_flow, vid_train.VideoTrainGraph)

The forward is the one of engine.FlowNetPlan with a training-friendly layout: every conv writes a buffer of its own (one gradient
buffer per activation), the three-way concatenations (conv_out | deconv_out | upflow, :228-231) are explicit copies with a CUSTOM tape
entry, and the ConvTranspose2d(4, stride 2, pad 1) layers run as what they are -- the data gradient of a 4x4 / stride-2 convolution V
whose weight tensor IS the ConvTranspose weight (cin, cout, 4, 4) read as (Cout_V, Cin_V, 4, 4):

  forward        y = V^T x      llv_conv2d_nhwc_dgrad with the filter packed by llv_pack_conv_weights mode 2 (+ LeakyReLU in place)
  data gradient  dx += V dy     llv_conv2d_nhwc_fwd, 4x4 / stride 2, filter packed by mode 0
  weight grad.   dW = wgrad_V(input = dy, output gradient = x)   llv_conv2d_nhwc_wgrad; unpacked as V's (Cout_V, Cin_V, 4, 4) = the
                 ConvTranspose2d parameter's own layout

so no host-side re-packing happens between optimizer steps. crop_like (:16-20: the transposed conv's output cut to its skip tensor's
size from the top-left corner, e.g. 24 -> 23 rows at the reference's 360 x 640 scaled frames) is a tape entry of its own: the transposed
conv writes its full 2h x 2w map, the crop copies the kept window, and its backward adds the window's gradient into the full map's
(the cut rows / columns keep the zeros zero_act_grads left)."""
import ctypes as C
import math

import torch

from . import backward as B
from . import engine, ops, train_engine
from ._lib import CONV_LEAKY, CONV_PIXSHUF, CONV_RES_ADD, LlvError, check, lib

SLOPE = 0.1


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _sptr(t):
    return t.untyped_storage().data_ptr()


class _Deconv:
    """One ConvTranspose2d(cin, cout, 4, 2, 1, bias=False) of the network: x (n, h, w, x_ld) -> y (n, 2h, 2w, y_ld)."""

    def __init__(self, graph, wkey, x, y, act):
        self.g, self.wkey, self.x, self.y, self.act = graph, wkey, x, y, act
        w = graph.params[wkey]
        self.cin_d, self.cout_d = int(w.shape[0]), int(w.shape[1])
        dev = graph.device
        n, h, wd, x_ld = x.shape
        _, h2, w2, y_ld = y.shape
        if (h2, w2) != (2 * h, 2 * wd):
            raise LlvError(f'{wkey}: transposed conv {h}x{wd} -> {2 * h}x{2 * wd} written into a {h2}x{w2} buffer (crop_like is a tape entry of its own)')
        self.n, self.h, self.w, self.x_ld, self.y_ld = n, h, wd, x_ld, y_ld
        self.seg = (h2, w2, h, wd, 0, 0)                         # V's forward geometry: (2h, 2w) -> (h, w)
        # V^T (the layer's forward): window of the stride-1 conv over x that emits the four output phases
        KH, KW, pd, _, _ = B.dgrad_geometry(4, 4, 2, 1)
        self.cy = ops.round_up(self.cout_d, 8)                   # channels of y that V reads (V's padded Cin)
        rows = self.cy * 4
        w_rows, k_pad = ops.round_up(rows, 16), ops.round_up(KH * KW * x_ld, 32)
        self.pcd = ops.PackedConv(w=torch.zeros(1, w_rows, k_pad, dtype=torch.float16, device=dev), bias=torch.zeros(rows, dtype=torch.float32, device=dev),
                                  cin=x_ld, cout=rows, w_rows=w_rows, k_pad=k_pad, kh=KH, kw=KW, stride=1, pad=pd, groups=1, true_macs_per_pixel=0)
        # V (the data gradient): 4x4 / stride 2 over y -> x's channels
        cs = ops.round_up(self.cin_d, 4)
        wr, kp = ops.round_up(cs, 16), ops.round_up(16 * self.cy, 32)
        self.pcv = ops.PackedConv(w=torch.zeros(1, wr, kp, dtype=torch.float16, device=dev), bias=torch.zeros(cs, dtype=torch.float32, device=dev),
                                  cin=self.cy, cout=cs, w_rows=wr, k_pad=kp, kh=4, kw=4, stride=2, pad=1, groups=1, true_macs_per_pixel=0)
        self.dw = torch.zeros(1, wr, kp, dtype=torch.float32, device=dev)

    def repack(self):
        w = self.g.params[self.wkey].detach()
        B.pack_weights(w, None, None, 1e-5, self.pcd.w[0], cpad=self.pcd.cin, k_pad=self.pcd.k_pad, row0=0, mode=2, pad=1, KH=self.pcd.kh,
                       KW=self.pcd.kw, padp=self.pcd.pad, rows=self.pcd.cout)
        B.pack_weights(w, None, None, 1e-5, self.pcv.w[0], cpad=self.pcv.cin, k_pad=self.pcv.k_pad, row0=0, mode=0)

    def forward(self, stream=None):
        B.conv_dgrad(self.x, self.pcd, self.y, self.n, [self.seg], self.x_ld, self.y_ld, CONV_PIXSHUF, accumulate=False)
        if self.act:                                             # leaky(y) in place: y *= (y > 0 ? 1 : slope)
            B.act_bwd(self.y[..., :self.cy], self.y[..., :self.cy], None, None, SLOPE)
        return 0

    def backward(self, g):
        gy = g.gview(self.y)
        if self.act:
            B.act_bwd(gy[..., :self.cy], self.y[..., :self.cy], None, None, SLOPE)
        B.conv_wgrad(gy, self.x, self.dw, cin_pad=self.cy, cout=self.cin_d, w_rows=self.pcv.w_rows, k_pad=self.pcv.k_pad, kh=4, kw=4, stride=2, pad=1,
                     groups=1, batch=self.n, seg=self.seg, x_ld=gy.stride(-2), dy_ld=self.x_ld)
        if _sptr(self.x) in g.req:
            gx = g.gview(self.x)
            d = ops.make_conv_desc(self.pcv, [gy], gx, self.n, [self.seg], gy.stride(-2), gx.stride(-2), gx, gx.stride(-2), CONV_RES_ADD)
            wsb = ops.conv_workspace_bytes(d)
            if wsb:
                ws = torch.empty((wsb,), device=gy.device, dtype=torch.uint8)
                d.workspace, d.workspace_bytes = ws.data_ptr(), wsb
                d._ext.workspace = wsb
            ops.launch_conv(d)

    def grad(self):
        w = self.g.params[self.wkey]
        out = torch.empty_like(w, dtype=torch.float32)
        B.unpack_wgrad_device(self.dw[0], None, w.detach(), None, None, None, 1e-5, out, None, None, None, cpad=self.cy, k_pad=self.pcv.k_pad)
        return out


def _deconv_bwd(g, op):
    op['deconv'].backward(g)


def _crop_fwd(full, out, stream=None):
    """crop_like (flownet_simple.py:16-20): out = full[:, :h, :w]"""
    out.copy_(full[:, :out.shape[1], :out.shape[2], :])
    return 0


def _crop_bwd(g, op):
    full, out = op['xs'][0], op['y']
    if _sptr(full) in g.req:
        g.gview(full)[:, :out.shape[1], :out.shape[2], :].add_(g.gview(out))


def _concat_bwd(g, op):
    gcat = g.gview(op['y'])
    o = 0
    for src in op['xs']:
        c = src.shape[-1]
        if _sptr(src) in g.req:
            gs = g.gview(src)
            rows = src.numel() // c
            check(lib().llv_add_f16(gs.data_ptr(), gcat[..., o:].data_ptr(), gs.data_ptr(), rows, c, gs.stride(-2), gcat.stride(-2), gs.stride(-2), _stream()),
                  'llv_add_f16')
        o += c


class FlowNetTrainGraph(train_engine.TrainGraph):
    CONV = dict(inplanes=(6, 64, 128, 256, 512, 512), kernel_size=(7, 5, 5, 3, 3, 3), num_convs=(1, 1, 2, 2, 2, 2))

    def __init__(self, motion, batch, height, width, device, img_norm_cfg=None, img_shape=None, loss_scale=512.0):
        nc = img_norm_cfg or dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375])
        self._init_common(motion, dict(motion.named_parameters()), dict(motion.named_buffers()), device, batch, loss_scale)
        s = float(motion.img_scale_factor)
        ho, wo = int(math.floor(height * s)), int(math.floor(width * s))
        b = engine.Builder(device)
        self.builder = b
        sd = engine._cpu_sd(motion.state_dict())
        self.pairs = b.buf(batch, 6, height, width, dtype=torch.float32)
        fstd, fmean = motion.flow_img_norm_std, motion.flow_img_norm_mean
        sc = [nc['std'][c % 3] / fstd[c % 3] for c in range(6)]
        sh = [nc['mean'][c % 3] / fstd[c % 3] - fmean[c % 3] for c in range(6)]
        self.scale = torch.tensor(sc, device=device, dtype=torch.float32)
        self.shift = torch.tensor(sh, device=device, dtype=torch.float32)
        x0 = b.buf(batch, ho, wo, 8)
        ih, iw = (img_shape[0], img_shape[1]) if img_shape is not None else (height, width)
        b.plan.keep += [self.scale, self.shift]
        b.plan.add('flow.prepare', lib().llv_flow_prepare, (C.c_void_p(self.pairs.data_ptr()), C.c_void_p(x0.data_ptr()), batch, height, width, ho, wo,
                                                           ih, iw, C.c_void_p(self.scale.data_ptr()), C.c_void_p(self.shift.data_ptr()), C.c_float(s)),
                   0.0, 4.0 * self.pairs.numel())
        inpl, ks, ncv = self.CONV['inplanes'], self.CONV['kernel_size'], self.CONV['num_convs']
        x, skips = x0, {}
        for i in range(6):
            for j in range(ncv[i]):
                k = ks[i] if j == 0 else (3 if i == 2 else ks[i])
                pc = ops.pack_conv(sd[f'conv{i + 1}.{j}.conv.weight'], sd.get(f'conv{i + 1}.{j}.conv.bias'), 2 if j == 0 else 1, (k - 1) // 2)
                n_, h_, w_, c_ = x.shape
                oh, ow = ops.conv_out_size(h_, w_, pc)
                out = b.buf(batch, oh, ow, pc.cout)
                b.conv_desc(f'flow.conv{i + 1}.{j}', pc, [x], out, batch, [(h_, w_, oh, ow, 0, 0)], c_, pc.cout, flags=CONV_LEAKY, act_slope=SLOPE)
                x = out
            skips[i + 1] = x
        self.deconvs = []
        cat = skips[6]
        for lvl in (5, 4, 3, 2):
            tgt = skips[lvl]
            _, th, tw, tc = tgt.shape
            n_, ch, cw, cc = cat.shape
            pcf = ops.pack_conv(sd[f'predict_flow{lvl + 1}.conv.weight'], None, 1, 1, cout_store=8)      # cat's padding channels are the default in_map's
            fl = b.buf(batch, ch, cw, 8)
            b.conv_desc(f'flow.predict_flow{lvl + 1}', pcf, [cat], fl, batch, [(ch, cw, ch, cw, 0, 0)], cc, 8)
            up = b.buf(batch, th, tw, 8)
            dpl = int(self.params[f'deconv{lvl}.conv.weight'].shape[1])
            dec = b.buf(batch, th, tw, dpl)
            if th > 2 * ch or tw > 2 * cw:
                raise LlvError(f'FlowNetSimple level {lvl}: skip tensor {th}x{tw} larger than the transposed conv\'s {2 * ch}x{2 * cw}')
            for nm, xin, yout, act in ((f'upsample_flow{lvl}', fl, up, False), (f'deconv{lvl}', cat, dec, True)):
                crop = (th, tw) != (2 * ch, 2 * cw)
                yfull = b.buf(batch, 2 * ch, 2 * cw, yout.shape[-1]) if crop else yout
                dc = _Deconv(self, nm + '.conv.weight', xin, yfull, act)
                self.deconvs.append(dc)
                b.plan.add('flow.' + nm, dc.forward, (), 0.0, 0.0)
                b.plan.keep += [dc.pcd.w, dc.pcd.bias, dc.pcv.w, dc.pcv.bias]
                b.tape.append(dict(name='flow.' + nm, custom=True, xs=[xin], y=yfull, bwd=_deconv_bwd, deconv=dc))
                if crop:
                    b.plan.add('flow.' + nm + '.crop_like', _crop_fwd, (yfull, yout), 0.0, 4.0 * yout.numel())
                    b.tape.append(dict(name='flow.' + nm + '.crop_like', custom=True, xs=[yfull], y=yout, bwd=_crop_bwd))
            ncat = b.buf(batch, th, tw, tc + dpl + 8)
            o = 0
            for src in (tgt, dec, up):
                c = src.shape[-1]
                zero = b.buf(*src.shape)
                b.plan.add(f'flow.cat{lvl}.{o}', lib().llv_add_f16, (C.c_void_p(src.data_ptr()), C.c_void_p(zero.data_ptr()), C.c_void_p(ncat[..., o:].data_ptr()),
                                                                   src.numel() // c, c, c, c, ncat.shape[-1]), 0.0, 4.0 * src.numel())
                o += c
            b.tape.append(dict(name=f'flow.cat{lvl}', custom=True, xs=[tgt, dec, up], y=ncat, bwd=_concat_bwd))
            cat = ncat
        n_, ch, cw, cc = cat.shape
        pcf = ops.pack_conv(sd['predict_flow.conv.weight'], None, 1, 1, cout_store=8)
        self.fl = b.buf(batch, ch, cw, 8)
        b.conv_desc('flow.predict_flow', pcf, [cat], self.fl, batch, [(ch, cw, ch, cw, 0, 0)], cc, 8)
        self.fl32 = b.buf(batch, 2, ch, cw, dtype=torch.float32)
        b.plan.add('flow.unpack', lib().llv_unpack_nhwc_f16_to_nchw_f32, (C.c_void_p(self.fl.data_ptr()), C.c_void_p(self.fl32.data_ptr()), batch, 2, ch, cw, 8),
                   0.0, 0.0)
        self.up = 4.0 / s
        self.mul = self.up * float(motion.flow_scale_factor)
        fh, fw = int(math.floor(ch * self.up)), int(math.floor(cw * self.up))
        self.flow = b.buf(batch, 2, fh, fw, dtype=torch.float32)
        b.plan.add('flow.upsample', lib().llv_bilinear_resize_f32, (C.c_void_p(self.fl32.data_ptr()), C.c_void_p(self.flow.data_ptr()), batch * 2, ch, cw, fh, fw,
                                                                   C.c_float(self.up), C.c_float(self.mul)), 0.0, 0.0)
        self._finish(b, [])
        self.trainable = self.trainable + [d.wkey for d in self.deconvs if self.params[d.wkey].requires_grad]
        self.dfl32 = torch.zeros_like(self.fl32)

    def _repack_each(self, ops_list=None):
        super()._repack_each(ops_list)
        for d in self.deconvs:
            d.repack()

    def forward_flow(self, pairs):
        self.pairs.copy_(pairs.float())
        self.plan.run()
        return self.flow.clone()

    def zero_param_grads(self):
        super().zero_param_grads()
        for d in self.deconvs:
            d.dw.zero_()

    def backward_flow(self, dflow):
        """dflow (batch, 2, fh, fw) fp32 = d(loss)/d(flow) -> weight gradients on the tape (times loss_scale)."""
        self.zero_act_grads()
        b, _, ch, cw = self.fl32.shape
        self.dfl32.zero_()
        check(lib().llv_bilinear_resize_f32_bwd(ops._ptr(dflow.float().contiguous()), ops._ptr(self.dfl32), b * 2, ch, cw, self.flow.shape[2], self.flow.shape[3],
                                                self.up, self.mul, _stream()), 'llv_bilinear_resize_f32_bwd')
        # loss scale of THIS pass: the gradient that enters here has no fixed magnitude (the flow is x8 x flow_scale_factor of the conv output,
        # and the detector's loss may be a mean or a sum), so the power of two that puts its largest element near 2^10 is taken each time
        m = float(self.dfl32.abs().max().item())
        if m > 0.0 and math.isfinite(m):
            self.loss_scale = float(2.0 ** max(-10, min(16, math.floor(math.log2(1024.0 / m)))))
        g = self.gview(self.fl)
        g[..., :2] = (self.dfl32.permute(0, 2, 3, 1) * self.loss_scale).to(torch.float16)
        self.backward_ops(list(reversed(self.tape)))

    def param_grads(self, only_op=None, dst=None):
        out = super().param_grads(only_op, dst)
        if only_op is None:
            for d in self.deconvs:
                if self.params[d.wkey].requires_grad:
                    out[d.wkey] = d.grad()
        return out


class FlowNetFunction(torch.autograd.Function):
    """FlowNetSimple.forward as one differentiable op: (image pairs, parameters) -> flow; backward = the tape."""

    @staticmethod
    def forward(ctx, graph, pairs, *params):
        graph.repack()
        ctx.graph = graph
        return graph.forward_flow(pairs)

    @staticmethod
    def backward(ctx, dflow):
        g = ctx.graph
        g.zero_param_grads()
        g.backward_flow(dflow)
        pg = g.param_grads()
        inv = 1.0 / g.loss_scale
        return (None, None) + tuple((pg[k] * inv) if k in pg else None for k in g.trainable)


def flownet_train_forward(motion, pairs, img_metas):
    """motion(pairs, img_metas) with gradients into the flow network's parameters (the training path of DFF / FGFA when `motion` is not
    frozen). pairs (N, 6, H, W) normalised image pairs on the device -> flow (N, 2, H', W')."""
    n, c, h, w = pairs.shape
    assert c == 6
    meta = img_metas[0]
    key = ('train', n, h, w, str(pairs.device), tuple(meta.get('img_norm_cfg', {}).get('mean', ())), tuple(meta['img_shape'][:2]))
    for k in [k for k in motion._plans if not (isinstance(k, tuple) and k and k[0] == 'train')]:
        del motion._plans[k]
    graph = motion._plan(key, lambda: FlowNetTrainGraph(motion, n, h, w, pairs.device, meta.get('img_norm_cfg'), meta['img_shape']))
    return FlowNetFunction.apply(graph, pairs, *[graph.params[k] for k in graph.trainable])
