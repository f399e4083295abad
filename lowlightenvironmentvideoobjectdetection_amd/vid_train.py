"""Training of the flow / aggregation video detectors on the device: the forward_train bodies of

  DFF     mmtracking/mmtrack/models/vid/dff.py:49-182     flow = motion(cat(img, ref)); x_l = flow_warp_feats(extract_feat(ref)_l, flow)
  FGFA    mmtracking/mmtrack/models/vid/fgfa.py:52-194    x_l = aggregator(feat(img)_l, flow_warp_feats(feat(refs)_l, flows))
  LLVOD   mmtracking/mmtrack/models/vid/llvod.py:45-154   x_l = aggregator(feat(key)_l, feat(other frames)_l)      (no flow)

for a dense-head detector, each followed by bbox_head.forward_train(x, ...) (AnchorHead.loss). One launch graph per geometry
(train_engine.TrainGraph with CUSTOM tape entries between the convolutions):

  frames -> ResNet -> FPN (batch = all frames) -> [warp the reference pyramids]  -> [stack | embed convs | aggregate] -> RetinaHead
                                                   llv_flow_warp / _bwd             llv_embed_aggregate_ring / _bwd

and one autograd Function around it, so the returned losses support .backward() and fill param.grad of detector AND aggregator
exactly as the reference's module tree does. The flow network trains too unless it is frozen (frozen_modules=['motion'], base.py:39-54):
the warp's backward also yields d(loss)/d(flow) (llv_flow_warp_bwd_flow: grid_sample's grid gradient through the /W, /H grid and the
bilinear resize), which flow_train.FlowNetTrainGraph carries through FlowNetSimple's convolutions and transposed convolutions.

The authors' forward_train of FGFA / LLVOD first passes the frames through general_clean_noise_pairs (random Poisson / Gaussian /
row noise; core/utils/add_noise.py:13-52 -- the import is commented out in fgfa.py:8 / llvod.py:6, so the reference raises
NameError there as it stands). It is restated as `add_noise.general_clean_noise_pairs` on the device and applied to every training
batch, as the reference's body does, unless train_cfg['add_noise'] is false; parity tests switch it off (random draws cannot be
shared with the CPU oracle)."""
import ctypes as C

import torch

from . import engine, ops, train_engine
from ._lib import CONV_RELU, LlvError, check, lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class VideoTrainGraph(train_engine.TrainGraph):
    """mode 'dff': one reference frame, warped.  mode 'fgfa': key frame (slot 0) + nr warped references, aggregated.
    mode 'llvod': n frames, the key frame is slot `key_slot`, aggregated without warping."""

    def __init__(self, model, mode, n_frames, height, width, device, key_slot=0, warp=True, loss_scale=512.0):
        det = model.detector
        bb = det.backbone
        self._init_common(det, dict(model.named_parameters()), dict(model.named_buffers()), device, 1, loss_scale)
        self.mode, self.n_frames, self.key_slot, self.warp = mode, n_frames, key_slot, bool(warp)
        b = engine.Builder(device)
        self.builder = b
        sd = engine._cpu_sd(model.state_dict())
        nf = n_frames
        self.x_in = b.buf(nf, height, width, 8)
        feats = engine.build_resnet(b, sd, self.x_in, p='detector.backbone.', **bb.plan_kwargs())
        pyr = self._build_neck(b, sd, det, feats)
        self.frame_pyramid = pyr
        c = pyr.buf.shape[-1]
        hw = pyr.level_hw
        L = lib()
        n_warp = {'dff': 1, 'fgfa': nf - 1, 'llvod': 0}[mode] if self.warp else 0
        self.flow = None
        # ---- the frames' pyramids, references warped to the key frame (dff / fgfa)
        if mode == 'llvod' or not self.warp:
            stack = pyr
        else:
            stack = engine.Pyramid(b.buf(pyr.total, c), hw, pyr.level_off, nf)
            self._flow_shape = None
            first = 1 if mode == 'fgfa' else 0            # fgfa: slot 0 is the key frame itself (copied)
            steps = []
            for l, (hh, ww) in enumerate(hw):
                src, dst = pyr.level(l), stack.level(l)
                if first:
                    b.plan.add(f'warp.copy{l}', L.llv_copy_d2d, (C.c_void_p(dst[0:1].data_ptr()), C.c_void_p(src[0:1].data_ptr()),
                                                                C.c_size_t(src[0:1].numel() * 2)), 0.0, 4.0 * src[0:1].numel())
                steps.append((l, src[first:], dst[first:], hh, ww))
            self._warp_levels = steps
            self._warp_step0 = len(b.plan.steps)
            self._n_warp = n_warp
            # the flow field's size is only known at the first call: the warp launches are added then (see set_flow)
            for l, src, dst, hh, ww in steps:
                b.plan.add(f'warp.level{l}', None, (), 0.0, 4.0 * src.numel())
            b.tape.append(dict(name='warp', custom=True, xs=[pyr.buf], y=stack.buf, bwd=VideoTrainGraph._warp_bwd))
        self.stack = stack
        # ---- temporal aggregation (fgfa / llvod): embed convs over every slot, softmax over the references
        if mode in ('fgfa', 'llvod'):
            from .aggregators import embed_packed_convs
            ag = model.aggregator
            emb = stack.buf
            for i, (pc, flags) in enumerate(embed_packed_convs(sd, ag.num_convs, prefix='aggregator.')):
                nxt = b.buf(stack.total, pc.cout)
                b.conv_desc(f'aggregator.embed{i}', pc, [emb], nxt, nf, stack.segs(), emb.shape[-1], nxt.shape[-1], flags=flags)
                emb = nxt
            off1 = [sum(a * bb_ for a, bb_ in hw[:i]) for i in range(len(hw))]
            agg = engine.Pyramid(b.buf(sum(a * bb_ for a, bb_ in hw), c), hw, off1, 1)
            for l, (hh, ww) in enumerate(hw):
                e_l, s_l = stack.level(l, emb), stack.level(l)
                b.plan.add(f'aggregate{l}', L.llv_embed_aggregate_ring,
                           (ops._ptr(e_l), ops._ptr(s_l), ops._ptr(agg.level(l)), nf, key_slot, hh * ww, c, emb.shape[-1], c, c), 0.0, 0.0)
            self.emb = emb
            b.tape.append(dict(name='aggregate', custom=True, xs=[emb, stack.buf], y=agg.buf, bwd=VideoTrainGraph._aggregate_bwd))
            head_in = agg
        else:
            head_in = stack                                # dff: the warped pyramid of the one reference frame
        self.pyramid = head_in                             # what TrainGraph.forward / set_logit_grads index the logits with
        self._build_head(b, sd, det, head_in)
        fs = getattr(bb, 'frozen_stages', -1)
        frozen = (['stem.'] if fs >= 0 else []) + [f'detector.backbone.layer{i}.' for i in range(1, fs + 1)]
        self._finish(b, frozen)
        self._dx32 = {}
        self.want_dflow, self.dflow = False, None

    # ---- what sits on either side of the temporal part: FPN + RetinaHead here, ChannelMapper + RPN convs in TwoStageVideoGraph
    def _build_neck(self, b, sd, det, feats):
        neck = det.neck
        return engine.build_fpn(b, sd, feats, neck.start_level, -1, neck.num_outs, neck.add_extra_convs, neck.relu_before_extra_convs,
                                p='detector.neck.')

    def _build_head(self, b, sd, det, head_in):
        self.cls_buf, self.reg_buf = engine.build_retina_head(b, sd, head_in, det.bbox_head.stacked_convs, True, p='detector.bbox_head.',
                                                              pingpong=False)

    # ---- the flow field (an input of every step, not a parameter)
    def set_flow(self, flow):
        """flow: (n_warp, 2, Hf, Wf) fp32 on the device, flow of (key frame -> reference i) as FlowNetSimple returns it."""
        if not self.warp or self.mode == 'llvod':
            return
        flow = flow.float().contiguous()
        if self.flow is None or self.flow.shape != flow.shape:
            self.flow = torch.empty_like(flow)
            self.plan.keep.append(self.flow)
            L = lib()
            for k, (l, src, dst, hh, ww) in enumerate(self._warp_levels):
                cc = src.shape[-1]
                name, _, _, fl, nb = self.plan.steps[self._warp_step0 + k]
                self.plan.steps[self._warp_step0 + k] = (name, L.llv_flow_warp,
                                                         (ops._ptr(src), ops._ptr(self.flow), ops._ptr(dst), src.shape[0], hh, ww, cc, cc,
                                                          flow.shape[2], flow.shape[3]), fl, nb)
        assert flow.shape[0] == self._n_warp, f'{flow.shape[0]} flow fields for {self._n_warp} warped frames'
        self.flow.copy_(flow)

    @staticmethod
    def _warp_bwd(g, op):
        """d(frame pyramids) += scatter of d(warped pyramids) along the flow (llv_flow_warp_bwd, fp32 atomics); fgfa's key slot is
        a copy: its gradient passes straight through."""
        L = lib()
        gy, gx = g.gview(g.stack.buf), g.gview(g.frame_pyramid.buf)
        first = 1 if g.mode == 'fgfa' else 0
        for l, src, dst, hh, ww in g._warp_levels:
            gyl, gxl = g.stack.level(l, gy), g.frame_pyramid.level(l, gx)
            cc = src.shape[-1]
            if first:
                gxl[0:1].add_(gyl[0:1])
            n = src.shape[0]
            key = (l, n)
            if key not in g._dx32:
                g._dx32[key] = torch.empty((n, hh, ww, cc), dtype=torch.float32, device=g.device)
            dx = g._dx32[key]
            dx.zero_()
            check(L.llv_flow_warp_bwd(ops._ptr(gyl[first:]), ops._ptr(g.flow), ops._ptr(dx), n, hh, ww, cc, gyl.shape[-1],
                                      g.flow.shape[2], g.flow.shape[3], _stream()), 'llv_flow_warp_bwd')
            check(L.llv_add_f32_to_f16(ops._ptr(dx), ops._ptr(gxl[first:]), n * hh * ww, cc, gxl.shape[-1], _stream()), 'llv_add_f32_to_f16')
            if g.want_dflow:                              # the flow network trains too: d(loss)/d(flow) of this level (llv_flow_warp_bwd_flow)
                if g.dflow is None or g.dflow.shape != g.flow.shape:
                    g.dflow = torch.zeros_like(g.flow)
                check(L.llv_flow_warp_bwd_flow(ops._ptr(gyl[first:]), ops._ptr(src), ops._ptr(g.flow), ops._ptr(g.dflow), n, hh, ww, cc, gyl.shape[-1],
                                               src.shape[-1], g.flow.shape[2], g.flow.shape[3], _stream()), 'llv_flow_warp_bwd_flow')

    @staticmethod
    def _aggregate_bwd(g, op):
        L = lib()
        gout = g.gview(g.pyramid.buf)
        gemb, gstack = g.gview(g.emb), g.gview(g.stack.buf)
        c = g.stack.buf.shape[-1]
        for l, (hh, ww) in enumerate(g.stack.level_hw):
            check(L.llv_embed_aggregate_bwd(ops._ptr(g.stack.level(l, g.emb)), ops._ptr(g.stack.level(l)), ops._ptr(g.pyramid.level(l, gout)),
                                            ops._ptr(g.stack.level(l, gemb)), ops._ptr(g.stack.level(l, gstack)), g.n_frames, g.key_slot,
                                            hh * ww, c, g.emb.shape[-1], c, gout.shape[-1], _stream()), 'llv_embed_aggregate_bwd')

    def forward(self, frames, flow=None):
        """frames (n_frames, 3, H, W) fp32 normalised on the device (+ the flow fields) -> per-level fp32 NHWC logits (cls, reg)."""
        self.set_flow(flow) if flow is not None else None
        if self.warp and self.mode != 'llvod' and self.flow is None:
            raise LlvError('VideoTrainGraph.forward: this mode warps the reference pyramids and needs the flow field')
        return super().forward(frames)


class VideoFunction(torch.autograd.Function):
    """The whole conv body of a video detector's training step as one differentiable op (train_engine.DetectorFunction with the
    frames and the flow as inputs): outputs the per-level logits, backward returns the parameter gradients."""

    @staticmethod
    def forward(ctx, graph, frames, flow, *params):
        graph.repack()
        cls, reg = graph.forward(frames, flow)
        ctx.graph, ctx.n_levels = graph, len(cls)
        return tuple(t.clone() for t in cls) + tuple(t.clone() for t in reg)

    @staticmethod
    def backward(ctx, *grads):
        g = ctx.graph
        n = ctx.n_levels
        g.zero_param_grads()
        g.want_dflow = bool(ctx.needs_input_grad[2]) and g.flow is not None
        if g.want_dflow and g.dflow is not None:
            g.dflow.zero_()
        g.backward([x.contiguous() for x in grads[:n]], [x.contiguous() for x in grads[n:]])
        pg = g.param_grads()
        inv = 1.0 / g.loss_scale
        dflow = (g.dflow * inv) if (g.want_dflow and g.dflow is not None) else None
        return (None, None, dflow) + tuple((pg[k] * inv) if k in pg else None for k in g.trainable)


def video_forward_train(model, mode, frames, flow, key_slot, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, warp=True):
    """Shared tail of DFF / FGFA / LLVOD.forward_train for a dense head: graph (cached per geometry), logits, AnchorHead.loss."""
    from .training import anchor_head_loss
    det = model.detector
    if frames.device.type != 'cuda':
        raise LlvError('forward_train runs on the MI355X (conv forward/backward kernels); no CPU fallback')
    if not hasattr(det, 'bbox_head'):
        raise LlvError('forward_train of the video detectors is built for a dense-head (single-stage) detector')
    nf, _, h, w = frames.shape
    for m in (model, det, getattr(model, 'aggregator', None)):       # inference plans hold packed copies of the parameters
        if m is not None and hasattr(m, '_plans'):
            for k in [k for k in m._plans if not (isinstance(k, tuple) and k and k[0] == 'train')]:
                del m._plans[k]
    key = ('train', mode, nf, h, w, key_slot, bool(warp), str(frames.device))
    graph = model._plan(key, lambda: VideoTrainGraph(model, mode, nf, h, w, frames.device, key_slot=key_slot, warp=warp))
    outs = VideoFunction.apply(graph, frames, flow, *[graph.params[k] for k in graph.trainable])
    n = len(outs) // 2
    head = det.bbox_head
    cls_scores = [t.permute(0, 3, 1, 2)[:, :head.num_anchors * head.cls_out_channels] for t in outs[:n]]
    bbox_preds = [t.permute(0, 3, 1, 2)[:, :head.num_anchors * 4] for t in outs[n:]]
    dev = frames.device
    losses = anchor_head_loss(head, cls_scores, bbox_preds, [b.to(dev) for b in gt_bboxes], [l.to(dev) for l in gt_labels], img_metas,
                              gt_bboxes_ignore)
    if losses is None:
        raise LlvError('no valid anchors inside the image')
    return losses


def motion_flow(model, pairs, img_metas):
    """flow = motion(pairs): under no_grad when the flow network is frozen (frozen_modules=['motion']), otherwise through
    flow_train.flownet_train_forward (FlowNetSimple on the training tape; the warp's backward then also returns d(loss)/d(flow))."""
    if any(p.requires_grad for p in model.motion.parameters()):
        from .flow_train import flownet_train_forward
        return flownet_train_forward(model.motion, pairs, img_metas)
    with torch.no_grad():
        return model.motion(pairs, img_metas)
