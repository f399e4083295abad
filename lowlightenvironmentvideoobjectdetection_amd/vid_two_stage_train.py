"""Training of the flow / aggregation video detectors around a TWO-STAGE detector -- the branch the reference's own configs take
(`configs/vid/dff/*.py`, `configs/vid/fgfa/*.py` on `_base_/models/faster_rcnn_r50_dc5.py`):

  DFF.forward_train     mmtracking/mmtrack/models/vid/dff.py:139-173     x = flow_warp_feats(extract_feat(ref), flow)
  FGFA.forward_train    mmtracking/mmtrack/models/vid/fgfa.py:149-185    x = aggregator(feat(img), flow_warp_feats(feat(refs), flows))
  LLVOD.forward_train   mmtracking/mmtrack/models/vid/llvod.py:87-145    x = aggregator(feat(key), feat(other frames))
  then  rpn_head.forward_train(x, ...) -> (rpn losses, proposals);  roi_head.forward_train(x, img_metas, proposals, gt ...)
        (mmdetection/mmdet/models/dense_heads/base_dense_head.py:23-59, roi_heads/standard_roi_head.py:70-139)

One launch graph per geometry (vid_train.VideoTrainGraph with the neck / head hooks replaced):

  frames -> ResNet (DC5) -> ChannelMapper -> [warp references] -> [embed convs | aggregate] -> x (1, C, h, w) -> RPN convs
                                              llv_flow_warp / _bwd    llv_embed_aggregate_ring / _bwd

Outputs of the one autograd Function around it: x (the feature map the RoI head pools from), RPN objectness and deltas. The RoI
head (RoIAlign forward / backward kernels, the head's layers under autograd) hangs off x; loss.backward() returns d(x) -- RoIAlign's
scatter -- and d(RPN logits) into the graph, whose backward pass fills the parameter gradients of backbone, neck, aggregator and RPN
and, when the flow network trains, d(loss)/d(flow)."""
import torch

from . import engine, ops, train_engine
from ._lib import CONV_OUT_F32, CONV_RELU, LlvError
from .vid_train import VideoTrainGraph


class TwoStageVideoGraph(VideoTrainGraph):
    """VideoTrainGraph for a detector with `neck = ChannelMapper` (one conv on the last backbone map), `rpn_head` on one level."""

    def _build_neck(self, b, sd, det, feats):
        neck, rpn = det.neck, det.rpn_head
        if neck is None or len(neck.convs) != 1 or len(rpn.anchor_generator.strides) != 1:
            raise LlvError('two-stage video training graph: a one-conv ChannelMapper on the last backbone output and a single RPN level '
                           '(the geometry of the reference\'s faster_rcnn_r50_dc5 configs)')
        nconv = neck.convs[0].conv
        f = b.conv('detector.neck.conv0', feats[-1], ops.pack_conv(sd['detector.neck.convs.0.conv.weight'], sd.get('detector.neck.convs.0.conv.bias'),
                                                                   1, (nconv.kernel_size[0] - 1) // 2), CONV_RELU if neck.act else 0)
        n, h, w, c = f.shape
        self.f_channels = nconv.out_channels
        return engine.Pyramid(f.view(n * h * w, c), [(h, w)], [0], n)

    def _build_head(self, b, sd, det, head_in):
        rpn = det.rpn_head
        self.x_out = head_in.level(0)                                                      # (1, h, w, C): what the RoI head pools from
        assert self.x_out.shape[0] == 1
        pk = lambda name, pad: ops.pack_conv(sd[f'detector.rpn_head.{name}.weight'], sd[f'detector.rpn_head.{name}.bias'], 1, pad)
        t = b.conv('detector.rpn_head.rpn_conv', self.x_out, pk('rpn_conv', 1), CONV_RELU)
        self.rpn_cls = b.conv('detector.rpn_head.rpn_cls', t, pk('rpn_cls', 0), CONV_OUT_F32)
        self.rpn_reg = b.conv('detector.rpn_head.rpn_reg', t, pk('rpn_reg', 0), CONV_OUT_F32)
        self.n_cls, self.n_reg = rpn.num_anchors * rpn.cls_out_channels, rpn.num_anchors * 4

    def forward_frames(self, frames, flow=None):
        if flow is not None:
            self.set_flow(flow)
        if self.warp and self.mode != 'llvod' and self.flow is None:
            raise LlvError('TwoStageVideoGraph.forward_frames: this mode warps the reference features and needs the flow field')
        ops.pack_input(frames.float().contiguous(), 8, out=self.x_in)
        self.plan.run()
        x = ops.unpack_output(self.x_out.contiguous(), self.f_channels)                     # (1, C, h, w) fp32
        cls = self.rpn_cls[..., :self.n_cls].permute(0, 3, 1, 2).contiguous()
        reg = self.rpn_reg[..., :self.n_reg].permute(0, 3, 1, 2).contiguous()
        return x, cls, reg

    def backward_outputs(self, dx, dcls, dreg):
        self.zero_act_grads()
        s = self.loss_scale
        if dx is not None and train_engine._sptr(self.x_out) in self.grads:
            g = self.gview(self.x_out)
            g[..., :self.f_channels] = (dx.permute(0, 2, 3, 1) * s).to(torch.float16)
        for buf, d, n in ((self.rpn_cls, dcls, self.n_cls), (self.rpn_reg, dreg, self.n_reg)):
            if d is not None:
                g = self.grads[train_engine._sptr(buf)]
                g.view(buf.shape[0], buf.shape[1], buf.shape[2], -1)[..., :n] = (d.permute(0, 2, 3, 1) * s).to(torch.float16)
        self.backward_ops(list(reversed(self.tape)))


class TwoStageVideoFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, frames, flow, *params):
        graph.repack()
        ctx.graph = graph
        return graph.forward_frames(frames, flow)

    @staticmethod
    def backward(ctx, dx, dcls, dreg):
        g = ctx.graph
        g.zero_param_grads()
        g.want_dflow = bool(ctx.needs_input_grad[2]) and g.flow is not None
        if g.want_dflow and g.dflow is not None:
            g.dflow.zero_()
        g.backward_outputs(dx, dcls, dreg)
        pg = g.param_grads()
        inv = 1.0 / g.loss_scale
        dflow = (g.dflow * inv) if (g.want_dflow and g.dflow is not None) else None
        return (None, None, dflow) + tuple((pg[k] * inv) if k in pg else None for k in g.trainable)


def video_two_stage_forward_train(model, mode, frames, flow, key_slot, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, proposals=None,
                                  warp=True, gt_masks=None, **kwargs):
    """Shared tail of DFF / FGFA / LLVOD.forward_train for a two-stage detector (dff.py:153-173, fgfa.py:165-185, llvod.py:124-145)."""
    det = model.detector
    if frames.device.type != 'cuda':
        raise LlvError('forward_train runs on the MI355X (conv forward/backward kernels); no CPU fallback')
    nf, _, h, w = frames.shape
    dev = frames.device
    for m in (model, det, det.backbone, det.neck, det.rpn_head, getattr(model, 'aggregator', None)):   # inference plans hold packed parameters
        if m is not None and hasattr(m, '_plans'):
            for k in [k for k in m._plans if not (isinstance(k, tuple) and k and k[0] == 'train2v')]:
                del m._plans[k]
    key = ('train2v', mode, nf, h, w, key_slot, bool(warp), str(dev))
    graph = model._plan(key, lambda: TwoStageVideoGraph(model, mode, nf, h, w, dev, key_slot=key_slot, warp=warp))
    x, rpn_cls, rpn_reg = TwoStageVideoFunction.apply(graph, frames, flow, *[graph.params[k] for k in graph.trainable])
    x = (x,)
    gt_bboxes = [b.to(dev) for b in gt_bboxes]
    gt_labels = [l.to(dev) for l in gt_labels]
    losses = dict()
    if det.with_rpn:
        tc, sc = det.train_cfg or {}, det.test_cfg or {}
        proposal_cfg = tc.get('rpn_proposal', sc.get('rpn'))
        rpn_losses, proposal_list = det.rpn_head.forward_train(x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=gt_bboxes_ignore,
                                                               proposal_cfg=proposal_cfg, outs=([rpn_cls], [rpn_reg]))
        losses.update(rpn_losses)
    else:
        proposal_list = proposals
    losses.update(det.roi_head.forward_train(x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore, gt_masks, **kwargs))
    return losses
