"""Training of the two-stage SELSA video detector on the device:

  SELSA.forward_train               mmtracking/mmtrack/models/vid/selsa.py:47-170
  RPNHead.loss / forward_train      mmdetection/mmdet/models/dense_heads/rpn_head.py:49-80, base_dense_head.py:23-59
  SelsaRoIHead.forward_train        mmtracking/mmtrack/models/roi_heads/selsa_roi_head.py:10-111
  BBoxHead.get_targets / loss       mmdetection/mmdet/models/roi_heads/bbox_heads/bbox_head.py:103-268

Split of the work:
  * every CONVOLUTION (backbone incl. the dilated DC5 stage, ChannelMapper, the RPN's three convs) is a launch of ONE training graph
    (TwoStageBody below = train_engine.TrainGraph with three outputs): forward on the conv engine with kept activations, backward =
    wgrad / dgrad kernels, eval-mode-BN affine gradients, as for the dense detector ("Backward pass");
  * RoIAlign is the device kernel both ways (llv_roi_align / llv_roi_align_bwd through roi_extractors._RoIAlignFn);
  * the RoI head's fully connected layers and proposal attention (SelsaBBoxHead.forward_autograd) are plain library GEMMs through
    torch on the device under autograd -- the in-tree GEMM path of the test step (selsa.linear_pc / selsa_attention) has no backward;
  * targets: llv_iou_assign (MaxIoUAssigner) for anchors and proposals, RandomSampler, llv_softmax_ce / BCE / SmoothL1 losses.
The graph's outputs (feature maps of all frames, RPN logits of the key frame) are leaves of one autograd Function; loss.backward()
sends d(feature map) -- the sum of RoIAlign's scatter for key and reference RoIs -- and d(RPN logits) back into it."""
import torch

from . import engine, ops, train_engine
from ._lib import CONV_OUT_F32, CONV_RELU, LlvError


class TwoStageBody(train_engine.TrainGraph):
    """frames (n, 3, H, W) -> backbone -> ChannelMapper -> feature map f (n, C, h, w); RPN convs on frame `key` -> objectness / deltas."""

    def __init__(self, model, n_frames, height, width, device, loss_scale=512.0, key=0):
        det = model.detector
        bb, neck, rpn = det.backbone, det.neck, det.rpn_head
        # backbone maps in front of the neck's inputs (the authors' config: out_indices=(3, 3) with a one-conv ChannelMapper,
        # TwoStageDetector.extract_feat) are returned beside the neck's output and not used by forward_train
        if neck is None or len(neck.convs) != 1 or len(bb.out_indices) < 1 or len(rpn.anchor_generator.strides) != 1:
            raise LlvError('two-stage training graph: a one-conv ChannelMapper on the last backbone output and a single RPN level '
                           '(the geometry of the SELSA configs)')
        self.key = None if key is None else int(key)       # None: the RPN runs on every frame (a plain Faster R-CNN batch)
        self._init_common(det, dict(model.named_parameters()), dict(model.named_buffers()), device, 1, loss_scale)
        self.n_frames = n_frames
        b = engine.Builder(device)
        self.builder = b
        sd = engine._cpu_sd(model.state_dict())
        self.x_in = b.buf(n_frames, height, width, 8)
        feats = engine.build_resnet(b, sd, self.x_in, p='detector.backbone.', **bb.plan_kwargs())
        nconv = neck.convs[0].conv
        f = b.conv('detector.neck.conv0', feats[-1], ops.pack_conv(sd['detector.neck.convs.0.conv.weight'], sd.get('detector.neck.convs.0.conv.bias'),
                                                                   1, (nconv.kernel_size[0] - 1) // 2), CONV_RELU if neck.act else 0)
        self.f, self.f_channels = f, nconv.out_channels
        pk = lambda name, pad: ops.pack_conv(sd[f'detector.rpn_head.{name}.weight'], sd[f'detector.rpn_head.{name}.bias'], 1, pad)
        t = b.conv('detector.rpn_head.rpn_conv', f if self.key is None else f[self.key:self.key + 1], pk('rpn_conv', 1), CONV_RELU)
        self.rpn_cls = b.conv('detector.rpn_head.rpn_cls', t, pk('rpn_cls', 0), CONV_OUT_F32)
        self.rpn_reg = b.conv('detector.rpn_head.rpn_reg', t, pk('rpn_reg', 0), CONV_OUT_F32)
        self.n_cls, self.n_reg = rpn.num_anchors * rpn.cls_out_channels, rpn.num_anchors * 4
        fs = getattr(bb, 'frozen_stages', -1)
        self._finish(b, (['stem.'] if fs >= 0 else []) + [f'detector.backbone.layer{i}.' for i in range(1, fs + 1)])

    def forward_frames(self, frames):
        ops.pack_input(frames.float().contiguous(), 8, out=self.x_in)
        self.plan.run()
        f = ops.unpack_output(self.f, self.f_channels)                                    # (n, C, h, w) fp32
        cls = self.rpn_cls[..., :self.n_cls].permute(0, 3, 1, 2).contiguous()
        reg = self.rpn_reg[..., :self.n_reg].permute(0, 3, 1, 2).contiguous()
        return f, cls, reg

    def backward_outputs(self, df, dcls, dreg):
        """Gradients of the loss with respect to the three outputs (NCHW fp32, any may be None) -> parameter gradients on the tape."""
        self.zero_act_grads()
        s = self.loss_scale
        if df is not None:
            g = self.gview(self.f)
            g[..., :self.f_channels] = (df.permute(0, 2, 3, 1) * s).to(torch.float16)
        for buf, d, n in ((self.rpn_cls, dcls, self.n_cls), (self.rpn_reg, dreg, self.n_reg)):
            if d is not None:
                g = self.grads[train_engine._sptr(buf)]
                g.view(buf.shape[0], buf.shape[1], buf.shape[2], -1)[..., :n] = (d.permute(0, 2, 3, 1) * s).to(torch.float16)
        self.backward_ops(list(reversed(self.tape)))


class TwoStageFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, frames, *params):
        graph.repack()
        f, cls, reg = graph.forward_frames(frames)
        ctx.graph = graph
        return f, cls, reg

    @staticmethod
    def backward(ctx, df, dcls, dreg):
        g = ctx.graph
        g.zero_param_grads()
        g.backward_outputs(df, dcls, dreg)
        pg = g.param_grads()
        inv = 1.0 / g.loss_scale
        return (None, None) + tuple((pg[k] * inv) if k in pg else None for k in g.trainable)


def _two_stage_losses(model, all_imgs, key, img_metas, gt_bboxes, gt_labels, ref_img_metas, gt_bboxes_ignore, proposals, ref_proposals):
    """The shared body of selsa.py:133-170 and selsa_fastdvd_darkfarm.py:176-227: frames -> training graph -> RPN loss + proposals on
    frame `key`, proposals only on the other frames, SelsaRoIHead.forward_train."""
    det = model.detector
    nf, _, h, w = all_imgs.shape
    dev = all_imgs.device
    for m in (model, det, det.backbone, det.neck, det.rpn_head):        # inference plans hold packed copies of the parameters
        if hasattr(m, '_plans'):
            for k in [k for k in m._plans if not (isinstance(k, tuple) and k and k[0] == 'train2')]:
                del m._plans[k]
    graph = model._plan(('train2', nf, h, w, key, str(dev)), lambda: TwoStageBody(model, nf, h, w, dev, key=key))
    f, rpn_cls, rpn_reg = TwoStageFunction.apply(graph, all_imgs, *[graph.params[k] for k in graph.trainable])
    others = [i for i in range(nf) if i != key]
    x, ref_x = (f[key:key + 1],), (f[others],)
    losses = dict()
    gt_bboxes = [b.to(dev) for b in gt_bboxes]
    gt_labels = [l.to(dev) for l in gt_labels]
    rpn = det.rpn_head
    if det.with_rpn:                                   # selsa.py:146 (the `proposals=` arguments are read only without an RPN)
        tc, sc = det.train_cfg or {}, det.test_cfg or {}
        proposal_cfg = tc.get('rpn_proposal', sc.get('rpn'))
        rpn_losses, proposal_list = rpn.forward_train(x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=gt_bboxes_ignore,
                                                      proposal_cfg=proposal_cfg, outs=([rpn_cls], [rpn_reg]))
        losses.update(rpn_losses)
        with torch.no_grad():
            ref_proposals_list = rpn.simple_test_rpn(tuple(t.detach() for t in ref_x), ref_img_metas[0])
    else:
        proposal_list, ref_proposals_list = proposals, ref_proposals
    roi_losses = det.roi_head.forward_train(x, ref_x, img_metas, proposal_list, ref_proposals_list, gt_bboxes, gt_labels, gt_bboxes_ignore)
    losses.update(roi_losses)
    return losses


def selsa_forward_train(model, img, img_metas, gt_bboxes, gt_labels, ref_img, ref_img_metas, ref_gt_bboxes=None, ref_gt_labels=None,
                        gt_bboxes_ignore=None, proposals=None, ref_proposals=None):
    """selsa.py:47-170: key + reference frames through backbone + neck together; RPN loss and proposals on the key frame, proposals
    only on the references; SelsaRoIHead.forward_train on (x, ref_x). Returns the dict of loss tensors (loss_rpn_cls, loss_rpn_bbox,
    loss_cls, acc, loss_bbox)."""
    if img.device.type != 'cuda':
        raise LlvError('forward_train runs on the MI355X (conv forward/backward kernels); no CPU fallback')
    assert len(img) == 1, 'selsa video detector only supports 1 batch size per gpu for now.'
    all_imgs = torch.cat((img.float(), ref_img[0].float()), dim=0).contiguous()
    return _two_stage_losses(model, all_imgs, 0, img_metas, gt_bboxes, gt_labels, ref_img_metas, gt_bboxes_ignore, proposals, ref_proposals)


def selsa_fastdvd_forward_train(model, img, img_metas, gt_bboxes, gt_labels, ref_img, ref_img_metas, ref_gt_bboxes=None, ref_gt_labels=None,
                                gt_bboxes_ignore=None, proposals=None, ref_proposals=None):
    """selsa_fastdvd_darkfarm.py:65-227: img (1, 2c, H, W) = noisy ‖ clean key frame, ref_img (1, 4, 2c, H, W) the four references.
    The five frames [ref0, ref1, key, ref2, ref3] are denoised from their cyclic 5-stacks (:163-169, SelsaFastDVDnetDetect.
    denoise_cyclic: the inference plan, the cleaner is FROZEN here as in FastDVDnetDetect.forward_train -- its loss term is reported,
    no gradient reaches its weights), all five go through the detector's training graph, the key frame is image 2."""
    import torch.nn.functional as F
    if img.device.type != 'cuda':
        raise LlvError('forward_train runs on the MI355X (conv forward/backward kernels); no CPU fallback')
    assert len(img) == 1, 'selsa video detector only supports 1 batch size per gpu for now.'
    assert ref_img.shape[1] == 4, 'FastDVDnet support frames is 4.'
    if img.shape[1] % 4 == 0:
        raise LlvError('SelsaFastDVDnetDetect.forward_train: RAW (4-channel) input is not on this path (FastDVDnet plan takes RGB frames)')
    if model.with_cleaner and any(p.requires_grad for p in model.cleaner.parameters()):
        # the reference trains an unfrozen cleaner through the detection losses and the denoising loss (:171-200); its backward pass is
        # not built here, and a loss_denoising term without a gradient behind it would look live in the logs
        raise LlvError("SelsaFastDVDnetDetect.forward_train: freeze the cleaner (frozen_modules='cleaner', as the authors' config does); "
                       "training it is not built")
    noise_img, clean_img = img[:, :3], img[:, 3:]
    noise_ref, clean_ref = ref_img[:, :, :3], ref_img[:, :, 3:]
    clean_imgs = torch.cat([clean_ref[0][:2], clean_img, clean_ref[0][2:]], dim=0).float()
    noise_imgs = torch.cat([noise_ref[0][:2], noise_img, noise_ref[0][2:]], dim=0).float()
    with torch.no_grad():
        denoised = model.denoise_cyclic(noise_imgs).float()
    losses = dict()
    if model.with_cleaner:
        fn = {'l1': F.l1_loss, 'l2': F.mse_loss, 'smooth_l1': F.smooth_l1_loss}.get(model.loss_type)
        if fn is None:
            raise NotImplementedError()
        losses[f'loss_denoising_{model.loss_type}'] = fn(denoised, clean_imgs)
    losses.update(_two_stage_losses(model, denoised.contiguous(), 2, img_metas, gt_bboxes, gt_labels, ref_img_metas, gt_bboxes_ignore,
                                    proposals, ref_proposals))
    return losses


class _AsVideoModel(torch.nn.Module):
    """A detector under the name the training graph's parameter keys use (`detector.*`)."""

    def __init__(self, det):
        super().__init__()
        self.detector = det


def faster_rcnn_forward_train(det, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None, proposals=None, **kwargs):
    """TwoStageDetector.forward_train (mmdetection/mmdet/models/detectors/two_stage.py:115-176) for a batch of still images with a
    StandardRoIHead: backbone + neck + RPN convs of all images as one training graph (TwoStageBody with the RPN on every frame), RPN loss
    and proposals per image, StandardRoIHead.forward_train on the graph's feature map."""
    from .roi_heads import StandardRoIHead
    if img.device.type != 'cuda':
        raise LlvError('forward_train runs on the MI355X (conv forward/backward kernels); no CPU fallback')
    if not isinstance(det.roi_head, StandardRoIHead):
        raise LlvError('the two-stage detector with a SELSA RoI head is trained through the SELSA video detectors (key + reference frames)')
    n, _, h, w = img.shape
    dev = img.device
    for m in (det, det.backbone, det.neck, det.rpn_head):               # inference plans hold packed copies of the parameters
        if hasattr(m, '_plans'):
            for k in [k for k in m._plans if not (isinstance(k, tuple) and k and k[0] == 'train2')]:
                del m._plans[k]
    wrap = det.__dict__.get('_llv_as_video')
    if wrap is None:
        wrap = det.__dict__['_llv_as_video'] = _AsVideoModel(det)
    graph = det._plan(('train2', n, h, w, None, str(dev)), lambda: TwoStageBody(wrap, n, h, w, dev, key=None))
    f, rpn_cls, rpn_reg = TwoStageFunction.apply(graph, img.float().contiguous(), *[graph.params[k] for k in graph.trainable])
    x = (f,)
    gt_bboxes = [b.to(dev) for b in gt_bboxes]
    gt_labels = [l.to(dev) for l in gt_labels]
    losses = dict()
    if det.with_rpn:
        tc, sc = det.train_cfg or {}, det.test_cfg or {}
        rpn_losses, proposal_list = det.rpn_head.forward_train(x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=gt_bboxes_ignore,
                                                               proposal_cfg=tc.get('rpn_proposal', sc.get('rpn')), outs=([rpn_cls], [rpn_reg]))
        losses.update(rpn_losses)
    else:
        proposal_list = proposals
    losses.update(det.roi_head.forward_train(x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore, gt_masks, **kwargs))
    return losses
