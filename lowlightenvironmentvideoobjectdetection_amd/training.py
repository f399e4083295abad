"""Training side of the dense head: targets + losses on the device kernels.

  AnchorHead.get_anchors / get_targets / _get_targets_single / loss_single / loss
        mmdetection/mmdet/models/dense_heads/anchor_head.py:145-493
  heavy parts -> kernels: MaxIoUAssigner (llv_iou_assign: k x n IoU, both arg-maxes, low-quality match, no Python loop
  over gts) and the sigmoid focal loss over (N*anchors, C) logits (llv_sigmoid_focal_loss_fwd/bwd, autograd Function).
  The remaining glue touches only the positives (a few hundred rows) and stays in torch.
The convolution backward pass lives in train_engine.py (llv_conv2d_nhwc_wgrad + forward kernels on re-packed weights);
SingleStageDetector.forward_train goes through it (single_stage_forward_train below).
"""
import torch

from . import ops
from ._lib import LlvError, check, lib
from .core.anchor import anchor_inside_flags, images_to_levels


def _bbox2delta(proposals, gt, means, stds):
    """mmdetection/mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:88-130 (positives only: tiny)"""
    proposals, gt = proposals.float(), gt.float()
    px = (proposals[..., 0] + proposals[..., 2]) * 0.5
    py = (proposals[..., 1] + proposals[..., 3]) * 0.5
    pw = proposals[..., 2] - proposals[..., 0]
    ph = proposals[..., 3] - proposals[..., 1]
    gx = (gt[..., 0] + gt[..., 2]) * 0.5
    gy = (gt[..., 1] + gt[..., 3]) * 0.5
    gw = gt[..., 2] - gt[..., 0]
    gh = gt[..., 3] - gt[..., 1]
    deltas = torch.stack([(gx - px) / pw, (gy - py) / ph, torch.log(gw / pw), torch.log(gh / ph)], dim=-1)
    return deltas.sub_(deltas.new_tensor(means).unsqueeze(0)).div_(deltas.new_tensor(stds).unsqueeze(0))


def _delta2bbox(rois, deltas, means, stds, wh_ratio_clip=16 / 1000):
    """delta_xywh_bbox_coder.py:134-237 without max_shape (the training-time decode of anchor_head.py:404-409)"""
    import numpy as np
    means = deltas.new_tensor(means).view(1, -1)
    stds = deltas.new_tensor(stds).view(1, -1)
    d = deltas * stds + means
    dx, dy, dw, dh = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
    max_ratio = abs(float(np.log(wh_ratio_clip)))
    dw = dw.clamp(min=-max_ratio, max=max_ratio)
    dh = dh.clamp(min=-max_ratio, max=max_ratio)
    px, py = (rois[:, 0] + rois[:, 2]) * 0.5, (rois[:, 1] + rois[:, 3]) * 0.5
    pw, ph = rois[:, 2] - rois[:, 0], rois[:, 3] - rois[:, 1]
    gw, gh = pw * dw.exp(), ph * dh.exp()
    gx, gy = px + pw * dx, py + ph * dy
    return torch.stack([gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5], dim=-1)


def get_targets_single(head, flat_anchors, valid_flags, gt_bboxes, gt_labels, img_meta):
    """anchor_head.py:175-271 with PseudoSampler; assignment over the inside anchors via a validity mask
    (identical to assigning the subset and unmapping: outside anchors neither win a gt nor get a label)."""
    tc = head.train_cfg
    inside = anchor_inside_flags(flat_anchors, valid_flags, img_meta['img_shape'][:2], tc['allowed_border'])
    if not bool(inside.any()):
        return (None,) * 6
    a = head.assigner
    gt_inds, _ = ops.iou_assign(gt_bboxes, flat_anchors, inside, a.pos_iou_thr, a.neg_iou_thr, a.min_pos_iou,
                                a.match_low_quality, a.gt_max_assign_all)
    n = flat_anchors.shape[0]
    sampler = getattr(head, 'sampler', None)
    if head.sampling and sampler is not None and type(sampler).__name__ != 'PseudoSampler':
        # anchor_head.py:217-219 with a real sampler (the RPN: RandomSampler(num=256, pos_fraction=0.5, add_gt_as_proposals=False)):
        # anchors outside the image carry gt_inds == -1 and are neither positives nor negatives
        from .core.assign import AssignResult
        sr = sampler.sample(AssignResult(gt_bboxes.shape[0], gt_inds, None, None), flat_anchors, gt_bboxes)
        pos_inds, neg_inds = sr.pos_inds, sr.neg_inds
    else:
        pos_inds = torch.nonzero(gt_inds > 0, as_tuple=False).squeeze(-1)
        neg_inds = torch.nonzero(gt_inds == 0, as_tuple=False).squeeze(-1)
    bbox_targets = torch.zeros_like(flat_anchors)
    bbox_weights = torch.zeros_like(flat_anchors)
    labels = flat_anchors.new_full((n,), head.num_classes, dtype=torch.long)
    label_weights = flat_anchors.new_zeros(n, dtype=torch.float)
    if len(pos_inds) > 0:
        pos_gt = gt_inds[pos_inds] - 1
        if head.reg_decoded_bbox:                        # IoU-family losses take boxes, anchor_head.py:246-249
            bbox_targets[pos_inds, :] = gt_bboxes[pos_gt, :].float()
        else:
            bbox_targets[pos_inds, :] = _bbox2delta(flat_anchors[pos_inds], gt_bboxes[pos_gt, :], head.bbox_coder.means,
                                                    head.bbox_coder.stds)
        bbox_weights[pos_inds, :] = 1.0
        labels[pos_inds] = gt_labels[pos_gt] if gt_labels is not None else 0
        label_weights[pos_inds] = 1.0 if tc['pos_weight'] <= 0 else tc['pos_weight']
    if len(neg_inds) > 0:
        label_weights[neg_inds] = 1.0
    return labels, label_weights, bbox_targets, bbox_weights, pos_inds, neg_inds


def anchor_head_loss(head, cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None):
    """AnchorHead.loss (anchor_head.py:425-493). cls_scores / bbox_preds: per-level NCHW GPU tensors."""
    if head.train_cfg is None:
        raise LlvError('AnchorHead.loss needs train_cfg (assigner, allowed_border, pos_weight)')
    featmap_sizes = [f.size()[-2:] for f in cls_scores]
    assert len(featmap_sizes) == head.anchor_generator.num_levels
    device = cls_scores[0].device
    if device.type != 'cuda':
        raise LlvError('the loss path runs on the MI355X (iou-assign and focal-loss kernels); no CPU fallback')
    mlvl_anchors = head.anchor_generator.grid_anchors(featmap_sizes, device)
    num_level_anchors = [a.size(0) for a in mlvl_anchors]
    flat_anchors = torch.cat(mlvl_anchors)
    per_img = []
    for i, meta in enumerate(img_metas):
        flags = torch.cat(head.anchor_generator.valid_flags(featmap_sizes, meta['pad_shape'], device))
        per_img.append(get_targets_single(head, flat_anchors, flags, gt_bboxes[i].to(device),
                                          gt_labels[i].to(device) if gt_labels is not None else None, meta))
    if any(r[0] is None for r in per_img):
        return None
    num_total_pos = sum(max(r[4].numel(), 1) for r in per_img)
    num_total_neg = sum(max(r[5].numel(), 1) for r in per_img)
    num_total_samples = num_total_pos + num_total_neg if head.sampling else num_total_pos
    labels_list = images_to_levels([r[0] for r in per_img], num_level_anchors)
    label_weights_list = images_to_levels([r[1] for r in per_img], num_level_anchors)
    bbox_targets_list = images_to_levels([r[2] for r in per_img], num_level_anchors)
    bbox_weights_list = images_to_levels([r[3] for r in per_img], num_level_anchors)
    losses_cls, losses_bbox = [], []
    anchors_list = images_to_levels([flat_anchors] * len(img_metas), num_level_anchors)
    for cls, reg, anc, lab, lw, bt, bw in zip(cls_scores, bbox_preds, anchors_list, labels_list, label_weights_list,
                                              bbox_targets_list, bbox_weights_list):          # loss_single, anchor_head.py:375-423
        cls = cls.permute(0, 2, 3, 1).reshape(-1, head.cls_out_channels)
        losses_cls.append(head.loss_cls(cls, lab.reshape(-1), lw.reshape(-1), avg_factor=num_total_samples))
        reg = reg.permute(0, 2, 3, 1).reshape(-1, 4)
        if head.reg_decoded_bbox:
            reg = _delta2bbox(anc.reshape(-1, 4), reg, head.bbox_coder.means, head.bbox_coder.stds)
        losses_bbox.append(head.loss_bbox(reg, bt.reshape(-1, 4), bw.reshape(-1, 4), avg_factor=num_total_samples))
    return dict(loss_cls=losses_cls, loss_bbox=losses_bbox)


class FusedHeadLoss:
    """AnchorHead.loss for the training step WITHOUT a host round trip (train_engine.TrainStep): per image llv_iou_assign over the
    cached flat anchors, then llv_retina_loss -- labels / weights / box targets, FocalLoss and L1 (SmoothL1) terms, their derivatives
    written straight into the fp16 gradient twins of the logit buffers (times loss_weight / avg_factor * loss_scale), avg_factor counted
    on the device. Same arithmetic as anchor_head_loss above (tests/test_gpu_train.py compares the two); what it replaces is the
    ~350 element-wise launches and the four host synchronisations of the module-level path."""

    @staticmethod
    def eligible(head):
        lc, lb, a = head.loss_cls, head.loss_bbox, getattr(head, 'assigner', None)
        return (type(lc).__name__ == 'FocalLoss' and getattr(lc, 'use_sigmoid', False) and lc.reduction == 'mean'
                and type(lb).__name__ in ('L1Loss', 'SmoothL1Loss') and lb.reduction == 'mean' and not head.reg_decoded_bbox
                and not head.sampling and type(a).__name__ == 'MaxIoUAssigner' and a.ignore_iof_thr <= 0 and head.train_cfg is not None)

    def __init__(self, graph, head, device):
        import ctypes as C
        from ._lib import RetinaLossDesc
        self.graph, self.head, self.device = graph, head, device
        pyr = graph.pyramid
        hw = [tuple(int(v) for v in x) for x in pyr.level_hw]
        assert len(hw) == head.anchor_generator.num_levels
        A = head.num_anchors
        self.flat_anchors = torch.cat(head.anchor_generator.grid_anchors(hw, device)).float().contiguous()
        self.a_total = self.flat_anchors.shape[0]
        self.hw, self.batch = hw, graph.batch
        n = graph.batch
        self.gt_inds = torch.zeros((n, self.a_total), dtype=torch.int64, device=device)
        self.max_ov = torch.zeros((self.a_total,), dtype=torch.float32, device=device)
        self.max_gt = 0
        self._grow(64)
        self.scratch = torch.zeros(2 * len(hw) * 2 + 4, dtype=torch.float32, device=device)     # sums | losses | avg_factor
        L = len(hw)
        self.losses = self.scratch[2 * L:4 * L]
        self.avg_factor = self.scratch[4 * L:4 * L + 1]
        gc, gr = graph.grads[_sptr(graph.cls_buf)], graph.grads[_sptr(graph.reg_buf)]
        d = RetinaLossDesc()
        d.cls, d.reg, d.dcls, d.dreg = graph.cls_buf.data_ptr(), graph.reg_buf.data_ptr(), gc.data_ptr(), gr.data_ptr()
        d.cls_ld, d.reg_ld, d.dcls_ld, d.dreg_ld = graph.cls_buf.shape[-1], graph.reg_buf.shape[-1], gc.shape[-1], gr.shape[-1]
        d.batch, d.n_levels, d.num_anchors, d.num_classes = n, L, A, head.cls_out_channels
        a0 = 0
        for i, (h, w) in enumerate(hw):
            d.level_h[i], d.level_w[i], d.level_off[i], d.level_anchor0[i] = h, w, int(pyr.level_off[i]), a0
            a0 += h * w * A
        d.anchors_per_image = self.a_total
        d.anchors, d.gt_inds = self.flat_anchors.data_ptr(), self.gt_inds.data_ptr()
        lc, lb, tc = head.loss_cls, head.loss_bbox, head.train_cfg
        d.gamma, d.alpha, d.cls_weight, d.reg_weight = lc.gamma, lc.alpha, lc.loss_weight, lb.loss_weight
        d.beta = float(getattr(lb, 'beta', 0.0)) if type(lb).__name__ == 'SmoothL1Loss' else 0.0
        d.pos_weight, d.grad_scale = float(tc['pos_weight']), graph.loss_scale
        for i in range(4):
            d.means[i], d.stds[i] = float(head.bbox_coder.means[i]), float(head.bbox_coder.stds[i])
        d.sums, d.avg_factor, d.losses = self.scratch.data_ptr(), self.avg_factor.data_ptr(), self.losses.data_ptr()
        self.counts = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
        d.counts = self.counts.data_ptr()
        self.partials = torch.zeros(2 * len(hw) * 2048, dtype=torch.float32, device=device)     # a slot per workgroup and term: no atomics
        d.partials, d.partials_cap = self.partials.data_ptr(), self.partials.numel()
        self.desc = d
        self._set_gt_ptrs()
        self._inside = {}
        wsb = lib().llv_iou_assign_workspace_bytes(4096, self.a_total)
        self.ws = torch.empty((max(wsb, 256),), device=device, dtype=torch.uint8)
        self.wsb = wsb

    def _grow(self, k):
        self.max_gt = k
        self.gt_boxes = torch.zeros((self.batch, k, 4), dtype=torch.float32, device=self.device)
        self.gt_labels = torch.zeros((self.batch, k), dtype=torch.int64, device=self.device)

    def _set_gt_ptrs(self):
        self.desc.gt_boxes, self.desc.gt_labels, self.desc.max_gt = self.gt_boxes.data_ptr(), self.gt_labels.data_ptr(), self.max_gt

    def inside_flags(self, meta):
        """valid_flags & anchor_inside_flags of one image geometry (anchor_head.py:145-173, 206-211), cached; raises when no anchor
        lies inside (the module-level path returns None there)."""
        tc = self.head.train_cfg
        key = (tuple(meta['img_shape'][:2]), tuple(meta['pad_shape'][:2]), tc['allowed_border'])
        if key not in self._inside:
            flags = torch.cat(self.head.anchor_generator.valid_flags(self.hw, meta['pad_shape'], self.device))
            inside = anchor_inside_flags(self.flat_anchors, flags, meta['img_shape'][:2], tc['allowed_border'])
            if not bool(inside.any()):
                raise LlvError('no valid anchors inside the image')
            self._inside[key] = inside.to(torch.uint8).contiguous()
        return self._inside[key]

    def run(self, gt_bboxes, gt_labels, img_metas):
        """-> (2 * levels,) fp32 device tensor: loss_cls per level, then loss_bbox per level. The gradient twins of the logits hold
        d(total loss)/d(logit) * loss_scale afterwards."""
        L, a, st = lib(), self.head.assigner, _stream()
        k_max = max(int(b.shape[0]) for b in gt_bboxes)
        if k_max > self.max_gt:
            self._grow(max(k_max, 2 * self.max_gt))
            self._set_gt_ptrs()
        for n, (gb, gl, meta) in enumerate(zip(gt_bboxes, gt_labels, img_metas)):
            k = int(gb.shape[0])
            if k > 4096:
                raise LlvError('more than 4096 ground-truth boxes in one image')
            if k:
                self.gt_boxes[n, :k].copy_(gb, non_blocking=True)
                self.gt_labels[n, :k].copy_(gl, non_blocking=True)
            check(L.llv_iou_assign(ops._ptr(self.gt_boxes[n]), k, ops._ptr(self.flat_anchors), ops._ptr(self.inside_flags(meta)), self.a_total,
                                   a.pos_iou_thr, a.neg_iou_thr, a.min_pos_iou, int(a.match_low_quality), int(a.gt_max_assign_all),
                                   ops._ptr(self.gt_inds[n]), ops._ptr(self.max_ov), ops._ptr(self.ws), self.wsb, st), 'llv_iou_assign')
        import ctypes as C
        check(L.llv_retina_loss(C.byref(self.desc), st), 'llv_retina_loss')
        return self.losses


def _stream():
    import ctypes as C
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _sptr(t):
    return t.untyped_storage().data_ptr()


def single_stage_forward_train(det, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
    """single_stage.py:68-96. The detector body runs as one autograd op on the device kernels (train_engine), so the
    returned losses support `.backward()` and fill `param.grad` exactly as the reference's module tree does."""
    from . import train_engine
    if img.device.type != 'cuda':
        raise LlvError('forward_train runs on the MI355X (conv forward/backward kernels); no CPU fallback')
    n, _, h, w = img.shape
    # parameters are about to change: inference plans hold packed copies of them, drop those (the training graph
    # re-packs from the live parameters on every forward)
    for k in [k for k in det._plans if not (isinstance(k, tuple) and k and k[0] == 'train')]:
        del det._plans[k]
    graph = det._plan(('train', n, h, w, str(img.device)), lambda: train_engine.TrainGraph(det, n, h, w, img.device))
    dev = img.device
    return train_engine.forward_train(det, graph, img, img_metas, [b.to(dev) for b in gt_bboxes], [l.to(dev) for l in gt_labels],
                                      gt_bboxes_ignore)
