"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): fp32 CPU restatement of the YOLOv3 stand-in for config #5.

  Darknet-53              mmdetection/mmdet/models/backbones/darknet.py:14-199
  YOLOV3Neck              mmdetection/mmdet/models/necks/yolo_neck.py:11-136
  YOLOV3Head              mmdetection/mmdet/models/dense_heads/yolo_head.py:150-170 (forward), :210-366 (_get_bboxes)
  YOLOAnchorGenerator     mmdetection/mmdet/core/anchor/anchor_generator.py:590-665 (+ grid_anchors of the base class)
  YOLOBBoxCoder.decode    mmdetection/mmdet/core/bbox/coder/yolo_bbox_coder.py:61-89
ConvModule here = conv (no bias) -> BatchNorm (eval) -> LeakyReLU(0.1) (mmcv, not vendored: semantics restated).
Pinning: YOLOBBoxCoder.decode by the reference's mmdetection/tests/test_utils/test_coder.py:8-23 (four expected boxes) and the
YOLOAnchorGenerator base anchors by tests/test_utils/test_anchor.py:148-188 (tests/test_oracle_known_answers.py runs both through this
file and through the product). The reference holds no vector for Darknet / the neck / _get_bboxes' selection or the grid assigner:
those stay unpinned restatements (F.conv2d / F.batch_norm compositions + the pinned pieces + oracle/nms.py).
"""
import torch
import torch.nn.functional as F

from .nms import multiclass_nms

ARCH = {53: ((1, 2, 8, 8, 4), ((32, 64), (64, 128), (128, 256), (256, 512), (512, 1024)))}   # darknet.py:83-86


def conv_module(x, sd, p, stride=1, pad=0, eps=1e-5, slope=0.1):
    x = F.conv2d(x, sd[p + '.conv.weight'], None, stride, pad)
    x = F.batch_norm(x, sd[p + '.bn.running_mean'], sd[p + '.bn.running_var'], sd[p + '.bn.weight'], sd[p + '.bn.bias'], False, 0.0, eps)
    return F.leaky_relu(x, slope)


def darknet_forward(sd, x, depth=53, out_indices=(3, 4, 5), prefix='backbone.'):
    layers, _ = ARCH[depth]
    outs = []
    x = conv_module(x, sd, prefix + 'conv1', 1, 1)
    if 0 in out_indices:
        outs.append(x)
    for i, n in enumerate(layers):
        q = f'{prefix}conv_res_block{i + 1}.'
        x = conv_module(x, sd, q + 'conv', 2, 1)                       # darknet.py:189-193
        for r in range(n):                                              # ResBlock, darknet.py:47-53
            y = conv_module(x, sd, f'{q}res{r}.conv1', 1, 0)
            y = conv_module(y, sd, f'{q}res{r}.conv2', 1, 1)
            x = y + x
        if i + 1 in out_indices:
            outs.append(x)
    return tuple(outs)


def detection_block(x, sd, p):                                          # yolo_neck.py:47-53
    x = conv_module(x, sd, p + 'conv1', 1, 0)
    x = conv_module(x, sd, p + 'conv2', 1, 1)
    x = conv_module(x, sd, p + 'conv3', 1, 0)
    x = conv_module(x, sd, p + 'conv4', 1, 1)
    return conv_module(x, sd, p + 'conv5', 1, 0)


def yolo_neck_forward(sd, feats, prefix='neck.'):                       # yolo_neck.py:110-131
    outs = []
    out = detection_block(feats[-1], sd, prefix + 'detect1.')
    outs.append(out)
    for i, x in enumerate(reversed(feats[:-1])):
        tmp = conv_module(out, sd, f'{prefix}conv{i + 1}', 1, 0)
        tmp = F.interpolate(tmp, scale_factor=2)
        tmp = torch.cat((tmp, x), 1)
        out = detection_block(tmp, sd, f'{prefix}detect{i + 2}.')
        outs.append(out)
    return tuple(outs)


def yolo_head_forward(sd, feats, prefix='bbox_head.'):                  # yolo_head.py:150-170
    maps = []
    for i, x in enumerate(feats):
        x = conv_module(x, sd, f'{prefix}convs_bridge.{i}', 1, 1)
        maps.append(F.conv2d(x, sd[f'{prefix}convs_pred.{i}.weight'], sd[f'{prefix}convs_pred.{i}.bias']))
    return maps


def yolo_grid_anchors(featmap_sizes, base_sizes, strides):
    """YOLOAnchorGenerator: base anchors centred at (stride/2, stride/2) (anchor_generator.py:639-665), shifted over the
    grid in (y, x, anchor) order (single_level_grid_anchors, anchor_generator.py:237-270)."""
    out = []
    for (h, w), sizes, st in zip(featmap_sizes, base_sizes, strides):
        xc = yc = st / 2.
        base = torch.stack([torch.Tensor([xc - 0.5 * bw, yc - 0.5 * bh, xc + 0.5 * bw, yc + 0.5 * bh]) for bw, bh in sizes], 0)
        sx = torch.arange(0, w) * st
        sy = torch.arange(0, h) * st
        xx = sx.repeat(len(sy))
        yy = sy.view(-1, 1).repeat(1, len(sx)).view(-1)
        shifts = torch.stack([xx, yy, xx, yy], dim=-1).type_as(base)
        out.append((base[None, :, :] + shifts[:, None, :]).view(-1, 4))
    return out


def yolo_decode(bboxes, pred, stride):                                  # yolo_bbox_coder.py:61-89
    xc = (bboxes[..., 0] + bboxes[..., 2]) * 0.5
    yc = (bboxes[..., 1] + bboxes[..., 3]) * 0.5
    w = bboxes[..., 2] - bboxes[..., 0]
    h = bboxes[..., 3] - bboxes[..., 1]
    px = (pred[..., 0] - 0.5) * stride + xc
    py = (pred[..., 1] - 0.5) * stride + yc
    pw = torch.exp(pred[..., 2]) * w
    ph = torch.exp(pred[..., 3]) * h
    return torch.stack((px - pw / 2, py - ph / 2, px + pw / 2, py + ph / 2), dim=-1)


def yolo_get_bboxes(pred_maps, base_sizes, strides, num_classes, test_cfg, scale_factors=None, rescale=False, return_inds=False):
    """YOLOV3Head._get_bboxes (yolo_head.py:210-366). Returns per image (dets (k,5), labels (k,))."""
    na = 5 + num_classes
    bs = pred_maps[0].shape[0]
    anchors = yolo_grid_anchors([m.shape[-2:] for m in pred_maps], base_sizes, strides)
    nms_pre = test_cfg.get('nms_pre', -1)
    boxes_l, cls_l, conf_l = [], [], []
    for i, pm in enumerate(pred_maps):
        pm = pm.permute(0, 2, 3, 1).reshape(bs, -1, na)
        pm = torch.cat([torch.sigmoid(pm[..., :2]), pm[..., 2:]], dim=-1)
        box = yolo_decode(anchors[i].expand_as(pm[..., :4]), pm[..., :4], strides[i])
        conf = torch.sigmoid(pm[..., 4])
        cls = torch.sigmoid(pm[..., 5:]).view(bs, -1, num_classes)
        if nms_pre > 0 and conf.shape[1] > nms_pre:
            _, ti = conf.topk(nms_pre)
            bi = torch.arange(bs).view(-1, 1).expand_as(ti).long()
            box, cls, conf = box[bi, ti, :], cls[bi, ti, :], conf[bi, ti]
        boxes_l.append(box); cls_l.append(cls); conf_l.append(conf)
    boxes, scores, confs = torch.cat(boxes_l, 1), torch.cat(cls_l, 1), torch.cat(conf_l, 1)
    if rescale:
        boxes = boxes / boxes.new_tensor(scale_factors).unsqueeze(1)
    scores = torch.cat([scores, scores.new_zeros(bs, scores.shape[1], 1)], dim=-1)
    out = []
    for b, s, c in zip(boxes, scores, confs):
        thr = test_cfg.get('conf_thr', -1)
        if thr > 0:
            ci = c.ge(thr).nonzero(as_tuple=False).squeeze(1)
            b, s, c = b[ci, :], s[ci, :], c[ci]
        out.append(multiclass_nms(b, s, test_cfg['score_thr'], test_cfg['nms'], test_cfg['max_per_img'], score_factors=c))
    return out


def yolov3_simple_test(sd, img, num_classes, base_sizes, strides, test_cfg, scale_factors=None, rescale=False):
    feats = darknet_forward(sd, img)
    feats = yolo_neck_forward(sd, feats)
    return yolo_get_bboxes(yolo_head_forward(sd, feats), base_sizes, strides, num_classes, test_cfg, scale_factors, rescale)


# ---------------------------------------------------------------------------------------------------------------------
# training targets and losses of YOLOV3Head (yolo_head.py:369-560), restated on the CPU
# ---------------------------------------------------------------------------------------------------------------------
def yolo_responsible_flags(featmap_sizes, gt_bboxes, strides, num_base_anchors):
    """YOLOAnchorGenerator.responsible_flags (anchor_generator.py:667-728): the anchors of the grid cell a ground-truth centre falls in."""
    out = []
    for (h, w), st, nb in zip(featmap_sizes, strides, num_base_anchors):
        cx = (gt_bboxes[:, 0] + gt_bboxes[:, 2]) * 0.5
        cy = (gt_bboxes[:, 1] + gt_bboxes[:, 3]) * 0.5
        gx = torch.floor(cx / st).long()
        gy = torch.floor(cy / st).long()
        grid = torch.zeros(h * w, dtype=torch.uint8)
        grid[gy * w + gx] = 1
        out.append(grid[:, None].expand(h * w, nb).contiguous().view(-1))
    return out


def grid_assign(bboxes, flags, gt_bboxes, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0, gt_max_assign_all=True):
    """GridAssigner.assign (grid_assigner.py:42-155): 1-based gt index per box, 0 negative, -1 ignored."""
    from .bbox import bbox_overlaps
    k, n = gt_bboxes.size(0), bboxes.size(0)
    overlaps = bbox_overlaps(gt_bboxes, bboxes)
    assigned = overlaps.new_full((n,), -1, dtype=torch.long)
    if k == 0 or n == 0:
        if k == 0:
            assigned[:] = 0
        return assigned
    max_ov, _ = overlaps.max(dim=0)
    assigned[(max_ov >= 0) & (max_ov <= neg_iou_thr)] = 0
    fl = flags.type(torch.bool)
    overlaps[:, ~fl] = -1.
    max_ov, argmax_ov = overlaps.max(dim=0)
    gt_max, gt_argmax = overlaps.max(dim=1)
    pos = (max_ov > pos_iou_thr) & fl
    assigned[pos] = argmax_ov[pos] + 1
    for i in range(k):
        if gt_max[i] > min_pos_iou:
            if gt_max_assign_all:
                assigned[(overlaps[i, :] == gt_max[i]) & fl] = i + 1
            elif fl[gt_argmax[i]]:
                assigned[gt_argmax[i]] = i + 1
    return assigned


def yolo_encode(bboxes, gt_bboxes, stride, eps=1e-6):
    """YOLOBBoxCoder.encode (yolo_bbox_coder.py:26-58)."""
    xg = (gt_bboxes[..., 0] + gt_bboxes[..., 2]) * 0.5
    yg = (gt_bboxes[..., 1] + gt_bboxes[..., 3]) * 0.5
    wg = gt_bboxes[..., 2] - gt_bboxes[..., 0]
    hg = gt_bboxes[..., 3] - gt_bboxes[..., 1]
    xc = (bboxes[..., 0] + bboxes[..., 2]) * 0.5
    yc = (bboxes[..., 1] + bboxes[..., 3]) * 0.5
    w = bboxes[..., 2] - bboxes[..., 0]
    h = bboxes[..., 3] - bboxes[..., 1]
    wt = torch.log((wg / w).clamp(min=eps))
    ht = torch.log((hg / h).clamp(min=eps))
    xt = ((xg - xc) / stride + 0.5).clamp(eps, 1 - eps)
    yt = ((yg - yc) / stride + 0.5).clamp(eps, 1 - eps)
    return torch.stack([xt, yt, wt, ht], dim=-1)


def yolo_loss(pred_maps, gt_bboxes, gt_labels, base_sizes, strides, num_classes, pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.0,
              one_hot_smoother=0.0, w_cls=1.0, w_conf=1.0, w_xy=2.0, w_wh=2.0):
    """YOLOV3Head.loss (yolo_head.py:369-462) with the losses of configs/yolo/yolov3_d53_mstrain-608_273e_coco.py:27-60: sigmoid cross
    entropy for classes / objectness / centre offsets, MSE for log sizes, all reduction='sum'. Returns the four per-level lists."""
    na = 5 + num_classes
    sizes = [tuple(p.shape[-2:]) for p in pred_maps]
    anchors = yolo_grid_anchors(sizes, base_sizes, strides)
    nb = [len(b) for b in base_sizes]
    nlev = [a.size(0) for a in anchors]
    targets, negs = [], []
    for img in range(pred_maps[0].shape[0]):
        flags = torch.cat(yolo_responsible_flags(sizes, gt_bboxes[img], strides, nb))
        cat = torch.cat(anchors)
        st = torch.cat([torch.full((a.size(0),), float(s)) for a, s in zip(anchors, strides)])
        assigned = grid_assign(cat, flags, gt_bboxes[img], pos_iou_thr, neg_iou_thr, min_pos_iou)
        pos = torch.nonzero(assigned > 0, as_tuple=False).squeeze(-1).unique()
        neg = torch.nonzero(assigned == 0, as_tuple=False).squeeze(-1).unique()
        tm = cat.new_zeros(cat.size(0), na)
        gi = assigned[pos] - 1
        tm[pos, :4] = yolo_encode(cat[pos], gt_bboxes[img][gi], st[pos])
        tm[pos, 4] = 1
        oh = F.one_hot(gt_labels[img], num_classes=num_classes).float()
        if one_hot_smoother != 0:
            oh = oh * (1 - one_hot_smoother) + one_hot_smoother / num_classes
        tm[pos, 5:] = oh[gi]
        nm = cat.new_zeros(cat.size(0), dtype=torch.uint8)
        nm[neg] = 1
        targets.append(tm); negs.append(nm)
    out = dict(loss_cls=[], loss_conf=[], loss_xy=[], loss_wh=[])
    o = 0
    for l, pm in enumerate(pred_maps):
        n = pm.shape[0]
        tmap = torch.stack([t[o:o + nlev[l]] for t in targets])            # images_to_levels
        nmap = torch.stack([t[o:o + nlev[l]] for t in negs])
        o += nlev[l]
        p = pm.permute(0, 2, 3, 1).reshape(n, -1, na)
        pos_mask = tmap[..., 4]
        pn = (nmap.float() + pos_mask).clamp(min=0., max=1.)
        pm1 = pos_mask.unsqueeze(-1)
        bce = lambda a, b: F.binary_cross_entropy_with_logits(a, b, reduction='none')
        out['loss_cls'].append(w_cls * (bce(p[..., 5:], tmap[..., 5:]) * pm1).sum())
        out['loss_conf'].append(w_conf * (bce(p[..., 4], tmap[..., 4]) * pn).sum())
        out['loss_xy'].append(w_xy * (bce(p[..., :2], tmap[..., :2]) * pm1).sum())
        out['loss_wh'].append(w_wh * (F.mse_loss(p[..., 2:4], tmap[..., 2:4], reduction='none') * pm1).sum())
    return out
