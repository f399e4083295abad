"""RetinaNet R-50+FPN inference restated as pure functions of a reference-named state_dict (fp32, CPU).
Test infrastructure only.

Follows, operation by operation:
  ResNet.forward / Bottleneck.forward / BasicBlock.forward  mmdetection/mmdet/models/backbones/resnet.py:622-640,260-299,66-92
  ResLayer (downsample = conv1x1(stride) + BN)              mmdetection/mmdet/models/utils/res_layer.py:36-58
  FPN.forward                                               mmdetection/mmdet/models/necks/fpn.py:170-221
  RetinaHead.forward_single                                 mmdetection/mmdet/models/dense_heads/retina_head.py:93-114
  AnchorHead.get_bboxes/_get_bboxes                         mmdetection/mmdet/models/dense_heads/anchor_head.py:495-726
  SingleStageDetector.simple_test                           mmdetection/mmdet/models/detectors/single_stage.py:98-129
BatchNorm uses running statistics (norm_eval=True: resnet.py:642-651), eps = 1e-5.
"""
import torch
import torch.nn.functional as F

from .anchors import AnchorGenerator
from .bbox import DeltaXYWHBBoxCoder, bbox2result
from .nms import multiclass_nms

ARCH = {18: ('basic', (2, 2, 2, 2)), 34: ('basic', (3, 4, 6, 3)), 50: ('bottleneck', (3, 4, 6, 3)),
        101: ('bottleneck', (3, 4, 23, 3)), 152: ('bottleneck', (3, 8, 36, 3))}  # resnet.py:354-360


def _bn(x, sd, p, eps=1e-5):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        False, 0.0, eps)


def resnet_forward(sd, x, depth=50, out_indices=(0, 1, 2, 3), strides=(1, 2, 2, 2), style='pytorch', prefix='backbone.',
                   after_stage=None, dilations=(1, 1, 1, 1), dark_stages=(), num_stages=4):
    """after_stage(i, num_blocks, x) -> x: the plugin InsertResNet appends to stage i
    (mmtracking/mmtrack/models/backbones/insert_resnet.py:93-97); an out index may repeat (:410-415)."""
    kind, blocks = ARCH[depth]
    blocks = blocks[:num_stages]
    p = prefix
    x = F.conv2d(x, sd[p + 'conv1.weight'], None, 2, 3)          # resnet.py:559-566 (7x7/s2, bias=False)
    x = F.relu(_bn(x, sd, p + 'bn1'))
    x = F.max_pool2d(x, 3, 2, 1)                                 # resnet.py:570
    outs = []
    for i, nb in enumerate(blocks):
        for b in range(nb):
            q = f'{p}layer{i + 1}.{b}.'
            stride = strides[i] if b == 0 else 1
            dil = dilations[i]                         # resnet.py:151-167: conv2 (Bottleneck) / conv1 (BasicBlock) padding = dilation
            identity = x
            if kind == 'bottleneck' and i in dark_stages:
                # DenoisingBlock of DarkResNet (mmtracking/mmtrack/models/backbones/dark_resnet.py:226-285): the 3x3 conv is the gate
                # convolution of a ConvLSTM along the batch; all four gates through a sigmoid (:237-240); h_0 = c_0 = 0
                s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)
                out = F.relu(_bn(F.conv2d(x, sd[q + 'conv1.weight'], None, s1, 0), sd, q + 'bn1'))
                hs, h_cur, c_cur = [], None, None
                for t in range(out.shape[0]):
                    xt = out[t:t + 1]
                    if h_cur is None:
                        h_cur, c_cur = torch.zeros_like(xt), torch.zeros_like(xt)
                    cc = F.conv2d(torch.cat([xt, h_cur], dim=1), sd[q + 'conv2.weight'], None, s2, dil, dil)
                    ci, cf, co, cg = torch.split(cc, xt.shape[1], dim=1)
                    c_cur = torch.sigmoid(cf) * c_cur + torch.sigmoid(ci) * torch.sigmoid(cg)
                    h_cur = torch.sigmoid(co) * torch.tanh(c_cur)
                    hs.append(h_cur)
                out = F.relu(_bn(torch.cat(hs, 0), sd, q + 'bn2'))
                out = _bn(F.conv2d(out, sd[q + 'conv3.weight'], None, 1, 0), sd, q + 'bn3')
            elif kind == 'bottleneck':
                s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)    # resnet.py:151-156
                out = F.relu(_bn(F.conv2d(x, sd[q + 'conv1.weight'], None, s1, 0), sd, q + 'bn1'))
                out = F.relu(_bn(F.conv2d(out, sd[q + 'conv2.weight'], None, s2, dil, dil), sd, q + 'bn2'))
                out = _bn(F.conv2d(out, sd[q + 'conv3.weight'], None, 1, 0), sd, q + 'bn3')
            else:
                out = F.relu(_bn(F.conv2d(x, sd[q + 'conv1.weight'], None, stride, dil, dil), sd, q + 'bn1'))
                out = _bn(F.conv2d(out, sd[q + 'conv2.weight'], None, 1, 1), sd, q + 'bn2')
            if (q + 'downsample.0.weight') in sd:                                # res_layer.py:36-58
                identity = _bn(F.conv2d(x, sd[q + 'downsample.0.weight'], None, stride, 0), sd, q + 'downsample.1')
            x = F.relu(out + identity)
        if after_stage is not None:
            x = after_stage(i, nb, x)
        for j in out_indices:
            if i == j:
                outs.append(x)
    return tuple(outs)


def fpn_forward(sd, inputs, start_level=1, num_outs=5, add_extra_convs='on_input', relu_before_extra_convs=False,
                prefix='neck.'):
    p = prefix
    n_lat = len(inputs) - start_level
    laterals = [F.conv2d(inputs[i + start_level], sd[f'{p}lateral_convs.{i}.conv.weight'],
                         sd[f'{p}lateral_convs.{i}.conv.bias']) for i in range(n_lat)]
    for i in range(n_lat - 1, 0, -1):                                            # fpn.py:180-191
        laterals[i - 1] = laterals[i - 1] + F.interpolate(laterals[i], size=laterals[i - 1].shape[2:], mode='nearest')
    outs = [F.conv2d(laterals[i], sd[f'{p}fpn_convs.{i}.conv.weight'], sd[f'{p}fpn_convs.{i}.conv.bias'], 1, 1)
            for i in range(n_lat)]
    if num_outs > len(outs):                                                     # fpn.py:199-220
        if not add_extra_convs:
            for _ in range(num_outs - n_lat):
                outs.append(F.max_pool2d(outs[-1], 1, stride=2))
        else:
            if add_extra_convs == 'on_input':
                src = inputs[-1]
            elif add_extra_convs == 'on_lateral':
                src = laterals[-1]
            else:
                src = outs[-1]
            outs.append(F.conv2d(src, sd[f'{p}fpn_convs.{n_lat}.conv.weight'], sd[f'{p}fpn_convs.{n_lat}.conv.bias'], 2, 1))
            for i in range(n_lat + 1, num_outs):
                x = F.relu(outs[-1]) if relu_before_extra_convs else outs[-1]
                outs.append(F.conv2d(x, sd[f'{p}fpn_convs.{i}.conv.weight'], sd[f'{p}fpn_convs.{i}.conv.bias'], 2, 1))
    return tuple(outs)


def retina_head_forward(sd, feats, stacked_convs=4, prefix='bbox_head.'):
    p = prefix
    cls_scores, bbox_preds = [], []
    for x in feats:                                                              # retina_head.py:93-114
        c, r = x, x
        for i in range(stacked_convs):
            c = F.relu(F.conv2d(c, sd[f'{p}cls_convs.{i}.conv.weight'], sd[f'{p}cls_convs.{i}.conv.bias'], 1, 1))
            r = F.relu(F.conv2d(r, sd[f'{p}reg_convs.{i}.conv.weight'], sd[f'{p}reg_convs.{i}.conv.bias'], 1, 1))
        cls_scores.append(F.conv2d(c, sd[p + 'retina_cls.weight'], sd[p + 'retina_cls.bias'], 1, 1))
        bbox_preds.append(F.conv2d(r, sd[p + 'retina_reg.weight'], sd[p + 'retina_reg.bias'], 1, 1))
    return cls_scores, bbox_preds


def decode_candidates(cls_scores, bbox_preds, anchor_generator, bbox_coder, img_shape, scale_factor, num_classes,
                      nms_pre=1000, rescale=False):
    """anchor_head.py:636-711 for ONE image (batch 1), use_sigmoid_cls. Returns (boxes (n,4), scores (n,C+1),
    anchor index per candidate). top-k uses a stable sort = (score desc, anchor asc)."""
    featmap_sizes = [c.shape[-2:] for c in cls_scores]
    mlvl_anchors = anchor_generator.grid_anchors(featmap_sizes)
    mlvl_bboxes, mlvl_scores, mlvl_idx = [], [], []
    a0 = 0
    for cls_score, bbox_pred, anchors in zip(cls_scores, bbox_preds, mlvl_anchors):
        cls_score = cls_score.permute(0, 2, 3, 1).reshape(1, -1, num_classes)
        scores = cls_score.sigmoid()
        bbox_pred = bbox_pred.permute(0, 2, 3, 1).reshape(1, -1, 4)
        anchors = anchors.expand_as(bbox_pred)
        idx = torch.arange(scores.shape[1])
        if nms_pre > 0 and scores.shape[-2] > nms_pre:
            max_scores, _ = scores.max(-1)
            _, topk_inds = torch.sort(max_scores[0], descending=True, stable=True)
            topk_inds = topk_inds[:nms_pre]
            anchors = anchors[:, topk_inds, :]
            bbox_pred = bbox_pred[:, topk_inds, :]
            scores = scores[:, topk_inds, :]
            idx = topk_inds
        bboxes = bbox_coder.decode(anchors, bbox_pred, max_shape=[img_shape])
        mlvl_bboxes.append(bboxes)
        mlvl_scores.append(scores)
        mlvl_idx.append(idx + a0)
        a0 += cls_score.shape[1]
    boxes = torch.cat(mlvl_bboxes, dim=1)
    if rescale:
        boxes = boxes / boxes.new_tensor(scale_factor).unsqueeze(0).unsqueeze(1)
    scores = torch.cat(mlvl_scores, dim=1)
    padding = scores.new_zeros(1, scores.shape[1], 1)
    scores = torch.cat([scores, padding], dim=-1)
    return boxes[0], scores[0], torch.cat(mlvl_idx)


def get_bboxes(cls_scores, bbox_preds, anchor_generator, bbox_coder, img_shape, scale_factor, num_classes, test_cfg,
               rescale=False, return_inds=False):
    boxes, scores, _ = decode_candidates(cls_scores, bbox_preds, anchor_generator, bbox_coder, img_shape, scale_factor,
                                         num_classes, test_cfg.get('nms_pre', -1), rescale)
    return multiclass_nms(boxes, scores, test_cfg['score_thr'], test_cfg['nms'], test_cfg['max_per_img'],
                          return_inds=return_inds)


DEFAULT_ANCHOR = dict(octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0], strides=[8, 16, 32, 64, 128])
DEFAULT_TEST_CFG = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5),
                        max_per_img=100)   # mmdetection/configs/_base_/models/retinanet_r50_fpn.py:54-60


def retinanet_extract_feat(sd, img, depth=50):
    return fpn_forward(sd, resnet_forward(sd, img, depth))


def retinanet_simple_test(sd, img, img_meta, num_classes=80, depth=50, test_cfg=None, rescale=False,
                          anchor_cfg=None, stacked_convs=4):
    """single_stage.py:98-129 for one image -> list[num_classes] of (k,5) float32 arrays."""
    test_cfg = test_cfg or DEFAULT_TEST_CFG
    ag = AnchorGenerator(**(anchor_cfg or DEFAULT_ANCHOR))
    coder = DeltaXYWHBBoxCoder()
    feats = retinanet_extract_feat(sd, img, depth)
    cls_scores, bbox_preds = retina_head_forward(sd, feats, stacked_convs)
    dets, labels = get_bboxes(cls_scores, bbox_preds, ag, coder, img_meta['img_shape'], img_meta['scale_factor'],
                              num_classes, test_cfg, rescale)
    return bbox2result(dets, labels, num_classes)
