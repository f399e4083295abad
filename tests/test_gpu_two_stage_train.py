"""Training of the two-stage SELSA video detectors: SELSA.forward_train (selsa.py:47-170) and the authors'
SelsaFastDVDnetDetect.forward_train (selsa_fastdvd_darkfarm.py:65-227) on the device against autograd over the fp32 oracle composition
(oracle/two_stage.py): every loss term and d(loss)/d(parameter) of backbone, ChannelMapper, RPN and RoI head.

Both sides see the SAME proposals (the oracle's, put in place of RPNHead.get_bboxes / simple_test_rpn on the device): the proposal
attention makes every RoI logit depend on the exact proposal set, and the device RPN has its own parity test (tests/test_rpn.py).
The samplers' `num` is sized so that every candidate is taken (RandomSampler draws with randperm otherwise: oracle/two_stage.py)."""
import pytest
import torch

from helpers import grad_metrics, img_meta, randomize, smooth_image

pytestmark = pytest.mark.gpu

H, W = 128, 192
TRAIN_CFG = dict(
    rpn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, ignore_iof_thr=-1),
             sampler=dict(type='RandomSampler', num=4096, pos_fraction=0.5, neg_pos_ub=-1, add_gt_as_proposals=False),
             allowed_border=0, pos_weight=-1, debug=False),
    rpn_proposal=dict(nms_across_levels=False, nms_pre=600, nms_post=60, max_num=60, nms_thr=0.7, min_bbox_size=0),
    rcnn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, ignore_iof_thr=-1),
              sampler=dict(type='RandomSampler', num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True),
              pos_weight=-1, debug=False))
GTB = [torch.tensor([[10., 12., 90., 100.], [60., 30., 150., 120.], [100., 8., 180., 60.]])]
GTL = [torch.tensor([1, 3, 0])]


def _seed_heads(model, g):
    with torch.no_grad():
        model.detector.rpn_head.rpn_cls.bias.add_(-1.0)
        model.detector.rpn_head.rpn_reg.weight.mul_(0.3)
        for n, p in model.detector.roi_head.named_parameters():
            if p.dim() == 4:
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / (9 * p.shape[1])) ** 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * ((1.5 / p.shape[1]) ** 0.5 if p.dim() == 2 else 0.1))
        model.detector.roi_head.bbox_head.fc_reg.weight.mul_(0.3)


def _oracle_step(sd, frames, key, depth_kw, p_ext, num_fcs, ncls):
    """frames (n, 3, H, W) -> losses (dict of scalars), proposals of the key frame and of the other frames."""
    from oracle.anchors import AnchorGenerator
    from oracle.bbox import DeltaXYWHBBoxCoder
    from oracle.detector import resnet_forward
    from oracle.rpn import channel_mapper_forward, rpn_get_bboxes, rpn_head_forward
    from oracle.training import MaxIoUAssigner
    from oracle.two_stage import RandomSampler, rpn_loss, selsa_roi_head_forward_train
    ag = AnchorGenerator(strides=[16], ratios=[0.5, 1.0, 2.0], scales=[4, 8, 16, 32])
    rpn_cfg = dict(nms_pre=600, max_per_img=60, nms=dict(type='nms', iou_threshold=0.7), min_bbox_size=0)
    f = channel_mapper_forward(sd, resnet_forward(sd, frames, prefix='detector.backbone.', **depth_kw)[-1:], 'detector.neck.')[0]
    others = [i for i in range(frames.shape[0]) if i != key]
    x, ref_x = f[[key]], f[others]
    cls, reg = rpn_head_forward(sd, [x], 'detector.rpn_head.')
    meta = img_meta(H, W)
    losses = rpn_loss(cls, reg, GTB, [meta], ag, MaxIoUAssigner(0.7, 0.3, 0.3, ignore_iof_thr=-1), RandomSampler(4096, 0.5, -1, False), allowed_border=0)
    losses = dict(loss_rpn_cls=sum(losses['loss_rpn_cls']), loss_rpn_bbox=sum(losses['loss_rpn_bbox']))

    def props(c, r):
        with torch.no_grad():
            return rpn_get_bboxes([t.detach() for t in c], [t.detach() for t in r], ag, DeltaXYWHBBoxCoder(), (H, W, 3), rpn_cfg)
    p_key = props(cls, reg)
    p_ref = [props(*rpn_head_forward(sd, [ref_x[i:i + 1]], 'detector.rpn_head.')) for i in range(len(others))]
    roi_losses, aux = selsa_roi_head_forward_train(sd, 'detector.roi_head.bbox_head.', x, ref_x, p_key, p_ref, GTB[0], GTL[0],
                                                   MaxIoUAssigner(0.5, 0.5, 0.5, ignore_iof_thr=-1), RandomSampler(512, 0.25, -1, True), ncls,
                                                   16, num_fcs, 8, 'smooth_l1', p_ext, 4)      # bbox_head.py:35 default SmoothL1Loss(beta=1.0)
    losses.update(roi_losses)
    return losses, p_key, p_ref, aux


def _compare(model, dev, run, ref_losses, ref_grads, p_key, p_ref, frozen_prefixes, loss_tol, grad_tol, cos_tol=0.997):
    rpn = model.detector.rpn_head
    rpn.get_bboxes = lambda *a, **k: [p_key.to(dev)]
    rpn.simple_test_rpn = lambda *a, **k: [p.to(dev) for p in p_ref]
    losses = run()
    for k, v in ref_losses.items():
        got = losses[k]
        got = sum(got) if isinstance(got, (list, tuple)) else got
        print(f'{k}: {float(got):.5f} vs {float(v):.5f}')
        assert abs(float(got) - float(v)) <= loss_tol * max(abs(float(v)), 1e-2), (k, float(got), float(v))
    total = sum(sum(v) if isinstance(v, (list, tuple)) else v for k, v in losses.items() if 'loss' in k and 'denoising' not in k)
    total.backward()
    params = dict(model.named_parameters())
    top = max(v.abs().max().item() for v in ref_grads.values() if v is not None)
    errs, checked = [], 0
    for k, r in ref_grads.items():
        if k.startswith(frozen_prefixes):
            assert params[k].grad is None, k
            continue
        if r is None:                                   # not on the path (a stage behind the last out index): none or all zero
            assert params[k].grad is None or params[k].grad.abs().max().item() == 0, k
            continue
        g = params[k].grad
        assert g is not None, k
        scale = r.abs().max().item()
        if scale < 1e-6 * top:                          # e.g. ref_fc_embed.bias: the softmax over the references cancels it exactly
            assert g.abs().max().item() < 1e-4 * top, k
            continue
        e_, cos, l2, _ = grad_metrics(g, r)
        errs.append((e_, k, scale, cos, l2))
        checked += 1
    errs.sort(reverse=True)
    print('largest relative gradient errors:', [(round(e, 3), k, round(sc, 5)) for e, k, sc, _, _ in errs[:8]])
    print('lowest cosines:', [(k, round(c, 5), round(l2, 3)) for _, k, _, c, l2 in sorted(errs, key=lambda t: t[3])[:6]])
    for err, k, scale, cos, l2 in errs:
        big = scale > 1e-4 * top
        assert err < (grad_tol if big else 2 * grad_tol), (k, err, scale)
        # cosine / relative L2 beside the max-abs bound: a mis-routed gradient shows up as a cosine far from 1
        assert cos > (cos_tol if big else 1 - 10 * (1 - cos_tol)) and l2 < (1.2 * grad_tol if big else 3 * grad_tol), (k, cos, l2, scale)
    return checked


def test_selsa_forward_train_matches_oracle(dev):
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    from test_selsa import NCLS, _selsa_cfg
    cfg = _selsa_cfg()
    cfg['detector']['train_cfg'] = TRAIN_CFG
    model = M.build_model(cfg)
    randomize(model.detector.backbone, seed=171)
    randomize(model.detector.neck, seed=172)
    randomize(model.detector.rpn_head, seed=173)
    _seed_heads(model, torch.Generator().manual_seed(174))
    NREF = 2
    frames = smooth_image(NREF + 1, H, W, seed=175)
    sd = {k: v.detach().clone().float() for k, v in model.state_dict().items()}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    for n in names:
        sd[n].requires_grad_(True)
    ref_losses, p_key, p_ref, _ = _oracle_step(sd, frames, 0, dict(depth=18, out_indices=(2,)), None, 2, NCLS)
    sum(v for k, v in ref_losses.items() if 'loss' in k).backward()
    ref_grads = {n: sd[n].grad for n in names}
    model = model.to(dev).train()
    meta = img_meta(H, W)

    def run():
        return model(img=frames[:1].to(dev), img_metas=[meta], gt_bboxes=[b.to(dev) for b in GTB], gt_labels=[l.to(dev) for l in GTL],
                     ref_img=frames[1:].unsqueeze(0).to(dev), ref_img_metas=[[meta] * NREF], ref_gt_bboxes=None, ref_gt_labels=None, return_loss=True)
    checked = _compare(model, dev, run, ref_losses, ref_grads, p_key, p_ref, ('cleaner.',), 2e-2, 8e-2)
    assert checked > 50, checked
    # layer4 is not on the path of out_indices=(2,): no gradient on either side; the RoI head and the RPN get theirs
    p = dict(model.named_parameters())
    for k in ('detector.roi_head.bbox_head.aggregator.0.fc_embed.weight', 'detector.rpn_head.rpn_cls.weight', 'detector.neck.convs.0.conv.weight',
              'detector.backbone.layer1.0.conv1.weight'):
        assert p[k].grad is not None and p[k].grad.abs().max().item() > 0, k


def test_selsa_fastdvdnet_detect_forward_train_matches_oracle(dev):
    """The authors' model: noisy ‖ clean 6-channel frames, cyclic 5-stacks through the (frozen) FastDVDnet, DC5 backbone, TemporalRoIAlign.
    TemporalRoIAlign's top-2 picks flip where two reference pixels tie within fp16 rounding of the feature maps, so the RoI-side
    tolerances are wider than SELSA's."""
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    import torch.nn.functional as F
    from oracle.cleaner import fastdvdnet_forward
    from test_selsa import NCLS, _authors_cfg
    cfg = _authors_cfg()
    cfg['detector']['train_cfg'] = TRAIN_CFG
    model = M.build_model(cfg)
    randomize(model.cleaner, seed=191)
    randomize(model.detector.backbone, seed=192)
    randomize(model.detector.neck, seed=193)
    randomize(model.detector.rpn_head, seed=194)
    g = torch.Generator().manual_seed(195)
    _seed_heads(model, g)
    clean = smooth_image(5, H, W, seed=196)
    noisy = clean + 0.2 * torch.randn(5, 3, H, W, generator=g)
    both = torch.cat([noisy, clean], 1)                                     # (5, 6, H, W): frames [ref0, ref1, key, ref2, ref3]
    img, ref_img = both[[2]], both[[0, 1, 3, 4]].unsqueeze(0)
    sd = {k: v.detach().clone().float() for k, v in model.state_dict().items()}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert not any(n.startswith('cleaner.') for n in names)                 # frozen_modules='cleaner'
    for n in names:
        sd[n].requires_grad_(True)
    sd_c = {k[len('cleaner.'):]: v for k, v in sd.items() if k.startswith('cleaner.')}
    stacks = torch.cat([torch.cat([noisy[[(i + j) % 5]] for j in (-2, -1, 0, 1, 2)], 1) for i in range(5)], 0)
    with torch.no_grad():
        den = fastdvdnet_forward(sd_c, stacks)
    ref_losses, p_key, p_ref, _ = _oracle_step(sd, den, 2, dict(depth=18, out_indices=(3,), strides=(1, 2, 2, 1), dilations=(1, 1, 1, 2)),
                                               'detector.roi_head.bbox_roi_extractor.', 3, NCLS)
    sum(v for k, v in ref_losses.items() if 'loss' in k).backward()
    ref_grads = {n: sd[n].grad for n in names}
    ref_losses['loss_denoising_l2'] = F.mse_loss(den, clean)
    model = model.to(dev).train()
    meta = img_meta(H, W)

    def run():
        return model(img=img.to(dev), img_metas=[meta], gt_bboxes=[b.to(dev) for b in GTB], gt_labels=[l.to(dev) for l in GTL],
                     ref_img=ref_img.to(dev), ref_img_metas=[[meta] * 4], ref_gt_bboxes=None, ref_gt_labels=None, return_loss=True)
    checked = _compare(model, dev, run, ref_losses, ref_grads, p_key, p_ref, ('cleaner.',), 5e-2, 2e-1, cos_tol=0.98)
    assert checked > 50, checked
    p = dict(model.named_parameters())
    assert p['detector.roi_head.bbox_roi_extractor.embed_network.conv.weight'].grad is not None
    assert all(q.grad is None for n, q in p.items() if n.startswith('cleaner.'))
