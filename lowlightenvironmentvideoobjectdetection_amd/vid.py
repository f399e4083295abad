"""Video detectors with the reference's interface (mmtracking/mmtrack/models/vid/base.py:15-326) and the
enhance -> detect wiring of mmtracking/mmtrack/models/vid/selsa_fastdvd_darkfarm.py:151-184,294-319.

DFF / FGFA / LLVOD take either kind of detector, as the reference's do (`hasattr(self.detector, 'roi_head')`, dff.py:153,245,
fgfa.py:165,344, llvod.py:124,225): a dense-head one (RetinaNet; everything after the frame features is one launch plan) or a
two-stage one (the reference's own `configs/vid/dff`, `configs/vid/fgfa` on faster_rcnn_r50_dc5: feature maps at module level, then
`rpn_head.simple_test_rpn` and `roi_head.simple_test`, `TwoStageVideoMixin` below; training in vid_two_stage_train.py).

The reference's SelsaFastDVDnetDetect asserts a two-stage detector (`roi_head`, :37-38) and its single-stage test
branches are broken, so the dense-head variant is registered here as `FastDVDnetDetect` with the
same constructor arguments (detector, cleaner, pretrains, frozen_modules, train_cfg, test_cfg, loss_type)."""
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

from . import engine, ops
from ._lib import LlvError
from .core.bbox import bbox2result
from .planned import PlannedModule
from .registry import MODELS, build_cleaner, build_detector


class BaseVideoDetector(PlannedModule):
    def __init__(self):
        super().__init__()

    def init_module(self, module, pretrain=None):
        """base.py:22-37"""
        if pretrain is not None:
            from .checkpoint import load_checkpoint
            print(f'load {module} from: {pretrain}')
            load_checkpoint(getattr(self, module), pretrain, strict=False)
        else:
            getattr(self, module).init_weights()
        self.invalidate_plans()

    def freeze_module(self, module):
        """base.py:39-54"""
        if isinstance(module, str):
            modules = [module]
        elif isinstance(module, (list, tuple)):
            modules = module
        else:
            raise TypeError('module must be a str or a list.')
        for name in modules:
            m = getattr(self, name)
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    with_detector = property(lambda self: hasattr(self, 'detector') and self.detector is not None)
    with_motion = property(lambda self: hasattr(self, 'motion') and self.motion is not None)
    with_aggregator = property(lambda self: hasattr(self, 'aggregator') and self.aggregator is not None)
    with_cleaner = property(lambda self: hasattr(self, 'cleaner') and self.cleaner is not None)

    def forward_test(self, imgs, img_metas, **kwargs):
        """base.py:97-138"""
        if isinstance(imgs, torch.Tensor):
            imgs = [imgs]
        elif not isinstance(imgs, list):
            raise TypeError(f'imgs must be a list or tensor, but got {type(imgs)}')
        assert isinstance(img_metas, list)
        if isinstance(img_metas[0], dict):
            img_metas = [img_metas]
        elif not isinstance(img_metas[0], list):
            raise TypeError('img_metas must be a List[List[dict]] or List[dict]')
        num_augs = len(imgs)
        if num_augs != len(img_metas):
            raise ValueError(f'num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})')
        if num_augs == 1:
            if 'proposals' in kwargs:
                kwargs['proposals'] = kwargs['proposals'][0]
            return self.simple_test(imgs[0], img_metas[0], **kwargs)
        assert imgs[0].size(0) == 1
        assert 'proposals' not in kwargs
        return self.aug_test(imgs, img_metas, **kwargs)

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        """base.py:140-156"""
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def _parse_losses(self, losses):
        """base.py:158-190 (same contract as the detector's)."""
        log_vars = OrderedDict()
        for loss_name, loss_value in losses.items():
            if isinstance(loss_value, torch.Tensor):
                log_vars[loss_name] = loss_value.mean()
            elif isinstance(loss_value, list):
                log_vars[loss_name] = sum(_loss.mean() for _loss in loss_value)
            else:
                raise TypeError(f'{loss_name} is not a tensor or list of tensors')
        loss = sum(_value for _key, _value in log_vars.items() if 'loss' in _key)
        log_vars['loss'] = loss
        for loss_name, loss_value in log_vars.items():
            if dist.is_available() and dist.is_initialized():
                loss_value = loss_value.data.clone()
                dist.all_reduce(loss_value.div_(dist.get_world_size()))
            log_vars[loss_name] = loss_value.item()
        return loss, log_vars

    def train_step(self, data, optimizer):
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))

    def aug_test(self, imgs, img_metas, **kwargs):
        raise NotImplementedError('test-time augmentation is outside the hot path')

    def reset_video(self):
        """Forget the current video (what frame_id == 0 does inside simple_test): the per-video memo and any carried streaming
        state. apis.single_gpu_test calls it before a clip segment, whose first frame need not be frame 0."""
        self.memo = None
        self._stream_state = None


def _add_noise(train_cfg):
    """The authors' forward_train of LLVOD / FGFA passes EVERY training batch through general_clean_noise_pairs (llvod.py:91,
    fgfa.py:141-147); no reference config carries a switch, so the default is on. train_cfg['add_noise'] = False turns it off."""
    return bool((train_cfg or {}).get('add_noise', True))


def flow_warp_feats(x, flow):
    """mmtracking/mmtrack/core/motion/flow.py:4-41 at module level: x (N, C, h, w) fp32 on the GPU, flow (N, 2, Hf, Wf) -> warped
    (N, C, h, w) fp32 (llv_flow_warp on the NHWC fp16 form of the map)."""
    c = x.shape[1]
    cp = ops.round_up(c, 8)
    y = ops.flow_warp(ops.pack_input(x.float().contiguous(), cp), flow.float().contiguous())
    return ops.unpack_output(y, c)


class TwoStageVideoMixin:
    """The `hasattr(self.detector, 'roi_head')` branches of the reference's flow / aggregation video detectors."""

    @property
    def two_stage(self):
        return hasattr(self.detector, 'roi_head')

    def _detect_two_stage(self, x, img_metas, proposals, rescale):
        """dff.py:245-253 = fgfa.py:344-352 = llvod.py:225-233."""
        det = self.detector
        proposal_list = det.rpn_head.simple_test_rpn(x, img_metas) if proposals is None else proposals
        outs = det.roi_head.simple_test(x, proposal_list, img_metas, rescale=rescale)
        return dict(bbox_results=outs[0])

    def _check_two_stage(self):
        from .roi_heads import StandardRoIHead
        if self.two_stage and not isinstance(self.detector.roi_head, StandardRoIHead):
            raise LlvError(f'{type(self).__name__}: a two-stage detector needs a StandardRoIHead here (roi_head.simple_test(x, proposals, '
                           f'img_metas)); {type(self.detector.roi_head).__name__} takes reference features and belongs to SELSA')


@MODELS.register_module()
class FastDVDnetDetect(BaseVideoDetector):
    """FastDVDnet (5-frame window) -> single-stage detector.

    Test-time call, as mmtrack's data pipeline delivers it (selsa_fastdvd_darkfarm.py:262-301):
        model(return_loss=False, img=[T(1,3,H,W)], img_metas=[[meta]], ref_img=[T(1,4,3,H,W)], ref_img_metas=...)
    The reference caches the four reference frames at frame 0 of a video in self.memo and re-uses them for every
    later frame (:295-301); pass ref_img with every call to slide the window instead (both are supported: a call
    without ref_img uses the memo). Returns {'bbox_results': list[num_classes] of (k,5) float32 arrays}.
    """

    def __init__(self, detector, cleaner, pretrains=None, frozen_modules=None, train_cfg=None, test_cfg=None,
                 loss_type=None):
        super().__init__()
        self.detector = build_detector(detector)
        self.cleaner = build_cleaner(cleaner)
        if hasattr(self.detector, 'roi_head'):
            raise LlvError('FastDVDnetDetect drives a dense-head (single-stage) detector; two-stage is out of scope')
        self.train_cfg, self.test_cfg, self.loss_type = train_cfg, test_cfg, loss_type
        self.memo = None
        self.init_weights(pretrains)
        if frozen_modules is not None:
            self.freeze_module(frozen_modules)

    def init_weights(self, pretrain):
        """selsa_fastdvd_darkfarm.py:47-62"""
        if pretrain is None:
            pretrain = dict()
        assert isinstance(pretrain, dict), '`pretrain` must be a dict.'
        if self.with_detector and pretrain.get('detector', False):
            self.init_module('detector', pretrain['detector'])
        if self.with_cleaner and pretrain.get('cleaner', False):
            self.init_module('cleaner', pretrain['cleaner'])

    def invalidate_plans(self):
        super().invalidate_plans()
        for m in self.children():
            if isinstance(m, PlannedModule):
                m.invalidate_plans()

    def make_plan(self, batch, height, width, device, use_graph=False):
        d = self.detector
        return engine.EnhanceDetectPlan(
            self.cleaner.state_dict(), d.state_dict(), d.bbox_head.cls_out_channels, batch, height, width, device,
            detector_kwargs=dict(depth=d.backbone.depth, strides=d.backbone.strides, style=d.backbone.style,
                                 out_indices=d.backbone.out_indices, start_level=d.neck.start_level, num_outs=d.neck.num_outs,
                                 add_extra_convs=d.neck.add_extra_convs, stacked_convs=d.bbox_head.stacked_convs,
                                 anchor_generator=d.bbox_head.anchor_generator,
                                 test_cfg=dict(d.bbox_head.test_cfg) if d.bbox_head.test_cfg else None,
                                 target_means=d.bbox_head.bbox_coder.means, target_stds=d.bbox_head.bbox_coder.stds),
            use_graph=use_graph)

    def denoise_cyclic(self, noise_imgs):
        """(5, 3, H, W) fp32 on the GPU -> (5, 3, H, W): frame i denoised from the stack of frames (i-2 .. i+2) mod 5
        (selsa_fastdvd_darkfarm.py:190-196,303-308), one FastDVDnetPlan over the 9-frame cyclic sequence [3, 4, 0, 1, 2, 3, 4, 0, 1]."""
        n, c, h, w = noise_imgs.shape
        assert n == 5 and c == 3
        plan = self._plan(('den5', h, w, str(noise_imgs.device)), lambda: engine.FastDVDnetPlan(self.cleaner.state_dict(), 5, h, w, noise_imgs.device))
        idx = torch.tensor([3, 4, 0, 1, 2, 3, 4, 0, 1], device=noise_imgs.device)
        ops.pack_input(noise_imgs.float()[idx].contiguous(), 8, out=plan.frames)
        plan.plan.run()
        return ops.unpack_output(plan.out, 3)

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, ref_img, ref_img_metas=None, ref_gt_bboxes=None, ref_gt_labels=None,
                      gt_bboxes_ignore=None, **kwargs):
        """selsa_fastdvd_darkfarm.py:64-227 for a dense-head detector: `img` (1, 6, H, W) = noisy | clean frame stacked on the channel
        axis, `ref_img` (1, 4, 6, H, W) the four reference frames likewise. The five noisy frames are denoised from their cyclic
        5-stacks, `loss_denoising_<loss_type>` compares them with the clean ones, the detector is trained on the denoised key frame
        (single_stage.py:68-96 through the device training graph). The cleaner must be frozen (`frozen_modules='cleaner'`, as in the
        authors' config): its gradient path is not built."""
        self._check_input(img)
        assert len(img) == 1, 'selsa video detector only supports 1 batch size per gpu for now.'
        assert ref_img.shape[1] == 4, 'FastDVDnet support frames is 4.'
        if any(p.requires_grad for p in self.cleaner.parameters()):
            raise LlvError("FastDVDnetDetect.forward_train: freeze the cleaner (frozen_modules='cleaner'); training it is not built")
        if img.shape[1] % 4 == 0:
            raise LlvError('FastDVDnetDetect.forward_train: 4-channel RAW input is outside the hot path')
        noise_img, clean_img = img[:, :3], img[:, 3:]
        noise_ref, clean_ref = ref_img[0][:, :3], ref_img[0][:, 3:]
        clean_imgs = torch.cat([clean_ref[:2], clean_img, clean_ref[2:]], dim=0)
        noise_imgs = torch.cat([noise_ref[:2], noise_img, noise_ref[2:]], dim=0)
        den = self.denoise_cyclic(noise_imgs)
        losses = dict()
        fn = dict(l1=torch.nn.functional.l1_loss, l2=torch.nn.functional.mse_loss, smooth_l1=torch.nn.functional.smooth_l1_loss).get(self.loss_type)
        if fn is None:
            raise NotImplementedError(f'loss_type {self.loss_type}')
        losses[f'loss_denoising_{self.loss_type}'] = fn(den, clean_imgs.float())
        from .training import single_stage_forward_train
        losses.update(single_stage_forward_train(self.detector, den[[2]].contiguous(), img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore))
        return losses

    def simple_test(self, img, img_metas, ref_img=None, ref_img_metas=None, rescale=False, **kwargs):
        self._check_input(img)
        assert len(img) == 1, 'video detectors take 1 image per GPU (selsa_fastdvd_darkfarm.py:152-153)'
        frame_id = img_metas[0].get('frame_id', -1)
        if ref_img is not None:
            refs = ref_img[0] if isinstance(ref_img, (list, tuple)) else ref_img
            refs = refs[0] if refs.dim() == 5 else refs
            assert refs.shape[0] == 4, 'FastDVDnet support frames is 4.'
            self.memo = dict(ref_imgs=refs)
        elif self.memo is None or frame_id == 0:
            raise LlvError('first frame of a video needs ref_img (4 reference frames)')
        refs = self.memo['ref_imgs']
        n, c, h, w = img.shape
        plan = self._plan((h, w, str(img.device)), lambda: self.make_plan(1, h, w, img.device))
        # window order [ref0, ref1, img, ref2, ref3] (selsa_fastdvd_darkfarm.py:300)
        window = torch.cat([refs[:2], img, refs[2:]], dim=0).float().contiguous()
        # a window that is the previous one slid by a frame (the next frame of the same video with its own references) shares two
        # of its three first-stage DenBlock results with it: they are carried over (FastDVDnetPlan.enable_stream), bit-identical
        packed = ops.pack_input(window, 8)
        prev = getattr(self, '_stream_state', None)
        carry = (prev is not None and prev[0] is plan and frame_id == prev[1] + 1 and frame_id > 0
                 and bool(torch.equal(packed[:4], plan.den.frames[1:])))
        plan.den.frames.copy_(packed)
        plan.det.post.set_metas(img_metas, rescale)
        if carry:
            plan.den.enable_stream().run()
        else:
            plan.den.plan.run()
        plan.det.plan.run()
        self._stream_state = (plan, frame_id)
        dets, labels = plan.det.results()[0]
        self.memo['denoising_img'] = plan.den.out
        return dict(bbox_results=bbox2result(dets, labels, self.detector.bbox_head.num_classes))


@MODELS.register_module()
class LLVOD(TwoStageVideoMixin, BaseVideoDetector):
    """Temporal feature aggregation without optical flow (mmtracking/mmtrack/models/vid/llvod.py:17-254). Dense-head
    detector: the pyramid features of the last `memo_len` frames (reference: 32, llvod.py:163) live in a
    device ring buffer together with their embeddings; each new frame costs one backbone+FPN pass and one embedding
    conv, then per level `EmbedAggregator(x_key, refs)` (key excluded from the references, llvod.py:201-206), the
    RetinaHead on the aggregated pyramid, decode and NMS.

    Call (test): model(return_loss=False, img=[T(1,3,H,W)], img_metas=[[meta]], ref_img=[T(1,N,3,H,W)] or None).
    `ref_img` frames are pushed into the ring before `img`; frame_id == 0 resets it. The reference's own control
    flow for this class cannot run (assert frame_id == 1 next to `if frame_id == 0`, NameError on
    general_clean_noise_pairs), so the window bookkeeping here is the plain sliding one.
    Two-stage detector (llvod.py:124-145,225-233): the same sliding window over module-level feature maps
    (`_simple_test_two_stage`), `aggregator(x_key, refs)` per level, then RPN + StandardRoIHead.
    """

    def __init__(self, detector, aggregator, pretrains=None, frozen_modules=None, train_cfg=None, test_cfg=None,
                 memo_len=32):
        super().__init__()
        from .registry import build_aggregator
        self.detector = build_detector(detector)
        self.aggregator = build_aggregator(aggregator)
        self._check_two_stage()
        self.train_cfg, self.test_cfg, self.memo_len = train_cfg, test_cfg, memo_len
        self.memo = None
        self.init_weights(pretrains)
        if frozen_modules is not None:
            self.freeze_module(frozen_modules)

    def init_weights(self, pretrain):
        if pretrain is None:
            pretrain = dict()
        assert isinstance(pretrain, dict), '`pretrain` must be a dict.'
        if self.with_detector and pretrain.get('detector', False):
            self.init_module('detector', pretrain['detector'])
        if self.with_aggregator and pretrain.get('aggregator', False):
            self.init_module('aggregator', pretrain['aggregator'])

    def invalidate_plans(self):
        super().invalidate_plans()
        self.memo = None
        for m in self.children():
            if isinstance(m, PlannedModule):
                m.invalidate_plans()

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, ref_img, ref_img_metas=None, ref_gt_bboxes=None, ref_gt_labels=None,
                      gt_bboxes_ignore=None, gt_masks=None, proposals=None, **kwargs):
        """llvod.py:45-154: the key frame is put in the middle of its references (:87-88), every frame goes
        through backbone + FPN, per level `aggregator(feat[key], feat[others])` (:107-112), then bbox_head.forward_train (:148-150).
        The reference's body also (a) adds synthetic noise (general_clean_noise_pairs, :91: applied here when train_cfg['add_noise']
        is true, on un-normalised frames as the reference's data pipeline delivers them) and (b) regresses the noisy frames' features
        onto a `cleaner`'s (:93-94,114-118), which needs a backbone returning two feature sets and exists only with the authors'
        DarkResNet: without a cleaner (`with_cleaner` false) that term is absent there too.
        Input contract of the noise step: un-normalised 0..255 frames (BGR as the loader delivers them); `train_cfg=dict(add_noise=False)`
        switches it off (the parity tests do: random draws cannot be shared with the CPU oracle)."""
        from . import vid_train
        self._check_input(img)
        assert len(img) == 1, 'video detector only support 1 batch size per gpu for now.'
        index = ref_img.shape[1] // 2
        all_imgs = torch.cat((ref_img[0][:index], img, ref_img[0][index:]), dim=0)
        if _add_noise(self.train_cfg):
            from .core.add_noise import general_clean_noise_pairs
            all_imgs, _ = general_clean_noise_pairs(all_imgs, constant=[0.5, 0.5])
        if self.two_stage:
            from .vid_two_stage_train import video_two_stage_forward_train
            return video_two_stage_forward_train(self, 'llvod', all_imgs.float().contiguous(), None, index, img_metas, gt_bboxes, gt_labels,
                                                 gt_bboxes_ignore, proposals, gt_masks=gt_masks)
        return vid_train.video_forward_train(self, 'llvod', all_imgs.float().contiguous(), None, index, img_metas, gt_bboxes, gt_labels,
                                             gt_bboxes_ignore)

    def _make_plan(self, h, w, device):
        from .aggregators import embed_packed_convs
        d, ag = self.detector, self.aggregator
        sd = engine._cpu_sd(d.state_dict())
        bb, neck, head = d.backbone, d.neck, d.bbox_head
        # stage A: backbone + FPN + embedding convs of ONE frame
        ba = engine.Builder(device)
        x_in = ba.buf(1, h, w, 8)
        feats = engine.build_resnet(ba, sd, x_in, p='backbone.', **bb.plan_kwargs())
        pyr = engine.build_fpn(ba, sd, feats, p='neck.', **neck.plan_kwargs())
        emb = pyr.buf
        for i, (pc, flags) in enumerate(embed_packed_convs(engine._cpu_sd(ag.state_dict()), ag.num_convs)):
            nxt = ba.buf(pyr.total, pc.cout)
            ba.conv_desc(f'embed{i}', pc, [emb], nxt, 1, pyr.segs(), emb.shape[-1], nxt.shape[-1], flags=flags)
            emb = nxt
        # ring buffers per level: features and embeddings of the last memo_len frames
        c = pyr.buf.shape[-1]
        ring_f = [torch.zeros((self.memo_len, hh, ww, c), device=device, dtype=torch.float16) for hh, ww in pyr.level_hw]
        ring_e = [torch.zeros((self.memo_len, hh, ww, emb.shape[-1]), device=device, dtype=torch.float16) for hh, ww in pyr.level_hw]
        # stage B: head + decode + NMS on the aggregated pyramid
        bbld = engine.Builder(device)
        agg = engine.Pyramid(bbld.buf(pyr.total, c), pyr.level_hw, pyr.level_off, 1)
        cls_buf, reg_buf = engine.build_retina_head(bbld, sd, agg, head.stacked_convs, True, p='bbox_head.')
        post = engine.PostProcess(bbld, cls_buf, reg_buf, agg, head.cls_out_channels, head.anchor_generator,
                                  dict(head.test_cfg) if head.test_cfg else engine.DEFAULT_TEST_CFG, head.bbox_coder.means,
                                  head.bbox_coder.stds, [(h, w, 3)], None, False, head.bbox_coder.clip_border)
        return dict(a=ba.plan, b=bbld.plan, x_in=x_in, pyr=pyr, emb=emb, ring_f=ring_f, ring_e=ring_e, agg=agg, post=post)

    def _push(self, plan, frame):
        """One frame (1,3,H,W) through stage A into the next ring slot; returns the slot."""
        ops.pack_input(frame.float().contiguous(), 8, out=plan['x_in'])
        plan['a'].run()
        m = self.memo
        slot = m['next']
        pyr = plan['pyr']
        for l in range(len(pyr.level_hw)):
            plan['ring_f'][l][slot].copy_(pyr.level(l)[0])
            plan['ring_e'][l][slot].copy_(pyr.level(l, plan['emb'])[0])
        m['next'] = (slot + 1) % self.memo_len
        m['valid'] = min(m['valid'] + 1, self.memo_len)
        return slot

    def _simple_test_two_stage(self, img, img_metas, ref_img, proposals, rescale):
        """The sliding window of `simple_test` below over module-level feature maps: every frame (references first, then the key frame)
        goes through `detector.extract_feat` once and joins the last `memo_len` frames' maps; per level aggregator(x_key, the others)."""
        frame_id = img_metas[0].get('frame_id', -1)
        if self.memo is None or frame_id == 0 or 'feats' not in self.memo:     # (apis.prepare_segment leaves the ring form's counters)
            self.memo = dict(feats=None)
        new = []
        if ref_img is not None:
            refs = ref_img[0] if isinstance(ref_img, (list, tuple)) else ref_img
            refs = refs[0] if refs.dim() == 5 else refs
            new += [refs[i:i + 1] for i in range(refs.shape[0])]
        new.append(img)
        for f in new:
            x = self.detector.extract_feat(f.float())
            m = self.memo['feats']
            self.memo['feats'] = [t.clone() for t in x] if m is None else [torch.cat((a, t), 0)[-self.memo_len:] for a, t in zip(m, x)]
        feats = self.memo['feats']
        if feats[0].shape[0] < 2:
            agg = tuple(f[-1:] for f in feats)                          # no references yet: the frame's own features
        else:
            agg = tuple(self.aggregator(f[-1:], f[:-1].contiguous()) for f in feats)
        return self._detect_two_stage(agg, img_metas, proposals, rescale)

    def simple_test(self, img, img_metas, ref_img=None, ref_img_metas=None, proposals=None, rescale=False, **kwargs):
        self._check_input(img)
        assert len(img) == 1, 'video detectors take 1 image per GPU (llvod.py:65)'
        if self.two_stage:
            return self._simple_test_two_stage(img, img_metas, ref_img, proposals, rescale)
        n, c, h, w = img.shape
        plan = self._plan((h, w, str(img.device)), lambda: self._make_plan(h, w, img.device))
        frame_id = img_metas[0].get('frame_id', -1)
        if self.memo is None or frame_id == 0:
            self.memo = dict(next=0, valid=0)
        if ref_img is not None:
            refs = ref_img[0] if isinstance(ref_img, (list, tuple)) else ref_img
            refs = refs[0] if refs.dim() == 5 else refs
            for i in range(refs.shape[0]):
                self._push(plan, refs[i:i + 1])
        key = self._push(plan, img)
        pyr, agg = plan['pyr'], plan['agg']
        nv = self.memo['valid']
        for l, (hh, ww) in enumerate(pyr.level_hw):
            out = agg.level(l)
            if nv < 2:
                out.copy_(plan['ring_f'][l][key:key + 1])           # no references yet: the frame's own features
            else:
                ops.check(ops.lib().llv_embed_aggregate_ring(ops._ptr(plan['ring_e'][l]), ops._ptr(plan['ring_f'][l]), ops._ptr(out),
                                                             nv, key, hh * ww, out.shape[-1], plan['ring_e'][l].shape[-1],
                                                             plan['ring_f'][l].shape[-1], out.shape[-1], ops._stream()),
                          'llv_embed_aggregate_ring')
        plan['post'].set_metas(img_metas, rescale)
        plan['b'].run()
        dets, labels = plan['post'].results()[0]
        return dict(bbox_results=bbox2result(dets, labels, self.detector.bbox_head.num_classes))


class _StagePlans:
    """Plan pieces shared by the flow-based video detectors: stage A = backbone + FPN of `batch` frames into a pyramid,
    stage B = RetinaHead + decode + NMS reading ONE frame's pyramid from `cur`."""

    def __init__(self, det, h, w, device, batch_a):
        sd = engine._cpu_sd(det.state_dict())
        bb, neck, head = det.backbone, det.neck, det.bbox_head
        self.a = {}
        for n in sorted(set(batch_a)):
            ba = engine.Builder(device)
            x_in = ba.buf(n, h, w, 8)
            feats = engine.build_resnet(ba, sd, x_in, p='backbone.', **bb.plan_kwargs())
            pyr = engine.build_fpn(ba, sd, feats, p='neck.', **neck.plan_kwargs())
            self.a[n] = (ba.plan, x_in, pyr)
        p1 = self.a[min(self.a)][2]
        bh = engine.Builder(device)
        self.cur = engine.Pyramid(bh.buf(sum(hh * ww for hh, ww in p1.level_hw), p1.buf.shape[-1]), p1.level_hw,
                                  [sum(a * b for a, b in p1.level_hw[:i]) for i in range(len(p1.level_hw))], 1)
        cls_buf, reg_buf = engine.build_retina_head(bh, sd, self.cur, head.stacked_convs, True, p='bbox_head.')
        self.post = engine.PostProcess(bh, cls_buf, reg_buf, self.cur, head.cls_out_channels, head.anchor_generator,
                                       dict(head.test_cfg) if head.test_cfg else engine.DEFAULT_TEST_CFG, head.bbox_coder.means,
                                       head.bbox_coder.stds, [(h, w, 3)], None, False, head.bbox_coder.clip_border)
        self.b = bh.plan
        self.num_classes = head.num_classes

    def extract(self, frames):
        """frames (n,3,H,W) fp32 GPU -> the stage-A pyramid of batch n (valid until the next extract of that batch)."""
        n = frames.shape[0]
        plan, x_in, pyr = self.a[n]
        ops.pack_input(frames.float().contiguous(), 8, out=x_in)
        plan.run()
        return pyr

    def detect(self, img_metas, rescale):
        self.post.set_metas(img_metas, rescale)
        self.b.run()
        dets, labels = self.post.results()[0]
        return dict(bbox_results=bbox2result(dets, labels, self.num_classes))


def _warp_pyramid(src_levels, flow, dst_levels):
    """flow_warp_feats per FPN level (mmtracking/mmtrack/core/motion/flow.py:4-41): src/dst lists of (n,h,w,C) fp16."""
    for s, d in zip(src_levels, dst_levels):
        n, hh, ww, c = s.shape
        ops.check(ops.lib().llv_flow_warp(ops._ptr(s), ops._ptr(flow), ops._ptr(d), n, hh, ww, c, c, flow.shape[2], flow.shape[3],
                                          ops._stream()), 'llv_flow_warp')


@MODELS.register_module()
class DFF(TwoStageVideoMixin, BaseVideoDetector):
    """Deep Feature Flow (mmtracking/mmtrack/models/vid/dff.py:11-282), dense-head or two-stage detector: every
    `test_cfg.key_frame_interval`-th frame runs backbone + FPN and caches image and pyramid; the frames in between
    run FlowNetSimple on (frame, key frame) and warp the cached pyramid (dff.py:205-216), then the head."""

    def __init__(self, detector, motion, pretrains=None, frozen_modules=None, train_cfg=None, test_cfg=None):
        super().__init__()
        from .registry import build_motion
        self.detector = build_detector(detector)
        self.motion = build_motion(motion)
        self._check_two_stage()
        self.train_cfg, self.test_cfg = train_cfg, test_cfg or dict()
        self.memo = None
        self.init_weights(pretrains)
        if frozen_modules is not None:
            self.freeze_module(frozen_modules)

    def init_weights(self, pretrain):
        if pretrain is None:
            pretrain = dict()
        assert isinstance(pretrain, dict), '`pretrain` must be a dict.'
        if self.with_detector and pretrain.get('detector', False):
            self.init_module('detector', pretrain['detector'])
        if self.with_motion:
            self.init_module('motion', pretrain.get('motion', None))

    def invalidate_plans(self):
        super().invalidate_plans()
        self.memo = None
        for m in self.children():
            if isinstance(m, PlannedModule):
                m.invalidate_plans()

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, ref_img, ref_img_metas=None, ref_gt_bboxes=None, ref_gt_labels=None,
                      gt_bboxes_ignore=None, gt_masks=None, proposals=None, **kwargs):
        """dff.py:49-182: flow = motion(cat(img, ref)); the REFERENCE frame's features are warped to the key frame (or used as they are
        for still-image data, :143-144) and fed, with the key frame's ground truth, to bbox_head.forward_train (dense head, :174-177) or
        to rpn_head.forward_train + roi_head.forward_train (two-stage, :153-173: vid_two_stage_train.py).
        The flow network trains with the detector unless it is frozen (vid_train.motion_flow)."""
        from . import vid_train
        self._check_input(img)
        assert len(img) == 1, 'Dff video detectors only support 1 batch size per gpu for now.'
        is_video_data = img_metas[0].get('is_video_data', True)
        ref = ref_img[:, 0].float().contiguous()
        flow = None
        if is_video_data:
            flow = vid_train.motion_flow(self, torch.cat((img.float(), ref), dim=1), img_metas)
        if self.two_stage:
            from .vid_two_stage_train import video_two_stage_forward_train
            return video_two_stage_forward_train(self, 'dff', ref, flow, 0, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, proposals,
                                                 warp=is_video_data, gt_masks=gt_masks)
        return vid_train.video_forward_train(self, 'dff', ref, flow, 0, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, warp=is_video_data)

    def extract_feats(self, img, img_metas):
        """dff.py:184-217 at module level (the two-stage branch): tuple of NCHW fp32 maps."""
        key_frame_interval = self.test_cfg.get('key_frame_interval', 10)
        frame_id = img_metas[0].get('frame_id', -1)
        assert frame_id >= 0
        if frame_id % key_frame_interval == 0 or self.memo is None:
            x = self.detector.extract_feat(img.float())
            self.memo = dict(img=img.float().clone(), feats=[t.clone() for t in x])
            return tuple(x)
        flow = self.motion(torch.cat((img.float(), self.memo['img']), dim=1), img_metas)
        return tuple(flow_warp_feats(f, flow) for f in self.memo['feats'])

    def simple_test(self, img, img_metas, ref_img=None, ref_img_metas=None, proposals=None, rescale=False, **kwargs):
        self._check_input(img)
        assert len(img) == 1, 'Dff video detectors only support 1 batch size per gpu for now.'
        if self.two_stage:
            return self._detect_two_stage(self.extract_feats(img, img_metas), img_metas, proposals, rescale)
        n, c, h, w = img.shape
        st = self._plan(('stages', h, w, str(img.device)), lambda: _StagePlans(self.detector, h, w, img.device, [1]))
        key_frame_interval = self.test_cfg.get('key_frame_interval', 10)
        frame_id = img_metas[0].get('frame_id', -1)
        assert frame_id >= 0
        is_key_frame = frame_id % key_frame_interval == 0
        if is_key_frame or self.memo is None:
            pyr = st.extract(img)
            self.memo = dict(img=img.float().clone(), feats=[pyr.level(l).clone() for l in range(len(pyr.level_hw))])
            st.cur.buf.copy_(pyr.buf)
        else:
            flow_img = torch.cat((img.float(), self.memo['img']), dim=1)          # dff.py:211
            flow = self.motion(flow_img, img_metas)
            _warp_pyramid(self.memo['feats'], flow.contiguous(), [st.cur.level(l) for l in range(len(st.cur.level_hw))])
        return st.detect(img_metas, rescale)


@MODELS.register_module()
class FGFA(TwoStageVideoMixin, BaseVideoDetector):
    """Flow-Guided Feature Aggregation (mmtracking/mmtrack/models/vid/fgfa.py:12-384), dense-head or two-stage detector,
    both test-time branches of extract_feats (fgfa.py:196-284). Adaptive stride (`frame_stride < 1`): the reference
    frames given with frame 0 stay in the memo; every frame: backbone + FPN, FlowNetSimple on (frame, each memo frame),
    warp the memo pyramids, EmbedAggregator(x, cat(x, warped)) per level, head. Fixed stride: the memo is a sliding
    window that gains the new reference frame every `frame_stride` frames; the key frame's features come out of the
    window (slot num_left_ref_imgs) and replace that slot of the warped stack."""

    def __init__(self, detector, motion, aggregator, pretrains=None, frozen_modules=None, train_cfg=None, test_cfg=None):
        super().__init__()
        from .registry import build_aggregator, build_motion
        self.detector = build_detector(detector)
        self.motion = build_motion(motion)
        self.aggregator = build_aggregator(aggregator)
        self._check_two_stage()
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.memo = None
        self.init_weights(pretrains)
        if frozen_modules is not None:
            self.freeze_module(frozen_modules)

    def init_weights(self, pretrain):
        if pretrain is None:
            pretrain = dict()
        assert isinstance(pretrain, dict), '`pretrain` must be a dict.'
        if self.with_detector and pretrain.get('detector', False):
            self.init_module('detector', pretrain['detector'])
        if self.with_motion:
            self.init_module('motion', pretrain.get('motion', None))
        if self.with_aggregator and pretrain.get('aggregator', False):
            self.init_module('aggregator', pretrain['aggregator'])

    def invalidate_plans(self):
        super().invalidate_plans()
        self.memo = None
        for m in self.children():
            if isinstance(m, PlannedModule):
                m.invalidate_plans()

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, ref_img, ref_img_metas=None, ref_gt_bboxes=None, ref_gt_labels=None,
                      gt_bboxes_ignore=None, gt_masks=None, proposals=None, **kwargs):
        """fgfa.py:52-194: flows = motion(cat(img, ref_i)) for every reference, all frames through backbone + neck, per level
        `aggregator(feat[img], flow_warp_feats(feat[refs], flows))` (:155-160), then bbox_head.forward_train (dense head, :186-189) or
        rpn_head.forward_train + roi_head.forward_train (two-stage, :165-185: vid_two_stage_train.py). The authors' noise injection
        (:141-147, every training batch) runs unless train_cfg['add_noise'] is false (input contract: un-normalised 0..255 frames).
        The flow network trains too unless it is frozen."""
        from . import vid_train
        self._check_input(img)
        assert len(img) == 1, 'fgfa video detectors only support 1 batch size per gpu for now.'
        refs = ref_img[0].float()
        img = img.float()
        if _add_noise(self.train_cfg):
            from .core.add_noise import general_clean_noise_pairs
            noise, _ = general_clean_noise_pairs(torch.cat([img, refs], dim=0), constant=[0.5, 0.5])
            img, refs = noise[[0]], noise[1:]
        nr = refs.shape[0]
        flows = vid_train.motion_flow(self, torch.cat((img.repeat(nr, 1, 1, 1), refs), dim=1), img_metas)
        all_imgs = torch.cat((img, refs), dim=0).contiguous()
        if self.two_stage:
            from .vid_two_stage_train import video_two_stage_forward_train
            return video_two_stage_forward_train(self, 'fgfa', all_imgs, flows, 0, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, proposals,
                                                 gt_masks=gt_masks)
        return vid_train.video_forward_train(self, 'fgfa', all_imgs, flows, 0, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore)

    def _agg_plan(self, st, n_slots, key_slot, device):
        """Embedding convs over the n_slots frames of every level in one multi-segment launch + the tail per level.
        key_slot = which slot holds the key frame's own features (0 in the adaptive-stride branch, where the stack is
        cat(x, warped refs); num_left_ref_imgs in the fixed-stride branch, where x replaces that warped slot)."""
        from .aggregators import embed_packed_convs
        ag = self.aggregator
        nb = n_slots
        hw = st.cur.level_hw
        b = engine.Builder(device)
        offs, tot = [], 0
        for hh, ww in hw:
            offs.append(tot)
            tot += nb * hh * ww
        c = st.cur.buf.shape[-1]
        stack = engine.Pyramid(b.buf(tot, c), hw, offs, nb)
        emb = stack.buf
        for i, (pc, flags) in enumerate(embed_packed_convs(engine._cpu_sd(ag.state_dict()), ag.num_convs)):
            nxt = b.buf(tot, pc.cout)
            b.conv_desc(f'embed{i}', pc, [emb], nxt, nb, stack.segs(), emb.shape[-1], nxt.shape[-1], flags=flags)
            emb = nxt
        for l, (hh, ww) in enumerate(hw):
            e_l, s_l = stack.level(l, emb), stack.level(l)
            b.plan.add(f'aggregate{l}', ops.lib().llv_embed_aggregate,
                       (ops._ptr(e_l[key_slot:key_slot + 1]), ops._ptr(e_l), ops._ptr(s_l), ops._ptr(st.cur.level(l)), nb, hh * ww, c,
                        emb.shape[-1], c, c))
        b.plan.keep += [emb]
        return b.plan, stack

    def extract_feats(self, img, img_metas, ref_img, ref_img_metas=None):
        """fgfa.py:196-284 at module level (the two-stage branch): tuple of aggregated NCHW fp32 maps. ref_img (N, C, H, W) or None."""
        frame_id = img_metas[0].get('frame_id', -1)
        assert frame_id >= 0
        num_left = img_metas[0].get('num_left_ref_imgs', -1)
        frame_stride = img_metas[0].get('frame_stride', -1)
        ext1 = lambda t: self.detector.extract_feat(t.float())
        # references one frame per launch set, as every later reference enters the window (tile / split-K choices follow the launch
        # size): a clip segment that starts mid-video then rebuilds bit for bit the window the one-rank run has slid to
        ext = lambda t: ext1(t) if t.shape[0] == 1 else tuple(torch.cat(l, 0) for l in zip(*[ext1(t[i:i + 1]) for i in range(t.shape[0])]))
        if frame_stride < 1:                                            # adaptive stride
            if frame_id == 0 or self.memo is None:
                if ref_img is None:
                    raise LlvError('frame 0 of a video needs ref_img (fgfa.py:224-230,234-246)')
                self.memo = dict(img=ref_img.float(), feats=[t.clone() for t in ext(ref_img)])
            x = ext(img)
        else:
            # Fixed stride: the window (fgfa.py:243-257 `torch.cat((memo, new), dim=0)[k:]`) is a RING here. Window position u (in units of
            # the stride; the key frame of step j = frame_id // stride is u = j, the window u = j - num_left .. j + right) lives in slot
            # u % N: a slide overwrites the slot of the position that leaves instead of copying the N - 1 that stay (31 x 512 x 48 x 80
            # fp32 features + 31 frames: 0.6 GB per frame for the reference's 30-reference test sampler). Flow, warp and the softmax over
            # the references see the same N frames in slot order; a clip segment that starts mid-video fills the slots the one-rank run
            # has at that frame, so both take every sum in the same order.
            j = frame_id // frame_stride
            if frame_id == 0 or self.memo is None:
                if ref_img is None:
                    raise LlvError('frame 0 of a video needs ref_img (fgfa.py:224-230,234-246)')
                n = ref_img.shape[0]
                slots = torch.tensor([(j - num_left + i) % n for i in range(n)], device=ref_img.device)
                ring = lambda t: torch.empty_like(t).index_copy_(0, slots, t)
                self.memo = dict(img=ring(ref_img.float()), feats=[ring(t) for t in ext(ref_img)])
            elif frame_id % frame_stride == 0:
                assert ref_img is not None
                ref_x = ext(ref_img)
                n, k = self.memo['img'].shape[0], ref_img.shape[0]
                right = n - 1 - num_left
                slots = torch.tensor([(j + right - k + 1 + m) % n for m in range(k)], device=ref_img.device)
                self.memo['img'].index_copy_(0, slots, ref_img.float())
                for m_, r_ in zip(self.memo['feats'], ref_x):
                    m_.index_copy_(0, slots, r_)
            else:
                assert ref_img is None
            key_slot = j % self.memo['img'].shape[0]
            x = ext(img) if (frame_id % frame_stride and frame_id != 0) else [f[[key_slot]] for f in self.memo['feats']]
        nr = self.memo['img'].shape[0]
        flows = self.motion(torch.cat((img.float().repeat(nr, 1, 1, 1), self.memo['img']), dim=1), img_metas)
        agg_x = []
        for i in range(len(x)):
            stack = flow_warp_feats(self.memo['feats'][i], flows)
            if frame_stride < 1:
                stack = torch.cat((x[i], stack), dim=0)
            else:
                stack[key_slot] = x[i][0]
            agg_x.append(self.aggregator(x[i], stack.contiguous()))
        return tuple(agg_x)

    def simple_test(self, img, img_metas, ref_img=None, ref_img_metas=None, proposals=None, rescale=False, **kwargs):
        self._check_input(img)
        assert len(img) == 1, 'fgfa video detectors only support 1 batch size per gpu for now.'
        if self.two_stage:
            refs = None
            if ref_img is not None:
                refs = ref_img[0] if isinstance(ref_img, (list, tuple)) else ref_img
                refs = refs[0] if refs.dim() == 5 else refs
            if (self.test_cfg or {}).get('add_noise', False):
                # the authors' evaluation hack (fgfa.py:330-335: clean test frames darkened / noised on the fly). Unconditional there --
                # and with `ref_img.size(0)` on a frame without references it raises -- so it is opt-in here (test_cfg['add_noise'])
                from .core.add_noise import general_clean_noise_pairs
                noise, _ = general_clean_noise_pairs(img if refs is None else torch.cat([img, refs], dim=0), constant=[0.5, 0.5])
                img, refs = noise[[0]], (None if refs is None else noise[1:])
            return self._detect_two_stage(self.extract_feats(img, img_metas, refs), img_metas, proposals, rescale)
        n, c, h, w = img.shape
        frame_id = img_metas[0].get('frame_id', -1)
        assert frame_id >= 0
        frame_stride = img_metas[0].get('frame_stride', -1)
        num_left = img_metas[0].get('num_left_ref_imgs', -1)
        dev = str(img.device)

        def refs_of(ref_img):
            refs = ref_img[0] if isinstance(ref_img, (list, tuple)) else ref_img
            return (refs[0] if refs.dim() == 5 else refs).float().contiguous()

        def stages(nr, batches):
            return self._plan(('stages', h, w, nr, tuple(batches), dev), lambda: _StagePlans(self.detector, h, w, img.device, batches))

        fixed = frame_stride >= 1
        x_levels = None                   # fixed-stride branches that take x out of the memo
        if frame_id == 0 or self.memo is None:
            if ref_img is None:
                raise LlvError('frame 0 of a video needs ref_img (fgfa.py:224-230,234-246)')
            refs = refs_of(ref_img)
            nr = refs.shape[0]
            st = stages(nr, [1])
            # one frame per launch set, the way every later reference enters the window: a frame's features then never depend on
            # which other frames shared its launches (tile / split-K choices follow the launch size), so a clip segment that starts
            # mid-video (apis.sharded_clip_test) rebuilds bit for bit the window the one-rank run has slid to
            per = []
            for i in range(nr):
                pyr = st.extract(refs[i:i + 1])
                per.append([pyr.level(l).clone() for l in range(len(pyr.level_hw))])
            self.memo = dict(img=refs, nr=nr, feats=[torch.cat([p[l] for p in per], 0) for l in range(len(per[0]))])
            if fixed:                     # the key frame is one of the references (fgfa.py:243-246)
                assert 0 <= num_left < nr
                x_levels = [f[num_left:num_left + 1] for f in self.memo['feats']]
        elif fixed and frame_id % frame_stride == 0:
            # slide the window: append the new reference(s), drop the oldest (fgfa.py:247-257)
            assert ref_img is not None
            new = refs_of(ref_img)
            k = new.shape[0]
            st = stages(self.memo['nr'], [1])
            for i in range(k):
                pyr = st.extract(new[i:i + 1])
                for l in range(len(pyr.level_hw)):
                    self.memo['feats'][l] = torch.cat((self.memo['feats'][l], pyr.level(l)), 0)[1:].contiguous()
            self.memo['img'] = torch.cat((self.memo['img'], new), 0)[k:].contiguous()
            x_levels = [f[num_left:num_left + 1] for f in self.memo['feats']]
        elif fixed:
            assert ref_img is None        # fgfa.py:258-260
        nr = self.memo['nr']
        st = stages(nr, [1])
        n_slots, key_slot = (nr, num_left) if fixed else (nr + 1, 0)
        agg_plan, stack = self._plan(('agg', h, w, n_slots, key_slot, dev), lambda: self._agg_plan(st, n_slots, key_slot, img.device))
        if x_levels is None:
            x = st.extract(img)
            x_levels = [x.level(l) for l in range(len(x.level_hw))]
        nl = len(x_levels)
        flow_imgs = torch.cat((img.float().repeat(nr, 1, 1, 1), self.memo['img']), dim=1)      # fgfa.py:264-267
        flows = self.motion(flow_imgs, img_metas).contiguous()
        if fixed:                         # warped references, with the key frame's slot replaced by x (fgfa.py:275-276)
            _warp_pyramid(self.memo['feats'], flows, [stack.level(l) for l in range(nl)])
            for l in range(nl):
                stack.level(l)[key_slot:key_slot + 1].copy_(x_levels[l])
        else:                             # cat(x, warped references) (fgfa.py:273-274)
            for l in range(nl):
                stack.level(l)[0:1].copy_(x_levels[l])
            _warp_pyramid(self.memo['feats'], flows, [stack.level(l)[1:] for l in range(nl)])
        agg_plan.run()
        return st.detect(img_metas, rescale)
