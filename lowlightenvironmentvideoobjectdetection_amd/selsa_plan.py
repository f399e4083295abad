"""One key frame of the SELSA test step (memo mode: the reference frames of the video stay fixed, mmtracking/mmtrack/models/vid/
selsa.py:172-205,260-330 with `test_with_adaptive_stride`) as ONE launch list / HIP graph of fixed shapes:

  frame -> ResNet (DC5) -> ChannelMapper -> RPN convs -> decode + top-k -> NMS (<= P proposals, count on the device) -> rois
        -> RoIAlign -> shared fcs, each followed by a SelsaAggregator over [static reference rows | this frame's own proposals]
        -> fc_cls / fc_reg -> class softmax -> per-class delta2bbox -> multiclass NMS -> results in pinned memory

Every buffer is sized for P = the RPN's max_per_img proposals; the rows behind the frame's actual count are padding: their RoIs are
zero, they are masked out of the attention (llv_selsa_softmax's live count) and score 0 in every class (llv_softmax_rows_masked), so
the detections equal those of the module-level path, which works on exactly `count` rows (tests/test_selsa.py). The matrix products
run on llv_conv2d_nhwc_fwd (selsa.linear_pc / selsa.selsa_attention's layout); nothing in the step reads the device on the host.

Eligibility (otherwise SELSA.simple_test keeps the module-level path): one backbone output and a one-conv ChannelMapper, a single RPN
level with a sigmoid objectness, min_bbox_size 0, SingleRoIExtractor with one level, class-specific box regression."""
import ctypes as C

import torch

from . import engine, ops
from ._lib import CONV_OUT_F32, CONV_RELU, CONV_RES_ADD, LlvError, check, lib


def eligible(model):
    det = model.detector
    try:
        from .roi_extractors import SingleRoIExtractor, TemporalRoIAlign
        rpn, rh = det.rpn_head, det.roi_head
        ex, head = rh.bbox_roi_extractor, rh.bbox_head
        cfg = dict(rpn.test_cfg)
        return (det.neck is not None and len(det.neck.convs) == 1 and len(det.backbone.out_indices) == 1
                and len(rpn.anchor_generator.strides) == 1 and getattr(rpn, 'use_sigmoid_cls', True) and rpn.cls_out_channels == 1
                and cfg.get('min_bbox_size', 0) <= 0 and type(ex) is SingleRoIExtractor and not isinstance(ex, TemporalRoIAlign)
                and len(ex.roi_layers) == 1 and not head.reg_class_agnostic and head.num_shared_fcs >= 1)
    except AttributeError:
        return False


class SelsaKeyPlan:
    def __init__(self, model, height, width, device, static_branch, img_meta, rescale=False, use_graph=True):
        det = model.detector
        rpn, rh = det.rpn_head, det.roi_head
        ex, head = rh.bbox_roi_extractor, rh.bbox_head
        L = lib()
        b = engine.Builder(device)
        self.builder, self.plan = b, b.plan
        sd = engine._cpu_sd(det.state_dict())
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        # ---- backbone + neck
        self.x_in = b.buf(1, height, width, 8)
        feats = engine.build_resnet(b, sd, self.x_in, fuse_blocks=True, p='backbone.', **det.backbone.plan_kwargs())
        nconv = det.neck.convs[0].conv
        f = b.conv('neck.conv0', feats[-1], ops.pack_conv(sd['neck.convs.0.conv.weight'], sd.get('neck.convs.0.conv.bias'), 1,
                                                          (nconv.kernel_size[0] - 1) // 2), CONV_RELU if det.neck.act else 0)
        _, hf, wf, cf = f.shape
        # ---- RPN: convs, decode + top-k, NMS (the level plays the class, rpn_head.py:215-229: one level here)
        pk = lambda m, pad: ops.pack_conv(m.weight.detach().float().cpu(), m.bias.detach().float().cpu(), 1, pad)
        t = b.conv('rpn.conv', f, pk(rpn.rpn_conv, 1), CONV_RELU)
        cls = b.conv('rpn.cls', t, pk(rpn.rpn_cls, 0), CONV_OUT_F32)
        reg = b.conv('rpn.reg', t, pk(rpn.rpn_reg, 0), CONV_OUT_F32)
        rcfg = dict(rpn.test_cfg)
        nms = rcfg['nms'] if 'nms' in rcfg else dict(type='nms', iou_threshold=rcfg['nms_thr'])
        P = int(rcfg.get('max_per_img', rcfg.get('max_num')))
        self.P = P
        img_shape = tuple(int(v) for v in img_meta['img_shape'])
        pyr1 = engine.Pyramid(cls.view(hf * wf, cls.shape[-1]), [(hf, wf)], [0], 1)
        self.rpn_post = engine.PostProcess(b, cls.view(hf * wf, cls.shape[-1]), reg.view(hf * wf, reg.shape[-1]), pyr1, 1, rpn.anchor_generator,
                                           dict(nms_pre=rcfg.get('nms_pre', -1), score_thr=0.0, nms=nms, max_per_img=P), rpn.bbox_coder.means,
                                           rpn.bbox_coder.stds, [img_shape], None, False, rpn.bbox_coder.clip_border)
        nb_ = self.rpn_post.nb
        self.count = nb_.count
        self.rois = b.buf(P, 5, dtype=torch.float32)
        self.boxes4 = b.buf(P, 4, dtype=torch.float32)
        b.plan.add('rois', L.llv_dets_to_rois, (ops._ptr(nb_.dets), ops._ptr(nb_.count), ops._ptr(self.rois), ops._ptr(self.boxes4), P, 0))
        # ---- RoIAlign of the frame's proposals on its own feature map
        rl = ex.roi_layers[0]
        ph, pw = rl.output_size
        roi = b.buf(P, ph, pw, cf)
        b.plan.add('roi_align', L.llv_roi_align, (ops._ptr(f), ops._ptr(self.rois), ops._ptr(roi), 1, hf, wf, cf, cf, P, ph, pw,
                                                  C.c_float(rl.spatial_scale), int(rl.sampling_ratio), 1 if rl.aligned else 0, 0))
        # ---- head. x: the key proposals' chain (aggregated); r: the same rows as REFERENCES of themselves (selsa.py:312-315: the key
        # frame is among its own references), whose chain has no aggregation (selsa_bbox_head.py:70-76)
        from .selsa import linear_pc
        nf = head.num_shared_fcs
        self._kv = []
        def merged_pc(*fcs):
            """several fully connected layers over the same rows as ONE 1x1 convolution: output channels side by side (the K sum of
            every output is the layer's own: only the launch count changes -- each of these small-M products is launch-bound)"""
            wt = torch.cat([m.weight.detach().float().cpu() for m in fcs], 0)
            bs = torch.cat([m.bias.detach().float().cpu() for m in fcs], 0)
            return ops.pack_conv(wt.view(wt.shape[0], wt.shape[1], 1, 1), bs, 1, 0)

        def conv_view(name, xv, x_ld, pc, flags=0):
            """b.conv on a channel slice of a wider buffer (row stride x_ld)"""
            n_, h_, w_, _ = xv.shape
            out = b.buf(n_, h_, w_, pc.groups * pc.cout, dtype=torch.float32 if flags & CONV_OUT_F32 else torch.float16)
            b.conv_desc(name, pc, [xv], out, n_, [(h_, w_, h_, w_, 0, 0)], x_ld, out.shape[-1], None, 0, flags)
            return out

        x, r = roi, roi
        xr = None                              # rows [0, P): the key chain's input to the next fc, rows [P, 2P): the reference chain's
        for i, fc in enumerate(head.shared_fcs):
            pc = linear_pc(fc, roi_hw=(ph, pw) if i == 0 else None)
            if i == 0:
                x_pre = b.conv(f'head.fc{i}', x, pc)                           # (P, 1, 1, F) fp16, no activation: it is also the residual
                r_pre = x_pre                                                  # the reference chain's pre-activation
            else:                                                              # both chains through the same weights: one launch over 2P rows
                xr_pre = b.conv(f'head.fc{i}', xr, pc)
                x_pre, r_pre = xr_pre[:P], xr_pre[P:]
            ag = head.aggregator[i]
            nbk, Fc = ag.num_attention_blocks, x_pre.shape[-1]
            d = Fc // nbk
            ref_pre_s, r_embed_s, r_new_s = static_branch[i]
            ms = r_embed_s.shape[0]
            m_pad, kp = ops.round_up(ms + P, 32), ops.round_up(d, 32)
            kw = b.buf(nbk, m_pad, kp)
            vw = b.buf(nbk, ops.round_up(d, 16), m_pad)
            self._kv.append((kw, vw, ms, nbk, d, Fc, m_pad, kp))             # the static rows are packed by set_static
            if i == 0:                         # fc_embed, ref_fc_embed and ref_fc all read x_pre
                e3 = b.conv(f'head.agg{i}.embed3', x_pre, merged_pc(ag.fc_embed, ag.ref_fc_embed, ag.ref_fc))
                ld = e3.shape[-1]
                xe, xe_ld = e3[..., :Fc], ld
                re_l, rn_l, r_ld = e3[..., Fc:2 * Fc], e3[..., 2 * Fc:3 * Fc], ld
            else:
                xe, xe_ld = b.conv(f'head.agg{i}.fc_embed', x_pre, linear_pc(ag.fc_embed)), Fc
                e2 = b.conv(f'head.agg{i}.ref_embed2', r_pre, merged_pc(ag.ref_fc_embed, ag.ref_fc))
                re_l, rn_l, r_ld = e2[..., :Fc], e2[..., Fc:2 * Fc], e2.shape[-1]
            b.plan.add(f'head.agg{i}.pack_k', L.llv_selsa_pack_kv, (ops._ptr(re_l), ops._ptr(kw), P, ms, nbk, d, r_ld, m_pad, kp, 0))
            b.plan.add(f'head.agg{i}.pack_v', L.llv_selsa_pack_kv, (ops._ptr(rn_l), ops._ptr(vw), P, ms, nbk, d, r_ld, vw.shape[1], m_pad, 1))
            pck = ops.PackedConv(w=kw, bias=torch.zeros(nbk * m_pad, device=device), cin=d, cout=m_pad, w_rows=m_pad, k_pad=kp, kh=1, kw=1,
                                 stride=1, pad=0, groups=nbk, true_macs_per_pixel=(ms + P) * Fc)
            pcv = ops.PackedConv(w=vw, bias=torch.zeros(nbk * d, device=device), cin=m_pad, cout=d, w_rows=vw.shape[1], k_pad=m_pad, kh=1, kw=1,
                                 stride=1, pad=0, groups=nbk, true_macs_per_pixel=(ms + P) * Fc)
            s = conv_view(f'head.agg{i}.scores', xe.as_strided((1, P, 1, Fc), (P * xe_ld, xe_ld, xe_ld, 1)), xe_ld, pck, CONV_OUT_F32)
            pwt = b.buf(1, P, 1, nbk * m_pad)
            b.plan.add(f'head.agg{i}.softmax', L.llv_selsa_softmax, (ops._ptr(s), ops._ptr(pwt), P, nbk, m_pad, ms, P, ops._ptr(self.count),
                                                                    C.c_float(1.0 / d ** 0.5)))
            x_new = b.conv(f'head.agg{i}.values', pwt, pcv)                    # (1, P, 1, F)
            # relu(x_pre + fc(x_new)) = the residual + ReLU epilogue of the aggregator's last fc (selsa_bbox_head.py:78-80)
            if i + 1 < nf:
                xr = b.buf(2 * P, 1, 1, Fc)
                x = b.conv(f'head.agg{i}.fc', x_new.view(P, 1, 1, Fc), linear_pc(ag.fc), CONV_RELU | CONV_RES_ADD, res=x_pre.view(P, 1, 1, Fc),
                           out=xr[:P])
                # the reference chain's input to the next fc: relu(r_pre), element-wise on the rows already there (y * (y > 0) =
                # llv_relu_bwd with dy = y; bit-identical to the fc's ReLU epilogue, whose re-run of the 12544-wide first fc cost
                # 73 us per frame)
                b.plan.add(f'head.fc{i}.ref_relu', L.llv_relu_bwd,
                           (ops._ptr(r_pre), None, ops._ptr(r_pre), ops._ptr(xr[P:]), C.c_int64(r_pre.numel())), 0.0, 4.0 * r_pre.numel())
            else:
                x = b.conv(f'head.agg{i}.fc', x_new.view(P, 1, 1, Fc), linear_pc(ag.fc), CONV_RELU | CONV_RES_ADD, res=x_pre.view(P, 1, 1, Fc))
        ncl = head.num_classes
        cls_l = b.conv('head.fc_cls', x, linear_pc(head.fc_cls), CONV_OUT_F32)                  # (P, 1, 1, >= ncl + 1) fp32
        reg_l = b.conv('head.fc_reg', x, linear_pc(head.fc_reg), CONV_OUT_F32)                  # (P, 1, 1, 4 * ncl)
        assert reg_l.shape[-1] == 4 * ncl, 'class-specific deltas: 4 * num_classes is a multiple of 4'
        self.scores = b.buf(P, ncl + 1, dtype=torch.float32)
        b.plan.add('head.softmax', L.llv_softmax_rows_masked, (ops._ptr(cls_l), ops._ptr(self.scores), P, ncl + 1, cls_l.shape[-1], ncl + 1,
                                                               ops._ptr(self.count)))
        self.boxes = b.buf(P, 4 * ncl, dtype=torch.float32)
        f4 = C.c_float * 4
        bc = head.bbox_coder
        sf = img_meta.get('scale_factor', 1.0)
        import numpy as np
        sfv = np.asarray(sf, dtype=np.float32).reshape(-1)
        sfv = np.repeat(sfv, 4) if sfv.size == 1 else sfv[:4]
        self._means, self._stds, self._sf = f4(*[float(v) for v in bc.means]), f4(*[float(v) for v in bc.stds]), f4(*[float(v) for v in sfv])
        clip = 1 if bc.clip_border else 0
        b.plan.add('head.delta2bbox', L.llv_delta2bbox, (ops._ptr(self.boxes4), ops._ptr(reg_l), ops._ptr(self.boxes), P, ncl, self._means, self._stds,
                                                         C.c_float(16 / 1000), C.c_float(float(img_shape[0]) if clip else 0.0),
                                                         C.c_float(float(img_shape[1]) if clip else 0.0), clip, self._sf if rescale else None))
        tc = dict(rh.test_cfg)
        nms2 = tc['nms']
        self.out = ops.NmsBuffers(P, ncl, tc['max_per_img'], device)
        b.plan.keep += [self.out, self._means, self._stds, self._sf]
        b.plan.add('head.nms', L.llv_multiclass_nms_batched_ex,
                   (ops._ptr(self.boxes), ops._ptr(self.scores), None, 1, 1, P, ncl, ncl + 1, C.c_float(tc['score_thr']),
                    C.c_float(nms2.get('iou_threshold', nms2.get('iou_thr'))), tc['max_per_img'], ops._ptr(self.out.dets), ops._ptr(self.out.labels),
                    ops._ptr(self.out.count), ops._ptr(self.out.keep_flat), ops._ptr(self.out.ws), C.c_size_t(self.out.ws_bytes)))
        k = tc['max_per_img']
        self.res_host = torch.zeros((1, k, 6), dtype=torch.float32).pin_memory()
        self.cnt_host = torch.zeros((1,), dtype=torch.int32).pin_memory()
        b.plan.keep += [self.res_host, self.cnt_host]
        b.plan.add('results', L.llv_pack_results, (ops._ptr(self.out.dets), ops._ptr(self.out.labels), ops._ptr(self.out.count), 1, k,
                                                   C.c_void_p(self.res_host.data_ptr()), C.c_void_p(self.cnt_host.data_ptr())))
        self.guard_host = torch.zeros((1,), dtype=torch.int32).pin_memory()
        b.plan.keep.append(self.guard_host)
        b.plan.add('guards', L.llv_guard_trips_to, (C.c_void_p(self.guard_host.data_ptr()),))
        self.num_classes = ncl
        self._graph = None
        self.use_graph = use_graph
        self.static_rows = tuple(k[2] for k in self._kv)
        self.set_static(static_branch)

    def set_static(self, static_branch):
        """The reference frames of a (new) video: their rows of the packed attention operands (the graph reads the same buffers)."""
        L = lib()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for (kw, vw, ms, nbk, d, Fc, m_pad, kp), (_, r_embed_s, r_new_s) in zip(self._kv, static_branch):
            if r_embed_s.shape[0] != ms:
                raise LlvError(f'SelsaKeyPlan.set_static: {r_embed_s.shape[0]} static reference rows, the plan was built for {ms}')
            re16, rn16 = r_embed_s.to(torch.float16).contiguous(), r_new_s.to(torch.float16).contiguous()
            check(L.llv_selsa_pack_kv(ops._ptr(re16), ops._ptr(kw), ms, 0, nbk, d, Fc, m_pad, kp, 0, st), 'llv_selsa_pack_kv')
            check(L.llv_selsa_pack_kv(ops._ptr(rn16), ops._ptr(vw), ms, 0, nbk, d, Fc, vw.shape[1], m_pad, 1, st), 'llv_selsa_pack_kv')
        torch.cuda.current_stream().synchronize()

    def run(self, img):
        """img (1, 3, H, W) fp32 normalised on the device -> (dets (k, 5), labels (k,)) on the host."""
        ops.pack_input(img.float().contiguous(), 8, out=self.x_in)
        if self.use_graph and self._graph is None:
            self.plan.run()                      # warm-up (one-time attribute calls), then capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.plan.run()
            self._graph = g
        if self._graph is not None:
            self.plan.epoch += 1
            self._graph.replay()
        else:
            self.plan.run()
        torch.cuda.current_stream().synchronize()
        if int(self.guard_host[0]):              # the step's last kernel wrote the capacity-guard count next to the results
            ops.raise_on_guard_trips('SelsaKeyPlan.run')
        k = int(self.cnt_host[0])
        r = self.res_host[0, :k]
        return r[:, :5].clone(), r[:, 5].long()
