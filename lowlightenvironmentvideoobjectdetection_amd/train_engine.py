"""Training step of the single-stage detector on the device kernels.

  SingleStageDetector.forward_train          mmdetection/mmdet/models/detectors/single_stage.py:68-96
  BaseDetector._parse_losses / train_step    mmdetection/mmdet/models/detectors/base.py:185-253
  ResNet.train / _freeze_stages (norm_eval)  mmdetection/mmdet/models/backbones/resnet.py:573-590,642-651
  SGD(lr, momentum 0.9, weight_decay 1e-4)   mmdetection/configs/_base_/schedules/schedule_1x.py

Forward = the same launch list the inference plan uses (engine.build_resnet / build_fpn / build_retina_head), with
every activation kept. Backward walks that launch tape in reverse: ReLU mask, residual / top-down-add routing, weight
gradient (llv_conv2d_nhwc_wgrad), bias gradient, data gradient (forward kernels on re-packed weights). BatchNorm is in
eval mode (running statistics, folded into the conv) with trainable affine parameters, exactly as norm_eval=True trains
in the reference: d(gamma), d(beta) and d(W) are recovered from the folded conv's gradients.

Gradients are fp16 with a static loss scale (the reference's fp16 mode: Fp16OptimizerHook(loss_scale=512.)), weight
gradients accumulate in fp32. The autograd bridge (`DetectorFunction`) makes the whole thing look like one differentiable
op to torch: `loss.backward()` fills `param.grad` of the module, so the reference's runner / OptimizerHook / DDP work
unchanged; `TrainStep.step_fused()` is the all-device path `bench.py --mode train` measures.
"""
import contextlib
import ctypes as C
import os
from collections import OrderedDict
from collections.abc import Mapping
from typing import Dict, List

import torch
import torch.distributed as dist

from . import backward as B
from . import engine, ops
from ._lib import LlvError, CONV_ACT_FIRST, CONV_LEAKY, CONV_RELU, CONV_RES_ADD, CONV_RES_UPS, CONV_PIXSHUF, CONV_OUT_F32


def _sptr(t: torch.Tensor) -> int:
    return t.untyped_storage().data_ptr()


class TrainGraph:
    """Forward launch list + everything the backward pass needs, for one (batch, H, W)."""

    def __init__(self, det, batch: int, height: int, width: int, device, loss_scale: float = 512.0, frozen_prefixes=None):
        head, neck, bb = det.bbox_head, det.neck, det.backbone
        self._init_common(det, dict(det.named_parameters()), dict(det.named_buffers()), device, batch, loss_scale)
        b = engine.Builder(device)
        self.builder = b
        sd = engine._cpu_sd(det.state_dict())
        self.x_in = b.buf(batch, height, width, 8)
        feats = engine.build_resnet(b, sd, self.x_in, bb.depth, bb.strides, bb.style, bb.out_indices, p='backbone.')
        self.pyramid = engine.build_fpn(b, sd, feats, neck.start_level, -1, neck.num_outs, neck.add_extra_convs,
                                        neck.relu_before_extra_convs, p='neck.')
        self.cls_buf, self.reg_buf = engine.build_retina_head(b, sd, self.pyramid, head.stacked_convs, True, p='bbox_head.',
                                                              pingpong=False)
        # frozen part: frozen_stages of the backbone (stem + layer1 .. layerK), resnet.py:573-590
        if frozen_prefixes is None:
            fs = getattr(bb, 'frozen_stages', -1)
            frozen_prefixes = (['stem.'] if fs >= 0 else []) + [f'backbone.layer{i}.' for i in range(1, fs + 1)]
        self._finish(b, frozen_prefixes)

    def _init_common(self, det, params, buffers, device, batch, loss_scale):
        self.det, self.device, self.batch, self.loss_scale = det, device, batch, float(loss_scale)
        self.params: Dict[str, torch.Tensor] = params
        self.buffers: Dict[str, torch.Tensor] = buffers
        for t in list(self.params.values()) + list(self.buffers.values()):
            if t.device.type != 'cuda':
                raise LlvError('training runs on the MI355X: move the detector to the GPU first (det.cuda()); no CPU fallback')

    def _finish(self, b, frozen_prefixes):
        """Everything the backward pass needs, from the builder's tape. Besides convolution launches the tape may hold CUSTOM
        entries (dict(name, custom=True, xs=[input buffers], y=output buffer, bwd=callable(graph, op))): parameter-free ops between
        the convolutions (feature warp, temporal aggregation) whose forward launches already sit in the plan and whose backward is
        the callable, which ADDS into the gradient twins of its inputs (self.gview)."""
        device = self.device
        self.plan = b.plan
        self.tape = b.tape
        self.frozen_prefixes = tuple(frozen_prefixes)
        bases = {}
        for t in b.plan.keep:
            if isinstance(t, torch.Tensor) and t.device.type == 'cuda':
                bases.setdefault(_sptr(t), t)
        self.bases = bases
        # which tensors need a gradient
        req = set()
        for op in self.tape:
            if op.get('custom'):
                op['frozen'], op['need_dx'], op['res'] = True, False, None
            else:
                op['frozen'] = bool(self.frozen_prefixes) and op['name'].startswith(self.frozen_prefixes)
                if op['pc'].parts is None:
                    raise LlvError(f"{op['name']}: launch has no parameter provenance (training needs PackedConv.parts)")
            inp = any(_sptr(x) in req for x in op['xs']) or (op['res'] is not None and _sptr(op['res']) in req)
            if not op['frozen'] or inp:
                req.add(_sptr(op['y']))
        self.req = req
        # gradient buffers (fp16), one per activation buffer that needs one; fp32 logits get padded fp16 twins
        # (one fp16 arena, so that zero_act_grads is one fill instead of one per buffer)
        self.grads: Dict[int, torch.Tensor] = {}
        shapes = {}
        for op in self.tape:
            p = _sptr(op['y'])
            if p in req and p not in shapes:
                base = bases[p]
                shapes[p] = base.shape[:-1] + (ops.round_up(base.shape[-1], 8),) if base.dtype == torch.float32 else tuple(base.shape)
        sizes = {p: ops.round_up(int(torch.Size(sh).numel()), 64) for p, sh in shapes.items()}
        self._grad_arena = torch.zeros(sum(sizes.values()), dtype=torch.float16, device=device)
        o = 0
        for p, sh in shapes.items():
            self.grads[p] = self._grad_arena[o:o + int(torch.Size(sh).numel())].view(sh)
            o += sizes[p]
        # per-launch backward state
        live = [op for op in self.tape if not op.get('custom') and not op['frozen']]
        wsz = [ops.round_up(op['pc'].groups * op['pc'].w_rows * op['pc'].k_pad, 64) + ops.round_up(op['pc'].groups * op['pc'].cout, 64) for op in live]
        self._wgrad_arena = torch.zeros(sum(wsz), dtype=torch.float32, device=device)        # every dW / db: zero_param_grads is one fill
        o = 0
        for op, n in zip(live, wsz):
            pc = op['pc']
            nw = pc.groups * pc.w_rows * pc.k_pad
            op['dw'] = self._wgrad_arena[o:o + nw].view(pc.groups, pc.w_rows, pc.k_pad)
            o2 = o + ops.round_up(nw, 64)
            op['db'] = self._wgrad_arena[o2:o2 + pc.groups * pc.cout]
            o += n
        self._tables = {}
        for op in self.tape:
            if op.get('custom'):
                continue
            pc = op['pc']
            op['need_dx'] = any(_sptr(x) in req for x in op['xs'])
            if op['need_dx']:
                assert len(op['xs']) == 1 and pc.n_src == 1, 'multi-source launches are not on the training path'
                dyc = self._dy_ld(op) // pc.groups
                KH, KW, pd, _, _ = B.dgrad_geometry(pc.kh, pc.kw, pc.stride, pc.pad)
                dil = max(int(pc.dilation), 1)
                if dil > 1:                  # dilated stride-1 conv (the DC5 stage): the flipped filter keeps the dilation
                    if pc.stride != 1:
                        raise LlvError(f"{op['name']}: data gradient of a dilated stride-{pc.stride} conv is not built")
                    pd = dil * (pc.kh - 1) - pc.pad
                rows = pc.cin * (4 if pc.stride == 2 else 1)
                w_rows, k_pad = ops.round_up(rows, 16), ops.round_up(KH * KW * dyc, 32)
                op['pcd'] = ops.PackedConv(w=torch.zeros(pc.groups, w_rows, k_pad, dtype=torch.float16, device=device),
                                           bias=torch.zeros(pc.groups * rows, dtype=torch.float32, device=device), cin=dyc, cout=rows,
                                           w_rows=w_rows, k_pad=k_pad, kh=KH, kw=KW, stride=1, pad=pd, groups=pc.groups,
                                           true_macs_per_pixel=0, dilation=dil)
                op['dflags'] = CONV_PIXSHUF if pc.stride == 2 else 0
        self._plan_mask_fusion()
        self.trainable = self._trainable_keys()

    # A plain conv + ReLU whose output feeds stride-1 convolutions only: its output gradient is written by those launches' data
    # gradients, which take the ReLU mask in their epilogue (LLV_CONV_MASK_POS) -- no llv_relu_bwd pass over it (30 of the 43 mask
    # launches of a RetinaNet R-50 step). LLV_FUSE_RELU_MASK=0: every mask its own launch.
    fuse_masks = os.environ.get('LLV_FUSE_RELU_MASK', '1') != '0'

    def _plan_mask_fusion(self):
        def same_view(a, b):
            return _sptr(a) == _sptr(b) and a.storage_offset() == b.storage_offset() and a.shape == b.shape and a.stride() == b.stride()
        users: Dict[int, list] = {}
        for op in self.tape:
            for x in list(op['xs']) + ([op['res']] if not op.get('custom') and op.get('res') is not None else []):
                users.setdefault(_sptr(x), []).append(op)
        # activations whose gradient is also written from outside the tape (the feature map a RoI head hangs off): never fused
        fed = {_sptr(t) for t in (getattr(self, 'f', None), getattr(self, 'x_out', None)) if torch.is_tensor(t)}
        for a in self.tape:
            if not self.fuse_masks or a.get('custom') or _sptr(a['y']) not in self.req or a['y'].dtype != torch.float16 or _sptr(a['y']) in fed:
                continue
            fl = a['flags']
            if not (fl & CONV_RELU) or (fl & (CONV_LEAKY | CONV_RES_ADD | CONV_ACT_FIRST | CONV_PIXSHUF)):
                continue
            cons = users.get(_sptr(a['y']), [])
            ok = bool(cons)
            for b in cons:
                ok = ok and not b.get('custom') and len(b['xs']) == 1 and same_view(b['xs'][0], a['y']) and b.get('need_dx', False) \
                    and b.get('dflags', 0) == 0 and (b.get('res') is None or _sptr(b['res']) != _sptr(a['y'])) and b['pc'].stride == 1
            if ok:
                a['mask_fused'] = True
                for b in cons:
                    b['mask_dx'] = True
        # Identity blocks: out = relu(conv3(..) + x) with x = the previous block's output, itself relu(conv3 + ..), read by this block's
        # first conv and by nothing else. g(x) = mask(x) * (dgrad_conv1 + g(out)) then comes out of ONE launch: conv1's data gradient takes
        # g(out) as its residual and x's mask in the epilogue, written over g(x); neither block needs its mask + identity-add pass
        # (`relu_res_bwd`) for that edge. Stage boundaries (downsample / lateral consumers) keep theirs.
        prod = {}
        for op in self.tape:
            if not op.get('custom'):
                prod.setdefault(_sptr(op['y']), []).append(op)
        plain = lambda fl: (fl & CONV_RELU) and (fl & CONV_RES_ADD) and not (fl & (CONV_LEAKY | CONV_RES_UPS | CONV_ACT_FIRST | CONV_PIXSHUF))
        for c3 in self.tape:
            if not self.fuse_masks or c3.get('custom') or not plain(c3['flags']) or c3.get('res') is None or _sptr(c3['y']) not in self.req:
                continue
            x = c3['res']
            px = _sptr(x)
            if px not in self.req or px in fed or x.dtype != torch.float16 or len(c3['segs']) != 1 or len(prod.get(px, [])) != 1:
                continue
            p3 = prod[px][0]
            if not plain(p3['flags']) or not same_view(p3['y'], x) or len(p3['segs']) != 1 or x.shape != c3['y'].shape \
                    or not x.is_contiguous() or not c3['y'].is_contiguous():
                continue
            us = users.get(px, [])
            firsts = [b for b in us if not b.get('custom') and len(b['xs']) == 1 and same_view(b['xs'][0], x)]
            if len(us) != 2 or len(firsts) != 1 or c3 not in us or firsts[0] is c3:
                continue
            c1 = firsts[0]
            if not c1.get('need_dx', False) or c1.get('dflags', 0) != 0 or c1['pc'].stride != 1 or c1.get('res') is not None and _sptr(c1['res']) == px:
                continue
            c3['id_via_conv1'] = True             # the identity gradient of this block joins in conv1's launch
            c1['dx_res_of'] = c3['y']             # ... as the residual g(out), with x's mask, written over g(x)
            c1['mask_dx'] = True
            p3['grad_final'] = True               # g(x) arrives masked and complete at the block that produced x

    # ---- helpers -------------------------------------------------------------------------------
    def _dy_ld(self, op):
        return self.grads[_sptr(op['y'])].shape[-1] if _sptr(op['y']) in self.grads else ops.round_up(op['y'].shape[-1], 8)

    def gview(self, t: torch.Tensor) -> torch.Tensor:
        """The gradient twin of an activation tensor / view (same rows; fp32 logits map to the padded fp16 twin)."""
        base, g = self.bases[_sptr(t)], self.grads[_sptr(t)]
        if base.dtype == torch.float32:
            assert t.shape == base.shape
            return g
        return g.as_strided(t.shape, t.stride(), g.storage_offset() + t.storage_offset() - base.storage_offset())

    def _trainable_keys(self) -> List[str]:
        keys = []
        for op in self.tape:
            if op['frozen']:
                continue
            for q in op['pc'].parts:
                ks = [q['wkey'], q['bkey']] + (list(q['bn'][:2]) if q['bn'] else [])
                for k in ks:
                    if k is not None and k in self.params and self.params[k].requires_grad and k not in keys:
                        keys.append(k)
        return keys

    def _eff(self, q):
        """Effective (BN-folded) weight and bias of one part, fp32 on device (ops.pack_conv's folding)."""
        w = self.params[q['wkey']].detach()
        b = self.params[q['bkey']].detach() if q['bkey'] else None
        if q['bn']:
            gk, bk, mk, vk = q['bn']
            gamma, beta = self.params[gk].detach(), self.params[bk].detach()
            mean, var = self.buffers[mk], self.buffers[vk]
            inv = torch.rsqrt(var + q['eps'])
            s = gamma * inv
            we = w * s.view(-1, 1, 1, 1)
            be = beta + ((b if b is not None else 0.0) - mean) * s
            return we, be, (s, inv, mean)
        return w, (b if b is not None else torch.zeros(w.shape[0], device=w.device)), None

    def _bn_tensors(self, q):
        if not q['bn']:
            return None, None, None, None
        gk, bk, mk, vk = q['bn']
        return self.params[gk].detach(), self.params[bk].detach(), self.buffers[mk], self.buffers[vk]

    def _sig(self):
        return tuple(t.data_ptr() for t in self.params.values()) + tuple(t.data_ptr() for t in self.buffers.values())

    dgrad_stream = None      # TrainStep: the side stream the data-gradient filters are re-packed on (beside the forward pass)

    def _batched(self, key, record, extra_sig=()):
        """Run `record()` (a sequence of B.pack_weights / fold_bias / unpack_wgrad_device calls) as job tables: recorded once, then one
        launch per table for as long as no parameter, buffer or destination tensor has moved."""
        sig = self._sig() + tuple(extra_sig)
        tab = self._tables.get(key)
        if tab is None or tab.sig != sig:
            with B.JobRecorder() as rec:
                record()
            tab = self._tables[key] = B.JobTables(rec, self.device)
            tab.sig = sig
        tab.launch(self.dgrad_stream)

    def repack(self, only_ops=None, key='repack'):
        """Write the current parameter values into every trainable launch's packed weights (+ its dgrad filter):
        llv_pack_conv_weights / llv_fold_bias over job tables (two launches; round 2 issued two or three per parameter tensor).
        only_ops: the launches of one all-reduce bucket (TrainStep re-packs a bucket right behind its SGD update)."""
        if only_ops is None:
            self._batched(key, self._repack_each)
        else:
            self._batched(key, lambda: self._repack_each(only_ops))

    def _repack_each(self, ops_list=None):
        for op in (self.tape if ops_list is None else ops_list):
            if op.get('custom'):
                continue
            if op['frozen'] and not op['need_dx']:
                continue
            pc = op['pc']
            for q in pc.parts:
                w = self.params[q['wkey']].detach()
                b = self.params[q['bkey']].detach() if q['bkey'] else None
                gamma, beta, mean, var = self._bn_tensors(q)
                g, r0, co = q['group'], q['row0'], q['cout']
                if not op['frozen']:
                    B.pack_weights(w, gamma, var, q['eps'], pc.w[g], cpad=pc.cin, k_pad=pc.k_pad, row0=r0, mode=0)
                    B.fold_bias(b, gamma, beta, mean, var, q['eps'], pc.bias[g * pc.cout + r0:g * pc.cout + r0 + co])
                if op['need_dx']:
                    pcd = op['pcd']
                    B.pack_weights(w, gamma, var, q['eps'], pcd.w[g], cpad=pcd.cin, k_pad=pcd.k_pad, row0=r0,
                                   mode=2 if pc.stride == 2 else 1, pad=pc.pad, KH=pcd.kh, KW=pcd.kw, padp=pcd.pad, rows=pcd.cout)

    # ---- forward / backward ---------------------------------------------------------------------
    def forward(self, img: torch.Tensor):
        """img: (N,3,H,W) fp32 normalised -> per-level fp32 NHWC logits views (cls, reg)."""
        ops.pack_input(img.float().contiguous(), 8, out=self.x_in)
        self.plan.run()
        pyr = self.pyramid
        return [pyr.level(i, self.cls_buf) for i in range(len(pyr.level_hw))], [pyr.level(i, self.reg_buf) for i in range(len(pyr.level_hw))]

    def backward(self, dcls: List[torch.Tensor], dreg: List[torch.Tensor], on_op_done=None):
        """dcls / dreg: per-level (N,h,w,C) fp32 gradients of the loss w.r.t. the logits. Fills op['dw'] / op['db']
        (scaled by loss_scale). on_op_done(op) is called as soon as a launch's weight gradient is final (launches run
        in reverse order), which is what lets the gradient all-reduce overlap the rest of the backward pass."""
        self.zero_act_grads()
        self.set_logit_grads(dcls, dreg)
        self.backward_ops(list(reversed(self.tape)), on_op_done)

    def zero_act_grads(self):
        self._grad_arena.zero_()

    def set_logit_grads(self, dcls, dreg):
        gc, gr = self.grads[_sptr(self.cls_buf)], self.grads[_sptr(self.reg_buf)]
        pyr = self.pyramid
        for i in range(len(pyr.level_hw)):
            h, w = pyr.level_hw[i]
            o, n = pyr.level_off[i], self.batch * h * w
            gc[o:o + n, :dcls[i].shape[-1]] = (dcls[i].reshape(n, -1) * self.loss_scale).to(torch.float16)
            gr[o:o + n, :dreg[i].shape[-1]] = (dreg[i].reshape(n, -1) * self.loss_scale).to(torch.float16)

    def backward_ops(self, ops_rev, on_op_done=None, phase='all'):
        """phase 'data': everything but the weight gradients (ReLU masks, residual / top-down routing, data gradients); 'weight': the
        weight gradients alone. A launch's dW depends only on its input activation and on dY as the data phase left it, nothing later
        in the pass writes that dY again, and dW is not read before the unpack -- so TrainStep replays the weight phase of one
        all-reduce bucket on a second stream beside the data phase of the next (single-image launches that each fill part of the chip)."""
        if phase == 'weight' and self.group_wgrad:
            self._wgrad_group(ops_rev).launch()
            return
        for op in ops_rev:
            yp = _sptr(op['y'])
            if yp not in self.req:
                continue
            if op.get('custom'):
                if phase != 'weight':
                    op['bwd'](self, op)
                continue
            pc, flags = op['pc'], op['flags']
            gy_base = self.grads[yp]
            gy = self.gview(op['y'])
            dy_ld = gy_base.shape[-1]
            if phase == 'weight':
                flags = 0                                        # masks and routing belong to the data phase
            # rows of y this launch wrote: per segment (batch*ho*wo rows at y offset)
            fused_res = False
            if flags & CONV_RELU:
                gres = None
                if (flags & CONV_RES_ADD) and not (flags & (CONV_RES_UPS | CONV_LEAKY)) and op['res'] is not None and _sptr(op['res']) in self.req:
                    gres = self.gview(op['res'])
                if gres is not None and (op.get('grad_final') or op.get('id_via_conv1')):
                    # identity-block edges planned by _plan_mask_fusion: dY arrives masked (grad_final) and / or the identity gradient
                    # joins in conv1's data-gradient launch (id_via_conv1); what is left of the pass, if anything, is one of its halves
                    if not op.get('grad_final'):
                        B.relu_bwd(gy, op['y'], gy)
                    if not op.get('id_via_conv1'):
                        B.relu_bwd(gres, None, gres, dy2=gy)
                    fused_res = True
                elif gres is not None and len(op['segs']) == 1 and op['segs'][0][5] == 0 and gy.is_contiguous() and gres.is_contiguous() \
                        and op['y'].is_contiguous() and gy.numel() == gres.numel() == op['y'].numel():
                    B.relu_res_bwd(gy, op['y'], gres)            # mask + the identity's gradient in one pass (a Bottleneck's output)
                    fused_res = True
                elif op.get('mask_fused'):
                    pass                                         # the consumers' data gradients wrote dY masked (_plan_mask_fusion)
                else:
                    # rows this launch wrote, neighbouring levels of a pyramid buffer as one range (the head towers: 1 launch, not 5)
                    spans = []
                    for (h, w, ho, wo, xo, yo) in sorted(op['segs'], key=lambda s_: s_[5]):
                        n = op['batch'] * ho * wo
                        if spans and spans[-1][0] + spans[-1][1] == yo:
                            spans[-1][1] += n
                        else:
                            spans.append([yo, n])
                    for yo, n in spans:
                        yv, gv = self._rows(op['y'], yo, n), self._rows(gy, yo, n)
                        B.relu_bwd(gv, yv, gv)
            if flags & CONV_LEAKY:
                # LeakyReLU epilogue (Darknet / YOLOV3Neck ConvModules); with LLV_CONV_ACT_FIRST the residual joins AFTER the activation
                # (ResBlock: y = res + act(v)): the residual takes dY as it is, the conv's own dY gets the activation's slope
                res = op['res'] if (flags & CONV_RES_ADD) else None
                if res is not None and not (flags & CONV_ACT_FIRST):
                    raise LlvError(f"{op['name']}: backward of LeakyReLU(v + res) is not built (ResBlocks add after the activation)")
                gres = self.gview(res) if res is not None and _sptr(res) in self.req else None
                B.act_bwd(gy, op['y'], res, gres, op.get('act_slope', 0.0))
            elif (flags & CONV_RES_ADD) and not fused_res and op['res'] is not None and _sptr(op['res']) in self.req:
                gres = self.gview(op['res'])
                if flags & CONV_RES_UPS:
                    B.upsample_add_bwd(gy, gres, accumulate=True)
                else:
                    B.relu_bwd(gres, None, gres, dy2=gy)
            if not op['frozen'] and phase != 'data':
                B.conv_wgrad(op['xs'][0], gy, op['dw'], cin_pad=pc.cin, cout=pc.cout,
                             w_rows=pc.w_rows, k_pad=pc.k_pad, kh=pc.kh, kw=pc.kw, stride=pc.stride, pad=pc.pad, groups=pc.groups,
                             batch=op['batch'], segs=[tuple(sg) for sg in op['segs']], x_ld=op['x_ld'], dy_ld=dy_ld, db=op['db'],
                             dilation=max(int(pc.dilation), 1))
            if phase == 'weight':
                continue
            if op['need_dx']:
                gx = self.gview(op['xs'][0])
                if op['dflags'] & CONV_PIXSHUF:
                    for seg in op['segs']:
                        B.conv_dgrad(gy, op['pcd'], gx, op['batch'], [seg], dy_ld, op['x_ld'], op['dflags'], accumulate=True)
                else:
                    rsrc = op.get('dx_res_of')
                    B.conv_dgrad(gy, op['pcd'], gx, op['batch'], op['segs'], dy_ld, op['x_ld'], 0, accumulate=rsrc is None,
                                 mask=op['xs'][0] if op.get('mask_dx') else None, res=self.gview(rsrc) if rsrc is not None else None)
            if on_op_done is not None and not op['frozen']:
                on_op_done(op)

    # the weight phase of a list of launches as one launch per kernel kind (B.WgradGroup); LLV_WGRAD_GROUP=0: one launch per conv
    group_wgrad = os.environ.get('LLV_WGRAD_GROUP', '1') != '0'

    def _wgrad_group(self, ops_list):
        key = ('wgrad',) + tuple(id(op) for op in ops_list)
        grp = self._tables.get(key)
        if grp is None:
            calls = []
            for op in ops_list:
                yp = _sptr(op['y'])
                if yp not in self.req or op.get('custom') or op['frozen']:
                    continue
                pc = op['pc']
                calls.append(dict(x=op['xs'][0], dy=self.gview(op['y']), dw=op['dw'], db=op['db'], cin_pad=pc.cin, cout=pc.cout, w_rows=pc.w_rows,
                                  k_pad=pc.k_pad, kh=pc.kh, kw=pc.kw, stride=pc.stride, pad=pc.pad, groups=pc.groups, batch=op['batch'],
                                  segs=[tuple(sg) for sg in op['segs']], x_ld=op['x_ld'], dy_ld=self.grads[yp].shape[-1],
                                  dilation=max(int(pc.dilation), 1)))
            grp = self._tables[key] = B.WgradGroup(calls, self.device)
        return grp

    @staticmethod
    def _rows(t: torch.Tensor, off: int, n: int) -> torch.Tensor:
        """rows [off, off+n) of a pixel-major buffer, whatever its leading shape (full-buffer launches have off = 0)."""
        c = t.shape[-1]
        flat = t.as_strided((t.numel() // c, c), (t.stride(-2), 1), t.storage_offset()) if t.dim() > 2 else t
        return flat[off:off + n]

    def zero_param_grads(self):
        self._wgrad_arena.zero_()

    def param_grads_batched(self, ops_list, dst: Dict[str, torch.Tensor]):
        """param_grads for a list of launches as ONE llv_unpack_wgrad_batched call (`dst` must name a tensor for every parameter)."""
        live = [op for op in ops_list if not op.get('custom') and not op['frozen']]
        if not live:
            return

        def record():
            for op in live:
                self.param_grads(only_op=op, dst=dst)
        self._batched(('unpack',) + tuple(id(op) for op in live), record, extra_sig=tuple(t.data_ptr() for t in dst.values()))

    def param_grads(self, only_op=None, dst: Dict[str, torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Module-parameter gradients (fp32, still multiplied by loss_scale) from the packed dW / db of every launch
        (or of one launch): llv_unpack_wgrad, one launch per parameter tensor. `dst` maps parameter names to the tensors
        to write (TrainStep: views of its flat gradient buffer); missing ones are allocated."""
        out: Dict[str, torch.Tensor] = {}

        def slot(k):
            if k is None or k not in self.params or not self.params[k].requires_grad:
                return None
            assert k not in out, f'{k}: a parameter feeds exactly one launch on this path'
            out[k] = dst[k] if dst is not None and k in dst else torch.empty_like(self.params[k], dtype=torch.float32)
            return out[k]

        for op in ([only_op] if only_op is not None else self.tape):
            if op['frozen']:
                continue
            pc = op['pc']
            for q in pc.parts:
                g, r0, co = q['group'], q['row0'], q['cout']
                w = self.params[q['wkey']].detach()
                gamma, beta, mean, var = self._bn_tensors(q)
                gw, gb = slot(q['wkey']), slot(q['bkey'])
                gg, gbt = (slot(q['bn'][0]), slot(q['bn'][1])) if q['bn'] else (None, None)
                B.unpack_wgrad_device(op['dw'][g, r0:], op['db'][g * pc.cout + r0:], w, gamma, mean, var, q['eps'], gw, gg, gbt, gb,
                                      cpad=pc.cin, k_pad=pc.k_pad)
        return out


def _param_grads_keys(self, op):
    ks = []
    for q in op['pc'].parts:
        for k in [q['wkey'], q['bkey']] + (list(q['bn'][:2]) if q['bn'] else []):
            if k is not None and k in self.params and self.params[k].requires_grad and k not in ks:
                ks.append(k)
    return ks


TrainGraph.param_grads_keys = _param_grads_keys


class DetectorFunction(torch.autograd.Function):
    """The detector body (backbone + neck + head convolutions) as one differentiable op: inputs = the trainable
    parameters, outputs = per-level logits; backward runs TrainGraph.backward and returns the parameter gradients."""

    @staticmethod
    def forward(ctx, graph: TrainGraph, img: torch.Tensor, *params):
        graph.repack()
        cls, reg = graph.forward(img)
        ctx.graph = graph
        ctx.n_levels = len(cls)
        return tuple(t.clone() for t in cls) + tuple(t.clone() for t in reg)

    @staticmethod
    def backward(ctx, *grads):
        g: TrainGraph = ctx.graph
        n = ctx.n_levels
        g.zero_param_grads()
        g.backward([x.contiguous() for x in grads[:n]], [x.contiguous() for x in grads[n:]])
        pg = g.param_grads()
        inv = 1.0 / g.loss_scale
        return (None, None) + tuple((pg[k] * inv) if k in pg else None for k in g.trainable)


def forward_train(det, graph: TrainGraph, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
    """single_stage.py:68-96: x = extract_feat(img); losses = bbox_head.forward_train(x, ...)."""
    from .training import anchor_head_loss
    outs = DetectorFunction.apply(graph, img, *[graph.params[k] for k in graph.trainable])
    n = len(outs) // 2
    cls_scores = [t.permute(0, 3, 1, 2)[:, :det.bbox_head.num_anchors * det.bbox_head.cls_out_channels] for t in outs[:n]]
    bbox_preds = [t.permute(0, 3, 1, 2)[:, :det.bbox_head.num_anchors * 4] for t in outs[n:]]
    losses = anchor_head_loss(det.bbox_head, cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore)
    if losses is None:
        raise LlvError('no valid anchors inside the image')
    return losses


def plan_buckets(keys_per_op, slices, limit):
    """All-reduce buckets over a flat gradient buffer laid out in backward order. keys_per_op[i] = parameter names whose
    gradients are final once the i-th launch of the backward pass is done; slices[name] = (offset, numel). Returns
    {i: [(start, end), ...]}: the ranges that may be all-reduced right after launch i. Every element is covered exactly
    once, a bucket is at least `limit` elements (except the last) and never contains a gradient that is not final."""
    out, start, done = {}, 0, set()
    total = max((o + n for o, n in slices.values()), default=0)
    for i, ks in enumerate(keys_per_op):
        done.update(ks)
        end = max((slices[k][0] + slices[k][1] for k in done), default=0)
        # the prefix [0, end) is final only if every key inside it is done (backward-ordered layout guarantees it)
        if end - start >= limit:
            out.setdefault(i, []).append((start, end))
            start = end
    if start < total:
        out.setdefault(len(keys_per_op) - 1, []).append((start, total))
    return out


class LazyLogVars(Mapping):
    """BaseDetector._parse_losses' log_vars (base.py:226-261: per key the sum of the per-level means, 'loss' = sum of the keys with
    'loss' in the name, averaged over the ranks) as a read-on-demand mapping: the values stay on the device until somebody looks, so
    a training loop that logs every n-th step does not stall the other n - 1 (the all-reduce itself is queued by every rank at once)."""

    def __init__(self, vec, world):
        # the average over the ranks is a collective: it is issued HERE, by every rank, asynchronously (a rank that never looks at its
        # values must still take part -- reading them lazily on rank 0 alone would hang the other ranks' next collective)
        self._vec, self._world, self._vals, self._work = vec, world, None, None
        if world > 1:
            self._vec = vec.div(world)
            self._work = dist.all_reduce(self._vec, async_op=True)

    def _resolve(self):
        if self._vals is None:
            if self._work is not None:
                self._work.wait()
            v = self._vec.tolist()
            nl = len(v) // 2
            lc, lb = float(sum(v[:nl])), float(sum(v[nl:]))
            self._vals = OrderedDict(loss_cls=lc, loss_bbox=lb, loss=lc + lb)
        return self._vals

    def __getitem__(self, k):
        return self._resolve()[k]

    def __iter__(self):
        return iter(self._resolve())

    def __len__(self):
        return 3


class TrainStep:
    """All-device training step: forward, loss, backward, gradient all-reduce over the ranks (RCCL when the process
    group is nccl), fused SGD on one flat fp32 parameter buffer. The parameters of `det` become views of that buffer.

    The flat buffer is laid out in BACKWARD order (head first), so a prefix of it is final long before the backward
    pass ends: it is cut into buckets and each bucket's all-reduce is issued asynchronously as soon as its last
    parameter gradient has been written, overlapping the collective (xGMI) with the remaining convolution kernels.
    A bucket is also the unit of the two-stream backward pass (its weight-gradient phase runs beside the next bucket's data-gradient
    phase): 18 MB buckets (9 for the 151 MB of RetinaNet R-50's trainable gradients) measured 2-3 % above 25 MB (6) on one GPU."""

    def __init__(self, det, batch, height, width, device, lr=0.01, momentum=0.9, weight_decay=1e-4, loss_scale=512.0,
                 bucket_mb=18.0, use_graph=True, fused_loss=True):
        self.det, self.device = det, device
        self.use_graph, self._g_fwd = use_graph, None
        self.fused_loss, self._fused = fused_loss, None       # training.FusedHeadLoss where the head's losses allow it
        self._wstream = torch.cuda.Stream(device=device) if os.environ.get('LLV_WGRAD_STREAM', '1') != '0' else None
        # SGD + re-pack of an all-reduce bucket on the side stream right behind its weight phase (and its all-reduce), beside the data
        # phase of the next bucket, instead of one update over all parameters and one re-pack between the backward and the next forward
        # pass (0.3 ms of a 4.4 ms step on the critical path). LLV_SGD_PER_BUCKET=0: the update after the backward pass.
        self.per_bucket_update = use_graph and self._wstream is not None and os.environ.get('LLV_SGD_PER_BUCKET', '1') != '0'
        self.params_changed = True            # set it again after load_state_dict / an external update: the next step re-packs everything
        self.graph = TrainGraph(det, batch, height, width, device, loss_scale)
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        g = self.graph
        keys = []
        for op in reversed(g.tape):
            if op['frozen']:
                continue
            for k in g.param_grads_keys(op):
                if k not in keys:
                    keys.append(k)
        assert set(keys) == set(g.trainable)
        sizes = [g.params[k].numel() for k in keys]
        self.flat_w = torch.zeros(sum(sizes), dtype=torch.float32, device=device)
        self.flat_g = torch.zeros_like(self.flat_w)
        self.flat_m = torch.zeros_like(self.flat_w)
        self.slices, o = {}, 0
        for k, n in zip(keys, sizes):
            p = g.params[k]
            self.flat_w[o:o + n] = p.detach().reshape(-1)
            p.data = self.flat_w[o:o + n].view_as(p)
            self.slices[k] = (o, n)
            o += n
        # buckets: [start, end) ranges of the flat buffer, closed when the op owning their last key is done
        live = [op for op in reversed(g.tape) if not op['frozen']]
        plan = plan_buckets([g.param_grads_keys(op) for op in live], self.slices, int(bucket_mb * 1024 * 1024 / 4))
        self.bucket_after_op = {id(live[i]): rng for i, rng in plan.items()}
        self.buckets = [r for rng in plan.values() for r in rng]
        self.steps_done = 0
        self._works = []
        self.gviews = {k: self.flat_g[o:o + n].view_as(g.params[k]) for k, (o, n) in self.slices.items()}

    def _dist(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    # ---- HIP-graph capture: the launch sequences are static (same buffers every step), only the loss in the middle is
    # dynamic (numbers of positives). Forward = one graph; backward = one graph per all-reduce bucket, so each bucket's
    # collective can start while the next graph runs.
    def _capture(self, img_shape):
        g = self.graph
        self.img_static = torch.zeros(img_shape, dtype=torch.float32, device=self.device)
        segs, cur = [], []
        for op in reversed(g.tape):
            if _sptr(op['y']) not in g.req:          # frozen region: nothing to launch
                continue
            cur.append(op)
            if id(op) in self.bucket_after_op:
                segs.append((cur, self.bucket_after_op[id(op)]))
                cur = []
        if cur:
            segs.append((cur, []))
        self._segs = segs

        def fwd(repack=True):
            # the data-gradient filters (a transposing re-pack of all trainable weights, 150 us) are first read by the backward pass:
            # forked onto the side stream here, joined at the end of the forward graph
            cur = torch.cuda.current_stream()
            if repack:
                g.dgrad_stream = self._wstream
                try:
                    g.repack()
                finally:
                    g.dgrad_stream = None
            ops.pack_input(self.img_static, 8, out=g.x_in)
            g.plan.run()
            if repack and self._wstream is not None:
                cur.wait_stream(self._wstream)

        def seg_data(k):
            g.backward_ops(self._segs[k][0], None, phase='data')

        def seg_weight(k):
            if k == 0:
                g.zero_param_grads()
            g.backward_ops(self._segs[k][0], None, phase='weight')
            g.param_grads_batched(self._segs[k][0], self.gviews)

        # warm-up on a side stream (lazy initialisation, split-K workspaces, job tables), then capture
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fwd()
            g.zero_act_grads()
            for k in range(len(segs)):
                seg_data(k)
                seg_weight(k)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._g_fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_fwd):
            fwd(repack=not self.per_bucket_update)       # (per-bucket update: the weights were re-packed behind their SGD update)
        if self.per_bucket_update:
            # job tables of the per-bucket re-packs, built outside any capture
            for k in range(len(segs)):
                g.repack(only_ops=segs[k][0], key=('repack', k))
            torch.cuda.synchronize()
        self._g_bwd, self._g_wgt = [], []
        for k in range(len(segs)):
            for fn, lst in ((seg_data, self._g_bwd), (seg_weight, self._g_wgt)):
                gk = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gk):
                    fn(k)
                lst.append(gk)

    def step_fused(self, img, img_metas, gt_bboxes, gt_labels):
        from .training import anchor_head_loss
        g, head = self.graph, self.det.bbox_head
        if self.use_graph and self._g_fwd is None:
            self._capture(tuple(img.shape))
        if self.use_graph:
            if self.per_bucket_update and self.params_changed:
                g.repack()                                       # parameters set from outside since the last step
            self.params_changed = False
            self.img_static.copy_(img)
            self._g_fwd.replay()
            pyr = g.pyramid
            cls = [pyr.level(i, g.cls_buf) for i in range(len(pyr.level_hw))]
            reg = [pyr.level(i, g.reg_buf) for i in range(len(pyr.level_hw))]
        else:
            g.repack()
            cls, reg = g.forward(img)
        from .training import FusedHeadLoss
        if self.fused_loss and self._fused is None and FusedHeadLoss.eligible(head):
            self._fused = FusedHeadLoss(g, head, self.device)
        self._works = []
        if self.fused_loss and self._fused is not None:
            # device-only loss: no element-wise torch launches, no host synchronisation before the backward graphs are queued
            g.zero_act_grads()
            vec = self._fused.run(gt_bboxes, gt_labels, img_metas)
            nl = vec.shape[0] // 2
            losses = dict(loss_cls=[vec[i] for i in range(nl)], loss_bbox=[vec[nl + i] for i in range(nl)])
        else:
            lc = [t.detach().requires_grad_() for t in cls]
            lr_ = [t.detach().requires_grad_() for t in reg]
            losses = anchor_head_loss(head, [t.permute(0, 3, 1, 2) for t in lc], [t.permute(0, 3, 1, 2) for t in lr_],
                                      gt_bboxes, gt_labels, img_metas)
            if losses is None:
                raise LlvError('no valid anchors inside the image')
            total = sum(sum(v) if isinstance(v, list) else v for k, v in losses.items() if 'loss' in k)
            total.backward()
            g.zero_act_grads()
            g.set_logit_grads([t.grad for t in lc], [t.grad for t in lr_])
        if self.use_graph:
            # per all-reduce bucket: the data phase on this stream, its weight phase (+ unpack + the bucket's all-reduce) on the side
            # stream beside the NEXT bucket's data phase
            main, side = torch.cuda.current_stream(), self._wstream
            for k, (ops_k, buckets) in enumerate(self._segs):
                self._g_bwd[k].replay()
                if side is not None:
                    side.wait_stream(main)
                with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                    self._g_wgt[k].replay()
                    works = [dist.all_reduce(self.flat_g[a:b], async_op=True) for (a, b) in buckets] if self._dist() else []
                    if self.per_bucket_update:
                        for w in works:
                            w.wait()                              # (orders this stream behind the collective; the host does not block)
                        world = dist.get_world_size() if self._dist() else 1
                        for (a, b) in buckets:
                            B.sgd_momentum(self.flat_w[a:b], self.flat_g[a:b], self.flat_m[a:b], self.lr, self.momentum, self.weight_decay,
                                           1.0 / (g.loss_scale * world), self.steps_done == 0)
                        if buckets:
                            g.repack(only_ops=ops_k, key=('repack', k))
                    else:
                        self._works += works
            if side is not None:
                main.wait_stream(side)
        else:
            g.zero_param_grads()
            g.backward_ops(list(reversed(g.tape)), self._op_done)
        for w in self._works:
            w.wait()
        world = dist.get_world_size() if self._dist() else 1
        if not (self.use_graph and self.per_bucket_update):
            B.sgd_momentum(self.flat_w, self.flat_g, self.flat_m, self.lr, self.momentum, self.weight_decay,
                           1.0 / (g.loss_scale * world), self.steps_done == 0)
        self.steps_done += 1
        if self.fused_loss and self._fused is not None:
            return dict(loss=vec.sum(), log_vars=LazyLogVars(vec.clone(), world), num_samples=len(img_metas))
        loss, log_vars = self.det._parse_losses({k: [x.detach() for x in v] if isinstance(v, list) else v.detach() for k, v in losses.items()})
        return dict(loss=loss, log_vars=log_vars, num_samples=len(img_metas))

    def _op_done(self, op):
        self.graph.param_grads(only_op=op, dst=self.gviews)
        if self._dist():
            for (a, b) in self.bucket_after_op.get(id(op), []):
                self._works.append(dist.all_reduce(self.flat_g[a:b], async_op=True))
