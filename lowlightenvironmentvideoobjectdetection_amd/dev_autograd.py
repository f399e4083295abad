"""Module-level differentiable device operations: the training path of the authors' DCN-based temporal fusion modules (SURVEY.md
section 8f item 3) -- DenoisingAggregator / Denoising2Aggergator (mmtracking/mmtrack/models/aggregators/denoising_aggregator.py:155-200,
denoising2_aggregator.py:157-244), the plugin stages of InsertResNet (backbones/insert_resnet.py) and DarkResNet's ConvLSTM
DenoisingBlock (backbones/dark_resnet.py:226-285).

The dense detectors train through ONE launch tape (train_engine.TrainGraph); these modules have data-dependent structure the tape does
not carry (dense concatenations, the N x N frame pairs of the temporal fusion, a recurrence over the frames), so here every
GEMM-shaped step is a torch.autograd.Function of its own whose forward AND backward are libllv.so launches on NHWC fp16 tensors:

  conv2d      llv_conv2d_nhwc_fwd                  | llv_conv2d_nhwc_wgrad + llv_unpack_wgrad, llv_conv2d_nhwc_dgrad (re-packed filter)
  mdcn        llv_modulated_deform_im2col + 1x1 conv | 1x1 dgrad / wgrad over the column tensor + llv_modulated_deform_col2im
              (mmcv modulated_deform_conv2d's three gradients: input, offsets, mask)
  lstm_gate   llv_convlstm_gate                    | llv_convlstm_gate_bwd

and the glue between them (ReLU masks, skip additions, channel concatenation, the softmax over frames) is element-wise torch on the
device under autograd. Activations and activation gradients are fp16, as in the tape; gradients travel multiplied by GRAD_SCALE (the
reference's fp16 mode: Fp16OptimizerHook(loss_scale=512.)): `to_dev` / `from_dev` apply and remove the factor at the module boundary,
and every parameter gradient is divided by it in fp32 before it is handed to autograd. No CPU path."""
import ctypes as C

import torch

from . import backward as B
from . import ops
from ._lib import CONV_OUT_F32, CONV_PIXSHUF, LlvError, check, lib

GRAD_SCALE = 512.0


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise LlvError('training of the denoising modules runs on the MI355X (conv / deformable-conv kernels); no CPU path')


class _ToDev(torch.autograd.Function):
    """(N, C, H, W) fp32 -> (N, H, W, Cp) fp16; backward removes GRAD_SCALE."""

    @staticmethod
    def forward(ctx, x):
        ctx.c = x.shape[1]
        return ops.pack_input(x.float().contiguous(), ops.round_up(x.shape[1], 8))

    @staticmethod
    def backward(ctx, g):
        return ops.unpack_output(g.contiguous(), ctx.c) * (1.0 / GRAD_SCALE)


class _FromDev(torch.autograd.Function):
    """(N, H, W, Cp) fp16 -> (N, C, H, W) fp32; backward applies GRAD_SCALE."""

    @staticmethod
    def forward(ctx, y, c):
        ctx.cp = y.shape[-1]
        return ops.unpack_output(y.contiguous(), c)

    @staticmethod
    def backward(ctx, g):
        return ops.pack_input((g.float() * GRAD_SCALE).contiguous(), ctx.cp), None


def to_dev(x):
    _need_gpu(x)
    return _ToDev.apply(x)


def from_dev(y, c):
    return _FromDev.apply(y, c)


def _geometry(w, cin_pad):
    co, ci, kh, kw = w.shape
    cout_store = ops.round_up(co, 4)
    return cout_store, ops.round_up(cout_store, 16), ops.round_up(kh * kw * cin_pad, 32)


def _packed_fwd(w, b, bn, cin_pad, stride, pad, eps):
    """The launch's packed weights, filled on the device from the live parameters (llv_pack_conv_weights / llv_fold_bias)."""
    co, ci, kh, kw = w.shape
    cout_store, w_rows, k_pad = _geometry(w, cin_pad)
    pc = ops.PackedConv(w=torch.zeros(1, w_rows, k_pad, dtype=torch.float16, device=w.device),
                        bias=torch.zeros(cout_store, dtype=torch.float32, device=w.device), cin=cin_pad, cout=cout_store, w_rows=w_rows,
                        k_pad=k_pad, kh=kh, kw=kw, stride=stride, pad=pad, groups=1, true_macs_per_pixel=co * ci * kh * kw)
    gamma, beta, mean, var = bn if bn is not None else (None, None, None, None)
    B.pack_weights(w.detach().contiguous(), gamma, var, eps, pc.w[0], cpad=cin_pad, k_pad=k_pad, row0=0, mode=0)
    B.fold_bias(None if b is None else b.detach(), gamma, beta, mean, var, eps, pc.bias[:co])
    return pc


def _packed_dgrad(w, bn, cin_pad, dy_ld, stride, pad, eps):
    co, ci, kh, kw = w.shape
    KH, KW, pd, _, _ = B.dgrad_geometry(kh, kw, stride, pad)
    rows = cin_pad * (4 if stride == 2 else 1)
    w_rows, k_pad = ops.round_up(rows, 16), ops.round_up(KH * KW * dy_ld, 32)
    pcd = ops.PackedConv(w=torch.zeros(1, w_rows, k_pad, dtype=torch.float16, device=w.device),
                         bias=torch.zeros(rows, dtype=torch.float32, device=w.device), cin=dy_ld, cout=rows, w_rows=w_rows, k_pad=k_pad,
                         kh=KH, kw=KW, stride=1, pad=pd, groups=1, true_macs_per_pixel=0)
    gamma, var = (bn[0], bn[3]) if bn is not None else (None, None)
    B.pack_weights(w.detach().contiguous(), gamma, var, eps, pcd.w[0], cpad=pcd.cin, k_pad=pcd.k_pad, row0=0, mode=2 if stride == 2 else 1,
                   pad=pad, KH=KH, KW=KW, padp=pd, rows=pcd.cout)
    return pcd


class _Conv(torch.autograd.Function):
    """y = [relu](conv(x, w [* bn scale]) + b'), NHWC fp16 in; fp16 out, or fp32 when out_f32 (the raw conv_offset output the
    deformable kernels read). bn = (gamma, beta, running_mean, running_var): an eval-mode BatchNorm folded into the launch, whose
    affine parameters train (resnet.py norm_eval=True)."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, mean, var, stride, pad, relu, out_f32, eps):
        _need_gpu(x, w)
        n, h, wd, cp = x.shape
        co, ci, kh, kw = w.shape
        if cp != ops.round_up(ci, 8) or x.dtype != torch.float16:
            raise LlvError(f'dev_autograd.conv2d: input of {cp} fp16 channels for a weight of {ci}')
        bn = (gamma.detach(), beta.detach(), mean, var) if gamma is not None else None
        pc = _packed_fwd(w, b, bn, cp, stride, pad, eps)
        ho, wo = ops.conv_out_size(h, wd, pc)
        ld = ops.round_up(co, 8)
        y = torch.zeros((n, ho, wo, ld), device=x.device, dtype=torch.float32 if out_f32 else torch.float16)
        from ._lib import CONV_RELU
        ops.conv2d(x.contiguous(), pc, flags=(CONV_RELU if relu else 0) | (CONV_OUT_F32 if out_f32 else 0), out=y)
        ctx.save_for_backward(x, w, b, gamma, beta, mean, var, y if relu else None)
        ctx.geom = (stride, pad, relu, eps, ho, wo, ld)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b, gamma, beta, mean, var, y = ctx.saved_tensors
        stride, pad, relu, eps, ho, wo, ld = ctx.geom
        n, h, wd, cp = x.shape
        co, ci, kh, kw = w.shape
        dy = dy.to(torch.float16).contiguous()
        if relu:                                   # into a new tensor: autograd may hand the same gradient object to another branch
            masked = torch.empty_like(dy)
            B.relu_bwd(dy, y, masked)
            dy = masked
        bn = (gamma.detach(), beta.detach(), mean, var) if gamma is not None else None
        cout_store, w_rows, k_pad = _geometry(w, cp)
        seg = (h, wd, ho, wo, 0, 0)
        gw = gb = gg = gbt = None
        if any(ctx.needs_input_grad[i] for i in (1, 2, 3, 4)):
            dw = torch.zeros(1, w_rows, k_pad, dtype=torch.float32, device=x.device)
            db = torch.zeros(cout_store, dtype=torch.float32, device=x.device)
            B.conv_wgrad(x, dy, dw, cin_pad=cp, cout=cout_store, w_rows=w_rows, k_pad=k_pad, kh=kh, kw=kw, stride=stride, pad=pad, groups=1,
                         batch=n, seg=seg, x_ld=cp, dy_ld=ld, db=db)
            gw = torch.empty_like(w, dtype=torch.float32)
            gb = torch.empty_like(b, dtype=torch.float32) if b is not None else None
            gg = torch.empty_like(gamma, dtype=torch.float32) if gamma is not None else None
            gbt = torch.empty_like(beta, dtype=torch.float32) if gamma is not None else None
            B.unpack_wgrad_device(dw[0], db, w.detach().contiguous(), bn[0] if bn else None, bn[2] if bn else None, bn[3] if bn else None, eps,
                                  gw, gg, gbt, gb, cpad=cp, k_pad=k_pad)
            inv = 1.0 / GRAD_SCALE
            gw, gb, gg, gbt = [None if t is None else t * inv for t in (gw, gb, gg, gbt)]
        dx = None
        if ctx.needs_input_grad[0]:
            pcd = _packed_dgrad(w, bn, cp, ld, stride, pad, eps)
            dx = torch.zeros((n, h, wd, cp), device=x.device, dtype=torch.float16)
            B.conv_dgrad(dy, pcd, dx, n, [seg], ld, cp, CONV_PIXSHUF if stride == 2 else 0, accumulate=False)
        return dx, gw, gb, gg, gbt, None, None, None, None, None, None, None


def conv2d(x, conv, relu=False, out_f32=False, bn=None, eps=1e-5):
    """`conv` = an nn.Conv2d parameter container (weight, bias, stride, padding); x NHWC fp16. bn: an nn.BatchNorm2d in eval mode."""
    s, p = conv.stride[0], conv.padding[0]
    if conv.dilation[0] != 1 or conv.groups != 1 or conv.stride[0] != conv.stride[1] or s not in (1, 2):
        raise LlvError('dev_autograd.conv2d: dense convolutions of stride 1 or 2 without dilation')
    if bn is not None:
        return _Conv.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, s, p, relu, out_f32, bn.eps)
    return _Conv.apply(x, conv.weight, conv.bias, None, None, None, None, s, p, relu, out_f32, eps)


class _Mdcn(torch.autograd.Function):
    """mmcv.ops.modulated_deform_conv2d (3x3, stride 1) with its three gradients. x (N, H, W, C) fp16; om (N, H, W, >= 3 dg K) fp32 = the raw
    conv_offset output (offsets, then mask logits: the sigmoid belongs to this op, as in llv_modulated_deform_im2col)."""

    @staticmethod
    def forward(ctx, x, om, w, b, dg, pad):
        _need_gpu(x, om, w)
        n, h, wd, c = x.shape
        co, ci, kh, kw = w.shape
        if ci != c:
            raise LlvError('dev_autograd.mdcn: the input must carry exactly the weight\'s channels (multiple of 8)')
        L = lib()
        K = kh * kw
        col = torch.empty((n, h, wd, K * c), device=x.device, dtype=torch.float16)
        check(L.llv_modulated_deform_im2col(ops._ptr(x), ops._ptr(om), ops._ptr(col), n, h, wd, c, c, om.shape[3], col.shape[3], kh, kw, 1, pad, 1,
                                            dg, h, wd, _stream()), 'llv_modulated_deform_im2col')
        # the weight as the 1x1 convolution over the column tensor: column (i*kw + j)*C + ch
        w2 = w.detach().permute(0, 2, 3, 1).reshape(co, K * c, 1, 1).contiguous()
        pc = _packed_fwd(w2, b, None, K * c, 1, 0, 1e-5)
        ld = ops.round_up(co, 8)
        y = torch.zeros((n, h, wd, ld), device=x.device, dtype=torch.float16)
        ops.conv2d(col, pc, out=y)
        ctx.save_for_backward(x, om, w, b, col, w2)
        ctx.geom = (dg, pad, ld)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, om, w, b, col, w2 = ctx.saved_tensors
        dg, pad, ld = ctx.geom
        n, h, wd, c = x.shape
        co, ci, kh, kw = w.shape
        K = kh * kw
        dy = dy.to(torch.float16).contiguous()
        cout_store, w_rows, k_pad = _geometry(w2, K * c)
        seg = (h, wd, h, wd, 0, 0)
        inv = 1.0 / GRAD_SCALE
        gw = gb = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            dw = torch.zeros(1, w_rows, k_pad, dtype=torch.float32, device=x.device)
            db = torch.zeros(cout_store, dtype=torch.float32, device=x.device)
            B.conv_wgrad(col, dy, dw, cin_pad=K * c, cout=cout_store, w_rows=w_rows, k_pad=k_pad, kh=1, kw=1, stride=1, pad=0, groups=1, batch=n,
                         seg=seg, x_ld=K * c, dy_ld=ld, db=db)
            gw2 = torch.empty_like(w2, dtype=torch.float32)
            gb = torch.empty_like(b, dtype=torch.float32) if b is not None else None
            B.unpack_wgrad_device(dw[0], db, w2, None, None, None, 1e-5, gw2, None, None, gb, cpad=K * c, k_pad=k_pad)
            gw = gw2.view(co, kh, kw, c).permute(0, 3, 1, 2).contiguous() * inv
            gb = None if gb is None else gb * inv
        dx = dom = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            pcd = _packed_dgrad(w2, None, K * c, ld, 1, 0, 1e-5)
            dcol = torch.zeros((n, h, wd, K * c), device=x.device, dtype=torch.float16)
            B.conv_dgrad(dy, pcd, dcol, n, [seg], ld, K * c, 0, accumulate=False)
            dx32 = torch.zeros((n, h, wd, c), device=x.device, dtype=torch.float32)
            dom = torch.zeros_like(om)
            check(lib().llv_modulated_deform_col2im(ops._ptr(dcol), ops._ptr(x), ops._ptr(om), ops._ptr(dx32), ops._ptr(dom), n, h, wd, c, c,
                                                    om.shape[3], dcol.shape[3], kh, kw, 1, pad, 1, dg, h, wd, _stream()), 'llv_modulated_deform_col2im')
            dx = dx32.to(torch.float16)
        return dx, dom, gw, gb, None, None


def modulated_dcn_pack(pack, x, extra_feat):
    """ModulatedDCNPack.forward (denoising_aggregator.py:103-111): conv_offset(extra_feat) -> offsets | mask logits -> deformable conv of x."""
    om = conv2d(extra_feat, pack.conv_offset, out_f32=True)
    return _Mdcn.apply(x, om, pack.weight, pack.bias, pack.deform_groups, pack.padding)


# ------------------------------------------------------------------------------------------------ the authors' modules
def rdb_forward(rdb, x):
    """RDB.forward (denoising_aggregator.py:59-66): dense layers cat([x, relu(conv(x))]) then x + lff(.)."""
    y = x
    for layer in rdb.layers:
        y = torch.cat([y, conv2d(y, layer.conv, relu=True)], dim=-1)
    return x + conv2d(y, rdb.lff)


def taf_forward(taf, x):
    """TemporalAttentionFusion.forward (denoising_aggregator.py:135-152) over the N frames of x (N, H, W, C)."""
    x = conv2d(x, taf.conv1, relu=True)
    n = x.shape[0]
    feat = []
    for i in range(n):
        x_ref = x[i:i + 1].expand(n, -1, -1, -1)
        x_set = conv2d(torch.cat([x, x_ref], dim=-1).contiguous(), taf.offset_conv)
        x_dcn = modulated_dcn_pack(taf.dcn_pack, x, x_set)
        x_cor = (x_dcn * x_ref).contiguous()
        for conv in taf.emb_conv:
            x_cor = conv2d(x_cor, conv)
        wgt = torch.softmax(x_cor.float(), dim=0)
        feat.append(torch.sum(wgt * x.float(), dim=0, keepdim=True).to(torch.float16))
    return conv2d(torch.cat(feat, dim=0).contiguous(), taf.conv2, relu=True)


def denoising_aggregator_forward(m, x):
    """DenoisingAggregator.forward (denoising_aggregator.py:186-200) on NHWC fp16."""
    identity = x
    x = conv2d(x, m.conv1, relu=True)
    x_f = x
    if m.with_rdb:
        for r in m.rdbs:
            x = rdb_forward(r, x)
    x = conv2d((x + x_f).contiguous(), m.conv2, relu=True)
    x_m = x
    if m.with_taf:
        x = taf_forward(m.taf, x)
    x = conv2d((x + x_m).contiguous(), m.conv3, relu=True)
    return x + identity


def denoising2_aggregator_forward(m, x_noise, last):
    """Denoising2Aggergator.forward (denoising2_aggregator.py:214-244) on NHWC fp16: -> (list of x_noise_out, all_x[-1] + last stage's output)."""
    outs, x_list = [], []
    for i in range(m.num_stage):
        l = m.layer_name[i]
        f = x_noise[0] if i == 0 else torch.cat([x_noise[i], x_list[i - 1]], dim=-1).contiguous()
        x = conv2d(f, m.layers[l + '_conv1'])
        if m.with_rdb[i]:
            for r in m.layers[l + '_rdb']:
                x = rdb_forward(r, x)
        if m.with_taf[i]:
            x = taf_forward(m.layers[l + '_taf'], x)
        outs.append(x + x_noise[i])
        x_list.append(conv2d((x if i == m.num_stage - 1 else x + x_noise[i]).contiguous(), m.layers[l + '_conv2']))
    return outs, last + x_list[-1]


class _LstmGate(torch.autograd.Function):
    """One frame of DarkResNet's ConvLSTM (dark_resnet.py:226-246,267-269): gates (1, H, W, 4P) fp16, c_prev (H*W, P) fp32 ->
    h (fp16), act = relu(norm2(h)) (fp16), c (fp32). scale / shift: the eval-mode BatchNorm as an affine, gradients returned for both."""

    @staticmethod
    def forward(ctx, gates, c_prev, scale, shift):
        n, h, w, p4 = gates.shape
        p = p4 // 4
        c = c_prev.detach().clone()
        hh = torch.empty((n, h, w, p), device=gates.device, dtype=torch.float16)
        act = torch.empty_like(hh)
        check(lib().llv_convlstm_gate(ops._ptr(gates), ops._ptr(c), ops._ptr(scale), ops._ptr(shift), ops._ptr(hh), ops._ptr(act), n * h * w, p, p4,
                                      p, p, _stream()), 'llv_convlstm_gate')
        ctx.save_for_backward(gates, c_prev, scale, shift)
        return hh, act, c

    @staticmethod
    def backward(ctx, dh, dact, dc):
        gates, c_prev, scale, shift = ctx.saved_tensors
        n, h, w, p4 = gates.shape
        p = p4 // 4
        dgates = torch.empty_like(gates)
        dc_prev = torch.empty_like(c_prev)
        dscale, dshift = torch.zeros_like(scale), torch.zeros_like(shift)
        dh = None if dh is None else dh.to(torch.float16).contiguous()
        dact = torch.zeros((n, h, w, p), device=gates.device, dtype=torch.float16) if dact is None else dact.to(torch.float16).contiguous()
        dc = None if dc is None else dc.float().contiguous()
        check(lib().llv_convlstm_gate_bwd(ops._ptr(gates), ops._ptr(c_prev), ops._ptr(scale), ops._ptr(shift), ops._ptr(dh), ops._ptr(dact),
                                          ops._ptr(dc), ops._ptr(dgates), ops._ptr(dc_prev), ops._ptr(dscale), ops._ptr(dshift), n * h * w, p, p4, p, p,
                                          _stream()), 'llv_convlstm_gate_bwd')
        inv = 1.0 / GRAD_SCALE
        return dgates, dc_prev, dscale * inv, dshift * inv


def lstm_gate(gates, c_prev, scale, shift):
    return _LstmGate.apply(gates, c_prev, scale, shift)


# ------------------------------------------------------------------------------------------------ the backbones around them
class _MaxPool(torch.autograd.Function):
    """3x3 / stride-2 / pad-1 max pooling (the ResNet stem, resnet.py:570): llv_maxpool3x3s2_nhwc | llv_maxpool3x3s2_nhwc_bwd."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.maxpool3x3s2(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        n, h, w, c = x.shape
        dy = dy.to(torch.float16).contiguous()
        dx = torch.zeros_like(x)
        check(lib().llv_maxpool3x3s2_nhwc_bwd(ops._ptr(x), ops._ptr(dy), ops._ptr(dx), n, h, w, c, c, c, c, _stream()), 'llv_maxpool3x3s2_nhwc_bwd')
        return dx


def _bottleneck_forward(blk, x):
    """Bottleneck.forward (resnet.py:260-299), every BatchNorm in eval mode and folded into its convolution."""
    identity = x
    out = conv2d(x, blk.conv1, relu=True, bn=blk.bn1)
    out = conv2d(out, blk.conv2, relu=True, bn=blk.bn2)
    out = conv2d(out, blk.conv3, bn=blk.bn3)
    if blk.downsample is not None:
        identity = conv2d(x, blk.downsample[0], bn=blk.downsample[1])
    return torch.relu(out + identity)


def _basic_block_forward(blk, x):
    """BasicBlock.forward (resnet.py:63-88)."""
    identity = x
    out = conv2d(x, blk.conv1, relu=True, bn=blk.bn1)
    out = conv2d(out, blk.conv2, bn=blk.bn2)
    if blk.downsample is not None:
        identity = conv2d(x, blk.downsample[0], bn=blk.downsample[1])
    return torch.relu(out + identity)


def _denoising_block_forward(blk, x):
    """DenoisingBlock.forward (dark_resnet.py:226-285): conv1 + norm1 + ReLU over all frames; a ConvLSTM along the frames of the batch
    (per frame conv2 over [x_t | h_{t-1}] -> four gates, llv_convlstm_gate: c, h, act = relu(norm2(h))); conv3 + norm3 + identity + ReLU."""
    identity = x
    t1 = conv2d(x, blk.conv1, relu=True, bn=blk.bn1)
    n, h, w, p = t1.shape
    if blk.conv2.stride[0] != 1 and n > 1:
        raise LlvError("DenoisingBlock with the stride on conv2 and more than one frame: the reference's torch.cat([x, h_cur]) (dark_resnet.py:233) "
                       "meets two resolutions and raises; use style='caffe' or stride 1")
    bn = blk.bn2
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias - bn.running_mean * scale
    ho, wo = (h - 1) // blk.conv2.stride[0] + 1, (w - 1) // blk.conv2.stride[0] + 1
    hcur = torch.zeros((1, h, w, p), device=x.device, dtype=torch.float16)
    c = torch.zeros((ho * wo, p), device=x.device, dtype=torch.float32)
    acts = []
    for t in range(n):
        gates = conv2d(torch.cat([t1[t:t + 1], hcur], dim=-1).contiguous(), blk.conv2)
        hcur, act, c = lstm_gate(gates, c, scale.float().contiguous(), shift.float().contiguous())
        acts.append(act)
    out = conv2d(torch.cat(acts, dim=0).contiguous(), blk.conv3, bn=blk.bn3)
    if blk.downsample is not None:
        identity = conv2d(x, blk.downsample[0], bn=blk.downsample[1])
    return torch.relu(out + identity)


def resnet_forward(bb, img):
    """ResNet / InsertResNet / DarkResNet .forward in training (resnet.py:622-640, insert_resnet.py:93-97, dark_resnet.py:386-709):
    img (N, 3, H, W) fp32 normalised frames (the batch is the clip for the temporal modules) -> tuple of NCHW fp32 stage outputs that carry
    the graph. BatchNorm uses running statistics (norm_eval=True, the setting of every reference config) with trainable affine parameters;
    frozen stages stay frozen through requires_grad."""
    from .backbones import BasicBlock, Bottleneck, DenoisingBlock
    _need_gpu(img)
    if not bb.norm_eval:
        raise LlvError('backbone training path: norm_eval=True (BatchNorm on running statistics, as the reference configs set)')
    if any(d != 1 for d in bb.dilations):
        raise LlvError('backbone training path (dev_autograd): dilated stages are trained through the launch tape (two_stage_train.py), not here')
    x = to_dev(img)
    x = conv2d(x, bb.conv1, relu=True, bn=bb.bn1)
    x = _MaxPool.apply(x)
    outs = []
    plugins = getattr(bb, '_plugins', {})
    for i, name in enumerate(bb.res_layers):
        for blk in getattr(bb, name):
            if isinstance(blk, DenoisingBlock):
                x = _denoising_block_forward(blk, x)
            elif isinstance(blk, Bottleneck):
                x = _bottleneck_forward(blk, x)
            elif isinstance(blk, BasicBlock):
                x = _basic_block_forward(blk, x)
            elif i in plugins and blk is plugins[i]:
                x = denoising_aggregator_forward(blk, x)
            else:
                raise LlvError(f'backbone training path: block {type(blk).__name__} is not built')
        if i in bb.out_indices:
            outs.append(from_dev(x, x.shape[-1]))
    return tuple(outs)
