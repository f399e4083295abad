"""Backward pass of the convolution engine: weight / data gradients on the device kernels.

Replaces torch autograd's conv backward under mmdet's ResNet / FPN / RetinaHead in ``BaseDetector.train_step``
(mmdetection/mmdet/models/detectors/base.py:220-253; resnet.py:260-299, fpn.py:170-221, retina_head.py:93-114).

* wgrad  -> ``llv_conv2d_nhwc_wgrad`` (csrc/conv_bwd.hip): pixel-reduction GEMM on MFMA, fp32 accumulation into the
  packed weight layout; :func:`unpack_wgrad` maps it back to the module's ``(Cout, Cin, kh, kw)``.
* dgrad  -> the FORWARD kernels on re-packed weights (:func:`dgrad_weight`): a stride-1 conv's data gradient is the conv
  of dY with the spatially flipped, in/out-transposed filter; a stride-2 conv's is a small stride-1 conv over dY that
  produces the four output phases at once, stored through the PixelShuffle epilogue.
* everything here works on device tensors; packing is torch indexing on a few MB of weights (plumbing).
"""
import ctypes as C
from typing import Optional, Tuple

import torch

from . import ops
from ._lib import lib, check, ConvDesc, CONV_MASK_POS, CONV_PIXSHUF, CONV_RES_ADD


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ----------------------------------------------------------------------------------------------
# device-side packing
# ----------------------------------------------------------------------------------------------
def pack_rows_device(weight: torch.Tensor, w_rows: int, cin_pad: int, k_pad: int, out: Optional[torch.Tensor] = None):
    """(O, I, kh, kw) fp32 device tensor -> fp16 [w_rows][k_pad] with k = (r*kw + q)*cin_pad + c (ops.pack_conv's layout)."""
    o, i, kh, kw = weight.shape
    if out is None:
        out = torch.zeros(w_rows, k_pad, dtype=torch.float16, device=weight.device)
    v = out[:o, :kh * kw * cin_pad].view(o, kh, kw, cin_pad)
    v[..., :i] = weight.permute(0, 2, 3, 1).to(torch.float16)
    return out


def unpack_wgrad(dw: torch.Tensor, cout: int, cin: int, kh: int, kw: int, cin_pad: int) -> torch.Tensor:
    """fp32 [w_rows][k_pad] block of the packed gradient -> (cout, cin, kh, kw)."""
    return dw[:cout, :kh * kw * cin_pad].view(cout, kh, kw, cin_pad)[..., :cin].permute(0, 3, 1, 2)


def dgrad_geometry(kh: int, kw: int, stride: int, pad: int):
    """Window of the stride-1 conv over dY that yields dX. Returns (KH', KW', pad_h, pad_w, taps) where, for stride 2,
    taps[(py, dy)] = r (filter row feeding output phase py from window row dy), same for columns."""
    if stride == 1:
        return kh, kw, kh - 1 - pad, kw - 1 - pad, None
    assert stride == 2, 'dgrad is built for stride 1 and 2 (ResNet / FPN)'

    def axis(k):
        m = {}
        for ph in range(2):
            for r in range(k):
                if (ph + pad - r) % 2 == 0:
                    m[(ph, (ph + pad - r) // 2)] = r      # forward output row = y + d
        lo = min(d for (_, d) in m) if m else 0
        hi = max(d for (_, d) in m) if m else 0
        return {(ph, d - lo): r for (ph, d), r in m.items()}, hi - lo + 1, -lo

    ty, KH, ph_ = axis(kh)
    tx, KW, pw_ = axis(kw)
    return KH, KW, ph_, pw_, (ty, tx)


def dgrad_weight(weight: torch.Tensor, stride: int, pad: int, cin_pad: int) -> Tuple[torch.Tensor, int, int, int, int, int]:
    """Filter of the conv that maps dY -> dX. weight: (Cout, Cin, kh, kw) fp32 (effective, BN folded).
    Returns (w', KH', KW', pad', flags) with w' of shape (rows, Cout, KH', KW'): rows = Cin for stride 1,
    4*cin_pad (conv channel 4*ci + 2*py + px) for stride 2 + PixelShuffle."""
    cout, cin, kh, kw = weight.shape
    KH, KW, ph_, pw_, taps = dgrad_geometry(kh, kw, stride, pad)
    if stride == 1:
        assert ph_ == pw_ and ph_ >= 0
        return weight.flip(2, 3).permute(1, 0, 2, 3).contiguous(), KH, KW, ph_, 0
    ty, tx = taps
    assert ph_ == pw_, 'square geometry expected'
    w = torch.zeros(cin_pad, 2, 2, cout, KH, KW, dtype=weight.dtype, device=weight.device)
    # window row d of dY is forward output row y + d - pad'; window position is what the forward kernel calls (r, q)
    for (py, dy), r in ty.items():
        for (px, dx), q in tx.items():
            w[:cin, py, px, :, dy, dx] = weight[:, :, r, q].t()
    return w.view(cin_pad * 4, cout, KH, KW), KH, KW, ph_, CONV_PIXSHUF


# ----------------------------------------------------------------------------------------------
# launches
# ----------------------------------------------------------------------------------------------
def conv_wgrad(x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, *, cin_pad: int, cout: int, w_rows: int, k_pad: int,
               kh: int, kw: int, stride: int, pad: int, groups: int, batch: int, seg: Optional[tuple] = None, x_ld: int = 0, dy_ld: int = 0,
               db: Optional[torch.Tensor] = None, dilation: int = 1, segs: Optional[list] = None):
    """Accumulate the weight gradient into dw (fp32 [groups][w_rows][k_pad]) and, if given, the bias gradient into db (fp32
    [groups*cout]). seg = (h, w, ho, wo, x_pixel_off, dy_pixel_off); `segs` = several of them (the pyramid levels a shared filter ran
    on): one call, and for 3x3 / stride-1 filters one launch whose pixel splits run over the patches of all levels."""
    segs = [seg] if segs is None else list(segs)
    d = _wgrad_desc(x, cin_pad=cin_pad, cout=cout, w_rows=w_rows, k_pad=k_pad, kh=kh, kw=kw, stride=stride, pad=pad, groups=groups,
                    batch=batch, segs=segs, x_ld=x_ld, dilation=dilation)
    check(lib().llv_conv2d_nhwc_wgrad(C.byref(d), dy.data_ptr(), dy_ld, dw.data_ptr(), db.data_ptr() if db is not None else None,
                                      _stream()), 'llv_conv2d_nhwc_wgrad')


def _wgrad_desc(x, *, cin_pad, cout, w_rows, k_pad, kh, kw, stride, pad, groups, batch, segs, x_ld, dilation):
    d = ConvDesc()
    d.x[0] = x.data_ptr()
    d.n_src, d.n_seg = 1, len(segs)
    d.x_ld = x_ld
    d.cin, d.cout, d.w_rows, d.k_pad = cin_pad, cout, w_rows, k_pad
    d.kh, d.kw, d.stride, d.pad, d.groups, d.batch = kh, kw, stride, pad, groups, batch
    d.dilation = dilation
    for i, sg in enumerate(segs):
        d.seg_h[i], d.seg_w[i], d.seg_ho[i], d.seg_wo[i], d.seg_xoff[i], d.seg_yoff[i] = sg
    return d


class WgradGroup:
    """The weight gradients of several convolutions as one launch per kernel kind (llv_conv2d_nhwc_wgrad_plan / _jobs): `calls` are
    the keyword sets of conv_wgrad (x, dy, dw first). Planned once -- every pointer is that of a static buffer of the training graph --
    then `launch()` per step. A convolution's pixel splits are sized by its share of the group's multiply-adds."""

    def __init__(self, calls, device):
        import numpy as np
        from ._lib import LLV_MAX_SEG, WgradJob
        self.keep = []
        work = []
        for c in calls:
            segs = c['segs']
            work.append(float(sum(c['batch'] * sg[2] * sg[3] for sg in segs)) * c['cin_pad'] * c['cout'] * c['groups'] * c['kh'] * c['kw'])

        def plan(c, share):
            d = _wgrad_desc(c['x'], **{k: c[k] for k in ('cin_pad', 'cout', 'w_rows', 'k_pad', 'kh', 'kw', 'stride', 'pad', 'groups', 'batch',
                                                          'segs', 'x_ld', 'dilation')})
            jobs = (WgradJob * LLV_MAX_SEG)()
            n = C.c_int(0)
            check(lib().llv_conv2d_nhwc_wgrad_plan(C.byref(d), c['dy'].data_ptr(), c['dy_ld'], c['dw'].data_ptr(),
                                                   c['db'].data_ptr() if c['db'] is not None else None, float(share), jobs, C.byref(n)),
                  'llv_conv2d_nhwc_wgrad_plan')
            return [jobs[i] for i in range(n.value)]
        kinds = [plan(c, 1.0)[0].kind for c in calls]
        tot = {k: sum(w for w, kk in zip(work, kinds) if kk == k) for k in set(kinds)}
        by_kind = {}
        for c, w, k in zip(calls, work, kinds):
            self.keep += [c['x'], c['dy'], c['dw'], c['db']]
            for j in plan(c, max(w / tot[k], 1e-3)):
                by_kind.setdefault(j.kind, []).append(j)
        self.tables = []
        for kind, jobs in sorted(by_kind.items()):
            assert len(jobs) < 65536
            begin, w2j = 0, []
            arr = (WgradJob * len(jobs))()
            for i, j in enumerate(jobs):
                C.memmove(C.byref(arr[i]), C.byref(j), C.sizeof(WgradJob))
                arr[i].wg_begin = begin
                n = (j.n_wg + 7) // 8 * 8
                w2j += [i] * n
                begin += n
            t_jobs = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(device)
            t_map = torch.from_numpy(np.asarray(w2j, dtype=np.uint16).view(np.int16).copy()).to(device)
            self.tables.append((kind, t_jobs, t_map, begin))

    def launch(self):
        L, st = lib(), _stream()
        for kind, t_jobs, t_map, total in self.tables:
            check(L.llv_conv2d_nhwc_wgrad_jobs(t_jobs.data_ptr(), t_map.data_ptr(), kind, total, st), 'llv_conv2d_nhwc_wgrad_jobs')


def relu_bwd(dy: torch.Tensor, y: Optional[torch.Tensor], dx: torch.Tensor, dy2: Optional[torch.Tensor] = None):
    """dx = (dy [+ dy2]) * (y > 0) over contiguous fp16 buffers (in place allowed)."""
    n = dy.numel()
    check(lib().llv_relu_bwd(dy.data_ptr(), dy2.data_ptr() if dy2 is not None else None, y.data_ptr() if y is not None else None,
                             dx.data_ptr(), n, _stream()), 'llv_relu_bwd')


def relu_res_bwd(dy: torch.Tensor, y: torch.Tensor, gres: torch.Tensor):
    """dy *= (y > 0) in place, then gres += dy: the block output relu(v + identity) in one pass (contiguous fp16 buffers)."""
    check(lib().llv_relu_res_bwd(dy.data_ptr(), y.data_ptr(), gres.data_ptr(), dy.numel(), _stream()), 'llv_relu_res_bwd')


def act_bwd(dy: torch.Tensor, y: torch.Tensor, res: Optional[torch.Tensor], gres: Optional[torch.Tensor], slope: float):
    """llv_act_bwd over (..., C) fp16 tensors whose last-but-one stride is their row pitch: gres += dy; dy *= (y - res > 0 ? 1 : slope)."""
    c = y.shape[-1]
    rows = y.numel() // c
    ld = lambda t: t.stride(-2) if t.dim() > 1 else c
    check(lib().llv_act_bwd(dy.data_ptr(), y.data_ptr(), _p(res), _p(gres), rows, c, ld(dy), ld(y), ld(res) if res is not None else 0,
                            ld(gres) if gres is not None else 0, float(slope), _stream()), 'llv_act_bwd')


def upsample2x_bwd(gdst: torch.Tensor, gsrc: torch.Tensor):
    """gsrc (n, h, w, c) += the four fine pixels of gdst (n, 2h, 2w, c) (views with their own row pitch allowed)."""
    n, h, w, c = gsrc.shape
    check(lib().llv_upsample2x_nhwc_bwd(gdst.data_ptr(), gsrc.data_ptr(), n, h, w, c, gdst.stride(-2), gsrc.stride(-2), _stream()),
          'llv_upsample2x_nhwc_bwd')


def bias_grad(dy: torch.Tensor, m: int, c: int, ld: int, out: torch.Tensor):
    check(lib().llv_bias_grad(dy.data_ptr(), m, c, ld, out.data_ptr(), _stream()), 'llv_bias_grad')


def upsample_add_bwd(dfine: torch.Tensor, dcoarse: torch.Tensor, accumulate: bool):
    n, H, W, c = dfine.shape
    _, h, w, c2 = dcoarse.shape
    assert c == c2 and dfine.is_contiguous() and dcoarse.is_contiguous()
    check(lib().llv_upsample_add_bwd(dfine.data_ptr(), dcoarse.data_ptr(), n, H, W, h, w, c, 1 if accumulate else 0, _stream()),
          'llv_upsample_add_bwd')


def sgd_momentum(w: torch.Tensor, g: torch.Tensor, mom: torch.Tensor, lr: float, momentum: float, weight_decay: float,
                 grad_scale: float, first_step: bool):
    assert w.dtype == g.dtype == mom.dtype == torch.float32 and w.is_contiguous() and g.is_contiguous() and mom.is_contiguous()
    check(lib().llv_sgd_momentum(w.data_ptr(), g.data_ptr(), mom.data_ptr(), w.numel(), lr, momentum, weight_decay, grad_scale,
                                 1 if first_step else 0, _stream()), 'llv_sgd_momentum')


def conv_dgrad(dy: torch.Tensor, pcd: ops.PackedConv, dx: torch.Tensor, batch: int, segs, dy_ld: int, dx_ld: int,
               flags: int, accumulate: bool, in_hw=None, fwd_stride: int = None, fwd_cin: int = None, mask: Optional[torch.Tensor] = None,
               res: Optional[torch.Tensor] = None):
    """dX (+)= data gradient through llv_conv2d_nhwc_dgrad. segs: the FORWARD launch's segments (h, w, ho, wo, x_off, y_off);
    pcd: the filter packed for the gradient (dgrad_weight / llv_pack_conv_weights modes 1, 2). flags: CONV_PIXSHUF for a
    stride-2 forward conv. mask: the forward conv's INPUT activation when that is a ReLU output (laid out like dx): dX also takes that
    ReLU's mask, so the launch it is the output gradient of needs no llv_relu_bwd pass (stride-1 convs only). res (laid out like dx,
    accumulate False): dX = gradient + res, e.g. the gradient of a residual block's output joining at its first conv."""
    s2 = bool(flags & CONV_PIXSHUF)
    f = ConvDesc()
    if mask is not None:
        f.x[0], f.x_ld, f.flags = mask.data_ptr(), dx_ld, CONV_MASK_POS
    if res is not None:
        assert not accumulate and not s2
        f.res, f.res_ld = res.data_ptr(), dx_ld
    f.n_src, f.n_seg = 1, len(segs)
    f.stride = 2 if s2 else 1
    f.cin = pcd.cout // 4 if s2 else pcd.cout
    f.groups, f.batch = pcd.groups, batch
    f.dilation = pcd.dilation
    for s, (h, w, ho, wo, xo, yo) in enumerate(segs):
        f.seg_h[s], f.seg_w[s], f.seg_ho[s], f.seg_wo[s], f.seg_xoff[s], f.seg_yoff[s] = h, w, ho, wo, xo, yo
    # split-K workspace of the underlying forward launch, if the planner wants one
    probe = ops.make_conv_desc(pcd, [dy], dx, batch,
                               [((ho, wo, (h + 1) // 2, (w + 1) // 2, yo, xo) if s2 else (ho, wo, h, w, yo, xo)) for (h, w, ho, wo, xo, yo) in segs],
                               dy_ld, dx_ld, dx if (accumulate or res is not None) else None, dx_ld if (accumulate or res is not None) else 0,
                               flags | (CONV_RES_ADD if (accumulate or res is not None) else 0), ps_out=(segs[0][0], segs[0][1]) if s2 else (0, 0))
    wsb = ops.conv_workspace_bytes(probe)
    ws = torch.empty((wsb,), device=dy.device, dtype=torch.uint8) if wsb else None
    check(lib().llv_conv2d_nhwc_dgrad(C.byref(f), pcd.w.data_ptr(), pcd.bias.data_ptr(), pcd.w_rows, pcd.k_pad, pcd.kh, pcd.kw, pcd.pad,
                                      dy.data_ptr(), dy_ld, dx.data_ptr(), dx_ld, 1 if accumulate else 0,
                                      ws.data_ptr() if ws is not None else None, wsb, _stream()), 'llv_conv2d_nhwc_dgrad')


def _p(t):
    return t.data_ptr() if t is not None else None


class JobRecorder:
    """While active, pack_weights / fold_bias / unpack_wgrad_device record their arguments as jobs instead of launching: the
    training graph replays its bookkeeping once under a recorder and from then on launches each table with ONE call
    (llv_*_batched). The recorded pointers are those of the live parameter / buffer / gradient tensors: `signature` lets the owner
    notice when one of them moved (load_state_dict into new storage, TrainStep's flat buffers) and record again."""
    active = None

    def __init__(self):
        self.pack, self.fold, self.unpack, self.tensors = [], [], [], []

    def __enter__(self):
        assert JobRecorder.active is None
        JobRecorder.active = self
        return self

    def __exit__(self, *exc):
        JobRecorder.active = None

    def note(self, *ts):
        self.tensors += [t for t in ts if t is not None]

    def signature(self):
        return tuple(t.data_ptr() for t in self.tensors)


class JobTables:
    """Device copies of a recorder's job lists + their launches."""

    def __init__(self, rec: JobRecorder, device):
        import ctypes as C
        import numpy as np
        self.rec = rec
        self.sig = rec.signature()

        def table(jobs, ty):
            if not jobs:
                return None
            arr = (ty * len(jobs))(*jobs)
            return torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(device)
        from ._lib import FoldJob, PackJob, UnpackJob
        fwd = [j for j in rec.pack if j.mode == 0]
        dgr = [j for j in rec.pack if j.mode != 0]                # transposes: the LDS-tiled kernel
        self.n = (len(fwd), len(rec.fold), len(rec.unpack), len(dgr))
        self.t_pack, self.t_fold, self.t_unpack = table(fwd, PackJob), table(rec.fold, FoldJob), table(rec.unpack, UnpackJob)
        self.t_dgrad = table(dgr, PackJob)
        self.b_pack = max([1] + [min(512, (j.cout * j.kh * j.kw * j.cpad + 255) // 256) for j in fwd])
        self.b_dgrad = max([1] + [((j.cout + 31) // 32) * ((j.cin + 31) // 32) for j in dgr])
        self.b_fold = max([1] + [min(64, (j.n + 255) // 256) for j in rec.fold])
        self.b_unpack = max([1] + [j.cout for j in rec.unpack])

    def valid(self):
        return self.sig == self.rec.signature()

    def launch(self, dgrad_stream=None):
        """dgrad_stream: the data-gradient filters are not read before the backward pass -- their table may run on this stream beside
        whatever follows on the current one (forked here; the caller joins it before the first data-gradient launch)."""
        L, st = lib(), _stream()
        if self.t_pack is not None:
            check(L.llv_pack_conv_weights_batched(self.t_pack.data_ptr(), self.n[0], self.b_pack, st), 'llv_pack_conv_weights_batched')
        if self.t_dgrad is not None:
            if dgrad_stream is not None:
                dgrad_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(dgrad_stream):
                    check(L.llv_pack_dgrad_weights_batched(self.t_dgrad.data_ptr(), self.n[3], self.b_dgrad, _stream()),
                          'llv_pack_dgrad_weights_batched')
            else:
                check(L.llv_pack_dgrad_weights_batched(self.t_dgrad.data_ptr(), self.n[3], self.b_dgrad, st), 'llv_pack_dgrad_weights_batched')
        if self.t_fold is not None:
            check(L.llv_fold_bias_batched(self.t_fold.data_ptr(), self.n[1], self.b_fold, st), 'llv_fold_bias_batched')
        if self.t_unpack is not None:
            check(L.llv_unpack_wgrad_batched(self.t_unpack.data_ptr(), self.n[2], self.b_unpack, st), 'llv_unpack_wgrad_batched')


def pack_weights(w: torch.Tensor, gamma, var, eps: float, dst: torch.Tensor, *, cpad: int, k_pad: int, row0: int, mode: int,
                 pad: int = 0, KH: int = 1, KW: int = 1, padp: int = 0, rows: int = 1):
    """llv_pack_conv_weights: (Cout,Cin,kh,kw) fp32 parameter (+ eval-mode BN scale) -> one block of a packed fp16 weight
    tensor (dst = the group's [w_rows][k_pad] matrix). mode 0 forward, 1 / 2 data gradient of a stride-1 / stride-2 conv."""
    cout, cin, kh, kw = w.shape
    assert w.is_contiguous() and w.dtype == torch.float32 and dst.dtype == torch.float16
    rec = JobRecorder.active
    if rec is not None:
        from ._lib import PackJob
        rec.pack.append(PackJob(w.data_ptr(), _p(gamma), _p(var), float(eps), dst.data_ptr(), cout, cin, kh, kw, cpad, k_pad, row0, mode, pad, KH, KW,
                                padp, rows))
        rec.note(w, gamma, var, dst)
        return
    check(lib().llv_pack_conv_weights(w.data_ptr(), _p(gamma), _p(var), float(eps), dst.data_ptr(), cout, cin, kh, kw, cpad, k_pad,
                                      row0, mode, pad, KH, KW, padp, rows, _stream()), 'llv_pack_conv_weights')


def fold_bias(b, gamma, beta, mean, var, eps: float, out: torch.Tensor):
    rec = JobRecorder.active
    if rec is not None:
        from ._lib import FoldJob
        rec.fold.append(FoldJob(_p(b), _p(gamma), _p(beta), _p(mean), _p(var), float(eps), out.data_ptr(), out.numel()))
        rec.note(b, gamma, beta, mean, var, out)
        return
    check(lib().llv_fold_bias(_p(b), _p(gamma), _p(beta), _p(mean), _p(var), float(eps), out.data_ptr(), out.numel(), _stream()),
          'llv_fold_bias')


def unpack_wgrad_device(dw: torch.Tensor, db, w: torch.Tensor, gamma, mean, var, eps: float, gw, ggamma, gbeta, gbias, *,
                        cpad: int, k_pad: int):
    """llv_unpack_wgrad: rows of the packed fp32 gradient (dw: first row of the part) -> parameter gradients."""
    cout, cin, kh, kw = w.shape
    rec = JobRecorder.active
    if rec is not None:
        from ._lib import UnpackJob
        rec.unpack.append(UnpackJob(dw.data_ptr(), _p(db), w.data_ptr(), _p(gamma), _p(mean), _p(var), float(eps), _p(gw), _p(ggamma), _p(gbeta),
                                    _p(gbias), cout, cin, kh, kw, cpad, k_pad))
        rec.note(dw, db, w, gamma, mean, var, gw, ggamma, gbeta, gbias)
        return
    check(lib().llv_unpack_wgrad(dw.data_ptr(), _p(db), w.data_ptr(), _p(gamma), _p(mean), _p(var), float(eps), _p(gw), _p(ggamma),
                                 _p(gbeta), _p(gbias), cout, cin, kh, kw, cpad, k_pad, _stream()), 'llv_unpack_wgrad')
