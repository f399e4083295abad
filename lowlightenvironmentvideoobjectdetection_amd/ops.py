"""Torch-facing wrappers over the C ABI (include/llv.h). Tensors only carry device memory and the
stream; all arithmetic happens in libllv.so. Every function raises if the library is missing or
the tensors are not on a GPU -- there is no CPU path here (the CPU restatement lives in oracle/).
"""
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import CONV_PS_EMAJOR  # noqa: F401
from ._lib import (CONV_LEAKY, CONV_OUT_F32, CONV_PIXSHUF, CONV_RELU, CONV_RES_ADD, CONV_RES_RSUB, CONV_RES_UPS,
                   ConvDesc, DecodeDesc, Extents, LlvError, check, lib)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise LlvError('llv ops need tensors on an MI355X device; the CPU restatement is oracle/, '
                           'not a fallback of this package')


def round_up(v, m):
    return (v + m - 1) // m * m


def tensor_extent(t: Optional[torch.Tensor]) -> int:
    """Bytes addressable from t.data_ptr() to the end of the allocation behind t (a view's storage): what llv_extents declares."""
    if t is None:
        return 0
    st = t.untyped_storage()
    return int(st.nbytes() - (t.data_ptr() - st.data_ptr()))


# ----------------------------------------------------------------------------------------------
# convolution
# ----------------------------------------------------------------------------------------------
@dataclass
class PackedConv:
    """Weights of one conv launch in the layout llv_conv2d_nhwc_fwd reads (see include/llv.h)."""
    w: torch.Tensor          # fp16 [groups, w_rows, k_pad]
    bias: torch.Tensor       # fp32 [groups * cout]
    cin: int                 # per-group padded input channels (multiple of 8)
    cout: int                # per-group stored output channels (multiple of 4)
    w_rows: int
    k_pad: int
    kh: int
    kw: int
    stride: int
    pad: int
    groups: int = 1
    n_src: int = 1
    src_cvec: int = 0
    true_macs_per_pixel: int = 0   # unpadded Cout*Cin/g*kh*kw, for FLOP accounting
    w_halo: Optional[torch.Tensor] = None   # second packing for the direct 3x3 kernel (add_halo_packing)
    halo_bn: int = 0
    dilation: int = 1        # > 1: dilated taps (forward only, implicit-GEMM kernels)
    # which module parameters each block of rows came from (training: repack after an optimizer step, route the
    # weight gradient back): dicts with wkey / bkey / bn (4 keys) / group / row0 / cout / cin / eps
    parts: Optional[list] = None

    def to(self, device):
        self.w = self.w.to(device)
        self.bias = self.bias.to(device)
        if self.w_halo is not None:
            self.w_halo = self.w_halo.to(device)
        return self


def halo_eligible(pc: 'PackedConv') -> bool:
    if not (pc.kh == 3 and pc.kw == 3 and pc.stride == 1 and pc.pad == 1):
        return False
    if not (pc.cin in (8, 16, 24) or pc.cin % 32 == 0):
        return False
    if pc.n_src > 1 and pc.groups != pc.n_src:
        # sources concatenated along C: every 32-channel chunk must lie inside one source
        if pc.groups != 1 or (pc.src_cvec * 8) % min(pc.cin, 32) != 0:
            return False
    return True


def add_halo_packing(pc: 'PackedConv') -> 'PackedConv':
    """Second packing of the same (already folded, fp16) weights for conv3x3_halo_kernel:
    [group][chunk][n-tile][bn rows][Kc_pad + 8], Kc = 9*CK, CK = min(cin, 32), k = tap*CK + c (see include/llv.h)."""
    if not halo_eligible(pc):
        return pc
    cin, g = pc.cin, pc.groups
    ck = min(cin, 32)
    nch = cin // ck
    kc = 9 * ck
    kc_pad = round_up(kc, 32)
    wrow = kc_pad + 8
    bn = 16 if pc.cout <= 16 else (32 if pc.cout <= 32 else 64)
    nt = (pc.cout + bn - 1) // bn
    w = pc.w.cpu()[:, :, :9 * cin].reshape(g, pc.w_rows, 9, cin)
    out = torch.zeros(g, nch, nt, bn, wrow, dtype=torch.float16)
    for ci in range(nch):
        for t in range(nt):
            r0, r1 = t * bn, min((t + 1) * bn, pc.w_rows)
            if r1 > r0:
                out[:, ci, t, :r1 - r0, :kc] = w[:, r0:r1, :, ci * ck:(ci + 1) * ck].reshape(g, r1 - r0, kc)
    pc.w_halo = out.contiguous().to(pc.w.device)
    pc.halo_bn = bn
    return pc


def pack_conv(weight: torch.Tensor, bias: Optional[torch.Tensor], stride=1, pad=0,
              in_map: Optional[Sequence[int]] = None, cout_store: Optional[int] = None,
              n_src: int = 1, bn=None, eps: float = 1e-5, dilation: int = 1) -> PackedConv:
    """Pack a dense conv weight [Cout, Cin, kh, kw] (fp32) for the implicit-GEMM kernel.

    in_map[j] = original input channel feeding padded channel j, or -1 for a zero channel; default is
    identity padded to a multiple of 8. `bn` = (gamma, beta, running_mean, running_var) folds an
    eval-mode BatchNorm2d that follows the conv (norm_eval / frozen stats, resnet.py:642-651):
    w' = w * gamma / sqrt(var + eps), b' = beta + (b - mean) * gamma / sqrt(var + eps).
    """
    weight_in, bn_in, in_map_default = weight, bn, in_map is None
    weight = weight.detach().double().cpu()
    cout, cin, kh, kw = weight.shape
    b = torch.zeros(cout, dtype=torch.float64) if bias is None else bias.detach().double().cpu()
    if bn is not None:
        gamma, beta, mean, var = [t.detach().double().cpu() for t in bn]
        s = gamma / torch.sqrt(var + eps)
        weight = weight * s.view(-1, 1, 1, 1)
        b = beta + (b - mean) * s
    if in_map is None:
        in_map = list(range(cin)) + [-1] * (round_up(cin, 8) - cin)
    cin_pad = len(in_map)
    assert cin_pad % 8 == 0
    if cout_store is None:
        cout_store = round_up(cout, 4)
    assert cout_store % 4 == 0 and cout_store >= cout
    w_rows = round_up(cout_store, 16)
    k = kh * kw * cin_pad
    k_pad = round_up(k, 32)
    wp = torch.zeros(w_rows, kh, kw, cin_pad, dtype=torch.float64)
    idx = torch.tensor([j for j, s in enumerate(in_map) if s >= 0], dtype=torch.long)
    src = torch.tensor([s for s in in_map if s >= 0], dtype=torch.long)
    wp[:cout][:, :, :, idx] = weight.permute(0, 2, 3, 1)[:, :, :, src]
    wfull = torch.zeros(w_rows, k_pad, dtype=torch.float64)
    wfull[:, :k] = wp.reshape(w_rows, k)
    bfull = torch.zeros(cout_store, dtype=torch.float64)
    bfull[:cout] = b
    pc = PackedConv(w=wfull.to(torch.float16).unsqueeze(0).contiguous(), bias=bfull.float().contiguous(),
                    cin=cin_pad, cout=cout_store, w_rows=w_rows, k_pad=k_pad, kh=kh, kw=kw, stride=stride,
                    pad=pad, groups=1, n_src=n_src, src_cvec=(cin_pad // 8 // n_src) if n_src > 1 else 0,
                    true_macs_per_pixel=cout * cin * kh * kw, dilation=int(dilation))
    wkey = getattr(weight_in, '_llv_key', None)
    if wkey is not None and in_map_default:
        pc.parts = [dict(wkey=wkey, bkey=getattr(bias, '_llv_key', None) if bias is not None else None,
                         bn=tuple(getattr(t, '_llv_key', None) for t in bn_in) if bn_in is not None else None,
                         group=0, row0=0, cout=cout, cin=cin, eps=eps)]
    return pc


def pixshuf_emajor(pc: PackedConv) -> PackedConv:
    """Re-order the output channels of a conv that feeds PixelShuffle(2) from torch's 4*cc + e to e*(Cout/4) + cc (e = 2*dy + dx):
    launched with CONV_PS_EMAJOR the tap kernel then writes whole output pixels per workgroup. Returns a new PackedConv."""
    assert pc.groups == 1 and pc.cout % 32 == 0 and pc.w_halo is None
    oc = pc.cout // 4
    perm = torch.tensor([4 * (j % oc) + j // oc for j in range(pc.cout)], dtype=torch.long)
    w = pc.w.clone()
    w[0, :pc.cout] = pc.w[0, perm.to(pc.w.device)]
    b = pc.bias.clone()
    b[:pc.cout] = pc.bias[perm.to(pc.bias.device)]
    import dataclasses
    return dataclasses.replace(pc, w=w.contiguous(), bias=b.contiguous(), parts=None)


def stack_groups(pcs: List[PackedConv]) -> PackedConv:
    """Several equal-shape convs as one grouped launch (RetinaHead cls/reg towers side by side)."""
    p0 = pcs[0]
    for p in pcs:
        assert (p.cin, p.cout, p.w_rows, p.k_pad, p.kh, p.kw, p.stride, p.pad, p.groups, p.n_src) == \
               (p0.cin, p0.cout, p0.w_rows, p0.k_pad, p0.kh, p0.kw, p0.stride, p0.pad, 1, 1)
    out = PackedConv(w=torch.cat([p.w for p in pcs], 0).contiguous(), bias=torch.cat([p.bias for p in pcs]).contiguous(),
                     cin=p0.cin, cout=p0.cout, w_rows=p0.w_rows, k_pad=p0.k_pad, kh=p0.kh, kw=p0.kw,
                     stride=p0.stride, pad=p0.pad, groups=len(pcs),
                     true_macs_per_pixel=sum(p.true_macs_per_pixel for p in pcs))
    if all(p.parts is not None for p in pcs):
        out.parts = [dict(q, group=g) for g, p in enumerate(pcs) for q in p.parts]
    return out


def concat_cout(pcs: List[PackedConv]) -> PackedConv:
    """Convs sharing the input, outputs concatenated along C (one launch, N = sum of couts)."""
    p0 = pcs[0]
    for p in pcs:
        assert (p.cin, p.k_pad, p.kh, p.kw, p.stride, p.pad, p.groups, p.n_src) == \
               (p0.cin, p0.k_pad, p0.kh, p0.kw, p0.stride, p0.pad, 1, 1)
    cout = sum(p.cout for p in pcs)
    w_rows = round_up(cout, 16)
    w = torch.zeros(1, w_rows, p0.k_pad, dtype=torch.float16)
    r = 0
    for p in pcs:
        w[0, r:r + p.cout] = p.w[0, :p.cout].cpu()
        r += p.cout
    out = PackedConv(w=w.contiguous(), bias=torch.cat([p.bias.cpu() for p in pcs]).contiguous(), cin=p0.cin,
                     cout=cout, w_rows=w_rows, k_pad=p0.k_pad, kh=p0.kh, kw=p0.kw, stride=p0.stride, pad=p0.pad,
                     true_macs_per_pixel=sum(p.true_macs_per_pixel for p in pcs))
    if all(p.parts is not None for p in pcs):
        out.parts, r = [], 0
        for p in pcs:
            out.parts += [dict(q, row0=r + q['row0']) for q in p.parts]
            r += p.cout
    return out


def conv_out_size(h, w, pc: PackedConv):
    d = max(int(getattr(pc, 'dilation', 1)), 1)
    return (h + 2 * pc.pad - d * (pc.kh - 1) - 1) // pc.stride + 1, (w + 2 * pc.pad - d * (pc.kw - 1) - 1) // pc.stride + 1


def make_conv_desc(pc: PackedConv, xs: Sequence[torch.Tensor], y: torch.Tensor, batch: int,
                   segs: Sequence[tuple], x_ld: int, y_ld: int, res: Optional[torch.Tensor] = None,
                   res_ld: int = 0, flags: int = 0, res_hw=(0, 0), tile: int = 0, k_splits: int = 0,
                   y_group_stride: int = 0, act_slope: float = 0.0, ps_out=(0, 0), clsmax=None) -> ConvDesc:
    """segs: [(h, w, ho, wo, x_pixel_off, y_pixel_off), ...]. clsmax = (clsmax uint32 [rows][A], seg_written uint8 [rows][tiles], cls_c,
    store_thr) with the LLV_CONV_CLSMAX flag (include/llv.h)."""
    d = ConvDesc()
    for i in range(_lib.LLV_MAX_SRC):
        d.x[i] = xs[i].data_ptr() if i < len(xs) else 0
    d.w = pc.w.data_ptr()
    d.bias = pc.bias.data_ptr()
    d.res = res.data_ptr() if res is not None else 0
    d.y = y.data_ptr()
    d.n_src, d.src_cvec = pc.n_src, pc.src_cvec
    d.x_ld, d.y_ld, d.res_ld = x_ld, y_ld, res_ld
    d.cin, d.cout, d.w_rows, d.k_pad = pc.cin, pc.cout, pc.w_rows, pc.k_pad
    d.kh, d.kw, d.stride, d.pad = pc.kh, pc.kw, pc.stride, pc.pad
    d.groups, d.batch = pc.groups, batch
    d.n_seg = len(segs)
    for s, (h, w, ho, wo, xo, yo) in enumerate(segs):
        d.seg_h[s], d.seg_w[s], d.seg_ho[s], d.seg_wo[s] = h, w, ho, wo
        d.seg_xoff[s], d.seg_yoff[s] = xo, yo
    d.flags = flags
    d.res_h, d.res_w = res_hw
    d.tile = tile
    d.k_splits = k_splits
    d.workspace, d.workspace_bytes = 0, 0
    d.y_group_stride = y_group_stride
    d.act_slope = act_slope
    d.ps_out_h, d.ps_out_w = ps_out
    d.dilation = max(int(getattr(pc, 'dilation', 1)), 1)
    if clsmax is not None:
        d.clsmax, d.seg_written, d.cls_c, d.store_thr = clsmax[0].data_ptr(), clsmax[1].data_ptr(), int(clsmax[2]), float(clsmax[3])
    if pc.w_halo is not None and d.dilation == 1 and len(segs) == 1 and tile == 0 and segs[0][0] == segs[0][2] and segs[0][1] == segs[0][3]:
        d.w_halo, d.halo_bn = pc.w_halo.data_ptr(), pc.halo_bn
    else:
        d.w_halo, d.halo_bn = 0, 0
    # declared buffer sizes (llv_extents): checked against the kernels' replayed addresses by llv_conv2d_extents, and before
    # every launch when LLV_CHECK_EXTENTS=1
    e = Extents()
    for i in range(_lib.LLV_MAX_SRC):
        e.x[i] = tensor_extent(xs[i]) if i < len(xs) else 0
    e.w, e.bias, e.res, e.y = tensor_extent(pc.w), tensor_extent(pc.bias), tensor_extent(res), tensor_extent(y)
    e.w_halo = tensor_extent(pc.w_halo) if pc.w_halo is not None else 0
    d._ext = e                      # keeps the struct alive as long as the descriptor
    d.extents = C.pointer(e)
    return d


def guard_trips(reset: bool = False) -> int:
    """llv_guard_trips: how many capacity guards of the decode / NMS kernels have fired (0 with consistent inputs)."""
    v = C.c_uint64(0)
    check(lib().llv_guard_trips(C.byref(v), 1 if reset else 0), 'llv_guard_trips')
    return int(v.value)


def raise_on_guard_trips(what: str):
    """Called by every results() hand-over after its synchronisation: a capacity guard that fired means the step's decode / NMS
    launches ran on counters or lists that were not consistent (include/llv.h, llv_guard_trips); the kernels stayed inside their
    buffers by clamping, so the detections of the step are wrong. Refuse to return them (the counter is reset so that the next
    step is judged on its own)."""
    n = guard_trips()
    if n:
        guard_trips(reset=True)
        raise LlvError(f'{what}: {n} post-processing capacity guard(s) fired during the step(s) since the last check; the '
                       f'detections are not trustworthy and are withheld (see DESIGN.md section 4)')


def conv_extents(desc) -> Extents:
    """llv_conv2d_extents / llv_convpair_extents: 1 + the highest byte the launch touches through each pointer (host replay)."""
    need = Extents()
    if isinstance(desc, _lib.InputCvDesc):
        check(lib().llv_convpair_extents(C.byref(desc), int(desc._output_block), C.byref(need)), 'llv_convpair_extents')
    else:
        check(lib().llv_conv2d_extents(C.byref(desc), C.byref(need)), 'llv_conv2d_extents')
    return need


def extents_violations(desc):
    """[(pointer name, bytes needed, bytes declared)] for the launch `desc` describes; an access below a pointer is reported
    with needed = -1. Empty list: every replayed address lies inside its declared buffer."""
    need = conv_extents(desc)
    have = (desc._keep if isinstance(desc, _lib.InputCvDesc) else desc)._ext
    out = []
    n, h = need.as_dict(), have.as_dict()
    for i, name in enumerate(Extents.NAMES):
        if need.below & (1 << i):
            out.append((name, -1, h[name]))
        if h[name] and n[name] > h[name]:
            out.append((name, n[name], h[name]))
    return out


def conv_workspace_bytes(desc: ConvDesc) -> int:
    return int(lib().llv_conv2d_workspace_bytes(C.byref(desc)))


def launch_conv(desc: ConvDesc):
    ws = None
    if not desc.workspace:
        wsb = conv_workspace_bytes(desc)
        if wsb:          # split-K launch without a caller workspace: allocate one for this call
            ws = torch.empty((wsb,), device=torch.device('cuda', torch.cuda.current_device()), dtype=torch.uint8)
            desc.workspace, desc.workspace_bytes = ws.data_ptr(), wsb
    check(lib().llv_conv2d_nhwc_fwd(C.byref(desc), _stream()), 'llv_conv2d_nhwc_fwd')
    if ws is not None:
        desc.workspace, desc.workspace_bytes = 0, 0


def conv2d(x, pc: PackedConv, res: Optional[torch.Tensor] = None, flags: int = 0, out: Optional[torch.Tensor] = None,
           tile: int = 0, k_splits: int = 0) -> torch.Tensor:
    """Single-segment convenience: x is (N,H,W,C) fp16 NHWC, or a list of n_src such tensors."""
    xs = list(x) if isinstance(x, (list, tuple)) else [x]
    _need_gpu(*xs, pc.w, pc.bias, res, out)
    n, h, w, cx = xs[0].shape
    for t in xs:
        assert t.dtype == torch.float16 and t.is_contiguous() and t.shape[:3] == (n, h, w)
    ho, wo = conv_out_size(h, w, pc)
    ctot = pc.groups * pc.cout
    if flags & CONV_PIXSHUF:
        oshape = (n, 2 * ho, 2 * wo, ctot // 4)
    else:
        oshape = (n, ho, wo, ctot)
    if out is None:
        out = torch.empty(oshape, device=xs[0].device, dtype=torch.float32 if flags & CONV_OUT_F32 else torch.float16)
    else:
        assert out.shape[:3] == oshape[:3] and out.is_contiguous()
    res_hw = (0, 0)
    if res is not None:
        assert res.dtype == torch.float16 and res.is_contiguous()
        if flags & CONV_RES_UPS:
            res_hw = (res.shape[1], res.shape[2])
    d = make_conv_desc(pc, xs, out, n, [(h, w, ho, wo, 0, 0)], cx, out.shape[3], res,
                       res.shape[3] if res is not None else 0, flags, res_hw, tile, k_splits)
    wsb = conv_workspace_bytes(d)
    if wsb:
        ws = torch.empty((wsb,), device=xs[0].device, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = ws.data_ptr(), wsb
        d._ext.workspace = wsb
    launch_conv(d)
    return out


_TILE_KERNEL = {
    -1: 'conv3x3_halo_kernel', 1: 'conv_igemm_glds_kernel<128,128>', 2: 'conv_igemm_glds_kernel<64,256>',
    3: 'conv_igemm_glds_kernel<32,256>', 4: 'conv_igemm_glds_kernel<16,256>', 5: 'conv_igemm_glds_kernel<64,64>',
    6: 'conv_igemm_kernel<128,128>', 7: 'conv_igemm_glds_kernel<128,128,K64>', 8: 'conv_igemm_glds_kernel<64,256,K64>',
    9: 'conv3x3_tap_kernel<128>', 10: 'conv3x3_tap_kernel<64>', 11: 'conv3x3_tap_kernel<64,R4>', 12: 'conv3x3_tap_kernel<64,G3>', 13: 'conv3x3_tap512_kernel',
}


def conv_kernel_name(desc: ConvDesc) -> str:
    """Name of the device kernel llv_conv2d_nhwc_fwd will launch for this descriptor (llv_conv2d_plan)."""
    tile, splits = C.c_int(0), C.c_int(0)
    check(lib().llv_conv2d_plan(C.byref(desc), C.byref(tile), C.byref(splits)), 'llv_conv2d_plan')
    name = _TILE_KERNEL.get(tile.value, f'tile{tile.value}')
    return name + ('+splitk' if splits.value > 1 else '')


def conv_flops(desc: ConvDesc) -> float:
    return lib().llv_conv2d_flops(C.byref(desc))


# ----------------------------------------------------------------------------------------------
# layout / pooling
# ----------------------------------------------------------------------------------------------
def maxpool3x3s2(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need_gpu(x)
    n, h, w, c = x.shape
    assert x.dtype == torch.float16 and x.is_contiguous()
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    if out is None:
        out = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.float16)
    check(lib().llv_maxpool3x3s2_nhwc(_ptr(x), _ptr(out), n, h, w, c, _stream()), 'llv_maxpool3x3s2_nhwc')
    return out


def pack_input(x: torch.Tensor, c_pad: int = 8, scale: Optional[torch.Tensor] = None,
               shift: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """NCHW fp32 -> NHWC fp16 with zero channel padding (optionally x*scale+shift per channel)."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    n, c, h, w = x.shape
    src = _nhwc_source(x, (n, h, w, c_pad)) if scale is None and shift is None else None
    if src is not None:            # x was unpacked from this very NHWC fp16 buffer, which still holds it: no conversion
        if out is None:
            return src
        out.copy_(src)
        return out
    if out is None:
        out = torch.empty((n, h, w, c_pad), device=x.device, dtype=torch.float16)
    check(lib().llv_pack_nchw_f32_to_nhwc_f16(_ptr(x), _ptr(out), n, c, h, w, c_pad, _ptr(scale), _ptr(shift),
                                               _stream()), 'llv_pack_nchw_f32_to_nhwc_f16')
    return out


def _nhwc_source(x: torch.Tensor, shape):
    """The NHWC fp16 plan buffer `x` (an NCHW fp32 module output) was unpacked from, if it still holds the same values: its plan has not
    run since (Plan.epoch), x has not been written to (tensor version), and the layout asked for is the buffer's. Module-level calls
    hand NCHW fp32 tensors to each other as the reference's modules do; between two modules of this package the consumer would only
    convert them back."""
    tag = getattr(x, '_llv_nhwc', None)
    if tag is None:
        return None
    buf, owner, epoch, version = tag
    if owner.epoch != epoch or x._version != version or tuple(buf.shape) != tuple(shape) or buf.device != x.device:
        return None
    return buf


def unpack_output(x: torch.Tensor, c: int, owner=None) -> torch.Tensor:
    """NHWC fp16/fp32 (first c channels) -> NCHW fp32. `owner` = the engine.Plan whose run() wrote x: the result then remembers where
    it came from (_nhwc_source), so that a pack_input of it by the next module is free."""
    _need_gpu(x)
    assert x.is_contiguous()
    n, h, w, ld = x.shape
    out = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    if x.dtype == torch.float16:
        check(lib().llv_unpack_nhwc_f16_to_nchw_f32(_ptr(x), _ptr(out), n, c, h, w, ld, _stream()), 'unpack f16')
    elif x.dtype == torch.float32:
        check(lib().llv_unpack_nhwc_f32_to_nchw_f32(_ptr(x), _ptr(out), n, c, h, w, ld, _stream()), 'unpack f32')
    else:
        raise LlvError('unpack_output: fp16 or fp32 only')
    if owner is not None and x.dtype == torch.float16:
        out._llv_nhwc = (x, owner, owner.epoch, out._version)
    return out


# ----------------------------------------------------------------------------------------------
# post-processing
# ----------------------------------------------------------------------------------------------
@dataclass
class DecodePlan:
    """Everything llv_retina_decode_topk needs for a fixed pyramid geometry (buffers are reused)."""
    desc: DecodeDesc
    boxes: torch.Tensor
    scores: torch.Tensor
    cand_anchor: torch.Tensor
    workspace: torch.Tensor
    base_anchors: torch.Tensor
    n_cand: int
    keep: list = field(default_factory=list)


def set_decode_image_params(d: DecodeDesc, img_shapes, scale_factors, rescale, clip=True):
    """Per-image clip size / scale factor of a (batched) decode descriptor; host-side struct edit, no launch."""
    batch = max(int(d.batch), 1)
    if batch > _lib.LLV_MAX_BATCH:
        raise LlvError(f'decode descriptor: {batch} images per step; the post-processing descriptors take at most LLV_MAX_BATCH = '
                       f'{_lib.LLV_MAX_BATCH} (include/llv.h); split the step')
    for n in range(batch):
        shp = img_shapes[n] if img_shapes is not None else None
        sf = np.asarray(scale_factors[n] if scale_factors is not None else 1.0, dtype=np.float32).reshape(-1)
        if sf.size == 1:
            sf = np.repeat(sf, 4)
        d.img_h_b[n] = float(shp[0]) if shp is not None else 0.0
        d.img_w_b[n] = float(shp[1]) if shp is not None else 0.0
        for e in range(4):
            d.scale_factor_b[n][e] = float(sf[e])
        if n == 0:
            d.img_h, d.img_w = d.img_h_b[0], d.img_w_b[0]
            for e in range(4):
                d.scale_factor[e] = float(sf[e])
    d.clip = 1 if (clip and img_shapes is not None) else 0
    d.rescale = 1 if rescale else 0


def make_decode_plan(cls: torch.Tensor, reg: torch.Tensor, level_hw, strides, cls_off, reg_off, cls_ld, reg_ld,
                     num_classes, num_anchors, base_anchors, nms_pre, means, stds, wh_ratio_clip, img_shape,
                     scale_factor, rescale, batch: int = 1, clsmax=None) -> DecodePlan:
    """batch > 1: img_shape / scale_factor are per-image lists; level l of image n starts n*h*w pixels after cls_off[l];
    outputs are (batch, n_cand, ...). clsmax = (clsmax, seg_written) tensors of the LLV_CONV_CLSMAX launch that wrote `cls`."""
    _need_gpu(cls, reg)
    if batch > _lib.LLV_MAX_BATCH:
        raise LlvError(f'{batch} images per step: the post-processing descriptors take at most LLV_MAX_BATCH = {_lib.LLV_MAX_BATCH} '
                       f'(include/llv.h); split the step')
    d = DecodeDesc()
    d.n_levels = len(level_hw)
    d.num_classes, d.num_anchors, d.nms_pre = num_classes, num_anchors, nms_pre
    for l, ((h, w), s) in enumerate(zip(level_hw, strides)):
        d.level_h[l], d.level_w[l] = h, w
        sw, sh = (s, s) if isinstance(s, int) else s
        d.stride_w[l], d.stride_h[l] = sw, sh
        d.cls_off[l], d.reg_off[l] = cls_off[l], reg_off[l]
    d.cls_ld, d.reg_ld = cls_ld, reg_ld
    assert cls.dtype == reg.dtype and cls.dtype in (torch.float16, torch.float32)
    d.logits_f32 = 1 if cls.dtype == torch.float32 else 0
    d.cls, d.reg = cls.data_ptr(), reg.data_ptr()
    ba = torch.stack([b.float() for b in base_anchors]).contiguous().to(cls.device)  # [L][A][4]
    d.base_anchors = ba.data_ptr()
    for e in range(4):
        d.means[e], d.stds[e] = float(means[e]), float(stds[e])
    d.max_ratio = float(np.float32(np.abs(np.log(wh_ratio_clip))))
    d.batch = batch
    if batch == 1 and (img_shape is None or not isinstance(img_shape[0], (tuple, list, np.ndarray))):
        shapes = [img_shape] if img_shape is not None else None
        sfs = [scale_factor]
    else:
        shapes, sfs = img_shape, scale_factor
    set_decode_image_params(d, shapes, sfs, rescale)
    n_cand = lib().llv_retina_decode_ncand(C.byref(d))
    if n_cand < 0:
        check(n_cand, 'llv_retina_decode_ncand')
    d.n_cand = n_cand
    dev = cls.device
    boxes = torch.empty((batch, n_cand, 4), device=dev, dtype=torch.float32)
    scores = torch.empty((batch, n_cand, num_classes), device=dev, dtype=torch.float32)
    cand = torch.empty((batch, n_cand), device=dev, dtype=torch.int32)
    wsb = lib().llv_retina_decode_workspace_bytes(C.byref(d))
    ws = torch.empty((max(wsb, 256),), device=dev, dtype=torch.uint8)
    d.boxes, d.scores, d.cand_anchor = boxes.data_ptr(), scores.data_ptr(), cand.data_ptr()
    d.workspace, d.workspace_bytes = ws.data_ptr(), wsb
    keep = [cls, reg]
    if clsmax is not None:
        d.clsmax, d.seg_written = clsmax[0].data_ptr(), clsmax[1].data_ptr()
        d.seg_tiles = (num_anchors * num_classes + 127) // 128
        keep += list(clsmax)
    if batch == 1:
        boxes, scores, cand = boxes[0], scores[0], cand[0]
    return DecodePlan(d, boxes, scores, cand, ws, ba, n_cand, keep=keep)


def retina_decode_topk(plan: DecodePlan):
    check(lib().llv_retina_decode_topk(C.byref(plan.desc), _stream()), 'llv_retina_decode_topk')
    return plan.boxes, plan.scores


class NmsBuffers:
    def __init__(self, n, num_classes, max_num, device, batch: int = 1):
        self.n, self.num_classes, self.max_num, self.batch = n, num_classes, max_num, batch
        self.dets = torch.zeros((batch, max_num, 5), device=device, dtype=torch.float32)
        self.labels = torch.zeros((batch, max_num), device=device, dtype=torch.int64)
        self.count = torch.zeros((batch,), device=device, dtype=torch.int32)
        self.keep_flat = torch.zeros((batch, max_num), device=device, dtype=torch.int32)
        self.ws_bytes = lib().llv_multiclass_nms_batched_workspace_bytes(batch, n, num_classes, max_num)
        self.ws = torch.empty((max(self.ws_bytes, 256),), device=device, dtype=torch.uint8)
        if batch == 1:
            self.dets, self.labels, self.keep_flat = self.dets[0], self.labels[0], self.keep_flat[0]


def multiclass_nms_into(boxes, scores, score_thr, iou_thr, bufs: NmsBuffers, num_classes=None):
    """Stream-ordered, no host sync; results land in bufs (dets, labels, count, keep_flat).
    boxes (n,4)/(batch,n,4), scores (n,c_ld)/(batch,n,c_ld)."""
    _need_gpu(boxes, scores)
    assert boxes.dtype == torch.float32 and scores.dtype == torch.float32
    assert boxes.is_contiguous() and scores.is_contiguous()
    n = boxes.shape[-2]
    ncls = num_classes if num_classes is not None else scores.shape[-1]
    assert n <= bufs.n and ncls == bufs.num_classes
    assert (boxes.dim() == 3 and boxes.shape[0] == bufs.batch) or (boxes.dim() == 2 and bufs.batch == 1)
    check(lib().llv_multiclass_nms_batched(_ptr(boxes), _ptr(scores), bufs.batch, n, ncls, scores.shape[-1], float(score_thr),
                                           float(iou_thr), bufs.max_num, _ptr(bufs.dets), _ptr(bufs.labels), _ptr(bufs.count),
                                           _ptr(bufs.keep_flat), _ptr(bufs.ws), bufs.ws_bytes, _stream()),
          'llv_multiclass_nms_batched')


_NMS_BUFFERS = {}


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, return_inds=False):
    """Same signature and return values as mmdet.core.multiclass_nms (bbox_nms.py:7-93).

    multi_scores carries the background column last, which is ignored, as in the reference.
    """
    _need_gpu(multi_bboxes, multi_scores)
    num_classes = multi_scores.shape[1] - 1
    n = multi_bboxes.shape[0]
    class_specific = multi_bboxes.shape[1] > 4        # (n, #class*4): the two-stage bbox heads with reg_class_agnostic=False
    if class_specific and multi_bboxes.shape[1] != 4 * num_classes:
        raise LlvError(f'multiclass_nms: boxes of shape (n, {multi_bboxes.shape[1]}) with {num_classes} classes')
    iou_thr = nms_cfg.get('iou_threshold', nms_cfg.get('iou_thr'))
    if nms_cfg.get('type', 'nms') != 'nms':
        raise LlvError(f"nms type {nms_cfg.get('type')} not supported")
    # the per-class kept lists are merged in LDS: num_classes * cap <= 8192 keys (kMaxMerge, nms.hip) and n <= 16384 boxes. With
    # max_num <= 0 (mmdet's default: keep everything) a class can keep at most n boxes, so cap = min(n, 8192 // num_classes) is
    # exact whenever n * num_classes <= 8192 and refused otherwise (instead of failing inside the library with "unsupported")
    if n > 16384:
        raise LlvError(f'multiclass_nms: {n} boxes; the device NMS takes at most 16384 per image')
    if max_num > 0:
        cap = max_num
    else:
        cap = min(max(n, 1), 8192 // max(num_classes, 1))
        if cap < n:
            raise LlvError(f'multiclass_nms(max_num={max_num}): keeping all of {n} x {num_classes} candidates exceeds the merge limit '
                           f'(num_classes * max_num <= 8192); pass max_num <= {8192 // max(num_classes, 1)}')
    if num_classes * cap > 8192:
        raise LlvError(f'multiclass_nms: num_classes * max_num = {num_classes * cap} > 8192 (merge limit of the device NMS)')
    if lib().llv_multiclass_nms_lds_bytes_ex(n, cap, num_classes) > lib().llv_multiclass_nms_lds_limit():
        raise LlvError(f'multiclass_nms: {n} boxes with max_num={cap} need {lib().llv_multiclass_nms_lds_bytes_ex(n, cap, num_classes)} bytes of LDS per class '
                       f'list (limit {lib().llv_multiclass_nms_lds_limit()}); lists above 8192 boxes take max_num <= 1536')
    # the buffers of a (n, classes, cap) geometry are kept between calls, per device and per stream (reuse is stream-ordered; the
    # results are cloned out below, so a caller on another stream never sees a buffer this one is still writing)
    bkey = (max(n, 1), num_classes, max(cap, 1), str(multi_bboxes.device), torch.cuda.current_stream().cuda_stream)
    bufs = _NMS_BUFFERS.get(bkey)
    if bufs is None:
        if len(_NMS_BUFFERS) >= 16:
            _NMS_BUFFERS.clear()
        bufs = _NMS_BUFFERS[bkey] = NmsBuffers(max(n, 1), num_classes, max(cap, 1), multi_bboxes.device)
    if class_specific:
        bx, sc = multi_bboxes.float().contiguous(), multi_scores.float().contiguous()
        check(lib().llv_multiclass_nms_batched_ex(_ptr(bx), _ptr(sc), None, 1, 1, n, num_classes, sc.shape[-1], float(score_thr),
                                                  float(iou_thr), bufs.max_num, _ptr(bufs.dets), _ptr(bufs.labels), _ptr(bufs.count),
                                                  _ptr(bufs.keep_flat), _ptr(bufs.ws), bufs.ws_bytes, _stream()),
              'llv_multiclass_nms_batched_ex')
    else:
        multiclass_nms_into(multi_bboxes.float().contiguous(), multi_scores.float().contiguous(), score_thr, iou_thr, bufs,
                            num_classes)
    k = int(bufs.count.item())
    dets, labels = bufs.dets[:k].clone(), bufs.labels[:k].clone()
    if return_inds:
        return dets, labels, bufs.keep_flat[:k].long()
    return dets, labels



def delta2bbox(rois: torch.Tensor, deltas: torch.Tensor, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None,
               wh_ratio_clip=16 / 1000, clip_border=True, scale_factor=None) -> torch.Tensor:
    """mmdet delta2bbox (delta_xywh_bbox_coder.py:134-237) on the device: rois (n, 4), deltas (n, 4k) fp32 -> (n, 4k) fp32;
    max_shape (h, w[, c]) clips, scale_factor (4 values) divides (the `rescale=True` of get_bboxes). One launch (llv_delta2bbox)."""
    _need_gpu(rois, deltas)
    n, k4 = deltas.shape
    assert rois.shape == (n, 4) and k4 % 4 == 0
    rois, deltas = rois.float().contiguous(), deltas.float().contiguous()
    out = torch.empty_like(deltas)
    f4 = C.c_float * 4
    clip = 1 if (clip_border and max_shape is not None) else 0
    sf = f4(*[float(v) for v in np.asarray(scale_factor, dtype=np.float32).reshape(-1)[:4]]) if scale_factor is not None else None
    check(lib().llv_delta2bbox(_ptr(rois), _ptr(deltas), _ptr(out), n, k4 // 4, f4(*[float(m) for m in means]), f4(*[float(v) for v in stds]),
                               float(wh_ratio_clip), float(max_shape[0]) if clip else 0.0, float(max_shape[1]) if clip else 0.0, clip, sf,
                               _stream()), 'llv_delta2bbox')
    return out


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float, offset: int = 0):
    """mmcv.ops.nms(boxes, scores, iou_threshold, offset=0) -> (dets (k,5), keep (k,) int64)."""
    _need_gpu(boxes, scores)
    if offset != 0:
        raise LlvError('offset=1 (legacy areas) is not used on this path')
    n = boxes.shape[0]
    boxes = boxes.float().contiguous()
    scores = scores.float().contiguous()
    keep = torch.empty((max(n, 1),), device=boxes.device, dtype=torch.int64)
    count = torch.zeros((1,), device=boxes.device, dtype=torch.int32)
    wsb = lib().llv_nms_workspace_bytes(n)
    ws = torch.empty((max(wsb, 256),), device=boxes.device, dtype=torch.uint8)
    check(lib().llv_nms(_ptr(boxes), _ptr(scores), n, float(iou_threshold), _ptr(keep), _ptr(count), _ptr(ws), wsb,
                        _stream()), 'llv_nms')
    k = int(count.item())
    keep = keep[:k]
    dets = torch.cat([boxes[keep], scores[keep, None]], dim=1)
    return dets, keep


# ----------------------------------------------------------------------------------------------
# temporal
# ----------------------------------------------------------------------------------------------
def embed_aggregate(x_embed, ref_embed, ref, out=None):
    """x_embed (1,H,W,C), ref_embed (N,H,W,C), ref (N,H,W,C) fp16 NHWC -> (1,H,W,C)."""
    _need_gpu(x_embed, ref_embed, ref)
    n, h, w, c = ref.shape
    assert x_embed.shape == (1, h, w, c) and ref_embed.shape == (n, h, w, c)
    for t in (x_embed, ref_embed, ref):
        assert t.dtype == torch.float16 and t.is_contiguous()
    if out is None:
        out = torch.empty((1, h, w, c), device=ref.device, dtype=torch.float16)
    check(lib().llv_embed_aggregate(_ptr(x_embed), _ptr(ref_embed), _ptr(ref), _ptr(out), n, h * w, c, c, c, c,
                                    _stream()), 'llv_embed_aggregate')
    return out


def flow_warp(x, flow, out=None):
    """x (N,H,W,C) fp16 NHWC, flow (N,2,Hf,Wf) fp32 NCHW -> warped (N,H,W,C) fp16."""
    _need_gpu(x, flow)
    n, h, w, c = x.shape
    assert x.dtype == torch.float16 and x.is_contiguous() and flow.dtype == torch.float32 and flow.is_contiguous()
    assert flow.shape[0] == n and flow.shape[1] == 2
    if out is None:
        out = torch.empty_like(x)
    check(lib().llv_flow_warp(_ptr(x), _ptr(flow), _ptr(out), n, h, w, c, c, flow.shape[2], flow.shape[3], _stream()),
          'llv_flow_warp')
    return out


# ----------------------------------------------------------------------------------------------
# training-side
# ----------------------------------------------------------------------------------------------
def sigmoid_focal_loss_fwd(logits, labels, gamma=2.0, alpha=0.25):
    _need_gpu(logits, labels)
    logits = logits.float().contiguous()
    labels = labels.long().contiguous()
    n, c = logits.shape
    out = torch.empty_like(logits)
    check(lib().llv_sigmoid_focal_loss_fwd(_ptr(logits), _ptr(labels), _ptr(out), n, c, gamma, alpha, _stream()),
          'llv_sigmoid_focal_loss_fwd')
    return out


def sigmoid_focal_loss_bwd(logits, labels, gamma=2.0, alpha=0.25):
    _need_gpu(logits, labels)
    logits = logits.float().contiguous()
    labels = labels.long().contiguous()
    n, c = logits.shape
    out = torch.empty_like(logits)
    check(lib().llv_sigmoid_focal_loss_bwd(_ptr(logits), _ptr(labels), _ptr(out), n, c, gamma, alpha, _stream()),
          'llv_sigmoid_focal_loss_bwd')
    return out


def softmax_ce_fwd(logits, labels, class_weight=None):
    """F.cross_entropy(logits, labels, weight=class_weight, reduction='none') -> (n,) fp32 (llv_softmax_ce_fwd)."""
    _need_gpu(logits, labels, class_weight)
    logits = logits.float().contiguous()
    labels = labels.long().contiguous()
    n, c = logits.shape
    out = torch.empty(n, dtype=torch.float32, device=logits.device)
    check(lib().llv_softmax_ce_fwd(_ptr(logits), _ptr(labels), _ptr(class_weight), _ptr(out), n, c, c, _stream()), 'llv_softmax_ce_fwd')
    return out


def softmax_ce_bwd(logits, labels, grad_loss, class_weight=None):
    _need_gpu(logits, labels, grad_loss, class_weight)
    logits = logits.float().contiguous()
    labels = labels.long().contiguous()
    grad_loss = grad_loss.float().contiguous()
    n, c = logits.shape
    out = torch.empty_like(logits)
    check(lib().llv_softmax_ce_bwd(_ptr(logits), _ptr(labels), _ptr(class_weight), _ptr(grad_loss), _ptr(out), n, c, c, _stream()),
          'llv_softmax_ce_bwd')
    return out


def iou_assign(gt_bboxes, anchors, valid=None, pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0,
               match_low_quality=True, gt_max_assign_all=True):
    _need_gpu(gt_bboxes, anchors, valid)
    gt = gt_bboxes.float().contiguous()
    an = anchors.float().contiguous()
    k, n = gt.shape[0], an.shape[0]
    v = valid.to(torch.uint8).contiguous() if valid is not None else None
    assigned = torch.empty((n,), device=an.device, dtype=torch.int64)
    max_ov = torch.empty((n,), device=an.device, dtype=torch.float32)
    wsb = lib().llv_iou_assign_workspace_bytes(k, n)
    ws = torch.empty((max(wsb, 256),), device=an.device, dtype=torch.uint8)
    check(lib().llv_iou_assign(_ptr(gt), k, _ptr(an), _ptr(v), n, pos_iou_thr, neg_iou_thr, min_pos_iou,
                               int(match_low_quality), int(gt_max_assign_all), _ptr(assigned), _ptr(max_ov), _ptr(ws), wsb,
                               _stream()), 'llv_iou_assign')
    return assigned, max_ov


def pack_deconv4x4s2(weight: torch.Tensor, bias: Optional[torch.Tensor] = None, in_map=None) -> PackedConv:
    """nn.ConvTranspose2d(Cin, Cout, 4, stride=2, padding=1) as a 3x3/pad-1 conv to 4*Cout channels followed by
    PixelShuffle(2) (LLV_CONV_PIXSHUF): output (2y+dy, 2x+dx) only sees inputs y-1..y+1, x-1..x+1, with
    tap r (input row y+r-1) <- kernel row {dy=0: r0->3, r1->1; dy=1: r1->2, r2->0} (same for columns), zero elsewhere.
    weight: (Cin, Cout, 4, 4) as torch stores it (flownet_simple.py:95-104 deconv layers)."""
    cin, cout = weight.shape[0], weight.shape[1]
    w = weight.detach().double().cpu()
    w3 = torch.zeros(cout * 4, cin, 3, 3, dtype=torch.float64)
    kmap = {0: {0: 3, 1: 1}, 1: {1: 2, 2: 0}}       # parity -> {3x3 tap: 4x4 kernel index}
    for dy in (0, 1):
        for dx in (0, 1):
            for r, ky in kmap[dy].items():
                for q, kx in kmap[dx].items():
                    w3[dy * 2 + dx::4, :, r, q] = w[:, :, ky, kx].t()
    b3 = None
    if bias is not None:
        b3 = bias.detach().double().cpu().repeat_interleave(4)
    pc = pack_conv(w3.float(), b3.float() if b3 is not None else None, 1, 1, in_map=in_map)
    pc.true_macs_per_pixel = cout * cin * 16 // 4 * 4   # per INPUT pixel: 16 taps x Cin x Cout (4 outputs x 4 taps each)
    return pc


# ------------------------------------------------------------------------------------------------
# modulated deformable convolution
# ------------------------------------------------------------------------------------------------
def pack_dcn_weight(weight: torch.Tensor, bias: Optional[torch.Tensor]) -> PackedConv:
    """A (Co, C, kh, kw) deformable-conv weight as the 1x1 conv over the column tensor of llv_modulated_deform_im2col
    (channel k*Cpad + c, k = i*kw + j): the same tap-major / channel-fastest order a kh x kw conv is packed in."""
    co, c, kh, kw = weight.shape
    cp = round_up(c, 8)
    w = torch.zeros(co, kh * kw, cp, dtype=torch.float32)
    w[:, :, :c] = weight.detach().float().cpu().permute(0, 2, 3, 1).reshape(co, kh * kw, c)
    pc = pack_conv(w.reshape(co, kh * kw * cp, 1, 1), bias, 1, 0)
    pc.true_macs_per_pixel = co * c * kh * kw
    return pc


def modulated_deform_conv2d(x: torch.Tensor, om: torch.Tensor, pc: PackedConv, kh=3, kw=3, stride=1, pad=1, dilation=1,
                            deform_groups=1, flags: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """mmcv.ops.modulated_deform_conv2d on NHWC fp16: x (n,h,w,c); om (n,ho,wo,>=3*dg*K) fp32, the raw conv_offset output
    (offsets then mask logits, see include/llv.h); pc from pack_dcn_weight. Column step on llv_modulated_deform_im2col,
    the GEMM on the implicit-GEMM conv kernel as a 1x1 conv."""
    _need_gpu(x, om, pc.w, pc.bias)
    assert x.dtype == torch.float16 and x.is_contiguous() and om.dtype == torch.float32 and om.is_contiguous()
    n, h, w, c = x.shape
    ho = (h + 2 * pad - (dilation * (kh - 1) + 1)) // stride + 1
    wo = (w + 2 * pad - (dilation * (kw - 1) + 1)) // stride + 1
    assert om.shape[:3] == (n, ho, wo)
    col = torch.empty((n, ho, wo, kh * kw * c), device=x.device, dtype=torch.float16)
    check(lib().llv_modulated_deform_im2col(_ptr(x), _ptr(om), _ptr(col), n, h, w, c, c, om.shape[3], col.shape[3], kh, kw, stride,
                                            pad, dilation, deform_groups, ho, wo, _stream()), 'llv_modulated_deform_im2col')
    return conv2d(col, pc, flags=flags, out=out)


# ------------------------------------------------------------------------------------------------
# RoIAlign (first piece)
# ------------------------------------------------------------------------------------------------
def roi_align(feat: torch.Tensor, rois: torch.Tensor, output_size, spatial_scale: float = 1.0, sampling_ratio: int = 0,
              aligned: bool = True, out_f32: bool = False) -> torch.Tensor:
    """mmcv.ops.roi_align (avg) on NHWC fp16: feat (n,h,w,c), rois (k,5) fp32 = (batch index, x1, y1, x2, y2) ->
    (k, ph, pw, c) fp16 (fp32 with out_f32)."""
    _need_gpu(feat, rois)
    assert feat.dtype == torch.float16 and feat.is_contiguous() and rois.dtype == torch.float32 and rois.shape[1] == 5
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    n, h, w, c = feat.shape
    k = rois.shape[0]
    rois = rois.contiguous()
    out = torch.empty((k, ph, pw, c), device=feat.device, dtype=torch.float32 if out_f32 else torch.float16)
    check(lib().llv_roi_align(_ptr(feat), _ptr(rois), _ptr(out), n, h, w, c, c, k, ph, pw, float(spatial_scale), int(sampling_ratio),
                              1 if aligned else 0, 1 if out_f32 else 0, _stream()), 'llv_roi_align')
    return out
