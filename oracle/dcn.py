"""TEST INFRASTRUCTURE (oracle): fp32 CPU restatement of the authors' DCN-based temporal fusion.

Follows /root/reference/mmtracking/mmtrack/models/aggregators/denoising_aggregator.py
  :10-34    DenseLayer            cat([x, relu(conv3x3(x))], 1)
  :37-66    RDB                   x + lff(layers(x))
  :69-111   ModulatedDCNPack      conv_offset(extra_feat) -> chunk 3 -> offset = cat(o1, o2), mask = sigmoid -> modulated_deform_conv2d
  :114-152  TemporalAttentionFusion
  :155-200  DenoisingAggregator

`modulated_deform_conv2d` is an op of mmcv-full (pinned >=1.2.4,<=1.4.0, authors ran 1.3.0), which is not under
/root/reference and not installed. Its published algorithm (modulated_deformable_im2col + GEMM; CUDA kernels
modulated_deformable_im2col_gpu_kernel / dmcn_im2col_bilinear of mmcv 1.3) is restated in `modulated_deform_conv2d`
below. Parity unpinned: no reference test holds a vector for it. What pins the restatement instead (tests/test_dcn.py):
zero offsets with mask logits -> plain conv2d * sigmoid; integer offsets -> conv2d of the shifted image; the border rule.

Only tests/ may import this file.
"""
import torch
import torch.nn.functional as F


def _bilinear_mmcv(img, py, px):
    """dmcn_im2col_bilinear on a whole map: img (C, H, W); py, px (...,) sampling positions. Returns (C, ...).
    Positions outside (-1, H) x (-1, W) give zero; each of the four corners counts only if it lies inside the image."""
    c, h, w = img.shape
    inside = (py > -1) & (px > -1) & (py < h) & (px < w)
    y0, x0 = torch.floor(py), torch.floor(px)
    ly, lx = py - y0, px - x0
    hy, hx = 1 - ly, 1 - lx
    y0, x0 = y0.long(), x0.long()
    y1, x1 = y0 + 1, x0 + 1

    def corner(yy, xx, wt):
        ok = inside & (yy >= 0) & (yy <= h - 1) & (xx >= 0) & (xx <= w - 1)
        v = img[:, yy.clamp(0, h - 1), xx.clamp(0, w - 1)]
        return v * (wt * ok.to(img.dtype))
    return corner(y0, x0, hy * hx) + corner(y0, x1, hy * lx) + corner(y1, x0, ly * hx) + corner(y1, x1, ly * lx)


def modulated_deform_conv2d(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deform_groups=1):
    """x (N,C,H,W); offset (N, 2*dg*K, Ho, Wo) with channel (g*2K + 2k) = dy, +1 = dx; mask (N, dg*K, Ho, Wo) (already
    sigmoid-ed by the caller); weight (Co, C/groups, kh, kw). Returns (N, Co, Ho, Wo)."""
    n, c, h, w = x.shape
    co, _, kh, kw = weight.shape
    K = kh * kw
    ho = (h + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    wo = (w + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    assert offset.shape == (n, 2 * deform_groups * K, ho, wo) and mask.shape == (n, deform_groups * K, ho, wo)
    cg = c // deform_groups
    oy = torch.arange(ho, dtype=x.dtype).view(ho, 1) * stride - padding
    ox = torch.arange(wo, dtype=x.dtype).view(1, wo) * stride - padding
    cols = x.new_zeros(n, c, K, ho, wo)
    for b in range(n):
        for g in range(deform_groups):
            img = x[b, g * cg:(g + 1) * cg]
            for k in range(K):
                i, j = divmod(k, kw)
                py = oy + i * dilation + offset[b, g * 2 * K + 2 * k]
                px = ox + j * dilation + offset[b, g * 2 * K + 2 * k + 1]
                cols[b, g * cg:(g + 1) * cg, k] = _bilinear_mmcv(img, py, px) * mask[b, g * K + k]
    out = x.new_zeros(n, co, ho, wo)
    cpg, opg = c // groups, co // groups
    for gg in range(groups):
        wg = weight[gg * opg:(gg + 1) * opg].reshape(opg, cpg * K)
        cg_cols = cols[:, gg * cpg:(gg + 1) * cpg].reshape(n, cpg * K, ho * wo)
        out[:, gg * opg:(gg + 1) * opg] = torch.matmul(wg, cg_cols).view(n, opg, ho, wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


def _conv(x, sd, p, relu=False, padding=1):
    y = F.conv2d(x, sd[p + '.weight'], sd.get(p + '.bias'), padding=padding)
    return F.relu(y) if relu else y


def modulated_dcn_pack(sd, p, x, extra_feat, deform_groups=8):
    """denoising_aggregator.py:103-111."""
    out = _conv(extra_feat, sd, p + '.conv_offset')
    o1, o2, mask = torch.chunk(out, 3, dim=1)
    return modulated_deform_conv2d(x, torch.cat((o1, o2), dim=1), torch.sigmoid(mask), sd[p + '.weight'], sd[p + '.bias'],
                                   1, 1, 1, 1, deform_groups)


def temporal_attention_fusion(sd, p, x, emb_nums=3, deform_groups=8):
    """denoising_aggregator.py:135-152."""
    x = _conv(x, sd, p + '.conv1', relu=True)
    feat = []
    for i in range(x.shape[0]):
        x_ref = x[[i]].repeat(x.shape[0], 1, 1, 1)
        x_set = _conv(torch.cat([x, x_ref], dim=1), sd, p + '.offset_conv')
        x_dcn = modulated_dcn_pack(sd, p + '.dcn_pack', x, x_set, deform_groups)
        x_cor = x_dcn * x_ref
        for e in range(emb_nums):
            x_cor = _conv(x_cor, sd, f'{p}.emb_conv.{e}')
        x_cor = torch.softmax(x_cor, dim=0)
        feat.append(torch.sum(x_cor * x, dim=0, keepdim=True))
    return _conv(torch.cat(feat, dim=0), sd, p + '.conv2', relu=True)


def rdb(sd, p, x, num_layers):
    """denoising_aggregator.py:37-66."""
    y = x
    for l in range(num_layers):
        y = torch.cat([y, _conv(y, sd, f'{p}.layers.{l}.conv', relu=True)], 1)
    return x + _conv(y, sd, p + '.lff', padding=0)


def denoising_aggregator_forward(sd, x, num_blocks=1, num_layers=6, emb_nums=3, with_rdb=True, with_taf=True, prefix=''):
    """denoising_aggregator.py:186-200. x (N, C, H, W): the N frames' feature maps."""
    p = prefix
    identity = x
    x = _conv(x, sd, p + 'conv1', relu=True)
    x_f = x
    if with_rdb:
        for b in range(num_blocks):
            x = rdb(sd, f'{p}rdbs.{b}', x, num_layers)
    x = _conv(x + x_f, sd, p + 'conv2', relu=True)
    x_m = x
    if with_taf:
        x = temporal_attention_fusion(sd, p + 'taf', x, emb_nums)
    x = _conv(x + x_m, sd, p + 'conv3', relu=True)
    return x + identity


def denoising2_aggregator_forward(sd, x_noise, all_x, in_channel, layer_name, rdb_blocks, taf_embs, downsample, with_rdb, with_taf, prefix=''):
    """Denoising2Aggergator.forward (mmtracking/mmtrack/models/aggregators/denoising2_aggregator.py:214-244): per backbone stage
    conv1 (on the stage's noisy features, from the second stage on concatenated with the previous stage's output) -> RDBs (3 dense
    layers each) -> TemporalAttentionFusion -> + the stage's features; conv2 (stride 2 where `downsample`) feeds the next stage;
    every entry of all_x gets all_x[-1] + the last stage's output. x_noise: tuple of (N, C_i, H_i, W_i)."""
    p = prefix + 'layers.'
    x_noise_out, x_list = [], []
    ns = len(in_channel)
    for i in range(ns):
        l = layer_name[i]
        f = x_noise[0] if i == 0 else torch.cat([x_noise[i], x_list[i - 1]], 1)
        x = _conv(f, sd, f'{p}{l}_conv1')
        if with_rdb[i]:
            for b in range(rdb_blocks[i]):
                x = rdb(sd, f'{p}{l}_rdb.{b}', x, 3)
        if with_taf[i]:
            x = temporal_attention_fusion(sd, f'{p}{l}_taf', x, taf_embs[i])
        x_noise_out.append(x + x_noise[i])
        w, bias = sd[f'{p}{l}_conv2.weight'], sd.get(f'{p}{l}_conv2.bias')
        x_list.append(F.conv2d(x if i == ns - 1 else x + x_noise[i], w, bias, 2 if downsample[i] else 1, 1))
    return tuple(x_noise_out), tuple(all_x[-1] + x_list[-1] for _ in all_x)
