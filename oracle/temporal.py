"""Temporal aggregation restated (fp32 CPU). Test infrastructure only.
  EmbedAggregator.forward  mmtracking/mmtrack/models/aggregators/embed_aggregator.py:50-81
  flow_warp_feats          mmtracking/mmtrack/core/motion/flow.py:4-41 (verbatim operation order, including the
                           division by W/H -- not W-1/H-1 -- before an align_corners=True grid_sample)"""
import torch
import torch.nn.functional as F


def embed_aggregate(sd, x, ref_x, num_convs=1, prefix=''):
    """x (1,C,H,W), ref_x (N,C,H,W). embed convs: ConvModule(3x3, pad 1), ReLU on all but the last (:33-48)."""
    def embed(t):
        for i in range(num_convs):
            t = F.conv2d(t, sd[f'{prefix}embed_convs.{i}.conv.weight'], sd[f'{prefix}embed_convs.{i}.conv.bias'], 1,
                         (sd[f'{prefix}embed_convs.{i}.conv.weight'].shape[-1] - 1) // 2)
            if i != num_convs - 1:
                t = F.relu(t)
        return t
    x_embed = embed(x)
    x_embed = x_embed / x_embed.norm(p=2, dim=1, keepdim=True)
    ref_x_embed = embed(ref_x)
    ref_x_embed = ref_x_embed / ref_x_embed.norm(p=2, dim=1, keepdim=True)
    ada_weights = torch.sum(ref_x_embed * x_embed, dim=1, keepdim=True)
    ada_weights = ada_weights.softmax(dim=0)
    return torch.sum(ref_x * ada_weights, dim=0, keepdim=True)


def embed_aggregate_tail(x_embed, ref_x_embed, ref_x):
    """Lines :70-81 only (what llv_embed_aggregate computes): inputs are the embed-conv outputs."""
    x_embed = x_embed / x_embed.norm(p=2, dim=1, keepdim=True)
    ref_x_embed = ref_x_embed / ref_x_embed.norm(p=2, dim=1, keepdim=True)
    w = torch.sum(ref_x_embed * x_embed, dim=1, keepdim=True).softmax(dim=0)
    return torch.sum(ref_x * w, dim=0, keepdim=True)


def flow_warp_feats(x, flow):
    """flow.py:4-41"""
    assert len(x.shape) == 4
    assert len(flow.shape) == 4 and flow.shape[1] == 2
    scale_factor = float(x.shape[-1]) / flow.shape[-1]
    flow = F.interpolate(flow, scale_factor=scale_factor, mode='bilinear', align_corners=False)
    flow = flow * scale_factor
    H, W = x.shape[-2:]
    h_grid, w_grid = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    h_grid = h_grid.float()[None, None, ...]
    w_grid = w_grid.float()[None, None, ...]
    grid = torch.cat((w_grid, h_grid), dim=1)
    grid = grid + flow
    grid[:, 0] = grid[:, 0] / W * 2 - 1
    grid[:, 1] = grid[:, 1] / H * 2 - 1
    grid = grid.permute(0, 2, 3, 1)
    return F.grid_sample(x, grid, padding_mode='border', align_corners=True)
