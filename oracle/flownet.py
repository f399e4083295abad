"""FlowNetSimple restated as a pure function of a reference-named state_dict (fp32, CPU). Test infrastructure only.
mmtracking/mmtrack/models/motion/flownet_simple.py:149-251."""
import torch
import torch.nn.functional as F

CONV = dict(inplanes=(6, 64, 128, 256, 512, 512), kernel_size=(7, 5, 5, 3, 3, 3), num_convs=(1, 1, 2, 2, 2, 2))


def prepare_imgs(imgs, img_meta, img_scale_factor, flow_mean=(0.411, 0.432, 0.450), flow_std=(255.0, 255.0, 255.0)):
    """flownet_simple.py:149-190"""
    mean = torch.tensor(img_meta['img_norm_cfg']['mean']).repeat(2)[None, :, None, None]
    std = torch.tensor(img_meta['img_norm_cfg']['std']).repeat(2)[None, :, None, None]
    fmean = torch.tensor(flow_mean).repeat(2)[None, :, None, None]
    fstd = torch.tensor(flow_std).repeat(2)[None, :, None, None]
    flow_img = imgs * std + mean
    flow_img = flow_img / fstd - fmean
    flow_img[:, :, img_meta['img_shape'][0]:, :] = 0.0
    flow_img[:, :, :, img_meta['img_shape'][1]:] = 0.0
    return F.interpolate(flow_img, scale_factor=img_scale_factor, mode='bilinear', align_corners=False)


def _crop_like(x, target):
    return x if x.shape[2:] == target.shape[2:] else x[:, :, :target.size(2), :target.size(3)]


def flownet_forward(sd, imgs, img_meta, img_scale_factor=0.5, flow_scale_factor=5.0, prefix=''):
    """flownet_simple.py:192-244"""
    p = prefix
    x = prepare_imgs(imgs, img_meta, img_scale_factor)
    outs = []
    for i in range(6):
        for j in range(CONV['num_convs'][i]):
            k = CONV['kernel_size'][i] if j == 0 else (3 if i == 2 else CONV['kernel_size'][i])
            x = F.leaky_relu(F.conv2d(x, sd[f'{p}conv{i + 1}.{j}.conv.weight'], sd[f'{p}conv{i + 1}.{j}.conv.bias'],
                                      2 if j == 0 else 1, (k - 1) // 2), 0.1)
        if i + 1 in (2, 3, 4, 5, 6):
            outs.append(x)
    num_outs = len(outs)
    concat_out = outs[-1]
    for i, lvl in zip(range(1, num_outs)[::-1], (5, 4, 3, 2)):
        flow = F.conv2d(concat_out, sd[f'{p}predict_flow{lvl + 1}.conv.weight'], None, 1, 1)
        upflow = _crop_like(F.conv_transpose2d(flow, sd[f'{p}upsample_flow{lvl}.conv.weight'], None, 2, 1), outs[i - 1])
        deconv_out = _crop_like(F.leaky_relu(F.conv_transpose2d(concat_out, sd[f'{p}deconv{lvl}.conv.weight'], None, 2, 1), 0.1),
                                outs[i - 1])
        concat_out = torch.cat((outs[i - 1], deconv_out, upflow), dim=1)
    flow = F.conv2d(concat_out, sd[f'{p}predict_flow.conv.weight'], None, 1, 1)
    flow = F.interpolate(flow, scale_factor=4 / img_scale_factor, mode='bilinear', align_corners=False)
    flow = flow * (4 / img_scale_factor)
    return flow * flow_scale_factor
