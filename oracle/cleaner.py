"""FastDVDnet restated as pure functions of a reference-named state_dict (fp32, CPU). Test infrastructure only.
mmtracking/mmtrack/models/cleaners/FastDVDnet.py:7-149. BatchNorm in eval mode (the cleaner is frozen at test
time: llvod_fastdvdnet_darkfarm.py `frozen_modules='cleaner'`), eps = 1e-5."""
import torch
import torch.nn.functional as F


def _bn(x, sd, p, eps=1e-5):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        False, 0.0, eps)


def _cbr(x, sd, conv, bn, stride=1, groups=1):
    return F.relu(_bn(F.conv2d(x, sd[conv + '.weight'], None, stride, 1, 1, groups), sd, bn))


def _cvblock(x, sd, p):          # FastDVDnet.py:7-20: convblock.{0,1,3,4}
    x = _cbr(x, sd, p + '0', p + '1')
    return _cbr(x, sd, p + '3', p + '4')


def denblock_forward(sd, p, in0, in1, in2):
    """DenBlock.forward FastDVDnet.py:102-116"""
    x = torch.cat([in0, in1, in2], dim=1)
    x0 = _cbr(x, sd, p + 'inc.convblock.0', p + 'inc.convblock.1', groups=3)        # :27-28 grouped 9->90
    x0 = _cbr(x0, sd, p + 'inc.convblock.3', p + 'inc.convblock.4')                 # 90->32

    def down(x, q):                                                                 # DownBlock :39-49
        x = _cbr(x, sd, q + 'convblock.0', q + 'convblock.1', stride=2)
        return _cvblock(x, sd, q + 'convblock.3.convblock.')

    def up(x, q):                                                                   # UpBlock :52-61
        x = _cvblock(x, sd, q + 'convblock.0.convblock.')
        x = F.conv2d(x, sd[q + 'convblock.1.weight'], None, 1, 1)
        return F.pixel_shuffle(x, 2)

    x1 = down(x0, p + 'downc0.')
    x2 = down(x1, p + 'downc1.')
    x2 = up(x2, p + 'upc2.')
    x1 = up(x1 + x2, p + 'upc1.')
    x = _cbr(x0 + x1, sd, p + 'outc.convblock.0', p + 'outc.convblock.1')           # OutputCvBlock :64-74
    x = F.conv2d(x, sd[p + 'outc.convblock.3.weight'], None, 1, 1)
    return in1 - x                                                                  # :115


def fastdvdnet_forward(sd, x, prefix=''):
    """FastDVDnet.forward FastDVDnet.py:138-149. x: (N, 15, H, W)."""
    f = [x[:, 3 * m:3 * m + 3] for m in range(5)]
    x20 = denblock_forward(sd, prefix + 'temp1.', f[0], f[1], f[2])
    x21 = denblock_forward(sd, prefix + 'temp1.', f[1], f[2], f[3])
    x22 = denblock_forward(sd, prefix + 'temp1.', f[2], f[3], f[4])
    return denblock_forward(sd, prefix + 'temp2.', x20, x21, x22)
