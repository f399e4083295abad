"""Golden vectors for the evaluation and deformable-conv rows, produced by the oracles:
    python tests/golden/make_golden_eval.py
eval_small.npz: a seeded detection / annotation set in flat arrays (dets [n,7] = image, class, x1, y1, x2, y2, score; gts [m,6];
crowd / ignore regions [k,6]) with eval_map's per-class AP (PASCAL and ImageNet rules, with and without scale ranges) and the COCO
protocol's 12 statistics. dcn_small.npz: inputs and output of modulated_deform_conv2d."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))


def unflatten(dets, gts, ign, n_img, K):
    det_results = [[dets[(dets[:, 0] == i) & (dets[:, 1] == c)][:, 2:].astype(np.float32) for c in range(K)] for i in range(n_img)]
    anns = []
    for i in range(n_img):
        g, q = gts[gts[:, 0] == i], ign[ign[:, 0] == i]
        anns.append(dict(bboxes=g[:, 2:].astype(np.float32), labels=g[:, 1].astype(np.int64),
                         bboxes_ignore=q[:, 2:].astype(np.float32), labels_ignore=q[:, 1].astype(np.int64)))
    return det_results, anns


def main():
    from oracle import coco_eval, evaluation
    from oracle.dcn import modulated_deform_conv2d
    g = np.random.RandomState(20261004)
    n_img, K = 30, 4
    dets, gts, ign = [], [], []
    for i in range(n_img):
        n = g.randint(1, 7)
        xy = g.rand(n, 2) * 400
        wh = np.exp(g.rand(n, 2) * 4 + 1.5)
        lab = g.randint(0, K, n)
        for j in range(n):
            gts.append([i, lab[j], xy[j, 0], xy[j, 1], xy[j, 0] + wh[j, 0], xy[j, 1] + wh[j, 1]])
            for _ in range(g.randint(0, 4)):
                b = np.array([xy[j, 0], xy[j, 1], xy[j, 0] + wh[j, 0], xy[j, 1] + wh[j, 1]]) + g.randn(4) * g.choice([1.0, 5.0, 15.0])
                b[2:] = np.maximum(b[2:], b[:2] + 1)
                dets.append([i, lab[j] if g.rand() < 0.9 else g.randint(0, K), *b, np.round(g.rand(), 3)])
        for _ in range(g.randint(0, 3)):
            p = g.rand(2) * 400
            dets.append([i, g.randint(0, K), p[0], p[1], p[0] + g.rand() * 80 + 3, p[1] + g.rand() * 80 + 3, np.round(g.rand(), 3)])
        if g.rand() < 0.4:
            p = g.rand(2) * 300
            ign.append([i, g.randint(0, K), p[0], p[1], p[0] + 120, p[1] + 90])
    dets, gts, ign = np.array(dets, np.float32), np.array(gts, np.float32), np.array(ign, np.float32).reshape(-1, 6)
    det_results, anns = unflatten(dets, gts, ign, n_img, K)
    out = dict(dets=dets, gts=gts, ign=ign, n_img=n_img, K=K)
    for name, kw in [('voc', dict()), ('vid', dict(dataset='vid')), ('voc07', dict(dataset='voc07')),
                     ('scales', dict(scale_ranges=[(0, 32), (32, 96), (96, 1e5)]))]:
        m, res = evaluation.eval_map(det_results, anns, iou_thr=0.5, **kw)
        out[f'map_{name}'] = np.asarray(m, np.float64)
        out[f'ap_{name}'] = np.stack([np.atleast_1d(r['ap']) for r in res]).astype(np.float64)
    c = coco_eval.coco_bbox_eval([[d.astype(np.float64) for d in im] for im in det_results], anns)
    out['coco_stats'] = c['stats']
    np.savez_compressed(os.path.join(HERE, 'eval_small.npz'), **out)
    print('eval:', {k: np.round(v, 4).tolist() for k, v in out.items() if k.startswith('map_')}, np.round(c['stats'][:6], 4))

    t = torch.Generator().manual_seed(7)
    n, ch, co, h, w, dg = 2, 16, 8, 7, 9, 2
    x = torch.randn(n, ch, h, w, generator=t).half().float()
    wgt = (torch.randn(co, ch, 3, 3, generator=t) * 0.1).half().float()
    bias = torch.randn(co, generator=t) * 0.1
    om = torch.cat([torch.randn(n, 2 * dg * 9, h, w, generator=t) * 1.5, torch.randn(n, dg * 9, h, w, generator=t)], 1)
    y = modulated_deform_conv2d(x, om[:, :2 * dg * 9], torch.sigmoid(om[:, 2 * dg * 9:]), wgt, bias, 1, 1, 1, 1, dg)
    np.savez_compressed(os.path.join(HERE, 'dcn_small.npz'), x=x.numpy(), w=wgt.numpy(), b=bias.numpy(), om=om.numpy(), y=y.numpy(), dg=dg)
    print('dcn:', tuple(y.shape), float(y.abs().max()))


if __name__ == '__main__':
    main()
