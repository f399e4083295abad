import sys, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import *
import lowlightenvironmentvideoobjectdetection_amd.models
from lowlightenvironmentvideoobjectdetection_amd import registry as reg, ops
from oracle.detector import *
from oracle.anchors import AnchorGenerator
from oracle.bbox import DeltaXYWHBBoxCoder
dev = torch.device('cuda:0')
h, w, ncls = 128, 160, 4
det = randomize(reg.build_detector(retinanet_cfg(num_classes=ncls)), seed=7, cls_bias=-1.5)
img = smooth_image(1, h, w, seed=8)
calibrate_cls(det, img)
meta = img_meta(h - 8, w, h, w, scale=1.25)
sd = {k: v.clone() for k, v in det.state_dict().items()}
feats = retinanet_extract_feat(sd, img)
rc, rr = retina_head_forward(sd, feats, 4)
det = det.to(dev).eval()
res = det(img=[img.to(dev)], img_metas=[[meta]], return_loss=False, rescale=True)
plan = det._get_plan(img.to(dev))
pyr = plan.pyramid
for i in range(5):
    c = pyr.level(i, plan.cls_buf).float().cpu().permute(0, 3, 1, 2)
    r = pyr.level(i, plan.reg_buf).float().cpu().permute(0, 3, 1, 2)
    print('lvl', i, 'cls rel', ((c - rc[i]).abs().max() / rc[i].abs().max()).item(), 'abs', (c - rc[i]).abs().max().item(),
          'reg rel', ((r - rr[i]).abs().max() / rr[i].abs().max()).item(), 'logit range', rc[i].min().item(), rc[i].max().item())
ag = AnchorGenerator(**DEFAULT_ANCHOR)
b_ref, s_ref, idx_ref = decode_candidates(rc, rr, ag, DeltaXYWHBBoxCoder(), meta['img_shape'], meta['scale_factor'], ncls, 1000, True)
dp = plan.post.decode[0]
got = dp.cand_anchor.cpu().long()
print('ncand', len(got), len(idx_ref), 'same idx frac', (got == idx_ref).float().mean().item(),
      'set overlap', len(set(got.tolist()) & set(idx_ref.tolist())) / len(idx_ref))
print('score stats ref', s_ref[:, :ncls].max(1)[0].min().item(), s_ref.max().item(), 'n > thr', int((s_ref[:, :ncls] > 0.05).sum()))
ref = retinanet_simple_test(sd, img, meta, num_classes=ncls, rescale=True)
allref = np.concatenate(ref); allres = np.concatenate(res[0])
print('ref scores', np.sort(allref[:, 4])[::-1][:5], np.sort(allref[:, 4])[:5])
print('res scores', np.sort(allres[:, 4])[::-1][:5], np.sort(allres[:, 4])[:5])
for thr in (0.9, 0.7, 0.5):
    print('match iou', thr, match_detections(res[0], ref, thr, 0.05))
# oracle NMS on the device candidates
from oracle.nms import multiclass_nms
sc = torch.cat([dp.scores.cpu(), torch.zeros(dp.n_cand, 1)], 1)
d2, l2 = multiclass_nms(dp.boxes.cpu(), sc, 0.05, dict(type='nms', iou_threshold=0.5), 100)
nb = plan.post.nms[0]
k = int(nb.count.item())
print('device nms == oracle nms on device candidates:', torch.equal(d2, nb.dets[:k].cpu()), torch.equal(l2, nb.labels[:k].cpu()))
same = (got == idx_ref).nonzero().view(-1)
print('box diff on same-idx cands', (dp.boxes.cpu()[same] - b_ref[same]).abs().max().item())
print('ref box sample', b_ref[same[:3]]); print('dev box sample', dp.boxes.cpu()[same[:3]])
print('desc', dp.desc.clip, dp.desc.img_h, dp.desc.img_w, list(dp.desc.scale_factor), dp.desc.rescale)
print('ref dets c0', ref[0][:3]); print('res dets c0', res[0][0][:3])
print('labels hist ref', [len(r) for r in ref], 'res', [len(r) for r in res[0]])
