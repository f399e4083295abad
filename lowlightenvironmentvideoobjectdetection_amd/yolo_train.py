"""Training step of the YOLOv3 stand-in (config #5) on the device: SingleStageDetector.forward_train (mmdetection/mmdet/models/detectors/
single_stage.py:68-96) for YOLOV3 = Darknet-53 (backbones/darknet.py), YOLOV3Neck (necks/yolo_neck.py:110-131) and the head's bridge /
prediction convs (dense_heads/yolo_head.py:150-170) as ONE training graph (train_engine.TrainGraph over yolo.build_yolo_body(train=True)),
then YOLOV3Head.loss (yolo_head.py:369-462: GridAssigner, YOLOBBoxCoder.encode, sigmoid cross entropy / MSE terms) on its prediction maps.

What the tape needed beyond the RetinaNet graph: the LeakyReLU epilogue and the ResBlock's add-after-activation (llv_act_bwd), the neck's
2x nearest upsampling + concatenation as a CUSTOM tape entry (llv_upsample2x_nhwc_bwd + a strided add). BatchNorm runs with its running
statistics folded into the conv (Darknet's norm_eval=True, darknet.py:66,197-205) and trainable affine parameters, as in the ResNet graph."""
import torch

from . import engine, ops, train_engine
from ._lib import LlvError
from .yolo import build_yolo_body


class YoloTrainGraph(train_engine.TrainGraph):
    # the head's terms are SUMS (reduction='sum', configs/yolo/yolov3_d53_mstrain-608_273e_coco.py:27-60): d(loss)/d(activation) reaches the
    # hundreds, and 512 x that leaves fp16's range (the RetinaNet graph's losses are means: 512 there) -- 8 keeps both ends representable
    def __init__(self, det, batch, height, width, device, loss_scale=8.0):
        bb, neck, head = det.backbone, det.neck, det.bbox_head
        if height % 32 or width % 32:
            raise LlvError('YOLOv3 input must be a multiple of 32 (Pad(size_divisor=32))')
        self._init_common(det, dict(det.named_parameters()), dict(det.named_buffers()), device, batch, loss_scale)
        b = engine.Builder(device)
        self.builder = b
        sd = engine._cpu_sd(det.state_dict())
        self.x_in = b.buf(batch, height, width, 8)
        self.pred, self.level_hw = build_yolo_body(b, sd, bb, neck, head, self.x_in, train=True)
        self.n_out = head.num_anchors * head.num_attrib
        fs = getattr(bb, 'frozen_stages', -1)                      # darknet.py:181-188: conv1 + conv_res_block1..fs
        frozen = ['backbone.conv1'] if fs >= 0 else []
        frozen += [f'backbone.conv_res_block{i}.' for i in range(1, fs + 1)]
        self._finish(b, frozen)

    def forward_maps(self, img):
        """img (N, 3, H, W) fp32 normalised on the device -> per-level prediction maps (N, A * (5 + C), h, w) fp32."""
        ops.pack_input(img.float().contiguous(), 8, out=self.x_in)
        self.plan.run()
        return [pm[..., :self.n_out].permute(0, 3, 1, 2).contiguous() for pm in self.pred]

    def backward_maps(self, dmaps):
        self.zero_act_grads()
        for pm, d in zip(self.pred, dmaps):
            if d is None:
                continue
            g = self.grads[train_engine._sptr(pm)]
            g[..., :self.n_out] = (d.permute(0, 2, 3, 1) * self.loss_scale).to(torch.float16)
        self.backward_ops(list(reversed(self.tape)))


class YoloFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, img, *params):
        graph.repack()
        ctx.graph = graph
        return tuple(graph.forward_maps(img))

    @staticmethod
    def backward(ctx, *dmaps):
        g = ctx.graph
        g.zero_param_grads()
        g.backward_maps(dmaps)
        pg = g.param_grads()
        inv = 1.0 / g.loss_scale
        return (None, None) + tuple((pg[k] * inv) if k in pg else None for k in g.trainable)


def yolo_forward_train(det, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
    if img.device.type != 'cuda':
        raise LlvError('forward_train runs on the MI355X (conv forward/backward kernels); no CPU fallback')
    n, _, h, w = img.shape
    if len(gt_bboxes) != n or len(gt_labels) != n or len(img_metas) != n:
        raise LlvError(f'YOLOV3.forward_train: {n} images, {len(gt_bboxes)} gt_bboxes, {len(gt_labels)} gt_labels, {len(img_metas)} img_metas')
    for k in [k for k in det._plans if not (isinstance(k, tuple) and k and k[0] == 'train')]:      # inference plans hold packed copies
        del det._plans[k]
    graph = det._plan(('train', n, h, w, str(img.device)), lambda: YoloTrainGraph(det, n, h, w, img.device))
    maps = YoloFunction.apply(graph, img, *[graph.params[k] for k in graph.trainable])
    dev = img.device
    return det.bbox_head.loss(list(maps), [b.to(dev) for b in gt_bboxes], [l.to(dev) for l in gt_labels], img_metas, gt_bboxes_ignore)
