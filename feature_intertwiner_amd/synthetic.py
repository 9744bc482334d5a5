"""Synthetic COCO-shaped training batches: there is no dataset and no
pretrained checkpoint on the GPU box, so inputs are seeded random tensors of the right
shape and the network is randomly initialised.

Because a randomly initialised RPN proposes nothing object-like, `SyntheticProposals`
supplies jittered copies of the ground-truth boxes as EXTERNAL proposals that compete with
the RPN's candidates before NMS, so that NMS sees realistically clustered boxes and the
target sampler finds enough positive RoIs to fill TRAIN_ROIS_PER_IMAGE -- i.e. the heads,
RoIAlign and the intertwiner run at their full configured size.  Every stage still executes
(RPN convs, selection of the best candidates, decode, clip, NMS, sampling); only the VALUES
of part of the candidate boxes are synthetic.
"""
import math

import torch


def synthetic_batch(batch, image_size, n_gt=20, num_classes=81, mini_mask=56, device="cuda", seed=2000):
    """images N(0,1)*64 [b,3,S,S]; 20 GT boxes per image, log-uniform side 16..512 px, aspect
    0.5..2, classes uniform in 1..80, filled-ellipse mini-masks (56x56); pixel (y1,x1,y2,x2)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    S = float(image_size)
    images = torch.randn(batch, 3, image_size, image_size, generator=g) * 64.0
    side = torch.exp(torch.empty(batch, n_gt).uniform_(math.log(16.0), math.log(min(512.0, S / 2)), generator=g))
    asp = torch.exp(torch.empty(batch, n_gt).uniform_(math.log(0.5), math.log(2.0), generator=g))
    h = side / asp.sqrt()
    w = side * asp.sqrt()
    y1 = torch.rand(batch, n_gt, generator=g) * (S - h)
    x1 = torch.rand(batch, n_gt, generator=g) * (S - w)
    gt_boxes = torch.stack([y1, x1, y1 + h, x1 + w], 2)
    gt_class_ids = torch.randint(1, num_classes, (batch, n_gt), generator=g)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, mini_mask), torch.linspace(-1, 1, mini_mask), indexing="ij")
    ellipse = ((yy ** 2 + xx ** 2) <= 1.0).float()
    gt_masks = ellipse.view(1, 1, mini_mask, mini_mask).expand(batch, n_gt, -1, -1).contiguous()
    return (images.to(device), gt_class_ids.to(device), gt_boxes.to(device), gt_masks.to(device))


class SyntheticProposals(object):
    """External proposal source for the synthetic benchmark / tests (MaskRCNN.external_proposals): every call
    returns `n_plant` jittered copies of the ground-truth boxes per image as rows (y1, x1, y2, x2, score) with
    scores 2.0 .. 1.5 -- above any RPN probability -- so that they take the top of the pre-NMS ranking and the
    best PRE_NMS_LIMIT - n_plant RPN candidates follow in their own order.  The proposal layer itself knows
    nothing about this: it receives the rows as `extra_dets` (layers.proposal_layer)."""

    def __init__(self, gt_boxes, image_size, n_plant=3000, jitter=0.2, seed=7, cycle=0):
        """cycle = n > 0: n sets of rows are drawn here, once, and handed out in turn -- the benchmark's inputs are
        resident in HBM before the timed region starts, like its images and ground truth (drawing a set takes ~35 small
        launches that are the data source's, not the framework's)."""
        self.gt_boxes = gt_boxes
        self.size = float(image_size)
        self.n_plant = n_plant
        self.jitter = jitter
        self.gen = torch.Generator(device=gt_boxes.device).manual_seed(seed)
        self.pos = 0
        self.sets = [self._draw() for _ in range(cycle)] if cycle > 0 else None

    def get_state(self):
        return self.gen.get_state(), self.pos

    def set_state(self, state):
        self.gen.set_state(state[0])
        self.pos = state[1]

    def __call__(self):
        if self.sets is not None:
            rows = self.sets[self.pos % len(self.sets)]
            self.pos += 1
            return rows
        return self._draw()

    def _draw(self):
        b, G = self.gt_boxes.size(0), self.gt_boxes.size(1)
        k = self.n_plant
        dev = self.gt_boxes.device
        which = torch.randint(0, G, (b, k), device=dev, generator=self.gen)
        gt = torch.gather(self.gt_boxes, 1, which.unsqueeze(2).expand(-1, -1, 4))
        h = gt[..., 2] - gt[..., 0]
        w = gt[..., 3] - gt[..., 1]
        cy = gt[..., 0] + 0.5 * h
        cx = gt[..., 1] + 0.5 * w
        r = lambda: (torch.rand(b, k, device=dev, generator=self.gen) * 2 - 1) * self.jitter
        cy = cy + r() * h
        cx = cx + r() * w
        h = h * (1 + r())
        w = w * (1 + r())
        planted = torch.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], 2).clamp(0, self.size)
        scores = torch.linspace(2.0, 1.5, k, device=dev).view(1, k, 1).expand(b, -1, -1)
        return torch.cat([planted, scores], 2).contiguous()
