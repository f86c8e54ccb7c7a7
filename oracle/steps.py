"""Oracle restatement of the LightningModule glue around the component nets.

The reference's LightningModules cannot be imported here (pytorch_lightning,
torchvision, test_tube are absent), so these functions are written from the
source text.  Each cites the lines it follows.  Test infrastructure only.
"""
import numpy as np
import torch
from torch.nn import functional as F

VIEW_ORDER = (0, 1, 2, 5, 4, 3)   # roadmap_bce_v2.py:58, autoencoder.py:55, spatial_w_rm.py:59


def collate(batch):
    """helper.py:22-23 -- a batch of (sample, target, road_image) triples becomes three tuples."""
    return tuple(zip(*batch))


def wide_stitch(views):
    """[B,6,C,H,W] (or a tuple of B [6,C,H,W]) -> [B,C,H,6W], views re-ordered to a 180 degree sweep.

    roadmap_bce_v2.py:53-64, spatial_w_rm.py:54-65, autoencoder.py:55-57.
    """
    if isinstance(views, (tuple, list)):
        views = torch.stack(tuple(views), dim=0)
    x = views[:, list(VIEW_ORDER)]
    b, n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1, 4).reshape(b, c, h, n * w)


def six_to_one_task(views, rng=np.random):
    """Masked-view pretext task.  autoencoder.py:53-73.

    ``np.random.randint(0, 5)`` has an exclusive upper bound, so only views 0..4 of the
    stitched image are ever blanked -- reproduced, not fixed.  The view width is hard-coded
    306 in the reference (autoencoder.py:60-61); we use the actual view width, which is the
    same number at the reference's input size.
    """
    x = wide_stitch(views).clone()
    vw = views.shape[-1]
    t = int(rng.randint(0, 5))
    y = x[..., t * vw:(t + 1) * vw].clone()
    x[..., t * vw:(t + 1) * vw] = 0.0
    return x, y, t


def ae_loss(encoder, decoder, views, rng=np.random, masks=None):
    """BasicAE._run_step: mse_loss(y, decoder(encoder(x))).  autoencoder.py:78-93."""
    x, y, _ = six_to_one_task(views, rng)
    m = masks or {}
    z = encoder(x, m.get("enc", (None, None)))
    y_hat = decoder(z, m.get("dec", (None, None)))
    return F.mse_loss(y, y_hat), y_hat


def roadmap_forward(encoder, head, sample, masks=(None, None)):
    """RoadMapBCE.forward -> (logits [B,800,800], sigmoid(logits)).  roadmap_bce_v2.py:66-81."""
    z = encoder(wide_stitch(sample), masks)
    y = F.linear(z, head.weight, head.bias)
    y = y.reshape(y.size(0), 800, 800)
    return y, torch.sigmoid(y)


def roadmap_bce_loss(encoder, head, batch, masks=(None, None)):
    """RoadMapBCE._run_step: BCE-with-logits over the flattened maps.  roadmap_bce_v2.py:83-108."""
    sample, _target, road_image = batch
    target = torch.stack(tuple(road_image), dim=0).to(head.weight.dtype)
    logits, probs = roadmap_forward(encoder, head, sample, masks)
    b = target.size(0)
    loss = F.binary_cross_entropy_with_logits(logits.reshape(b, -1), target.reshape(b, -1))
    return loss, target, logits, probs


def roadmap_mse_loss(encoder, head, batch, masks=(None, None)):
    """RoadMap (roadmap_pretrain_ae.py:67-110): sigmoid inside forward, mse_loss(target, pred)."""
    sample, _target, road_image = batch
    target = torch.stack(tuple(road_image), dim=0).to(head.weight.dtype)
    _, probs = roadmap_forward(encoder, head, sample, masks)
    return F.mse_loss(target, probs), target, probs


def bbox_forward(encoder, space_map, box_merge, views, rm):
    """BBSpatialRoadMap.forward: [B,6,3,H,W],[B,1,800,800] -> [B,800,800].  spatial_w_rm.py:67-83."""
    assert encoder.c3_only
    space_rep = space_map(views)
    ssr = encoder(wide_stitch(views))
    return box_merge(ssr, space_rep, rm).squeeze(1)


def bbox_loss(encoder, space_map, box_merge, views, rm, target_img, mse=False):
    """spatial_w_rm.py:122-131: binary_cross_entropy on probabilities, or mse_loss(pred, target)."""
    pred = bbox_forward(encoder, space_map, box_merge, views, rm)
    b = pred.size(0)
    p, t = pred.reshape(b, -1), target_img.reshape(b, -1)
    return (F.mse_loss(p, t) if mse else F.binary_cross_entropy(p, t)), pred


def joint_loss(encoder, head, space_map, box_merge, batch, target_img, masks=(None, None), branch=None, box_branch=None):
    """BASELINE config 4, the joint roadmap + bounding-box step.  The reference has no joint model (its ``c3_only`` switch,
    components.py:44-45, makes the two heads exclusive users of the encoder): per SURVEY.md 8(d) this is the composition of
    RoadMapBCE._run_step (roadmap_bce_v2.py:66-108) and BBSpatialRoadMap._run_step (spatial_w_rm.py:67-131) on ONE pass of the
    encoder's conv stack, losses added -- so every encoder gradient is the sum of the two single-head gradients.
    -> (loss, loss_roadmap, loss_boxes)."""
    from .branch import PLAIN
    branch = PLAIN if branch is None else branch
    box_branch = branch if box_branch is None else box_branch
    sample, _target, road_image = batch
    views = torch.stack(tuple(sample), dim=0) if isinstance(sample, (tuple, list)) else sample
    target_rm = torch.stack(tuple(road_image), dim=0).to(head.weight.dtype)
    b = target_rm.size(0)
    feat = encoder.conv_stack(wide_stitch(views), branch)                              # one pass, two consumers
    logits = F.linear(encoder.tail(feat, masks, branch), head.weight, head.bias)       # roadmap_bce_v2.py:75-81
    loss_rm = F.binary_cross_entropy_with_logits(logits.reshape(b, -1), target_rm.reshape(b, -1))
    pred = box_merge(feat, space_map(views, branch=box_branch), target_rm.unsqueeze(1), branch=box_branch).squeeze(1)   # spatial_w_rm.py:67-83
    loss_bb = F.binary_cross_entropy(pred.reshape(b, -1), target_img.reshape(b, -1))   # spatial_w_rm.py:131
    return loss_rm + loss_bb, loss_rm, loss_bb


def threat_score(a, b):
    """helper.py:74-77."""
    tp = (a * b).sum()
    return tp * 1.0 / (a.sum() + b.sum() - tp)
