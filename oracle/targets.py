"""Oracle: target assignment and losses of the detection head.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Vectorised PyTorch-CPU restatement for
the layout every shipped rv-* config uses (one FPN level of stride 1, GAUSSIAN affinity,
``k = inf``, ``normalize_affinities = false``, ``fpn_assignment_method = null``), with
fp64 exactly where the reference uses it.  Reference files followed (relative to
``/root/reference/src/torchbox3d``):

* ``utils/polars.py:9-22``                   polars_to_torch (quaternion -> yaw, fp64)
* ``math/polytope.py:76-107``                cuboids_to_vertices (fp32)
* ``math/polytope.py:14-56``                 compute_interior_points_mask (fp64)
* ``nn/heads/detection_head.py:452-493``     rotate / encode_regression_targets
* ``nn/heads/detection_head.py:496-665``     compute_targets        -> :func:`compute_targets`
* ``math/ops/assignment.py:76-161``          compute_classification_targets / _gaussian
* ``nn/functional/__init__.py:8-27``         varifocal_loss         -> :func:`varifocal_loss`
* ``nn/heads/detection_head.py:202-367``     DetectionHead.loss / compute_*_loss
* ``nn/heads/detection_head.py:370-449``     reduce_multiscale_loss -> :func:`detection_loss`
"""

from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from .decode import decode_range_view

_UNIT = torch.tensor(
    [[+1, +1, +1], [+1, -1, +1], [+1, -1, -1], [+1, +1, -1], [-1, +1, +1], [-1, -1, +1], [-1, -1, -1], [-1, +1, -1]],
    dtype=torch.float32,
)


def annotations_to_cuboids(ann: Tensor) -> Tensor:
    """(M,13) [xyz,lwh,qw,qx,qy,qz,task,offset,batch] fp64 -> (M,10) [xyz,lwh,yaw,task,offset,batch].

    yaw = atan2(2(wz+xy), 1-2(y^2+z^2)) (kornia ``euler_from_quaternion``, ``utils/polars.py:18``).
    """
    if ann.shape[0] == 0:
        return ann.new_empty((0, 10))
    w, x, y, z = ann[:, 6], ann[:, 7], ann[:, 8], ann[:, 9]
    yaw = torch.atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
    return torch.cat([ann[:, :6], yaw[:, None], ann[:, 10:]], dim=-1)


def cuboids_to_vertices(cub: Tensor) -> Tensor:
    """(M,7) fp32 [xyz,lwh,yaw] -> (M,8,3) vertices; rotation via the yaw-only quaternion."""
    half = cub[:, 6] * 0.5
    qw, qz = torch.cos(half), torch.sin(half)
    norm = torch.sqrt(qw * qw + qz * qz)
    qw, qz = qw / norm, qz / norm
    zero, one = torch.zeros_like(qw), torch.ones_like(qw)
    rot = torch.stack(
        [1 - 2 * qz * qz, -2 * qw * qz, zero, 2 * qw * qz, 1 - 2 * qz * qz, zero, zero, zero, one], dim=-1
    ).view(-1, 3, 3)
    verts_obj = cub[:, None, 3:6] / 2.0 * _UNIT[None]
    return verts_obj @ rot.transpose(2, 1) + cub[:, None, :3]


def interior_points_mask(points: Tensor, verts: Tensor) -> Tensor:
    """(N,3) fp64 points, (K,8,3) fp64 vertices -> (K,N) bool (three slab tests, either orientation)."""
    corner = verts[:, [6, 3, 1]]  # (K,3,3)
    ref = verts[:, 2:3]  # (K,1,3)
    uvw = ref - corner
    d_ref = uvw @ ref.transpose(1, 2)  # (K,3,1)
    d_cor = torch.diagonal(uvw @ corner.transpose(1, 2), 0, 1, 2)[..., None]
    d_pts = uvw @ points.T  # (K,3,N)
    a = torch.logical_and(d_ref <= d_pts, d_pts <= d_cor)
    b = torch.logical_and(d_ref >= d_pts, d_pts >= d_cor)
    return torch.logical_or(a, b).all(dim=1)


def encode_regression_targets(cub: Tensor, pts: Tensor, azimuth_invariant: bool) -> Tensor:
    """(M,10) fp64 cuboids, (N,3) fp32 points -> (M,N,8) fp32 targets.

    [R(-az)(ctr - pt), log lwh, sin(yaw - az), cos(yaw - az)]; the centre difference is
    taken in fp32, the log / sin / cos in fp64 then stored as fp32.
    """
    off = cub[:, None, :3].float() - pts
    rots = cub[:, None, 6:7]
    if azimuth_invariant:
        az = torch.atan2(pts[:, 1:2], pts[:, 0:1])
        rots = rots - az
        c, s = torch.cos(az).squeeze(1), torch.sin(az).squeeze(1)
        off = torch.stack([c * off[..., 0] + s * off[..., 1], -s * off[..., 0] + c * off[..., 1], off[..., 2]], dim=-1)
    out = pts.new_zeros((cub.shape[0], pts.shape[0], 8))
    out[:, :, :3] = off
    out[:, :, 3:6] = cub[:, None, 3:6].log()
    out[:, :, 6:7] = torch.sin(rots)
    out[:, :, 7:8] = torch.cos(rots)
    return out


def compute_targets(
    cart: Tensor, annotations: Tensor, num_classes: int, azimuth_invariant: bool = True
) -> Dict[str, Tensor]:
    """Per-pixel labels / instance ids / regression targets for one task at stride 1.

    ``annotations`` is the (M,13) fp64 table in batch order.  Boxes of one sweep are
    ordered by interior-point count ascending (stable); a pixel inside several boxes is
    given to the first of them (fewest points); ``panoptics`` ids are 1-based in that
    order, 0 = background; background label = ``num_classes``.
    """
    B, _, H, W = cart.shape
    out = {
        "points_per_obj": torch.zeros((B, 1, H, W), dtype=torch.int64),
        "panoptics": torch.zeros((B, 1, H, W), dtype=torch.int64),
        "classification_labels": torch.full((B, H, W), num_classes, dtype=torch.int64),
        "regression_targets": torch.zeros((B, 8, H, W)),
    }
    cub = annotations_to_cuboids(annotations)
    if cub.shape[0] == 0:
        return out
    verts = cuboids_to_vertices(cub[:, :7].float())
    for b in cub[:, -1].unique().long().tolist():
        sel = cub[:, -1] == b
        cub_b, verts_b = cub[sel], verts[sel]
        pts = cart[b].flatten(1, 2).t().contiguous()  # (HW,3) fp32
        inside = interior_points_mask(pts.double(), verts_b.double())  # (M,HW)
        n_pts = inside.sum(dim=-1)
        _, perm = n_pts.sort(stable=True, descending=False)
        n_pts, cub_b, inside = n_pts[perm], cub_b[perm], inside[perm]
        M = inside.shape[0]
        ids = torch.where(inside, torch.arange(1, M + 1)[:, None], torch.full((1, 1), M + 1))
        winner = ids.min(dim=0).values  # (HW,) in 1..M, M+1 = none
        fg = winner <= M
        w0 = (winner - 1).clamp(0, M - 1)
        pan = torch.where(fg, winner, torch.zeros_like(winner))
        labels = torch.where(fg, cub_b[:, -2].long()[w0], torch.full_like(winner, num_classes))
        reg = encode_regression_targets(cub_b, pts, azimuth_invariant)  # (M,HW,8)
        reg = reg[w0, torch.arange(pts.shape[0])] * fg[:, None]
        out["classification_labels"][b] = labels.view(H, W)
        out["panoptics"][b, 0] = pan.view(H, W)
        out["regression_targets"][b] = reg.t().reshape(8, H, W)
        out["points_per_obj"][b, 0] = torch.where(fg, n_pts[w0], torch.zeros_like(winner)).view(H, W)
    return out


def classification_targets(
    regressands: Tensor,
    targets: Dict[str, Tensor],
    cart: Tensor,
    mask: Tensor,
    num_classes: int,
    sigma: float = 0.75,
    azimuth_invariant: bool = True,
) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Soft classification targets from the (detached) predictions.

    With ``k = inf`` every pixel of an instance takes part, so the per-instance loop of
    ``assignment.py:118-141`` reduces to a per-pixel map:
    ``aff = exp(-||ctr_pred - ctr_gt|| / sigma^2)`` on instance pixels, 0 elsewhere;
    foreground = ``aff != 0``; background = ``!foreground & mask``.
    Returns (soft targets (B,C,H,W), foreground (B,1,H,W), background, regression weights).
    """
    labels, pan = targets["classification_labels"], targets["panoptics"]
    one_hot = F.one_hot(labels, num_classes + 1).permute(0, 3, 1, 2)[:, :-1].float()
    pds = decode_range_view(regressands.detach(), cart, True)
    gts = decode_range_view(targets["regression_targets"], cart, azimuth_invariant)
    dist = torch.linalg.norm(pds[:, :3] - gts[:, :3], dim=1, keepdim=True)
    aff = torch.exp(-dist / sigma**2) * (pan > 0)
    fg = (aff != 0).to(aff.dtype)
    bg = torch.logical_and(fg.logical_not(), mask)
    return aff * one_hot, fg, bg, one_hot.any(dim=1, keepdim=True)


def varifocal_loss(logits: Tensor, target: Tensor, alpha: float = 0.75, gamma: float = 2.0) -> Tensor:
    """``[t>0] t bce + alpha [t==0] sigmoid(x)^gamma bce`` with bce = BCE-with-logits, no reduction."""
    bce = F.binary_cross_entropy_with_logits(logits, target, reduction="none")
    p = logits.sigmoid()
    return (target > 0.0) * target * bce + alpha * (target == 0) * p.pow(gamma) * bce


def detection_loss(
    logits: Tensor,
    regressands: Tensor,
    cart: Tensor,
    mask: Tensor,
    targets: Dict[str, Tensor],
    num_classes: int,
    classification_weight: float = 1.0,
    regression_weight: float = 1.0,
    coding_weights: Sequence[float] = (1.0,) * 8,
    additive_smoothing: float = 1.0,
    sigma: float = 0.75,
    alpha: float = 0.75,
    gamma: float = 2.0,
    azimuth_invariant: bool = True,
) -> Dict[str, Tensor]:
    """Losses of ``DetectionHead.loss`` + ``reduce_multiscale_loss`` for one stride / one task.

    classification: ``w * VFL * mask / (sum(fg) + smoothing)``; regression:
    ``L1 * w * reg_mask * 1/(points_per_obj + smoothing) [fp64] * mask * coding_w / 8 /
    max(total_objects, 1)``; ``total_objects`` = number of distinct instance ids per sweep
    (``detection_head.py:379-399``: the first value of ``unique()`` is dropped, assumed to
    be the background id 0).  The total is fp64 because of the normaliser.
    """
    soft, fg, bg, reg_w = classification_targets(regressands, targets, cart, mask, num_classes, sigma, azimuth_invariant)
    cls = classification_weight * varifocal_loss(logits, soft, alpha, gamma) * mask
    cw = regressands.new_tensor(list(coding_weights)).view(1, -1, 1, 1)
    norm = (targets["points_per_obj"] + additive_smoothing).double().reciprocal()
    reg = (
        F.l1_loss(regressands, targets["regression_targets"], reduction="none")
        * regression_weight * reg_w * norm * mask * cw / cw.shape[1]
    )
    total_objects = torch.as_tensor([x.unique()[1:].shape[0] for x in targets["panoptics"]]).sum().clamp(1.0)
    total_fg = fg.sum() + additive_smoothing
    cls = cls / total_fg
    reg = reg / total_objects
    coord, dim, rot = reg.sum(dim=[2, 3]).sum(dim=0).split([3, 3, 2], dim=-1)
    coord, dim, rot = coord.sum(), dim.sum(), rot.sum()
    cls_sum = cls.sum()
    return {
        "loss": cls_sum + (coord + dim + rot),
        "classification_loss": cls_sum.detach(),
        "foreground_loss": (cls * fg).sum().detach(),
        "background_loss": (cls * bg).sum().detach(),
        "regression_loss": (coord + dim + rot).detach(),
        "coordinate_loss": coord.detach(),
        "dimension_loss": dim.detach(),
        "rotation_loss": rot.detach(),
        "total_fg": total_fg,
        "total_objects": total_objects,
        "targets": soft,
        "foreground": fg,
        "background": bg,
    }
