"""CPU restatement of the reference's supervised loss `get_loss` -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(omni-pq_amd/models/loss_helper_pq.py -> csrc/loss_rows.hip) never does.  Pinned by tests/golden/get_loss.npz, which
tests/golden/make_golden_get_loss.py generates by running the reference's own get_loss (every loss term, the labels,
the collision count and the gradient with respect to every prediction tensor).

Plain torch on the CPU (f32, autograd for the gradients), written as a restatement rather than a transcription: the
reference recomputes the proposal -> ground-truth assignment seven times (once per prediction head; identical results,
loss_helper_pq.py:52-71 / :198-242) and walks B x 256 x 256 Python scalars for the physical-constraint term (:392-408);
here the assignment is made once, the seven heads are one stacked tensor and the constraint term is one broadcast.

    vote loss                      loss_helper_pq.py:24-44
    objectness labels + loss       :47-86
    box + semantic class loss      :89-192
    quad labels + score loss       :196-246
    quad centre / normal / size    :249-299
    box footprint, projection      :302-353
    physical constraints           :355-410
    total                          :412-486
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import loss_oracle

FAR_THRESHOLD = 0.6                     # :17-21
NEAR_THRESHOLD = 0.3
OBJECTNESS_CLS_WEIGHTS = (0.2, 0.8)
GT_VOTE_FACTOR = 3
QUAD_CLS_WEIGHTS = (0.4, 0.6)
NOT_SOLID = (5, 6, 8, 11)               # door, window, picture, curtain (:354-357)


def prefixes(num_layer=6):
    return ["proposal_", "last_"] + [f"{i}head_" for i in range(num_layer - 1)]


def smoothl1(err):
    """models/utils/losses.py:5-13 with delta = 1."""
    a = err.abs()
    return torch.where(a < 1.0, 0.5 * a * a, a - 0.5)


def assign(query, gt, num_gt):
    """Nearest ground truth of every query point and the NEAR / FAR labelling (:60-71, :207-217)."""
    d1, i1, _, _ = loss_oracle.nn_distance(query.detach().numpy(), gt.detach().numpy())
    d1, i1 = torch.from_numpy(d1), torch.from_numpy(i1)
    e = torch.sqrt(d1 + 1e-6)
    label = ((e < NEAR_THRESHOLD) & (i1 < num_gt)).long()
    mask = ((e < NEAR_THRESHOLD) | (e > FAR_THRESHOLD)).float()
    assignment = torch.where(label == 0, torch.full_like(i1, gt.shape[1] - 1), i1)
    return label, mask, assignment


def stack(ep, key, pfx):
    return torch.stack([ep[p + key] for p in pfx])                   # (P, B, K, ...)


def weighted_ce(scores, label, weights):
    """nn.CrossEntropyLoss(weight, reduction='none') on (..., C) scores."""
    logp = F.log_softmax(scores, dim=-1)
    picked = torch.gather(logp, -1, label.unsqueeze(-1)).squeeze(-1)
    if weights is None:
        return -picked
    return -picked * torch.tensor(weights)[label]


def vote_loss(ep):
    B, S, _ = ep["seed_xyz"].shape
    inds = ep["seed_inds"].long()
    m = torch.gather(ep["vote_label_mask"], 1, inds).float()
    gt = torch.gather(ep["vote_label"], 1, inds[..., None].expand(-1, -1, 3 * GT_VOTE_FACTOR))
    gt = (gt + ep["seed_xyz"].repeat(1, 1, 3)).view(B * S, GT_VOTE_FACTOR, 3)
    votes = ep["vote_xyz"].view(B * S, -1, 3)
    d = (votes[:, :, None, :] - gt[:, None, :, :]).abs().sum(-1)     # l1=True  (utils/nn_distance.py:56-57)
    dist2 = d.min(dim=1)[0]                                          # (B*S, GT_VOTE_FACTOR)
    per_seed = dist2.min(dim=1)[0].view(B, S)
    return (per_seed * m).sum() / (m.sum() + 1e-6)


def footprint(size, center):
    """The four xy corners of every box (:302-321): (+,+), (+,-), (-,+), (-,-)."""
    sx = torch.tensor([0.5, 0.5, -0.5, -0.5], dtype=size.dtype)
    sy = torch.tensor([0.5, -0.5, 0.5, -0.5], dtype=size.dtype)
    x = (size[..., 0:1] * sx).float() + center[..., 0:1]
    y = (size[..., 1:2] * sy).float() + center[..., 1:2]
    return torch.stack([x, y], dim=-1)                               # (B, K, 4, 2) f32


def physical_constraints(ep, config):
    """:355-410.  For every scene: the corners of the boxes that are objects and not door / window / picture / curtain,
    against every predicted quad labelled as a quad; a corner behind the quad's line (delta < 0) whose projection lies
    within size[0] of the quad centre is penalised by its depth.  Each scene's sum is divided by its number of boxes."""
    means = torch.from_numpy(np.asarray(config.mean_size_arr))                      # float64, as the reference keeps it
    cls = ep["last_size_scores"].argmax(-1)
    res = torch.gather(ep["last_size_residuals"], 2, cls[..., None, None].expand(-1, -1, 1, 3)).squeeze(2)
    size = means[cls] + res                                                          # f64 + f32 -> f64
    corners = footprint(size, ep["last_center"])                                     # f32
    sem = torch.gather(ep["sem_cls_label"], 1, ep["last_object_assignment"])
    solid = torch.ones_like(sem, dtype=torch.bool)
    for c in NOT_SOLID:
        solid &= sem != c
    use = (ep["last_objectness_label"] > 0) & solid                                  # (B, K)
    n_box = use.sum(1).float()                                                       # (B,)
    qc, nv, qs = ep["last_quad_center"], ep["last_normal_vector"], ep["last_quad_size"]
    a, b = nv[..., 0:1], nv[..., 1:2]                                                # (B, Q, 1)
    d = -(a * qc[..., 0:1] + b * qc[..., 1:2])
    px = corners[..., 0].flatten(1)[:, None, :]                                      # (B, 1, 4K)
    py = corners[..., 1].flatten(1)[:, None, :]
    k = -(a * px + b * py + d)
    tx, ty = px + a * k, py + b * k
    w = torch.sqrt((tx - qc[..., 0:1]) ** 2 + (ty - qc[..., 1:2]) ** 2)
    inside = (w < qs[..., 0:1]).float()
    delta = px * a + py * b + d
    pen = torch.relu(-delta) * inside                                                # (B, Q, 4K)
    live = use.repeat_interleave(4, dim=1)[:, None, :] & (ep["last_quad_label"] > 0)[..., None]
    pen = pen * live.float()
    per_scene = pen.sum((1, 2)) / n_box.clamp(min=1.0)
    collisions = ((pen > 1e-4) & live).sum()
    return per_scene.sum(), collisions


def get_loss(ep, config, pc_loss=True, num_layer=6):
    """-> (loss, end_points) with the reference's keys (:412-486)."""
    pfx = prefixes(num_layer)
    P = len(pfx)
    ep["vote_loss"] = vote_loss(ep) if "vote_xyz" in ep else 0.0

    # ---- objectness (:47-86)
    o_label, o_mask, o_assign = assign(ep["aggregated_vote_xyz"], ep["center_label"][:, :, 0:3], ep["num_gt_boxes"])
    for p in pfx:
        ep[p + "objectness_label"], ep[p + "objectness_mask"], ep[p + "object_assignment"] = o_label, o_mask, o_assign
    ce = weighted_ce(stack(ep, "objectness_scores", pfx), o_label.expand(P, -1, -1), OBJECTNESS_CLS_WEIGHTS)
    obj = (ce * o_mask).sum((1, 2)) / (o_mask.sum() + 1e-6)                                         # (P,)

    # ---- boxes and semantic classes (:89-192)
    pos = o_label.float()
    n_pos = pos.sum() + 1e-6
    take = lambda key: torch.gather(ep[key], 1, o_assign)                                          # noqa: E731
    take3 = lambda key: torch.gather(ep[key], 1, o_assign[..., None].expand(-1, -1, 3))            # noqa: E731
    center = (smoothl1(take3("center_label") - stack(ep, "center", pfx)) * pos[..., None]).sum((1, 2, 3)) / n_pos
    h_cls = take("heading_class_label")
    heading_cls = (weighted_ce(stack(ep, "heading_scores", pfx), h_cls.expand(P, -1, -1), None) * pos).sum((1, 2)) / n_pos
    h_res = take("heading_residual_label") / (np.pi / config.num_heading_bin)
    h_pred = torch.gather(stack(ep, "heading_residuals_normalized", pfx), -1, h_cls.expand(P, -1, -1)[..., None])[..., 0]
    heading_reg = (smoothl1(h_pred - h_res) * pos).sum((1, 2)) / n_pos
    s_cls = take("size_class_label")
    size_cls = (weighted_ce(stack(ep, "size_scores", pfx), s_cls.expand(P, -1, -1), None) * pos).sum((1, 2)) / n_pos
    s_pred = torch.gather(stack(ep, "size_residuals_normalized", pfx), 3,
                          s_cls.expand(P, -1, -1)[..., None, None].expand(-1, -1, -1, 1, 3))[:, :, :, 0]
    means = torch.from_numpy(np.asarray(config.mean_size_arr).astype(np.float32))
    s_res = take3("size_residual_label") / means[s_cls]
    size_reg = (smoothl1(s_pred - s_res) * pos[..., None]).sum((1, 2, 3)) / n_pos
    sem = (weighted_ce(stack(ep, "sem_cls_scores", pfx), take("sem_cls_label").expand(P, -1, -1), None) * pos).sum((1, 2)) / n_pos
    box = center + 0.1 * heading_cls + heading_reg + 0.1 * size_cls + size_reg

    # ---- quads (:196-299)
    q_label, q_mask, q_assign = assign(ep["aggregated_sample_xyz"], ep["gt_quad_centers"][:, :, 0:3], ep["num_gt_quads"])
    for p in pfx:
        ep[p + "quad_label"], ep[p + "quad_mask"], ep[p + "quad_assignment"] = q_label, q_mask, q_assign
    ce = weighted_ce(stack(ep, "quad_scores", pfx), q_label.expand(P, -1, -1), QUAD_CLS_WEIGHTS)
    q_score = (ce * q_mask).sum((1, 2)) / (q_mask.sum() + 1e-6)
    qpos = q_label.float()
    n_q = qpos.sum() + 1e-6
    qa3 = q_assign[..., None].expand(-1, -1, 3)
    q_center = (smoothl1(torch.gather(ep["gt_quad_centers"][:, :, 0:3], 1, qa3) - stack(ep, "quad_center", pfx))
                * qpos[..., None]).sum((1, 2, 3)) / n_q
    gt_n = torch.gather(ep["gt_normal_vectors"], 1, qa3)
    cos = F.cosine_similarity(stack(ep, "normal_vector", pfx), gt_n.expand(P, -1, -1, -1), dim=3)
    q_vec = ((1.0 - cos) * qpos).sum((1, 2)) / n_q
    gt_s = torch.gather(ep["gt_quad_sizes"], 1, q_assign[..., None].expand(-1, -1, 2))
    q_size = (smoothl1(stack(ep, "quad_size", pfx) - gt_s) * qpos[..., None]).sum((1, 2, 3)) / n_q

    for i, p in enumerate(pfx):
        ep[p + "objectness_loss"] = obj[i]
        ep[p + "center_loss"], ep[p + "heading_cls_loss"], ep[p + "heading_reg_loss"] = center[i], heading_cls[i], heading_reg[i]
        ep[p + "size_cls_loss"], ep[p + "size_reg_loss"] = size_cls[i], size_reg[i]
        ep[p + "box_loss"], ep[p + "sem_cls_loss"] = box[i], sem[i]
        ep[p + "quad_scores_loss"], ep[p + "quad_center_loss"] = q_score[i], q_center[i]
        ep[p + "normal_vector_loss"], ep[p + "quad_size_loss"] = q_vec[i], q_size[i]
    ep["objectness_loss"], ep["box_loss"], ep["sem_cls_loss_sum"] = obj.sum(), box.sum(), sem.sum()
    ep["quad_score_loss_sum"] = q_score.sum()
    ep["quad_center_loss_sum"], ep["quad_vector_loss_sum"], ep["quad_size_loss_sum"] = q_center.sum(), q_vec.sum(), q_size.sum()
    ep["quad_loss_sum"] = ep["quad_center_loss_sum"] + ep["quad_vector_loss_sum"] + ep["quad_size_loss_sum"]
    if pc_loss:
        pc, collisions = physical_constraints(ep, config)
    else:
        pc, collisions = 0.0, 0
    ep["physical_constraints_loss"], ep["collisions"] = pc, collisions
    object_loss = ep["box_loss"] + 0.1 * ep["sem_cls_loss_sum"] + 0.5 * ep["objectness_loss"]
    quad_loss = ep["quad_loss_sum"] + 0.5 * ep["quad_score_loss_sum"]
    loss = 10 * (pc + ep["vote_loss"] + 1.0 / (num_layer + 1) * (0.9 * object_loss + 0.1 * quad_loss))
    ep["loss"] = loss
    return loss, ep
