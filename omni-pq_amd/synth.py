"""Synthetic point clouds for tests and bench (no dataset ships with the reference).

The reference feeds `PQ_Transformer` ScanNet / ARKitScenes scans re-sampled to a fixed
number of points (scannet/scannet_detection_dataset.py:86-312, utils/pc_util.py:36-44 --
sampling *with replacement* when a scan is short, hence exact duplicate points).  These
generators imitate the geometry that matters to the hot path: surface-like point sets in a
room-sized box, duplicates, and (optionally) points inside the 1e-3 squared-norm ball that
furthest-point sampling ignores (sampling_gpu.cu:105-106).

All randomness comes from a CPU `torch.Generator` seeded per scene, so the same
(seed, scene) always yields the same cloud regardless of batch composition.
"""
import torch


def _uniform(gen, shape, lo, hi):
    return torch.rand(shape, generator=gen, dtype=torch.float32) * (hi - lo) + lo


def _box_surface(gen, n, size, center):
    """n points uniform on the 6 faces of an axis-aligned box (area-weighted)."""
    sx, sy, sz = [float(v) for v in size]
    areas = torch.tensor([sy * sz, sy * sz, sx * sz, sx * sz, sx * sy, sx * sy])
    face = torch.multinomial(areas / areas.sum(), n, replacement=True, generator=gen)
    uvw = torch.rand((n, 3), generator=gen, dtype=torch.float32) - 0.5
    axis = face // 2
    side = (face % 2).to(torch.float32) - 0.5
    uvw[torch.arange(n), axis] = side
    pts = uvw * torch.tensor([sx, sy, sz], dtype=torch.float32)
    return pts + torch.tensor(center, dtype=torch.float32)


def room_scene(gen, n):
    """One ScanNet-like scene: room shell + box-shaped furniture, jitter, 2 % duplicates."""
    w, l = [float(v) for v in _uniform(gen, (2,), 3.0, 8.0)]
    h = float(_uniform(gen, (1,), 2.4, 3.0))
    n_shell = int(n * 0.6)
    pts = [_box_surface(gen, n_shell, (w, l, h), (0.0, 0.0, h / 2))]
    n_obj = int(torch.randint(8, 21, (1,), generator=gen))
    rest = n - n_shell
    per = [rest // n_obj + (1 if i < rest % n_obj else 0) for i in range(n_obj)]
    for cnt in per:
        sz = _uniform(gen, (3,), 0.3, 2.0)
        sz[2] = min(float(sz[2]), h * 0.9)
        cx = float(_uniform(gen, (1,), -w / 2 + 0.2, w / 2 - 0.2))
        cy = float(_uniform(gen, (1,), -l / 2 + 0.2, l / 2 - 0.2))
        pts.append(_box_surface(gen, cnt, sz.tolist(), (cx, cy, float(sz[2]) / 2)))
    pts = torch.cat(pts, 0)
    pts = pts + torch.randn(pts.shape, generator=gen, dtype=torch.float32) * 0.005
    pts = pts[torch.randperm(n, generator=gen)]
    n_dup = n // 50
    dst = torch.randperm(n, generator=gen)[:n_dup]
    src = torch.randint(0, n, (n_dup,), generator=gen)
    pts[dst] = pts[src]
    return pts.contiguous()


def uniform_scene(gen, n):
    pts = torch.rand((n, 3), generator=gen, dtype=torch.float32)
    return (pts * torch.tensor([4.0, 4.0, 2.5])).contiguous()


def make_clouds(seed, batch, n, extra_channels=0, kind="room", first_scene=0):
    """-> (batch, n, 3 + extra_channels) float32 CPU tensor.

    Scene i of the batch is generated from seed `1000 * seed + first_scene + i`, so ranks of a
    data-parallel job can draw disjoint scenes with `first_scene = rank * batch`.
    """
    out = []
    for i in range(batch):
        gen = torch.Generator().manual_seed(1000 * int(seed) + first_scene + i)
        xyz = room_scene(gen, n) if kind == "room" else uniform_scene(gen, n)
        if extra_channels:
            colour = _uniform(gen, (n, min(3, extra_channels)), -0.5, 0.5)
            feats = [colour]
            if extra_channels > 3:
                nrm = torch.randn((n, extra_channels - 3), generator=gen, dtype=torch.float32)
                nrm = nrm / nrm.norm(dim=1, keepdim=True).clamp_min(1e-6)
                feats.append(nrm)
            xyz = torch.cat([xyz] + feats, 1)
        out.append(xyz)
    return torch.stack(out, 0).contiguous()


def adversarial_cloud(seed, batch, n):
    """Clouds built to hit the reference kernels' edge cases in one go:

    * exact duplicate points (ties in FPS / 3-NN, resolved by the launch geometry);
    * points with squared norm <= 1e-3 (skipped by FPS, sampling_gpu.cu:105-106), incl. index 0;
    * an isolated far-away point (empty balls for every other centre);
    * a tight cluster with more than `nsample` neighbours inside any radius.
    """
    gen = torch.Generator().manual_seed(77000 + int(seed))
    pts = torch.rand((batch, n, 3), generator=gen, dtype=torch.float32) * 2.0 - 1.0
    q = max(n // 16, 1)
    pts[:, :q] = pts[:, q:2 * q]                      # duplicates of other points
    pts[:, 0] = 0.0                                   # the FPS seed itself is "skipped"
    pts[:, 2 * q:2 * q + q // 2] *= 0.01              # inside the 1e-3 ball
    pts[:, 3 * q:4 * q] = pts[:, 3 * q:3 * q + 1] + \
        torch.randn((batch, q, 3), generator=gen) * 1e-3   # dense cluster
    pts[:, -1] = torch.tensor([50.0, 50.0, 50.0])     # outlier
    # a regular lattice patch: many exactly equal distances
    g = torch.arange(4, dtype=torch.float32) * 0.25
    lat = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    k = min(lat.shape[0], n // 4)
    pts[:, 5 * q:5 * q + k] = lat[:k] + 0.5
    return pts.contiguous()


def make_labels(point_clouds, seed, max_obj=64, max_quad=32, num_class=18, mean_size_arr=None):
    """Synthetic supervision for `point_clouds` (B, N, 3+C) in the format of the reference's data loader
    (scannet/scannet_detection_dataset.py:256-304: MAX_NUM_OBJ = 64 box slots, MAX_NUM_QUAD = 32 quad slots, zero padded):
    8-23 boxes per scene centred on points of the cloud, point votes towards the nearest box centre within 0.6 m, the four
    walls of the cloud's xy bounding box plus up to four partitions as quads.  -> dict of CPU tensors with the keys
    `get_loss` reads (models/loss_helper_pq.py:412-486)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    xyz = point_clouds[..., :3].cpu().numpy().astype(np.float64)
    B, N, _ = xyz.shape
    if mean_size_arr is None:
        mean_size_arr = np.full((num_class, 3), 0.8)
    out = {"center_label": np.zeros((B, max_obj, 3), np.float32),
           "heading_class_label": np.zeros((B, max_obj), np.int64),
           "heading_residual_label": np.zeros((B, max_obj), np.float32),
           "size_class_label": np.zeros((B, max_obj), np.int64),
           "size_residual_label": np.zeros((B, max_obj, 3), np.float32),
           "sem_cls_label": np.zeros((B, max_obj), np.int64),
           "num_gt_boxes": np.zeros((B, 1), np.int64),
           "vote_label": np.zeros((B, N, 9), np.float32),
           "vote_label_mask": np.zeros((B, N), np.int64),
           "gt_quad_centers": np.zeros((B, max_quad, 3), np.float32),
           "gt_normal_vectors": np.zeros((B, max_quad, 3), np.float32),
           "gt_quad_sizes": np.zeros((B, max_quad, 2), np.float32),
           "num_gt_quads": np.zeros((B, 256), np.int64),           # the count once per quad proposal (dataset :301-304)
           "num_total_quads": np.zeros((B, 256), np.int64),
           "horizontal_quads": np.zeros((B, 4, 4, 3), np.float32)}
    for b in range(B):
        n = int(rs.randint(8, 24))
        centres = xyz[b, rs.choice(N, n, replace=False)]
        cls = rs.randint(0, num_class, size=n)
        out["center_label"][b, :n] = centres
        out["size_class_label"][b, :n] = cls
        out["sem_cls_label"][b, :n] = cls
        out["size_residual_label"][b, :n] = 0.15 * rs.randn(n, 3) * mean_size_arr[cls]
        out["num_gt_boxes"][b, 0] = n
        d = ((xyz[b][:, None, :] - centres[None]) ** 2).sum(-1)
        near = d.argmin(1)
        inside = d.min(1) < 0.36
        vote = (centres[near] - xyz[b]) * inside[:, None]
        out["vote_label"][b] = np.tile(vote, (1, 3))
        out["vote_label_mask"][b] = inside
        lo, hi = xyz[b, :, :2].min(0), xyz[b, :, :2].max(0)
        mid, ext = (lo + hi) / 2, hi - lo
        qc = [[lo[0], mid[1], 1.3], [hi[0], mid[1], 1.3], [mid[0], lo[1], 1.3], [mid[0], hi[1], 1.3]]
        qn = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0]]
        qs = [[ext[1], 2.6], [ext[1], 2.6], [ext[0], 2.6], [ext[0], 2.6]]
        for _ in range(int(rs.randint(0, 5))):
            ang = rs.rand() * 2 * np.pi
            qc.append([mid[0] + (rs.rand() - 0.5) * ext[0] * 0.6, mid[1] + (rs.rand() - 0.5) * ext[1] * 0.6, 1.3])
            qn.append([np.cos(ang), np.sin(ang), 0.0])
            qs.append([1 + rs.rand(), 2.6])
        m = len(qc)
        out["gt_quad_centers"][b, :m], out["gt_normal_vectors"][b, :m], out["gt_quad_sizes"][b, :m] = qc, qn, qs
        out["num_gt_quads"][b, :] = m
        out["num_total_quads"][b, :] = m + 2                        # + floor and ceiling
        ring = np.array([[lo[0], lo[1]], [hi[0], lo[1]], [lo[0], hi[1]], [hi[0], hi[1]]])
        out["horizontal_quads"][b, 0] = np.concatenate([ring, np.full((4, 1), 2.6)], 1)
        out["horizontal_quads"][b, 1] = np.concatenate([ring, np.zeros((4, 1))], 1)
    return {k: torch.from_numpy(v) for k, v in out.items()}
