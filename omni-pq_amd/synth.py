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
