"""Golden vectors for the native point-set operators that do NOT pass through this repo's oracle:
outputs of the REFERENCE'S OWN pure-PyTorch implementations

    farthest_point_sample   /root/reference/models/utils/pointnet_util.py:71-91
    query_ball_point        /root/reference/models/utils/pointnet_util.py:94-114
    3-NN by sorting         /root/reference/models/utils/pointnet_util.py:310-317  (square_distance + sort)

imported in place (no bytecode written, nothing copied) and run here, in the build container.  They differ from
the CUDA ops by construction -- random start index, no small-norm skip, `>` r^2 (inclusive ball), empty ball ->
index N, `argmax` / `sort` tie rules, matmul-form distances -- so the inputs are chosen where the two formulations
must agree, and every row that could still depend on a rounding-level difference is masked out EXPLICITLY:

  * coordinates uniform in [0.5, 3.5]^3: every norm >> 1e-3 (no skip), no exact duplicates;
  * FPS: the start index is pinned to 0 (torch.randint patched during the call); a case is kept only if the
    oracle's three distance-contraction forms all reproduce it (no near-tie decided by one rounding);
  * ball query: rows with a neighbour whose d^2 lies within 2e-5 (absolute: the matmul-form distance of the reference
    carries an error of a few 1e-6 at these coordinates) of r^2 are masked; empty balls do not occur (the centre itself
    is a member);
  * 3-NN: rows whose 1st..4th neighbour distances are not separated by more than 2e-5 are masked.

Stored (tests/golden/pure_ops.npz): generator seeds + an f64 checksum of every input, the reference's index
outputs as int32, and the row masks.  The oracle (CPU suite) and the HIP kernels (GPU suite) are compared with
this DATA (tests/test_pure_golden.py); a case at BASELINE's full size (1 x 40 000 -> 2048, r = 0.2, nsample 64)
is included.

    python tests/golden/make_golden_pure.py
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("OMNIPQ_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "models"))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(8)


def load_reference_pure():
    """models/utils/pointnet_util.py imports the reference's pointnet2_utils at module level (whose CUDA extension does
    not exist here): the import guard the reference itself provides lets it load without `_ext`
    (pointnet2_utils.py:25-33, as tests/golden/make_golden.py does); only pure-torch functions are called below."""
    import builtins
    builtins.__POINTNET2_SETUP__ = True
    sys.path.insert(0, os.path.join(REF, "pointnet2"))
    from utils import pointnet_util
    assert pointnet_util.__file__.startswith(REF), pointnet_util.__file__
    return pointnet_util


def cloud(seed, b, n):
    gen = torch.Generator().manual_seed(seed)
    return torch.rand((b, n, 3), generator=gen) * 3.0 + 0.5


# The reference's square_distance is the matmul form |a|^2 + |b|^2 - 2ab: with |a|^2 ~ 10 its f32 rounding error is a few
# 1e-6 ABSOLUTE, whatever d^2 is.  Decisions closer than this band to their threshold are masked.
BAND = 2e-5

CASES = [
    # name, seed, B, N, npoint, radius, nsample, unknown points for 3-NN
    ("small", 123, 3, 1500, 200, 0.35, 24, 700),
    ("sa2like", 321, 2, 2048, 1024, 0.4, 32, 1024),
    ("full40k", 4001, 1, 40000, 2048, 0.2, 64, 4096),
]


def main():
    from oracle import oracle_ext
    pure = load_reference_pure()
    out = {}
    for name, seed, B, N, npoint, radius, nsample, n_unknown in CASES:
        while True:
            xyz = cloud(seed, B, N)
            real_randint = torch.randint
            torch.randint = lambda lo, hi, size, **kw: torch.zeros(size, dtype=kw.get("dtype", torch.long))
            try:
                fps = pure.farthest_point_sample(xyz, npoint).int()
            finally:
                torch.randint = real_randint
            same = True
            for form in (0, 1, 2):
                oracle_ext.set_dist_form(form)
                same &= bool(torch.equal(oracle_ext.furthest_point_sampling(xyz, npoint), fps))
            oracle_ext.set_dist_form(1)
            if same:
                break
            print(f"{name}: seed {seed} has a rounding-sensitive FPS pick, trying the next")
            seed += 1
        new_xyz = torch.gather(xyz, 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        # ---- ball query (scene by scene: the reference materialises an (S, N) int64 matrix)
        ball, ball_ok = [], []
        for b in range(B):
            q, p = new_xyz[b:b + 1], xyz[b:b + 1]
            ball.append(pure.query_ball_point(radius, nsample, p, q).int())
            d2 = ((q.double()[0][:, None, :] - p.double()[0][None, :, :]) ** 2).sum(-1)
            near = ((d2 - radius * radius).abs() <= BAND).any(-1)
            ball_ok.append(~near)
        ball = torch.cat(ball)
        ball_ok = torch.stack(ball_ok)
        assert int(ball.max()) < N, "an empty ball: the centre itself should always be inside"
        # ---- 3-NN of the first n_unknown points among the sampled centres
        unknown = xyz[:, :n_unknown].contiguous()
        d = pure.square_distance(unknown, new_xyz)
        dsort, isort = d.sort(dim=-1)
        nn_idx = isort[:, :, :3].int()
        d4 = ((unknown.double()[:, :, None, :] - new_xyz.double()[:, None, :, :]) ** 2).sum(-1).sort(dim=-1)[0][:, :, :4]
        nn_ok = ((d4[:, :, 1:] - d4[:, :, :-1]) > BAND).all(-1)
        out[f"{name}.meta"] = np.array([seed, B, N, npoint, nsample, n_unknown], dtype=np.int64)
        out[f"{name}.radius"] = np.array([radius], dtype=np.float64)
        out[f"{name}.checksum"] = np.array([float(xyz.double().sum()), float((xyz.double() ** 2).sum())])
        out[f"{name}.fps"] = fps.numpy()
        out[f"{name}.ball"] = ball.numpy()
        out[f"{name}.ball_ok"] = ball_ok.numpy()
        out[f"{name}.nn_idx"] = nn_idx.numpy()
        out[f"{name}.nn_ok"] = nn_ok.numpy()
        print(f"{name}: seed {seed}, FPS {tuple(fps.shape)}, ball rows kept {float(ball_ok.float().mean()):.4f}, "
              f"3-NN rows kept {float(nn_ok.float().mean()):.4f}")
    path = os.path.join(HERE, "pure_ops.npz")
    np.savez_compressed(path, **out)
    print(f"pure_ops.npz: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
