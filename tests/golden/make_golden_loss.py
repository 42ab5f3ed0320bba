"""Generate tests/golden/loss_nn_distance.npz from the REFERENCE's own utils/nn_distance.py (imported in place from
/root/reference; nothing is copied, no bytecode written).  Runs only in the build container.

    python tests/golden/make_golden_loss.py

Cases: the reference's demo inputs (utils/nn_distance.py:63-69: np.random.seed(0), 5 x 3 and 6 x 3 points), and seeded
clouds at the shapes loss_helper_pq.py calls it with (votes (B*1024, 3, 3) against 3 ground-truth votes, :39; 256
aggregated votes against 64 box centres, :61 / :208), each in the three distance modes; plus huber_loss samples.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("OMNIPQ_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "utils"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import nn_distance as ref  # noqa: E402  (the reference's)


def main():
    out = {}
    np.random.seed(0)                                    # the demo's own inputs
    demo1 = np.random.random((1, 5, 3)).astype(np.float32)
    demo2 = np.random.random((1, 6, 3)).astype(np.float32)
    gen = torch.Generator().manual_seed(20260928)
    cases = {"demo": (demo1, demo2),
             "votes": (torch.randn(64, 3, 3, generator=gen).numpy(), torch.randn(64, 3, 3, generator=gen).numpy()),
             "centres": (torch.randn(8, 256, 3, generator=gen).numpy() * 2, torch.randn(8, 64, 3, generator=gen).numpy() * 2),
             "wide": (torch.randn(2, 33, 5, generator=gen).numpy(), torch.randn(2, 70, 5, generator=gen).numpy())}
    for name, (a, b) in cases.items():
        out[f"{name}.pc1"], out[f"{name}.pc2"] = a, b
        for mode, kw in (("l2", {}), ("huber", {"l1smooth": True, "delta": 0.7}), ("l1", {"l1": True})):
            d1, i1, d2, i2 = ref.nn_distance(torch.from_numpy(a), torch.from_numpy(b), **kw)
            out[f"{name}.{mode}.dist1"], out[f"{name}.{mode}.idx1"] = d1.numpy(), i1.numpy()
            out[f"{name}.{mode}.dist2"], out[f"{name}.{mode}.idx2"] = d2.numpy(), i2.numpy()
    err = torch.linspace(-3, 3, 61)
    out["huber.error"] = err.numpy()
    for delta in (1.0, 0.25):
        out[f"huber.delta{delta}"] = ref.huber_loss(err, delta).numpy()
    np.savez_compressed(os.path.join(HERE, "loss_nn_distance.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
