#!/usr/bin/env python
"""Run-to-run reproducibility of one training step: the same batch, the same weights, the same dropout counters, twice through
forward + backward (deferred grouped weight gradients, as the benchmark runs them) -- which end_points and which parameter
gradients differ bit for bit, and by how much.  python tools/repro_check.py [--points 20000] [--batch 2] [--runs 3]"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("omni-pq_amd", "omni-pq_amd/pointnet2", "omni-pq_amd/models"):
    sys.path.insert(0, os.path.join(REPO, p))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--dropout", action="store_true", help="keep dropout on (the counters advance between runs: expect differences)")
    args = ap.parse_args()
    import sa_fused
    import synth
    from procedural import load_procedural
    from test_oracle_golden import zero_dropout
    dev = torch.device("cuda", 0)
    net = load_procedural(bench.build_model(0)).to(dev).train()
    if not args.dropout:
        zero_dropout(net)
    pc = synth.make_clouds(60, args.batch, args.points, kind="room").to(dev)
    state = {k: v.detach().clone() for k, v in net.state_dict().items()}
    runs = []
    for r in range(args.runs):
        net.load_state_dict(state)
        for p in net.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ep = net({"point_clouds": pc})
            loss = bench.loss_of(ep)
        with sa_fused.deferred_wgrads():
            loss.backward()
        torch.cuda.synchronize()
        runs.append(({k: v.detach().clone() for k, v in ep.items() if torch.is_tensor(v)},
                     {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}, float(loss)))
    ref = runs[0]

    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))

    for r, (ep, gr, loss) in enumerate(runs[1:], 1):
        bad_ep = [(k, rel(ep[k].float(), ref[0][k].float())) for k in ref[0] if not torch.equal(ep[k], ref[0][k])]
        bad_gr = [(k, rel(gr[k], ref[1][k])) for k in ref[1] if not torch.equal(gr[k], ref[1][k])]
        print(f"run {r} vs run 0: loss {loss!r} vs {ref[2]!r}; end_points that differ: {len(bad_ep)} of {len(ref[0])}; "
              f"gradients that differ: {len(bad_gr)} of {len(ref[1])}")
        for k, e in sorted(bad_ep, key=lambda kv: -kv[1])[:8]:
            print(f"    ep   {k:48s} rel-L2 {e:.2e}")
        for k, e in sorted(bad_gr, key=lambda kv: -kv[1])[:12]:
            print(f"    grad {k:48s} rel-L2 {e:.2e}")
        # which subsystem: first layer (in forward order) whose gradient differs is the LAST one backward reached unharmed
        groups = {}
        for k, e in bad_gr:
            groups.setdefault(k.split(".")[0] + "." + k.split(".")[1] if k.startswith(("backbone", "decoder")) else k.split(".")[0], []).append(e)
        print("    by module:", {g: (len(v), f"{max(v):.1e}") for g, v in sorted(groups.items())})


if __name__ == "__main__":
    main()
