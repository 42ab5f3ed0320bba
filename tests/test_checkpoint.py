"""Checkpoint round trip in the reference's format (train.py:153-207).

tests/golden/reference_state_spec.npz holds the names / shapes / dtypes of the REFERENCE model's checkpointed state_dict
(SyncBatchNorm-converted, `module.`-prefixed, as a released .pth has them) and its two AdamW parameter groups, produced by
instantiating the reference model (tests/golden/make_golden_state_spec.py).  A file of exactly that shape must load into
this repo's model, tensor for tensor, and a file written here must have exactly those entries."""
import argparse
import os

import numpy as np
import pytest
import torch

from conftest import REPO  # noqa: F401  (sys.path set-up)

SPEC = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_state_spec.npz"))


def spec(tag):
    names = [str(n) for n in SPEC[f"{tag}.names"]]
    shapes = [tuple(int(d) for d in str(s).split(",") if d) for s in SPEC[f"{tag}.shapes"]]
    dtypes = [getattr(torch, str(d).split(".")[1]) for d in SPEC[f"{tag}.dtypes"]]
    return names, shapes, dtypes


def released_style_checkpoint(tag, epoch):
    gen = torch.Generator().manual_seed(7)
    model = {}
    names, shapes, dtypes = spec(tag)
    alias_of = SPEC[f"{tag}.alias_of"]             # modules the reference registers under two names share their tensors
    for i, (n, s, d) in enumerate(zip(names, shapes, dtypes)):
        if alias_of[i] != i:
            model[n] = model[names[alias_of[i]]]
        else:
            model[n] = (torch.randn(s, generator=gen) if d.is_floating_point else torch.full(s, 3, dtype=d))
    return {"config": argparse.Namespace(max_epoch=600), "save_path": "", "model": model, "optimizer": None,
            "scheduler": None, "epoch": epoch}


def build(extra, oracle_backend):
    import bench
    return bench.build_model(extra)


@pytest.mark.parametrize("tag,extra", [("c0", 0), ("c6", 6)])
def test_released_style_checkpoint_loads_tensor_for_tensor(tag, extra, tmp_path, oracle_backend):
    import checkpoint
    net = build(extra, oracle_backend)
    ck = released_style_checkpoint(tag, "last")
    path = str(tmp_path / "ckpt.pth")
    torch.save(ck, path)
    args = argparse.Namespace(checkpoint_path=path, ema=True)
    ema = build(extra, oracle_backend)
    assert checkpoint.load_checkpoint(args, net, None, None, ema_model=ema) == 600 and args.start_epoch == 601
    for target in (net, ema):                      # no 'ema_model' in the file: the EMA model starts from 'model' (train.py:172-174)
        sd = target.state_dict()
        assert len(sd) == len(ck["model"])
        for k, v in ck["model"].items():
            assert torch.equal(sd[k[len("module."):]], v), k
    ck["model"].pop(next(k for k in ck["model"] if k.endswith("conv.weight")))
    torch.save(ck, path)
    with pytest.raises(RuntimeError, match="does not match"):
        checkpoint.load_checkpoint(args, net, None, None)


def test_checkpoint_written_here_has_the_reference_entries_and_round_trips(tmp_path, oracle_backend):
    import checkpoint
    net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(build(0, oracle_backend))
    names, shapes, dtypes = spec("c0")
    sd = net.state_dict()
    assert ["module." + k for k in sd] == names                                    # same entries, same ORDER
    assert [tuple(v.shape) for v in sd.values()] == shapes and [v.dtype for v in sd.values()] == dtypes
    first = {}
    alias = [first.setdefault((v.data_ptr(), tuple(v.shape)), i) if v.numel() else i for i, v in enumerate(sd.values())]
    assert alias == list(SPEC["c0.alias_of"])                                      # the same modules are tied
    plain = [n for n, p in net.named_parameters() if "decoder" not in n]
    deco = [n for n, p in net.named_parameters() if "decoder" in n]
    assert plain == [str(n) for n in SPEC["c0.group_plain"]] and deco == [str(n) for n in SPEC["c0.group_decoder"]]

    class Wrapper(torch.nn.Module):                # the `module.` prefix DistributedDataParallel gives the saved names
        def __init__(self, module):
            super().__init__()
            self.module = module

    named = dict(net.named_parameters())
    optimizer = torch.optim.AdamW([{"params": [named[n] for n in plain]},
                                   {"params": [named[n] for n in deco], "lr": 2e-4}], lr=4e-3, weight_decay=5e-4)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=100, eta_min=1e-6)
    for p in net.parameters():
        p.grad = torch.ones_like(p) * 1e-3
    optimizer.step()
    scheduler.step()
    args = argparse.Namespace(log_dir=str(tmp_path), save_freq=10, ema=True)
    ema = build(0, oracle_backend)
    assert checkpoint.save_checkpoint(args, 7, Wrapper(net), optimizer, scheduler) is None          # 7 % 10 != 0
    path = checkpoint.save_checkpoint(args, 20, Wrapper(net), optimizer, scheduler, ema_model=ema)
    assert path.endswith("ckpt_epoch_20.pth")
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert list(ck.keys()) == ["config", "save_path", "model", "optimizer", "scheduler", "epoch", "ema_model"]
    assert list(ck["model"].keys()) == names

    net2 = torch.nn.SyncBatchNorm.convert_sync_batchnorm(build(0, oracle_backend))
    named2 = dict(net2.named_parameters())
    opt2 = torch.optim.AdamW([{"params": [named2[n] for n in plain]}, {"params": [named2[n] for n in deco], "lr": 2e-4}],
                             lr=4e-3, weight_decay=5e-4)
    sch2 = torch.optim.lr_scheduler.CosineAnnealingLR(opt2, T_max=100, eta_min=1e-6)
    ema2 = build(0, oracle_backend)
    args2 = argparse.Namespace(checkpoint_path=path, ema=True)
    assert checkpoint.load_checkpoint(args2, net2, opt2, sch2, ema_model=ema2) == 20 and args2.start_epoch == 21
    for (k, a), b in zip(net.state_dict().items(), net2.state_dict().values()):
        assert torch.equal(a, b), k
    for a, b in zip(ema.state_dict().values(), ema2.state_dict().values()):
        assert torch.equal(a, b)
    s1, s2 = optimizer.state_dict(), opt2.state_dict()
    assert s1["param_groups"] == s2["param_groups"]
    for i in s1["state"]:
        assert torch.equal(s1["state"][i]["exp_avg"], s2["state"][i]["exp_avg"])
    assert scheduler.state_dict() == sch2.state_dict()
