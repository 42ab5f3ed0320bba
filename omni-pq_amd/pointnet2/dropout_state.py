"""Seed bookkeeping for the hash-based dropout of the hand-written kernels (attention probabilities,
LayerNorm branches, feed-forward activations).

The 64-bit seed lives in DEVICE memory -- a captured hipGraph reads the current value on every replay -- and
is advanced once per training step (`advance`, one tiny kernel); the host-side `salt` counter tells apart the
dropout sites that share one seed.  The first seed is drawn from torch's generator, so `torch.manual_seed`
governs the masks.
"""
import torch


class _DropoutState:
    def __init__(self):
        self.seeds = {}
        self.salt = 0

    def seed(self, device):
        device = torch.device(device)
        t = self.seeds.get(device)
        if t is None:
            t = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(device)
            self.seeds[device] = t
        return t

    def advance(self, device):
        """New seed for the next training step; call once per forward of the model."""
        self.seed(device).add_(0x9E3779B97F4A7C15 >> 2)
        self.salt = 0

    def next_salt(self):
        self.salt += 1
        return self.salt


STATE = _DropoutState()


def seed(device):
    return STATE.seed(device)


def next_salt():
    return STATE.next_salt()


def advance(device):
    STATE.advance(device)
