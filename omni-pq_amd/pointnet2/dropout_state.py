"""Seed bookkeeping for the hash-based dropout of the hand-written kernels (attention probabilities,
LayerNorm branches, feed-forward activations).

The 64-bit seed lives in DEVICE memory -- a captured hipGraph reads the current value on every replay -- and
is advanced once per forward pass of the model (`advance`: one in-place add on the persistent counter plus one
copy that the forward and its backward keep); the host-side `salt` counter tells apart the dropout sites that
share one seed.  The first seed is drawn from torch's generator, so `torch.manual_seed`
governs the masks.
"""
import torch


class _DropoutState:
    """`state[device]`: the persistent 64-bit counter (static address: a captured hipGraph advances it on every replay).
    `cur[device]`: the seed of the forward pass in progress -- an IMMUTABLE copy taken by `advance`, which is what the
    kernels of that forward and of ITS backward read.  A later forward (the no-grad teacher of the mean-teacher step,
    train.py:489-491, or a second micro-batch under gradient accumulation) advances the counter and takes its own copy;
    it can no longer change the masks an earlier forward's backward pass rebuilds."""

    def __init__(self):
        self.state = {}
        self.cur = {}
        self.salts = {}
        self.slot = None            # `use(slot)`: a second, independent counter (see there)

    @property
    def salt(self):
        return self.salts.get(self.slot, 0)

    @salt.setter
    def salt(self, v):
        self.salts[self.slot] = v

    def use(self, slot):
        """with STATE.use("teacher"): ...  -- the forward passes issued inside draw from a counter of their own (device
        memory and host-side salt).  For a network whose forward runs on ANOTHER STREAM at the same time as the main
        network's (the mean-teacher step, train_step.CapturedStep): two streams adding to one device counter would race."""
        state = self

        class _Use:
            def __enter__(self):
                self.prev, state.slot = state.slot, slot

            def __exit__(self, *exc):
                state.slot = self.prev

        return _Use()

    def _state(self, device):
        t = self.state.get((self.slot, device))
        if t is None:
            t = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(device)
            self.state[(self.slot, device)] = t
        return t

    def seed(self, device):
        device = torch.device(device)
        t = self.cur.get((self.slot, device))
        if t is None:
            t = self.cur[(self.slot, device)] = self._state(device).clone()
        return t

    def advance(self, device):
        """New seed for the next training step; call once per forward of the model."""
        device = torch.device(device)
        st = self._state(device)
        st.add_(0x9E3779B97F4A7C15 >> 2)
        self.cur[(self.slot, device)] = st.clone()
        self.salt = 0

    def set_state(self, device, value):
        """(tests) restart the counter from a known value"""
        device = torch.device(device)
        self._state(device).fill_(int(value))
        self.cur.pop((self.slot, device), None)

    def reset(self):
        """(tests) forget every counter: the next use draws a new one from torch's generator"""
        self.state.clear()
        self.cur.clear()
        self.salts.clear()

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
