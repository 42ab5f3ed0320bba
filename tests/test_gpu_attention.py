"""GPU: the hand-written attention kernels (csrc/attention.hip) against a plain PyTorch f32 restatement of
the reference's attention core (models/utils/multi_head_attention.py:375-391):
    w = softmax((q * d^-0.5) k^T);  w = dropout(w, p);  out = w v
Tolerances: operands are bf16 and the probabilities are rounded to bf16 before the second contraction, so
outputs agree with the f32 evaluation of the SAME bf16 inputs to ~2^-8 relative (checked as relative L2
<= 1e-2 and max-abs <= 3e-2 of the tensor's range); gradients likewise (relative L2 <= 2e-2).
"""
import pytest
import torch

from conftest import REPO  # noqa: F401  (sys.path set-up)

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def rel_l2(a, b):
    a, b = a.detach(), b.detach()
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def ref_attention(q, k, v, H, mask=None, p=0.0):
    """f32 restatement on (L,N,E)/(S,N,E) tensors; mask (N*H, L, S) keep mask or None."""
    L, N, E = q.shape
    S = k.shape[0]
    D = E // H
    qh = q.float().reshape(L, N * H, D).transpose(0, 1) * (D ** -0.5)
    kh = k.float().reshape(S, N * H, D).transpose(0, 1)
    vh = v.float().reshape(S, N * H, D).transpose(0, 1)
    w = torch.softmax(torch.bmm(qh, kh.transpose(1, 2)), dim=-1)
    if mask is not None:
        w = w * mask.float() / (1.0 - p)
    return torch.bmm(w, vh).transpose(0, 1).reshape(L, N, E)


def make_qkv(L, S, N, H, D, seed, packed=False):
    gen = torch.Generator().manual_seed(seed)
    E = H * D
    if packed:         # q, k, v as chunks of one projection output, like the self-attention call site
        big = (1.5 * torch.randn((L, N, 3 * E), generator=gen)).to(torch.bfloat16).to(dev()).requires_grad_(True)
        q, k, v = big.chunk(3, dim=-1)
        return big, q, k, v
    q = (1.5 * torch.randn((L, N, E), generator=gen)).to(torch.bfloat16).to(dev()).requires_grad_(True)
    k = (1.5 * torch.randn((S, N, E), generator=gen)).to(torch.bfloat16).to(dev()).requires_grad_(True)
    v = torch.randn((S, N, E), generator=gen).to(torch.bfloat16).to(dev()).requires_grad_(True)
    return None, q, k, v


CASES = [  # L, S, N, H, D
    (256, 256, 8, 8, 36),        # decoder self attention
    (256, 1024, 8, 8, 36),       # decoder cross attention
    (32, 32, 1, 1, 4),
    (70, 45, 2, 3, 36),          # ragged: partial query and key blocks, fewer key blocks than waves
    (33, 130, 1, 2, 48),
    (1, 1, 1, 1, 8),
]


@pytest.mark.parametrize("L,S,N,H,D", CASES)
def test_attention_forward_backward_matches_f32(L, S, N, H, D):
    from utils import fused_attention
    packed = L == S
    big, q, k, v = make_qkv(L, S, N, H, D, 11 + L + S, packed)
    assert fused_attention.usable(q, k, v, H)
    out = fused_attention.attention(q, k, v, H, 0.0)
    want = ref_attention(q, k, v, H)
    assert out.shape == (L, N, H * D) and out.dtype == torch.bfloat16
    assert rel_l2(out, want) < 1e-2
    assert float((out.float() - want).abs().max()) <= 3e-2 * float(want.abs().max())
    gen = torch.Generator().manual_seed(5)
    g = torch.randn((L, N, H * D), generator=gen).to(torch.bfloat16).to(dev())
    leaves = [big] if packed else [q, k, v]
    got = torch.autograd.grad(out, leaves, g)
    ref = torch.autograd.grad(want, leaves, g.float())
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        assert rel_l2(a, b) < 2e-2, rel_l2(a, b)


@pytest.mark.parametrize("L,S,N,H,D", [(256, 1024, 8, 8, 36), (256, 256, 8, 8, 36), (70, 45, 2, 4, 36)])
def test_xcd_aware_workgroup_placement_is_a_relabelling_of_the_grid(L, S, N, H, D):
    """csrc/attention.hip att_block: the workgroups of one (batch, head) are placed on one XCD (a bijection of the grid when
    batch * heads is a multiple of 8, the plain reading otherwise).  Which workgroup computes a tile must not show in a
    single bit of the output or of the three gradients."""
    from utils import fused_attention
    from sa_fused import _lib
    _, q, k, v = make_qkv(L, S, N, H, D, 3)
    g = torch.randn((L, N, H * D), generator=torch.Generator().manual_seed(9)).to(torch.bfloat16).to(dev())
    got = {}
    try:
        for mode in (0, 1):
            _lib.omnipq_attn_block_map(mode)
            out = fused_attention.attention(q, k, v, H, 0.0)
            got[mode] = (out.detach().clone(),) + tuple(t.clone() for t in torch.autograd.grad(out, [q, k, v], g))
    finally:
        _lib.omnipq_attn_block_map(1)
    for a, b in zip(got[0], got[1]):
        assert torch.equal(a, b)


def test_attention_dropout_mask_is_consistent_forward_and_backward():
    """With dropout the kernels must use ONE mask in forward and backward: recover it through the test
    entry point and compare against the f32 restatement that applies the same mask."""
    from utils import fused_attention
    L, S, N, H, D, p = 96, 200, 2, 4, 36, 0.25
    _, q, k, v = make_qkv(L, S, N, H, D, 3)
    torch.manual_seed(7)
    fused_attention.STATE.reset()
    fused_attention.STATE.advance(dev())
    out = fused_attention.attention(q, k, v, H, p)
    salt = fused_attention.STATE.salt
    seed = fused_attention.STATE.seed(dev())
    mask = fused_attention.dropout_mask(N, H, L, S, p, seed, salt)
    keep = float(mask.float().mean())
    assert abs(keep - (1 - p)) < 0.01, keep
    # rows and columns are not correlated: every query keeps about the same share
    per_row = mask.float().mean(dim=2)
    assert float(per_row.min()) > 0.55 and float(per_row.max()) < 0.92
    want = ref_attention(q, k, v, H, mask, p)
    assert rel_l2(out, want) < 1e-2
    g = torch.randn((L, N, H * D), generator=torch.Generator().manual_seed(9)).to(torch.bfloat16).to(dev())
    got = torch.autograd.grad(out, [q, k, v], g)
    ref = torch.autograd.grad(want, [q, k, v], g.float())
    for a, b in zip(got, ref):
        assert rel_l2(a, b) < 2e-2, rel_l2(a, b)
    # a second call draws another mask (new salt), a new step another one again (new seed)
    out2 = fused_attention.attention(q, k, v, H, p)
    assert not torch.equal(out, out2)
    m2 = fused_attention.dropout_mask(N, H, L, S, p, seed, fused_attention.STATE.salt)
    assert 0.3 < float((m2 == mask).float().mean()) < 0.8
    fused_attention.STATE.advance(dev())
    m3 = fused_attention.dropout_mask(N, H, L, S, p, fused_attention.STATE.seed(dev()), salt)
    assert 0.3 < float((m3 == mask).float().mean()) < 0.8


def test_backward_rebuilds_the_masks_of_its_own_forward():
    """Mean-teacher order (reference train.py:489-491): student forward, then a train-mode no-grad teacher forward that
    advances the dropout seed, then the student's backward.  The backward kernels rebuild the masks from the seed
    tensor their forward saw -- an immutable per-forward copy -- so the gradients equal those of a run without the
    teacher in between (they did not while one device tensor was advanced in place)."""
    from utils import fused_attention
    L, S, N, H, D, p = 64, 128, 2, 4, 36, 0.3
    _, q, k, v = make_qkv(L, S, N, H, D, 5)
    g = torch.randn((L, N, H * D), generator=torch.Generator().manual_seed(11)).to(torch.bfloat16).to(dev())

    def student(with_teacher):
        fused_attention.STATE.set_state(dev(), 123456789)
        fused_attention.STATE.advance(dev())
        out = fused_attention.attention(q, k, v, H, p)
        if with_teacher:
            fused_attention.STATE.advance(dev())
            with torch.no_grad():
                fused_attention.attention(q, k, v, H, p)
        return out, torch.autograd.grad(out, [q, k, v], g)

    out_a, grads_a = student(False)
    out_b, grads_b = student(True)
    assert torch.equal(out_a, out_b)
    for a, b in zip(grads_a, grads_b):
        assert rel_l2(a, b) < 1e-6, rel_l2(a, b)


def test_multihead_attention_module_uses_the_kernels_under_autocast():
    """The module-level switch: under bf16 autocast MultiheadAttention.forward goes through the kernels and
    agrees with its own f32 math path (eval mode, no dropout)."""
    from utils import fused_attention, multi_head_attention
    torch.manual_seed(0)
    mha = multi_head_attention.MultiheadAttention(288, 8, dropout=0.1).to(dev()).eval()
    x = torch.randn(256, 4, 288, device=dev())
    mem = torch.randn(1024, 4, 288, device=dev())
    calls = []
    orig = fused_attention.attention
    fused_attention.attention = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            got = mha(x, mem, mem, need_weights=False)[0]
    finally:
        fused_attention.attention = orig
    assert calls
    with torch.no_grad():
        want = mha(x, mem, mem, need_weights=True)[0]
    assert rel_l2(got, want) < 2e-2
