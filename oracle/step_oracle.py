"""CPU restatement of the pieces of the reference's training step that sit right next to the hot path
(SURVEY.md 8f-1).  TEST INFRASTRUCTURE ONLY: tests/ (and nothing in the product) imports this.

update_ema_variables follows train.py:435-439 of the reference statement by statement:

    alpha = min(1 - 1 / (global_step + 1), alpha)
    for ema_param, param in zip(ema_model.parameters(), model.parameters()):
        ema_param.data.mul_(alpha).add_(1 - alpha, param.data)

in numpy with the roundings the reference's two in-place CUDA ops perform: the scalar factors are Python doubles
rounded to f32 by the elementwise kernels, `mul_` rounds its product to f32, and `add_(alpha, other)` is
`a + alpha * b` in one kernel, which nvcc contracts to a fused multiply-add by default (evaluated here in float64
and rounded once; the product of two f32 is exact in f64).  Parity: pinned by running the reference's own function
(imported from /root/reference/train.py is not possible -- it parses the command line and imports the dataset
stack at import time -- so the three statements above are exercised through torch on CPU in
tests/test_oracle_golden.py::test_ema_oracle_matches_the_reference_statements).
"""
import numpy as np


def ema_alpha(alpha, global_step):
    """Use the true average until the exponential average is more correct (train.py:436-437)."""
    return min(1.0 - 1.0 / (global_step + 1), alpha)


def update_ema_variables(ema_params, params, alpha, global_step):
    """In place on the list of float32 numpy arrays `ema_params`."""
    a = ema_alpha(alpha, global_step)
    a32, b32 = np.float32(a), np.float32(1.0 - a)
    for e, p in zip(ema_params, params):
        assert e.dtype == np.float32 and p.dtype == np.float32 and e.shape == p.shape
        np.multiply(e, a32, out=e)                    # ema_param.data.mul_(alpha)
        e[...] = (e.astype(np.float64) + np.float64(b32) * p.astype(np.float64)).astype(np.float32)   # fma
    return a
