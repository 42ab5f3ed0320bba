"""CPU ORACLE binding -- test infrastructure, NOT product code.

Exposes the C restatement in ``pointops_oracle.c`` with the call signatures of
the reference's pybind module ``pointnet2._ext`` (pointnet2/_ext_src/src/
bindings.cpp:11-24) so that it can be plugged in wherever that module is
expected:

* under the *reference's own* Python layers when golden vectors are generated
  (tests/golden/make_golden.py);
* under this repo's Python layers in the CPU test-suite and in bench.py's
  ``cpu_baseline`` leg.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's cpu_baseline leg may
import this module; the product path (omni-pq_amd/) never does.

Allocation conventions follow the reference's C++ wrappers: outputs come from
``torch.zeros`` (sampling.cpp:33-35, ball_query.cpp:27-29, ...) and the FPS
scratch from ``torch.full(1e10)`` (sampling.cpp:80-82).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpointops_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "pointops_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_opt_n_threads.restype = ctypes.c_int
        _lib.oracle_get_dist_form.restype = ctypes.c_int
    return _lib


def set_dist_form(form):
    """0 = uncontracted, 1 = fma(c,c,fma(a,a,b*b)) (default), 2 = fma(c,c,fma(b,b,a*a))."""
    lib().oracle_set_dist_form(ctypes.c_int(int(form)))


def set_num_threads(n):
    """Cap the OpenMP team of the oracle (batch / query loops)."""
    lib().oracle_set_num_threads(ctypes.c_int(int(n)))


def opt_n_threads(work_size):
    return int(lib().oracle_opt_n_threads(ctypes.c_int(int(work_size))))


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _chk(t, name, dtype):
    if t.device.type != "cpu":
        raise RuntimeError(f"oracle: {name} must be a CPU tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != dtype:
        kind = "float" if dtype == torch.float32 else "int"
        raise RuntimeError(f"{name} must be a{'n' if kind == 'int' else ''} {kind} tensor")


def gather_points(points, idx):
    _chk(points, "points", torch.float32)
    _chk(idx, "idx", torch.int32)
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.zeros((b, c, m), dtype=torch.float32)
    lib().oracle_gather_points(b, c, n, m, _p(points), _p(idx), _p(out))
    return out


def gather_points_grad(grad_out, idx, n):
    _chk(grad_out, "grad_out", torch.float32)
    _chk(idx, "idx", torch.int32)
    b, c, m = grad_out.shape
    out = torch.zeros((b, c, n), dtype=torch.float32)
    lib().oracle_gather_points_grad(b, c, int(n), m, _p(grad_out), _p(idx), _p(out))
    return out


def furthest_point_sampling(points, nsamples):
    _chk(points, "points", torch.float32)
    b, n, _ = points.shape
    out = torch.zeros((b, nsamples), dtype=torch.int32)
    tmp = torch.full((b, n), 1e10, dtype=torch.float32)
    lib().oracle_furthest_point_sampling(b, n, int(nsamples), _p(points), _p(tmp), _p(out))
    return out


def three_nn(unknowns, knows):
    _chk(unknowns, "unknowns", torch.float32)
    _chk(knows, "knows", torch.float32)
    b, n, _ = unknowns.shape
    m = knows.shape[1]
    idx = torch.zeros((b, n, 3), dtype=torch.int32)
    dist2 = torch.zeros((b, n, 3), dtype=torch.float32)
    lib().oracle_three_nn(b, n, m, _p(unknowns), _p(knows), _p(dist2), _p(idx))
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    _chk(points, "points", torch.float32)
    _chk(idx, "idx", torch.int32)
    _chk(weight, "weight", torch.float32)
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.zeros((b, c, n), dtype=torch.float32)
    lib().oracle_three_interpolate(b, c, m, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    _chk(grad_out, "grad_out", torch.float32)
    _chk(idx, "idx", torch.int32)
    _chk(weight, "weight", torch.float32)
    b, c, n = grad_out.shape
    out = torch.zeros((b, c, int(m)), dtype=torch.float32)
    lib().oracle_three_interpolate_grad(b, c, n, int(m), _p(grad_out), _p(idx), _p(weight), _p(out))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    _chk(new_xyz, "new_xyz", torch.float32)
    _chk(xyz, "xyz", torch.float32)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = torch.zeros((b, m, int(nsample)), dtype=torch.int32)
    lib().oracle_ball_query(b, n, m, ctypes.c_float(radius), int(nsample),
                            _p(new_xyz), _p(xyz), _p(idx))
    return idx


def group_points(points, idx):
    _chk(points, "points", torch.float32)
    _chk(idx, "idx", torch.int32)
    b, c, n = points.shape
    _, npoints, nsample = idx.shape
    out = torch.zeros((b, c, npoints, nsample), dtype=torch.float32)
    lib().oracle_group_points(b, c, n, npoints, nsample, _p(points), _p(idx), _p(out))
    return out


def group_points_grad(grad_out, idx, n):
    _chk(grad_out, "grad_out", torch.float32)
    _chk(idx, "idx", torch.int32)
    b, c, npoints, nsample = grad_out.shape
    out = torch.zeros((b, c, int(n)), dtype=torch.float32)
    lib().oracle_group_points_grad(b, c, int(n), npoints, nsample, _p(grad_out), _p(idx), _p(out))
    return out
