"""ctypes view of libomnipq_pointops.so for tests: raw device pointers + stream, nothing else."""
import ctypes
import os
import re

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(REPO, "omni-pq_amd", "lib", "libomnipq_pointops.so")
INCLUDE = os.path.join(REPO, "include")


def declared_symbols():
    """Every function the public headers (include/*.h) declare."""
    names = set()
    for fn in sorted(os.listdir(INCLUDE)):
        if not fn.endswith(".h"):
            continue
        text = open(os.path.join(INCLUDE, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(omnipq_\w+)\s*\(", text))
    return sorted(names)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.omnipq_error_string.restype = ctypes.c_char_p
    return _lib


def P(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def plan_aware():
    """Entry points of include/omnipq_sa.h that take `const omnipq_row_plan *plan` (the argument before the stream)."""
    text = re.sub(r"/\*.*?\*/", " ", open(os.path.join(INCLUDE, "omnipq_sa.h")).read(), flags=re.S)
    return frozenset(m.group(1) for m in re.finditer(r"\b(omnipq_\w+)\s*\(([^;{}()]*)\)\s*;", text)
                     if "omnipq_row_plan *plan" in m.group(2))


PLAN_AWARE = plan_aware()


def call(name, *args, plan=None):
    """plan: a ctypes pointer to an omnipq_row_plan for the plan-aware entry points (None = every row)"""
    if name in PLAN_AWARE:
        args = args + (plan,)
    rc = getattr(lib(), name)(*args, stream())
    return rc


def ok(name, *args, plan=None):
    rc = call(name, *args, plan=plan)
    assert rc == 0, f"{name} -> {rc}: {lib().omnipq_error_string(rc).decode()}"


def fps(xyz, m, flags=None):
    b, n, _ = xyz.shape
    out = torch.full((b, m), -7, device=xyz.device, dtype=torch.int32)
    tmp = torch.full((b, n), 1e10, device=xyz.device, dtype=torch.float32)
    if flags is None:
        ok("omnipq_furthest_point_sampling", b, n, m, P(xyz), P(tmp), P(out))
    else:
        ok("omnipq_furthest_point_sampling_ex", b, n, m, P(xyz), P(tmp), P(out), ctypes.c_uint(flags))
    rc = lib().omnipq_fps_check(stream())
    assert rc == 0, lib().omnipq_error_string(rc).decode()
    return out, tmp


def ball_query(new_xyz, xyz, radius, nsample):
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = torch.full((b, m, nsample), -7, device=xyz.device, dtype=torch.int32)   # NOT zero-filled
    ok("omnipq_ball_query", b, n, m, ctypes.c_float(radius), nsample, P(new_xyz), P(xyz), P(idx))
    return idx


def ball_query_grid(new_xyz, xyz, radius, nsample):
    """the hash-grid variant of omnipq_ball_query: must give the same indices"""
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = torch.full((b, m, nsample), -7, device=xyz.device, dtype=torch.int32)   # NOT zero-filled
    lib().omnipq_ball_query_grid_workspace_bytes.restype = ctypes.c_longlong
    ws = torch.empty(max(int(lib().omnipq_ball_query_grid_workspace_bytes(b, n)), 16), device=xyz.device, dtype=torch.uint8)
    ok("omnipq_ball_query_grid", b, n, m, ctypes.c_float(radius), nsample, P(new_xyz), P(xyz), P(idx), P(ws))
    return idx


def group_points(points, idx):
    b, c, n = points.shape
    _, m, s = idx.shape
    out = torch.empty((b, c, m, s), device=points.device)
    ok("omnipq_group_points", b, c, n, m, s, P(points), P(idx), P(out))
    return out


def group_points_grad(grad_out, idx, n):
    b, c, m, s = grad_out.shape
    out = torch.zeros((b, c, n), device=grad_out.device)
    ok("omnipq_group_points_grad", b, c, n, m, s, P(grad_out), P(idx), P(out))
    return out


def gather_points(points, idx):
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.empty((b, c, m), device=points.device)
    ok("omnipq_gather_points", b, c, n, m, P(points), P(idx), P(out))
    return out


def gather_points_grad(grad_out, idx, n):
    b, c, m = grad_out.shape
    out = torch.zeros((b, c, n), device=grad_out.device)
    ok("omnipq_gather_points_grad", b, c, n, m, P(grad_out), P(idx), P(out))
    return out


def three_nn(unknown, known):
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = torch.empty((b, n, 3), device=unknown.device)
    idx = torch.empty((b, n, 3), device=unknown.device, dtype=torch.int32)
    ok("omnipq_three_nn", b, n, m, P(unknown), P(known), P(d2), P(idx))
    return d2, idx


def three_interpolate(points, idx, weight):
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.empty((b, c, n), device=points.device)
    ok("omnipq_three_interpolate", b, c, m, n, P(points), P(idx), P(weight), P(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    b, c, n = grad_out.shape
    out = torch.zeros((b, c, m), device=grad_out.device)
    ok("omnipq_three_interpolate_grad", b, c, n, m, P(grad_out), P(idx), P(weight), P(out))
    return out
