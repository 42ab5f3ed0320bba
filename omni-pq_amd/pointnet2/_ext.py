"""`pointnet2._ext` for MI355X: the nine native point-set operators, on gfx950 HIP kernels.

Drop-in for the reference's pybind/CUDA module of the same name (pointnet2/_ext_src/src/
bindings.cpp:11-24): same function names, argument order, return values, dtype/contiguity
checks and the same refusal of CPU tensors ("CPU not supported", e.g. ball_query.cpp:35-37).
The work is done by `libomnipq_pointops.so` (hand-written HIP, see ../csrc and
include/omnipq_pointops.h), reached through its C ABI with raw device pointers and the
calling thread's current HIP stream -- this file is the thin binding a maintainer of the
reference would write against that ABI.

There is no fallback: if the shared library is missing or a tensor is not on a GPU the call
raises.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get(
    "OMNIPQ_POINTOPS_LIB",
    os.path.join(os.path.dirname(_HERE), "lib", "libomnipq_pointops.so"))

if not os.path.exists(_LIB_PATH):
    raise ImportError(
        f"pointnet2._ext: {_LIB_PATH} not found -- build it with `python omni-pq_amd/build.py` "
        "(hipcc --offload-arch=gfx950).  There is no CPU or PyTorch fallback for these operators.")

ABI_VERSION = 3          # include/omnipq_pointops.h: OMNIPQ_ABI_VERSION


def _load(path):
    lib = ctypes.CDLL(path)
    lib.omnipq_error_string.restype = ctypes.c_char_p
    lib.omnipq_abi_version.restype = ctypes.c_int
    if lib.omnipq_abi_version() != ABI_VERSION:
        raise ImportError(f"pointnet2._ext: {path} reports ABI version {lib.omnipq_abi_version()}, this binding is written "
                          f"against {ABI_VERSION} (include/omnipq_pointops.h) -- rebuild with `python omni-pq_amd/build.py`")
    lib.omnipq_plan_aware_entry_points.restype = ctypes.c_char_p
    return lib


# The library is built once per 16-bit element type (omni-pq_amd/build.py, csrc/common.h: e16_t): bfloat16 in
# libomnipq_pointops.so -- which also serves every index / f32 operator below -- and IEEE half in its `_f16` twin.
_LIB_F16_PATH = _LIB_PATH[:-3] + "_f16.so"
_LIBS = {torch.bfloat16: _load(_LIB_PATH)}
if os.path.exists(_LIB_F16_PATH):
    _LIBS[torch.float16] = _load(_LIB_F16_PATH)


class _Elem16(threading.local):
    """The element type the hand-written 16-bit kernels are running in: `E16.dtype`.  It follows torch.autocast:
    every entry into the hand-written path asks `E16.autocast()` -- True when CUDA autocast is on with a dtype a library
    exists for, which also makes that dtype current -- and the autograd nodes re-select the type their saved tensors
    have before they launch anything in backward.  Two element types interleaved on ONE thread between a forward and its
    backward are therefore fine, and the state is per thread (`threading.local`: every thread starts at bfloat16 and
    the autograd engine's worker threads re-select per node), so a teacher in fp16 next to a student in bf16 on two
    threads do not see each other's choice."""

    def __init__(self):
        self.dtype = torch.bfloat16

    def available(self, dtype):
        return dtype in _LIBS

    def select(self, dtype):
        if dtype is not self.dtype:
            if dtype not in _LIBS:
                raise RuntimeError(f"pointnet2._ext: no library for element type {dtype} ({_LIB_F16_PATH} missing?)")
            self.dtype = dtype
        return dtype

    def autocast(self):
        if not torch.is_autocast_enabled("cuda"):
            return False
        dt = torch.get_autocast_dtype("cuda")
        if dt not in _LIBS:
            return False
        self.dtype = dt
        return True

    def is16(self, dtype):
        """Is `dtype` the current element type?"""
        return dtype is self.dtype


E16 = _Elem16()


class _Entry:
    """One C-ABI entry point, resolved at call time in the library of the current element type."""
    __slots__ = ("__name__", "_fns")

    def __init__(self, name):
        self.__name__ = name
        self._fns = {dt: getattr(lib, name) for dt, lib in _LIBS.items()}

    def __call__(self, *args):
        return self._fns[E16.dtype](*args)

    @property
    def restype(self):
        return self._fns[torch.bfloat16].restype

    @restype.setter
    def restype(self, value):
        for fn in self._fns.values():
            fn.restype = value

    @property
    def argtypes(self):
        return self._fns[torch.bfloat16].argtypes

    @argtypes.setter
    def argtypes(self, value):
        for fn in self._fns.values():
            fn.argtypes = value


class _Libs:
    def __getattr__(self, name):
        entry = _Entry(name)
        object.__setattr__(self, name, entry)
        return entry


_lib = _Libs()
# which entry points take a row plan in front of the stream: the loaded libraries say so themselves (build-time list)
PLAN_AWARE = frozenset(_LIBS[torch.bfloat16].omnipq_plan_aware_entry_points().decode().split())
for _l in _LIBS.values():
    if frozenset(_l.omnipq_plan_aware_entry_points().decode().split()) != PLAN_AWARE:
        raise ImportError("pointnet2._ext: the bf16 and f16 libraries were built from different headers")
_lib0 = _LIBS[torch.bfloat16]        # element-type independent entry points (index ops, FPS state) always live here

LIB_PATH = _LIB_PATH
LIB_F16_PATH = _LIB_F16_PATH if torch.float16 in _LIBS else None


def _check(x, name, dtype=None, cuda_like=None):
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if dtype is torch.float32 and x.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float tensor")
    if dtype is torch.int32 and x.dtype != torch.int32:
        raise RuntimeError(f"{name} must be an int tensor")
    if cuda_like is not None and cuda_like.is_cuda and not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")


def _need_gpu(x):
    if not x.is_cuda:
        raise RuntimeError("CPU not supported")


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device_index=None):
    """The calling thread's current HIP stream as a raw handle (fast path: no Stream object)."""
    if _raw_stream is not None:
        if device_index is None:
            device_index = torch.cuda.current_device()
        return ctypes.c_void_p(_raw_stream(device_index))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional per-call device timing (bench.py's roofline leg): when a list is installed here every
# C-ABI call is bracketed by two events recorded on the stream the kernel is launched on.
_timing_sink = None
timing_tag = ""          # set by callers (e.g. "sa" inside the fused SA stage) to label sink entries


def set_timing_sink(sink):
    """sink: None, or a list that receives (entry_point_name, int_args, start_event, end_event)."""
    global _timing_sink
    _timing_sink = sink


def _run(fn, anchor, *args):
    """Call a C-ABI entry point on `anchor`'s device and current stream; raise on error."""
    dev = anchor.device.index
    if dev != torch.cuda.current_device():
        with torch.cuda.device(anchor.device):
            return _run(fn, anchor, *args)
    sink = _timing_sink
    if sink is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args, _stream(dev))
        e1.record()
        sink.append((fn.__name__ + timing_tag, tuple(a for a in args if isinstance(a, int)), e0, e1))
    else:
        rc = fn(*args, _stream(dev))
    if rc != 0:
        raise RuntimeError(f"{fn.__name__} failed: {_lib0.omnipq_error_string(rc).decode()} ({rc})")


def gather_points(points, idx):
    """(B,C,N) f32, (B,M) i32 -> (B,C,M)   [sampling.cpp:22-46]"""
    _check(points, "points", torch.float32)
    _check(idx, "idx", torch.int32, cuda_like=points)
    _need_gpu(points)
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.empty((b, c, m), device=points.device, dtype=torch.float32)
    _run(_lib0.omnipq_gather_points, points, b, c, n, m, _ptr(points), _ptr(idx), _ptr(out))
    return out


def gather_xyz(xyz, idx, out=None):
    """(B,N,3) f32, (B,M) i32 -> (B,M,3): xyz[b, idx[b, m]] (no gradient; the reference's transpose / gather / transpose).
    out: write into this contiguous (B,M,3) f32 tensor."""
    _check(xyz, "xyz", torch.float32)
    _check(idx, "idx", torch.int32, cuda_like=xyz)
    _need_gpu(xyz)
    b, n, three = xyz.shape
    if three != 3:
        raise ValueError("gather_xyz: xyz must be (B, N, 3)")
    m = idx.shape[1]
    if out is None:
        out = torch.empty((b, m, 3), device=xyz.device, dtype=torch.float32)
    elif not (out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (b, m, 3) and out.device == xyz.device):
        raise ValueError("gather_xyz: out must be a contiguous (B, M, 3) float32 tensor on the inputs' device")
    _run(_lib0.omnipq_gather_xyz, xyz, b, n, m, _ptr(xyz), _ptr(idx), _ptr(out))
    return out


def gather_points_grad(grad_out, idx, n):
    """(B,C,M), (B,M) -> (B,C,n) scatter-add   [sampling.cpp:48-71]"""
    _check(grad_out, "grad_out", torch.float32)
    _check(idx, "idx", torch.int32, cuda_like=grad_out)
    _need_gpu(grad_out)
    b, c, m = grad_out.shape
    out = torch.zeros((b, c, int(n)), device=grad_out.device, dtype=torch.float32)
    _run(_lib0.omnipq_gather_points_grad, grad_out, b, c, int(n), m, _ptr(grad_out), _ptr(idx), _ptr(out))
    return out


_fps_ready = set()


def fps_poll():
    """Raise if a multi-workgroup FPS launch on the current device has reported a hand-off timeout since the last
    poll (non-blocking: the flag is pinned host memory the kernel writes through)."""
    rc = _lib0.omnipq_fps_poll()
    if rc != 0:
        raise RuntimeError(f"furthest point sampling: {_lib0.omnipq_error_string(rc).decode()} ({rc}) -- a "
                           "multi-workgroup launch gave up waiting for its sibling workgroups; its indices are invalid")


def _fps_prepare(device):
    if device.index not in _fps_ready:
        with torch.cuda.device(device):
            rc = _lib0.omnipq_fps_init()            # allocations happen here, outside any stream capture
        if rc != 0:
            raise RuntimeError(f"omnipq_fps_init failed: {_lib0.omnipq_error_string(rc).decode()} ({rc})")
        _fps_ready.add(device.index)
    fps_poll()


def furthest_point_sampling(points, nsamples, out=None, small_footprint=False):
    """(B,N,3) f32 -> (B,nsamples) i32   [sampling.cpp:72-93].  `out` (extension): write into an existing
    int32 tensor instead of allocating one.  small_footprint (extension): for clouds of more than 8192 points, fewer
    workgroups per scene with more points each -- same indices, ~40 % longer rounds on 40 % fewer compute units: for a
    sampling chain that runs underneath other work with time to spare (omnipq_furthest_point_sampling_ex: flags)."""
    _check(points, "points", torch.float32)
    _need_gpu(points)
    _fps_prepare(points.device)
    b, n = points.shape[0], points.shape[1]
    if out is None:
        out = torch.zeros((b, int(nsamples)), device=points.device, dtype=torch.int32)
    else:
        _check(out, "out", torch.int32, cuda_like=points)
        assert tuple(out.shape) == (b, int(nsamples))
    tmp = torch.full((b, n), 1e10, device=points.device, dtype=torch.float32)
    _run(_lib0.omnipq_furthest_point_sampling_ex, points, b, n, int(nsamples), _ptr(points), _ptr(tmp), _ptr(out),
         ctypes.c_uint(1 if small_footprint else 0))          # flags: OMNIPQ_FPS_SMALL_FOOTPRINT
    return out


def three_nn(unknowns, knows):
    """(B,n,3), (B,m,3) -> [dist2 (B,n,3) squared, idx (B,n,3) i32]   [interpolate.cpp:22-48]"""
    _check(unknowns, "unknowns", torch.float32)
    _check(knows, "knows", torch.float32, cuda_like=unknowns)
    _need_gpu(unknowns)
    b, n = unknowns.shape[0], unknowns.shape[1]
    m = knows.shape[1]
    idx = torch.empty((b, n, 3), device=unknowns.device, dtype=torch.int32)
    dist2 = torch.empty((b, n, 3), device=unknowns.device, dtype=torch.float32)
    _run(_lib0.omnipq_three_nn, unknowns, b, n, m, _ptr(unknowns), _ptr(knows), _ptr(dist2), _ptr(idx))
    return [dist2, idx]


def three_nn_weights(unknowns, knows, out=None):
    """(B,n,3), (B,m,3) -> (weight (B,n,3) f32, idx (B,n,3) i32): three_nn and the normalised inverse-distance weights of
    pointnet2_modules.py:395-397 in one launch.  out = (weight, idx): write into these contiguous tensors."""
    _check(unknowns, "unknowns", torch.float32)
    _check(knows, "knows", torch.float32, cuda_like=unknowns)
    _need_gpu(unknowns)
    b, n = unknowns.shape[0], unknowns.shape[1]
    m = knows.shape[1]
    dist2 = torch.empty((b, n, 3), device=unknowns.device, dtype=torch.float32)
    if out is None:
        idx = torch.empty((b, n, 3), device=unknowns.device, dtype=torch.int32)
        weight = torch.empty((b, n, 3), device=unknowns.device, dtype=torch.float32)
    else:
        weight, idx = out
        if not (weight.dtype == torch.float32 and idx.dtype == torch.int32 and weight.is_contiguous() and idx.is_contiguous()
                and tuple(weight.shape) == (b, n, 3) and tuple(idx.shape) == (b, n, 3)):
            raise ValueError("three_nn_weights: out must be contiguous (B, n, 3) float32 / int32 tensors")
    _run(_lib0.omnipq_three_nn_weights, unknowns, b, n, m, _ptr(unknowns), _ptr(knows), _ptr(dist2), _ptr(idx), _ptr(weight))
    return weight, idx


def three_interpolate(points, idx, weight):
    """(B,C,m), (B,n,3) i32, (B,n,3) f32 -> (B,C,n)   [interpolate.cpp:50-78]"""
    _check(points, "points", torch.float32)
    _check(idx, "idx", torch.int32, cuda_like=points)
    _check(weight, "weight", torch.float32, cuda_like=points)
    _need_gpu(points)
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.empty((b, c, n), device=points.device, dtype=torch.float32)
    _run(_lib0.omnipq_three_interpolate, points, b, c, m, n, _ptr(points), _ptr(idx), _ptr(weight), _ptr(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """(B,C,n) -> (B,C,m) scatter-add   [interpolate.cpp:80-107]"""
    _check(grad_out, "grad_out", torch.float32)
    _check(idx, "idx", torch.int32, cuda_like=grad_out)
    _check(weight, "weight", torch.float32, cuda_like=grad_out)
    _need_gpu(grad_out)
    b, c, n = grad_out.shape
    out = torch.zeros((b, c, int(m)), device=grad_out.device, dtype=torch.float32)
    _run(_lib0.omnipq_three_interpolate_grad, grad_out, b, c, n, int(m), _ptr(grad_out), _ptr(idx),
         _ptr(weight), _ptr(out))
    return out


_BQ_GRID_MIN = 8192      # points per scene from which the grid pays
_lib0.omnipq_ball_query_grid_workspace_bytes.restype = ctypes.c_longlong


def ball_query(new_xyz, xyz, radius, nsample, out=None):
    """(B,M,3), (B,N,3) -> (B,M,nsample) i32   [ball_query.cpp:16-40]; out: write into this contiguous int32 tensor"""
    _check(new_xyz, "new_xyz", torch.float32)
    _check(xyz, "xyz", torch.float32, cuda_like=new_xyz)
    _need_gpu(new_xyz)
    b, n = xyz.shape[0], xyz.shape[1]
    m = new_xyz.shape[1]
    if out is None:
        idx = torch.empty((b, m, int(nsample)), device=new_xyz.device, dtype=torch.int32)
    else:
        if not (out.dtype == torch.int32 and out.is_contiguous() and tuple(out.shape) == (b, m, int(nsample)) and
                out.device == new_xyz.device):
            raise ValueError("ball_query: out must be a contiguous (B, M, nsample) int32 tensor on the inputs' device")
        idx = out
    if n >= _BQ_GRID_MIN and radius > 0 and b <= 65535:
        # large clouds: the same indices through a hash grid (csrc/ball_query.hip) instead of n tests per centre
        ws = torch.empty((int(_lib0.omnipq_ball_query_grid_workspace_bytes(b, n)),), device=new_xyz.device,
                         dtype=torch.uint8)
        _run(_lib0.omnipq_ball_query_grid, new_xyz, b, n, m, ctypes.c_float(radius), int(nsample), _ptr(new_xyz),
             _ptr(xyz), _ptr(idx), _ptr(ws))
        return idx
    _run(_lib0.omnipq_ball_query, new_xyz, b, n, m, ctypes.c_float(radius), int(nsample), _ptr(new_xyz),
         _ptr(xyz), _ptr(idx))
    return idx


def group_points(points, idx):
    """(B,C,N), (B,M,S) i32 -> (B,C,M,S)   [group_points.cpp:19-42]"""
    _check(points, "points", torch.float32)
    _check(idx, "idx", torch.int32, cuda_like=points)
    _need_gpu(points)
    b, c, n = points.shape
    npoints, nsample = idx.shape[1], idx.shape[2]
    out = torch.empty((b, c, npoints, nsample), device=points.device, dtype=torch.float32)
    _run(_lib0.omnipq_group_points, points, b, c, n, npoints, nsample, _ptr(points), _ptr(idx), _ptr(out))
    return out


def group_points_grad(grad_out, idx, n):
    """(B,C,M,S), (B,M,S) -> (B,C,n) scatter-add   [group_points.cpp:44-67]"""
    _check(grad_out, "grad_out", torch.float32)
    _check(idx, "idx", torch.int32, cuda_like=grad_out)
    _need_gpu(grad_out)
    b, c, npoints, nsample = grad_out.shape
    out = torch.zeros((b, c, int(n)), device=grad_out.device, dtype=torch.float32)
    _run(_lib0.omnipq_group_points_grad, grad_out, b, c, int(n), npoints, nsample, _ptr(grad_out), _ptr(idx),
         _ptr(out))
    return out


def fps_check():
    """Raise if a multi-workgroup FPS launch on this device reported a hand-off timeout."""
    rc = _lib0.omnipq_fps_check(_stream())
    if rc != 0:
        raise RuntimeError(f"omnipq_fps_check: {_lib0.omnipq_error_string(rc).decode()} ({rc})")
