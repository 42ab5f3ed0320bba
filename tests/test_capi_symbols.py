"""CPU: the C-ABI library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports
every function include/omnipq_pointops.h declares.  No kernel is launched here."""
import ctypes
import capi


def both(built_lib):
    """the bfloat16 library and its IEEE-half twin (same sources, -DOMNIPQ_ELEM_F16: omni-pq_amd/build.py)"""
    return [built_lib, built_lib[:-3] + "_f16.so"]


def test_libraries_export_every_declared_symbol(built_lib):
    syms = capi.declared_symbols()
    assert len(syms) >= 25 and "omnipq_furthest_point_sampling" in syms and "omnipq_sa_gather" in syms
    assert "omnipq_gemm_nt_e16_stats" in syms
    for path in both(built_lib):
        lib = ctypes.CDLL(path)
        missing = [s for s in syms if not hasattr(lib, s)]
        assert not missing, (path, missing)


def test_libraries_have_gfx950_code_objects_with_their_mfma_opcode(built_lib, tmp_path):
    """Both libraries carry gfx950 code objects, and each one's matrix instructions are those of ITS element type only
    (disassembly of the bundled code objects: llvm-objdump --offloading extracts them next to its input)."""
    import glob
    import os
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    for path, opcode, other in zip(both(built_lib), ("v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x16_f16"),
                                   ("v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x16_bf16")):
        blob = open(path, "rb").read()
        assert b"amdgcn-amd-amdhsa--gfx950" in blob
        if not os.path.exists(objdump):
            continue
        work = tmp_path / os.path.basename(path)
        work.mkdir()
        shutil.copy(path, work)
        subprocess.run([objdump, "--offloading", os.path.basename(path)], cwd=work, capture_output=True)
        asm = "".join(subprocess.run([objdump, "-d", f], capture_output=True, text=True).stdout
                      for f in glob.glob(str(work / "*gfx950")))
        if "v_mfma" not in asm:
            continue                                  # this objdump could not open the bundles: nothing to check
        assert asm.count(opcode) >= 100 and other not in asm, (path, asm.count(opcode), asm.count(other))


def test_library_names_its_plan_aware_entry_points(built_lib):
    """the binding takes the set of plan-aware entry points from the LIBRARY it loaded (a build-time constant),
    and refuses a library of another ABI version; here: that constant equals what the header of this tree declares, in
    both element-type libraries, and the binding's set is the library's."""
    import sys
    for path in both(built_lib):
        lib = ctypes.CDLL(path)
        lib.omnipq_plan_aware_entry_points.restype = ctypes.c_char_p
        names = frozenset(lib.omnipq_plan_aware_entry_points().decode().split())
        assert names == capi.plan_aware() and len(names) >= 20, sorted(names ^ capi.plan_aware())
        assert all(hasattr(lib, n) for n in names)
    import pointnet2_utils
    assert pointnet2_utils._ext.PLAN_AWARE == capi.plan_aware()
    assert pointnet2_utils._ext.ABI_VERSION == lib.omnipq_abi_version() == 3


def test_host_helpers_without_gpu(built_lib):
    from oracle import oracle_ext
    lib = capi.lib()
    assert lib.omnipq_abi_version() == 3
    assert b"invalid" in lib.omnipq_error_string(10001)
    # the FPS tie geometry must agree between library, oracle and the reference's formula
    for n in list(range(1, 70)) + [100, 127, 128, 255, 256, 511, 512, 513, 1000, 1024, 2048, 4096, 8192,
                                   40000, 50000, 80000, 1 << 20]:
        assert lib.omnipq_opt_n_threads(n) == oracle_ext.opt_n_threads(n), n
    assert lib.omnipq_opt_n_threads(40000) == 512 and lib.omnipq_opt_n_threads(100) == 64


def test_argument_validation_needs_no_gpu(built_lib):
    """Every entry point rejects malformed arguments with OMNIPQ_EINVAL / OMNIPQ_ETOOLARGE BEFORE it touches the
    device (the reference's wrappers `exit(-1)` from inside the launch, cuda_utils.h:35-44).  Pointers below
    are never dereferenced: validation comes first."""
    lib = capi.lib()
    EINVAL, ETOOLARGE = 10001, 10002
    p = ctypes.c_void_p(0x1000)                       # "some non-null pointer"
    null = ctypes.c_void_p(0)
    f = ctypes.c_float
    ll = ctypes.c_longlong
    strides = (ll * 8)(288, 2304, 288, 2304, 288, 2304, 288, 2304)
    # attention: head dim not a multiple of 4 / too wide, dropout outside [0, 1), problem beyond 32-bit indexing
    assert lib.omnipq_attn_fwd(8, 8, 256, 256, 37, p, p, p, p, strides, p, f(0.0), null, 0, null) == EINVAL
    assert lib.omnipq_attn_fwd(8, 8, 256, 256, 52, p, p, p, p, strides, p, f(0.0), null, 0, null) == EINVAL
    assert lib.omnipq_attn_fwd(8, 8, 256, 256, 36, p, p, p, p, strides, p, f(1.0), p, 0, null) == EINVAL
    assert lib.omnipq_attn_fwd(8, 8, 256, 256, 36, p, p, p, p, strides, p, f(0.1), null, 0, null) == EINVAL   # no seed
    assert lib.omnipq_attn_fwd(64, 16, 4096, 4096, 36, p, p, p, p, strides, p, f(0.0), null, 0, null) == ETOOLARGE
    # GEMMs: contraction length must be a multiple of the K step, leading dimensions of 8
    assert lib.omnipq_gemm_nt_e16(128, 128, 33, p, 40, p, 40, p, 128, null, null) == EINVAL
    assert lib.omnipq_gemm_nt_e16_stats(128, 128, 32, p, 32, p, 32, p, 128, null, null, null, null, null) == EINVAL  # no sums
    assert lib.omnipq_gemm_tn_e16(100, 128, 64, p, 100, p, 128, p, p, null, null) == EINVAL                      # M % 8
    lib.omnipq_gemm_nt_stats_workspace_floats.restype = ll
    assert lib.omnipq_gemm_nt_stats_workspace_floats(64 * 128, 256) == 0
    assert lib.omnipq_gemm_nt_stats_workspace_floats(64 * 128 + 1, 256) == 65 * 2 * 256
    # row kernels
    assert lib.omnipq_add_dropout_layernorm(ll(10), 290, p, p, p, p, f(1e-5), f(0.0), null, 0, p, p, null, null, p, p,
                                            null) == EINVAL                                                   # C % 4
    assert lib.omnipq_add_dropout_layernorm(ll(10), 288, p, p, p, p, f(1e-5), f(0.0), null, 0, p, p, p, null, p, p,
                                            null) == EINVAL                                     # pe without its output
    assert lib.omnipq_relu_dropout(ll(10), p, f(0.0), null, 0, null) == EINVAL                                 # n % 4
    assert lib.omnipq_interp_rows(2, 10, 5, 64, p, p, p, p, 64, 8, null) == EINVAL             # columns do not fit
    assert lib.omnipq_place_rows(ll(10), 64, p, p, 100, 0, null) == EINVAL                               # pitch % 8
    assert lib.omnipq_colsum_f32(ll(10), 12, p, p, null) == EINVAL
    # loss helper: more views than the kernel-argument table holds
    ptrs = (ctypes.c_void_p * 73)(*([0x1000] * 73))
    ones = (ctypes.c_int * (73 * 4))(*([1] * 73 * 4))
    flags = (ctypes.c_int * 73)()
    assert lib.omnipq_sum_of_means(73, ptrs, ones, ones, flags, p, null) == EINVAL
    # zero-sized problems are no-ops that succeed without a device
    assert lib.omnipq_gemm_nt_e16(0, 128, 32, p, 32, p, 32, p, 128, null, null) == 0
    assert lib.omnipq_interp_rows(0, 10, 5, 64, p, p, p, p, 64, 0, null) == 0
    assert lib.omnipq_relu_dropout(ll(0), p, f(0.0), null, 0, null) == 0
