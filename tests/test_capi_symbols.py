"""CPU: the C-ABI library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports
every function include/omnipq_pointops.h declares.  No kernel is launched here."""
import ctypes
import capi


def test_library_exports_every_declared_symbol(built_lib):
    syms = capi.declared_symbols()
    assert len(syms) >= 25 and "omnipq_furthest_point_sampling" in syms and "omnipq_sa_gather" in syms
    lib = ctypes.CDLL(built_lib)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_library_has_gfx950_code_object(built_lib):
    blob = open(built_lib, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob


def test_host_helpers_without_gpu(built_lib):
    from oracle import oracle_ext
    lib = capi.lib()
    assert lib.omnipq_abi_version() == 1
    assert b"invalid" in lib.omnipq_error_string(10001)
    # the FPS tie geometry must agree between library, oracle and the reference's formula
    for n in list(range(1, 70)) + [100, 127, 128, 255, 256, 511, 512, 513, 1000, 1024, 2048, 4096, 8192,
                                   40000, 50000, 80000, 1 << 20]:
        assert lib.omnipq_opt_n_threads(n) == oracle_ext.opt_n_threads(n), n
    assert lib.omnipq_opt_n_threads(40000) == 512 and lib.omnipq_opt_n_threads(100) == 64
