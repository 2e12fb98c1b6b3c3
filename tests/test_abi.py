"""CPU tier: the C-ABI library loads and exports every symbol include/varnet_hip.h declares;
without a GPU the engine fails loudly instead of falling back (no compute calls here)."""
import ctypes as C
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'varnet_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(vn_[a-z0-9_]+)\s*\(', txt)))


def _lib():
    from varnet_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return engine, engine.load_library()


def test_header_symbols_exported():
    engine, lib = _lib()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), 'missing export ' + n
    assert set(names) == set(engine.ABI_SYMBOLS), set(names) ^ set(engine.ABI_SYMBOLS)
    assert lib.vn_abi_version() == engine.VN_ABI_VERSION == 7
    hdr = open(os.path.join(ROOT, 'include', 'varnet_hip.h')).read()
    assert re.search(r'#define\s+VN_ABI_VERSION\s+7\b', hdr)


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    engine, lib = _lib()
    cfg = engine.VnConfig()
    cfg.dim, cfg.d_in, cfg.n_layers, cfg.integ_num = 1, 2, 1, 16
    cfg.widths[0] = 5
    cfg.lr, cfg.beta1, cfg.beta2, cfg.eps = 1e-3, 0.9, 0.999, 1e-8
    h = C.c_void_p()
    rc = lib.vn_create(C.byref(cfg), C.byref(h))
    assert rc != 0 and b'no CPU fallback' in lib.vn_last_error()
    with pytest.raises(Exception):
        engine.VNEngine(1, 2, [5], True, 16)


def test_argument_validation_without_gpu():
    engine, lib = _lib()
    cfg = engine.VnConfig()
    cfg.dim, cfg.d_in, cfg.n_layers, cfg.integ_num = 1, 2, 17, 16    # more layers than a vn_config can describe
    h = C.c_void_p()
    assert lib.vn_create(C.byref(cfg), C.byref(h)) == 1
    cfg.n_layers = 1
    cfg.widths[0] = 5000                                              # wider than VN_MAX_WIDTH
    assert lib.vn_create(C.byref(cfg), C.byref(h)) == 1
    cfg.widths[0], cfg.d_in = 20, 40                                  # more inputs than VN_MAX_DIN
    assert lib.vn_create(C.byref(cfg), C.byref(h)) == 1
    # nets beyond the kernels' range (9 layers, 500 wide) are legal configs now: they pass validation and fail
    # only for want of a GPU here (VN_EHIP = 2), not as bad arguments
    cfg.d_in, cfg.n_layers = 2, 9
    cfg.lr, cfg.beta1, cfg.beta2, cfg.eps = 1e-3, 0.9, 0.999, 1e-8
    for i in range(9):
        cfg.widths[i] = 500
    assert lib.vn_create(C.byref(cfg), C.byref(h)) in (0, 2)
    if h.value:
        lib.vn_destroy(h)
    assert lib.vn_create(None, C.byref(h)) == 1


def test_zero_initialised_config_is_rejected_not_trained():
    """the Adam hyper-parameters are taken literally (no defaulting), so a memset-zero vn_config -- which a
    round-1 C host could pass -- must be refused (eps = 0 gives 0/0 = NaN for a zero-gradient parameter), with a message
    that names the field; beta outside [0, 1) likewise; RMSProp ignores the three fields; lr = 0 stays legal."""
    engine, lib = _lib()
    h = C.c_void_p()
    cfg = engine.VnConfig()                                            # ctypes zero-initialises: lr = beta = eps = 0
    cfg.dim, cfg.d_in, cfg.n_layers, cfg.integ_num = 1, 2, 1, 16
    cfg.widths[0] = 5
    assert lib.vn_create(C.byref(cfg), C.byref(h)) == 1 and b'epsilon' in lib.vn_last_error()
    cfg.eps = 1e-8
    cfg.beta1 = 1.0
    assert lib.vn_create(C.byref(cfg), C.byref(h)) == 1 and b'beta1' in lib.vn_last_error()
    cfg.beta1, cfg.beta2 = 0.9, -0.1
    assert lib.vn_create(C.byref(cfg), C.byref(h)) == 1 and b'beta2' in lib.vn_last_error()
    cfg.beta2 = 0.999                                                  # lr = 0: legal (TFModel.py:130 only rejects negatives)
    assert lib.vn_create(C.byref(cfg), C.byref(h)) in (0, 2)           # 2 = no GPU here
    if h.value:
        lib.vn_destroy(h)
        h = C.c_void_p()
    cfg.beta1 = cfg.beta2 = cfg.eps = 0.0
    cfg.optimizer = 1                                                  # VN_OPT_RMSPROP: TF-1 constants, fields unused
    assert lib.vn_create(C.byref(cfg), C.byref(h)) in (0, 2)
    if h.value:
        lib.vn_destroy(h)


def test_stale_library_is_refused(tmp_path):
    """load_library checks vn_abi_version before binding: a side build with another vn_config layout must not
    be driven through this binding.  (A stub that reports ABI 3 stands in for a stale build.)"""
    import subprocess
    engine, _ = _lib()
    src = tmp_path / 'stale.c'
    src.write_text('int vn_abi_version(void) { return 3; }\n')
    so = tmp_path / 'libvarnet_hip_stale.so'
    subprocess.run(['gcc', '-shared', '-fPIC', str(src), '-o', str(so)], check=True)
    with pytest.raises(engine.VNError, match='ABI version 3'):
        engine.load_library(str(so))


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, 'varnet_amd')
    for f in os.listdir(pkg):
        if f.endswith('.py'):
            src = open(os.path.join(pkg, f)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f


def test_product_library_knows_no_vendor_gemm_library():
    """the env-switched rocBLAS twin of the layer-by-layer route was a dual path inside the product
    library.  Neither the sources of the package nor the built .so (strings, DT_NEEDED) name a BLAS library any more; the
    only library the engine opens by name is RCCL (the one collective)."""
    import subprocess
    for d in ('varnet_amd', os.path.join('varnet_amd', 'csrc'), 'include'):
        for f in os.listdir(os.path.join(ROOT, d)):
            if f.endswith(('.py', '.hip', '.h', '.c', '.cpp')) or f == 'Makefile':
                assert not re.search(r'(?i)blas', open(os.path.join(ROOT, d, f)).read()), os.path.join(d, f)
    lib = os.path.join(ROOT, 'varnet_amd', 'libvarnet_hip.so')
    blob = open(lib, 'rb').read()
    assert b'rocblas' not in blob.lower() and b'hipblas' not in blob.lower()
    needed = subprocess.run(['readelf', '-d', lib], capture_output=True, text=True).stdout
    assert 'NEEDED' in needed and not re.search(r'(?i)blas', needed)


@pytest.mark.gpu
def test_plain_c_host_trains_through_the_abi(tmp_path):
    """examples/c_host_step.c: gcc-compiled C program (HIP runtime API for memory, dlopen of the library) runs 200
    training steps through the C ABI -- no Python, no PyTorch in that process -- and the loss falls."""
    from tests import conftest, rank_worker as rw
    if conftest.FORKSERVER is None:
        pytest.skip('no fork server')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    q = conftest.FORKSERVER.Queue()
    p = conftest.FORKSERVER.Process(target=rw.run_c_host, args=(root, str(tmp_path), q))
    p.start()
    rc, text = q.get(timeout=600)
    p.join(60)
    assert rc == 0 and 'C_HOST_OK' in text, text
