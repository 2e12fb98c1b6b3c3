"""In-process A/B timing of two builds of libvarnet_hip (same device, interleaved rounds):
   python tools/ab_perf.py <nameA> <nameB> [kernel] [rounds]    (libs: varnet_amd/libvarnet_hip_<name>.so)"""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from varnet_amd import engine
if ',' in sys.argv[1]:                      # python tools/ab_perf.py a,b,c [kernel] [rounds]
    names = sys.argv[1].split(','); rest = sys.argv[2:]
else:
    names = sys.argv[1:3]; rest = sys.argv[3:]
kernel = int(rest[0]) if len(rest) > 0 else 0
rounds = int(rest[1]) if len(rest) > 1 else 5
d_in, dim, widths, q, n_k, nB = 3, 2, [50]*5, 64, 100000, 14000
n = n_k*q
g = torch.Generator(device='cuda'); g.manual_seed(0)
Input = torch.rand(n, d_in, device='cuda', generator=g)*2-1
gcoef = torch.randn(n, dim, device='cuda', generator=g)
bi = torch.rand(nB, d_in, device='cuda', generator=g)*2-1
bl = torch.randn(nB, device='cuda', generator=g)
rng = np.random.default_rng(0)
N1, dN1 = rng.uniform(0, 1, q), rng.standard_normal(q)
engs = []
for nm in names:
    path = os.path.join(os.path.dirname(os.path.abspath(engine.__file__)), 'libvarnet_hip_%s.so' % nm)
    import ctypes
    engine.VN_ABI_VERSION = ctypes.CDLL(path).vn_abi_version()      # an older build may report an older ABI; the entry points used here have not changed
    engine._lib = None
    _probe = ctypes.CDLL(path)
    _all = getattr(engine, '_SIGS_ALL', None) or dict(engine._SIGS)
    engine._SIGS_ALL = _all
    engine._SIGS.clear(); engine._SIGS.update({k: v for k, v in _all.items() if hasattr(_probe, k)})   # older builds lack newer entry points
    engine._lib = engine.load_library(path)
    e = engine.VNEngine(dim, d_in, widths, True, q, kernel=kernel)
    e.init_params(0); e.set_fe_table(N1, dN1); e.set_interior(0, Input, gcoef, None, n_k=n_k, detJ=1e-6)
    e.set_bic(bi, bl, 9000, 2.0); e.set_weights([1, 1, 1])
    for _ in range(3): e.train_step(0)
    engs.append(e)
torch.cuda.synchronize()
res = {nm: [] for nm in names}
for r in range(rounds):
    for nm, e in zip(names, engs):
        e.profile_begin()
        for _ in range(8): e.train_step(0)
        ms, nl, kn = e.profile_end()
        res[nm].append(ms)
for nm in names:
    v = np.array(res[nm]); print('%-12s kernel ms: median %.4f  min %.4f  max %.4f' % (nm, np.median(v), v.min(), v.max()))
a = np.median(res[names[0]])
for nm in names[1:]: print('%s / %s = %.4f' % (nm, names[0], np.median(res[nm]) / a))
