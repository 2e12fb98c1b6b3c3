"""In-process A/B timing of builds of libvarnet_hip on the small-step workloads of tools/step_timeline.py (whole steps: fused
kernel + reduce/optimizer kernel, HIP events around vn_train_epoch of 200 steps, interleaved rounds):
   python tools/ab_small.py <name,name,...> <workload,...> [rounds]   (libs: varnet_amd/libvarnet_hip_<name>.so; '' = shipped)"""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from varnet_amd import engine
WL = {'cfg2': ([50] * 4, 2, 1, 16, 10000, 450), 'mor': ([10, 20, 30], 3, 1, 16, 6000, 1750), 'cfg1': ([20], 2, 1, 16, 6000, 620),
      'w32': ([32] * 3, 3, 1, 16, 6000, 1750), 'deep20': ([20] * 6, 2, 1, 16, 6000, 620)}
names = sys.argv[1].split(',')
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
for wl in sys.argv[2].split(','):
    widths, d_in, dim, q, n_k, nB = WL[wl]
    n = n_k * q
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    X = torch.rand(n, d_in, device='cuda', generator=g) * 2 - 1
    G = torch.randn(n, dim, device='cuda', generator=g)
    bi = torch.rand(nB, d_in, device='cuda', generator=g) * 2 - 1
    bl = torch.randn(nB, device='cuda', generator=g)
    rng = np.random.default_rng(0)
    N1, dN1 = rng.uniform(0, 1, q), rng.standard_normal(q)
    engs = []
    for nm in names:
        lib = 'libvarnet_hip_%s.so' % nm if nm else 'libvarnet_hip.so'
        path = os.path.join(os.path.dirname(os.path.abspath(engine.__file__)), lib)
        import ctypes
        _probe = ctypes.CDLL(path)
        engine.VN_ABI_VERSION = _probe.vn_abi_version()     # an older build may report an older ABI; the entry points used here have not changed
        _all = getattr(engine, '_SIGS_ALL', None) or dict(engine._SIGS)
        engine._SIGS_ALL = _all
        engine._SIGS.clear(); engine._SIGS.update({k: v for k, v in _all.items() if hasattr(_probe, k)})
        engine._lib = None
        engine._lib = engine.load_library(path)
        e = engine.VNEngine(dim, d_in, widths, True, q)
        e.init_params(0); e.set_fe_table(N1, dN1); e.set_interior(0, X, G, None, n_k=n_k, detJ=1e-3)
        e.set_bic(bi, bl, nB // 2, 2.0); e.set_weights([1, 1, 1])
        engs.append(e)
    acc = torch.zeros((), device='cuda')
    for e in engs: e.train_epoch([0] * 20, acc)
    torch.cuda.synchronize()
    res = {nm: [] for nm in names}
    for r in range(rounds):
        for nm, e in zip(names, engs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e.train_epoch([0] * 200, acc); e1.record(); torch.cuda.synchronize()
            res[nm].append(e0.elapsed_time(e1) / 200 * 1e3)
    th = [e.get_params() for e in engs]
    for nm in names:
        v = np.array(res[nm])
        print('%-8s %-10s us/step: median %.2f  min %.2f  max %.2f   max|theta - theta[%s]| %.2e' % (wl, nm or 'shipped', np.median(v), v.min(), v.max(), names[0] or 'shipped', np.abs(th[names.index(nm)] - th[0]).max()))
    for e in engs: e.close()
