"""In-process A/B of the 8-wave kernel and the one-wave-per-SIMD kernel (VN_PW=0 / 1) on config-3- and config-2-sized synthetic
inputs: kernel time from HIP events on the engine stream, interleaved rounds.
    python tools/pw_perf.py [rounds]"""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from varnet_amd.engine import VNEngine
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = torch.Generator(device='cuda'); g.manual_seed(0)
rng = np.random.default_rng(0)


def make(pw, d_in, dim, widths, q, n_k, nB, bDof):
    os.environ['VN_PW'] = '1' if pw else '0'
    n = n_k * q
    e = VNEngine(dim, d_in, widths, True, q)
    e.init_params(0)
    e.set_fe_table(rng.uniform(0, 1, q), rng.standard_normal(q))
    return e


for name, (d_in, dim, widths, q, n_k, nB, bDof, steps) in {
        'config 3 (6.4M points, 5x50)': (3, 2, [50] * 5, 64, 100000, 14000, 9000, 8),
        'config 2 (160k points, 4x50)': (2, 1, [50] * 4, 16, 10000, 450, 400, 200),
        'W=8 shard (800k points, 5x50)': (3, 2, [50] * 5, 64, 12500, 14000, 9000, 30)}.items():
    n = n_k * q
    Input = torch.rand(n, d_in, device='cuda', generator=g) * 2 - 1
    gcoef = torch.randn(n, dim, device='cuda', generator=g)
    bi = torch.rand(nB, d_in, device='cuda', generator=g) * 2 - 1
    bl = torch.randn(nB, device='cuda', generator=g)
    engs = {}
    for pw in (0, 1):
        e = make(pw, d_in, dim, widths, q, n_k, nB, bDof)
        e.set_interior(0, Input, gcoef, None, n_k=n_k, detJ=1e-6)
        e.set_bic(bi, bl, bDof, 2.0); e.set_weights([1, 1, 1])
        e.train_epoch([0] * 3, None)
        engs[pw] = e
    torch.cuda.synchronize()
    res = {0: [], 1: []}
    wall = {0: [], 1: []}
    import time
    for r in range(rounds):
        for pw, e in engs.items():
            e.profile_begin()
            t0 = time.perf_counter()
            e.train_epoch([0] * steps, None)
            torch.cuda.synchronize()
            wall[pw].append((time.perf_counter() - t0) / steps * 1e3)
            ms, nl, kn = e.profile_end()
            res[pw].append(ms)
    F_pt = 2 * (d_in * widths[0] + sum(a * b for a, b in zip(widths[:-1], widths[1:])) + widths[-1])
    flop = 6.0 * F_pt * n + 3.0 * F_pt * nB
    print(name)
    for pw in (0, 1):
        v = np.array(res[pw]); w = np.array(wall[pw])
        print('   VN_PW=%d  kernel ms: median %.4f  min %.4f  max %.4f   step ms %.4f   kernel frac of 157.3 TF/s: %.4f'
              % (pw, np.median(v), v.min(), v.max(), np.median(w), flop / (np.median(v) * 1e-3) / 157.3e12))
    print('   pw / 8-wave = %.4f' % (np.median(res[1]) / np.median(res[0])))
    th0, th1 = engs[0].get_params(), engs[1].get_params()
    print('   max |theta_pw - theta_8w| after the same steps: %.3e (max |theta| %.3e)' % (np.max(np.abs(th0 - th1)), np.max(np.abs(th0))))
    for e in engs.values():
        e.close()
