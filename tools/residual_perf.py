"""Time of the strong-residual and fp64 entry points (monitors, residual-driven sampling, config 5's fp64 check) on n points, 5x50 net:
vn_residual (fp32, second-order forward mode: value, gradient and Laplacian), vn_residual_f64, vn_forward_f64.   python tools/residual_perf.py [n]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from varnet_amd.engine import VNEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d_in, dim, widths = 3, 2, [50] * 5
g = torch.Generator(device='cuda'); g.manual_seed(0)
X = torch.rand(n, d_in, device='cuda', generator=g) * 2 - 1
diff = torch.rand(n, device='cuda', generator=g) * 0.01
vel = torch.randn(n, dim, device='cuda', generator=g)
e = VNEngine(dim, d_in, widths, True, 64)
e.init_params(0)
F_pt = 2 * (d_in * 50 + 4 * 2500 + 50)


def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps


from varnet_amd.engine import _ptr
u32, r32 = torch.empty(n, device='cuda'), torch.empty(n, device='cuda')
X64, diff64, vel64 = X.double(), diff.double(), vel.double()
u64, r64 = torch.empty(n, device='cuda', dtype=torch.float64), torch.empty(n, device='cuda', dtype=torch.float64)
a = t(lambda: e._ck(e.lib.vn_residual(e.h, _ptr(X), _ptr(diff), _ptr(vel), None, None, n, _ptr(u32), _ptr(r32))))
# round 6: hidden widths 33..64 run the bf16-piece kernels (vn_split16.hip); route 2 = the f32-MFMA kernels on the same inputs
# (the f32-MFMA forms of this network live in the tests' cross-check library: a second engine of that library, same parameters)
r_split, u_split = r32.clone(), u32.clone()
ex = VNEngine(dim, d_in, widths, True, 64, xcheck=True)
ex.set_params(e.get_params())
ex.debug_point_route(2)
a32 = t(lambda: ex._ck(ex.lib.vn_residual(ex.h, _ptr(X), _ptr(diff), _ptr(vel), None, None, n, _ptr(u32), _ptr(r32))))
d32 = t(lambda: ex.forward(X))
uf32 = ex.forward(X)
ex.close()
uf = e.forward(X)
torch.cuda.synchronize()
print('bf16-piece kernels against the f32-MFMA kernels on the same inputs: residual max |diff| / max |res| %.2e, forward %.2e'
      % (float((r_split - r32).abs().max() / r32.abs().max()), float((uf - uf32).abs().max() / uf32.abs().max())))
print('vn_residual      fp32, f32-MFMA kernel (vn_taylor16) %8.3f ms   vn_forward, f32-MFMA kernel (vn_pgrad16 value sweep) %8.3f ms' % (a32 * 1e3, d32 * 1e3))
b = t(lambda: e._ck(e.lib.vn_residual_f64(e.h, _ptr(X64), _ptr(diff64), _ptr(vel64), None, None, n, _ptr(u64), _ptr(r64))))
c = t(lambda: e.forward_f64(X64))
d = t(lambda: e.forward(X))
# second-order forward mode: value + dim first + dim second tangents = (1 + 2 dim) streams of F_pt
print('n = %d, 5x50, dim %d' % (n, dim))
print('vn_residual      fp32 %8.3f ms  %.3e points/s  (%.1f TFLOP/s of (1 + 2 dim) F_pt)' % (a * 1e3, n / a, (1 + 2 * dim) * F_pt * n / a / 1e12))
print('vn_residual_f64       %8.3f ms  %.3e points/s' % (b * 1e3, n / b))
print('vn_forward_f64        %8.3f ms  %.3e points/s  (%.1f TFLOP/s of F_pt)' % (c * 1e3, n / c, F_pt * n / c / 1e12))
print('vn_forward       fp32 %8.3f ms  %.3e points/s  (%.1f TFLOP/s of F_pt)' % (d * 1e3, n / d, F_pt * n / d / 1e12))
e.close()
