"""Diagnostic: per-phase cycle shares of the fused kernel (needs libvarnet_hip_stamps.so built with -DVN_STAMPS)."""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from varnet_amd import engine
engine.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(engine.__file__)), os.environ.get("VN_STAMPS_LIB", "libvarnet_hip_stamps.so"))
from varnet_amd.engine import VNEngine
L = int(sys.argv[1]) if len(sys.argv) > 1 else 5
H = int(sys.argv[2]) if len(sys.argv) > 2 else 50
d_in, dim, widths, q, n_k, nB = 3, 2, [H]*L, 64, 100000, 14000
wl = sys.argv[5] if len(sys.argv) > 5 else ''          # optional small-step workloads (L, H ignored)
if wl == 'mor':  d_in, dim, widths, q, n_k, nB = 3, 1, [10, 20, 30], 16, 6000, 1750
if wl == 'cfg2': d_in, dim, widths, q, n_k, nB = 2, 1, [50]*4, 16, 10000, 450
if wl == 'cfg1': d_in, dim, widths, q, n_k, nB = 2, 1, [20], 16, 6000, 620
n = n_k*q
g = torch.Generator(device='cuda'); g.manual_seed(0)
Input = torch.rand(n, d_in, device='cuda', generator=g)*2-1
gcoef = torch.randn(n, dim, device='cuda', generator=g)
eng = VNEngine(dim, d_in, widths, True, q, kernel=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
eng.init_params(0)
rng = np.random.default_rng(0)
eng.set_fe_table(rng.uniform(0,1,q), rng.standard_normal(q))
eng.set_interior(0, Input, gcoef, None, n_k=n_k, detJ=1e-6)
eng.set_bic(torch.rand(nB, d_in, device='cuda')*2-1, torch.randn(nB, device='cuda'), nB*9//14, 2.0)
eng.set_weights([1,1,1])
for _ in range(3): eng.train_step(0)
torch.cuda.synchronize()
st = np.array(eng.debug_stamps(), dtype=np.float64)
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 1
names = ['inputs+forward', 'out layer + int1', 'wait barrier 1', 'R_k', 'wait barrier 2', 'seeds', 'zbar_L', 'rest of reverse'] if mode == 3 else ['inputs+forward', 'epilogue', 'wgrad publish', 'wait publish barrier', 'wgrad contraction', 'wait release barrier', 'bwd GEMMs + zbar', '-'] if mode == 2 else ['inputs', 'fwd GEMMs', 'output+epilogue', 'zbar_L', 'wgrad out', 'wgrad hidden', 'bwd-data+zbar', 'wgrad L1']
tiles = (n/128 + nB/128)/256
print('workload', wl or '%dx%d' % (L, H), 'tiles per workgroup %.2f' % tiles)
tot = st.sum()
print('cycles/tile:', tot/tiles)
for nm, v in zip(names, st): print('%-18s %6.2f%%  %10.0f /tile' % (nm, 100*v/tot, v/tiles))
