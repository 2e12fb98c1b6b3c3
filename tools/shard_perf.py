"""Per-rank compute time of the strong-scaling bench at world sizes 1,2,4,8, measured on ONE GPU by giving
the engine rank 0's shard only (no collective): an upper bound on the scaling the 8-GPU run can show."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
base = None
for world in (1, 2, 4, 8):
    vn, _ = bench.build_problem(3)
    vn.world, vn.rank = world, 0                      # shard as rank 0 of `world`; vn.dist stays None
    fd, eng = vn.fixData, vn.engine
    td = vn._build_tdata(); td.select_mor(0)
    w = np.array([1.0, 1.0, 1.0]); w[:2] /= world
    eng.set_weights(w)
    gb = eng.bind_grad_buffer()
    def step():
        eng.grad(0); eng.apply()
    for _ in range(3): step()
    torch.cuda.synchronize()
    eng.profile_begin()
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    kms, nl, kn = eng.profile_end()
    n0, n1 = td.block(0)
    base = base or dt
    print('world %d: %6d test functions/rank  step %.3f ms  kernel %.3f ms  -> speed-up before the collective %.2fx' % (world, n1 - n0, dt * 1e3, kms, base / dt))
    eng.close()
