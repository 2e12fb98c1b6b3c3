#!/usr/bin/env python
"""
bench.py -- training-points/sec of the VarNet variational-loss step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.md cfg 3 / cfg 4): Operator_2Dt advection-diffusion problem, rectangle
[0,2]x[-.5,.5], kappa=1e-3, v=(1,0), T=1.5, inlet BC; discNum=[50,40], bDiscNum=40, tDiscNum=50
-> 100 000 test functions x 64 quadrature points = 6.4 M training points per step, 5x50
sigmoid MLP, TF-1 Adam lr 1e-3, fp32.  All arrays are produced by the package's own problem
layer (synthetic in the sense that no dataset is read; weights are glorot-uniform seed 0).
A "step" = forward + weak-form loss + backward + (gradient SUM all-reduce) + Adam on the whole
training set.  With N > 1 ranks the SAME 100 000 test functions are sharded contiguously
(strong scaling, BASELINE cfg 4) and the flat gradient is all-reduced over RCCL.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: Peak FP32 (matrix), dense


def build_problem(cfg):
    from varnet_amd.domain import Domain1D, PolygonDomain2D
    from varnet_amd.adpde import ADPDE
    from varnet_amd.varnet import VarNet
    if cfg == 3:
        verts = np.array([[0.0, -0.5], [0.0, -0.2], [0.0, 0.2], [0.0, 0.5], [2.0, 0.5], [2.0, -0.5]])
        BC = [[], [0.0, 1.0, 1.0], [], [], [], []]
        pde = ADPDE(PolygonDomain2D(verts), diff=1e-3, vel=[1., 0.], tInterval=[0, 1.5], BCs=BC, IC=0.0)
        vn = VarNet(pde, layerWidth=[50] * 5, discNum=[50, 40], bDiscNum=40, tDiscNum=50)
        name = '2D+t AD-PDE (Operator_2Dt), 5x50 MLP, 1e5 test functions x 64 quadrature points'
    elif cfg == 2:
        pde = ADPDE(Domain1D(), diff=0.1 / np.pi, vel=1.0, tInterval=[0, 2.0], IC=lambda x: -np.sin(np.pi * x))
        vn = VarNet(pde, layerWidth=[50] * 4, discNum=50, bDiscNum=None, tDiscNum=200)
        name = '1D+t AD-PDE (Operator_1Dt), 4x50 MLP, 1e4 test functions x 16 quadrature points'
    else:
        raise SystemExit('unknown --config')
    return vn, name


def cpu_baseline(vn, tdata, budget_s=20.0):
    """The oracle (fp32 PyTorch-CPU autograd restatement of the reference graph) timed on this
    host's cores on a bounded sample of the same workload: the first n_s test functions."""
    import torch
    from oracle import tf1_graph as og
    fd = vn.fixData
    q = fd.integNum
    d = tdata.mor[0]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # a one-GPU box exposes every host core but grants a share of 16 (more threads only thrash)
    cores = int(os.environ.get('VN_CPU_THREADS', min(cores, 16)))
    torch.set_num_threads(cores)
    flat = vn.engine.get_params()
    w = np.array([1.0, 1.0, 1.0])

    def step(n_s):
        rows = n_s * q
        kw = dict(Input=d['Input'][:rows].cpu().numpy(), gcoef=d['gcoef'][:rows].cpu().numpy(), source=None,
                  N=np.tile(fd.N, n_s).reshape(rows, 1).astype(np.float32),
                  dNt=np.tile(fd.dNt, n_s).reshape(rows, 1).astype(np.float32), integW=None,
                  intShape=[n_s, q], detJ=float(fd.detJ), detJvec=False,
                  biInput=d['biInput'].cpu().numpy(), biLabel=d['biLabel'].cpu().numpy().reshape(-1, 1),
                  bDof=fd.bDofsum, biDimVal=float(fd.biDimVal), w=w, dim=vn.dim, time_dependent=True,
                  is_source=False, integWflag=False)
        t0 = time.perf_counter()
        og.loss_and_grad(flat, vn.inpDim, vn.layerWidth, torch.float32, **kw)
        return time.perf_counter() - t0

    n_s = min(10000, fd.nt)
    step(n_s)                                   # warm-up
    t1 = step(n_s)
    reps = max(1, min(8, int(budget_s / max(t1, 1e-3))))
    ts = [step(n_s) for _ in range(reps)]
    dt = float(np.median(ts))
    return {"value": n_s * q / dt, "unit": "training-points/s", "cores": cores, "kind": "port",
            "sample": "%d of %d test functions (%d points) + all %d BC/IC points per step, %d timed steps, "
                      "fp32 PyTorch-CPU autograd restatement of TFModel.py:515-714 (oracle/tf1_graph.py)"
                      % (n_s, fd.nt, n_s * q, d['biInput'].shape[0], reps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-dedup', action='store_true', help='skip the extra de-duplicated-formulation timing')
    args = ap.parse_args()

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d' % (args.gpus, args.gpus))
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)          # several ranks may share a GPU only in the gloo rehearsal below
    torch.cuda.set_device(local)
    os.environ['LOCAL_RANK'] = str(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # backend 'nccl' IS RCCL on ROCm; VN_DIST_BACKEND=gloo rehearses the N>1 path on one GPU
        backend = os.environ.get('VN_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    vn, wname = build_problem(args.config)
    fd, eng = vn.fixData, vn.engine
    tdata = vn._build_tdata()                     # shards by rank when world > 1
    tdata.select_mor(0)
    w = np.array([1.0, 1.0, 1.0])
    w[:2] /= world                                # VarNetUtility.py:900-901 (BC/IC replicated per tower)
    eng.set_weights(w)
    gb = eng.bind_grad_buffer()
    P = eng.P
    n0, n1 = tdata.block(0)
    rows_local = (n1 - n0) * fd.integNum
    nB = tdata.mor[0]['biInput'].shape[0]
    nT_total = fd.nT

    def step():
        eng.grad(0)
        if world > 1:
            dist.all_reduce(gb)
        eng.apply()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    eng.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms, klaunches, kname = eng.profile_end()
    loss_after = float(gb[P].item())
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- extra, separately reported: de-duplicated formulation (one network evaluation per unique
    # quadrature point; SURVEY.md 8d "honest accounting").  Never mixed into `value`.
    dd = None
    if not args.no_dedup and world == 1:          # the scaling runs time the headline formulation only
        U_local = tdata.enable_dedup()
        if U_local:
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            dtd = time.perf_counter() - t1
            if world > 1:
                t = torch.tensor([dtd, float(U_local)], dtype=torch.float64, device='cuda')
                tm = t.clone()
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                dtd, U_tot = float(tm[0].item()), int(t[1].item())
            else:
                U_tot = int(U_local)
            dd = (dtd, U_tot, float(gb[P].item()))

    if rank == 0:
        F_pt = 2 * (vn.inpDim * vn.layerWidth[0] + sum(a * b for a, b in zip(vn.layerWidth[:-1], vn.layerWidth[1:]))
                    + vn.layerWidth[-1])
        flop_launch = 6.0 * F_pt * rows_local + 3.0 * F_pt * nB
        achieved = flop_launch / (kms * 1e-3) / 1e12 if kms > 0 else None
        # HBM bytes per launch come from the separate rocprofv3 --pmc passes (FETCH_SIZE doubled per the
        # gfx950 correction, WRITE_SIZE as is) committed under profiles/; only quoted for the exact
        # workload and kernel they were collected on.
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'r1_pmc_traffic.json')
        if args.config == 3 and world == 1 and kname.startswith('vn_fused') and os.path.exists(tfile):
            traffic = json.load(open(tfile)).get('hbm_bytes_per_launch')
        out = {
            "metric": "training-points/sec (test-funcs x quad-pts), 2D+t AD-PDE" if args.config == 3
            else "training-points/sec (test-funcs x quad-pts), 1D+t AD-PDE",
            "value": nT_total * args.steps / dt,
            "unit": "training-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (problem-layer generated Operator_%s inputs, glorot-uniform seed-0 weights)"
                    % ('2Dt' if args.config == 3 else '1Dt'),
            "config": {"workload": wname, "test_functions": int(fd.nt), "quad_points_per_test_function": int(fd.integNum),
                       "training_points_per_step": int(nT_total), "bc_ic_points": int(nB),
                       "net": "%dx%d sigmoid MLP, d_in=%d, %d params" % (len(vn.layerWidth), vn.layerWidth[0], vn.inpDim, P),
                       "optimizer": "TF1-Adam lr=1e-3", "sharding": "contiguous test-function blocks per rank, SUM all-reduce of %d floats" % (P + 4),
                       "test_functions_per_sec": fd.nt * args.steps / dt, "loss_after": loss_after},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": (achieved / PEAK_FP32_MFMA_TFLOPS) if achieved else None, "traffic": traffic,
                         "kernel": kname, "kernel_ms": kms, "launches_timed": klaunches,
                         "algorithmic_flop_per_launch": flop_launch,
                         "note": "6*F_pt per interior point + 3*F_pt per BC/IC point, F_pt=%d; HIP events on the engine stream" % F_pt},
        }
        if dd is not None:
            dtd, U_tot, loss_dd = dd
            # formulation actually run: per unique point dim forward passes (2 F_pt) + dim reverse passes (6 F_pt)
            flop_dd = 8.0 * vn.dim * F_pt * U_tot + 3.0 * F_pt * nB
            out["dedup"] = {
                "value": nT_total * args.steps / dtd, "unit": "training-points/s (reference units: rows per step / time)",
                "ms_per_step": dtd / args.steps * 1e3, "unique_points": U_tot,
                "rows_per_unique_point": nT_total / max(U_tot, 1), "loss_after": loss_dd,
                "achieved_TFLOPs_of_formulation_run": flop_dd / (dtd / args.steps) / 1e12,
                "note": "separate speed-up, not the headline: each unique quadrature point is evaluated once "
                        "(value + input gradient, one tangent pass per spatial dimension) instead of once per "
                        "(test function, point) row; same loss and gradient up to fp32 rounding "
                        "(tests/test_engine_gpu.py::test_dedup_formulation_parity)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(vn, tdata)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
