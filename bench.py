#!/usr/bin/env python
"""
bench.py -- training-points/sec of the VarNet variational-loss step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3] [--no-cpu-baseline]
        (N > 1: starts its own N ranks, one per GPU, before anything touches the GPU)
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


# which files determine the code of a profiled kernel (the instantiation's name starts with the key): its translation unit, every
# header that unit includes, and the compile flags of the Makefile (-fno-slp-vectorize and friends decide the code too)
_F16 = ('varnet_amd/csrc/vn_fused16.hip', 'varnet_amd/csrc/vn_fused16_common.h', 'varnet_amd/csrc/vn_internal.h', 'include/varnet_hip.h')
KERNEL_SOURCES = {'vn_fused16_kernel': _F16,
                  'vn_pgrad16_kernel': ('varnet_amd/csrc/vn_pgrad16.hip', 'varnet_amd/csrc/vn_pgrad16.h', 'varnet_amd/csrc/vn_points16.h') + _F16[1:],
                  'vn_split16_': ('varnet_amd/csrc/vn_split16.hip', 'varnet_amd/csrc/vn_split16.h', 'varnet_amd/csrc/vn_points16.h') + _F16[1:],
                  'vn_dedup_': ('varnet_amd/csrc/vn_dedup.hip', 'varnet_amd/csrc/vn_dedup.h') + _F16[2:]}


def split16_serves(layerWidth):
    """Labels only (the engine decides: vn_api.hip, vn_split16_supported): hidden widths 33..64 with 2..7 hidden layers (6 beyond 50
    wide) run their point kernels -- vn_forward, vn_residual, vn_forward_grad / the de-duplicated step's (u, grad u) pass -- on the bf16
    matrix pipe as six products of exact bf16 pieces (vn_split16.hip); the others on the f32-MFMA kernels."""
    W = list(layerWidth)
    return 33 <= max(W) <= 64 and 2 <= len(W) <= (7 if max(W) <= 50 else 6)
# counter files bench.py may quote `roofline.traffic` from, by --config (tools/collect_profiles.sh + tools/summarise_profiles.py)
TRAFFIC_FILES = {3: 'pmc_traffic.json', 2: 'pmc_traffic_cfg2.json'}
DEDUP_TRAFFIC_FILE = 'pmc_traffic_dedup.json'


def strip_comments(src):
    """C/C++ source without comments and with runs of white space collapsed (string and character literals kept):
    what is left decides the generated code, a reworded comment or a re-wrapped line does not."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c == '/' and i + 1 < n and src[i + 1] == '/':
            while i < n and src[i] != '\n':
                i += 1
        elif c == '/' and i + 1 < n and src[i + 1] == '*':
            j = src.find('*/', i + 2)
            i = n if j < 0 else j + 2
            out.append(' ')
        elif c in '"\'':
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == '\\' else 1
            out.append(src[i:j + 1])
            i = j + 1
        else:
            out.append(c)
            i += 1
    return ' '.join(''.join(out).split())


def effective_cxxflags():
    """The compile flags of varnet_amd/csrc/Makefile as a plain `make` resolves them (ARCH = gfx950, EXTRA empty)."""
    import re
    mk = open(os.path.join(ROOT, 'varnet_amd', 'csrc', 'Makefile')).read()
    var = {'ARCH': 'gfx950', 'EXTRA': ''}
    m = re.search(r'^CXXFLAGS\s*=\s*(.*)$', mk, re.M)
    flags = m.group(1) if m else ''
    flags = re.sub(r'\$\((\w+)\)', lambda mm: var.get(mm.group(1), ''), flags)
    return ' '.join(flags.split())


def kernel_source_hash(kernel='vn_fused16_kernel'):
    """sha256 over what determines `kernel`'s code: the comment-stripped, white-space-normalised translation unit and headers
    (KERNEL_SOURCES) and the Makefile's effective compile flags.  Ties a committed counter file to the code it measured; edits to
    other kernels' files, comments and layout do not move it."""
    import hashlib
    key = [k for k in KERNEL_SOURCES if str(kernel).startswith(k)]
    if not key:
        return None
    h = hashlib.sha256()
    for f in KERNEL_SOURCES[key[0]]:
        h.update(f.encode())
        h.update(strip_comments(open(os.path.join(ROOT, f)).read()).encode())
    h.update(('CXXFLAGS ' + effective_cxxflags()).encode())
    return h.hexdigest()


def static_traffic(kname, config, world):
    """HBM bytes per launch of the dominant kernel.  PMC counters cannot be read from inside the run (they need a
    rocprofv3 --pmc pass of their own), so the figure is the committed one from profiles/pmc_traffic*.json
    (tools/collect_profiles.sh + tools/summarise_profiles.py: FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE as
    is) -- and ONLY if that file was collected on this exact kernel instantiation, this workload and this kernel's code.
    Returns (traffic or None, traffic_source string)."""
    if world != 1 or config not in TRAFFIC_FILES:
        return None, 'none: counter passes are committed for the N=1 launches of configs %s only' % sorted(TRAFFIC_FILES)
    rel = 'profiles/' + TRAFFIC_FILES[config]
    tfile = os.path.join(ROOT, rel)
    if not os.path.exists(tfile):
        return None, 'none: %s absent' % rel
    js = json.load(open(tfile))
    norm = lambda n: ''.join(str(n).split())
    if js.get('config') != config:
        return None, 'none: %s holds the config-%s launch' % (rel, js.get('config'))
    if norm(js.get('kernel')) != norm(kname):
        return None, 'none: %s was collected on %s, this run launches %s' % (rel, js.get('kernel'), kname)
    if js.get('kernel_source_sha256') != kernel_source_hash(kname):
        return None, ('none: %s (%s) predates the current code of this kernel (sha256 of its comment-stripped sources differs): '
                      're-run tools/collect_profiles.sh' % (rel, js.get('round')))
    return js.get('hbm_bytes_per_launch'), ('%s (static: rocprofv3 --pmc passes of round %s on %s, sha256 of its comment-stripped '
                                            'sources %s... = this build)' % (rel, js.get('round'), js.get('kernel'), js['kernel_source_sha256'][:12]))


def dedup_traffic(config):
    """HBM bytes per step of the de-duplicated formulation's four kernels (committed counter passes, profiles/pmc_traffic_dedup.json),
    quoted only when every kernel's file still hashes to what the counters were collected on.  (traffic or None, source string)"""
    rel = 'profiles/' + DEDUP_TRAFFIC_FILE
    f = os.path.join(ROOT, rel)
    if not os.path.exists(f):
        return None, 'none: %s absent' % rel
    js = json.load(open(f))
    if js.get('config') != config:
        return None, 'none: %s holds the config-%s launches' % (rel, js.get('config'))
    for k in js.get('kernels', []):
        if k.get('kernel_source_sha256') != kernel_source_hash(k['kernel']):
            return None, 'none: %s predates the current code of %s: re-run tools/collect_profiles.sh' % (rel, k['kernel'].split('<')[0])
    return js.get('hbm_bytes_per_step'), '%s (static: rocprofv3 --pmc passes of round %s; per kernel: %s)' % (
        rel, js.get('round'), ', '.join('%s %.1f MB' % (k['kernel'].split('<')[0].split('(')[0], k['hbm_bytes_per_launch'] / 1e6) for k in js['kernels']))


def issue_model(kname, key, cal=None):
    """What bounds a kernel whose f32 MFMAs and f32 vector instructions share one datapath (gfx950: DESIGN.md 3.2) is matrix-pipe
    cycles + vector-instruction issue cycles per SIMD, not the MFMA peak alone; for nets up to 32 wide the vector share is large
    (elementwise work scales with H, matrix work with H^2).  The committed counter passes (profiles/pmc_issue.json, written by
    tools/summarise_profiles.py) give both terms per launch; quoted ONLY for the kernel instantiation and kernel code they
    were collected on.  frac_of_issue_bound = (matrix + vector cycles) / kernel cycles of the same profiled launches."""
    f = os.path.join(ROOT, 'profiles', 'pmc_issue.json')
    if not os.path.exists(f):
        return {"available": False, "why": "profiles/pmc_issue.json absent"}
    js = json.load(open(f)).get(key)
    norm = lambda n: ''.join(str(n).split())
    if not js:
        return {"available": False, "why": "no counter pass for %s" % key}
    if norm(js['kernel']) != norm(kname):
        return {"available": False, "why": "counters were collected on %s, this run launches %s" % (js['kernel'], kname)}
    if js['kernel_source_sha256'] != kernel_source_hash(kname):
        return {"available": False, "why": "counter pass predates the current code of this kernel: re-run tools/collect_profiles.sh"}
    return {"available": True, "bound_that_applies": "fp32 issue: matrix-pipe cycles + vector-instruction cycles on the shared datapath",
            "matrix_cycles_per_simd": js['matrix_cycles_per_simd'],
            "vector_instructions_per_simd": js['vector_instructions_per_simd'],
            "cycles_per_vector_instruction": js['cycles_per_vector_instruction'],
            "issue_bound_cycles": js['issue_bound_cycles'], "kernel_cycles_same_pass": js['kernel_cycles_same_pass'],
            "frac_of_issue_bound": js['frac_of_issue_bound'],
            # the bound depends on what a vector instruction costs on the shared datapath: 4 cycles is the guide's ONE-wave issue
            # cost, the kernels run two waves per SIMD, where independent v_fma_f32 measure ~2.7 (tools/micro/overlap2.hip;
            # re-measured live: roofline.calibration) -- so the fraction is a RANGE, its low end the honest one
            "frac_of_issue_bound_range": [(js['matrix_cycles_per_simd'] + c * js['vector_instructions_per_simd']) / js['kernel_cycles_same_pass']
                                          for c in ((cal or {}).get('cycles_per_vector_instruction_2_waves_per_simd') or 2.7, 4.0)],
            "cycles_per_vector_instruction_range": [(cal or {}).get('cycles_per_vector_instruction_2_waves_per_simd') or 2.7, 4.0],
            "matrix_pipe_busy": js['matrix_pipe_busy'],
            "source": "profiles/pmc_issue.json[%s] (rocprofv3 --pmc passes of round %s on %s)" % (key, js.get('round'), js['kernel']),
            "note": "f32 MFMA and f32 VALU share one datapath on gfx950: the bound of a launch is matrix-pipe cycles + c cycles per "
                    "vector instruction per SIMD, c between the measured two-waves-per-SIMD issue cost (~2.7) and the guide's "
                    "one-wave 4 (transcendentals cost more); frac_of_issue_bound uses c = 4, frac_of_issue_bound_range = [c measured, 4]; "
                    "`frac` above prices the same launch against the MFMA peak alone"}


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
    elif cfg == 1:
        pde = ADPDE(Domain1D(), diff=0.1 / np.pi, vel=1.0, tInterval=[0, 2.0], IC=lambda x: -np.sin(np.pi * x))
        vn = VarNet(pde, layerWidth=[20] * 3, discNum=20, bDiscNum=None, tDiscNum=300)
        name = '1D+t AD-PDE (Operator_1Dt), 3x20 MLP, 6e3 test functions x 16 quadrature points'
    elif cfg == 5:
        from varnet_amd.mor import MOR

        def diffFun(x, t=0, D=0.01):
            return D * np.ones([np.shape(x)[0], 1])

        def disc(discNum=6):
            return np.array([0.003 * (11 ** (n / (discNum - 1))) for n in range(discNum)])[np.newaxis].T
        mor = MOR(diffFun, ['D'], [[0.003, 0.033]])
        pde = ADPDE(Domain1D(), diff=diffFun, vel=1.0, timeDependent=True, tInterval=[0, 2.0], IC=lambda x: -np.sin(np.pi * x),
                    MORvar=mor)
        vn = VarNet(pde, layerWidth=[10, 20, 30], discNum=150, bDiscNum=None, tDiscNum=800, MORdiscScheme=disc)
        name = '1D+t parametric-kappa AD-PDE (Operator_1DtMOR), [10,20,30] MLP, 6 kappa x 20 mini-batches of 6e3 test functions x 16 points'
    else:
        raise SystemExit('unknown --config')
    return vn, name


def host_cores():
    """(physical cores lscpu reports, cores this job may actually use).  A one-GPU box exposes every core of
    the host but grants a share (cgroup cpu.max / affinity); threads beyond the share only thrash."""
    import subprocess
    phys = None
    try:
        out = subprocess.run(['lscpu', '-p=core,socket'], capture_output=True, text=True, timeout=10).stdout
        phys = len({ln for ln in out.splitlines() if ln and not ln.startswith('#')}) or None
    except Exception:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(f).read().split()
            if f.endswith('cpu.max'):
                if txt[0] != 'max':
                    usable = min(usable, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    usable = min(usable, max(1, q // int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())))
            break
        except Exception:
            continue
    if phys is None:
        phys = usable
    return phys, usable


def cpu_baseline(vn, tdata, budget_s=25.0):
    """
    The oracle (fp32 PyTorch-CPU autograd restatement of the reference graph + TF-1 Adam, oracle/tf1_graph.py)
    timed on this host's cores on a bounded sample of the same workload -- the first n_s test functions and
    all BC/IC points -- as a TRAINING run from the bench's theta_0: >= 20 timed steps after 5 warm-up steps
    at the full thread count, 2 timed steps at one thread.  The HIP engine then trains on exactly that sample
    from the same theta_0 and the per-step loss trajectories are compared.
    """
    import torch
    from oracle import tf1_graph as og
    fd, eng = vn.fixData, vn.engine
    q = fd.integNum
    d = tdata.mor[0]
    phys, usable = host_cores()
    # a one-GPU box grants a share of 16 host cores whatever the affinity mask says
    cores = int(os.environ.get('VN_CPU_THREADS', min(phys, usable, 16)))
    n_s = min(int(os.environ.get('VN_CPU_SAMPLE', 2000)), fd.nt)
    rows = n_s * q
    nB = d['biInput'].shape[0]
    w = np.array([1.0, 1.0, 1.0])
    kw = dict(Input=d['Input'][:rows].cpu().numpy(), gcoef=d['gcoef'][:rows].cpu().numpy(),
              source=None if d['source'] is None else d['source'][:rows].cpu().numpy().reshape(rows, 1),
              N=np.tile(fd.N, n_s).reshape(rows, 1).astype(np.float32),
              dNt=np.tile(fd.dNt, n_s).reshape(rows, 1).astype(np.float32), integW=None,
              intShape=[n_s, q], detJ=float(fd.detJ), detJvec=False,
              biInput=d['biInput'].cpu().numpy(), biLabel=d['biLabel'].cpu().numpy().reshape(-1, 1),
              bDof=fd.bDofsum, biDimVal=float(fd.biDimVal), w=w, dim=vn.dim, time_dependent=vn.PDE.timeDependent,
              is_source=vn.lossOpt['isSource'], integWflag=False)
    eng.init_params(seed=0)
    theta0 = eng.get_params()

    def run(nsteps, threads):
        torch.set_num_threads(threads)
        theta = theta0.copy()
        adam = og.TF1Adam(theta.size, lr=vn.learning_rate, dtype=np.float32)
        losses, times = [], []
        for _ in range(nsteps):
            t0 = time.perf_counter()
            res, g = og.loss_and_grad(theta, vn.inpDim, vn.layerWidth, torch.float32, **kw)
            theta = adam.step(theta, g)
            times.append(time.perf_counter() - t0)
            losses.append(res['loss'])
        return np.array(losses), np.array(times)

    warm = 5
    _, t_w = run(2, cores)                                     # page-in + thread pool start
    per = float(np.min(t_w))
    n_timed = int(max(20, min(60, budget_s / max(per, 1e-3) - warm)))
    if per * (20 + warm) > 3 * budget_s:                       # very slow host: keep the run bounded, say so
        n_timed = max(3, int(budget_s / per))
    losses, times = run(warm + n_timed, cores)
    dt = float(np.mean(times[warm:]))
    _, t1 = run(3, 1)
    dt1 = float(np.mean(t1[1:]))

    # the same training run on the HIP engine: same sample, same theta_0, same optimizer
    eng.set_interior(1, d['Input'][:rows], d['gcoef'][:rows], None if d['source'] is None else d['source'][:rows],
                     n_k=n_s, detJ=float(fd.detJ))
    eng.set_weights(w)
    eng.init_params(seed=0)
    lg = torch.zeros(1, dtype=torch.float32, device=eng.device)
    gl = []
    for _ in range(warm + n_timed):
        eng.train_step(1, lg)
        gl.append(float(lg.item()))
    gl = np.array(gl)
    dev = np.abs(gl - losses) / np.abs(losses)
    full = None
    if os.environ.get('VN_CPU_FULL', '1') != '0':
        full = cpu_full_size(vn, tdata, kw, theta0, cores)
    return {"value": rows / dt, "unit": "training-points/s", "cores": cores, "kind": "port", "full_size": full,
            "value_1_thread": rows / dt1,
            "physical_cores_lscpu": phys, "cores_usable_by_this_job": usable,
            "timed_steps": n_timed, "warmup_steps": warm, "ms_per_step": dt * 1e3,
            "loss_traj_max_rel_dev": float(dev.max()), "loss_traj_rel_dev_last": float(dev[-1]),
            "loss_first_last_cpu": [float(losses[0]), float(losses[-1])],
            "loss_first_last_gpu": [float(gl[0]), float(gl[-1])],
            "sample": "%d of %d test functions (%d points) + all %d BC/IC points per step, %d timed Adam steps after %d "
                      "warm-up at %d threads (host has %d physical cores, this job may use %d), 2 timed steps at 1 "
                      "thread; fp32 PyTorch-CPU autograd restatement of TFModel.py:515-714 + TF-1 Adam "
                      "(oracle/tf1_graph.py); loss_traj_*: HIP engine vs this CPU run, same sample, same theta_0"
                      % (n_s, fd.nt, rows, nB, n_timed, warm, cores, phys, usable)}


def dedup_leg(vn, tdata, eng, step_fn, steps, warmup, nT_total, nB, F_pt, loss_of, config=None):
    """The de-duplicated formulation on the workload just timed (SURVEY.md 8d "honest accounting"): separately reported,
    never `value`.  Per unique quadrature point 8 F_pt: value + input gradient in one pass (2 F_pt, vn_pgrad16), weak-form
    assembly over rows and seed gather (HBM-bound), ONE reverse launch of the fused kernel with the per-point direction
    (6 F_pt incl. the recomputed forward).  Its own roofline: the formulation's FLOPs over the HIP-event time of its kernel
    sequence (vn_profile_* brackets steps 1-4 of run_dedup on the engine stream)."""
    import torch
    U = tdata.enable_dedup()
    if not U:
        return None
    for _ in range(warmup):
        step_fn()
    torch.cuda.synchronize()
    eng.profile_begin()
    t1 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    dtd = time.perf_counter() - t1
    kms, kl, kname = eng.profile_end()
    loss_dd = loss_of()
    tdata.disable_dedup()                          # what runs after this leg is the row-wise formulation again
    flop_dd = 8.0 * F_pt * U + 3.0 * F_pt * nB
    traffic, traffic_source = dedup_traffic(config)
    return {
        "value": nT_total * steps / dtd, "unit": "training-points/s (reference units: rows per step / time)",
        "ms_per_step": dtd / steps * 1e3, "steps": steps, "unique_points": int(U),
        "rows_per_unique_point": nT_total / max(U, 1), "loss_after": loss_dd,
        "roofline": {"bound": "mfma", "achieved": flop_dd / (kms * 1e-3) / 1e12 if kms else None, "peak": PEAK_FP32_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": flop_dd / (kms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS if kms else None,
                     "whole_step_frac": flop_dd / (dtd / steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                     "kernels": [("vn_split16_pgrad_kernel" if split16_serves(vn.layerWidth) else "vn_pgrad16_kernel") +
                                 " (u and du/dx_d at the unique points: value forward + value-adjoint sweep, 2 F_pt)",
                                 "vn_dedup_seed_kernel (rows: weak-form integrand, R_k, loss, per-row seeds; HBM-bound)",
                                 "vn_dedup_gather_kernel (unique points: CSR sum of their rows' seeds; HBM-bound)",
                                 "%s, reverse mode with the per-point direction sg and tangent seed 1 (6 F_pt, BC/IC tiles ride along)" % kname],
                     "kernel_sequence_ms": kms, "launches_timed": kl, "formulation_flop_per_step": flop_dd,
                     "traffic": traffic, "traffic_source": traffic_source,
                     "note": "FLOPs of the formulation that ran (8 F_pt per unique point + 3 F_pt per BC/IC point), not of the row-wise "
                             "one; per-kernel times: profiles/r6_dedup_kernel_stats.csv"},
        "note": "separate speed-up, not the headline: each unique quadrature point is evaluated once instead of once per "
                "(test function, point) row; same loss and gradient up to fp32 rounding "
                "(tests/test_engine_gpu.py::test_dedup_formulation_parity)"}


def cpu_full_size(vn, tdata, kw_sample, theta0, cores, budget_s=45.0):
    """SURVEY.md 8(d) "same inputs / seed / step count": the oracle on the FULL workload (all test functions, all BC/IC points),
    1 warm-up + 3 timed Adam steps at the granted thread count -- the sample's per-point rate checked at full size, not
    extrapolated.  One autograd graph over 6.4 M rows would hold ~60 tensors of 1.3 GB; the set is therefore fed as the
    reference feeds towers: T contiguous blocks of whole test functions, gradients SUMMED (TFModel.py:370), BC/IC weights / T
    (VarNetUtility.py:819-857, 900-901) -- the same arithmetic the reference runs with processors=[...] * T.  The HIP engine
    takes the same steps on the full batch from the same theta_0; the per-step losses are compared."""
    import torch
    from oracle import tf1_graph as og
    fd, eng = vn.fixData, vn.engine
    q, nt = fd.integNum, fd.nt
    d = tdata.mor[0]
    # block size: 2 000 test functions (128 k rows), the sample's -- the host's per-point rate is 2.3 x lower on 10 blocks of 640 k
    # rows (tensors of 128 MB each: DRAM-bound; 2.28e5 points/s, 28 s per step on the round-5 box) than on blocks that stay in
    # cache, and the faster way to run the same arithmetic is the fairer baseline
    T = int(os.environ.get('VN_CPU_FULL_TOWERS', max(1, int(np.ceil(nt / 2000.0)))))
    blk = -(-nt // T)
    feeds = []
    for t in range(T):
        k0, k1 = t * blk, min(nt, (t + 1) * blk)
        if k1 <= k0:
            continue
        r0, r1 = k0 * q, k1 * q
        kw = dict(kw_sample)
        kw.update(Input=d['Input'][r0:r1].cpu().numpy(), gcoef=d['gcoef'][r0:r1].cpu().numpy(),
                  source=None if d['source'] is None else d['source'][r0:r1].cpu().numpy().reshape(r1 - r0, 1),
                  N=np.tile(fd.N, k1 - k0).reshape(r1 - r0, 1).astype(np.float32),
                  dNt=np.tile(fd.dNt, k1 - k0).reshape(r1 - r0, 1).astype(np.float32), intShape=[k1 - k0, q])
        feeds.append(kw)
    T = len(feeds)
    w = np.array([1.0 / T, 1.0 / T, 1.0])
    torch.set_num_threads(cores)

    def one_step(theta, adam):
        loss, g = 0.0, 0.0
        for kw in feeds:
            kw['w'] = w
            res, gt = og.loss_and_grad(theta, vn.inpDim, vn.layerWidth, torch.float32, **kw)
            loss += res['loss']
            g = g + gt
        return adam.step(theta, g), loss

    theta = theta0.copy()
    adam = og.TF1Adam(theta.size, lr=vn.learning_rate, dtype=np.float32)
    losses, times = [], []
    n_timed = 3
    i = 0
    while i < 1 + n_timed:
        t0 = time.perf_counter()
        theta, l = one_step(theta, adam)
        times.append(time.perf_counter() - t0)
        losses.append(l)
        if i == 0 and times[0] * 3 > budget_s:          # slow host: keep the run bounded, say so
            n_timed = max(1, int(budget_s / times[0]))
        i += 1
    dt = float(np.mean(times[1:]))
    # the same steps on the HIP engine: full batch 0, same theta_0, weights [1, 1, 1]
    eng.set_weights(np.array([1.0, 1.0, 1.0]))
    eng.init_params(seed=0)
    lg = torch.zeros(1, dtype=torch.float32, device=eng.device)
    gl = []
    for _ in range(len(losses)):
        eng.train_step(0, lg)
        gl.append(float(lg.item()))
    dev = np.abs(np.array(gl) - np.array(losses)) / np.abs(np.array(losses))
    rows = nt * q
    return {"value": rows / dt, "unit": "training-points/s", "cores": cores, "ms_per_step": dt * 1e3, "timed_steps": int(len(times) - 1),
            "warmup_steps": 1, "towers": T, "rows_per_step": int(rows), "loss_traj_max_rel_dev": float(dev.max()),
            "loss_first_last_cpu": [float(losses[0]), float(losses[-1])], "loss_first_last_gpu": [float(gl[0]), float(gl[-1])],
            "sample": "ALL %d test functions (%d points) + all BC/IC points per step, fed as %d tower blocks of <= %d test functions whose "
                      "gradients are summed (TFModel.py:370; BC/IC weights / %d): %d timed Adam step(s) after 1 warm-up at %d threads"
                      % (nt, rows, T, blk, T, len(times) - 1, cores)}


def inference_line(vn, tdata, eng, F_pt):
    """The callers either side of the training step, on the headline's own points: `vn_forward` (VarNet.evaluate: value-only sweep of
    vn_pgrad16, F_pt per point) on all training rows and `vn_residual` (strong residual incl. the Laplacian: every monitor of the training
    loop and the residual-driven re-sampling; second-order forward mode on the matrix pipe, vn_taylor16, (3 dim + 2) F_pt per point) on
    the first 10^6 of them.  HIP-side time only (inputs resident); reported under `extra`, never `value`."""
    import torch
    from varnet_amd.engine import _ptr
    d = tdata.mor[0]
    X = d['Input']
    n = int(X.shape[0])
    m = min(n, 1000000)
    dim = vn.dim
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    diff = torch.full((m,), 1e-3, device='cuda'); vel = torch.randn(m, dim, device='cuda', generator=g)
    u = torch.empty(n, device='cuda'); r = torch.empty(m, device='cuda')

    def t(fn, reps=5):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    tf = t(lambda: eng._ck(eng.lib.vn_forward(eng.h, _ptr(X), n, _ptr(u))))
    tr = t(lambda: eng._ck(eng.lib.vn_residual(eng.h, _ptr(X), _ptr(diff), _ptr(vel), None, None, m, _ptr(u), _ptr(r))))
    # streams of F_pt a residual point costs: vn_taylor16 runs dim + 1 passes (3 streams per spatial direction, 2 for time); the
    # bf16-piece kernel lets the time tangent ride in pass 0 (3 dim + 1 streams, vn_split16.hip)
    # which matrix-pipe form serves these two calls (labels only; the engine decides: vn_api.hip, vn_split16_supported): hidden
    # widths 33..64 with 2..7 hidden layers (6 beyond 50 wide) run on the bf16 pipe as six products of exact bf16 pieces
    split = split16_serves(vn.layerWidth)
    td = 1 if vn.PDE.timeDependent else 0
    nd = (3 * dim + td) if split else (3 * dim + 2 * td)
    # the fp64 checking path (BASELINE config 5's fp64 residual check runs these entry points): vn_taylor16d on the fp64 matrix pipe.
    # Kernel time from events on the engine's stream (= torch's current stream, VNEngine.use_current_stream); priced against the
    # fp64 MFMA rate THIS box sustains (vn_debug_calibrate_f64: the guide quotes no fp64 matrix peak), FLOPs stated both ways.
    X64, diff64, vel64 = X[:m].double(), diff.double(), vel.double()
    u64 = torch.empty(m, device='cuda', dtype=torch.float64); r64 = torch.empty(m, device='cuda', dtype=torch.float64)

    def tev(fn, reps=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    tf64 = tev(lambda: eng._ck(eng.lib.vn_forward_f64(eng.h, _ptr(X64), m, _ptr(u64))))
    tr64 = tev(lambda: eng._ck(eng.lib.vn_residual_f64(eng.h, _ptr(X64), _ptr(diff64), _ptr(vel64), None, None, m, _ptr(u64), _ptr(r64))))
    cal64 = eng.calibrate_f64(eng.calibrate()["clock_ghz_implied_by_the_mfma_loop"])
    pk64 = cal64["mfma_f64_tflops"]
    W = list(vn.layerWidth)
    ks, mt = -(-max(W) // 4), -(-max(W) // 16)
    # matrix-pipe FLOPs vn_taylor16d issues per point and stream: every layer padded to 4*ks inputs x 16*mt outputs (no padding branches)
    Fx_pt = 2.0 * 16 * mt * (4 * (-(-vn.inpDim // 4)) + (len(W) - 1) * 4 * ks)
    on_pipe = max(W) <= 64 and len(W) <= 8
    nd64 = 3 * dim + 2 * td                            # vn_taylor16d keeps a pass of its own for the time direction
    f64 = {"peak_measured_tflops": pk64, "calibration": cal64,
           "forward_f64": {"points": m, "kernel_ms": tf64 * 1e3, "points_per_s": m / tf64, "flop_algorithmic": F_pt * m,
                           "flop_issued_on_the_matrix_pipe": Fx_pt * m, "tflops_algorithmic": F_pt * m / tf64 / 1e12,
                           "frac_of_peak_measured": F_pt * m / tf64 / 1e12 / pk64,
                           "frac_of_peak_measured_issued": Fx_pt * m / tf64 / 1e12 / pk64},
           "residual_f64": {"points": m, "kernel_ms": tr64 * 1e3, "points_per_s": m / tr64, "flop_algorithmic": nd64 * F_pt * m,
                            "flop_issued_on_the_matrix_pipe": nd64 * Fx_pt * m, "tflops_algorithmic": nd64 * F_pt * m / tr64 / 1e12,
                            "frac_of_peak_measured": nd64 * F_pt * m / tr64 / 1e12 / pk64,
                            "frac_of_peak_measured_issued": nd64 * Fx_pt * m / tr64 / 1e12 / pk64},
           "kernel": "vn_taylor16d_kernel (v_mfma_f64_16x16x4_f64; libm exp / tanh, true division)" if on_pipe else "per-thread kernels",
           "note": "peak = what a loop of independent v_mfma_f64_16x16x4_f64 sustains on this box in this process (two waves per SIMD); "
                   "(3 dim + 2) F_pt per residual point (a pass of its own for the time direction); `issued` counts the padded 16-row tiles the kernel really runs"}
    return {"fp64": f64, "forward": {"points": n, "ms": tf * 1e3, "points_per_s": n / tf, "tflops_of_F_pt": F_pt * n / tf / 1e12,
                        "frac_of_peak": F_pt * n / tf / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                        "kernel": "vn_split16_kernel<NS = 1> (hidden layers as six v_mfma_f32_16x16x32_bf16 products of exact bf16 pieces; "
                                  "priced against the f32 MFMA peak for comparison with rounds 1-5)" if split else "vn_pgrad16_kernel (value-only sweep)"},
            "residual": {"points": m, "ms": tr * 1e3, "points_per_s": m / tr, "tflops_executed": nd * F_pt * m / tr / 1e12,
                         "frac_of_peak": nd * F_pt * m / tr / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         "streams_of_F_pt_per_point": nd,
                         "kernel": ("vn_split16_kernel<NS = 3> (bf16 pieces; six v_mfma_f32_16x16x32_bf16 products per layer product: `frac_of_peak` "
                                    "prices the algorithmic f32 FLOPs against the f32 MFMA peak for comparison with round 5, the kernel itself "
                                    "runs on the bf16 pipe): (3 dim + 1) F_pt per point, dim passes of three chained streams, the time tangent a "
                                    "fourth stream of pass 0" if split else
                                    "vn_taylor16_kernel: (3 dim + 2) F_pt per point, one pass of three chained streams per coordinate direction")},
            "note": "vn_forward = VarNet.evaluate; vn_residual = TFModel.py:743-754, what every training monitor (VarNet.py:1363) and the "
                    "residual-driven re-sampling (VarNet.py:1696-1868) call; profiles/r5_forward_perf.txt, r5_residual_perf.txt"}


def small_step_line(cfg, steps, warmup, cal=None, with_dedup=False):
    """One more workload in the same process, reported under `extra`: BASELINE config 2 (1D+t, 4x50, 160 k points), a step
    of ~0.16 ms where per-step fixed cost, not the tile loop, decides.  Same timing rules as the headline (inputs resident,
    K steps between synchronisations, whole-step time; the kernel's own time from HIP events on the engine stream)."""
    import torch
    vn, wname = build_problem(cfg)
    fd, eng = vn.fixData, vn.engine
    tdata = vn._build_tdata()
    tdata.select_mor(0)
    eng.set_weights(np.array([1.0, 1.0, 1.0]))
    nB = tdata.mor[0]['biInput'].shape[0]
    eng.train_epoch([0] * warmup, None)
    torch.cuda.synchronize()
    eng.profile_begin()
    t0 = time.perf_counter()
    eng.train_epoch([0] * steps, None)                 # optimIter's one host call per pass (VarNetUtility.py:1043-1045)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms, kl, kname = eng.profile_end()
    F_pt = 2 * (vn.inpDim * vn.layerWidth[0] + sum(a * b for a, b in zip(vn.layerWidth[:-1], vn.layerWidth[1:])) + vn.layerWidth[-1])
    flop = 6.0 * F_pt * fd.nT + 3.0 * F_pt * nB
    traffic, traffic_source = static_traffic(kname, cfg, 1)
    out = {"config": {"workload": wname, "training_points_per_step": int(fd.nT), "bc_ic_points": int(nB)},
           "value": fd.nT * steps / dt, "unit": "training-points/s", "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3,
           "roofline": {"bound": "mfma", "bound_note": "the contract's vocabulary (hbm | mfma); for nets this small the ceiling that applies is "
                                                    "issue_model.frac_of_issue_bound, not the MFMA peak",
                        "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "kernel": kname,
                        "kernel_ms": kms, "launches_timed": kl, "algorithmic_flop_per_launch": flop,
                        "achieved": flop / (kms * 1e-3) / 1e12 if kms else None,
                        "frac": flop / (kms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS if kms else None,
                        "whole_step_frac": flop / (dt / steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                        "traffic_source": traffic_source, "issue_model": issue_model(kname, 'config%d' % cfg, cal)}}
    if with_dedup:
        gb = eng.bind_grad_buffer()
        out["dedup"] = dedup_leg(vn, tdata, eng, lambda: eng.train_epoch([0], None), steps, warmup, fd.nT, nB, F_pt,
                                 lambda: float(gb[eng.P].item()))
    eng.close()
    return out


def mor_epoch_line(epochs, warmup, cal=None):
    """BASELINE config 5 under `extra`: one epoch = 6 diffusivities x 20 mini-batches, each mini-batch its own Adam step
    (VarNetUtility.py:1021-1047: batch loop inside the MOR loop), driven like VarNet.train drives it."""
    import torch
    vn, wname = build_problem(5)
    fd, eng = vn.fixData, vn.engine
    td = vn._build_tdata(batchNum=20)
    eng.set_weights(np.array([1.0, 1.0, 1.0]))
    acc = torch.zeros((), device='cuda')

    def epoch():
        for mb in range(fd.MORbatchNum):
            td.select_mor(mb)
            vn.optimIter(td, mb, acc)
    for _ in range(warmup):
        epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        epoch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / epochs
    # the kernel's own time from one more epoch with HIP events around every launch (kept out of the timed epochs: two event
    # records per 39-us step cost ~5 us of it)
    eng.profile_begin()
    epoch()
    torch.cuda.synchronize()
    kms, kl, kname = eng.profile_end()
    steps = fd.MORbatchNum * td.batchNum
    F_pt = 2 * (vn.inpDim * vn.layerWidth[0] + sum(a * b for a, b in zip(vn.layerWidth[:-1], vn.layerWidth[1:])) + vn.layerWidth[-1])
    nB = td.mor[0]['biInput'].shape[0]
    flop = (6.0 * F_pt * fd.nT + 3.0 * F_pt * nB * td.batchNum) * fd.MORbatchNum
    out = {"config": {"workload": wname, "training_points_per_epoch": int(fd.nT * fd.MORbatchNum), "adam_steps_per_epoch": int(steps),
                      "training_points_per_step": int(fd.nT // td.batchNum), "bc_ic_points": int(nB)},
           "value": fd.nT * fd.MORbatchNum / dt, "unit": "training-points/s", "epochs": epochs, "warmup": warmup,
           "ms_per_epoch": dt * 1e3, "us_per_step": dt / steps * 1e6,
           "roofline": {"bound": "mfma", "bound_note": "the contract's vocabulary (hbm | mfma); for nets this small the ceiling that applies is "
                                                    "issue_model.frac_of_issue_bound, not the MFMA peak",
                        "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "kernel": kname, "kernel_ms": kms,
                        "launches_timed": kl, "algorithmic_flop_per_launch": flop / steps,
                        "frac": (flop / steps) / (kms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS if kms else None,
                        "whole_step_frac": flop / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                        "issue_model": issue_model(kname, 'config5_minibatch', cal),
                        "note": "whole epoch incl. the host loop over mini-batches; a step is ~16 us of fixed cost (two dependent kernel "
                                "boundaries: 7.6 us floor) plus 3 tiles per workgroup"}}
    eng.close()
    return out


def main():
    try:
        _main()
    except SystemExit:
        raise
    except BaseException as e:                                  # noqa: BLE001
        # a rank that dies must leave a diagnosis on stdout (the driver keeps the tail) and a non-zero status
        import traceback
        print(json.dumps({"error": "%s: %s" % (type(e).__name__, e), "rank": int(os.environ.get('RANK', '0')),
                          "world": int(os.environ.get('WORLD_SIZE', '1')),
                          "traceback_tail": traceback.format_exc().splitlines()[-6:]}), flush=True)
        sys.exit(1)


def _main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-dedup', action='store_true', help='skip the extra de-duplicated-formulation timing')
    ap.add_argument('--no-extra', action='store_true', help='skip the calibration launches, the 2-s sustained window and the extra small-step workloads (configs 1, 2, 5)')
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves, as fresh children, BEFORE anything in this
        # process touches the GPU (this parent never imports torch); rank 0 prints the JSON line.
        from varnet_amd import launch
        # bench.py is a short job by contract: besides the launcher's bootstrap deadline it opts into the overall wall-clock limit
        # (a collective that wedges in the timed region must still end in ONE JSON line that says `timed`, not at the driver's limit)
        rc = launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                overall_s=float(os.environ.get('VN_LAUNCH_OVERALL_S', '480')))
        if rc != 0:       # the failing rank has printed its own {"error": ...} line if it got as far as Python
            rep = launch.last_report or {}
            print(json.dumps({"error": "a rank of the %d-rank launch exited with status %d (its peers were ended): %s"
                                       % (args.gpus, rc, rep.get('reason')),
                              "n_gpus": args.gpus, "last_stage": rep.get('last_stage'), "exit_status": rep.get('exit_status'),
                              "alive_when_ended": rep.get('alive')}), flush=True)
        raise SystemExit(rc)

    from varnet_amd.launch import mark_stage, rank_watchdog, IMPORTS_DONE
    mark_stage('start')
    import torch
    mark_stage(IMPORTS_DONE)
    # N > 1 under somebody else's launcher (torch.distributed.run): the rank ends itself with a one-line diagnosis
    # (its last stage) instead of sitting in a bootstrap until the caller's limit kills it silently
    # (short job by contract: the rank opts into the overall limit as well -- a wedge past the bootstrap still leaves a diagnosis)
    disarm = rank_watchdog(what='bench rank', overall_s=float(os.environ.get('VN_RANK_OVERALL_S', '420'))) if args.gpus > 1 else (lambda: None)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    ndev = torch.cuda.device_count()
    # backend 'nccl' IS RCCL on ROCm; VN_DIST_BACKEND=gloo rehearses the N>1 path with ranks sharing a GPU
    backend = os.environ.get('VN_DIST_BACKEND', 'nccl')
    if world > 1 and backend == 'nccl' and ndev < world:
        raise SystemExit('--gpus %d needs %d GPUs, %d visible (VN_DIST_BACKEND=gloo rehearses the N>1 path on fewer)'
                         % (world, world, ndev))
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    os.environ['LOCAL_RANK'] = str(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        from datetime import timedelta
        pg_timeout = timedelta(seconds=float(os.environ.get('VN_PG_TIMEOUT_S', '120')))     # torch's default is 10 minutes
        mark_stage('pg_init')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local), timeout=pg_timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=pg_timeout)

    mark_stage('build_problem')
    vn, wname = build_problem(args.config)        # joins the engine's RCCL communicator when world > 1
    mark_stage('data')
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
    in_engine = world == 1 or vn.comm == 'rccl'
    ar_ev = []

    def step():
        if in_engine:
            eng.train_epoch([0], None)            # gradient (+ RCCL SUM) + TF-1 Adam: one host call, one stream
        else:
            eng.grad(0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(gb)
            e1.record()
            ar_ev.append((e0, e1))
            eng.apply()

    mark_stage('warmup')
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ar_ev.clear()
    mark_stage('timed')
    eng.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mark_stage('report')
    comm_ms = None
    if world > 1:
        comm_ms = eng.profile_comm()[0] if in_engine else float(np.mean([a.elapsed_time(b) for a, b in ar_ev]))
    kms, klaunches, kname = eng.profile_end()
    loss_after = float(gb[P].item())
    per_rank = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "rows": int(rows_local), "kernel_ms": kms, "allreduce_ms": comm_ms})
        n_ranks = eng.comm_size()[0] if vn.comm == 'rccl' else dist.get_world_size()
    else:
        n_ranks = 1

    F_pt = 2 * (vn.inpDim * vn.layerWidth[0] + sum(a * b for a, b in zip(vn.layerWidth[:-1], vn.layerWidth[1:])) + vn.layerWidth[-1])
    cal, sustained, dd = None, None, None
    if world == 1 and not args.no_extra:
        # what this GPU sustains on the two instruction streams the kernel is priced against (vn_calib.hip, ~30 ms)
        cal = eng.calibrate()
        # ---- the same workload for >= 2 s in ONE timed window: does the 20-step rate hold under seconds of load?
        n_sus = int(max(args.steps, min(2000, np.ceil(2.2 / max(dt / args.steps, 1e-6)))))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        eng.train_epoch([0] * n_sus, None)
        torch.cuda.synchronize()
        dts = time.perf_counter() - t2
        sustained = {"steps": n_sus, "window_s": dts, "ms_per_step": dts / n_sus * 1e3, "value": nT_total * n_sus / dts,
                     "unit": "training-points/s", "ratio_to_the_%d_step_value" % args.steps: (dt / args.steps) / (dts / n_sus),
                     "note": "one timed window, one host call (vn_train_epoch), synchronised at both ends"}
    # ---- extra, separately reported: de-duplicated formulation (one network evaluation per unique
    # quadrature point; SURVEY.md 8d "honest accounting").  Never mixed into `value`.
    if not args.no_dedup and world == 1:          # the scaling runs time the headline formulation only
        dd = dedup_leg(vn, tdata, eng, step, args.steps, args.warmup, nT_total, nB, F_pt, lambda: float(gb[P].item()), args.config)

    if rank == 0:
        flop_of = lambda rows: 6.0 * F_pt * rows + 3.0 * F_pt * nB
        flop_launch = flop_of(rows_local)
        achieved = flop_launch / (kms * 1e-3) / 1e12 if kms > 0 else None
        traffic, traffic_source = static_traffic(kname, args.config, world)
        out = {
            "metric": "training-points/sec (test-funcs x quad-pts), 2D+t AD-PDE" if args.config == 3
            else "training-points/sec (test-funcs x quad-pts), 1D+t AD-PDE",
            "value": nT_total * args.steps / dt,
            "unit": "training-points/s",
            "n_gpus": int(n_ranks), "steps": args.steps, "warmup": args.warmup,
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
                         "frac": (achieved / PEAK_FP32_MFMA_TFLOPS) if achieved else None,
                         # the datasheet peak beside what a loop of independent v_mfma_f32_16x16x4_f32 sustains on this box, two
                         # waves per SIMD on all 1 024 SIMDs, measured in this process (vn_debug_calibrate)
                         "peak_measured": cal["mfma_f32_tflops"] if cal else None,
                         "frac_of_peak_measured": (achieved / cal["mfma_f32_tflops"]) if (cal and achieved) else None,
                         "calibration": cal, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "kernel": kname, "kernel_ms": kms, "launches_timed": klaunches,
                         "algorithmic_flop_per_launch": flop_launch,
                         "issue_model": issue_model(kname, 'config%d' % args.config, cal) if world == 1 else None,
                         "note": "rank 0's launch: 6*F_pt per interior point + 3*F_pt per BC/IC point, F_pt=%d; "
                                 "HIP events on the engine stream" % F_pt},
        }
        if world > 1:
            out["comm"] = {
                "backend": ("RCCL communicator inside the engine (vn_comm_init / vn_allreduce_grad)" if vn.comm == 'rccl'
                            else "torch.distributed " + dist.get_backend()),
                "ranks_reported": int(n_ranks), "payload_bytes": (P + 4) * 4,
                # '' when the engine's own communicator is up; else why all ranks kept the SUM in torch.distributed
                "in_engine_rccl_skipped_because": getattr(vn, 'comm_why', ''),
                # what summed the gradient: the version of the RCCL library this process loaded (ncclGetVersion) and the
                # (ranks, rank) the engine's communicator reports (vn_comm_size; [1, 0] when the collective is torch's)
                "rccl_version": eng.comm_version(), "vn_comm_size": list(eng.comm_size()),
                "bc_ic_weight_divisor": world,          # VarNetUtility.py:900-901: every tower is fed the whole BC/IC set
                "allreduce_ms": comm_ms,
                "note": "HIP events around the collective on the engine stream (includes waiting for the slowest rank)"}
            for r in per_rank:
                fl = flop_of(r["rows"])
                r["roofline_frac"] = (fl / (r["kernel_ms"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS) if r["kernel_ms"] else None
            out["per_rank"] = per_rank
            if backend != 'nccl':
                out["rehearsal"] = "ranks share %d GPU(s) over %s: a plumbing check, not a scaling number" % (ndev, backend)
        if dd is not None:
            out["dedup"] = dd
        inference = None
        if world == 1 and not args.no_extra:
            inference = inference_line(vn, tdata, eng, F_pt)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(vn, tdata)
        if world == 1 and args.config == 3 and not args.no_extra:
            eng.close()
            out["extra"] = {"sustained": sustained, "inference": inference,
                            "config2_small_step": small_step_line(2, 400, 40, cal, with_dedup=not args.no_dedup),
                            "config1_small_step": small_step_line(1, 1000, 100, cal),
                            "config5_mor_epoch": mor_epoch_line(5, 2, cal)}
        elif sustained is not None:
            out["extra"] = {"sustained": sustained, "inference": inference}
        print(json.dumps(out), flush=True)
    if world > 1:
        mark_stage('teardown')
        dist.barrier()
        if vn.comm == 'rccl':
            eng.comm_destroy()
        dist.destroy_process_group()
    mark_stage('done')
    disarm()
    if getattr(eng, '_comm_abandoned', False):     # a helper thread still sits in ncclCommInitRank: do not wait for it at exit
        sys.stdout.flush()
        os._exit(0)


if __name__ == '__main__':
    main()
