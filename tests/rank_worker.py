"""
Rank body shared by the multi-process tests (CPU: gloo + oracle-backed engine; GPU: the real
VNEngine, ranks sharing one GPU over gloo or one GPU each over RCCL).  Importable by name so that
`multiprocessing` start methods that pickle by reference (spawn / forkserver) can run it.
"""
import os

import numpy as np


def run_rank(rank, world, port, outdir, backend, engine, problem, train_kw, tag, env=None):
    os.environ.update(env or {})          # (fork-server children do not see later changes of the parent's environment)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['RANK'], os.environ['WORLD_SIZE'] = str(rank), str(world)
    if os.environ.get('VN_TEST_BREAK_RCCL_ON_RANK') == str(rank):       # this rank alone cannot load RCCL
        os.environ['VN_RCCL_LIB'] = '/nonexistent/librccl.so'
    import torch
    import torch.distributed as dist
    from tests.test_varnet_host import op1dt, op2dt
    from varnet_amd.varnet import VarNet

    torch.set_num_threads(1)
    if engine == 'hip':
        ndev = torch.cuda.device_count()
        local = rank % max(ndev, 1)
        torch.cuda.set_device(local)
        os.environ['LOCAL_RANK'] = str(local)
    else:
        from tests.oracle_engine import OracleEngine

        def make(self, processors):
            fd = self.fixData
            return OracleEngine(self.dim, self.inpDim, self.layerWidth, self.PDE.timeDependent, fd.integNum,
                                isSource=self.lossOpt['isSource'], integWflag=self.lossOpt['integWflag'],
                                learning_rate=self.learning_rate)
        VarNet._make_engine = make
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', int(os.environ['LOCAL_RANK'])))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    np.random.seed(1000 + rank)            # ranks start from DIFFERENT NumPy streams: sampling must still agree
    kind, pk = problem
    vn = (op1dt if kind == '1dt' else op2dt)(**pk)
    res = vn.train(os.path.join(outdir, '%s_w%d' % (tag, world)), **train_kw)
    theta = vn.engine.theta if engine != 'hip' else vn.engine.get_params()
    td = vn.tData
    if engine == 'hip' and train_kw.get('dedup'):
        assert td.dedup_on, 'train(dedup=True) did not engage the de-duplicated formulation on rank %d' % rank
    inp = td.mor[0]['Input']
    inp = inp.cpu().numpy() if hasattr(inp, 'cpu') else np.asarray(inp)
    np.savez(os.path.join(outdir, '%s_w%d_r%d.npz' % (tag, world, rank)), theta=np.asarray(theta, dtype=np.float64),
             loss=np.array(res.lossAll), w=np.asarray(res.trainWeight), Input=inp, comm=np.array(vn.comm),
             block=np.array(td.block(0)))
    if world > 1:
        dist.barrier()
        if vn.comm == 'rccl':
            vn.engine.comm_destroy()
        dist.destroy_process_group()


def free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(ctx, world, outdir, backend, engine, problem, train_kw, tag, timeout=600, env=None):
    """Start `world` ranks from the multiprocessing context `ctx`, wait, assert they all succeeded."""
    port = free_port()
    procs = [ctx.Process(target=run_rank, args=(r, world, port, outdir, backend, engine, problem, train_kw, tag, env))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
    bad = [p.exitcode for p in procs if p.exitcode != 0]
    for p in procs:
        if p.is_alive():
            p.terminate()
    assert not bad, 'rank exit codes %s' % [p.exitcode for p in procs]


def run_controller(outdir, backend, ngpu_entries, q):
    """The reference's single-process multi-GPU call: VarNet(..., processors=[...]) in a process that has not
    touched the GPU; the object forks its towers (varnet_amd/towers.py)."""
    try:
        os.environ['VN_DIST_BACKEND'] = backend
        import torch
        from tests.test_varnet_host import cExact, pi
        from varnet_amd import ADPDE, Domain1D, VarNet
        pde = ADPDE(Domain1D(), diff=0.1 / pi, vel=1.0, timeDependent=True, tInterval=[0, 2.0],
                    IC=lambda x: -np.sin(pi * x), cEx=cExact)               # lambdas: only a fork can carry them
        # the reference's own list, whatever the backend: over a rehearsal backend on a box with fewer cards the towers map
        # GPU:n onto the cards there are, and that wrap must reach each tower's ENGINE (varnet_amd/towers.py)
        procs = ['GPU:%d' % i for i in range(ngpu_entries)]
        vn = VarNet(pde, layerWidth=[20, 20, 20], discNum=20, bDiscNum=None, tDiscNum=30, processors=procs,
                    controller='GPU:0')
        assert not torch.cuda.is_initialized()                             # the controller never touches the GPU
        res = vn.train(os.path.join(outdir, 'ctl'), weight=[10., 10., 1.], epochNum=30, saveFreq=10, verbose=False)
        u = vn.evaluate()
        r, rv, err, ca = vn.residual()
        sim = vn.simRes(tcoord=[0.5])
        n = vn.loadModel()
        np.savez(os.path.join(outdir, 'ctl.npz'), loss=np.array(res.lossAll), u=u, err=err, r=r,
                 cApp=sim['cApp'][0], n=n, w=np.asarray(res.trainWeight))
        assert not torch.cuda.is_initialized()
        vn._towers.close()
        q.put((0, 'ok'))
    except Exception:
        import traceback
        q.put((1, traceback.format_exc()))


def run_c_host(root, outdir, q):
    """Build and run examples/c_host_step.c: a plain-C host on the C ABI, no Python / PyTorch in the process."""
    import subprocess
    exe = os.path.join(outdir, 'c_host_step')
    cc = subprocess.run(['gcc', '-O2', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I' + os.path.join(root, 'include'),
                         os.path.join(root, 'examples', 'c_host_step.c'), '-L/opt/rocm/lib', '-lamdhip64', '-ldl', '-lm',
                         '-o', exe], capture_output=True, text=True)
    if cc.returncode != 0:
        q.put((cc.returncode, 'gcc failed:\n' + cc.stderr))
        return
    env = dict(os.environ, LD_LIBRARY_PATH='/opt/rocm/lib:' + os.environ.get('LD_LIBRARY_PATH', ''))
    r = subprocess.run([exe, os.path.join(root, 'varnet_amd', 'libvarnet_hip.so')], capture_output=True, text=True,
                       env=env, timeout=300)
    q.put((r.returncode, r.stdout + '\n' + r.stderr))


class FailingTower:
    """Stand-in for VarNet inside a forked tower (tests/test_distributed_gloo.py): rank 1 raises in `train` while rank 0
    sits in a collective that can then never complete."""

    def __init__(self):
        self.rank = int(os.environ['RANK'])

    def train(self):
        import torch
        import torch.distributed as dist
        if self.rank == 1:
            raise RuntimeError('boom in tower 1')
        dist.all_reduce(torch.zeros(1))               # the peer never joins
        return 'unreachable'

    def ok(self):
        return 'pong %d' % self.rank

    def die(self):
        if self.rank == 1:
            os._exit(7)
        import torch
        import torch.distributed as dist
        dist.all_reduce(torch.zeros(1))
