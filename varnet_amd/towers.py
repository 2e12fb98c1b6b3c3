"""
`processors=['GPU:0', 'GPU:1', ...]` in ONE user process.

The reference builds all its towers inside one TF-1 process (/root/reference/TFModel.py:120-165, 253-289;
`VarNet(..., processors=[...], controller=...)`, VarNet.py:201-203).  This engine runs one process per GPU, so
a `VarNet` constructed with several processors and no launcher becomes a *controller*: it forks one child per
listed GPU (before anything in the parent has touched the GPU -- a forked child cannot inherit an initialised
HIP runtime), every child builds the real `VarNet` on its GPU and joins the process group, and the public
methods (`train`, `evaluate`, `residual`, `loadModel`, `simRes`, `saveNNparam`) are forwarded to all children;
rank 0's result is returned.  `fork` is used on purpose: the PDE holds user callables (lambdas) that cannot be
pickled but are inherited by a fork.  `controller` is accepted and unused: the gradient is summed by an
all-reduce and the optimizer step is repeated on every rank, there is no controller device.
"""
import multiprocessing
import multiprocessing.connection
import os
import socket
import traceback


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tower_main(rank, world, port, conn, cls, args, kwargs, device, backend):
    try:
        os.environ.update({'RANK': str(rank), 'WORLD_SIZE': str(world), 'LOCAL_RANK': str(device),
                           'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port)})
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        import torch
        import torch.distributed as dist
        # the GPU index is validated HERE, in the tower: the controller must not ask the runtime for a device count
        # (on ROCm that call can initialise HIP/HSA without torch noticing, and a fork must not inherit that)
        if backend == 'nccl':
            if device >= torch.cuda.device_count():
                raise ValueError('requested processor GPU:%d is unavailable!' % device)      # TFModel.py:121-123
            torch.cuda.set_device(device)
        elif torch.cuda.is_available():
            # rehearsal over another backend (VN_DIST_BACKEND=gloo): the towers share the cards there are, as bench.py's
            # ranks do; a CPU-only rehearsal has no device to pick.  The wrap must reach the ENGINE: VarNet takes
            # processors[rank] = 'GPU:k' and creates its engine on device k, so this tower's entry of `processors` is rewritten
            # (['GPU:0', 'GPU:1'] on a one-GPU box used to set device 0 here and create the engine on device 1)
            device = device % torch.cuda.device_count()
            os.environ['LOCAL_RANK'] = str(device)
            torch.cuda.set_device(device)
            procs = kwargs.get('processors')
            if isinstance(procs, (list, tuple)) and len(procs) == world:
                procs = list(procs)
                procs[rank] = 'GPU:%d' % device
                kwargs = dict(kwargs, processors=procs)
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        vn = cls(*args, **kwargs)                       # inside a rank: takes its own entry of `processors`
        conn.send(('ready', None))
    except Exception as e:
        # (tag, traceback text, exception type name, message): the controller re-raises argument errors under their
        # own type, as the reference's single process would have raised them (TFModel.py:111-134)
        conn.send(('err', traceback.format_exc(), type(e).__name__, str(e)))
        return
    while True:
        msg = conn.recv()
        if msg is None:
            break
        name, a, k = msg
        try:
            r = getattr(vn, name)(*a, **k)
            if name == 'simRes' and isinstance(r, dict):
                r = {key: v for key, v in r.items() if key != 'grid'}
            conn.send(('ok', r if rank == 0 else None))
        except Exception as e:
            conn.send(('err', traceback.format_exc(), type(e).__name__, str(e)))
    try:
        if getattr(vn, 'comm', 'none') == 'rccl':
            vn.engine.comm_destroy()
        dist.destroy_process_group()
    except Exception:
        pass


class TowerGroup:
    def __init__(self, cls, args, kwargs, processors):
        import torch
        if torch.cuda.is_initialized():
            raise RuntimeError('processors=%s: the towers are forked from this process, which has already initialised '
                               'the GPU; construct the multi-GPU VarNet before any other GPU work, or start one rank '
                               'per GPU with `python -m varnet_amd.launch --gpus N script.py`' % (processors,))
        devices = []
        for p in processors:
            kind, _, idx = str(p).partition(':')
            if kind.upper() != 'GPU':
                raise ValueError('requested processor %s is unavailable!' % p)        # TFModel.py:121-123
            devices.append(int(idx or 0))
        backend = os.environ.get('VN_DIST_BACKEND', 'nccl')
        if backend == 'nccl' and len(set(devices)) != len(devices):
            raise ValueError('processors %s name a GPU twice' % (processors,))
        # (no device count here: each tower checks its own index and reports through the 'err' message)
        ctx = multiprocessing.get_context('fork')
        port = _free_port()
        self.world = len(devices)
        self.conns, self.procs = [], []
        for r, dev in enumerate(devices):
            here, there = ctx.Pipe()
            p = ctx.Process(target=_tower_main, args=(r, self.world, port, there, cls, args, kwargs, dev, backend),
                            daemon=True)
            p.start()
            there.close()
            self.conns.append(here)
            self.procs.append(p)
        self._gather()

    def _gather(self):
        """One reply per tower.  All pipes and all process sentinels are watched together: a tower that raises or dies
        while its peers sit inside a collective (the training all-reduce, the RCCL bootstrap) would otherwise leave the
        controller blocked on a healthy tower's pipe forever.  On the first error or death the remaining towers are
        terminated (exact PIDs) and the error is raised."""
        pending = dict(enumerate(self.conns))
        sentinel = {p.sentinel: r for r, p in enumerate(self.procs)}
        out, err, exc = None, None, RuntimeError
        while pending and err is None:
            ready = multiprocessing.connection.wait(list(pending.values()) + [s for s, r in sentinel.items() if r in pending])
            for obj in ready:
                if obj in sentinel:
                    r = sentinel[obj]
                    if r in pending and not pending[r].poll():            # died without a reply
                        err = 'tower %d exited (exit code %s)' % (r, self.procs[r].exitcode)
                        break
                    continue
                r = next(k for k, c in pending.items() if c is obj)
                try:
                    msg = obj.recv()
                except EOFError:
                    msg = ('err', 'tower %d exited' % r)
                tag, val = msg[0], msg[1]
                del pending[r]
                if tag == 'err':
                    err = 'tower %d failed:\n%s' % (r, val)
                    if len(msg) >= 4 and msg[2] == 'ValueError':      # a rejected argument stays a ValueError (TFModel.py:124)
                        exc, err = ValueError, '%s (tower %d)\n%s' % (msg[3], r, val)
                    break
                if r == 0:
                    out = val
        if err is not None:
            self.close(kill=True)
            raise exc(err)
        return out

    def call(self, name, *a, **k):
        for c in self.conns:
            c.send((name, a, k))
        return self._gather()

    def close(self, kill=False):
        """kill=True: a tower failed -- its peers may be blocked inside a collective and would never read the
        shutdown message, so they are terminated at once (the exact processes this group started)."""
        for c in self.conns:
            try:
                c.send(None)
            except Exception:
                pass
        for p in self.procs:
            if not kill:
                p.join(20)
            if p.is_alive():
                p.terminate()
                p.join(5)
                if p.is_alive():
                    p.kill()
        self.conns, self.procs = [], []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
