"""
One process per GPU: start N ranks of a script on this node.

    python -m varnet_amd.launch --gpus N script.py [args ...]

The reference drives all its towers from ONE TF-1 process (`processors=['GPU:0','GPU:1']`,
/root/reference/TFModel.py:120-165, 253-289).  Here every GPU has its own process; this module is
the smallest launcher for that: it sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
and starts the ranks as fresh children.  It never touches the GPU itself (no torch import), so it is
safe to call from a parent that has not initialised HIP -- `bench.py --gpus N` uses it to launch itself.
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 script.py` is equivalent.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(argv, nproc, env_extra=None, poll_s=0.2):
    """Run `sys.executable argv...` as `nproc` ranks; returns the first non-zero exit status (0 if all
    succeed).  When one rank fails the others are ended (exact PIDs), so a dead rank cannot leave its
    peers blocked in a collective."""
    port = free_port()
    procs = []
    for r in range(nproc):
        env = dict(os.environ)
        env.update({'RANK': str(r), 'LOCAL_RANK': str(r), 'WORLD_SIZE': str(nproc), 'LOCAL_WORLD_SIZE': str(nproc),
                    'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port)})
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // nproc)))
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env))
    status = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0 and status == 0:
                    status = rc
                    for q in live:                       # do not leave the peers blocked in a collective
                        q.terminate()
                    deadline = time.time() + 10
                    for q in live:
                        try:
                            q.wait(max(0.1, deadline - time.time()))
                        except subprocess.TimeoutExpired:
                            q.kill()
                    live = []
                    break
            time.sleep(poll_s)
    except KeyboardInterrupt:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGINT)
        status = 130
    return status


def main():
    a = sys.argv[1:]
    if len(a) < 3 or a[0] != '--gpus':
        raise SystemExit('usage: python -m varnet_amd.launch --gpus N script.py [args ...]')
    raise SystemExit(spawn_ranks(a[2:], int(a[1])))


if __name__ == '__main__':
    main()
