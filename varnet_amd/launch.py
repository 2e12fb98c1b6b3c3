"""
One process per GPU: start N ranks of a script on this node.

    python -m varnet_amd.launch --gpus N script.py [args ...]

The reference drives all its towers from ONE TF-1 process (`processors=['GPU:0','GPU:1']`,
/root/reference/TFModel.py:120-165, 253-289).  Here every GPU has its own process; this module is
the smallest launcher for that: it sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
and starts the ranks as fresh children.  It never touches the GPU itself (no torch import), so it is
safe to call from a parent that has not initialised HIP -- `bench.py --gpus N` uses it to launch itself.
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 script.py` is equivalent.

A first multi-GPU run that wedges must leave a diagnosis, not a kill at somebody else's limit:
  * every rank appends STAGE BREADCRUMBS (`mark_stage`) to its own file ($VN_STAGE_FILE, set here per rank);
  * `spawn_ranks` has a BOOTSTRAP DEADLINE ($VN_LAUNCH_DEADLINE_S, default 240 s): the longest a live rank may sit in ONE
    bootstrap stage (BOOTSTRAP_STAGES: up to the communicator agreement), counted from its last breadcrumb and not before
    every rank has finished its imports -- a fresh box spends 1-2 minutes paging PyTorch in -- or start +
    $VN_LAUNCH_IMPORT_GRACE_S (default 180 s), whichever comes first.  Ranks past the bootstrap run as long as they like
    (an overall limit is opt-in: $VN_LAUNCH_OVERALL_S).  When it expires the parent ends its children BY PID (terminate, then kill),
    prints ONE JSON line -- which ranks were alive, each rank's last stage -- and returns 124.  A process that has
    touched the GPU is ended and reported, never re-executed;
  * `rank_watchdog` gives the same deadline to a rank started by another launcher (torch.distributed.run): a daemon
    thread that prints the rank's own JSON line and ends the process.
"""
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
import threading
import time

IMPORTS_DONE = 'imports_done'          # the stage that starts the deadline clock
# The deadline is a BOOTSTRAP deadline: it applies to a rank whose last breadcrumb is one of these stages -- everything up to
# and including the communicator agreement, where a rank can sit in a call that has no timeout of its own -- and measures
# the time since THAT breadcrumb.  A rank past the bootstrap (comm_done, data, warmup, timed, ... or any stage a user script
# marks) is healthy however long it runs; a rank that writes no breadcrumbs at all is covered by the import grace + deadline
# counted from the start (a script that marks no stage and prepares for longer than that sets $VN_LAUNCH_DEADLINE_S=0: no bootstrap
# deadline).  An overall wall-clock limit is opt-in ($VN_LAUNCH_OVERALL_S / $VN_RANK_OVERALL_S).
BOOTSTRAP_STAGES = ('start', IMPORTS_DONE, 'pg_init', 'build_problem', 'probe', 'id_bcast', 'comm_init', 'comm_agree')


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def mark_stage(name):
    """Append a breadcrumb `<unix time> <name>` to this rank's stage file (no-op without $VN_STAGE_FILE); also kept in
    `mark_stage.last` for `rank_watchdog`."""
    mark_stage.last = name
    mark_stage.t_last = time.time()
    mark_stage.history.append(name)
    del mark_stage.history[:-32]
    f = os.environ.get('VN_STAGE_FILE')
    if not f:
        return
    try:
        with open(f, 'a') as fh:
            fh.write('%.3f %s\n' % (time.time(), name))
    except OSError:
        pass


mark_stage.last = None
mark_stage.t_last = time.time()
mark_stage.history = []


def read_stages(path):
    """[(time, stage), ...] of one rank's breadcrumb file ([] if it never wrote one)."""
    out = []
    try:
        for ln in open(path):
            t, _, s = ln.strip().partition(' ')
            if s:
                out.append((float(t), s))
    except (OSError, ValueError):
        pass
    return out


def rank_watchdog(deadline_s=None, what='rank', overall_s=None):
    """For a rank somebody else launched (torch.distributed.run): when this rank has sat in ONE bootstrap stage
    (BOOTSTRAP_STAGES) for `deadline_s` ($VN_RANK_DEADLINE_S, default 300 s) print ONE JSON line with that stage and end the
    process with status 124 (os._exit: the main thread may sit in a collective that never returns).  Past the bootstrap the
    rank may run as long as it likes unless `overall_s` ($VN_RANK_OVERALL_S; default: none) bounds the whole run.
    Returns a function that disarms it."""
    if deadline_s is None:
        deadline_s = float(os.environ.get('VN_RANK_DEADLINE_S', '300'))
    if overall_s is None:
        overall_s = float(os.environ.get('VN_RANK_OVERALL_S', '0')) or None
    done = threading.Event()
    t_arm = time.time()

    def run():
        while True:
            if done.wait(min(1.0, max(0.05, abs(deadline_s) / 4))):
                return
            now = time.time()
            in_bootstrap = mark_stage.last is None or mark_stage.last in BOOTSTRAP_STAGES
            stale = now - max(mark_stage.t_last, t_arm)
            if deadline_s > 0 and in_bootstrap and stale >= deadline_s:
                why = "%s deadline of %.0f s expired in bootstrap stage %s" % (what, deadline_s, mark_stage.last)
            elif overall_s and now - t_arm >= overall_s:
                why = "%s overall limit of %.0f s expired" % (what, overall_s)
            else:
                continue
            print(json.dumps({"error": why, "rank": int(os.environ.get('RANK', '0')),
                              "world": int(os.environ.get('WORLD_SIZE', '1')), "last_stage": mark_stage.last,
                              "s_in_last_stage": round(stale, 1)}), flush=True)
            os._exit(124)
    threading.Thread(target=run, daemon=True).start()
    return done.set


def _end(procs, grace_s=10.0):
    """terminate, then kill, the exact PIDs we started"""
    for q in procs:
        if q.poll() is None:
            q.terminate()
    t_end = time.time() + grace_s
    for q in procs:
        try:
            q.wait(max(0.1, t_end - time.time()))
        except subprocess.TimeoutExpired:
            q.kill()
            q.wait()


last_report = None      # diagnosis of the last spawn_ranks call (dict): stages per rank, exit codes, what ended it


def spawn_ranks(argv, nproc, env_extra=None, poll_s=0.2, deadline_s=None, import_grace_s=None, overall_s=None):
    """Run `sys.executable argv...` as `nproc` ranks; returns the first non-zero exit status (0 if all
    succeed, 124 if the deadline expired).  When one rank fails the others are ended (exact PIDs), so a dead rank cannot
    leave its peers blocked in a collective; when a live rank has sat in one bootstrap stage for `deadline_s` all are ended
    and ONE JSON line says who was alive and where every rank last was.  A job whose ranks are past the bootstrap is never
    ended for running long (`overall_s` / $VN_LAUNCH_OVERALL_S opts into a wall-clock limit).  `last_report` keeps the same
    diagnosis for the caller."""
    global last_report
    last_report = None
    if deadline_s is None:
        deadline_s = float(os.environ.get('VN_LAUNCH_DEADLINE_S', '240'))
    if import_grace_s is None:
        import_grace_s = float(os.environ.get('VN_LAUNCH_IMPORT_GRACE_S', '180'))
    if overall_s is None:
        overall_s = float(os.environ.get('VN_LAUNCH_OVERALL_S', '0')) or None
    port = free_port()
    procs, stage_files = [], []
    stage_dir = tempfile.mkdtemp(prefix='vn_stages_')
    for r in range(nproc):
        env = dict(os.environ)
        sf = os.path.join(stage_dir, 'rank%d.stage' % r)
        stage_files.append(sf)
        env.update({'RANK': str(r), 'LOCAL_RANK': str(r), 'WORLD_SIZE': str(nproc), 'LOCAL_WORLD_SIZE': str(nproc),
                    'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'VN_STAGE_FILE': sf})
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // nproc)))
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env))
    t_start = time.time()
    t_clock = None                       # when the deadline clock started
    status, ended_by = 0, None
    exit_codes = {}

    def report(reason):
        stages = [read_stages(f) for f in stage_files]
        return {"reason": reason, "ranks": nproc, "deadline_s": deadline_s, "elapsed_s": round(time.time() - t_start, 1),
                "alive": [r for r, p in enumerate(procs) if p.poll() is None],
                "exit_status": {str(r): rc for r, rc in sorted(exit_codes.items())},
                "last_stage": {str(r): (st[-1][1] if st else None) for r, st in enumerate(stages)},
                "s_in_last_stage": {str(r): (round(time.time() - st[-1][0], 1) if st else None) for r, st in enumerate(stages)}}
    try:
        live = list(procs)
        while live:
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                exit_codes[procs.index(p)] = rc
                if rc != 0 and status == 0:
                    status, ended_by = rc, 'rank %d exited with status %d' % (procs.index(p), rc)
                    last_report = report(ended_by)
                    _end(live)                           # do not leave the peers blocked in a collective
                    live = []
                    break
            if not live:
                break
            now = time.time()
            if t_clock is None:
                if now - t_start >= import_grace_s or all(any(s == IMPORTS_DONE for _, s in read_stages(f)) for f in stage_files):
                    t_clock = now
            else:
                # staleness of the live ranks that are still in their bootstrap: seconds since the rank's last breadcrumb
                # (since the clock started for a rank whose last breadcrumb is older, or that never wrote one)
                wedged = []
                for r, p in enumerate(procs):
                    if p.poll() is not None:
                        continue
                    st = read_stages(stage_files[r])
                    if st and st[-1][1] not in BOOTSTRAP_STAGES:
                        continue
                    if deadline_s > 0 and now - max(st[-1][0] if st else t_clock, t_clock) >= deadline_s:     # (<= 0: no bootstrap deadline)
                        wedged.append(r)
                if wedged:
                    status, ended_by = 124, 'launch deadline of %.0f s expired (rank%s %s in one bootstrap stage that long)' % (
                        deadline_s, 's' if len(wedged) > 1 else '', ', '.join(map(str, wedged)))
                elif overall_s and now - t_start >= overall_s:
                    status, ended_by = 124, 'launch overall limit of %.0f s expired' % overall_s
            if status == 124:
                last_report = report(ended_by)
                _end(live)
                live = []
                js = dict(last_report)
                js["error"] = "%s: the ranks still alive were ended by PID" % ended_by
                print(json.dumps(js), flush=True)
                break
            time.sleep(poll_s)
    except KeyboardInterrupt:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGINT)
        status = 130
    if last_report is None or status == 0:
        last_report = report(ended_by or 'all ranks exited with status 0')
    for f in stage_files:
        try:
            os.remove(f)
        except OSError:
            pass
    try:
        os.rmdir(stage_dir)
    except OSError:
        pass
    return status


def main():
    a = sys.argv[1:]
    if len(a) < 3 or a[0] != '--gpus':
        raise SystemExit('usage: python -m varnet_amd.launch --gpus N script.py [args ...]')
    raise SystemExit(spawn_ranks(a[2:], int(a[1])))


if __name__ == '__main__':
    main()
