"""Rank bodies for the CPU-tier launcher / bootstrap tests (tests/test_bench_cpu.py); started by varnet_amd.launch.spawn_ranks.

    launch_worker.py stuck          every rank joins a gloo group; rank 1 then sleeps forever in stage `comm_init`, rank 0 waits
                                    for it in a collective: what a wedged ncclCommInitRank looks like from outside
    launch_worker.py watchdog       one rank under rank_watchdog(1 s) that never finishes
    launch_worker.py bootstrap KIND VNEngine.comm_init_from_torch over gloo on a stand-in engine whose probes are scripted:
                                    KIND = distinct  the ranks share ORDINAL 0 but sit on different hosts / devices
                                    KIND = same      both ranks name the same (host, device)
                                    KIND = wedged    rank 1's comm_init never returns (a wedged ncclCommInitRank)
                                    prints {"rank":, "ok":, "why":} per rank
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from varnet_amd.launch import IMPORTS_DONE, mark_stage, rank_watchdog  # noqa: E402

mode = sys.argv[1]
rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
mark_stage('start')

if mode == 'watchdog':
    rank_watchdog(1.0, what='test rank')
    mark_stage('comm_init')
    time.sleep(600)
    sys.exit(0)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
mark_stage(IMPORTS_DONE)
mark_stage('pg_init')
dist.init_process_group('gloo', rank=rank, world_size=world)

if mode == 'stuck':
    mark_stage('probe')
    mark_stage('id_bcast')
    mark_stage('comm_init')
    if rank == 1:
        time.sleep(600)
    dist.barrier()
    sys.exit(0)

if mode == 'long_healthy':
    # a job that is past its bootstrap and simply runs longer than the launcher's deadline
    mark_stage('comm_done')
    mark_stage('timed')
    time.sleep(float(sys.argv[2]))
    dist.barrier()
    mark_stage('done')
    sys.exit(0)

if mode == 'step_fail':
    # rank 1's engine call fails inside a training step (VNEngine._ck) while rank 0 waits for it in the step's collective:
    # what a one-rank vn_grad failure under a communicator looks like from outside (include/varnet_hip.h)
    from varnet_amd.engine import VNEngine, VNError

    class Lib:
        @staticmethod
        def vn_last_error():
            return b'batch 7 was never registered'

    class StandIn:
        lib = Lib()
    mark_stage('comm_done')
    mark_stage('timed')
    if rank == 1:
        try:
            VNEngine._ck(StandIn(), 1)
        except VNError as e:
            print(json.dumps({"error": str(e), "rank": rank}), flush=True)
            sys.exit(3)
    # rank 0 "sits in the step's all-reduce": over RCCL a collective whose peer is gone does not return (gloo would raise on the closed
    # socket and race the launcher's poll), so the blocked rank is modelled by a sleep -- the launcher must end it by PID
    time.sleep(600)
    sys.exit(0)

if mode == 'bootstrap':
    from varnet_amd.engine import VNEngine
    kind = sys.argv[2]

    class Dev:
        index = 0                                     # every rank sees its card as cuda:0 (per-rank visibility mask)

    class Lib:
        @staticmethod
        def vn_last_error():
            return b''

    class StandIn:
        """the attributes comm_init_from_torch touches, scripted"""
        torch, device, lib = torch, Dev(), Lib()
        inited = None
        shared_gpus = staticmethod(VNEngine.shared_gpus)

        def comm_available(self):
            return True

        def comm_size(self):
            return (1, 0)

        def _make_current(self, ordinal):
            assert ordinal == 0

        def _gpu_identity(self, ordinal):
            # (host, identifiers, visibility mask, ordinal): per-rank masks, every rank sees its card as ordinal 0
            if kind == 'distinct':
                return ('node0', 'uuid:GPU-%04d|pci:0:%x:0' % (rank, 0x5a + rank), '%d|' % rank, 0)
            if kind == 'wedged':
                return ('node0', 'uuid:GPU-%04d|pci:0:%x:0' % (rank, 0x5a + rank), '0,1|', rank)
            return ('node0', 'uuid:GPU-0000|pci:0:5a:0', '%s|' % ('0' if rank == 0 else '0,1'), 0)

        def comm_unique_id(self):
            return bytes(range(128))

        destroyed = False

        def comm_init(self, r, w, uid):
            assert uid == bytes(range(128))
            if kind == 'wedged' and r == 1:
                time.sleep(600)
            self.inited = (r, w)

        def comm_destroy(self):
            self.inited = None
            self.destroyed = True

        abandon_calls = 0

        def comm_abandon(self, pending=None):         # VNEngine.comm_abandon: vn_comm_abandon + the exit guard
            self.abandon_calls += 1
            self._comm_abandoned = True

    eng = StandIn()
    ok, why = VNEngine.comm_init_from_torch(eng, dist)
    assert mark_stage.last == 'comm_done'            # whatever the outcome: past the launcher's bootstrap deadline
    print(json.dumps({"rank": rank, "ok": bool(ok), "why": why, "inited": eng.inited, "last_stage": mark_stage.history[-2],
                      "abandoned": bool(getattr(eng, '_comm_abandoned', False)), "destroyed": eng.destroyed,
                      "abandon_calls": eng.abandon_calls}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.flush()
    os._exit(0)                # (a helper thread may still sit in the scripted wedge)

raise SystemExit('unknown mode ' + mode)
