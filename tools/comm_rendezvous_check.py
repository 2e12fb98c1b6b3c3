"""Bootstrap check of the in-engine RCCL communicator on a ONE-GPU box: two ranks exchange the unique id over
gloo and call vn_comm_init on the same GPU.  RCCL must answer (on both ranks) with its "Duplicate GPU detected"
usage error -- which it can only do after the two ranks have found each other through the id, i.e. the
rendezvous works.  (A real 2-rank communicator needs 2 GPUs: tests/test_multi_gpu.py::test_two_ranks_two_gpus_rccl.)
   python tools/comm_rendezvous_check.py            # parent: starts 2 ranks"""
import os, sys
sys.path.insert(0, '.')
if 'RANK' not in os.environ:
    from varnet_amd.launch import spawn_ranks
    raise SystemExit(spawn_ranks([os.path.abspath(__file__)], 2, {'NCCL_DEBUG': 'WARN'}))
import torch, torch.distributed as dist
from varnet_amd.engine import VNEngine, VNError
rank = int(os.environ['RANK'])
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=rank, world_size=2)
eng = VNEngine(1, 2, [20, 20], True, 16)
ok, why = eng.comm_init_from_torch(dist)            # collective-safe: (False, reason) on EVERY rank if any rank failed
if ok:
    print('rank %d: communicator of %s ranks created (two GPUs visible?)' % (rank, eng.comm_size()), flush=True)
    eng.comm_destroy()
else:
    print('rank %d: vn_comm_init -> %s' % (rank, why), flush=True)
dist.barrier()
dist.destroy_process_group()
