"""
N>1 path on CPU: world_size-2 `gloo` run of the VarNet training loop (oracle-backed test
engine) must reproduce the single-process trajectory: contiguous test-function shards per rank
(VarNetUtility.py:830-838), BC/IC replicated with weights / puNum (VarNetUtility.py:900-901),
gradient SUM (TFModel.py:370).
"""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_varnet_host import op1dt
from tests.oracle_engine import OracleEngine
from varnet_amd.varnet import VarNet


def _patch():
    def make(self, processors):
        fd = self.fixData
        return OracleEngine(self.dim, self.inpDim, self.layerWidth, self.PDE.timeDependent, fd.integNum,
                            isSource=self.lossOpt['isSource'], integWflag=self.lossOpt['integWflag'],
                            learning_rate=self.learning_rate)
    VarNet._make_engine = make


def _run(rank, world, port, outdir, batchNum):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    _patch()
    vn = op1dt(layerWidth=[6, 5], discNum=5, tDiscNum=7)
    res = vn.train(os.path.join(outdir, 'w%d' % world), weight=[10., 10., 1.], epochNum=3, saveFreq=100,
                   verbose=False, batchNum=batchNum)
    if rank == 0:
        np.savez(os.path.join(outdir, 'out_w%d_b%s.npz' % (world, batchNum)),
                 theta=vn.engine.theta, loss=np.array(res.lossAll), w=res.trainWeight)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('batchNum', [None, 2])
def test_world2_matches_world1(tmp_path, batchNum):
    out = str(tmp_path)
    _run(0, 1, _free_port(), out, batchNum)
    mp.spawn(_run, args=(2, _free_port(), out, batchNum), nprocs=2, join=True)
    a = np.load(os.path.join(out, 'out_w1_b%s.npz' % batchNum))
    b = np.load(os.path.join(out, 'out_w2_b%s.npz' % batchNum))
    np.testing.assert_allclose(b['w'], a['w'], rtol=1e-10)
    if batchNum is None:
        # identical partition of the sum -> same trajectory up to fp64 summation order
        np.testing.assert_allclose(b['loss'], a['loss'], rtol=1e-9)
        np.testing.assert_allclose(b['theta'], a['theta'], rtol=1e-7, atol=1e-10)
    else:
        # with mini-batches the (batch, tower) blocks differ between world sizes
        # (block j = bi*puNum + rank), so only the first epoch's first loss is comparable in size
        assert np.isfinite(b['loss']).all() and b['loss'][-1] < b['loss'][0] * 1.5
