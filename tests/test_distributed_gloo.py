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
    res = vn.train(os.path.join(outdir, 'w%d' % world), weight=[10., 10., 1.], epochNum=3, saveFreq=2,
                   verbose=False, batchNum=batchNum)
    sim = vn.simRes(tcoord=[0.3], plot=False)                # every rank: the monitors it calls are rank-local
    if rank == 0:
        np.savez(os.path.join(outdir, 'out_w%d_b%s.npz' % (world, batchNum)),
                 theta=vn.engine.theta, loss=np.array(res.lossAll), w=res.trainWeight,
                 lossVec=np.asarray(res.lossVec[0]), lossField=np.asarray(sim['lossField'][0]))
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
    # the recorded weights are per feed: BC/IC entries divided by batchNum * puNum (VarNetUtility.py:900-901)
    np.testing.assert_allclose(b['w'] * np.array([2.0, 2.0, 1.0]), a['w'], rtol=1e-10)
    # the loss field of a monitor is the towers' fields concatenated in tower order, mini-batch by mini-batch
    # (TFModel.py:319, VarNetUtility.py:1085-1090): one entry per test function, whatever the world size
    assert b['lossVec'].shape == a['lossVec'].shape == (35, 1)
    if batchNum is None:
        # identical partition of the sum -> same trajectory up to fp64 summation order
        np.testing.assert_allclose(b['loss'], a['loss'], rtol=1e-9)
        np.testing.assert_allclose(b['theta'], a['theta'], rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(b['lossVec'], a['lossVec'], rtol=1e-6, atol=1e-14)
        # ... and simRes interpolates it under towers as it does on one processor (VarNet.py:2043-2047)
        assert np.abs(a['lossField']).max() > 0
        np.testing.assert_allclose(b['lossField'], a['lossField'], rtol=1e-6, atol=1e-12 * np.abs(a['lossField']).max())
    else:
        # with mini-batches the (batch, tower) blocks differ between world sizes
        # (block j = bi*puNum + rank), so only the first epoch's first loss is comparable in size
        assert np.isfinite(b['loss']).all() and b['loss'][-1] < b['loss'][0] * 1.5


def test_world4_uneven_blocks_match_world1(tmp_path):
    """Four towers over 35 test functions: blocks of 9, 9, 9, 8 (batchLen = ceil(nt / puNum), VarNetUtility.py:825-838);
    the rehearsal of more ranks the one-GPU boxes allow."""
    out = str(tmp_path)
    _run(0, 1, _free_port(), out, None)
    mp.spawn(_run, args=(4, _free_port(), out, None), nprocs=4, join=True)
    a = np.load(os.path.join(out, 'out_w1_bNone.npz'))
    b = np.load(os.path.join(out, 'out_w4_bNone.npz'))
    np.testing.assert_allclose(b['w'] * np.array([4.0, 4.0, 1.0]), a['w'], rtol=1e-10)
    np.testing.assert_allclose(b['loss'], a['loss'], rtol=1e-9)
    np.testing.assert_allclose(b['theta'], a['theta'], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(b['lossVec'], a['lossVec'], rtol=1e-6, atol=1e-14)     # ragged / empty blocks gathered in tower order
    np.testing.assert_allclose(b['lossField'], a['lossField'], rtol=1e-6, atol=1e-12 * np.abs(a['lossField']).max())


def test_world8_uneven_blocks_match_world1(tmp_path):
    """The driver's SCALE run ends at 8 ranks; the one-GPU boxes admit at most 6 processes on a card, so the 8-rank case
    is rehearsed here on the CPU engine: eight towers over 35 test functions = blocks of 5 x 7 and an EMPTY eighth tower
    (batchLen = ceil(35 / 8) = 5, VarNetUtility.py:825-838), BC/IC weights / 8 (:900-901), one SUM all-reduce per step."""
    out = str(tmp_path)
    _run(0, 1, _free_port(), out, None)
    mp.spawn(_run, args=(8, _free_port(), out, None), nprocs=8, join=True)
    a = np.load(os.path.join(out, 'out_w1_bNone.npz'))
    b = np.load(os.path.join(out, 'out_w8_bNone.npz'))
    np.testing.assert_allclose(b['w'] * np.array([8.0, 8.0, 1.0]), a['w'], rtol=1e-10)
    np.testing.assert_allclose(b['loss'], a['loss'], rtol=1e-9)
    np.testing.assert_allclose(b['theta'], a['theta'], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(b['lossVec'], a['lossVec'], rtol=1e-6, atol=1e-14)     # ragged / empty blocks gathered in tower order
    np.testing.assert_allclose(b['lossField'], a['lossField'], rtol=1e-6, atol=1e-12 * np.abs(a['lossField']).max())


# ---- cases the round-1 review asked for ------------------------------------------------------------
import multiprocessing
from tests import rank_worker as rw


def _spawn_ctx():
    return multiprocessing.get_context('spawn')


def test_world2_with_an_empty_shard(tmp_path):
    """nt < batchLen * world: rank 1's tower block is empty (the reference slices past the end,
    VarNetUtility.py:830-838).  The empty rank must still join every collective, and the run must equal
    the one-process run."""
    out = str(tmp_path)
    prob = ('1dt', dict(layerWidth=[6, 5], discNum=5, tDiscNum=7))            # nt = 35
    kw = dict(weight=[10., 10., 1.], epochNum=3, saveFreq=100, verbose=False, batchLen=35)
    rw.launch(_spawn_ctx(), 1, out, 'gloo', 'oracle', prob, kw, 'e')
    rw.launch(_spawn_ctx(), 2, out, 'gloo', 'oracle', prob, kw, 'e')
    a = np.load(os.path.join(out, 'e_w1_r0.npz'))
    b0 = np.load(os.path.join(out, 'e_w2_r0.npz'))
    b1 = np.load(os.path.join(out, 'e_w2_r1.npz'))
    assert list(b1['block']) == [35, 35] and list(b0['block']) == [0, 35]
    np.testing.assert_allclose(b0['loss'], a['loss'], rtol=1e-9)
    np.testing.assert_allclose(b0['theta'], a['theta'], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(b1['theta'], b0['theta'], rtol=0, atol=0)


def test_world1_with_an_empty_minibatch(tmp_path):
    """nt = 9, batchNum = 4 -> batchLen 3, the 4th mini-batch of a 2-tower run is empty for both towers."""
    out = str(tmp_path)
    prob = ('1dt', dict(layerWidth=[4], discNum=3, tDiscNum=3))               # nt = 9
    kw = dict(weight=[10., 10., 1.], epochNum=2, saveFreq=100, verbose=False, batchNum=4)
    rw.launch(_spawn_ctx(), 2, out, 'gloo', 'oracle', prob, kw, 'm')
    b0 = np.load(os.path.join(out, 'm_w2_r0.npz'))
    assert np.isfinite(b0['loss']).all()


def test_world2_optimal_sampling_draws_one_training_set(tmp_path):
    """smpScheme='optimal' at world 2: every rank must hold the SAME re-drawn training set (the reference
    samples once and slices per tower) although the ranks' NumPy streams start differently."""
    out = str(tmp_path)
    prob = ('1dt', dict(layerWidth=[6], discNum=5, tDiscNum=6))
    kw = dict(weight=[10., 10., 1.], smpScheme='optimal', epochNum=6, saveFreq=1, verbose=False, trainUpdelay=2,
              tolUpd=1e9, frac=0.5)
    rw.launch(_spawn_ctx(), 2, out, 'gloo', 'oracle', prob, kw, 'o')
    b0 = np.load(os.path.join(out, 'o_w2_r0.npz'))
    b1 = np.load(os.path.join(out, 'o_w2_r1.npz'))
    assert b0['Input'].shape[0] > 5 * 6 * 16                                  # the set was re-drawn (points added)
    np.testing.assert_array_equal(b0['Input'], b1['Input'])
    np.testing.assert_allclose(b1['theta'], b0['theta'], rtol=0, atol=0)
    np.testing.assert_allclose(b1['w'], b0['w'], rtol=1e-12)


def _tower_controller(q, method):
    """Runs in a fresh process (the controller forks its towers)."""
    import time
    os.environ['VN_DIST_BACKEND'] = 'gloo'
    from varnet_amd.towers import TowerGroup
    try:
        grp = TowerGroup(rw.FailingTower, (), {}, ['GPU:0', 'GPU:1'])
        assert grp.call('ok') == 'pong 0'
        pids = [p.pid for p in grp.procs]
        t0 = time.time()
        try:
            grp.call(method)
            q.put((1, 'no error raised'))
            return
        except RuntimeError as e:
            msg = str(e)
        dt = time.time() - t0
        time.sleep(0.5)
        alive = []
        for pid in pids:
            try:
                os.kill(pid, 0)
                alive.append(pid)
            except OSError:
                pass
        q.put((0, (msg, dt, alive)))
    except Exception:
        import traceback
        q.put((1, traceback.format_exc()))


@pytest.mark.parametrize('method,needle', [('train', 'boom in tower 1'), ('die', 'tower 1 exited')])
def test_failed_tower_does_not_wedge_the_controller(method, needle):
    """a tower that raises (or dies) while its peer is inside a collective must surface as an error in the
    controller within seconds, and the surviving tower must be terminated -- not block on the healthy tower's pipe."""
    ctx = _spawn_ctx()
    q = ctx.Queue()
    p = ctx.Process(target=_tower_controller, args=(q, method))
    p.start()
    rc, val = q.get(timeout=120)
    p.join(30)
    assert rc == 0, val
    msg, dt, alive = val
    assert needle in msg and dt < 30 and alive == [], (msg, dt, alive)


def _tower_bad_processor(q):
    os.environ['VN_DIST_BACKEND'] = 'nccl'               # the reference's case: real devices are asked for
    from varnet_amd.towers import TowerGroup
    try:
        TowerGroup(rw.FailingTower, (), {}, ['GPU:0', 'GPU:97'])
        q.put((1, 'no error raised'))
    except ValueError as e:
        q.put((0, str(e)))
    except Exception as e:                                # noqa: BLE001
        q.put((1, '%s: %s' % (type(e).__name__, e)))


def test_unavailable_processor_is_a_value_error_in_the_controller():
    """the device check lives in the forked tower (the controller must not initialise the GPU), but the
    reference raises ValueError('requested processor ... is unavailable!') for it (TFModel.py:121-124): the controller
    re-raises a tower's ValueError under its own type instead of wrapping it into RuntimeError."""
    ctx = _spawn_ctx()
    q = ctx.Queue()
    p = ctx.Process(target=_tower_bad_processor, args=(q,))
    p.start()
    rc, val = q.get(timeout=120)
    p.join(30)
    assert rc == 0, val
    assert 'requested processor GPU:' in val and 'is unavailable!' in val
