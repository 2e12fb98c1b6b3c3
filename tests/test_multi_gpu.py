"""
N > 1 path on the real HIP engine (the twin of tests/test_distributed_gloo.py, which runs the same
host logic on the CPU oracle engine):

  * in-engine RCCL communicator through the C ABI at world size 1 (vn_comm_unique_id / vn_comm_init /
    vn_allreduce_grad / vn_train_epoch with a communicator) -- runs on any GPU box;
  * two ranks sharing ONE GPU over gloo: contiguous test-function shards, BC/IC replicated with
    weights / puNum, gradient SUM, must reproduce the one-rank trajectory to fp32 rounding -- runs on any GPU
    box;
  * two ranks on two GPUs over RCCL (`nccl` backend + the engine's own communicator) -- skipped unless
    the box has >= 2 GPUs.
Ranks are started from the fork server created in conftest.py before this process touched the GPU.
"""
import os

import numpy as np
import pytest

from tests import conftest
from tests import rank_worker as rw

pytestmark = pytest.mark.gpu


def _problem():
    return ('1dt', dict(layerWidth=[20, 20, 20], discNum=20, tDiscNum=30))      # 600 test functions x 16


def _ndev():
    import torch
    return torch.cuda.device_count()


def test_in_engine_rccl_world1_matches_plain_step():
    """C ABI: a communicator of one rank.  gradient -> RCCL all-reduce -> Adam must equal the plain step."""
    import torch
    from tests.test_engine_gpu import synth
    from varnet_amd.engine import VNEngine
    d = synth(5, 3, 2, [50] * 5, 64, 40, 60, 25)
    outs = []
    for use_comm in (False, True):
        eng = VNEngine(2, 3, [50] * 5, True, 64)
        eng.init_params(seed=3)
        eng.set_fe_table(d['N1'], d['dNt1'], None)
        eng.set_interior(0, d['Input'], d['gcoef'], None, n_k=40, detJ=d['detJ'])
        eng.set_bic(d['biInput'], d['biLabel'], 25, 2.0)
        eng.set_weights(d['w'])
        if use_comm:
            uid = VNEngine.comm_unique_id()
            assert len(uid) == 128
            eng.comm_init(0, 1, uid)
            assert eng.comm_size() == (1, 0)
            with pytest.raises(Exception):
                eng.comm_init(0, 1, uid)                     # already initialised
        acc = torch.zeros((), dtype=torch.float32, device='cuda')
        for _ in range(5):
            eng.train_epoch([0], acc)
        if use_comm:
            eng.bind_grad_buffer()
            eng.grad(0)
            g0 = eng.gradbuf.clone()
            eng.allreduce_grad()                             # SUM over one rank: identity
            torch.cuda.synchronize()
            assert torch.equal(g0, eng.gradbuf)
            eng.comm_destroy()
            assert eng.comm_size() == (1, 0)
        torch.cuda.synchronize()
        outs.append((eng.get_params(), float(acc.item()), eng.step))
        eng.close()
    (t0, l0, s0), (t1, l1, s1) = outs
    assert s0 == s1 == 5
    np.testing.assert_allclose(l1, l0, rtol=1e-6)
    np.testing.assert_allclose(t1, t0, rtol=1e-6, atol=1e-7)


def _compare(out, tag, rtol_loss, atol_theta):
    a = np.load(os.path.join(out, '%s_w1_r0.npz' % tag))
    b0 = np.load(os.path.join(out, '%s_w2_r0.npz' % tag))
    b1 = np.load(os.path.join(out, '%s_w2_r1.npz' % tag))
    # recorded weights are per feed: BC/IC entries divided by batchNum * puNum (VarNetUtility.py:900-901)
    np.testing.assert_allclose(b0['w'] * np.array([2.0, 2.0, 1.0]), a['w'], rtol=1e-4)
    np.testing.assert_allclose(b0['loss'], a['loss'], rtol=rtol_loss)
    np.testing.assert_allclose(b0['theta'], a['theta'], rtol=0, atol=atol_theta)
    np.testing.assert_array_equal(b1['theta'], b0['theta'])          # replicas stay bitwise in sync
    return a, b0, b1


@pytest.mark.parametrize('batchNum', [None])
def test_two_ranks_share_one_gpu_gloo(tmp_path, batchNum):
    """Real VNEngine, world 2, gloo (both ranks on GPU 0): vn_grad -> all_reduce -> vn_apply reproduces
    the one-rank run (fp32 kernels: the shard partial sums are added in a different order)."""
    if conftest.FORKSERVER is None:
        pytest.skip('no fork server')
    out = str(tmp_path)
    kw = dict(weight=[10., 10., 1.], epochNum=30, saveFreq=1000, verbose=False, batchNum=batchNum)
    rw.launch(conftest.FORKSERVER, 1, out, 'gloo', 'hip', _problem(), kw, 'g')
    rw.launch(conftest.FORKSERVER, 2, out, 'gloo', 'hip', _problem(), kw, 'g')
    a, b0, b1 = _compare(out, 'g', 2e-4, 2e-4)
    assert str(b0['comm']) == 'torch' and list(b0['block']) == [0, 300] and list(b1['block']) == [300, 600]


def _bootstrap_world1(q):
    """forked child: a one-rank gloo group, then the whole bootstrap VarNet runs at world > 1 -- probes, id broadcast, the REAL
    ncclCommInitRank on its helper thread, agreement -- and training steps through the communicator made there"""
    try:
        import socket
        import torch
        import torch.distributed as dist
        from tests.test_engine_gpu import synth
        from varnet_amd.engine import VNEngine
        sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
        dist.init_process_group('gloo', rank=0, world_size=1, init_method='tcp://127.0.0.1:%d' % port)
        d = synth(5, 3, 2, [50] * 5, 64, 40, 60, 25)
        outs = []
        for use_comm in (False, True):
            eng = VNEngine(2, 3, [50] * 5, True, 64)
            eng.init_params(seed=3)
            eng.set_fe_table(d['N1'], d['dNt1'], None)
            eng.set_interior(0, d['Input'], d['gcoef'], None, n_k=40, detJ=d['detJ'])
            eng.set_bic(d['biInput'], d['biLabel'], 25, 2.0)
            eng.set_weights(d['w'])
            if use_comm:
                ok, why = eng.comm_init_from_torch(dist)
                assert ok and why == '', why
                assert eng.comm_size() == (1, 0) and not getattr(eng, '_comm_abandoned', False)
            acc = torch.zeros((), dtype=torch.float32, device='cuda')
            for _ in range(5):
                eng.train_epoch([0], acc)                    # main thread drives the communicator the helper thread created
            torch.cuda.synchronize()
            if use_comm:
                eng.comm_destroy()
            outs.append((eng.get_params(), float(acc.item())))
            eng.close()
        dist.destroy_process_group()
        np.testing.assert_allclose(outs[1][1], outs[0][1], rtol=1e-6)
        np.testing.assert_allclose(outs[1][0], outs[0][0], rtol=1e-6, atol=1e-7)
        q.put((0, 'ok'))
    except Exception:
        import traceback
        q.put((1, traceback.format_exc()))


def test_bootstrap_through_torch_with_the_real_rccl_world1():
    """VNEngine.comm_init_from_torch end to end against the REAL library (a communicator of one rank is the most a one-GPU box
    allows): since round 5 ncclCommInitRank runs on a helper thread with a timeout; the communicator it creates must serve the
    main thread's training steps."""
    if conftest.FORKSERVER is None:
        pytest.skip('no fork server')
    from varnet_amd.engine import VNEngine
    if not VNEngine.comm_available():
        pytest.skip('RCCL cannot be loaded on this box')
    q = conftest.FORKSERVER.Queue()
    p = conftest.FORKSERVER.Process(target=_bootstrap_world1, args=(q,))
    p.start()
    rc, text = q.get(timeout=300)
    p.join(60)
    assert rc == 0, text


def _abandon_world1(q):
    """forked child: the C layer of an abandoned communicator"""
    try:
        import torch
        from tests.test_engine_gpu import synth
        from varnet_amd.engine import VNEngine, VNError
        d = synth(5, 3, 2, [50] * 5, 64, 40, 60, 25)

        def make():
            eng = VNEngine(2, 3, [50] * 5, True, 64)
            eng.init_params(seed=3)
            eng.set_fe_table(d['N1'], d['dNt1'], None)
            eng.set_interior(0, d['Input'], d['gcoef'], None, n_k=40, detJ=d['detJ'])
            eng.set_bic(d['biInput'], d['biLabel'], 25, 2.0)
            eng.set_weights(d['w'])
            return eng

        def train(eng):
            acc = torch.zeros((), dtype=torch.float32, device='cuda')
            for _ in range(4):
                eng.train_epoch([0], acc)
            torch.cuda.synchronize()
            return eng.get_params(), float(acc.item())
        plain = make()
        ref = train(plain)
        plain.close()
        # (a) a communicator that had come up is withdrawn: the steps after run WITHOUT a collective, bit for bit the plain steps
        a = make()
        a.comm_init(0, 1, a.comm_unique_id())
        assert a.comm_size() == (1, 0)
        # a step whose gradient fails on this rank (batch never registered) returns BEFORE the collective and leaves the
        # optimizer state as it found it (include/varnet_hip.h, "Failure under a communicator")
        step0 = a.step
        a.profile_begin()
        with pytest.raises(VNError):
            a.train_step(7, None)
        with pytest.raises(VNError):
            a.train_epoch([0, 7])                    # fails at its SECOND step: the first one counted
        assert a.step == step0 + 1 and a.profile_comm()[1] == 1
        a.init_params(seed=3)                        # back to the plain engine's start (parameters, Adam slots, step 0)
        assert a.step == 0
        assert a.lib.vn_comm_abandon(a.h) == 0 and a.lib.vn_comm_abandon(a.h) == 0        # idempotent
        a.profile_begin()
        got = train(a)
        assert a.profile_comm()[1] == 0, 'a collective was enqueued on an abandoned communicator'
        np.testing.assert_array_equal(got[0], ref[0])
        with pytest.raises(VNError, match='abandoned'):
            a.comm_init(0, 1, a.comm_unique_id())
        with pytest.raises(VNError, match='no communicator'):
            a.allreduce_grad()
        # (b) vn_comm_init that returns AFTER the abandonment (the helper thread of a bootstrap that timed out) commits nothing
        b = make()
        uid = b.comm_unique_id()
        assert b.lib.vn_comm_abandon(b.h) == 0
        with pytest.raises(VNError, match='abandoned'):
            b.comm_init(0, 1, uid)
        assert b.comm_size() == (1, 0)
        b.profile_begin()
        got = train(b)
        assert b.profile_comm()[1] == 0
        np.testing.assert_array_equal(got[0], ref[0])
        # (handles of abandoned engines are left to the process exit, as VNEngine.close does after comm_abandon)
        q.put((0, 'ok'))
    except Exception:
        import traceback
        q.put((1, traceback.format_exc()))


def test_abandoned_communicator_is_never_used_by_the_engine():
    """after the bootstrap gives up on ncclCommInitRank, the engine itself must ignore the communicator --
    vn_train_step / vn_train_epoch key on it -- whether it had come up or comes up later on the helper thread."""
    if conftest.FORKSERVER is None:
        pytest.skip('no fork server')
    from varnet_amd.engine import VNEngine
    if not VNEngine.comm_available():
        pytest.skip('RCCL cannot be loaded on this box')
    q = conftest.FORKSERVER.Queue()
    p = conftest.FORKSERVER.Process(target=_abandon_world1, args=(q,))
    p.start()
    rc, text = q.get(timeout=300)
    p.join(60)
    assert rc == 0, text


def test_two_ranks_dedup_matches_one_rank_gloo(tmp_path):
    """train(dedup=True) under towers: every rank de-duplicates its own contiguous block of test functions (points on the
    seam between the blocks are evaluated once per rank), the gradient SUM is unchanged -- world 2 reproduces the one-rank
    de-duplicated run, and that run the row-wise one."""
    if conftest.FORKSERVER is None:
        pytest.skip('no fork server')
    out = str(tmp_path)
    kw = dict(weight=[10., 10., 1.], epochNum=30, saveFreq=1000, verbose=False, dedup=True)
    rw.launch(conftest.FORKSERVER, 1, out, 'gloo', 'hip', _problem(), kw, 'd')
    rw.launch(conftest.FORKSERVER, 2, out, 'gloo', 'hip', _problem(), kw, 'd')
    a, b0, b1 = _compare(out, 'd', 2e-4, 2e-4)
    rw.launch(conftest.FORKSERVER, 1, out, 'gloo', 'hip', _problem(), dict(kw, dedup=False), 'r')
    r = np.load(os.path.join(out, 'r_w1_r0.npz'))
    np.testing.assert_allclose(a['loss'], r['loss'], rtol=2e-4)


def test_rccl_missing_on_one_rank_falls_back_on_all_ranks(tmp_path):
    """VN_COMM=try attempts the in-engine RCCL communicator; rank 1 alone cannot load RCCL
    (VN_RCCL_LIB names a missing file).  The bootstrap must not hang or mismatch collectives: BOTH ranks skip the
    communicator, keep the gradient SUM in torch.distributed and reproduce the one-rank run."""
    if conftest.FORKSERVER is None:
        pytest.skip('no fork server')
    out = str(tmp_path)
    kw = dict(weight=[10., 10., 1.], epochNum=10, saveFreq=1000, verbose=False)
    rw.launch(conftest.FORKSERVER, 1, out, 'gloo', 'hip', _problem(), kw, 'f')
    rw.launch(conftest.FORKSERVER, 2, out, 'gloo', 'hip', _problem(), kw, 'f', timeout=240,
              env={'VN_COMM': 'try', 'VN_TEST_BREAK_RCCL_ON_RANK': '1'})
    a, b0, b1 = _compare(out, 'f', 2e-4, 2e-4)
    assert str(b0['comm']) == 'torch' and str(b1['comm']) == 'torch'


def test_two_ranks_with_an_empty_shard_gloo(tmp_path):
    """HIP engine with n_k == 0 on rank 1 (nt < batchLen * world): the BC/IC tiles still run there and the
    rank joins the collective."""
    if conftest.FORKSERVER is None:
        pytest.skip('no fork server')
    out = str(tmp_path)
    kw = dict(weight=[10., 10., 1.], epochNum=10, saveFreq=1000, verbose=False, batchLen=600)
    rw.launch(conftest.FORKSERVER, 1, out, 'gloo', 'hip', _problem(), kw, 'e')
    rw.launch(conftest.FORKSERVER, 2, out, 'gloo', 'hip', _problem(), kw, 'e')
    a, b0, b1 = _compare(out, 'e', 2e-4, 2e-4)
    assert list(b1['block']) == [600, 600]


def test_two_ranks_two_gpus_rccl(tmp_path):
    """The production N > 1 path: one process per GPU, `nccl` (= RCCL) process group for bootstrap, the engine's
    own RCCL communicator for the gradient SUM inside vn_train_epoch."""
    if conftest.FORKSERVER is None:
        pytest.skip('no fork server')
    if _ndev() < 2:
        pytest.skip('needs >= 2 GPUs (this box has %d)' % _ndev())
    out = str(tmp_path)
    kw = dict(weight=[10., 10., 1.], epochNum=30, saveFreq=1000, verbose=False)
    rw.launch(conftest.FORKSERVER, 1, out, 'nccl', 'hip', _problem(), kw, 'n')
    rw.launch(conftest.FORKSERVER, 2, out, 'nccl', 'hip', _problem(), kw, 'n')
    a, b0, b1 = _compare(out, 'n', 2e-4, 2e-4)
    assert str(b0['comm']) == 'rccl'
    rw.launch(conftest.FORKSERVER, 2, out, 'nccl', 'hip', _problem(), kw, 't', env={'VN_COMM': 'torch'})
    t0 = np.load(os.path.join(out, 't_w2_r0.npz'))
    assert str(t0['comm']) == 'torch'
    np.testing.assert_allclose(t0['theta'], b0['theta'], rtol=0, atol=1e-6)


def test_bench_self_launch_two_ranks_gloo(tmp_path):
    """`python bench.py --gpus 2` starts its own ranks (here sharing the GPU over gloo) and prints one JSON line."""
    import json
    import subprocess
    import sys
    if conftest.FORKSERVER is None:
        pytest.skip('no fork server')
    # this process has initialised the GPU: the bench parent is started by the clean fork server, not by us
    q = conftest.FORKSERVER.Queue()
    p = conftest.FORKSERVER.Process(target=_bench_child, args=(q,))
    p.start()
    rc, text = q.get(timeout=900)
    p.join(60)
    assert rc == 0, text[-2000:]
    line = [ln for ln in text.splitlines() if ln.startswith('{')][-1]
    js = json.loads(line)
    assert js['n_gpus'] == 2 and js['value'] > 0 and len(js['per_rank']) == 2
    assert js['comm']['allreduce_ms'] > 0 and 'rehearsal' in js
    assert js['per_rank'][0]['rows'] + js['per_rank'][1]['rows'] == js['config']['training_points_per_step']


def _bench_child(q, gpus=2, extra=()):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VN_DIST_BACKEND='gloo')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(gpus), '--steps', '3', '--warmup', '1'] + list(extra),
                       capture_output=True, text=True, env=env, timeout=800)
    q.put((r.returncode, r.stdout + '\n' + r.stderr))


def test_bench_three_ranks_uneven_blocks_reproduce_the_one_rank_loss():
    """Rehearsal of the driver's SCALE run with UNEVEN shards: 100 000 test functions over 3 ranks = 33 334 + 33 334 +
    33 332 (VarNetUtility.py:825-838: batchLen = ceil(nt / puNum), the last tower takes the rest), BC/IC set replicated with
    its weights / 3 (:900-901).  The three ranks share this box's GPU over gloo (the pool allows at most 6 processes on a
    card, so the 8-rank case is rehearsed on the CPU engine: tests/test_distributed_gloo.py).  The whole-job loss after
    the same 4 steps must equal the one-rank run's to fp32 summation order: a wrong shard boundary, a doubled BC/IC term
    or a missed weight division all show up there."""
    import json
    if conftest.FORKSERVER is None:
        pytest.skip('no fork server')
    lines = []
    for gpus in (3, 1):
        q = conftest.FORKSERVER.Queue()
        p = conftest.FORKSERVER.Process(target=_bench_child, args=(q, gpus, ('--no-cpu-baseline', '--no-dedup', '--no-extra')))
        p.start()
        rc, text = q.get(timeout=900)
        p.join(60)
        assert rc == 0, text[-2000:]
        lines.append(json.loads([ln for ln in text.splitlines() if ln.startswith('{')][-1]))
    js, one = lines
    q = js['config']['quad_points_per_test_function']
    assert js['n_gpus'] == 3 and [r['rows'] for r in js['per_rank']] == [33334 * q, 33334 * q, 33332 * q]
    assert sum(r['rows'] for r in js['per_rank']) == js['config']['training_points_per_step'] == 6400000
    assert js['comm']['bc_ic_weight_divisor'] == 3 and js['comm']['payload_bytes'] == (10451 + 4) * 4
    from varnet_amd.engine import VNEngine
    if VNEngine.comm_available():         # the shard test is about sharding: a box without RCCL reports None here and still passes
        assert isinstance(js['comm']['rccl_version'], int) and js['comm']['rccl_version'] > 20000
    else:
        assert js['comm']['rccl_version'] is None
    assert js['comm']['vn_comm_size'] == [1, 0]                     # gloo rehearsal: the collective is torch's, not the engine's
    assert one['n_gpus'] == 1 and 'comm' not in one
    la, lb = js['config']['loss_after'], one['config']['loss_after']
    assert abs(la - lb) <= 2e-5 * abs(lb), (la, lb)


def test_processors_list_in_one_user_process(tmp_path):
    """VarNet(..., processors=['GPU:0','GPU:1'], controller=...) -- the reference's way to go multi-GPU
    (TFModel.py:120-165) -- from one user process: the object forks one tower per entry and forwards train /
    evaluate / residual / simRes / loadModel.  Here both towers share GPU 0 over gloo unless the box has 2 GPUs."""
    if conftest.FORKSERVER is None:
        pytest.skip('no fork server')
    out = str(tmp_path)
    backend = 'nccl' if _ndev() >= 2 else 'gloo'
    q = conftest.FORKSERVER.Queue()
    p = conftest.FORKSERVER.Process(target=rw.run_controller, args=(out, backend, 2, q))
    p.start()
    rc, text = q.get(timeout=600)
    p.join(60)
    assert rc == 0, text
    kw = dict(weight=[10., 10., 1.], epochNum=30, saveFreq=10, verbose=False)
    rw.launch(conftest.FORKSERVER, 1, out, 'gloo', 'hip', _problem(), kw, 'c1')
    a = np.load(os.path.join(out, 'c1_w1_r0.npz'))
    c = np.load(os.path.join(out, 'ctl.npz'))
    np.testing.assert_allclose(c['w'] * np.array([2.0, 2.0, 1.0]), a['w'], rtol=1e-4)     # two towers: per-feed BC/IC weights halved
    np.testing.assert_allclose(c['loss'], a['loss'], rtol=2e-4)
    assert c['u'].shape == (600, 1) and np.isfinite(c['err']) and c['cApp'].shape == (51, 1) and int(c['n']) in (10, 20, 30)
