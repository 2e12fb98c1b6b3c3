"""CPU tier: what bench.py does when it cannot run (no GPU here; a dying rank on the GPU box): ONE JSON line with an
"error" key on stdout and a non-zero exit status, so that a first multi-GPU hardware run leaves a diagnosis; and the
static-traffic check that keeps `roofline.traffic` from going stale."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_bench_failure_leaves_a_json_line():
    if _has_gpu():
        pytest.skip('GPU present: the failure path needs a box without one')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0'], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode != 0
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    js = json.loads(line)
    assert 'error' in js and js['rank'] == 0 and js['world'] == 1 and js['traceback_tail']


def test_self_launch_reports_a_failed_rank():
    """`python bench.py --gpus 2` without GPUs: both ranks die, each prints its own error line, the parent adds the
    summary line and exits non-zero (varnet_amd/launch.py ends the peers of the first rank that fails)."""
    if _has_gpu():
        pytest.skip('GPU present')
    env = dict(os.environ, VN_DIST_BACKEND='gloo')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert lines and all('error' in js for js in lines)
    assert any('exited with status' in js['error'] and js.get('n_gpus') == 2 for js in lines)


WORKER = os.path.join(ROOT, 'tests', 'launch_worker.py')


def test_launch_deadline_says_where_every_rank_was(capfd):
    """a first N > 1 run that wedges in the communicator bootstrap must not be killed at somebody else's
    limit having printed nothing.  Rank 1 sleeps forever in stage `comm_init`, rank 0 waits for it in a collective: within the
    launcher's deadline the parent ends both BY PID, prints ONE JSON line (who was alive, every rank's last stage) and
    returns 124."""
    import time
    sys.path.insert(0, ROOT)
    from varnet_amd import launch
    t0 = time.time()
    # (the clock starts when both ranks have finished their imports; the gloo rendezvous behind that takes a few seconds on a
    # loaded machine, so the deadline leaves room for both ranks to reach `comm_init` before it expires)
    rc = launch.spawn_ranks([WORKER, 'stuck'], 2, deadline_s=15.0, import_grace_s=90.0)
    dt = time.time() - t0
    assert rc == 124 and dt < 150
    out = capfd.readouterr().out
    lines = [json.loads(ln) for ln in out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    js = lines[0]
    assert 'deadline' in js['error'] and js['alive'] == [0, 1] and js['ranks'] == 2
    assert js['last_stage'] == {'0': 'comm_init', '1': 'comm_init'}
    assert js['s_in_last_stage']['1'] >= 5.0
    assert launch.last_report['reason'].startswith('launch deadline')


def test_launch_reports_the_stage_of_a_rank_that_died():
    sys.path.insert(0, ROOT)
    from varnet_amd import launch
    rc = launch.spawn_ranks(['-c', 'import os, sys; sys.path.insert(0, %r); from varnet_amd.launch import mark_stage; '
                             'mark_stage("pg_init"); sys.exit(7 if os.environ["RANK"] == "1" else 0)' % ROOT], 2, deadline_s=30.0)
    assert rc == 7
    rep = launch.last_report
    assert rep['reason'] == 'rank 1 exited with status 7' and rep['last_stage']['1'] == 'pg_init' and rep['exit_status']['1'] == 7


def test_launch_deadline_is_a_bootstrap_deadline_not_a_wall_clock_limit():
    """a healthy job whose ranks are past the bootstrap must not be ended for running longer than the deadline
    (the staleness of a BOOTSTRAP stage is what the deadline bounds); an overall limit is opt-in."""
    sys.path.insert(0, ROOT)
    from varnet_amd import launch
    t0 = time.time()
    rc = launch.spawn_ranks([WORKER, 'long_healthy', '12'], 2, deadline_s=4.0)
    assert rc == 0 and time.time() - t0 >= 12.0
    assert launch.last_report['last_stage'] == {'0': 'done', '1': 'done'}
    rc = launch.spawn_ranks([WORKER, 'long_healthy', '60'], 2, deadline_s=4.0, overall_s=15.0)
    assert rc == 124 and 'overall limit' in launch.last_report['reason']


def test_one_rank_step_failure_ends_the_job_and_names_the_error(capfd):
    """VERDICT r5 (weak 6): under a communicator a rank whose gradient fails returns before the all-reduce and its peers wait
    in it.  Settled semantics (include/varnet_hip.h, "Failure under a communicator"): the failing rank's process exits
    non-zero, the launcher ends the peers by PID within seconds -- not at its deadline -- and its report carries the failing
    rank's stage, which VNEngine._ck set to the engine's error text."""
    sys.path.insert(0, ROOT)
    from varnet_amd import launch
    t0 = time.time()
    rc = launch.spawn_ranks([WORKER, 'step_fail'], 2, deadline_s=120.0)
    dt = time.time() - t0
    assert rc == 3 and dt < 90
    rep = launch.last_report
    assert rep['reason'] == 'rank 1 exited with status 3' and rep['alive'] == [0]      # rank 0 sat in the collective: ended by PID
    assert rep['last_stage']['1'].startswith('engine_error: varnet_hip error 1: batch 7 was never registered')
    assert rep['last_stage']['0'] == 'timed'
    lines = [json.loads(ln) for ln in capfd.readouterr().out.splitlines() if ln.startswith('{')]
    assert any(js.get('rank') == 1 and 'batch 7' in js.get('error', '') for js in lines)


def test_rank_watchdog_ends_a_rank_with_its_last_stage():
    """the same deadline for a rank started by another launcher (torch.distributed.run): the rank's own one-line diagnosis"""
    r = subprocess.run([sys.executable, WORKER, 'watchdog'], capture_output=True, text=True, timeout=120)
    assert r.returncode == 124
    js = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert js['last_stage'] == 'comm_init' and 'deadline' in js['error'] and js['rank'] == 0


@pytest.mark.parametrize('kind,expect_ok', [('distinct', True), ('same', False)])
def test_comm_bootstrap_compares_physical_gpus_not_ordinals(kind, expect_ok, capfd):
    """with per-rank visibility masks (SLURM --gpus-per-task) or several nodes, distinct physical GPUs share
    device ordinal 0.  The bootstrap must bring RCCL up there (compare (host, uuid)), and refuse only exact duplicates -- on
    every rank alike.  Runs VNEngine.comm_init_from_torch over gloo on a stand-in engine with scripted probes."""
    sys.path.insert(0, ROOT)
    from varnet_amd import launch
    rc = launch.spawn_ranks([WORKER, 'bootstrap', kind], 2, deadline_s=120.0)
    assert rc == 0
    lines = [json.loads(ln) for ln in capfd.readouterr().out.splitlines() if ln.startswith('{')]
    assert sorted(js['rank'] for js in lines) == [0, 1]
    for js in lines:
        assert js['ok'] is expect_ok
        if expect_ok:
            assert js['inited'] == [js['rank'], 2] and js['last_stage'] == 'comm_agree'
        else:
            assert js['inited'] is None and 'share a physical GPU' in js['why'] and js['last_stage'] == 'probe'


def test_comm_bootstrap_survives_a_wedged_comm_init(capfd):
    """ncclCommInitRank has no timeout.  Rank 1's comm_init never returns: within $VN_COMM_INIT_TIMEOUT_S BOTH ranks fall back
    to torch.distributed, and rank 0 -- whose communicator came up -- abandons it instead of calling ncclCommDestroy against
    a wedged peer.  A first multi-GPU run then still produces its numbers (on the three-call path) instead of a dead record."""
    sys.path.insert(0, ROOT)
    from varnet_amd import launch
    rc = launch.spawn_ranks([WORKER, 'bootstrap', 'wedged'], 2, deadline_s=120.0, env_extra={'VN_COMM_INIT_TIMEOUT_S': '3'})
    assert rc == 0
    lines = {js['rank']: js for js in (json.loads(ln) for ln in capfd.readouterr().out.splitlines() if ln.startswith('{'))}
    assert sorted(lines) == [0, 1]
    for r, js in lines.items():
        assert js['ok'] is False and js['abandoned'] is True and js['destroyed'] is False and js['last_stage'] == 'comm_agree'
        assert js['abandon_calls'] == 1      # the communicator is withdrawn INSIDE the engine (vn_comm_abandon), not only flagged
    assert 'did not return within 3 s on rank 1' in lines[1]['why'] and 'abandoned' in lines[0]['why']
    assert lines[0]['inited'] == [0, 2] and lines[1]['inited'] is None


def test_shared_gpus_flags_exact_duplicates_only():
    sys.path.insert(0, ROOT)
    from varnet_amd.engine import VNEngine
    sg = VNEngine.shared_gpus
    assert sg([('a', 'uuid:1'), ('b', 'uuid:1'), ('a', 'uuid:2')]) == []          # same uuid string on two hosts
    assert sg([('a', 'uuid:1'), ('a', 'uuid:2'), ('a', 'uuid:1')]) == [(0, 2)]
    assert sg([['a', 'x'], ['a', 'x']]) == [(0, 1)]                                # all_gather_object hands lists back
    # one host, ONE visibility mask: the ordinal decides, even when the runtime reports the same (degenerate) identifiers
    assert sg([('a', 'uuid:0', '0,1|', 0), ('a', 'uuid:0', '0,1|', 1)]) == []
    assert sg([('a', 'uuid:7', '0,1|', 1), ('a', 'uuid:9', '0,1|', 1)]) == [(0, 1)]
    # per-rank masks (every rank sees its card as ordinal 0): the identifiers decide
    assert sg([('a', 'uuid:1|pci:0:5a:0', '0|', 0), ('a', 'uuid:2|pci:0:5b:0', '1|', 0)]) == []
    assert sg([('a', 'uuid:1|pci:0:5a:0', '0|', 0), ('a', 'uuid:1|pci:0:5a:0', '0,1|', 0)]) == [(0, 1)]
    # 8 ranks of one node, the launcher's usual picture
    assert sg([('n', 'uuid:%d' % r, '|', r) for r in range(8)]) == []


def test_static_traffic_is_refused_for_another_kernel_or_config():
    sys.path.insert(0, ROOT)
    import bench
    js = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
    assert 'kernel_source_sha256' in js and '<' in js['kernel'] and js['config'] == 3        # template arguments recorded
    t, src = bench.static_traffic('vn_fused16_kernel<4, 13, false>', 3, 1)
    assert t is None and 'was collected on' in src
    t, src = bench.static_traffic(js['kernel'], 3, 2)
    assert t is None and 'N=1' in src
    t, src = bench.static_traffic(js['kernel'], 5, 1)
    assert t is None


def test_committed_counter_files_match_the_tree():
    """the driver's bench line lost `roofline.traffic` to a commit made after the counter pass.  The
    committed counter files must carry the hash of the CURRENT code of the kernel they profiled (comment-stripped translation
    unit + every header it includes + the Makefile's compile flags): a kernel edit without a re-run of tools/collect_profiles.sh +
    tools/summarise_profiles.py fails here, in the CPU tier, before the driver's run can print traffic: null."""
    sys.path.insert(0, ROOT)
    import bench
    for cfg, name in bench.TRAFFIC_FILES.items():
        path = os.path.join(ROOT, 'profiles', name)
        assert os.path.exists(path), 'no counter file for config %d: run tools/collect_profiles.sh' % cfg
        js = json.load(open(path))
        assert js['config'] == cfg
        assert js['kernel_source_sha256'] == bench.kernel_source_hash(js['kernel']), \
            '%s predates the kernel code in the tree: re-run tools/collect_profiles.sh + tools/summarise_profiles.py' % name
        t, src = bench.static_traffic(js['kernel'], cfg, 1)
        assert t == js['hbm_bytes_per_launch'] and t > 0 and src.startswith('profiles/' + name)
    # the de-duplicated formulation's four kernels (round 5): every one of them against its own sources
    path = os.path.join(ROOT, 'profiles', bench.DEDUP_TRAFFIC_FILE)
    assert os.path.exists(path), 'no counter file for the de-duplicated formulation: run tools/collect_profiles.sh'
    js = json.load(open(path))
    names = [k['kernel'].split('<')[0] for k in js['kernels']]
    assert names[0] in ('vn_split16_pgrad_kernel', 'vn_pgrad16_kernel')      # the (u, grad u) pass: bf16 pieces at hidden widths 33..64
    assert names[1:] == ['vn_dedup_seed_kernel', 'vn_dedup_gather_kernel', 'vn_fused16_kernel']
    for k in js['kernels']:
        assert k['kernel_source_sha256'] == bench.kernel_source_hash(k['kernel']), '%s predates the code of %s' % (bench.DEDUP_TRAFFIC_FILE, k['kernel'])
    t, src = bench.dedup_traffic(3)
    assert t == js['hbm_bytes_per_step'] and t > 0
    assert bench.dedup_traffic(2)[0] is None


def test_kernel_hash_ignores_comments_and_layout_only():
    sys.path.insert(0, ROOT)
    import bench
    a = 'int f(int x) { /* doc */ return x + 1;   // one\n}\nconst char* s = "// not a comment";'
    b = 'int f(int x) {\n    return x + 1;\n}\n// trailing words\nconst char* s = "// not a comment";'
    c = 'int f(int x) { return x + 2; }\nconst char* s = "// not a comment";'
    assert bench.strip_comments(a) == bench.strip_comments(b) != bench.strip_comments(c)
    assert '// not a comment' in bench.strip_comments(a)
    assert bench.kernel_source_hash('vn_fused16_kernel<5, 13, false>') == bench.kernel_source_hash('vn_fused16_kernel')
    assert bench.kernel_source_hash('some_other_kernel') is None
    # the headers every object depends on and the compile flags decide the code too
    assert 'include/varnet_hip.h' in bench.KERNEL_SOURCES['vn_fused16_kernel'] and 'varnet_amd/csrc/vn_fused16_common.h' in bench.KERNEL_SOURCES['vn_fused16_kernel']
    assert '-fno-slp-vectorize' in bench.effective_cxxflags() and '--offload-arch=gfx950' in bench.effective_cxxflags()
    assert len({bench.kernel_source_hash(k) for k in ('vn_fused16_kernel', 'vn_pgrad16_kernel', 'vn_dedup_seed_kernel', 'vn_split16_pgrad_kernel')}) == 4
