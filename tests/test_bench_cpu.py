"""CPU tier: what bench.py does when it cannot run (no GPU here; a dying rank on the GPU box): ONE JSON line with an
"error" key on stdout and a non-zero exit status, so that a first multi-GPU hardware run leaves a diagnosis; and the
static-traffic check that keeps `roofline.traffic` from going stale."""
import json
import os
import subprocess
import sys

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
    committed counter files must carry the hash of the CURRENT code of the kernel they profiled (comment-stripped
    vn_fused16.hip + vn_internal.h): a kernel edit without a re-run of tools/collect_profiles.sh +
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
