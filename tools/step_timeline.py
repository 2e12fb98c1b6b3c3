"""Device-side timeline of small training steps: which kernels run, how long, and the gaps between them.

  run    : rocprofv3 --kernel-trace -d gpurun_out/tl_<name> -o t --output-format csv -- python tools/step_timeline.py run <workload> [steps]
  report : python tools/step_timeline.py report gpurun_out/tl_<name> [skip]

workloads: empty550 (5x50, no rows: fixed cost only) | cfg2 (BASELINE config 2: 4x50, 160 k points) |
           mor (config-5 mini-batch: [10,20,30], 96 k points) | cfg1 ([20] net, 96 k points) | shard8 (config 3, 1/8 shard)
The report takes the kernel-trace CSV (Start_Timestamp / End_Timestamp in ns, device clock domain), drops the first
`skip` dispatches and prints, per kernel name, the median duration, and the median gap from the end of the previous
dispatch to its start; the last line is the median period of the fused kernel = device time per step.
"""
import csv
import glob
import os
import sys

import numpy as np


def run(workload, steps):
    import os
    if workload == 'empty550':
        os.environ['VN_FULL_GRID'] = '1'          # the fixed cost of a FULL launch (an empty batch would otherwise get one workgroup)
    import torch
    sys.path.insert(0, '.')
    from varnet_amd.engine import VNEngine
    cfg = {
        'empty550': ([50] * 5, 3, 2, 64, 0, 0),
        'cfg2': ([50] * 4, 2, 1, 16, 10000, 450),
        'mor': ([10, 20, 30], 3, 1, 16, 6000, 1750),
        'cfg1': ([20], 2, 1, 16, 6000, 620),
        'shard8': ([50] * 5, 3, 2, 64, 12500, 14000),
    }[workload]
    widths, d_in, dim, q, n_k, nB = cfg
    n = n_k * q
    e = VNEngine(dim, d_in, widths, True, q)
    e.init_params(0)
    rng = np.random.default_rng(0)
    e.set_fe_table(rng.uniform(0, 1, q), rng.standard_normal(q))
    X = torch.rand(max(n, 1), d_in, device='cuda') * 2 - 1
    G = torch.randn(max(n, 1), dim, device='cuda')
    e.set_interior(0, X[:n], G[:n], None, n_k=n_k, detJ=1e-3)
    if nB:
        bi = torch.rand(nB, d_in, device='cuda') * 2 - 1
        bl = torch.randn(nB, device='cuda')
        e.set_bic(bi, bl, nB // 2, 2.0)
    else:
        e.set_bic(None, None, 0, 1.0)
    e.set_weights([1, 1, 1])
    acc = torch.zeros((), device='cuda')
    e.train_epoch([0] * 10, acc)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    e.train_epoch([0] * steps, acc)
    ev1.record()
    torch.cuda.synchronize()
    print('%s: %d steps, %.2f us/step (events around one vn_train_epoch call)' % (workload, steps, ev0.elapsed_time(ev1) / steps * 1e3))
    e.close()


def report(d, skip):
    f = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True))
    if not f:
        sys.exit('no kernel trace under ' + d)
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    rows = rows[skip:]
    by = {}
    prev_end = None
    starts = []
    for r in rows:
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        name = name.split('(')[0].strip()
        s, e_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = None if prev_end is None else s - prev_end
        by.setdefault(name, []).append((e_ - s, gap))
        if 'vn_fused16_kernel' in name or 'vn_persist' in name:
            starts.append(s)
        prev_end = e_
    print('# %s (%d dispatches after skipping %d)' % (f[0], len(rows), skip))
    print('%-60s %8s %12s %14s' % ('kernel', 'calls', 'median us', 'gap before us'))
    for name, v in by.items():
        dur = np.median([a for a, _ in v]) / 1e3
        gaps = [b for _, b in v if b is not None]
        print('%-60s %8d %12.2f %14.2f' % (name[:60], len(v), dur, np.median(gaps) / 1e3 if gaps else float('nan')))
    if len(starts) > 2:
        print('median period of the fused kernel (device time per step): %.2f us' % (np.median(np.diff(starts)) / 1e3))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 200)
    else:
        report(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 30)
