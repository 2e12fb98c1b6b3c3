"""Copy the judged summaries of gpurun_out/prof_<tag>/ into profiles/ (tracked):
   python tools/summarise_profiles.py <tag> <round-prefix>      e.g.  r1b r1"""
import csv, collections, glob, json, os, shutil, sys
tag, pre = sys.argv[1], sys.argv[2]
src = 'gpurun_out/prof_%s' % tag
os.makedirs('profiles', exist_ok=True)
full = not (len(sys.argv) > 3 and sys.argv[3] == 'cfg2')      # 'cfg2': only the config-2 counter passes (tools/collect_cfg2_pmc.sh)
if full:
    shutil.copy(src + '/bench.json', 'profiles/%s_bench_default.json' % pre)
    for extra, dst in (('bench_cfg2.json', '%s_bench_cfg2.json'), ('shard_perf.txt', '%s_shard_perf.txt'), ('mor_perf.txt', '%s_mor_perf.txt'),
                       ('q216.txt', '%s_q216_perf.txt'), ('generic_w64.txt', '%s_generic_w64_perf.txt'), ('width60.txt', '%s_width60_perf.txt')):
        if os.path.exists(src + '/' + extra):
            shutil.copy(src + '/' + extra, 'profiles/' + dst % pre)
    for sub, dst in (('stats_q216', '%s_twopass_q216_kernel_stats.csv'), ('stats_generic', '%s_generic_w64_kernel_stats.csv')):
        g = glob.glob(src + '/' + sub + '/**/s_kernel_stats.csv', recursive=True)
        if g:
            shutil.copy(g[0], 'profiles/' + dst % pre)
    tp = {}
    for f in glob.glob(src + '/tpmc_*/**/p_counter_collection.csv', recursive=True):
        d = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            for kn in ('vn_fused16_kernel', 'vn_seed_kernel'):
                if kn in r['Kernel_Name']:
                    d[kn][r['Counter_Name']].append(float(r['Counter_Value']))
        for kn, dd in d.items():
            for k, v in dd.items():
                v = sorted(v)
                # the fused kernel runs twice per step (forward-only, then reverse with seeds): report both halves
                tp.setdefault(kn, {})[k] = {'min': v[0], 'median': v[len(v) // 2], 'max': v[-1], 'launches': len(v)}
    if tp:
        tp['note'] = ('two-pass route at integNum 216 (tools/q216_perf.py: 30 000 test functions x 216 points, 5x50): per launch; '
                      'vn_fused16_kernel is launched in forward-only mode (min column) and in reverse mode (max column) each step; '
                      'MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / (GRBM_GUI_ACTIVE / 8)')
        json.dump(tp, open('profiles/%s_pmc_twopass.json' % pre, 'w'), indent=1)
    gen = {}
    for f in glob.glob(src + '/gpmc_*/**/p_counter_collection.csv', recursive=True):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'vn_generic_bwd_kernel' in r['Kernel_Name']:
                d[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in d.items():
            v = sorted(v); gen[k] = v[len(v) // 2]
    if gen:
        cyc = gen.get('GRBM_GUI_ACTIVE', 0) / 8.0
        gen['derived'] = {'shader_cycles_per_launch': cyc,
                          'mfma_pipe_utilisation': gen.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024.0 / cyc if cyc else None,
                          'lds_utilisation': gen.get('SQ_LDS_IDX_ACTIVE', 0) / 256.0 / cyc if cyc else None,
                          'lds_conflict_share': gen.get('SQ_LDS_BANK_CONFLICT', 0) / max(gen.get('SQ_LDS_IDX_ACTIVE', 1), 1),
                          'wait_any_share': gen.get('SQ_WAIT_ANY', 0) / max(gen.get('SQ_WAVE_CYCLES', 1), 1),
                          'wait_inst_share': gen.get('SQ_WAIT_INST_ANY', 0) / max(gen.get('SQ_WAVE_CYCLES', 1), 1),
                          'hbm_bytes_per_launch': (2 * gen.get('FETCH_SIZE', 0) + gen.get('WRITE_SIZE', 0)) * 1024.0}
        gen['kernel'] = 'vn_generic_bwd_kernel, 3x64 net, config-3 sized inputs (tools/width_perf.py 64 3)'
        json.dump(gen, open('profiles/%s_pmc_generic_bwd.json' % pre, 'w'), indent=1)
    st = glob.glob(src + '/stats/**/s_kernel_stats.csv', recursive=True)[0]
    shutil.copy(st, 'profiles/%s_fused16_kernel_stats.csv' % pre)
    g = glob.glob(src + '/stats_dedup/**/s_kernel_stats.csv', recursive=True)
    if g:
        shutil.copy(g[0], 'profiles/%s_dedup_kernel_stats.csv' % pre)

sys.path.insert(0, '.')
import bench

# the hash that goes into a counter file is the one taken AT COLLECTION TIME (tools/collect_profiles.sh writes it before its
# first pass); a tree that has moved since then must not have its new code blessed by old counters
COLLECTED = {}
_hf = sorted(glob.glob(src + '/kernel_source_sha256*.json'))       # one file per collection call: parts A, B (and D: only the
if _hf:                                                             # de-duplicated formulation's passes, collected again later)
    for f in _hf:
        h = json.load(open(f))
        part_d = f.endswith('_D.json')
        for k, v in h.items():
            # the fused kernel is profiled by every part: all must name the same code; vn_pgrad16 / vn_dedup are profiled by A and,
            # when it ran, again by D, whose passes REPLACE A's (collect_profiles.sh D removes them first): D's hash is the one
            if k in COLLECTED and COLLECTED[k] != v and not (part_d and k != 'vn_fused16_kernel'):
                raise SystemExit('the collection calls of %s ran on different sources of %s (%s): collect again' % (src, k, _hf))
            COLLECTED[k] = v
    moved = [k for k, v in COLLECTED.items() if bench.kernel_source_hash(k) != v]
    if moved:
        raise SystemExit('the sources of %s changed after %s was collected: re-run tools/collect_profiles.sh, then summarise' % (moved, src))
else:
    print('WARNING: %s/kernel_source_sha256*.json absent (collected by an older script): hashing the tree as it is now' % src)


def collected_hash(kernel):
    key = [k for k in bench.KERNEL_SOURCES if str(kernel).startswith(k)]
    return COLLECTED.get(key[0]) if key and COLLECTED else bench.kernel_source_hash(kernel)


def pmc_summary(prefix, config, workload, alg_bytes, command, copy_csv):
    """Median per launch of every counter of the --pmc passes gpurun_out/prof_<tag>/<prefix>*/ for the dominant kernel."""
    tot, kfull = {}, None
    for f in glob.glob(src + '/' + prefix + '*/**/p_counter_collection.csv', recursive=True):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'vn_fused16_kernel' in r['Kernel_Name']:
                d[r['Counter_Name']].append(float(r['Counter_Value']))
                kn = r['Kernel_Name']
                kfull = kn[kn.index('vn_fused16_kernel'):].split('(')[0].strip()      # with its template arguments
        for k, v in d.items():
            v = sorted(v); tot[k] = v[len(v) // 2]
        if copy_csv:
            shutil.copy(f, 'profiles/%s_%s_fused16.csv' % (pre, [q for q in f.split('/') if q.startswith(prefix)][0].lower()))
    if 'FETCH_SIZE' not in tot or 'WRITE_SIZE' not in tot:
        return None
    out = {
        'kernel': kfull, 'kernel_source_sha256': collected_hash(kfull), 'config': config, 'workload': workload,
        'FETCH_SIZE_KB_per_launch': tot['FETCH_SIZE'], 'WRITE_SIZE_KB_per_launch': tot['WRITE_SIZE'],
        'correction': 'gfx950: FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE taken as is',
        'hbm_bytes_per_launch': (2 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024.0,
        'algorithmic_input_bytes_per_launch': alg_bytes, 'round': pre, 'command': command}
    if 'GRBM_GUI_ACTIVE' in tot:
        cyc = tot['GRBM_GUI_ACTIVE'] / 8.0
        out['mfma'] = {'SQ_VALU_MFMA_BUSY_CYCLES': tot['SQ_VALU_MFMA_BUSY_CYCLES'], 'GRBM_GUI_ACTIVE_sum_over_8_XCD': tot['GRBM_GUI_ACTIVE'],
                       'shader_cycles_per_launch': cyc, 'mfma_pipe_utilisation': tot['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cyc,
                       'SQ_LDS_IDX_ACTIVE': tot.get('SQ_LDS_IDX_ACTIVE'), 'SQ_LDS_BANK_CONFLICT': tot.get('SQ_LDS_BANK_CONFLICT'),
                       'lds_utilisation': tot['SQ_LDS_IDX_ACTIVE'] / 256.0 / cyc if 'SQ_LDS_IDX_ACTIVE' in tot else None}
    out['waves'] = {k: tot[k] for k in ('SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA',
                                       'SQ_INSTS_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES') if k in tot}
    return out


if full:
    out = pmc_summary('pmc_', 3, 'bench.py config 3 (6.4M points/step)', 128224000,
                      'tools/collect_profiles.sh (rocprofv3 --pmc <group> --kernel-trace, one pass per group) -- python bench.py --steps 3 --warmup 1 '
                      '--no-cpu-baseline --no-dedup --no-extra', True)
    json.dump(out, open('profiles/%s_pmc_traffic.json' % pre, 'w'), indent=1)
    json.dump(out, open('profiles/' + bench.TRAFFIC_FILES[3], 'w'), indent=1)      # the file bench.py quotes `traffic` from
    print(json.dumps(out, indent=1))
# config 2 (1D+t, 160 000 points, 4x50): Input [nT,2] + gcoef [nT,1] + BC/IC rows, f32
out2 = pmc_summary('c2pmc_', 2, 'bench.py --config 2 (160k points/step)', 160000 * 3 * 4 + 450 * 3 * 4,
                   'tools/collect_profiles.sh (rocprofv3 --pmc <group> --kernel-trace, one pass per group) -- python bench.py --config 2 --steps 50 '
                   '--warmup 5 --no-cpu-baseline --no-dedup', False)
if out2:
    json.dump(out2, open('profiles/%s_pmc_traffic_cfg2.json' % pre, 'w'), indent=1)
    json.dump(out2, open('profiles/' + bench.TRAFFIC_FILES[2], 'w'), indent=1)
    print(json.dumps(out2, indent=1))


# ---- de-duplicated formulation: HBM traffic and matrix-pipe use of its four kernels, per launch (one launch of each per step).
# The bench process runs the row-wise leg first, so the fused kernel's dispatches BEHIND the first vn_pgrad16 dispatch are the
# formulation's reverse launches.
def dedup_summary():
    # the (u, grad u) pass: vn_split16_pgrad_kernel where the net's hidden widths are 33..64 (round 6), else vn_pgrad16_kernel
    allrows = [r for f in glob.glob(src + '/ddpmc_*/**/p_counter_collection.csv', recursive=True) for r in csv.DictReader(open(f))]
    PG = 'vn_split16_pgrad_kernel' if any('vn_split16_pgrad_kernel' in r['Kernel_Name'] for r in allrows) else 'vn_pgrad16_kernel'
    DD = (PG, 'vn_dedup_seed_kernel', 'vn_dedup_gather_kernel', 'vn_fused16_kernel')
    per, names = collections.defaultdict(lambda: collections.defaultdict(list)), {}
    for f in glob.glob(src + '/ddpmc_*/**/p_counter_collection.csv', recursive=True):
        rows = list(csv.DictReader(open(f)))
        first = min([int(r['Dispatch_Id']) for r in rows if PG in r['Kernel_Name']] or [1 << 60])
        for r in rows:
            for kn in DD:
                if kn in r['Kernel_Name'] and (kn != 'vn_fused16_kernel' or int(r['Dispatch_Id']) > first):
                    per[kn][r['Counter_Name']].append(float(r['Counter_Value']))
                    full = r['Kernel_Name']
                    names[kn] = full[full.index(kn):].split('(')[0].strip()
    if any('FETCH_SIZE' not in per[kn] or 'WRITE_SIZE' not in per[kn] for kn in DD):
        return None
    med = lambda v: sorted(v)[len(v) // 2]
    ks = []
    for kn in DD:
        t = {k: med(v) for k, v in per[kn].items()}
        e = {'kernel': names[kn], 'kernel_source_sha256': collected_hash(names[kn]), 'FETCH_SIZE_KB_per_launch': t['FETCH_SIZE'],
             'WRITE_SIZE_KB_per_launch': t['WRITE_SIZE'], 'hbm_bytes_per_launch': (2 * t['FETCH_SIZE'] + t['WRITE_SIZE']) * 1024.0,
             'launches_profiled': len(per[kn]['FETCH_SIZE'])}
        if 'GRBM_GUI_ACTIVE' in t and t['GRBM_GUI_ACTIVE'] > 0:
            e['shader_cycles_per_launch'] = t['GRBM_GUI_ACTIVE'] / 8.0
            e['mfma_pipe_utilisation'] = t.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / 1024.0 / (t['GRBM_GUI_ACTIVE'] / 8.0)
        ks.append(e)
    return {'config': 3, 'round': pre, 'kernels': ks, 'hbm_bytes_per_step': sum(k['hbm_bytes_per_launch'] for k in ks),
            'correction': 'gfx950: FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE taken as is',
            'command': 'tools/collect_profiles.sh (rocprofv3 --pmc <group> --kernel-trace, one pass per group) -- python bench.py --steps 3 --warmup 1 '
                       '--no-cpu-baseline --no-extra',
            'note': 'one launch of each kernel per de-duplicated step; vn_fused16_kernel = its reverse-mode launches (dispatches behind the first '
                    '(u, grad u) dispatch)'}


if full:
    outd = dedup_summary()
    if outd:
        json.dump(outd, open('profiles/%s_pmc_traffic_dedup.json' % pre, 'w'), indent=1)
        json.dump(outd, open('profiles/' + bench.DEDUP_TRAFFIC_FILE, 'w'), indent=1)
        print(json.dumps(outd, indent=1))


# ---- issue-cycle model (review r3 item 3): what bounds a kernel whose f32 MFMAs and f32 vector instructions share one datapath is
# matrix-pipe cycles + vector-instruction issue cycles per SIMD, not the MFMA peak alone.  Per launch, from the counter passes:
#   matrix  = SQ_VALU_MFMA_BUSY_CYCLES / #SIMDs            vector = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / #SIMDs x 4 cycles
#   (SQ_INSTS_VALU counts the MFMAs too; 4 cycles = issue cost of one wave's vector instruction, MI355X_MICROARCH.md cycle constants;
#    transcendentals cost 8, so the bound is a lower one)    kernel = GRBM_GUI_ACTIVE / 8 of the same profiled launches
def issue_entry(prefix, workload):
    tot, kfull = {}, None
    for f in glob.glob(src + '/' + prefix + '*/**/p_counter_collection.csv', recursive=True):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'vn_fused16_kernel' in r['Kernel_Name']:
                d[r['Counter_Name']].append(float(r['Counter_Value']))
                kn = r['Kernel_Name']
                kfull = kn[kn.index('vn_fused16_kernel'):].split('(')[0].strip()
        for k, v in d.items():
            v = sorted(v[len(v) // 4:]); tot[k] = v[len(v) // 2]            # (the first quarter of the dispatches: warm-up)
    need = ('SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA')
    if any(k not in tot for k in need):
        return None
    nsimd = 4 * 256
    matrix = tot['SQ_VALU_MFMA_BUSY_CYCLES'] / nsimd
    vinst = (tot['SQ_INSTS_VALU'] - tot['SQ_INSTS_MFMA']) / nsimd
    kernel = tot['GRBM_GUI_ACTIVE'] / 8.0
    return {'kernel': kfull, 'kernel_source_sha256': collected_hash(kfull), 'workload': workload, 'round': pre,
            'matrix_cycles_per_simd': matrix, 'vector_instructions_per_simd': vinst, 'cycles_per_vector_instruction': 4.0,
            'vector_cycles_per_simd': 4.0 * vinst, 'issue_bound_cycles': matrix + 4.0 * vinst, 'kernel_cycles_same_pass': kernel,
            'frac_of_issue_bound': (matrix + 4.0 * vinst) / kernel, 'matrix_pipe_busy': matrix / kernel,
            'counters': {k: tot[k] for k in sorted(tot)}}


issue = {}
for key, prefix, wl in (('config3', 'pmc_', 'bench.py config 3 (6.4M points/step, 5x50)'),
                        ('config2', 'c2pmc_', 'bench.py --config 2 (160 k points/step, 4x50)'),
                        ('config1', 'c1pmc_', 'bench.py --config 1 (96 k points/step, 3x20)'),
                        ('config5_minibatch', 'c5pmc_', 'tools/step_timeline.py run mor: one Adam step on a config-5 mini-batch (96 k points, [10,20,30])')):
    e = issue_entry(prefix, wl)
    if e:
        issue[key] = e
if issue:
    old = {}
    if os.path.exists('profiles/pmc_issue.json'):
        old = json.load(open('profiles/pmc_issue.json'))
    old.update(issue)
    json.dump(old, open('profiles/pmc_issue.json', 'w'), indent=1)
    json.dump(issue, open('profiles/%s_pmc_issue.json' % pre, 'w'), indent=1)
    for k, e in issue.items():
        print('issue model %-18s %-36s matrix %.0f + vector %.0f = %.0f of %.0f cycles per SIMD: %.3f' % (
            k, e['kernel'], e['matrix_cycles_per_simd'], e['vector_cycles_per_simd'], e['issue_bound_cycles'], e['kernel_cycles_same_pass'],
            e['frac_of_issue_bound']))
