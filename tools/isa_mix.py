"""Instruction mix of the fused kernel's instantiations from the compiler's assembly (whole kernel: prologue + tile loop + flush):
   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -S --cuda-device-only varnet_amd/csrc/vn_fused16.hip -o /tmp/f16.s
   python tools/isa_mix.py /tmp/f16.s ILi3ELi8ELb0 ILi5ELi13ELb0
The last column prices the shared fp32 datapath: 32 cycles per 16x16x4 MFMA, 8 per 4x4x1, 4 per vector op, 16 per transcendental."""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_ZN\S+vn_fused16_kernel\S+:', l)]
for key in sys.argv[2:]:
    i0 = [i for i, n in starts if key in n][0]
    i1 = next(i for i in range(i0, len(lines)) if 's_endpgm' in lines[i])
    c = Counter()
    for ln in lines[i0 + 1:i1]:
        ln = ln.strip()
        if not ln or ln[0] in ';.' or ln.endswith(':'):
            continue
        op = ln.split()[0]
        k = ('mfma16' if op.startswith('v_mfma_f32_16x16') else 'mfma4' if op.startswith('v_mfma') else
             'trans' if op.startswith(('v_exp', 'v_rcp', 'v_log', 'v_sqrt', 'v_rsq')) else
             'dpp/perm' if ('dpp' in ln or op.startswith(('v_permlane', 'v_readlane', 'v_writelane', 'v_readfirstlane'))) else
             'vpk' if op.startswith('v_pk_') else 'valu' if op.startswith('v_') else 'ds' if op.startswith('ds_') else
             'vmem' if op.startswith(('global_', 'buffer_', 'scratch_')) else 'waitcnt' if op.startswith('s_waitcnt') else
             'barrier' if op.startswith('s_barrier') else 'salu' if op.startswith('s_') else 'other')
        c[k] += 1
    dp = c['mfma16'] * 32 + c['mfma4'] * 8 + (c['valu'] + c['vpk'] + c['dpp/perm']) * 4 + c['trans'] * 16
    print(key, dict(c), 'instructions', sum(c.values()), 'datapath cycles ~', dp)


def loop_histogram(path, key, top=70):
    """opcode histogram of the largest loop (the tile loop) of one instantiation"""
    lines = open(path).read().split('\n')
    i0 = next(i for i, l in enumerate(lines) if re.match(r'^_ZN\S+vn_fused16_kernel' + key, l))
    i1 = next(i for i in range(i0, len(lines)) if 's_endpgm' in lines[i])
    lab = {}
    for i in range(i0, i1):
        m = re.match(r'^(\.LBB\d+_\d+):', lines[i])
        if m:
            lab[m.group(1)] = i
    best = (0, 0, 0)
    for i in range(i0, i1):
        m = re.search(r's_c?branch\S*\s+(\.LBB\d+_\d+)', lines[i])
        if m and m.group(1) in lab and lab[m.group(1)] < i and i - lab[m.group(1)] > best[0]:
            best = (i - lab[m.group(1)], lab[m.group(1)], i)
    ops = Counter()
    for ln in lines[best[1]:best[2] + 1]:
        ln = ln.strip()
        if not ln or ln[0] in ';.' or ln.endswith(':'):
            continue
        ops[ln.split()[0]] += 1
    print('# tile loop of', key, ':', sum(ops.values()), 'instructions')
    for op, n in ops.most_common(top):
        print('%6d %s' % (n, op))


if len(sys.argv) > 2 and sys.argv[2] == 'loop':
    pass
