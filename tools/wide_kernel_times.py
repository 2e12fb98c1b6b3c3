"""Durations of the tile kernels (vn_wide.hip) on the interior rows of a config-3 sized step, read from a rocprofv3 kernel
trace:   rocprofv3 --kernel-trace -d DIR -o w --output-format csv -- python3 tools/layered_perf.py "128,128,128"
         python3 tools/wide_kernel_times.py DIR [label]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'vn_wide_fwd' in n or 'bwd_kernel' in n:
        agg['fwd' if 'fwd' in n else 'bwd'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
lab = sys.argv[2] if len(sys.argv) > 2 else ''
print('%-28s' % lab + '  '.join('%s %.2f ms (median of the %d interior launches)' % (k, sorted(x for x in v if x > 1.0)[len([x for x in v if x > 1.0]) // 2], len([x for x in v if x > 1.0])) for k, v in sorted(agg.items())))
