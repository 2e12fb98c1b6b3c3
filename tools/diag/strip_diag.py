import re, sys
src = open(sys.argv[1]).read().split('\n')
def is_diag(name): return name in ('VN_FIXSTAMPS', 'VN_STAMPS') or name.startswith('VN_ABL_')
out = []
# stack entries: (kind, keeping, diag)  kind: 'diag' or 'other'
stack = []
def emitting(): return all(k for _, k, _ in stack)
i = 0
while i < len(src):
    ln = src[i]
    m = re.match(r'\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b\s*(.*)', ln)
    if m:
        d, rest = m.group(1), m.group(2)
        name = rest.split()[0] if rest.split() else ''
        if d in ('ifdef', 'ifndef'):
            if is_diag(name):
                stack.append(('diag', d == 'ifndef', True))     # macro undefined: ifdef -> drop, ifndef -> keep
            else:
                stack.append(('other', True, False))
                if emitting(): out.append(ln)
        elif d == 'if':
            stack.append(('other', True, False))
            if emitting(): out.append(ln)
        elif d in ('elif', 'else'):
            kind, keep, diag = stack[-1]
            if diag:
                assert d == 'else'
                stack[-1] = (kind, not keep, diag)
            else:
                if emitting(): out.append(ln)
        elif d == 'endif':
            kind, keep, diag = stack.pop()
            if not diag and emitting(): out.append(ln)
        i += 1
        continue
    if emitting():
        out.append(ln)
    i += 1
assert not stack
txt = '\n'.join(out)
# the comments that introduce the ablation blocks go with them
txt = txt.replace('''    // (-DVN_ABL_NOPUB / -DVN_ABL_NOBAR / -DVN_ABL_NOCONTRACT: diagnostic ablations, results wrong: what the publish
    // stores, the two workgroup barriers and the contraction of a round cost; profiles/r2_round_ablation.txt)
''', '')
txt = txt.replace('''  // Diagnostic ablations (-DVN_ABL_FWD_W / -DVN_ABL_BWD_W; results are WRONG, only counters and time matter):
  // the weight-fragment reads of the forward / backward GEMMs become bank-conflict-free (32 lanes of a half on
  // 32 distinct banks), to attribute SQ_LDS_BANK_CONFLICT (profiles/r2_lds_conflict_ablation.md).
''', '')
# drop the no-op macro definitions and their uses
txt = re.sub(r'^#define (FIXSTAMP|STAMP|WSTAMP|ESTAMP)\(i\) do \{\} while \(0\)\n', '', txt, flags=re.M)
txt = re.sub(r'^#define STAMP_(PARAMS|ARGS)\n', '', txt, flags=re.M)
txt = re.sub(r'^[ \t]*(FIXSTAMP|STAMP|WSTAMP|ESTAMP)\(\d+\);[ \t]*(//.*)?\n', '', txt, flags=re.M)
txt = re.sub(r' STAMP_(PARAMS|ARGS)\b', '', txt)
assert not re.search(r'STAMP|VN_ABL', txt), re.findall(r'.*(?:STAMP|VN_ABL).*', txt)[:10]
open(sys.argv[2], 'w').write(txt)
