"""Counts instructions between '; QPMARK X_BEGIN' / 'X_END' comments of one kernel in a hipcc -save-temps .s file."""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().endswith(':') or (l.startswith('_Z') and key in l and ': ' in l)][0]
end = [i for i, l in enumerate(lines) if i > start and l.startswith('.Lfunc_end')][0]
body = lines[start:end]
is_instr = lambda l: bool(re.match(r'\s+[a-z_0-9]+(\s|$)', l)) and not l.strip().startswith(('.', ';'))
marks = {}
for n, l in enumerate(body):
    m = re.search(r'QPMARK (\w+)', l)
    if m:
        marks.setdefault(m.group(1), []).append(n)
tot = 0
for name in ('B1', 'F1', 'B2', 'F2'):
    if name + '_BEGIN' in marks and name + '_END' in marks:
        a, b = marks[name + '_BEGIN'][0], marks[name + '_END'][-1]
        seg = [l for l in body[a:b] if is_instr(l)]
        cnt = lambda pat: sum(1 for l in seg if re.search(pat, l))
        print('%s: %5d instr | valu64 %4d  lds %4d  vmem %3d  salu %4d  branch %3d  waitcnt %3d' % (
            name, len(seg), cnt(r'v_\w+_f64'), cnt(r'\bds_'), cnt(r'global_|scratch_'), cnt(r'^\s+s_(?!waitcnt|cbranch|branch|barrier)'),
            cnt(r's_cbranch|s_branch'), cnt('s_waitcnt')))
        tot += len(seg)
print('sum per stage-iteration:', tot)
