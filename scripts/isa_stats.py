"""Instruction statistics of one kernel from a hipcc -save-temps .s file: waitcnt forms, memory / LDS / FP64 op counts."""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = [i for i, l in enumerate(lines) if key in l and l.rstrip().endswith('@' + l.split(':')[0]) or (l.startswith('_Z') and key in l and ': ' in l)][0]
end = [i for i, l in enumerate(lines) if i > start and l.startswith('.Lfunc_end')][0]
body = lines[start:end]
print('lines', len(body))
c = collections.Counter()
for l in body:
    l = l.strip()
    if l.startswith('s_waitcnt'):
        c[l.split(';')[0].strip()] += 1
for k, v in c.most_common(16):
    print('%5d  %s' % (v, k))
cnt = lambda pat: sum(1 for l in body if re.search(pat, l))
print('global_load', cnt('global_load'), 'global_store', cnt('global_store'), 'scratch', cnt('scratch_'))
print('ds_read', cnt(r'ds_read|ds_load'), 'ds_write', cnt(r'ds_write|ds_store'))
print('fma64', cnt('v_fma_f64'), 'mul64', cnt('v_mul_f64'), 'add64', cnt('v_add_f64'), 'rcp', cnt('v_rcp_f64'), 'rsq',
      cnt('v_rsq_f64'), 'div_scale', cnt('v_div_scale'), 'cndmask', cnt('v_cndmask'), 's_barrier', cnt('s_barrier'))
