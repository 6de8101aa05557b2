"""Static instruction mix of one kernel in build/engine.s (cross-compiled, no GPU).  usage: isa_mix.py <mangled-name-prefix>"""
import re, collections, sys
txt = open('build/engine.s').read().split('\n')
want = sys.argv[1]
inside, cnt, n = False, collections.Counter(), 0
for ln in txt:
    if ln.startswith(want) and ':' in ln.split(';')[0]:
        inside = True
        continue
    if inside:
        if 's_endpgm' in ln:
            break
        m = re.match(r'\s+([a-z_0-9]+)\s', ln)
        if m:
            op = m.group(1); n += 1
            if re.match(r'v_(fma|mul|add|max|min)_f64', op): key = 'v_fma/mul/add_f64'
            elif 'f64' in op: key = 'v_other_f64 (rcp, cmp, cvt ...)'
            elif 'bpermute' in op or 'swizzle' in op: key = 'ds_bpermute'
            elif op.startswith('ds_read'): key = 'ds_read'
            elif op.startswith('ds_write'): key = 'ds_write'
            elif op.startswith('s_load'): key = 's_load'
            elif op.startswith('global'): key = 'global'
            elif 'scratch' in op: key = 'scratch'
            elif op == 's_waitcnt': key = 's_waitcnt'
            elif 'branch' in op: key = 'branch'
            elif op in ('v_mov_b32', 'v_cndmask_b32', 'v_accvgpr_write_b32', 'v_accvgpr_read_b32'): key = 'v_mov/cndmask/accvgpr'
            elif op.startswith('s_'): key = 'scalar other'
            else: key = 'vector other'
            cnt[key] += 1
print(n, 'instructions')
for k, v in cnt.most_common():
    print('%6d  %s' % (v, k))
