"""Audit of k_qp_ipm's generated code (cross-compiled, no GPU): per marked stage body (QPMARK comments in kernel_qp.hpp) the
vector-memory instructions, every s_waitcnt vmcnt(n) with its n, scratch accesses.  A vmcnt(0) inside a stage body means the
prefetch queue is drained there ("static vector-memory streams").
usage: python scripts/qp_asm_audit.py build/engine.s [kernel-name-substring]"""
import re, sys
path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else 'k_qp_ipmILi6ELi6'
txt = open(path).read().split('\n')
inside, region, stats = False, None, {}
order = []
for ln in txt:
    if re.match(r'^_ZN4smpc8' + want + r'.*:', ln) or (want in ln and ln.endswith(':') and ln.startswith('_Z') and not inside and 'k_qp_ipm' in ln):
        inside = True
        continue
    if not inside:
        continue
    if 's_endpgm' in ln:
        break
    m = re.search(r'QPMARK (\w+)_(BEGIN|END)', ln)
    if m:
        if m.group(2) == 'BEGIN':
            region = m.group(1) + '#%d' % sum(1 for k in order if k.startswith(m.group(1) + '#'))
            order.append(region)
            stats[region] = {'loads': 0, 'stores': 0, 'scratch': 0, 'waits': [], 'insts': 0, 'lds': 0}
        else:
            region = None
        continue
    if region is None:
        continue
    t = ln.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    st = stats[region]
    st['insts'] += 1
    if t.startswith('global_load') or t.startswith('flat_load') or t.startswith('buffer_load'):
        st['loads'] += 1
    elif t.startswith('global_store') or t.startswith('flat_store') or t.startswith('buffer_store'):
        st['stores'] += 1
    elif t.startswith('scratch_'):
        st['scratch'] += 1
    elif t.startswith('ds_'):
        st['lds'] += 1
    m = re.search(r's_waitcnt.*vmcnt\((\d+)\)', t)
    if m:
        st['waits'].append(int(m.group(1)))
for r in order:
    s = stats[r]
    print('%-8s insts %5d  lds %4d  loads %3d  stores %3d  scratch %3d  vmcnt waits %s' % (r, s['insts'], s['lds'], s['loads'], s['stores'], s['scratch'], s['waits']))
