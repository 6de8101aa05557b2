"""Timeline of one stage body of k_qp_ipm between full LDS waits (cross-compiled ISA, no GPU needed).

Every `s_waitcnt lgkmcnt(0)` ends a segment; per segment the script prints how many FP64 / other vector instructions,
LDS reads / writes, vector-memory instructions and branches it holds.  A lone wavefront pays a full LDS round trip per
segment, so the number of segments per stage is what DESIGN.md section 4 (point 6) counts.

usage:  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -Iinclude -save-temps -c -o /dev/null safe_mpc_amd/csrc/engine.hip
        python scripts/qp_wait_timeline.py engine-hip-amdgcn-amd-amdhsa-gfx950.s [B1|F1|B2|F2] [copy index] [kernel substring]
"""
import re, sys
path = sys.argv[1]
which = sys.argv[2] if len(sys.argv) > 2 else 'B1'
idx = int(sys.argv[3]) if len(sys.argv) > 3 else 1
want = sys.argv[4] if len(sys.argv) > 4 else '_ZN4smpc8k_qp_ipmILi6ELi6E'
txt = open(path).read().split('\n')
inside, marks = False, []
for n, l in enumerate(txt):
    if l.startswith(want) and ':' in l:
        inside = True
    if inside and 's_endpgm' in l:
        break
    if inside and 'QPMARK' in l:
        marks.append((n, l.strip()))
begins = [n for n, l in marks if which + '_BEGIN' in l]
ends = [n for n, l in marks if which + '_END' in l]
b = begins[idx]
e = [x for x in ends if x > b][0]      # (the forward sweeps' loops are rotated: take the next END after this BEGIN)
new = lambda: {'v64': 0, 'valu': 0, 'ldsr': 0, 'ldsw': 0, 'vmem': 0, 'br': 0, 'salu': 0}
seg, out = new(), []


def flush(tag):
    global seg
    out.append('%-6s fp64 %3d  valu %3d  lds reads %2d writes %2d  vmem %2d  branches %d' %
               (tag, seg['v64'], seg['valu'], seg['ldsr'], seg['ldsw'], seg['vmem'], seg['br']))
    seg = new()


for l in txt[b:e]:
    m = re.match(r'\s+([a-z_0-9]+)', l)
    if not m or l.strip().startswith(('.', ';')):
        continue
    op = m.group(1)
    if op == 's_waitcnt':
        if 'lgkmcnt(0)' in l:
            flush('wait')
        continue
    if op.startswith('ds_read'): seg['ldsr'] += 1
    elif op.startswith('ds_write'): seg['ldsw'] += 1
    elif op.startswith(('global_', 'scratch_')): seg['vmem'] += 1
    elif op.startswith('s_cbranch') or op == 's_branch': seg['br'] += 1
    elif re.match(r'v_\w+_f64', op): seg['v64'] += 1
    elif op.startswith('v_'): seg['valu'] += 1
    else: seg['salu'] += 1
flush('end')
print('\n'.join(out))
print(len(out), 'segments in', which, 'copy', idx)
