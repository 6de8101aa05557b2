"""Summarise `make -C safe_mpc_amd/csrc resource-usage` output: one line per kernel."""
import re
import sys

txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
blocks = re.split(r'remark: [^\n]*?Function Name: ', txt)[1:]
KEYS = [('VGPR', r'VGPRs'), ('AGPR', r'AGPRs'), ('SGPR', r'TotalSGPRs'), ('scratch', r'ScratchSize \[bytes/lane\]'),
        ('occ', r'Occupancy \[waves/SIMD\]'), ('LDS', r'LDS Size \[bytes/block\]')]
for b in blocks:
    name = re.sub(r'^_ZN4smpc\d+', '', b.split(' ')[0])[:26]
    vals = []
    for label, key in KEYS:
        m = re.search(key + r': (\d+)', b)
        vals.append('%s %5s' % (label, m.group(1) if m else '?'))
    print('%-28s %s' % (name, '  '.join(vals)))
