#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing: isa_mix.py file.s substring"""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
lines = s.split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and pat in l and l.rstrip().endswith(tuple([':'])) or (l.startswith('_Z') and pat in l and ': ' in l))
c = collections.Counter()
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith('s_endpgm'):
        break
    if not t or t[0] in '.;/' or t.split()[0].endswith(':'):
        continue
    c[t.split()[0]] += 1
tot = sum(c.values())
print('total', tot)
groups = collections.Counter()
for k, v in c.items():
    g = ('valu_fma' if k.startswith(('v_fma', 'v_fmac', 'v_pk_fma', 'v_mac')) else 'lds' if k.startswith('ds_') else
         'global' if k.startswith(('global_', 'buffer_', 'flat_')) else 'salu' if k.startswith('s_') else 'valu_other')
    groups[g] += v
print(dict(groups))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    print(f'  {k:30s}{v}')
