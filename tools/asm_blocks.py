#!/usr/bin/env python3
"""Basic-block instruction counts of one kernel in a hipcc -S dump: asm_blocks.py file.s mangled_prefix [min_depth]"""
import re, sys, collections
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2]); i = s.index('\n', i); j = s.index('s_endpgm', i)
mind = int(sys.argv[3]) if len(sys.argv) > 3 else 2
blocks = collections.OrderedDict(); name = 'entry'; blocks[name] = []
for l in s[i:j].split('\n'):
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.p2align'): continue
    m = re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?$', t)
    if m: name = m.group(1) + ' ' + (m.group(2) or ''); blocks[name] = []; continue
    if t.startswith('.'): continue
    blocks[name].append(t)
tv = 0
for k, v in blocks.items():
    m = re.search(r'Depth=(\d)', k)
    if m and int(m.group(1)) >= mind and len(v) >= 8:
        valu = sum(1 for x in v if x.startswith('v_')); mem = sum(1 for x in v if x.startswith(('global_', 'ds_', 'flat_', 'buffer_')))
        div = sum(1 for x in v if x.startswith('v_div_fixup')); tv += valu
        print('%-52s n=%3d valu=%3d mem=%d div=%d' % (k[:52], len(v), valu, mem, div))
print('sum valu', tv)
