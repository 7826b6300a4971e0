#!/usr/bin/env python3
"""Instruction mix of the innermost stepping loop of a walk kernel in a hipcc -S dump:
   asm_loop.py file.s mangled_prefix    -> per basic block of the deepest loop: VALU / SALU / branch / memory counts; blocks holding
   v_div_fixup (the IEEE-division fallbacks a wave normally skips) are listed but left out of the hot-path total."""
import re, sys, collections
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2]); i = s.index('\n', i); j = s.index('s_endpgm', i)
blocks = collections.OrderedDict(); name = 'entry'; blocks[name] = []
hdr = {}
for l in s[i:j].split('\n'):
    t = l.strip()
    if not t or t.startswith(';') and 'bb.' not in t or t.startswith('.p2align'): continue
    m = re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?$', t) or re.match(r'^; %(bb\.\d+):\s*(;.*)?$', t)
    if m: name = m.group(1); blocks[name] = []; hdr[name] = m.group(2) or ''; continue
    if t.startswith('.') or t.startswith(';'): continue
    blocks[name].append(t.split()[0])
depth = {k: int(re.search(r'Depth=(\d)', v).group(1)) for k, v in hdr.items() if re.search(r'Depth=(\d)', v)}
deep = max(depth.values())
heads = collections.Counter(re.search(r'Header=(\w+)', hdr[k]).group(1) for k in depth if depth[k] == deep and 'Header=' in hdr[k])
tot = collections.Counter()
# the stepping loop = the deepest loop with the most instructions
best = max(heads, key=lambda h: sum(len(blocks[k]) for k in depth if depth[k] == deep and ('Header=' + h) in hdr[k]))
for k in blocks:
    if depth.get(k) != deep or (('Header=' + best) not in hdr[k] and not k.endswith(best[2:])): continue
    ins = blocks[k]
    c = collections.Counter('valu' if x.startswith('v_') else 'branch' if x.startswith(('s_cbranch', 's_branch')) else 'wait' if x.startswith(('s_waitcnt', 's_nop')) else 'salu' if x.startswith('s_') else 'mem' for x in ins)
    slow = any(x.startswith('v_div_fixup') for x in ins)
    print('%-12s n=%3d valu=%3d salu=%3d branch=%2d mem=%2d wait=%2d %s' % (k, len(ins), c['valu'], c['salu'], c['branch'], c['mem'], c['wait'], 'SLOW PATH (not in total)' if slow else ''))
    if not slow: tot.update(c)
print('hot path of one trip:', dict(tot), 'all', sum(tot.values()))
