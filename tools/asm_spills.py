#!/usr/bin/env python3
"""Where a kernel's spill code sits in a hipcc -S dump: asm_spills.py file.s mangled_prefix
   -> per loop (header label, depth): instructions, scratch loads / stores, v_readlane / v_writelane (SGPR spills live in VGPR lanes), s_load;
   then the kernel's totals.  Used for profiles/r04_resources.txt."""
import re, sys, collections
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2] ); i = s.index('\n', i); j = s.index('s_endpgm', i)
name, hdr = 'entry', ''
rows = collections.OrderedDict()
def key(h):
    m = re.search(r'Depth=(\d)', h); d = int(m.group(1)) if m else 0
    m = re.search(r'Header=(\w+)', h); return (m.group(1) if m else ('top' if d == 0 else 'header'), d)
cur = ('top', 0)
lastlabel = ''
for l in s[i:j].split('\n'):
    t = l.strip()
    if not t or t.startswith('.p2align'): continue
    m = re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?$', t) or re.match(r'^; %(bb\.\d+):\s*(;.*)?$', t)
    if m:
        h = m.group(2) or ''
        if 'Loop Header' in h:
            d = int(re.search(r'Depth=(\d)', h).group(1)); cur = (m.group(1).replace('.L', ''), d)
        else:
            k = key(h); cur = k if k[0] != 'header' else cur
            if 'Depth' not in h: cur = ('top', 0)
        continue
    if t.startswith(('.', ';')): continue
    op = t.split()[0]
    c = rows.setdefault(cur, collections.Counter())
    c['n'] += 1
    if op.startswith('scratch_load'): c['sld'] += 1
    elif op.startswith('scratch_store'): c['sst'] += 1
    elif op.startswith('v_readlane'): c['rdl'] += 1
    elif op.startswith('v_writelane'): c['wrl'] += 1
    elif op.startswith('s_load'): c['sload'] += 1
tot = collections.Counter()
print('%-14s %5s %6s %8s %8s %8s %8s %7s' % ('loop', 'depth', 'instr', 'scr.load', 'scr.stor', 'readlane', 'writelan', 's_load'))
for (h, d), c in rows.items():
    tot.update(c)
    if c['n'] >= 8: print('%-14s %5d %6d %8d %8d %8d %8d %7d' % (h, d, c['n'], c['sld'], c['sst'], c['rdl'], c['wrl'], c['sload']))
print('%-14s %5s %6d %8d %8d %8d %8d %7d' % ('kernel', '', tot['n'], tot['sld'], tot['sst'], tot['rdl'], tot['wrl'], tot['sload']))
