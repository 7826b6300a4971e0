#!/usr/bin/env python3
"""Instruction mix of one loop of a hipcc -S dump by issue class: asm_mix.py file.s mangled_kernel_prefix loop_header_label (e.g. BB3_908)
   -> opcode histogram of the blocks of that loop (nested loops included) and the totals per class.  Classes as tools/micro/valu_rates.hip measures them on gfx950
   (profiles/r04_valu_rates.txt): full rate (v_fma / add / sub / mul / mov / and / or / xor / add_u32 / cndmask ...), half rate (v_min / max / med3 / cmp / shifts / mbcnt / 24-bit and
   32-bit integer multiplies / conversions / lane reads), quarter rate (v_rcp / sqrt / rsq / exp / log / sin / cos)."""
import re, sys, collections
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2]); j = s.index('s_endpgm', i)
hdr = sys.argv[3]
HALF = ('v_max', 'v_min', 'v_med3', 'v_cmp', 'v_lshl', 'v_lshr', 'v_ashr', 'v_mbcnt', 'v_mul_lo', 'v_mul_hi', 'v_mad_u', 'v_mad_i', 'v_bfe', 'v_cvt', 'v_mul_u32_u24', 'v_readlane', 'v_writelane', 'v_readfirstlane', 'v_div_')
QUARTER = ('v_rcp', 'v_sqrt', 'v_rsq', 'v_exp', 'v_log', 'v_sin', 'v_cos')
inner = {hdr}          # headers of loops nested in it
inloop = False
cnt = collections.Counter()
for l in s[i:j].split('\n'):
    t = l.strip()
    m = re.match(r'^\.L(BB\d+_\d+):\s*(;.*)?$', t)
    if m:
        c = m.group(2) or ''
        h = re.search(r'Header=(BB\d+_\d+)', c)
        p = re.search(r'Parent Loop (BB\d+_\d+)', c)
        if m.group(1) == hdr: inloop = True
        elif 'Loop Header' in c and re.findall(r'Parent Loop (BB\d+_\d+)', l) and any(x in inner for x in re.findall(r'Parent Loop (BB\d+_\d+)', l)): inner.add(m.group(1)); inloop = True
        elif h and h.group(1) in inner: inloop = True
        else: inloop = False
        continue
    if not inloop or not t or t.startswith(('.', ';')): continue
    cnt[t.split()[0]] += 1
cls = collections.Counter()
for op, c in cnt.items():
    if op.startswith('s_'): cls['scalar'] += c
    elif op.startswith(('ds_', 'global_', 'flat_', 'scratch_', 'buffer_')): cls['memory'] += c
    elif op.startswith(QUARTER): cls['quarter'] += c
    elif op.startswith(HALF): cls['half'] += c
    elif op.startswith('v_'): cls['full'] += c
    else: cls['other'] += c
valu = cls['full'] + cls['half'] + cls['quarter']
print('loop %s: %d instructions; vector %d = full rate %d + half rate %d + quarter rate %d; scalar %d, memory %d' % (hdr, sum(cnt.values()), valu, cls['full'], cls['half'], cls['quarter'], cls['scalar'], cls['memory']))
if valu:
    cyc = 2 * cls['full'] + 4 * cls['half'] + 8 * cls['quarter']
    print('issue cycles of its vector instructions at 2 / 4 / 8 cycles per class: %d = %.2f per instruction -> the issue peak for THIS mix is 1 per %.2f cycles, %.2f of the 1-per-2 v_fma_f32 peak' % (cyc, cyc / valu, cyc / valu, 2.0 * valu / cyc))
for op, c in cnt.most_common(int(sys.argv[4]) if len(sys.argv) > 4 else 40):
    print('  %-26s %4d' % (op, c))
