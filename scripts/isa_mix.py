"""Instruction mix per kernel of a device assembly file (hipcc -S --cuda-device-only): scripts/isa_mix.py file.s"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):\s*;.*?\n(.*?)^\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    ins = []
    for l in body.split('\n'):
        if not l.startswith('\t'): continue
        t = l.split()
        if not t or t[0].startswith('.') or t[0].startswith(';'): continue
        ins.append(t[0])
    v = [i for i in ins if i.startswith('v_')]
    c = Counter(v)
    g = lambda *ks: sum(n for k, n in c.items() if any(k.startswith(p) for p in ks))
    print("%-60s total %5d valu %5d pk %4d mul %4d add/sub %4d fma %4d rcp %3d sqrt %3d div_scale %3d ds %4d" % (
        name[:60], len(ins), len(v), g('v_pk'), g('v_mul_f32'), g('v_add_f32', 'v_sub_f32'), g('v_fma_f32', 'v_fmac_f32'), g('v_rcp_f32'),
        g('v_sqrt_f32'), g('v_div_scale_f32'), sum(1 for i in ins if i.startswith('ds_'))))
