"""instruction mix of the kernels in a hipcc -S listing: python tools/isa_count.py file.s [name filter]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    c = collections.Counter()
    for ln in body.split('\n'):
        ln = ln.strip()
        if not ln or ln[0] in '.;/' or ln.endswith(':'):
            continue
        c[ln.split()[0]] += 1
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    mul = sum(v for k, v in c.items() if k.startswith(('v_mad_u64', 'v_mul_lo', 'v_mul_hi', 'v_mad_u32')))
    print("%s\n   total %d  valu %d  multiplier %d  s_nop %d  v_mov %d  lds %d  global %d  scratch %d" % (
        name, sum(c.values()), valu, mul, c['s_nop'], c['v_mov_b32'] + c.get('v_mov_b64', 0),
        sum(v for k, v in c.items() if k.startswith('ds_')), sum(v for k, v in c.items() if k.startswith('global_')),
        sum(v for k, v in c.items() if k.startswith('scratch_'))))
    print("   " + ", ".join("%s %d" % kv for kv in c.most_common(30)))
