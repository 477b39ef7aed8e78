"""register / spill summary per kernel of a hipcc -S listing: python tools/isa_regs.py file.s [name filter ...]"""
import re
import sys
txt = open(sys.argv[1]).read()
flts = sys.argv[2:]
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size:\s+\d+', txt, re.S):
    blk = m.group(0)
    g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk).group(1)
    name = g('name')
    if flts and not any(f in name for f in flts):
        continue
    print("%-90s vgpr %4s spill %3s sgpr %4s scratch %5s" % (name[:90], g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_count'), g('private_segment_fixed_size')))
