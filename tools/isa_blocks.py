#!/usr/bin/env python3
"""tools/isa_blocks.py FILE.s KERNEL_SUBSTR [--dump LABEL] -- per basic block of a kernel: MFMA / DS / VMEM / VALU / SALU / s_nop
counts (the loop body is the block with the most MFMAs), or the text of one block."""
import re
import sys

s = open(sys.argv[1]).read()
sub = sys.argv[2]
m = re.search(r'^(\S*%s\S*):[^\n]*\n(.*?)s_endpgm' % re.escape(sub), s, re.S | re.M)
if not m:
    sys.exit("kernel not found")
blocks, cur, name = [], [], 'entry'
for l in m.group(2).split('\n'):
    if re.match(r'^\.LBB\S+:', l):
        blocks.append((name, cur))
        name, cur = l.split(':')[0], []
    elif l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.'):
        cur.append(l.split(';')[0].rstrip())
blocks.append((name, cur))
if '--dump' in sys.argv:
    want = sys.argv[sys.argv.index('--dump') + 1]
    for n, b in blocks:
        if n == want:
            print('\n'.join(b))
    sys.exit(0)
for n, b in blocks:
    c = lambda pat: sum(bool(re.match(r'\s+' + pat, x)) for x in b)      # noqa: E731
    if c('v_mfma'):
        print(f"{n:12s} insts {len(b):4d} mfma {c('v_mfma'):3d} ds_read {c('ds_read'):3d} ds_write {c('ds_write'):2d} "
              f"vmem {c('(global|buffer)_'):3d} accvgpr {c('v_accvgpr'):3d} valu {c('v_(?!mfma|accvgpr)'):3d} "
              f"salu {c('s_(?!waitcnt|barrier|nop)'):3d} nop {c('s_nop'):2d} wait {c('s_waitcnt'):2d} bar {c('s_barrier'):2d}")
