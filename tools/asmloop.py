#!/usr/bin/env python3
"""tools/asmloop.py FILE.s KERNEL_SUBSTR -- condensed view of the memory/MFMA/sync stream."""
import re, sys
s = open(sys.argv[1]).read()
sub = sys.argv[2]
m = None
for mm in re.finditer(r'^(\S*%s\S*):[^\n]*\n(.*?)s_endpgm' % re.escape(sub), s, re.S | re.M):
    m = mm
    break
if not m:
    sys.exit("kernel not found")
keys = ('global_load', 's_waitcnt', 's_barrier', 'ds_read', 'v_mfma', 's_cbranch', '.LBB', 'ds_write',
        'buffer_', 'global_store', 'scratch_', 's_setprio', 'global_atomic')
out = [l.strip()[:90] for l in m.group(2).split('\n') if any(k in l for k in keys)]
comp, prev, cnt, prevline = [], None, 0, None
for l in out:
    op = l.split()[0]
    if op == prev and not op.startswith('.LBB') and op != 's_waitcnt':
        cnt += 1
    else:
        if prev:
            comp.append(f"{prevline}  x{cnt}")
        prev, cnt, prevline = op, 1, l
comp.append(f"{prevline}  x{cnt}")
print('\n'.join(comp[: int(sys.argv[3]) if len(sys.argv) > 3 else 200]))
