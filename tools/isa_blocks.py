#!/usr/bin/env python3
"""Per-basic-block instruction census of one kernel in a hipcc -S listing (where are the MFMAs, the spills, the waits).
    python tools/isa_blocks.py file.s <mangled-name-substring> [--dump LABEL]"""
import re
import sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(pat) + r'\S*):.*?^\.Lfunc_end\d+:', s, re.S | re.M)
f = m.group(0)
print(m.group(1))
if len(sys.argv) > 4 and sys.argv[3] == '--dump':
    lab = sys.argv[4]
    i = f.index(lab + ':')
    nxt = re.search(r'^\.LBB\d+_\d+:', f[i + len(lab) + 1:], re.M)
    j = i + len(lab) + 1 + (nxt.start() if nxt else 4000)
    for l in f[i:j].split('\n'):
        t = l.strip()
        if t and not t.startswith(';'):
            print(t[:120])
    sys.exit(0)
blocks, cur = [], None
for l in f.split('\n'):
    if re.match(r'^\.LBB\d+_\d+:', l) or cur is None:
        cur = {'label': l.split(':')[0], 'mfma': 0, 'scratch': 0, 'ds': 0, 'wait': 0, 'valu': 0, 'vmem': 0, 'n': 0}
        blocks.append(cur)
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    cur['n'] += 1
    if t.startswith('v_mfma'): cur['mfma'] += 1
    elif t.startswith('scratch_'): cur['scratch'] += 1
    elif t.startswith('ds_'): cur['ds'] += 1
    elif t.startswith('s_waitcnt'): cur['wait'] += 1
    elif t.startswith('global_') or t.startswith('buffer_'): cur['vmem'] += 1
    elif t.startswith('v_'): cur['valu'] += 1
for b in blocks:
    if b['mfma'] or b['scratch'] or b['n'] > 40:
        print(b)
