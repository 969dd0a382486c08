#!/usr/bin/env python3
"""Loops in which a full wait (s_waitcnt vmcnt(0) / lgkmcnt(0)) comes within a few instructions of a load of the same counter: the wave asks for something and
waits for it on the spot, nothing overlaps.  Input: `llvm-objdump -d` of a code object.   python tools/isa_immediate_waits.py file.s [function-substring] [window]"""
import re
import sys

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
WIN = int(sys.argv[3]) if len(sys.argv) > 3 else 6
funcs, fn = {}, None
for line in open(path):
    m = re.match(r'^([0-9a-f]+) <(.+)>:', line)
    if m:
        fn = m.group(2); funcs[fn] = []; continue
    m = re.match(r'^\s+(.*?)\s*//\s*([0-9A-Fa-f]+):', line)
    if m and fn:
        funcs[fn].append((int(m.group(2), 16), m.group(1).strip()))
for fn, ins in funcs.items():
    if flt not in fn:
        continue
    idx = {a: i for i, (a, _) in enumerate(ins)}
    loops = set()
    for i, (a, t) in enumerate(ins):
        m = re.match(r's_cbranch_\w+\s+(\d+)', t) or re.match(r's_branch\s+(\d+)', t)
        if m:
            off = int(m.group(1))
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if off < 0 and tgt in idx:
                loops.add((idx[tgt], i))
    rep = []
    for lo, hi in sorted(loops):
        if hi - lo > 3000:
            continue
        hits = []
        for k in range(lo, hi + 1):
            t = ins[k][1]
            if not t.startswith('s_waitcnt'):
                continue
            vm0 = 'vmcnt(0)' in t
            lg0 = 'lgkmcnt(0)' in t
            if not (vm0 or lg0):
                continue
            for j in range(max(lo, k - WIN), k):
                u = ins[j][1]
                if (vm0 and u.startswith(('global_load', 'flat_load', 'buffer_load', 'scratch_load'))) or (lg0 and u.startswith(('ds_read', 'ds_bpermute', 's_load'))):
                    hits.append((k - lo, u.split()[0]))
                    break
        nm = sum('v_mfma' in ins[k][1] for k in range(lo, hi + 1))
        if hits:
            rep.append((hi - lo + 1, nm, hits))
    if rep:
        print(fn[:110])
        for n, nm, hits in rep:
            print(f'    loop of {n:5d} instructions ({nm} matrix ops): {len(hits)} immediate full waits at {[h[0] for h in hits][:12]} after {sorted(set(h[1] for h in hits))}')
