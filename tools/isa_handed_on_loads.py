#!/usr/bin/env python3
"""Loops whose body moves a register that a load of the SAME body wrote (prefetched operands handed on in registers: the move waits for the load)."""
import re, sys
def regs(tok):
    tok=tok.strip().rstrip(',')
    m=re.match(r'v\[(\d+):(\d+)\]$',tok)
    if m: return set(range(int(m.group(1)),int(m.group(2))+1))
    m=re.match(r'v(\d+)$',tok)
    if m: return {int(m.group(1))}
    return set()
def scan(path):
    fn=None; ins=[]  # (addr, text)
    out=[]
    def flush():
        if not ins: return
        addr_idx={a:i for i,(a,_) in enumerate(ins)}
        loops=[]
        for i,(a,t) in enumerate(ins):
            m=re.match(r'\s*s_cbranch_\w+\s+(\d+)',t) or re.match(r'\s*s_branch\s+(\d+)',t)
            if m:
                off=int(m.group(1))
                if off>=32768: off-=65536
                tgt=a+4+4*off
                if off<0 and tgt in addr_idx: loops.append((addr_idx[tgt],i))
        seen=set()
        for lo,hi in loops:
            if (lo,hi) in seen or hi-lo>1500: continue
            seen.add((lo,hi))
            loaded={}  # reg -> index of load
            hits=[]
            for k in range(lo,hi+1):
                t=ins[k][1].strip()
                op=t.split()[0] if t else ''
                args=t[len(op):]
                if op.startswith(('global_load','flat_load','buffer_load','ds_read','scratch_load')):
                    dst=args.split(',')[0]
                    for r in regs(dst): loaded[r]=k
                elif op.startswith(('v_mov_b32','v_mov_b64')) and 'dpp' not in t:
                    parts=[p.strip() for p in args.split(',')]
                    if len(parts)>=2:
                        src=regs(parts[1]); 
                        if src and all(r in loaded for r in src):
                            hits.append((k-lo, k-max(loaded[r] for r in src), t))
                        for r in regs(parts[0]): loaded.pop(r,None)
                else:
                    # any other write to a register clears its loaded state (first operand as destination, heuristically)
                    if op.startswith('v_') :
                        d=args.split(',')[0]
                        for r in regs(d): loaded.pop(r,None)
            if hits:
                out.append((fn,hi-lo+1,hits))
    cur_addr=None
    for line in open(path):
        m=re.match(r'^([0-9a-f]+) <(.+)>:',line)
        if m:
            flush(); ins=[]; fn=m.group(2); continue
        m=re.match(r'^\s+(.*?)\s*//\s*([0-9A-Fa-f]+):',line)
        if m:
            ins.append((int(m.group(2),16),m.group(1)))
    flush()
    return out
for p in sys.argv[1:]:
    res=scan(p)
    print('##',p,len(res),'loops with handed-on loads')
    for fn,n,hits in sorted(res,key=lambda r:-len(r[2]))[:25]:
        print(f'  {fn[:90]:90s} loop of {n:4d} instrs: {len(hits)} moves of loaded registers; distances load->move: {sorted(h[1] for h in hits)[:8]}')
