#!/usr/bin/env python3
"""usage: tools/peel_phase_count.py <isa.s from tools/peel_isa.sh> - static instruction counts between the phase markers of the chain step."""
import sys,re
lines=open(sys.argv[1]).read().split('\n')
marks=[(i,l.split('C2AMARK')[1].strip()) for i,l in enumerate(lines) if 'C2AMARK' in l]
def classify(op):
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op.startswith('v_readlane') or op.startswith('v_writelane'): return 'lane'
    if op.startswith('global_') or op.startswith('flat_') or op.startswith('buffer_') or op.startswith('s_atomic') or op.startswith('s_load'): return 'mem'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_'): return 'salu'
    return 'other'
for k in range(len(marks)-1):
    a,na=marks[k]; b,nb=marks[k+1]
    c={}
    n=0
    for l in lines[a+1:b]:
        t=l.strip()
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
        op=t.split()[0]; cl=classify(op); c[cl]=c.get(cl,0)+1; n+=1
    print(f"{na} -> {nb}: {n:5d}  {c}")
