#!/bin/bash
# loopsum.sh <file.s> <kernel-name-substring>: instruction-class sequence of the kernel's largest MFMA loop
awk -v k="$2" 'index($0,k) && /^_Z.*:/{p=1} p{print} p&&/s_endpgm/{exit}' "$1" > /tmp/k.s
python3 - <<'PY'
import re
lines=open('/tmp/k.s').read().split('\n')
# find basic blocks; pick the one with most MFMAs
blocks=[];cur=[]
for l in lines:
    if re.match(r'^\.LBB\d+_\d+:',l):
        blocks.append(cur);cur=[]
    cur.append(l)
blocks.append(cur)
b=max(blocks,key=lambda b:sum('v_mfma' in l for l in b))
def cls(l):
    t=l.split()
    if not t or t[0].startswith(';') : return None
    o=t[0]
    if o.startswith('v_mfma'): return 'M'
    if o.startswith('global_load_lds'): return 'DMA'
    if o.startswith('global_load'): return 'LD'
    if o.startswith('ds_read'): return 'r'
    if o.startswith('ds_write'): return 'W'
    if o=='s_waitcnt': return 'w(%s)'%' '.join(t[1:])
    if o=='s_barrier': return 'BAR'
    if o.startswith('v_'): return 'v'
    return None
out=[];last=None;n=0
for l in b:
    c=cls(l)
    if c is None: continue
    if c==last: n+=1
    else:
        if last: out.append('%s%s'%(n if n>1 else '',last))
        last=c;n=1
out.append('%s%s'%(n if n>1 else '',last))
print(' '.join(out))
PY
