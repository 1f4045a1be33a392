"""Loads-in-flight / wait structure of one kernel's ISA: for every s_waitcnt vmcnt the number of global loads (L),
stores (S), LDS ops (D) and VALU instructions (V) issued since the previous one, plus barriers and branch labels.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip [-DPIRGPU_LOGN=12] --cuda-device-only -S file.hip -o file.s
       python tools/isa_waits.py <mangled kernel name prefix> [file.s]"""
import re,sys
s=open(sys.argv[2] if len(sys.argv) > 2 else '/tmp/nk12.s').read().split('\n')
name=sys.argv[1]
start=[i for i,l in enumerate(s) if l.startswith(name) and ':' in l][0]
end=next(i for i in range(start,len(s)) if s[i].startswith('.Lfunc_end'))
out=[];cnt=0;ld=0;st=0;lds=0
for l in s[start:end]:
    t=l.strip()
    if t.startswith('global_load') or t.startswith('buffer_load'): ld+=1
    elif t.startswith('global_store'): st+=1
    elif t.startswith('ds_'): lds+=1
    elif t.startswith('s_waitcnt') and 'vmcnt' in t:
        out.append('L%d S%d D%d V%d W[%s]'%(ld,st,lds,cnt,t.split('s_waitcnt')[1].strip())); ld=0;cnt=0;st=0;lds=0
    elif t.startswith('s_barrier'): out.append('BAR')
    elif t.startswith('s_cbranch') or t.startswith('.LBB'): out.append(t[:24])
    elif t.startswith('v_'): cnt+=1
print(end-start); print(' | '.join(out))
