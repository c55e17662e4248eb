# ISA reading aid: python tools/diag/isa_seq.py <file.s> <kernel-name regex> [max lines] - the order of vector-memory instructions, waits,
# barriers and matrix instructions of one kernel (device assembly from `hipcc -S --offload-device-only` with the flags of mpg_amd/build.py)
import re,sys
s=open(sys.argv[1]).read()
pat=sys.argv[2]; nmax=int(sys.argv[3]) if len(sys.argv)>3 else 80
ks=re.split(r'\n(?=_Z[\w]+:\s)', s)
for k in ks:
    name=k.split(':')[0]
    if re.search(pat,name):
        lines=k.split('\n')
        print(name, len(lines))
        seq=[]
        for i,l in enumerate(lines):
            t=l.strip()
            if t.startswith(('s_waitcnt','v_mfma','s_barrier','global_load','buffer_load','global_store','scratch_','s_cbranch','.LBB')):
                seq.append((i,t.split('//')[0].strip()[:60]))
        out=[];prev=None;cnt=0
        for i,t in seq:
            key=t.split()[0] if not t.startswith(('s_waitcnt','.LBB','s_cbranch')) else t
            if key==prev: cnt+=1
            else:
                if prev: out.append('%s x%d'%(prev,cnt))
                prev=key;cnt=1
        out.append('%s x%d'%(prev,cnt))
        print('\n'.join(out[:nmax]))
        break
