# ISA reading aid: python tools/diag/isa_waits.py <file.s> <kernel-name regex> - what sits between a 256 KB weight-image load (a run of >= 16
# global_load_dwordx4) and the first f16 matrix instruction behind it: vector-memory instructions, vmcnt waits, barriers
import re,sys
s=open(sys.argv[1]).read()
pat=sys.argv[2]
ks=re.split(r'\n(?=_Z[\w]+:\s)', s)
for k in ks:
    name=k.split(':')[0]
    if not re.search(pat,name): continue
    lines=[l.split(';')[0].strip() for l in k.split('\n')]
    lines=[l for l in lines if l]
    print('==',name[:110])
    # find runs of global_load_dwordx4 of len>=16 (allow few interleaved non-load instrs)
    i=0;n=len(lines)
    while i<n:
        if lines[i].startswith('global_load_dwordx4'):
            j=i;cnt=0;last=i
            while j<n and j-last<12:
                if lines[j].startswith('global_load_dwordx4'): cnt+=1;last=j
                j+=1
            if cnt>=16:
                # report VM ops & waits until first f16 mfma
                out=[];m=last+1
                while m<n and not lines[m].startswith('v_mfma_f32_16x16x32'):
                    t=lines[m]
                    if t.startswith(('global_','buffer_','scratch_')): out.append(t.split()[0])
                    elif 'vmcnt' in t: out.append(re.search(r'vmcnt\(\d+\)',t).group(0))
                    elif t.startswith('s_barrier'): out.append('BAR')
                    elif t.startswith('v_mfma'): out.append('mfma32')
                    elif t.startswith(('.LBB','s_cbranch')): out.append(t.split()[0][:12])
                    m+=1
                print(' image x%d at line %d; until first f16 mfma (%d instrs):'%(cnt,i,m-last), ' '.join(out)[:600])
                i=m
                continue
            i=last+1
        else: i+=1
