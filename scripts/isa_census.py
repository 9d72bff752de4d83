import sys,re,collections
lines=open(sys.argv[1]).read().split('\n')
def census(a,b,title):
    c=collections.Counter()
    for l in lines[a-1:b]:
        l=l.strip()
        if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'): continue
        op=l.split()[0]
        if op.startswith('v_mfma'): k='mfma'
        elif op.startswith('v_'): k='valu'
        elif op.startswith('ds_'): k='lds'
        elif op.startswith('buffer_') or op.startswith('global_') or op.startswith('scratch_') or op.startswith('flat_'): k='vmem'
        elif op.startswith('s_waitcnt'): k='waitcnt'
        elif op.startswith('s_barrier'): k='barrier'
        elif op.startswith('s_cbranch') or op.startswith('s_branch'): k='branch'
        elif op.startswith('s_load'): k='smem'
        elif op.startswith('s_nop'): k='nop'
        elif op.startswith('s_'): k='salu'
        else: k='other'
        c[k]+=1
    print(title, dict(c))
for spec in sys.argv[2:]:
    t,a,b=spec.split(':'); census(int(a),int(b),t)
