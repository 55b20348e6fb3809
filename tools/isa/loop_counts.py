import re,sys,subprocess,collections
src=open(sys.argv[1]).read().split('\n'); pat=sys.argv[2]
funcs={}; cur=None
for l in src:
    m=re.match(r'^(_Z\w+):\s', l)
    if m: cur=m.group(1); funcs[cur]=[]
    elif cur is not None:
        funcs[cur].append(l)
        if 's_endpgm' in l: cur=None
for n,body in funcs.items():
    d=subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip().split('(')[0]
    if pat not in d: continue
    mf=[j for j,l in enumerate(body) if l.strip().startswith('v_mfma')]
    start=0
    for j in range(mf[0],-1,-1):
        if '=>This' in body[j]: start=j; break
    # end: first s_barrier after last mfma of the loop
    end=mf[-1]
    for j in range(mf[-1],len(body)):
        if body[j].strip().startswith('s_barrier'): end=j; break
    cnt=collections.Counter()
    for l in body[start:end+1]:
        t=l.strip()
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
        op=t.split()[0]
        if op.startswith('v_mfma'): k='mfma'
        elif op.startswith('v_'): k='valu'
        elif op.startswith('s_'): k='salu'
        elif op.startswith('ds_'): k='lds'
        elif op.startswith('global_') or op.startswith('buffer_'): k='vmem'
        else: k='other'
        cnt[k]+=1
    print(d, dict(cnt))
