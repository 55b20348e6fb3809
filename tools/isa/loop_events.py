import re,sys,subprocess
src=open(sys.argv[1]).read().split('\n')
pat=sys.argv[2]
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
    if not mf: continue
    start=0
    for j in range(mf[0],-1,-1):
        if '=>This' in body[j]: start=j; break
    s=''
    for j in range(start,len(body)):
        l=body[j].strip()
        if l.startswith('global_load') or l.startswith('buffer_load'): c='L'
        elif l.startswith('s_waitcnt') and 'vmcnt' in l: c='W'+re.search(r'vmcnt\((\d+)\)',l).group(1)
        elif l.startswith('v_mfma'): c='M'
        elif l.startswith('s_barrier'): c='|'
        elif l.startswith('ds_write'): c='s'
        elif l.startswith('s_cbranch'): c='?'
        else: continue
        s+=c
        if c=='|' and j>mf[-1]: break
    s=re.sub(r'M+',lambda m:'M%d '%len(m.group(0)),s); s=re.sub(r's+',lambda m:'s%d '%len(m.group(0)),s)
    print(d, '\n   ', s)
