import re,sys,subprocess
txt=open(sys.argv[1]).read()
pat=sys.argv[2] if len(sys.argv)>2 else ''
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
    name=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.split('(')[0].strip()
    if pat and not re.search(pat,name): continue
    body=m.group(2)
    nv=re.search(r'\.amdhsa_next_free_vgpr (\d+)',body).group(1)
    acc=re.search(r'\.amdhsa_accum_offset (\d+)',body)
    print('%-40s vgpr %s accum_off %s'%(name[:40],nv,acc.group(1) if acc else '-'),end=' ')
    mm=re.search(r'; ScratchSize: (\d+)', txt[txt.find(m.group(1)+':'):txt.find(m.group(1)+':')+400000])
    print('scratch',mm.group(1) if mm else '?')
