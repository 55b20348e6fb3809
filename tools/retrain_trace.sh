OUT=$GRAFT_REPO_ROOT/gpurun_out/rt1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/st -o s --output-format csv -- python $GRAFT_REPO_ROOT/tools/retrain_bench.py --steps 10 --warmup 3 > $OUT/run.log 2>&1
python - <<PY
import csv,glob,re,collections
f=glob.glob('$OUT/st/**/s_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    n=re.sub(r'<.*','',r['Name'].replace('void ','').split('(')[0])
    agg[n][0]+=int(r['Calls']); agg[n][1]+=float(r['TotalDurationNs'])
tot=sum(v[1] for v in agg.values())
print('total kernel ms per step', tot/13/1e6)
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:32]:
    print('%-44s %5d/step %8.3f ms/step  avg %7.1f us' % (k[:44], v[0]//13, v[1]/13/1e6, v[1]/v[0]/1e3))
PY
find $OUT -name "*.csv" -size +5M -delete
