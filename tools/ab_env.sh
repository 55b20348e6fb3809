# usage: tools/ab_env.sh VAR v1 v2 ...   -> bench (30 pairs) with VAR set to each value
V=$1; shift
for x in "$@"; do
  env $V=$x timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-dropin --no-width-sweep --no-bf16 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$x', d['value'], d['ms_per_step'], d['w_step_ms'], d['a_step_ms'])"
done
