# per-kernel averages of the depthwise weight-gradient kernels + pair time (run on the GPU box)
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/trs
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/trs -o s --output-format csv -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-dropin --no-width-sweep --no-bf16 > /tmp/trs.log 2>&1
cd $R; F=$(find /tmp/trs -name "*kernel_stats.csv" | head -1)
grep -i "k_dw_wgrad\|k_dws_wgrad" $F | awk -F'",' '{split($2,a,","); printf "%-40s calls %5d avg %9.1f us\n", substr($1,7,34), a[1], a[3]/1000}'
timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-dropin --no-width-sweep --no-bf16 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['w_step_ms'], d['a_step_ms'], d['kernel_ms_per_pair'].get('k_dw_wgrad'))"
