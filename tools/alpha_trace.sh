OUT=$GRAFT_REPO_ROOT/gpurun_out/al1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/tr -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dropin --no-width-sweep --no-retrain > $OUT/run.log 2>&1
TR=$(find $OUT/tr -name "*_kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/alpha_cells.py $TR > $OUT/alpha_cells.txt 2>&1
find $OUT -name "*.csv" -delete
cat $OUT/alpha_cells.txt | cut -c1-250
