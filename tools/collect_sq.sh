rm -rf gpurun_out/pmc_kernels
PMC_FILTER=k_ bash tools/pmc_kernels.sh 0 1 5 10 14 > gpurun_out/pmc_k.log 2>&1
python tools/pmc_kernels_summary.py gpurun_out/pmc_kernels gpurun_out/r3sq "tools/pmc_kernels.sh 0 1 5 10 14 (cell_family: soft + sampled fwd/bwd of cells 0, 1, 5, 10, 14), final round-3 build"
ls -la gpurun_out/r3sq_sq_counters.json
find gpurun_out/pmc_kernels -name "*.csv" -delete
