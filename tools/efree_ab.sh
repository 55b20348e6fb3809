cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_efree.py -x -q 2>&1 | tail -3
for e in "TFNAS_EFREE_STRIDE1=0" "TFNAS_EFREE_STRIDE1=1 TFNAS_EFREE_RING=0" "TFNAS_EFREE_STRIDE1=1 TFNAS_EFREE_RING=1"; do
  echo "== $e"
  env $e python tools/cell_family.py 1 3 2>&1 | grep -A7 "soft" | grep "cell\|k_dw_fwd\|k_expand_fwd\|k_dw_bwd_data\|k_reduce\|small"
done
