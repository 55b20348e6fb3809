#!/bin/bash
# A/B of the bf16-storage build on one GPU box: software vs hardware (v_cvt_pk_bf16_f32) narrowing.  usage: tools/bf16_ab.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO/tf-nas_amd/csrc
for v in "hw|" "sw|-DTFNAS_BF16_SW" "hw2|"; do
  name=${v%%|*}; flags=${v#*|}
  rm -f build/bf16_*.o
  make -j8 ../tfnas_amd/libtfnas_hip_bf16.so EXTRA="$flags" > /tmp/build_$name.log 2>&1 || { echo "BUILD FAILED $name"; tail -5 /tmp/build_$name.log; continue; }
  echo "=== $name ($flags)"
  (cd $REPO && timeout 400 python bench.py --steps 20 --warmup 5 --no-width-sweep --no-dropin --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('bf16',{}).get('value'), d.get('bf16',{}).get('ms_per_step'))")
done
