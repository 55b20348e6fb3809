#!/bin/bash
# GPU box: round-6 A/B of the w-step's serial sections, alternating runs on ONE box:
#   fused tail (tail.py), direct + spread stem weight gradients, one-launch Gram operator (k_gram1)
TAG=${1:-r6ab}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
AB_STEPS=${AB_STEPS:-12} bash $REPO/tools/ab_bench.sh $TAG \
  "A=base" "TFNAS_FUSED_TAIL=0 TFNAS_STEM_DIRECT=0 TFNAS_GRAM=2" \
  "A=base" "TFNAS_FUSED_TAIL=0" "TFNAS_STEM_DIRECT=0" "TFNAS_GRAM=2" "TFNAS_STEM_SPREAD=0" \
  "A=base" "TFNAS_FUSED_TAIL=0 TFNAS_STEM_DIRECT=0 TFNAS_GRAM=2"
