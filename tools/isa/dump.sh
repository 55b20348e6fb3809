#!/bin/bash
# Device ISA of one kernel source of the product build:  tools/isa/dump.sh gemm_kernels /tmp/gemm.s
# then e.g.  python tools/isa/loop_events.py /tmp/gemm.s "k_project_wgrad<2, 0>"   (order of loads / vmcnt waits / MFMAs / barriers
#                                                                                  in the main loop: L W<n> M<count> s<count> |)
#            python tools/isa/loop_counts.py /tmp/gemm.s "k_project_fwd<4, 1>"     (VALU / SALU / LDS / VMEM / MFMA per iteration)
#            python tools/isa/loop_hist.py   /tmp/gemm.s "k_project_fwd<4, 1>" 40  (opcode histogram of the loop)
#            python tools/isa/regs.py        /tmp/gemm.s "project_fwd|expand"      (VGPRs and scratch bytes per kernel)
cd "$(dirname "$0")/../../tf-nas_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics \
  -ffp-contract=fast -I../../include -I. $EXTRA -S --cuda-device-only -o "$2" "$1.hip"
