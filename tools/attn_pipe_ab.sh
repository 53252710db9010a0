#!/bin/bash
# The twelve-row 64 x 64 self-attention launch under the product kernel (attn_pipe = 0) and the two half-tile software-pipelined forms
# (1: compiler-scheduled, 2: sched_group_barrier placement).  Needs the ablation library:
#   python -m pnpinversion_amd.build --ablations && PNPI_LIBRARY=pnpinversion_amd/csrc/libpnpi_ablations.so bash tools/attn_pipe_ab.sh
for r in 1 2; do for p in 0 1 2; do PNPI_TUNE=attn_pipe=$p timeout 120 python tools/attn_probe.py 2>&1 | tail -1 | sed "s/^/attn_pipe=$p /"; done; done
