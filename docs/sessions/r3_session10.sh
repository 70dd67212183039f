#!/bin/bash
# Round-3 GPU session 10: what binds the diffuse convolution — hardware counters of the load-time kernels (profiles/r3i_pmc_conv.txt).
# (The same session also timed patch-shaped / lock-stepped workgroups and, later, a software-pipelined tap loop: template variants that are not kept
#  in the source — profiles/r3i_diffuse_block_shapes.jsonl, profiles/r3i_conv_kernels.md.)
O=gpurun_out/r3i; mkdir -p $O
bash scripts/pmc_conv.sh fast > $O/pmc_conv_fast.txt 2>&1; cat $O/pmc_conv_fast.txt
