#!/bin/bash
# round 6, call 20: the whole GPU suite on the tree with the new LQ panel
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" > gpurun_out/r06/gputests.log
tail -5 gpurun_out/r06/gputests.log
