#!/bin/bash
# round 6, call 27: the committed tree: smoke(), the whole GPU suite
mkdir -p gpurun_out/r06
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" > gpurun_out/r06/gputests.log
tail -3 gpurun_out/r06/gputests.log
