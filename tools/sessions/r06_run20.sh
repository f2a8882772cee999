#!/bin/bash
# round 6, call 24: the number of head workgroups of the look-ahead launch (compile-time OGSQP_LQ_HEADS), C3's first 10 iterations
mkdir -p gpurun_out/r06 tools/_build
R=${GRAFT_REPO_ROOT:-/root/repo}
for h in 4 6 12 16; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -DOGSQP_LQ_HEADS=$h \
      $R/opengoddard_amd/csrc/ogsqp.hip -o $R/tools/_build/libogsqp_heads$h.so -ldl 2>/dev/null &
done
wait
for h in 8 4 6 12 16; do
  lib=$R/opengoddard_amd/lib/libogsqp.so; [ $h != 8 ] && lib=$R/tools/_build/libogsqp_heads$h.so
  echo "== heads $h"
  OG_SQP_LIB=$lib bash tools/sqp_kstats.sh polar_tsto 10 r06_run20_heads$h 2>&1 | grep "k_lq_step16<12, 5>\|k_lq_step16<8, 3>\|k_lq_step16<4, 2>\|polar_tsto hip" | cut -c1-220
done
