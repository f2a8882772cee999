#!/bin/bash
# Sanitizer pass over the host side of the two C-ABI libraries (aux row "sanitizer build"):
# libogpsx.so and libogsqp.so compiled with -fsanitize=address,undefined (host code only: the offload arch has no
# sanitizer support without xnack+), loaded by the test suite through OG_CORE_LIB / OG_SQP_LIB with the ASan
# runtime preloaded.  Usage: tools/sanitize.sh [pytest arguments...]  (default: the CPU suite's native-library
# tests).  The log goes to stdout; "SANITIZER: CLEAN" is printed when neither sanitizer reported anything.
set -u
cd "$(dirname "$0")/.."
OUT=tools/_build/san
mkdir -p "$OUT"
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -Wno-option-ignored \
  -fsanitize=address,undefined -shared-libsan -fno-omit-frame-pointer"
echo "== build (hipcc $FLAGS)"
hipcc $FLAGS opengoddard_amd/csrc/ogpsx_core.hip -o $OUT/libogpsx.so -ldl || exit 2
hipcc $FLAGS opengoddard_amd/csrc/ogsqp.hip -o $OUT/libogsqp.so || exit 2
echo "== run (LD_PRELOAD=$RT)"
ARGS=("$@")
if [ ${#ARGS[@]} -eq 0 ]; then
  ARGS=(tests/test_lgl_and_layout.py tests/test_cabi_and_solve.py tests/test_slsqp_core.py -m "not gpu" -q -x -p no:cacheprovider)
fi
LOG=$OUT/run.log
# the ASan dlopen interceptor loses the RUNPATH of the calling library: torch finds its own libraries through this
TORCH_LIB=$(python -c "import importlib.util, os; print(os.path.join(os.path.dirname(importlib.util.find_spec('torch').origin), 'lib'))" 2>/dev/null)
export LD_LIBRARY_PATH=${TORCH_LIB}${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}
OG_CORE_LIB=$PWD/$OUT/libogpsx.so OG_SQP_LIB=$PWD/$OUT/libogsqp.so \
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:log_path=stderr UBSAN_OPTIONS=print_stacktrace=1 \
  LD_PRELOAD=$RT python -c 'import os, sys, pytest
rc = pytest.main(sys.argv[1:])
sys.stdout.flush(); sys.stderr.flush()
os._exit(int(rc))     # no interpreter teardown: the ROCm ASan runtime CHECK-fails (and hangs) when the HIP runtime unloads
' "${ARGS[@]}" --deselect tests/test_gpu_parity.py::test_hardware_probe 2>&1 | tee $LOG
rc=${PIPESTATUS[0]}
echo "== pytest exit code $rc"
if grep -q "runtime error:\|ERROR: AddressSanitizer\|ERROR: UndefinedBehaviorSanitizer" $LOG; then
  echo "SANITIZER: REPORTS FOUND"; exit 1
fi
[ $rc -eq 0 ] && echo "SANITIZER: CLEAN"
exit $rc
