#!/bin/bash
# Builds the library's HOST code under a sanitizer (the device code is untouched: GPU
# sanitizers are not available here) together with tests/context_stress, into build/<kind>/.
#   tests/tools/build_sanitized.sh asan   ->  -fsanitize=address,undefined
#   tests/tools/build_sanitized.sh tsan   ->  -fsanitize=thread   (builds; but the ROCm runtime finds no
#                                              device in a process laid out by ThreadSanitizer, so it cannot run)
# Run on a GPU box, from the repository root:
#   ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 build/asan/context_stress 4 10
#   ... build/asan/cereal_box_matcher; build/asan/headless_pipeline --frames 40 --lockstep;
#   ... build/asan/multi_gpu_frames --devices 0,0,0 --frames 24 --rows 700 --db 20000
#   TSAN_OPTIONS=report_signal_unsafe=0 build/tsan/context_stress 4 6
set -e
KIND=${1:-asan}
case $KIND in
  asan) SAN="address,undefined" ;;
  tsan) SAN="thread" ;;
  *) echo "usage: $0 asan|tsan"; exit 2 ;;
esac
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/build/$KIND
mkdir -p $OUT
SRC=$ROOT/poseestimation_amd/csrc
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
COMMON="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$SRC -Wall -Wno-unused-result -ffp-contract=off"
for f in knn_hamming knn_mfma posehip_api ransac_epnp; do
  EXTRA=""
  [ $f = ransac_epnp ] && EXTRA="-mllvm -amdgpu-sched-strategy=max-ilp"
  [ $f = knn_mfma ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form"
  # device code at the product's optimisation level (-O3 via -Xarch_device), host code instrumented
  $HIPCC $COMMON -Xarch_device -O3 -Xarch_host -fsanitize=$SAN -Xarch_host -fno-omit-frame-pointer $EXTRA \
      -c $SRC/$f.hip -o $OUT/$f.o &
done
wait
$HIPCC -shared -fPIC --offload-arch=gfx950 -fsanitize=$SAN -fno-gpu-sanitize $OUT/*.o -o $OUT/libposehip.so
${CLANGXX:-/opt/rocm/lib/llvm/bin/clang++} -std=c++17 -O1 -g -fsanitize=$SAN -fno-omit-frame-pointer -pthread $ROOT/tests/context_stress.cpp \
    -I$ROOT/include -L$OUT -lposehip -Wl,-rpath,'$ORIGIN' -Wl,-rpath-link,/opt/rocm/lib -o $OUT/context_stress
# the C++ examples against the instrumented library: every host entry point they use (matcher,
# filters, lifting, solvePnPRansac single and batched flows, contexts) runs under the sanitizer
for ex in cereal_box_matcher headless_pipeline multi_gpu_frames; do
  ${CLANGXX:-/opt/rocm/lib/llvm/bin/clang++} -std=c++17 -O1 -g -fsanitize=$SAN -fno-omit-frame-pointer -pthread \
      $ROOT/examples/$ex.cpp -I$ROOT/include -L$OUT -lposehip -Wl,-rpath,'$ORIGIN' \
      -Wl,-rpath-link,/opt/rocm/lib -o $OUT/$ex
done
ls -la $OUT/libposehip.so $OUT/context_stress $OUT/headless_pipeline
