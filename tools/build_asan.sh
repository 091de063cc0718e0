#!/bin/bash
# Host-side sanitizer build of the C-ABI layer: the same sources with AddressSanitizer + UBSan on the HOST
# code only (-fno-gpu-sanitize: device ASan needs xnack+, not available on this pool), into torch-tts_amd/lib/libttsdec_asan.so;
# then the host-logic tests run against it with the ASan runtime preloaded.  CPU box only (no GPU needed: the tests there make
# no compute calls).
#   tools/build_asan.sh [pytest args...]      default: tests/test_host_logic.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/torch-tts_amd/csrc
OUT=$ROOT/torch-tts_amd/lib
mkdir -p "$OUT/asan"
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -shared-libsan"
pids=()
for f in decode_kernels frame_kernel fused_kernels api encoder vits2; do
  /opt/rocm/bin/hipcc $FLAGS -c "$CSRC/$f.hip" -o "$OUT/asan/$f.o" & pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -o "$OUT/libttsdec_asan.so" "$OUT"/asan/*.o
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
echo "built $OUT/libttsdec_asan.so; runtime $RT"
cd "$ROOT"
ARGS=("$@"); [ ${#ARGS[@]} -eq 0 ] && ARGS=(tests/test_host_logic.py)
# (detect_leaks=0: the interpreter itself is not leak-clean; UBSan reports abort the run)
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  TTSDEC_LIB="$OUT/libttsdec_asan.so" python -m pytest "${ARGS[@]}" -x -q -m "not gpu"
