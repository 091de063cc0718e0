#!/bin/bash
# build_variant.sh <name> <extra flags...>: torch-tts_amd/lib/libttsdec_<name>.so - another build of the same library (e.g. with a
# -DTTSDEC_EXPERIMENT_* switch) for same-box A/B runs through TTSDEC_LIB=...; objects under /tmp, nothing of it is shipped.
set -e
cd "$(dirname "$0")/.."
N=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $*"
mkdir -p /tmp/ttsdec_$N
for s in decode_kernels frame_kernel fused_kernels api encoder vits2 conv256; do
  hipcc $F -c torch-tts_amd/csrc/$s.hip -o /tmp/ttsdec_$N/$s.o 2>/dev/null &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o torch-tts_amd/lib/libttsdec_$N.so /tmp/ttsdec_$N/*.o
echo built libttsdec_$N.so
