#!/bin/bash
# Debug build of libttsdec.so whose buffer-descriptor loaders (csrc/gemm_tile.h kBufDma) check every lane's first address of
# every K segment against the per-lane pointer form before issuing anything: a disagreeing lane is taken out of range and
# printed, never dereferenced.  Output: torch-tts_amd/lib/libttsdec_check.so (use with TTSDEC_LIB=...).
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DTTSDEC_CHECK_DMA"
mkdir -p /tmp/ttsdec_check
for s in decode_kernels frame_kernel fused_kernels api encoder vits2; do
  hipcc $F -c torch-tts_amd/csrc/$s.hip -o /tmp/ttsdec_check/$s.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o torch-tts_amd/lib/libttsdec_check.so /tmp/ttsdec_check/*.o
echo built torch-tts_amd/lib/libttsdec_check.so
