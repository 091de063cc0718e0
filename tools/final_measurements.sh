#!/bin/bash
# Runs on the GPU box (gpurun): the round's final numbers on the frozen sources, as tools/gpu_steps.sh sessions (a step that is
# killed at its limit ends the session).  Every rocprofv3 call has the program itself after "--" and collects counters without
# any trace domain but --kernel-trace.
#   tools/final_measurements.sh bench <session>     the bench lines
#   tools/final_measurements.sh prof <session>      kernel traces and PMC passes
S=${2:-r4_final}
O=gpurun_out/$S
if [ "$1" = bench ]; then
  exec tools/gpu_steps.sh $S \
    "bench_default|600|python bench.py" \
    "bench_batch64|200|python bench.py --batch 64 --no-cpu-baseline --no-extra-legs" \
    "bench_batch1|200|python bench.py --batch 1 --no-cpu-baseline --no-extra-legs" \
    "bench_split|300|python bench.py --precision split_f16 --postnet split_f16 --no-cpu-baseline --no-extra-legs" \
    "bench_postnet_bf16|200|python bench.py --postnet bf16 --no-cpu-baseline --no-extra-legs" \
    "bench_masks|300|python bench.py --dropout masks --no-cpu-baseline --no-extra-legs" \
    "bench_rdh|200|python bench.py --config rdh --no-cpu-baseline --no-extra-legs" \
    "bench_sandra|200|python bench.py --config sandra --no-cpu-baseline --no-extra-legs" \
    "bench_vits2|300|python bench.py --workload vits2" \
    "e2e_b1|200|python tools/e2e_latency.py"
fi
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
# (the reference's arithmetic first: exact fp32 at B = 256 and B = 64, then the opt-in split-fp16 mode)
exec tools/gpu_steps.sh $S \
  "trace|300|rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o d -- python3 bench.py --no-cpu-baseline" \
  "trace_vits2|300|rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o v -- python3 bench.py --workload vits2 --no-cpu-baseline" \
  "fetch_f32|300|rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_f32 -o f -- python3 tools/prof_kernels.py --precision f32 --iters 10" \
  "write_f32|300|rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_f32 -o w -- python3 tools/prof_kernels.py --precision f32 --iters 10" \
  "fetch_f32_b64|300|rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_f32_b64 -o f -- python3 tools/prof_kernels.py --precision f32 --batch 64 --iters 10" \
  "write_f32_b64|300|rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_f32_b64 -o w -- python3 tools/prof_kernels.py --precision f32 --batch 64 --iters 10" \
  "mfma_f32|300|rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/mfma_f32 -o s -- python3 tools/prof_kernels.py --precision f32 --iters 10" \
  "mfma_f32_b64|300|rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/mfma_f32_b64 -o s -- python3 tools/prof_kernels.py --precision f32 --batch 64 --iters 10" \
  "fetch|300|rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 tools/prof_kernels.py --precision split_f16 --iters 10" \
  "write|300|rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 tools/prof_kernels.py --precision split_f16 --iters 10" \
  "mfma_step|300|rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/mfma_step -o s -- python3 tools/prof_kernels.py --precision split_f16 --iters 10" \
  "mfma_post|300|rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/mfma_post -o p -- python3 tools/time_postnet.py --iters 3" \
  "mfma_vits2|300|rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/mfma_vits2 -o v -- python3 tools/time_vits2.py --iters 2"
