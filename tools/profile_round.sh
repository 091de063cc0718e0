#!/bin/bash
# Runs on the GPU box (gpurun): kernel trace of the default bench, PMC passes for HBM traffic and MFMA utilisation.
# Every rocprofv3 call has the program itself after "--" and collects counters without any trace domain but
# --kernel-trace (see the task's profiling rules).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_prof
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_traced.json 2> $O/trace.err; echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 tools/prof_kernels.py --precision split_f16 --iters 10 > $O/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 tools/prof_kernels.py --precision split_f16 --iters 10 > $O/write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma_step -o s -- python3 tools/prof_kernels.py --precision split_f16 --iters 10 > $O/mfma_step.log 2>&1; echo "mfma step rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma_post -o p -- python3 tools/time_postnet.py --iters 3 > $O/mfma_post.log 2>&1; echo "mfma postnet rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma_vits -o v -- python3 tools/time_vits2.py --iters 3 > $O/mfma_vits.log 2>&1; echo "mfma vits rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_vits -o vits -- python3 bench.py --workload vits2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_vits_traced.json 2> $O/trace_vits.err; echo "trace vits rc=$?"
find $O -name "*.csv" | head -30
