#!/bin/bash
# Runs on the GPU box (gpurun): the round's final numbers on the frozen sources.  Every rocprofv3 call has the program
# itself after "--" and collects counters without any trace domain but --kernel-trace.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_final6
mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
timeout -k 10 300 python bench.py --batch 64 --no-cpu-baseline --no-extra-legs > $O/bench_batch64.json 2>/dev/null
timeout -k 10 300 python bench.py --batch 1 --no-cpu-baseline --no-extra-legs > $O/bench_batch1.json 2>/dev/null
timeout -k 10 300 python bench.py --postnet bf16 --no-cpu-baseline --no-extra-legs > $O/bench_postnet_bf16.json 2>/dev/null
timeout -k 10 300 python bench.py --dropout masks --no-cpu-baseline --no-extra-legs > $O/bench_masks.json 2>/dev/null
timeout -k 10 300 python bench.py --config rdh --no-cpu-baseline --no-extra-legs > $O/bench_rdh.json 2>/dev/null
timeout -k 10 300 python bench.py --config sandra --no-cpu-baseline --no-extra-legs > $O/bench_sandra.json 2>/dev/null
timeout -k 10 300 python bench.py --workload vits2 > $O/bench_vits2.json 2>/dev/null
echo "benches done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o d -- python3 bench.py --no-cpu-baseline --no-extra-legs > $O/bench_under_rocprof.json 2> $O/trace.err; echo "trace rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o v -- python3 bench.py --workload vits2 --no-cpu-baseline > /dev/null 2>&1; echo "trace vits rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 tools/prof_kernels.py --precision split_f16 --iters 10 > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 tools/prof_kernels.py --precision split_f16 --iters 10 > $O/write.log 2>&1; echo "write rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma_step -o s -- python3 tools/prof_kernels.py --precision split_f16 --iters 10 > $O/mfma_step.log 2>&1; echo "mfma step rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma_post -o p -- python3 tools/time_postnet.py --iters 3 > $O/mfma_post.log 2>&1; echo "mfma postnet rc=$?"
find $O -name "*.csv" | head -30
