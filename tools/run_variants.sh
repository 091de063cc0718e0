set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_final
mkdir -p $O
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --postnet bf16 --no-extra-legs --no-cpu-baseline > $O/bench_postnet_bf16.json 2>> $O/bench.err; echo "bf16 rc=$?"
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --batch 1 --no-extra-legs --no-cpu-baseline > $O/bench_b1.json 2>> $O/bench.err; echo "b1 rc=$?"
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --batch 64 --no-extra-legs --no-cpu-baseline > $O/bench_b64_split.json 2>> $O/bench.err; echo "b64 rc=$?"
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --dropout masks --no-extra-legs --no-cpu-baseline > $O/bench_masks.json 2>> $O/bench.err; echo "masks rc=$?"
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --config rdh > $O/bench_rdh.json 2>> $O/bench.err; echo "rdh rc=$?"
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --config sandra > $O/bench_sandra.json 2>> $O/bench.err; echo "sandra rc=$?"
timeout -k 10 300 python bench.py --workload vits2 --steps 5 --warmup 2 > $O/bench_vits2.json 2>> $O/bench.err; echo "vits2 rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 tools/prof_kernels.py --precision split_f16 --iters 10 > $O/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 tools/prof_kernels.py --precision split_f16 --iters 10 > $O/write.log 2>&1; echo "write rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --steps 3 --warmup 1 --no-extra-legs --no-cpu-baseline > $O/bench_traced.json 2> $O/trace.err; echo "trace rc=$?"
