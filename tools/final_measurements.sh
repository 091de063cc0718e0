set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout -k 10 300 python bench.py --precision f32 --postnet f32 --no-cpu-baseline > $O/bench_exact_f32.json 2>/dev/null
timeout -k 10 300 python bench.py --batch 64 --no-cpu-baseline > $O/bench_batch64.json 2>/dev/null
timeout -k 10 300 python bench.py --postnet bf16 --no-cpu-baseline > $O/bench_postnet_bf16.json 2>/dev/null
timeout -k 10 300 python bench.py --config rdh > $O/bench_rdh.json 2>/dev/null
timeout -k 10 300 python bench.py --config sandra > $O/bench_sandra.json 2>/dev/null
timeout -k 10 300 python bench.py --workload vits2 > $O/bench_vits2.json 2>/dev/null
timeout -k 10 300 python bench.py --batch 1 --no-cpu-baseline > $O/bench_batch1.json 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o d -- python bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o v -- python bench.py --workload vits2 --no-cpu-baseline > /dev/null 2>&1
ls $O $O/prof
