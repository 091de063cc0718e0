set -o pipefail
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 4 --warmup 2 --no-extra-legs --no-cpu-baseline > gpurun_out/r2_m_$name.json 2>> gpurun_out/r2_m.err || echo "FAIL $name"; }
run ov1 TTSDEC_OVERLAP=1
run ov0 TTSDEC_OVERLAP=0
run ov1b TTSDEC_OVERLAP=1
run ov0b TTSDEC_OVERLAP=0
echo done
