set -o pipefail
run() { name=$1; b=$2; shift; shift; env "$@" timeout -k 10 200 python bench.py --batch $b --steps 3 --warmup 1 --no-extra-legs --no-cpu-baseline > gpurun_out/r2_v_$name.json 2>> gpurun_out/r2_v.err || echo "FAIL $name"; python - <<PY
import json
d=json.load(open("gpurun_out/r2_v_$name.json"))
print("$name", d["value"], d["roofline"]["decode_step"]["ms_in_loop"], {k:v["ms"] for k,v in d["roofline"]["per_kernel"].items()})
PY
}
timeout -k 10 500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "ljspeech_dims_vs_oracle or split_f16 or frame_kernel or golden or philox or bound" 2>&1 | tail -2
run b1 1 TTSDEC_X=1
run b1_ov1 1 TTSDEC_OVERLAP=1
run b16 16 TTSDEC_X=1
run b32 32 TTSDEC_X=1
run b32_ov0 32 TTSDEC_OVERLAP=0
echo done
