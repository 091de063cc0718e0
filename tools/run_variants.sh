set -o pipefail
run() { name=$1; b=$2; shift; shift; env "$@" timeout -k 10 200 python bench.py --batch $b --precision f32 --postnet f32 --steps 3 --warmup 1 --no-extra-legs --no-cpu-baseline > gpurun_out/r2_y_$name.json 2>> gpurun_out/r2_y.err || echo "FAIL $name"; python - <<PY
import json
d=json.load(open("gpurun_out/r2_y_$name.json"))
print("$name", d["value"], d["roofline"]["decode_step"]["ms_in_loop"], {k:v["ms"] for k,v in d["roofline"]["per_kernel"].items()})
PY
}
timeout -k 10 500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "ljspeech_dims_vs_oracle or frame_kernel or headline_600 or philox or shard or stop_rule or bound" 2>&1 | tail -2
run f32_b256 256 TTSDEC_X=1
run f32_b256_ov0 256 TTSDEC_OVERLAP=0
run f32_b256_ov2 256 TTSDEC_OVERLAP=2
run f32_b64 64 TTSDEC_X=1
run f32_b64_ov0 64 TTSDEC_OVERLAP=0
run f32_b64_ov1 64 TTSDEC_OVERLAP=1
run f32_b1 1 TTSDEC_X=1
run f32_b1_ov0 1 TTSDEC_OVERLAP=0
echo done
