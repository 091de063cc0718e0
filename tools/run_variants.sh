set -o pipefail
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-extra-legs --no-cpu-baseline > gpurun_out/r2_o_$name.json 2>> gpurun_out/r2_o.err || echo "FAIL $name"; python - <<PY
import json
d=json.load(open("gpurun_out/r2_o_$name.json"))
print("$name", d["value"], d["roofline"]["decode_step"]["ms_in_loop"])
PY
}
run base TTSDEC_OVERLAP=1
run a_2_24 TTSDEC_OVERLAP=1 TTSDEC_THROTTLE_A=2,24
run a_4_24 TTSDEC_OVERLAP=1 TTSDEC_THROTTLE_A=4,24
run a_8_24 TTSDEC_OVERLAP=1 TTSDEC_THROTTLE_A=8,24
run a_4_40 TTSDEC_OVERLAP=1 TTSDEC_THROTTLE_A=4,40
run d_0 TTSDEC_OVERLAP=2
run d_4_32 TTSDEC_OVERLAP=2 TTSDEC_THROTTLE_D=4,32
run d_8_32 TTSDEC_OVERLAP=2 TTSDEC_THROTTLE_D=8,32
run d_16_32 TTSDEC_OVERLAP=2 TTSDEC_THROTTLE_D=16,32
run d_8_48 TTSDEC_OVERLAP=2 TTSDEC_THROTTLE_D=8,48
run d_16_20 TTSDEC_OVERLAP=2 TTSDEC_THROTTLE_D=16,20
echo done
