#!/bin/bash
# Copies one final-measurement session (tools/final_measurements.sh prof <S>) from gpurun_out/ into profiles/r04_z_* and writes the
# traffic / MFMA summaries.   tools/collect_final.sh <prof session> <tests session>
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/$1
python3 - "$O" <<'PY'
import subprocess, json, sys
O=sys.argv[1]
def order(f): return ",".join(json.loads(open(f'{O}/{f}.log').read().strip().splitlines()[-1]).keys())
for fd,wd,prec,b,out in (('fetch_f32','write_f32','f32',256,'profiles/r04_traffic_f32_b256.json'),
                         ('fetch_f32_b64','write_f32_b64','f32',64,'profiles/r04_traffic_f32_b64.json'),
                         ('fetch','write','split_f16',256,'profiles/r04_traffic_split_f16_b256.json')):
    cmd=['python','tools/summarize_pmc.py','traffic',f'{O}/{fd}/f_counter_collection.csv',f'{O}/{wd}/w_counter_collection.csv','--out',out,'--precision',prec,'--batch',str(b),'--note','round 4 final sources','--order',order(fd)]
    r=subprocess.run(cmd,capture_output=True,text=True); assert r.returncode==0, r.stderr
    d=json.load(open(out)); print(out, d['sources_digest'], {k:v['hbm_bytes'] for k,v in d['per_launch'].items() if '+' in k or k=='query'})
for n,outn,pre in (('mfma_f32','r04_z_pmc_mfma_f32_b256','s'),('mfma_f32_b64','r04_z_pmc_mfma_f32_b64','s'),('mfma_step','r04_z_pmc_mfma_split_b256','s'),('mfma_post','r04_z_pmc_mfma_postnet','p'),('mfma_vits2','r04_z_pmc_mfma_vits2','v')):
    args=['python','tools/summarize_pmc.py','mfma',f'{O}/{n}/{pre}_counter_collection.csv','--out',f'profiles/{outn}.csv']
    if n!='mfma_post' and n!='mfma_vits2': args+=['--order',order(n)]
    r=subprocess.run(args,capture_output=True,text=True); assert r.returncode==0, r.stderr
PY
cp $O/trace/d_kernel_stats.csv profiles/r04_z_bench_default_kernel_stats.csv
cp $O/trace/v_kernel_stats.csv profiles/r04_z_bench_vits2_kernel_stats.csv
tail -1 $O/trace.log > profiles/r04_z_bench_under_rocprof.json
tail -1 $O/trace_vits2.log > profiles/r04_z_bench_vits2_under_rocprof.json
for p in fetch_f32 write_f32 fetch_f32_b64 write_f32_b64 fetch write; do pre=${p:0:1}; cp $O/$p/${pre}_counter_collection.csv profiles/r04_z_pmc_${p}_counter_collection.csv; done
[ -n "$2" ] && cp gpurun_out/$2/tests.log profiles/r04_z_gpu_tests.txt
echo collected
