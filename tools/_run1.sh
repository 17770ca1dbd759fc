set -x
mkdir -p gpurun_out/r3
python -m pytest tests/test_f32_mfma_gpu.py -x -q -m gpu > gpurun_out/r3/t_f32mfma.log 2>&1; tail -5 gpurun_out/r3/t_f32mfma.log
timeout -k 10 600 python bench.py --precision f16x3 --no-train --no-f32 --steps 10 --warmup 3 > gpurun_out/r3/bench_f16x3_a.json 2> gpurun_out/r3/bench_f16x3_a.err; tail -3 gpurun_out/r3/bench_f16x3_a.err; python - <<'PY'
import json
r=json.load(open('gpurun_out/r3/bench_f16x3_a.json'))
print(r['value'], r['ms_per_step'], r.get('accuracy'))
for k,v in r['roofline']['per_kernel'].items(): print(k, v)
PY
