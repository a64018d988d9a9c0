cd /root/repo
python -m pytest tests/test_forward_gpu.py -x -q 2>&1 | tail -5
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/two_stream.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/two_stream.json'))
print(d['value'], d['ms_per_step'], d.get('breakdown_ms'))
PY
