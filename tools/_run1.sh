mkdir -p gpurun_out/r3d
python -m pytest tests/test_gpu_ops.py tests/test_gpu_training.py tests/test_gpu_unet.py -x -q > gpurun_out/r3d/pytest.log 2>&1; tail -3 gpurun_out/r3d/pytest.log
python bench.py --config c3 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3d/c3_new.json 2> gpurun_out/r3d/c3_new.err
python bench.py --no-cpu-baseline > gpurun_out/r3d/c2_new.json 2> gpurun_out/r3d/c2_new.err
ANODDPM_NO_STEM_STATS=1 python bench.py --no-cpu-baseline > gpurun_out/r3d/c2_nostemstats.json 2> gpurun_out/r3d/c2_nostemstats.err
for f in gpurun_out/r3d/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    r=d.get('roofline',{})
    print(d['ms_per_step'], d['value'], r.get('frac'), [ (k['kernel'][:12], round(k['GBps'])) for k in r.get('hbm_kernels',[])])
    print({k:round(v,3) for k,v in r.get('class_ms_per_step',{}).items()})
except Exception as e: print('ERR',e)
PY
done
