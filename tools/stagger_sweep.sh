mkdir -p gpurun_out/stag
for S in 0 1 2 3 4 6; do
  QC_BWD2_STAGGER=$S python bench.py --no-cpu-baseline --no-other-configs --steps 100 --warmup 10 > gpurun_out/stag/b$S.json 2> gpurun_out/stag/b$S.err
  python -c "
import json;d=json.load(open('gpurun_out/stag/b$S.json'));print('stagger $S', round(d['ms_per_step'],4), round(d['kernels_ms']['stage_circuit_bwd']*1e3,1))"
done
