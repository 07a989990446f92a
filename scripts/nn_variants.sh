for v in 16,2,2 8,2,2 8,4,2 16,4,2 8,2,1 8,4,1; do
  MMK_NN_VARIANT=$v python bench.py --no-cpu-baseline --no-grid --no-parity --steps 10 > gpurun_out/r02_nnv.json 2> gpurun_out/r02_nnv.err
  python -c "
import json; d=json.load(open('gpurun_out/r02_nnv.json')); print('variant $v', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_us'],1))"
done
