# Round-4 evidence batch (run on the GPU box from the repo root): bench line, kernel-trace statistics of the same command,
# PMC passes of the NN kernel (dim 2) and of the U-Net pass, bench.py --gpus 2 started plainly (it launches its own ranks;
# gloo on the one GPU of the box), run-to-run sweep of the correspondences for dim 2 and dim 3.
set -e
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py > gpurun_out/r04_bench_n1.json 2> gpurun_out/r04_bench_n1.err
tail -c 300 gpurun_out/r04_bench_n1.err
bash scripts/prof_bench.sh r04 && cd $R
bash scripts/pmc_nn.sh r04 > gpurun_out/r04_pmc_nn.log 2>&1 && cd $R
bash scripts/pmc_unet.sh r04 > gpurun_out/r04_pmc_unet.log 2>&1 && cd $R
tail -3 gpurun_out/r04_pmc_unet.log
MMK_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_bench_2ranks_gloo.json 2> gpurun_out/r04_bench_2ranks_gloo.err
tail -c 300 gpurun_out/r04_bench_2ranks_gloo.err
python3 scripts/nn_sweep.py 2 > gpurun_out/r04_nn_sweep_dim2.txt 2>&1
python3 scripts/nn_sweep.py 3 > gpurun_out/r04_nn_sweep_dim3.txt 2>&1
tail -n 1 gpurun_out/r04_nn_sweep_dim2.txt; tail -n 1 gpurun_out/r04_nn_sweep_dim3.txt
echo done
