# Round-3 evidence batch (run on the GPU box from the repo root): bench line, kernel-trace statistics of the same command,
# PMC passes of the NN kernel and of the U-Net pass, the 2-rank gloo rehearsal of bench.py --gpus 2 on one GPU.
set -e
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/r03_bench_n1.json 2> gpurun_out/r03_bench_n1.err
tail -c 400 gpurun_out/r03_bench_n1.err
bash scripts/prof_bench.sh r03 && cd $R
bash scripts/pmc_nn.sh r03 > gpurun_out/r03_pmc_nn.log 2>&1 && cd $R
bash scripts/pmc_unet.sh r03 > gpurun_out/r03_pmc_unet.log 2>&1 && cd $R
tail -3 gpurun_out/r03_pmc_unet.log
MMK_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench_2ranks_gloo.json 2> gpurun_out/r03_bench_2ranks_gloo.err
tail -c 300 gpurun_out/r03_bench_2ranks_gloo.err
echo done
