# Round-5 evidence batch, part B (run on the GPU box from the repo root; part A = scripts/pmc_nn.sh r05, scripts/pmc_unet.sh r05,
# scripts/deep_stamps.py, scripts/ring_stamps.py, whose summaries profiles/r05_nn_traffic.json etc. are made of): the bench line,
# kernel-trace statistics of the same command, the one-rank RCCL rehearsal with RCCL's own log, same-box A/B against the round-4
# kernels (build_exp/lib_base.so = round-4 mmk_unet.hip behind this round's ABI), the loader figures.
set -e
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py > gpurun_out/r05_bench_n1.json 2> gpurun_out/r05_bench_n1.err
tail -c 300 gpurun_out/r05_bench_n1.err
bash scripts/prof_bench.sh r05 && cd $R
NCCL_DEBUG=INFO python3 bench.py --gpus 1 --force-dist --no-cpu-baseline --no-side --no-grid > gpurun_out/r05_bench_1rank_nccl.json 2> gpurun_out/r05_bench_1rank_nccl.err
tail -c 200 gpurun_out/r05_bench_1rank_nccl.err
if [ -f build_exp/lib_base.so ]; then
  bash scripts/ab_libs.sh 3 r04kernels:build_exp/lib_base.so r05: > gpurun_out/r05_same_box_ab.txt 2>&1
  cat gpurun_out/r05_same_box_ab.txt
  MMK_LIB=build_exp/lib_base.so python3 scripts/ab_lib.py 32 5 > gpurun_out/r05_ab_layers_r04.txt 2>&1
  python3 scripts/ab_lib.py 32 5 > gpurun_out/r05_ab_layers_r05.txt 2>&1
  for v in base final; do MMK_LIB=build_exp/lib_$v.so python3 scripts/time_unpack.py 2>/dev/null | tail -1; done > gpurun_out/r05_unpack_ab.txt
fi
python3 -m pytest tests -q -m gpu > gpurun_out/r05_gpu_tests.log 2>&1 || true      # (tests/test_gpu_loader.py writes gpurun_out/r05_loader.json)
tail -3 gpurun_out/r05_gpu_tests.log
echo done
