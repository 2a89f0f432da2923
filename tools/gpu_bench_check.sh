mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/b_default.log 2> gpurun_out/b_default.err; echo "[bench] rc=$?"
for S in weak strong; do
  UCF_BENCH_ONE_DEVICE=1 UCF_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 3 --scaling $S > gpurun_out/b_gpus2_$S.log 2> gpurun_out/b_gpus2_$S.err; echo "[bench --gpus 2 $S] rc=$?"
done
UCF_BENCH_ONE_DEVICE=1 UCF_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 3 > gpurun_out/b_gpus2_def.log 2> gpurun_out/b_gpus2_def.err; echo "[bench --gpus 2 default] rc=$?"
timeout -k 10 600 python -m pytest tests/test_sharding_gloo.py -m gpu -q --no-header -p no:cacheprovider > gpurun_out/pytest_shard.log 2>&1; echo "[pytest shard] rc=$? $(tail -1 gpurun_out/pytest_shard.log)"
