set -x
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --serialize-streams > gpurun_out/final/bench_serialized.json 2> gpurun_out/final/bench_serialized.err
for b in 16 32 64; do python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras --batch $b > gpurun_out/final/bench_b$b.json 2> gpurun_out/final/bench_b$b.err; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --mode F > gpurun_out/final/bench_modeF.json 2> gpurun_out/final/bench_modeF.err
bash tools/bench_16bit_configs.sh final/cfg > gpurun_out/final/cfg.txt 2>&1
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/final/gpu_tests.txt 2>&1
tail -3 gpurun_out/final/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1; tail -2 gpurun_out/final/smoke.txt
