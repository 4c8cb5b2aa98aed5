set -x
O=gpurun_out/r3a; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "outlier or render or likelihood" > $O/pytest_outlier.txt 2>&1; tail -5 $O/pytest_outlier.txt
ROFT_LIB_SO=$PWD/build_ab/k1prof.so timeout 200 python tools/k1_phase_profile.py 64 > $O/k1_phase.txt 2>&1
unset ROFT_LIB_SO
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pcie-frames 0 > $O/bench_driver.json 2> $O/bench_driver.err
ROFT_ONE_STREAM=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pcie-frames 0 > $O/bench_driver_one.json 2> /dev/null
ROFT_BENCH_DEVICE=0 ROFT_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_gloo.json 2> $O/bench_2rank.err
for n in 8 16 32 128 256; do timeout 300 python bench.py --steps 40 --warmup 8 --objects $n --no-cpu-baseline --pcie-frames 0 > $O/sweep_$n.json 2> $O/sweep_$n.err; done
tail -3 $O/*.err
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
