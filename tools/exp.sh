run() { timeout 200 python bench.py --no-cpu-baseline --pcie-frames 0 --no-kernel-timing "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4))"; }
for th in 512 256; do
cd roft_amd/csrc && touch k_mask.hip && make -j8 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DROFT_MASK_THREADS=$th" 2>&1 | grep -E "error" ; cd ../..
echo threads $th
timeout 300 python -m pytest tests/test_batch_gpu.py tests/test_parity_gpu.py -x -q --timeout 200 2>&1 | tail -1
for i in 1 2; do run --steps 20 --warmup 5; done
run --steps 60 --warmup 12; run --steps 240 --warmup 16; run --steps 240 --warmup 16
done
