timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_batch_gpu.py tests/test_engine_gpu.py tests/test_configs_gpu.py tests/test_engine_edge_gpu.py -x -q --timeout 300 2>&1 | tail -2
echo "base split"; bash tools/ab.sh "--steps 20 --warmup 5" base.so split.so;  bash tools/ab.sh "--steps 240 --warmup 16" base.so split.so
cd roft_amd/csrc && touch k_mask.hip && make -j8 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DROFT_MASK_PROFILE" 2>&1 | grep -E "error" ; cd ../..
PHASES=mask timeout 200 python tools/k1_phase_profile.py 64 | tail -2
