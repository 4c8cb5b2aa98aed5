cd roft_amd/csrc && touch k_mask.hip && make -j8 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DROFT_MASK_PROFILE -DROFT_EXP_DOUBLE_ATOMIC" 2>&1 | grep -E "error" ; cd ../..
PHASES=mask timeout 200 python tools/k1_phase_profile.py 64 | tail -2
FLOWFIX=1 PHASES=mask timeout 200 python tools/k1_phase_profile.py 64 | tail -2
