O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
for v in ukfprof_base ukfprof; do echo $v; ROFT_LIB_SO=$PWD/build_ab/$v.so timeout 200 python tools/ukf_phase_profile.py 2>&1 | grep -v amdgpu.ids | cut -c1-200; done
for v in skfprof_base skfprof; do echo $v; PHASES=skf ROFT_LIB_SO=$PWD/build_ab/$v.so timeout 200 python tools/k1_phase_profile.py 64 2>&1 | tail -2; done
cp roft_amd/csrc/libroft_hip.so build_ab/new.so
bash tools/ab.sh "--steps 20 --warmup 5 --no-extras" base.so new.so
bash tools/ab.sh "--steps 60 --warmup 12 --no-extras" base.so new.so
bash tools/ab.sh "--steps 240 --warmup 16 --no-extras" base.so new.so
