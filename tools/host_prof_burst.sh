ROFT_HOST_PROF=1+ python bench.py --steps 20 --warmup 5 --windows 1 --no-cpu-baseline --pcie-frames 0 --no-extras --no-kernel-timing --rehearsal-ms 0 --json-out "" 2>&1 | grep "roft host" | tail -12
