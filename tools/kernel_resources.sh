#!/bin/bash
# Register / LDS / scratch footprint of every kernel of libroft_hip.so as the compiler reports it (-Rpass-analysis=kernel-resource-usage;
# a kernel trace's VGPR column is granule-rounded and misses the AGPRs) -> csv on stdout.  Runs without a GPU.
#   bash tools/kernel_resources.sh > profiles/r04_kernel_resources.csv
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
for f in k_mask k_flow k_skf k_ukf k_render k_opticalflow engine engine_submit engine_step engine_results engine_ops engine_debug flow_producer; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-pass-failed -Rpass-analysis=kernel-resource-usage \
        -c $R/roft_amd/csrc/$f.hip -o $T/$f.o 2> $T/$f.ru &
done
wait
python3 - $T/*.ru <<'PY'
import re, subprocess, sys
print("kernel,vgprs,agprs,sgprs,scratch_bytes_per_lane,occupancy_waves_per_simd,static_lds_bytes")
for path in sys.argv[1:]:
    cur = None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: (.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            name = t.split(":", 1)[1].strip()
            try:
                name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            except OSError:
                pass
            cur = {"name": name.split("(")[0].replace("void ", "")}
        elif cur is not None:
            k, _, v = t.partition(":")
            cur[k.strip()] = v.strip()
            if k.strip().startswith("LDS Size"):
                if "device" not in cur["name"] or True:
                    print(",".join(['"%s"' % cur["name"] if "," in cur["name"] else cur["name"], cur.get("VGPRs", ""), cur.get("AGPRs", ""), cur.get("TotalSGPRs", ""),
                                    cur.get("ScratchSize [bytes/lane]", ""), cur.get("Occupancy [waves/SIMD]", ""), cur.get("LDS Size [bytes/block]", "")]))
                cur = None
PY
rm -rf $T
