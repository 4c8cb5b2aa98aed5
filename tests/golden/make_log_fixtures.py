"""Generates tests/golden/log_fixtures.json: the tracker's five log files as the C++ facade's bfl::Logger stand-in writes
them (tests/golden/log_writer.cpp over include/ROFT/Compat.h), and what the REFERENCE's own evaluation reader makes of
them -- evaluation/data_loader.py, DataLoader.load_generic / load_ours (:99-108, :186-241), imported here in the dev
container (numpy only).  Only the texts and the parsed arrays are committed; the reference source does not travel.

    python tests/golden/make_log_fixtures.py
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "/root/reference/evaluation")
from data_loader import DataLoader  # noqa: E402


def main():
    tmp = tempfile.mkdtemp()
    exe = os.path.join(tmp, "log_writer")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "golden", "log_writer.cpp"), "-o", exe])
    subprocess.check_call([exe, tmp])
    loader = DataLoader({"name": "ours", "config": {}})
    out = {}
    # the contents load_ours reads per video (data_loader.py:198-205); `pose_estimate_ycb` is `pose_estimate` after the
    # reference's frame conversion script: same row layout, its six leading velocity columns are dropped by the reader (:238-241)
    for name in ("pose_estimate", "velocity_estimate", "execution_times", "pose_measurements", "velocity_measurements"):
        path = os.path.join(tmp, name + ".txt")
        d = loader.load_generic(path)
        if name == "pose_estimate":
            d = d[:, 6:]
        out[name] = dict(text=open(path).read(), parsed=d.tolist())
    with open(os.path.join(ROOT, "tests", "golden", "log_fixtures.json"), "w") as f:
        json.dump(out, f, indent=1)
    print({k: (len(v["parsed"]), len(v["parsed"][0])) for k, v in out.items()})


if __name__ == "__main__":
    main()
