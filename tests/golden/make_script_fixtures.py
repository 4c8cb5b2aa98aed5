"""Generates tests/golden/script_fixtures.json by RUNNING two of the reference's dataset scripts in the dev container on
files written by roft_amd.io (only inputs and outputs are committed; the scripts do not travel):
  tools/dataset/dope_pose_finder/pose_finder.py   -- the initial pose test/test_ho3d.sh:68 starts the tracker from
  tools/dataset/data_txt_generation/generate_data_txt.py -- data.txt of a sequence

    python tests/golden/make_script_fixtures.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from roft_amd import io  # noqa: E402

REF = "/root/reference/tools/dataset"


def main():
    rng = np.random.default_rng(5)
    out = {"pose_finder": [], "data_txt": []}
    tmp = tempfile.mkdtemp()
    # detections at 5 fps in a 30 fps file, the first few missing; one case with a valid row OFF the 5 fps grid first
    for case, (n, first_valid, off_grid) in enumerate([(40, 0, None), (40, 12, None), (40, 18, 7), (30, 24, 3), (20, None, None)]):
        pose = np.zeros((n, 7))
        pose[:, 3] = 1.0
        ok = np.zeros(n, bool)
        for k in range(n):
            if first_valid is not None and k >= first_valid and k % 6 == 0:
                q = rng.normal(size=4)
                pose[k] = np.concatenate([rng.uniform(-0.3, 0.3, 3) + [0, 0, 0.7], q / np.linalg.norm(q)])
                ok[k] = True
        if off_grid is not None:
            q = rng.normal(size=4)
            pose[off_grid] = np.concatenate([[0.1, 0.2, 0.6], q / np.linalg.norm(q)])
            ok[off_grid] = True
        path = os.path.join(tmp, "poses_%d.txt" % case)
        io.write_poses(path, pose, ok)
        r = subprocess.run([sys.executable, os.path.join(REF, "dope_pose_finder", "pose_finder.py"), path, "5"], capture_output=True, text=True)
        out["pose_finder"].append(dict(poses_txt=open(path).read(), fps=5, stdout=r.stdout))
    for n in (1, 7, 100):
        seq = os.path.join(tmp, "seq_%d" % n)
        os.makedirs(os.path.join(seq, "gt"))
        open(os.path.join(seq, "gt", "poses.txt"), "w").write("0.0 0.0 0.0 0.0 0.0 0.0 0.0\n" * n)
        subprocess.check_call([sys.executable, os.path.join(REF, "data_txt_generation", "generate_data_txt.py"), seq])
        out["data_txt"].append(dict(frames=n, text=open(os.path.join(seq, "data.txt")).read()))
    with open(os.path.join(ROOT, "tests", "golden", "script_fixtures.json"), "w") as f:
        json.dump(out, f, indent=1)
    print([c["stdout"][:30] for c in out["pose_finder"]], [len(c["text"]) for c in out["data_txt"]])


if __name__ == "__main__":
    main()
