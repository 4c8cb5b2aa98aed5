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
    # the whole results tree through DataLoader("ours").load() (:186-254), run from a scratch working directory: the layout
    # test/test.sh:126-131 creates (<results>/<dataset>/<variant>/<object>/<content>.txt), cam_K.json as roft_amd.io writes it
    import shutil
    sys.path.insert(0, ROOT)
    from roft_amd import io, synth
    work = os.path.join(tmp, "work")
    cfg = {"dataset": "fastycb", "masks_set": "gt", "of_set": "nvof_1_slow", "pose_set": "dope", "no_outrej": True,
           "excluded_objects": ["004_sugar_box", "005_tomato_soup_can", "006_mustard_bottle", "009_gelatin_box", "010_potted_meat_can"]}
    variant = "full_mask_gt_of_nvof_1_slow_pose_dope_no_outrej"
    obj_dir = os.path.join(work, "results", "ROFT_results", "fastycb", variant, "003_cracker_box")
    os.makedirs(obj_dir)
    for name in out:
        shutil.copy(os.path.join(tmp, name + ".txt"), os.path.join(obj_dir, ("pose_estimate_ycb" if name == "pose_estimate" else name) + ".txt"))
    seq_dir = os.path.join(work, "dataset", "fast-ycb", "003_cracker_box")
    os.makedirs(seq_dir)
    cam = synth.Camera.shape_b()
    io.write_cam_k(os.path.join(seq_dir, "cam_K.json"), cam)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        data = DataLoader({"name": "ours", "config": cfg}).load()
    finally:
        os.chdir(cwd)
    video = data["003_cracker_box"][0]
    out["load_ours"] = dict(variant=variant, cam_k_json=open(os.path.join(seq_dir, "cam_K.json")).read(),
                            cam_intrinsics={k: video["cam_intrinsics"][k] for k in ("fx", "fy", "cx", "cy")},
                            shapes={k: list(video[k].shape) for k in ("time", "pose", "velocity", "pose_meas", "vel_meas")},
                            pose=video["pose"].tolist(), segmentation_path=video["segmentation_path"], rgb_path=video["rgb_path"])
    with open(os.path.join(ROOT, "tests", "golden", "log_fixtures.json"), "w") as f:
        json.dump(out, f, indent=1)
    print({k: (len(v["parsed"]), len(v["parsed"][0])) for k, v in out.items() if "parsed" in v}, out["load_ours"]["shapes"], out["load_ours"]["cam_intrinsics"])


if __name__ == "__main__":
    main()
