"""Writes tests/golden/oracle_tracker_small.json: the oracle tracker's own trajectory on a small seeded
stream (regression pin of the oracle, NOT a reference vector -- the reference cannot be built here).

    python tests/golden/make_oracle_fixture.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import util  # noqa: E402
from oracle import binding as ob  # noqa: E402

seed, n_frames, scale = 77, 14, 4
st = util.stream(seed, n_frames, scale=scale)
out = util.run_oracle_tracker(ob, st)
data = dict(seed=seed, n_frames=n_frames, scale=scale,
            pose_twist=[np.concatenate([o["pose"], o["twist"]]).tolist() for o in out],
            n=[int(o["n"]) for o in out], sel=[int(o["sel"]) for o in out])
with open(os.path.join(HERE, "oracle_tracker_small.json"), "w") as f:
    json.dump(data, f)
print("wrote", n_frames, "frames")
