import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from roft_amd import _lib as L, engine as E, synth
dev = torch.device("cuda", 0)
cam = synth.Camera.shape_a()
st = synth.make_stream(4000, 8, cam, device=dev)   # torch work first, like bench.py
def probe(tag):
    cfg = E.default_config(cam.width, cam.height, synth.FLOW_F32C2, max_objects=2, max_batch_frames=8)
    eng = E.ROFTFilterBatch(cfg)
    out = (C.c_double * 25)()
    L.check(L.lib().roft_debug_probe_streams(eng._h, out))
    m = np.array(out).reshape(5, 5)
    print(tag); print(np.round(m).astype(int))
    return eng
e1 = probe("first engine (pose0 pose1 vel mask up)")
e2 = probe("second engine while the first is alive")
e3 = probe("third engine")
