#!/usr/bin/env python3
"""Shader-cycle profile of the phases of one ukf_step launch.  Needs a library built with -DROFT_UKF_PROFILE:
  make -C roft_amd/csrc clean && make -C roft_amd/csrc CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DROFT_UKF_PROFILE"
Slots: 0 load, 1 process noise, 2 prediction square root, 3 fan-out + motion, 4 means, 5 covariance, 6 store,
7 correction square root, 8 fan-out + measurement, 9 means/deviations, 10 Py/Pxy, 11 Cholesky, 12 gain + update."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, util
from roft_amd import engine as E, synth, _lib as L
import test_engine_gpu as T
streams=[util.stream(100,14,scale=2)]
eng=T.make_engine(streams)
for k in range(14):
    depth,flow,mask,pose=util.frame_inputs(streams[0],k)
    eng.submit([dict(depth=depth,flow=flow,mask=mask,pose=pose,dt=streams[0].dt)]); eng.step(); eng.sync()
    if k in (4,5,11,13):
        # read ObjState.dbg: find offset via struct size: use hipMemcpy through roft_get_state? not exposed -> use debug export
        buf=(C.c_longlong*32)()
        L.lib().roft_debug_get_dbg(eng._h,0,buf)
        print(k,[int(x) for x in buf][:13], 'sweeps n12/n4/n10:', [int(x) for x in buf][16:19], 'corr guard th/wv/xx (1e-9):', [int(x) for x in buf][20:23], 'sub 23..26 (means | sigma perturbation | K solve | KPy):', [int(x) for x in buf][23:27])
