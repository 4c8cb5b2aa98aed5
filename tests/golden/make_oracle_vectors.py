#!/usr/bin/env python3
"""Conformance vectors of the ORACLE's filter algebra (tests/golden/oracle_vectors/): inputs + the outputs oracle/ computes for
them, for the operators whose third-party semantics (bfl: UTWeight, sigma_point, unscented_transform, quaternion sums and
differences -- SURVEY App. A.4, recalled, UNVERIFIED) the oracle restates without a reference build to hold them against.

Two readers:
  * tests/test_oracle_cpu.py::test_oracle_reproduces_its_conformance_vectors -- the oracle must keep producing them (a change
    of the restatement shows up as a diff of committed data, not silently);
  * tests/ref_kit/replay.cpp -- a harness for somebody who HAS the reference's dependencies (Eigen, bfl, RobotsIO): it replays
    the same inputs through the real bfl::UKFPrediction / ROFT::UKFCorrection / ROFT::SKFCorrection / bfl::sigma_point and
    prints the differences.  It cannot be built in this repository's container; it is how "parity unpinned" can be closed.

Every case is written twice: <name>.json (tests) and <name>.txt (the harness: `key rows cols` then the values, row-major --
no JSON parser needed in C++).  Run from the repository root: python tests/golden/make_oracle_vectors.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "oracle_vectors")

MEAS_VELOCITY, MEAS_POSE, MEAS_POSE_VELOCITY = 1, 2, 3


def random_belief(rng, scale):
    A = rng.normal(size=(12, 12))
    P = scale * (A @ A.T / 12.0 + 0.5 * np.eye(12))
    q = rng.normal(size=4)
    mean = np.concatenate([rng.normal(size=3) * 0.1, rng.normal(size=3) * 0.5, [0.05, -0.03, 0.7] + rng.normal(size=3) * 0.02, q / np.linalg.norm(q)])
    return mean, P


def cases():
    from oracle import binding as ob
    rng = np.random.default_rng(20251003)
    out = {}
    # ---- UT weights: the three augmented dimensions the filter uses, two parameter sets
    for ut in ((1.0, 2.0, 0.0), (0.5, 2.0, 1.0)):
        for n in (18, 21, 24):
            out["ut_weights_n%d_a%g" % (n, ut[0])] = dict(n=[n], ut=list(ut), c_wm0_wc0_wi=ob.ut_weights(n, ut))
    # ---- sigma sets: state + process noise (r = 9), state + diagonal measurement noise (r = 6, r = 12)
    psd, sig_w, T = [1.0, 1.0, 1.0], [1.0, 1.0, 1.0], 1.0 / 30.0
    Q = ob.process_noise(psd, sig_w, T)
    mean, P = random_belief(rng, 1e-3)
    out["sigma_points_process_noise"] = dict(mean=mean, P=P, noise=Q, ut=[1.0, 2.0, 0.0], sigma=ob.sigma_points(mean, P, Q))
    Rv = np.diag([0.1] * 3 + [1e-4] * 3)
    out["sigma_points_velocity_noise"] = dict(mean=mean, P=P, noise=Rv, ut=[1.0, 2.0, 0.0], sigma=ob.sigma_points(mean, P, Rv))
    # ---- quaternion helpers
    qa, qb = mean[9:], random_belief(rng, 1e-3)[0][9:]
    rv = rng.normal(size=3) * 0.4
    bp, df = np.zeros(4), np.zeros(3)
    ob.lib().ro_quat_boxplus(ob._p(ob._f64(qa)), ob._p(ob._f64(rv)), ob._p(bp))
    ob.lib().ro_quat_diff(ob._p(ob._f64(qa)), ob._p(ob._f64(qb)), ob._p(df))
    out["quaternion_sum_and_difference"] = dict(q=qa, r=rv, q_boxplus_r=bp, q_b=qb, diff_q_qb=df)
    # ---- prediction (bfl::UKFPrediction over ROFT::CartesianQuaternionModel, CartesianQuaternionModel.cpp:86-141)
    for i, (ut, scale) in enumerate((((1.0, 2.0, 0.0), 1e-3), ((1.0, 2.0, 0.0), 5e-2), ((0.5, 2.0, 1.0), 1e-3))):
        mean, P = random_belief(rng, scale)
        m1, P1 = ob.ukf_predict(mean, P, Q, T, ut)
        out["ukf_predict_%d" % i] = dict(mean=mean, P=P, psd_lin_acc=psd, sigma_ang_vel=sig_w, Q=Q, T=[T], ut=list(ut), mean_out=m1, P_out=P1)
    # ---- corrections (ROFT::UKFCorrection over CartesianQuaternionMeasurement, UKFCorrection.cpp:54-133, ...Measurement.cpp:357-487)
    rp, rv6 = [1e-3] * 3 + [1e-4] * 3, [0.1] * 3 + [1e-4] * 3
    for i in range(2):
        mean, P = random_belief(rng, 1e-3)
        q = mean[9:] + rng.normal(size=4) * 0.02
        pose = np.concatenate([mean[6:9] + rng.normal(size=3) * 0.01, q / np.linalg.norm(q)])
        twist = np.concatenate([mean[:3] + np.cross(mean[3:6], -mean[6:9]), mean[3:6]]) + rng.normal(size=6) * 0.02
        for name, mtype, meas, rd in (("velocity", MEAS_VELOCITY, twist, rv6), ("pose", MEAS_POSE, pose, rp),
                                      ("pose_velocity", MEAS_POSE_VELOCITY, np.concatenate([twist, pose]), rv6 + rp)):
            rc, m1, P1 = ob.ukf_correct(mean, P, mtype, meas, rd)
            out["ukf_correct_%s_%d" % (name, i)] = dict(mean=mean, P=P, type=[mtype], meas=meas, Rdiag=rd, ut=[1.0, 2.0, 0.0], status=[rc], mean_out=m1, P_out=P1)
    # sigma rotations beyond pi: the input deviation of such a column is the WRAPPED logarithm (tests/test_parity_gpu.py)
    mean, P = random_belief(rng, 1e-3)
    P[9:, 9:] += np.eye(3) * 0.9
    q = mean[9:] + rng.normal(size=4) * 0.05
    pose = np.concatenate([mean[6:9] + rng.normal(size=3) * 0.01, q / np.linalg.norm(q)])
    rc, m1, P1 = ob.ukf_correct(mean, P, MEAS_POSE, pose, rp)
    out["ukf_correct_pose_beyond_pi"] = dict(mean=mean, P=P, type=[MEAS_POSE], meas=pose, Rdiag=rp, ut=[1.0, 2.0, 0.0], status=[rc], mean_out=m1, P_out=P1)
    # ---- velocity filter (ROFT::SKFCorrection::correctStep, SKFCorrection.cpp:37-153), with and without Laplacian re-weighting
    N = 40
    x = rng.normal(size=6) * 0.05
    Pv = np.eye(6) * 1e-3 + 1e-4
    Hm = rng.normal(size=(2 * N, 6)) * 20.0
    y = Hm @ (x + rng.normal(size=6) * 0.02) + rng.normal(size=2 * N)
    y[10:14] += 25.0    # a few gross outliers: what the re-weighting is for
    for rw in (0, 1):
        rc, x1, P1 = ob.skf_correct(x, Pv, y, Hm, (1.0, 1.0), bool(rw))
        out["skf_correct_reweight%d" % rw] = dict(x_pred=x, P_pred=Pv, y=y, H=Hm, Rdiag=[1.0, 1.0], reweight=[rw], status=[rc], x_out=x1, P_out=P1)
    # ---- flow measurement + mask propagation on a small image (ImageOpticalFlowMeasurement.hpp:231-283, ...OFAidedSource.hpp:234-281)
    W, H = 32, 32
    yy, xx = np.mgrid[0:H, 0:W]
    mask = (((xx - 15) ** 2 / 70.0 + (yy - 14) ** 2 / 50.0) < 1.0).astype(np.uint8) * 255
    depth = (0.6 + 0.002 * xx + 0.001 * yy).astype(np.float32)
    depth[5::7, 3::5] = 0.0
    flows = [(np.stack([1.5 + 0.02 * yy, -0.8 + 0.03 * xx], -1) * (1.0 + 0.1 * k)).astype(np.float32) for k in range(3)]
    flows[1][14, 15] = [np.nan, 0.0]
    flows[2][16, 13] = [1e10, 1e10]
    cam = ob.camera(W, H, 40.0, 40.0, 16.0, 16.0)
    n, uv, yv, Hv = ob.flow_measurement(cam, mask, depth, flows[0], 1.0 / 30.0, radius=5.0)
    out["flow_measurement_32x32"] = dict(width=[W], height=[H], cam=[40.0, 40.0, 16.0, 16.0], dt=[1.0 / 30.0], radius=[5.0], depth_max=[2.0],
                                         mask=mask, depth=depth, flow=flows[0].reshape(H, 2 * W), n=[n], uv=uv.reshape(-1, 2), y=yv, H=Hv.reshape(-1, 6))
    m3 = mask.copy()
    m3[::3, ::2] = np.where(m3[::3, ::2] > 0, 1, 0)   # three-valued: the general, map-based propagation
    for name, m in (("binary", mask), ("three_valued", m3)):
        got = ob.mask_propagate(m, flows)
        out["mask_propagate_%s_32x32" % name] = dict(width=[W], height=[H], mask=m, flow0=flows[0].reshape(H, 2 * W), flow1=flows[1].reshape(H, 2 * W),
                                                     flow2=flows[2].reshape(H, 2 * W), mask_out=got)
    return out


def to_lists(v):
    a = np.asarray(v)
    if a.dtype.kind == "f":
        # (non-finite flow values: JSON has no NaN / 1e10 stays as it is)
        # (float32 images: the shortest decimal that reads back to the same float32, read as float64 and cast)
        f32 = a.dtype == np.float32
        return [[(None if not np.isfinite(x) else (float(np.format_float_positional(x, unique=True, trim="0")) if f32 else float(x))) for x in row]
                for row in np.atleast_2d(a)]
    return np.atleast_2d(a).astype(np.int64).tolist()


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, case in cases().items():
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump({k: to_lists(v) for k, v in case.items()}, f)
        with open(os.path.join(OUT, name + ".txt"), "w") as f:
            for k, v in case.items():
                a = np.atleast_2d(np.asarray(v))
                f.write("%s %d %d\n" % (k, a.shape[0], a.shape[1]))
                if a.dtype == np.float32:
                    f.write(" ".join("%.9g" % float(x) for x in a.ravel()) + "\n")   # (nan / 1e+10 read back by strtod)
                elif a.dtype.kind == "f":
                    f.write(" ".join(repr(float(x)) for x in a.astype(np.float64).ravel()) + "\n")
                else:
                    f.write(" ".join(str(int(x)) for x in a.ravel()) + "\n")
    print("wrote", len(os.listdir(OUT)), "files to", OUT)


if __name__ == "__main__":
    main()
