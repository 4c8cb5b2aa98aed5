"""An independent numpy statement of the pose UKF, written from SURVEY.md App. A.4 - A.6 (the specification of bfl's unscented
transform and of the reference's state / measurement models), NOT from oracle/ro_ukf.c: test infrastructure that holds the C
oracle against a second implementation of the same specification (tests/test_oracle_cpu.py).
State mean 13 = [v(3), w(3), x(3), q(w,x,y,z)], covariance 12 x 12 over [v, w, x, rotation vector]."""
import numpy as np


def qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])


def qexp(r):
    """Rotation vector -> quaternion (cos(|r|/2), sin(|r|/2) r/|r|), identity for r = 0 (A.4)."""
    n = np.linalg.norm(r)
    if n == 0.0:
        return np.array([1.0, 0.0, 0.0, 0.0])
    return np.concatenate([[np.cos(n / 2)], np.sin(n / 2) * np.asarray(r) / n])


def boxplus(q, r):
    """sum_quaternion_rotation_vector: exp(r) (x) q."""
    return qmul(qexp(r), q)


def qdiff(a, b):
    """diff_quaternion: rotation vector of a (x) b^-1, shortest arc (w >= 0 before the logarithm)."""
    d = qmul(a, np.array([b[0], -b[1], -b[2], -b[3]]))
    if d[0] < 0:
        d = -d
    n = np.linalg.norm(d[1:])
    if n == 0.0:
        return np.zeros(3)
    return 2.0 * np.arctan2(n, d[0]) * d[1:] / n


def ut_weights(n, alpha, beta, kappa):
    lam = alpha * alpha * (n + kappa) - n
    c = n + lam
    wm = np.full(2 * n + 1, 1.0 / (2.0 * c))
    wc = wm.copy()
    wm[0] = lam / c
    wc[0] = wm[0] + 1.0 - alpha * alpha + beta
    return c, wm, wc


def sigma_points(mean, P, N, c):
    """Columns [0, +sqrt(c) A, -sqrt(c) A] of the augmented Gaussian (state 12 dof + noise r), A = U diag(sqrt(s)) of the SVD."""
    r = N.shape[0]
    n = 12 + r
    Pa = np.zeros((n, n))
    Pa[:12, :12] = P
    Pa[12:, 12:] = N
    U, s, _ = np.linalg.svd(Pa)
    A = U * np.sqrt(s)
    cols = [np.zeros(n)] + [np.sqrt(c) * A[:, j] for j in range(n)] + [-np.sqrt(c) * A[:, j] for j in range(n)]
    states, noises = [], []
    for d in cols:
        st = np.empty(13)
        st[:9] = mean[:9] + d[:9]
        st[9:] = boxplus(mean[9:], d[9:12])
        states.append(st)
        noises.append(d[12:])
    return states, noises


def quat_mean(qs, wm):
    M = sum(w * np.outer(q, q) for q, w in zip(qs, wm))
    vals, vecs = np.linalg.eigh(M)
    return vecs[:, np.argmax(vals)]


def mean_and_deviations(cols, wm, n_lin):
    """Columns with n_lin linear rows followed by a quaternion: (mean, deviations [n_lin + 3, ncols])."""
    lin = sum(w * c[:n_lin] for c, w in zip(cols, wm))
    qm = quat_mean([c[n_lin:] for c in cols], wm)
    D = np.array([np.concatenate([c[:n_lin] - lin, qdiff(c[n_lin:], qm)]) for c in cols]).T
    return np.concatenate([lin, qm]), D


def process_noise(psd_lin_acc, sigma_ang_vel, T):
    Q = np.zeros((9, 9))
    a = np.diag(psd_lin_acc)
    Q[0:3, 0:3] = a * T
    Q[3:6, 3:6] = np.diag(sigma_ang_vel)
    Q[6:9, 6:9] = a * T ** 3 / 3.0
    Q[0:3, 6:9] = Q[6:9, 0:3] = a * T ** 2 / 2.0
    return Q


def motion(state, noise, T):
    v, w, x, q = state[0:3], state[3:6], state[6:9], state[9:13]
    out = np.empty(13)
    out[0:3] = v + noise[0:3]
    out[3:6] = w + noise[3:6]
    out[6:9] = x + noise[6:9] + v * T                  # the velocity WITHOUT its noise (A.5)
    wn = np.linalg.norm(w) + np.finfo(float).eps
    th = wn * T
    dq = np.concatenate([[np.cos(th / 2)], np.sin(th / 2) / wn * w])
    out[9:13] = qmul(dq, q)                              # the angular velocity without its noise
    return out


def predict(mean, P, Q, T, ut=(1.0, 2.0, 0.0)):
    c, wm, wc = ut_weights(12 + 9, *ut)
    states, noises = sigma_points(mean, P, Q, c)
    cols = [motion(s, n, T) for s, n in zip(states, noises)]
    m, D = mean_and_deviations(cols, wm, 9)
    return m, (D * wc) @ D.T


VELOCITY, POSE, POSE_VELOCITY = 1, 2, 3


def measure(state, noise, mtype):
    v, w, x, q = state[0:3], state[3:6], state[6:9], state[9:13]
    rows = []
    if mtype in (VELOCITY, POSE_VELOCITY):
        rows += list(v + np.cross(w, -x) + noise[0:3]) + list(w + noise[3:6])
    if mtype in (POSE, POSE_VELOCITY):
        off = 6 if mtype == POSE_VELOCITY else 0
        rows += list(x + noise[off:off + 3]) + list(boxplus(q, noise[off + 3:off + 6]))
    return np.array(rows)


def correct(mean, P, mtype, meas, Rdiag, ut=(1.0, 2.0, 0.0)):
    r = len(Rdiag)
    c, wm, wc = ut_weights(12 + r, *ut)
    states, noises = sigma_points(mean, P, np.diag(Rdiag), c)
    Y = [measure(s, n, mtype) for s, n in zip(states, noises)]
    has_pose = mtype in (POSE, POSE_VELOCITY)
    if has_pose:
        n_lin = len(Y[0]) - 4
        ym, D = mean_and_deviations(Y, wm, n_lin)
        innov = np.concatenate([np.asarray(meas[:n_lin]) - ym[:n_lin], qdiff(np.asarray(meas[n_lin:]), ym[n_lin:])])
    else:
        ym = sum(w * y for y, w in zip(Y, wm))
        D = np.array([y - ym for y in Y]).T
        innov = np.asarray(meas) - ym
    X = np.array([np.concatenate([s[:9] - mean[:9], qdiff(s[9:], mean[9:])]) for s in states]).T    # state dof rows only
    Py = (D * wc) @ D.T
    Pxy = (X * wc) @ D.T
    K = Pxy @ np.linalg.inv(Py)
    d = K @ innov
    out = np.empty(13)
    out[:9] = mean[:9] + d[:9]
    out[9:] = boxplus(mean[9:], d[9:12])
    return out, P - K @ Py @ K.T
