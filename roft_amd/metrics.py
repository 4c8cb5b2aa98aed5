"""Pose-error metrics of the reference's evaluation: ADD, ADD-S (BOP "adi") and their AUC.

Definitions follow tools/third_party/bop_pose_error.py:73-108 (add / adi, nearest neighbour from the
ground-truth points into the estimated points) and evaluation/metrics.py:303-344 (threshold 0.1 m ->
inf, sort, accuracy = cumsum(1)/n, VOCap * 100).  Pinned by tests/golden/bop_fixtures.json.
"""
import numpy as np
from scipy import spatial


def quat_to_rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def add(R_est, t_est, R_gt, t_gt, pts):
    pe = pts @ np.asarray(R_est).T + np.asarray(t_est)
    pg = pts @ np.asarray(R_gt).T + np.asarray(t_gt)
    return float(np.linalg.norm(pe - pg, axis=1).mean())


def adds(R_est, t_est, R_gt, t_gt, pts):
    pe = pts @ np.asarray(R_est).T + np.asarray(t_est)
    pg = pts @ np.asarray(R_gt).T + np.asarray(t_gt)
    d, _ = spatial.cKDTree(pe).query(pg, k=1)
    return float(d.mean())


def auc(distances, threshold=0.1):
    d = np.array(distances, dtype=np.float64)
    d[d > threshold] = np.inf
    d = np.sort(d)
    n = len(d)
    if n == 0:
        return 0.0
    acc = np.cumsum(np.ones((n,), np.float32)) / n
    fin = np.isfinite(d)
    rec, prec = d[fin], acc[fin]
    if len(rec) == 0:
        return 0.0
    mrec = np.concatenate([[0.0], rec, [threshold]])
    mpre = np.concatenate([[0.0], prec, [prec[-1]]])
    mpre = np.maximum.accumulate(mpre)
    i = np.where(mrec[1:] != mrec[:-1])[0] + 1
    return float(np.sum((mrec[i] - mrec[i - 1]) * mpre[i]) * 10.0 * 100.0)


def trajectory_adds(pose_est, pose_ref, pts):
    """pose_*: [F, 7] (x, q wxyz).  Returns the per-frame ADD-S distances."""
    out = np.zeros(len(pose_est))
    for k in range(len(pose_est)):
        out[k] = adds(quat_to_rot(pose_est[k, 3:]), pose_est[k, :3], quat_to_rot(pose_ref[k, 3:]), pose_ref[k, :3], pts)
    return out


# ---- RMSE metrics of evaluation/metrics.py (the step right after the filtering path) -----------------------
def _rot_angle_deg(q_ref, q_sig):
    """Geodesic angle of R_ref R_sig^T in degrees (metrics.py:114-144)."""
    d = abs(float(np.dot(q_ref, q_sig))) / (np.linalg.norm(q_ref) * np.linalg.norm(q_sig))
    return np.degrees(2.0 * np.arccos(min(1.0, d)))


def rmse_cartesian_3d(ref_x, sig_x):
    """RMSE of the 3D position error in centimetres (metrics.py:102-111, 217-222)."""
    err = np.linalg.norm((np.asarray(ref_x) - np.asarray(sig_x)) * 100.0, axis=1)
    return float(np.linalg.norm(err) / np.sqrt(len(err)))


def rmse_angular(ref_q, sig_q):
    """RMSE of the orientation error in degrees (metrics.py:114-144, 243-248); quaternions (w,x,y,z)."""
    err = np.array([_rot_angle_deg(a, b) for a, b in zip(ref_q, sig_q)])
    return float(np.linalg.norm(err) / np.sqrt(len(err)))


def object_velocity_from_twist(twist, position):
    """The filter's linear velocity is that of the object point at the camera origin; the evaluation
    moves the pole to the object position before comparing: v = v_O + w x r (evaluate.py:514-521)."""
    twist = np.asarray(twist, float)
    out = twist.copy()
    out[:, :3] = twist[:, :3] + np.cross(twist[:, 3:], np.asarray(position, float))
    return out


def rmse_linear_velocity(ref_v, sig_v):
    """cm/s (metrics.py:165-174, 251-256)."""
    err = np.linalg.norm((np.asarray(ref_v) - np.asarray(sig_v)) * 100.0, axis=1)
    return float(np.linalg.norm(err) / np.sqrt(len(err)))


def rmse_angular_velocity(ref_w, sig_w):
    """deg/s (metrics.py:177-186, 259-264)."""
    err = np.linalg.norm(np.degrees(np.asarray(ref_w) - np.asarray(sig_w)), axis=1)
    return float(np.linalg.norm(err) / np.sqrt(len(err)))


def time_metrics(exec_ms):
    """Mean execution time and the number of frames above the 33 ms real-time budget (metrics.py:347-369)."""
    t = np.asarray(exec_ms, float)
    return float(t.mean()), int((t > 33.0).sum())


# ---- the reference's Metric interface (evaluation/metrics.py:20-369) ---------------------------------------------------
class Metric:
    """`Metric(name).evaluate(object_name, reference, signal, time)` as evaluation/evaluate.py calls it.  `reference` / `signal`:
    arrays whose rows are what evaluation/data_loader.py hands over -- poses `x y z axis angle` (7 columns) for the pose
    metrics, `v w` (6 columns) for the velocity metrics --, `time`: rows `[execution_ms, loading_ms]`; for the object name
    'ALL' all three are dicts name -> array and the rows of all objects are pooled (metrics.py:72-81).  `auc_points`: name ->
    [P, 3] model points for 'add' / 'adi' (the reference reads YCB_Video_Models/<name>/points.xyz, metrics.py:47-49)."""

    NAMES = ("rmse_cartesian_3d", "rmse_cartesian_x", "rmse_cartesian_y", "rmse_cartesian_z", "rmse_angular", "rmse_linear_velocity",
             "rmse_angular_velocity", "max_linear_velocity", "max_angular_velocity", "add", "adi", "time", "excess_33_ms")

    def __init__(self, name, auc_points=None):
        if name not in self.NAMES:
            raise ValueError("Metric " + name + " does not exist.")
        self.name = name
        self.auc_points = auc_points or {}

    @staticmethod
    def _pool(object_name, x):
        if object_name == "ALL":
            return np.concatenate([np.asarray(x[k], float) for k in x], axis=0)
        return np.asarray(x, float)

    @staticmethod
    def _rot(aa):
        """axis (3) + angle -> rotation matrix, the axis used as given (pyquaternion normalises it; pose files hold unit axes)."""
        from .io import axis_angle_to_quat
        return quat_to_rot(axis_angle_to_quat(aa[:3], aa[3]))

    @staticmethod
    def _rms(err):
        return float(np.linalg.norm(err) / np.sqrt(err.shape[0]))

    def evaluate(self, object_name, reference, signal, time):
        n = self.name
        if n in ("time", "excess_33_ms"):
            t = self._pool(object_name, time)[:, 0]
            return float(t.mean()) if n == "time" else float((t > 33.0).sum())
        if n in ("add", "adi"):
            return self.auc(object_name, reference, signal, n)[1]
        ref, sig = self._pool(object_name, reference), self._pool(object_name, signal)
        if n == "rmse_cartesian_3d":
            return self._rms(np.linalg.norm((ref[:, :3] - sig[:, :3]) * 100.0, axis=1))                  # cm
        if n.startswith("rmse_cartesian_"):
            i = "xyz".index(n[-1])
            return self._rms((ref[:, i] - sig[:, i]) * 100.0)
        if n == "rmse_angular":
            err = np.empty(len(ref))
            for k in range(len(ref)):
                R = self._rot(ref[k, 3:7]) @ self._rot(sig[k, 3:7]).T
                err[k] = np.degrees(np.arccos(np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)))          # |log R|
            return self._rms(err)
        if n == "rmse_linear_velocity":
            return self._rms(np.linalg.norm((ref[:, :3] - sig[:, :3]) * 100.0, axis=1))                  # cm/s
        if n == "rmse_angular_velocity":
            return self._rms(np.linalg.norm(np.degrees(ref[:, 3:6] - sig[:, 3:6]), axis=1))              # deg/s
        if n == "max_linear_velocity":
            return float(np.linalg.norm(ref[:, :3], axis=1).max())                                       # of the REFERENCE (metrics.py:189-198)
        return float(np.degrees(np.linalg.norm(ref[:, 3:6], axis=1).max()))

    def auc(self, object_name, reference, signal, ad_name):
        """(distances, AUC x 100) of ADD ('add') or ADD-S ('adi'), pooled over the objects for 'ALL' (metrics.py:303-344)."""
        if object_name == "ALL":
            names = list(signal)
        else:
            names, signal, reference = [object_name], {object_name: signal}, {object_name: reference}
        dists = []
        for name in names:
            pts = np.asarray(self.auc_points[name], float)
            for r, s in zip(np.asarray(reference[name], float), np.asarray(signal[name], float)):
                f = add if ad_name == "add" else adds
                dists.append(f(self._rot(s[3:7]), s[:3], self._rot(r[3:7]), r[:3], pts))
        dists = np.array(dists)
        return dists, auc(dists)
