"""On-disk formats either side of the filtering path.

These are the files the reference's Dataset* sources read and its logger writes, so that the engine can
be fed from -- and its output compared with -- a Fast-YCB / HO-3D style directory:

  optical_flow/<set>/<index>.float   OpticalFlowUtils::read_flow / save_flow
                                      (src/roft-lib/src/OpticalFlowUtilities.cpp:26-136)
  depth/<index>.float                 tools/dataset/conversion/ho3d_utils.py:65-79
  masks/<set>/<object>_<index>.png    DatasetImageSegmentation::read_file (src/roft-lib/src/DatasetImageSegmentation.cpp:128-147)
  <set>/poses.txt                     x y z ax ay az angle per frame, all-zero row = invalid
                                      (tools/dataset/conversion/utils.py:150-163, tools/dataset/dope_pose_finder/pose_finder.py:23-27)
  data.txt                            stamp_rgb stamp_depth cam_x cam_y cam_z ax ay az angle
                                      (tools/dataset/data_txt_generation/generate_data_txt.py:20-24)
  <name>.obj                          Wavefront mesh: `v x y z [r g b]`, `vn`, `f a//a b//b c//c`
                                      (src/roft-lib/meshes/DOPE/*.obj, src/roft-lib/src/SICADModel.cpp:99-110)
  pose_estimate / velocity_estimate   ROFTFilter's log rows (src/roft-lib/src/ROFTFilter.cpp:386-394,448-451)

Pure numpy / stdlib (zlib for PNG); no OpenCV.
"""
import json
import os
import struct
import zlib

import numpy as np

CV_16SC2 = 11
CV_32FC2 = 13


# ---- optical flow .float ------------------------------------------------------------------------------
def save_flow(flow, path):
    """flow: [rows, cols, 2] int16 (S10.5 fixed point) or float32."""
    flow = np.ascontiguousarray(flow)
    if flow.ndim != 3 or flow.shape[2] != 2 or flow.dtype not in (np.int16, np.float32):
        raise ValueError("only CV_32FC2 or CV_16SC2 frames are supported")
    typ = CV_16SC2 if flow.dtype == np.int16 else CV_32FC2
    with open(path, "wb") as f:
        f.write(struct.pack("=i", typ))                                  # int frame_type
        f.write(struct.pack("=QQ", flow.shape[1], flow.shape[0]))        # size_t {cols, rows}
        f.write(flow.tobytes())


def read_flow(path):
    """Returns (valid, flow) like OpticalFlowUtils::read_flow; never raises on a missing / short file."""
    try:
        with open(path, "rb") as f:
            head = f.read(4 + 16)
            if len(head) != 20:
                return False, None
            typ, = struct.unpack("=i", head[:4])
            cols, rows = struct.unpack("=QQ", head[4:])
            if typ not in (CV_16SC2, CV_32FC2):
                return False, None
            dt = np.int16 if typ == CV_16SC2 else np.float32
            n = cols * rows * 2
            data = np.frombuffer(f.read(n * np.dtype(dt).itemsize), dtype=dt)
            if data.size != n:
                return False, None
            return True, data.reshape(rows, cols, 2).copy()
    except (OSError, ValueError):
        return False, None


def flow_format(flow, image_width):
    """(type, grid, scale) as DatasetImageOpticalFlow derives them (DatasetImageOpticalFlow.cpp:46-50)."""
    typ = CV_16SC2 if flow.dtype == np.int16 else CV_32FC2
    return typ, image_width // flow.shape[1], (32.0 if typ == CV_16SC2 else 1.0)


# ---- depth .float -----------------------------------------------------------------------------------------
def write_depth(path, depth):
    depth = np.ascontiguousarray(depth, np.float32)
    with open(path, "wb") as f:
        f.write(struct.pack("=Q", depth.shape[1]))
        f.write(struct.pack("=Q", depth.shape[0]))
        f.write(depth.tobytes())


def read_depth(path):
    with open(path, "rb") as f:
        w, h = struct.unpack("=QQ", f.read(16))
        d = np.frombuffer(f.read(w * h * 4), np.float32)
    if d.size != w * h:
        raise ValueError("truncated depth frame " + path)
    return d.reshape(h, w).copy()


# ---- poses.txt / data.txt ---------------------------------------------------------------------------------
def axis_angle_to_quat(axis, angle):
    axis = np.asarray(axis, float)
    n = np.linalg.norm(axis)
    if n == 0.0:
        return np.array([1.0, 0.0, 0.0, 0.0])
    s = np.sin(angle / 2.0)
    return np.concatenate([[np.cos(angle / 2.0)], s * axis / n])


def quat_to_axis_angle(q):
    """Eigen::AngleAxisd(Quaterniond): angle in [0, pi], axis (1,0,0) for the identity."""
    q = np.asarray(q, float)
    n = np.linalg.norm(q[1:])
    if n == 0.0:
        return np.array([1.0, 0.0, 0.0]), 0.0
    angle = 2.0 * np.arctan2(n, abs(q[0]))
    return q[1:] / (n if q[0] >= 0 else -n), angle


def read_poses(path, skip_rows=0, skip_cols=0):
    """Rows `[skip_cols values] x y z ax ay az angle`; returns (pose [F,7] = x + quaternion wxyz, valid [F]).
    An all-zero pose row is an invalid detection."""
    rows = []
    with open(path) as f:
        for i, line in enumerate(f):
            if i < skip_rows:
                continue
            vals = [float(v) if v.lower() != "nan" else np.nan for v in line.split()]
            if len(vals) >= skip_cols + 7:
                rows.append(vals[skip_cols:skip_cols + 7])
    a = np.array(rows, float).reshape(-1, 7)
    valid = np.any(a != 0.0, axis=1)
    pose = np.zeros((len(a), 7))
    pose[:, 3] = 1.0
    for k in np.nonzero(valid)[0]:
        pose[k, :3] = a[k, :3]
        pose[k, 3:] = axis_angle_to_quat(a[k, 3:6], a[k, 6])
    return pose, valid


def read_data_txt(path):
    """Returns (stamp_rgb [F], stamp_depth [F], camera_pose [F,7 axis-angle form])."""
    a = np.loadtxt(path, ndmin=2)
    return a[:, 0], a[:, 1], a[:, 2:9]


def read_log(path, skip_cols=0):
    """One of the tracker's log files (bfl::Logger rows: space-separated numbers, padded to a common width) as the
    reference's evaluation reads it (evaluation/data_loader.py:99-108, `load_generic`; `skip_cols=6` drops the velocity
    columns of `pose_estimate[_ycb]` like `load_ours` does, :238-241)."""
    rows = []
    with open(path, newline="") as f:
        for row in f:
            rows.append([float(tok.rstrip()) for tok in row.rstrip().split(sep=" ") if tok != ""])
    a = np.array(rows)
    return a[:, skip_cols:] if skip_cols else a


def pose_log_row(pose13):
    """One row of ROFTFilter's `pose_estimate` log: v w x axis angle (ROFTFilter.cpp:386-394)."""
    r = np.asarray(pose13, float)
    axis, angle = quat_to_axis_angle(r[9:13])
    return np.concatenate([r[:9], axis, [angle]])


def write_cam_k(path, c):
    """cam_K.json of a sequence (read by test/test.sh:46-49 and by evaluation/data_loader.py:110-121)."""
    with open(path, "w") as f:
        json.dump(dict(width=c.width, height=c.height, fx=c.fx, fy=c.fy, cx=c.cx, cy=c.cy), f)


def write_data_txt(path, n_frames, fps=30.0):
    """data.txt as tools/dataset/data_txt_generation/generate_data_txt.py:16-27 writes it: one row per frame,
    `stamp_rgb stamp_depth` = i / fps twice and the identity camera pose `0.0 0.0 0.0 1.0 0.0 0.0 0.0`."""
    with open(path, "w") as f:
        for i in range(n_frames):
            stamp = (1.0 / fps) * i
            f.write(str(stamp) + " " + str(stamp) + " 0.0 0.0 0.0 1.0 0.0 0.0 0.0\n")


def find_initial_pose(path, fps):
    """What test/test_ho3d.sh:68 asks tools/dataset/dope_pose_finder/pose_finder.py (:13-33) for: the first row of a 30 fps
    poses.txt that is not the invalid detection `0.0 0.0 0.0 0.0 0.0 0.0 0.0` AND lies on the grid of a source running at
    `fps`; returns (frame the tracker starts at -- the row index + 6, the delay of the source, unless it is row 0 --,
    the row's text) or None."""
    steps = (1.0 / fps) / (1 / 30.0)
    invalid = ("0.0 " * 7)[:-1]
    with open(path) as f:
        for i, line in enumerate(f.readlines()):
            line = line.rstrip()
            if line != invalid and i % steps == 0:
                return (i + 6 if i != 0 else 0), line
    return None


def write_estimate_logs(prefix, pose13, twist6):
    """ROFTFilter's `pose_estimate` (v w x axis angle, 13 columns) and `velocity_estimate` (6 columns)."""
    pose13 = np.atleast_2d(pose13)
    twist6 = np.atleast_2d(twist6)
    with open(prefix + "pose_estimate", "w") as fp:
        for r in pose13:
            axis, angle = quat_to_axis_angle(r[9:13])
            fp.write(" ".join("%.12g" % v for v in list(r[:9]) + list(axis) + [angle]) + "\n")
    with open(prefix + "velocity_estimate", "w") as fv:
        for r in twist6:
            fv.write(" ".join("%.12g" % v for v in r) + "\n")


def delivery_schedule(n_frames, original_fps=30.0, desired_fps=5.0, head_0=0, simulate_inference_time=True):
    """Frame index delivered at each frame by DatasetImageSegmentationDelayed / DatasetTransformDelayed, -1 = none
    (src/roft-lib/src/DatasetImageSegmentationDelayed.cpp:42-63)."""
    delay = int(original_fps / desired_fps)
    out = np.full(n_frames, -1, np.int64)
    for h in range(n_frames):
        head = head_0 + h
        index = head - delay if simulate_inference_time else head
        if abs(index - head_0) % delay != 0:   # C's % keeps the sign; only "== 0" matters
            continue
        if index < 0:
            index = head_0
        out[h] = index
    return out


# ---- Wavefront OBJ --------------------------------------------------------------------------------------
def load_obj(path):
    """Vertices (float32 [V,3]) and triangles (int32 [T,3], 0-based).  Faces with more than three corners
    are fanned; `v//vn`, `v/vt/vn`, `v/vt` and bare `v` corner forms are accepted; negative indices are
    relative to the end."""
    verts, tris = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                idx = []
                for c in line.split()[1:]:
                    i = int(c.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    tris.append((idx[0], idx[k], idx[k + 1]))
    return np.array(verts, np.float32).reshape(-1, 3), np.array(tris, np.int32).reshape(-1, 3)


# ---- PNG (8-bit, non-interlaced; gray, gray+alpha, RGB, RGBA, palette) ------------------------------------
def read_png(path):
    """Decodes to uint8 [H, W] (gray / palette index) or [H, W, C]."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file: " + path)
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        length, = struct.unpack(">I", data[pos:pos + 4])
        ctype = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        pos += 12 + length
        if ctype == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif ctype == b"IDAT":
            idat.append(body)
        elif ctype == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0:
        raise ValueError("only 8-bit non-interlaced PNGs are supported")
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    stride = w * ch
    raw = raw.reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:   # Sub: running sum per channel
            cur = (np.cumsum(line.reshape(-1, ch), axis=0) & 255).reshape(-1).astype(np.int32)
        else:
            cur = np.zeros(stride, np.int32)
            for x in range(stride):
                a = cur[x - ch] if x >= ch else 0
                b = prev[x]
                c = prev[x - ch] if x >= ch else 0
                if ft == 1:
                    pr = a
                elif ft == 3:
                    pr = (a + b) >> 1
                elif ft == 4:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                else:
                    raise ValueError("bad PNG filter")
                cur[x] = (line[x] + pr) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w) if ch == 1 else out.reshape(h, w, ch)


def write_png(path, img):
    """8-bit gray [H, W] or RGB(A) [H, W, C] PNG, filter 0 (used by the synthetic dataset writer and the tests)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    raw = np.zeros((h, w * ch + 1), np.uint8)
    raw[:, 1:] = img.reshape(h, w * ch)

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw.tobytes(), 3)) + chunk(b"IEND", b""))


def rgb_to_gray(img):
    """cv::cvtColor(..., COLOR_BGR2GRAY) on 8-bit data: fixed point, (R*4899 + G*9617 + B*1868 + 2^13) >> 14.
    `img` is [H, W, >=3] in R,G,B order (as PNG stores it); a gray image passes through."""
    if img.ndim == 2:
        return img
    c = img[..., :3].astype(np.int32)
    return ((c[..., 0] * 4899 + c[..., 1] * 9617 + c[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


def read_mask_png(path):
    """Mask as ImageSegmentationMeasurement sees it before thresholding: colour images are converted to gray
    (cv::cvtColor BGR2GRAY), src/roft-lib/src/ImageSegmentationMeasurement.cpp:62-63."""
    return rgb_to_gray(read_png(path))


# ---- a Fast-YCB style sequence --------------------------------------------------------------------------
class Sequence:
    """Iterates the per-frame inputs of one object of a Fast-YCB / HO-3D style directory in the form
    `roft_amd.engine.ROFTFilterBatch.submit` takes (HOST buffers), applying the reference's 5 fps / delayed
    delivery schedule to masks and poses."""

    def __init__(self, root, object_name, flow_set="nvof_1_slow", mask_set="mrcnn_ycbv_bop_pbr", pose_set="dope",
                 width=1280, height=720, original_fps=30.0, desired_fps=5.0, delayed=True, first_frame=0):
        """first_frame: where the tracker starts (test/test_ho3d.sh:142-160 passes it as `index_offset` of the camera, flow and
        mask sources and as `skip_rows` of the pose file): the mask schedule counts from that frame on file indices, the
        pose schedule on the rows that are left -- the first frame delivers ITS OWN row, later ones the row six frames back."""
        self.root, self.obj = root, object_name
        self.flow_dir = os.path.join(root, "optical_flow", flow_set)
        self.mask_dir = os.path.join(root, "masks", mask_set)
        self.depth_dir = os.path.join(root, "depth")
        self.stamp, _, _ = read_data_txt(os.path.join(root, "data.txt"))
        self.poses, self.pose_ok = read_poses(os.path.join(root, pose_set, "poses.txt"))
        self.n = len(self.stamp)
        self.size = (width, height)
        first, run = int(first_frame), self.n - int(first_frame)
        self.mask_src = np.full(self.n, -1, np.int64)
        self.pose_src = np.full(self.n, -1, np.int64)
        if delayed:
            self.mask_src[first:] = delivery_schedule(run, original_fps, desired_fps, head_0=first)
            rows = delivery_schedule(run, original_fps, desired_fps)
            self.pose_src[first:] = np.where(rows >= 0, rows + first, -1)
        else:
            self.mask_src[first:] = self.pose_src[first:] = np.arange(first, self.n)

    def __len__(self):
        return self.n

    def frame(self, k):
        ok, flow = read_flow(os.path.join(self.flow_dir, "%d.float" % k))
        mi, pi = self.mask_src[k], self.pose_src[k]
        mask = None
        if mi >= 0:
            p = os.path.join(self.mask_dir, "%s_%d.png" % (self.obj, mi))
            mask = read_mask_png(p) if os.path.exists(p) else None
        pose = None
        if pi >= 0 and self.pose_ok[pi]:
            pose = (self.poses[pi, :3], self.poses[pi, 3:])
        dt = float(self.stamp[k] - self.stamp[k - 1]) if k > 0 else 0.0
        return dict(depth=read_depth(os.path.join(self.depth_dir, "%d.float" % k)), flow=flow if ok else None,
                    mask=mask, pose=pose, dt=dt)


# ---- writing a stream in the Fast-YCB layout (synthetic data on disk, closes the loop with Sequence / the dumper) ----
def write_obj(path, verts, tris):
    with open(path, "w") as f:
        for v in np.asarray(verts, np.float64):
            f.write("v %.9g %.9g %.9g\n" % tuple(v))
        for t in np.asarray(tris, np.int64):
            f.write("f %d %d %d\n" % tuple(t + 1))


def write_poses(path, pose7, valid=None):
    """Rows `x y z ax ay az angle` (the layout read_poses / the reference's pose files use); invalid rows are all zero."""
    pose7 = np.atleast_2d(pose7)
    with open(path, "w") as f:
        for i, r in enumerate(pose7):
            if valid is not None and not valid[i]:
                # (spelled as the reference's tools recognise an invalid detection: tools/dataset/dope_pose_finder/pose_finder.py:23)
                f.write(" ".join(["0.0"] * 7) + "\n")
                continue
            axis, angle = quat_to_axis_angle(r[3:7])
            f.write(" ".join("%.17g" % v for v in list(r[:3]) + list(axis) + [angle]) + "\n")


def write_sequence(root, st, object_name, mask_set="gt", pose_set="dope", flow_set=None):
    """Writes a roft_amd.synth stream as a Fast-YCB style directory: data.txt, cam_K.json, rgb/<i>.png (when the stream
    carries gray images), depth/<i>.float, masks/<mask_set>/<object>_<i>.png, <pose_set>/poses.txt (the per-frame
    detections the delayed source replays), gt/poses.txt, model.obj and -- if flow_set is given -- the stream's own
    optical_flow/<flow_set>/<i>.float.  Returns the path of the mesh."""
    n = int(st.n_frames)
    for d in ("rgb", "depth", os.path.join("masks", mask_set), pose_set, "gt"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    write_data_txt(os.path.join(root, "data.txt"), n, 1.0 / st.dt)
    write_cam_k(os.path.join(root, "cam_K.json"), st.camera)
    depth, masks = st.depth.cpu().numpy(), st.mask_gt.cpu().numpy()
    gray = st.gray.cpu().numpy() if getattr(st, "gray", None) is not None else None
    flow = st.flow.cpu().numpy() if flow_set else None
    if flow_set:
        os.makedirs(os.path.join(root, "optical_flow", flow_set), exist_ok=True)
    for k in range(n):
        write_depth(os.path.join(root, "depth", "%d.float" % k), depth[k])
        write_png(os.path.join(root, "masks", mask_set, "%s_%d.png" % (object_name, k)), masks[k])
        if gray is not None:
            write_png(os.path.join(root, "rgb", "%d.png" % k), np.repeat(gray[k][..., None], 3, axis=2))
        if flow_set and st.flow_valid[k]:
            save_flow(flow[k], os.path.join(root, "optical_flow", flow_set, "%d.float" % k))
    # detections: the stream holds what is DELIVERED at frame k (= the detection of frame sched[k])
    sched = delivery_schedule(n)
    det = np.zeros((n, 7))
    det[:, 3] = 1.0
    ok = np.zeros(n, bool)
    for k in range(n):
        if sched[k] >= 0 and st.pose_valid[k]:
            det[sched[k]] = st.pose_meas[k]
            ok[sched[k]] = True
    write_poses(os.path.join(root, pose_set, "poses.txt"), det, ok)
    write_poses(os.path.join(root, "gt", "poses.txt"), np.concatenate([st.gt.x, st.gt.q], 1))
    mesh = os.path.join(root, "model.obj")
    write_obj(mesh, *st.mesh)
    return mesh
