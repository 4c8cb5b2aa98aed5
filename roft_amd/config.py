"""ROFT configuration files -> engine configuration.

The reference's executable reads a libconfig file (config/config_fast_ycb.cfg, config/config_ho3d.cfg) through its
ConfigParser -- every setting can be overridden on the command line as `--group::key value`, arrays as "x_1, ..., x_n"
(src/roft/include/ConfigParser.h:21-60, src/roft/src/ConfigParser.cpp:57-130) -- and packs the values into the
arguments of ROFTFilter's constructor (src/roft/src/main.cpp:43-147 keys, :286-325 packing, :346-392 sources).  This
module is that path for the MI355X engine: the same files and overrides in, `roft_config` + `roft_object_desc`
(include/roft_engine.h) out, plus the data-set settings the sequence reader needs.  Host-side Python, no arithmetic of
the hot path.
"""
import math
import re

from . import _lib as L

_TOKEN = re.compile(r"""\s*(?:(?P<str>"(?:[^"\\]|\\.)*")|(?P<num>[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)L{0,2})|
                        (?P<name>[A-Za-z_*][-A-Za-z0-9_*]*)|(?P<op>[{}\[\]():=;,]))""", re.X)


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = []
    for line in text.splitlines():
        quoted = False
        for i, ch in enumerate(line):
            if ch == '"' and (i == 0 or line[i - 1] != "\\"):
                quoted = not quoted
            if not quoted and (ch == "#" or line.startswith("//", i)):
                line = line[:i]
                break
        out.append(line)
    return "\n".join(out)


def _tokens(text):
    pos, text = 0, _strip_comments(text)
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                return
            raise ValueError("cfg: cannot parse near %r" % text[pos:pos + 30])
        pos = m.end()
        if m.group("str") is not None:
            yield "val", bytes(m.group("str")[1:-1], "utf-8").decode("unicode_escape")
        elif m.group("num") is not None:
            s = m.group("num").rstrip("L")
            yield "val", (int(s) if re.fullmatch(r"[-+]?\d+", s) else float(s))
        elif m.group("name") is not None:
            n = m.group("name")
            if n.lower() in ("true", "false"):
                yield "val", n.lower() == "true"
            else:
                yield "name", n
        else:
            yield "op", m.group("op")


def parse_cfg(text):
    """libconfig subset used by ROFT: `name = value;`, `name: { ... }` groups, `[a, b, ...]` arrays, `( ... )` lists,
    strings, integers, floats, booleans, # // /* */ comments -> nested dict."""
    toks = list(_tokens(text))
    pos = [0]

    def peek():
        return toks[pos[0]] if pos[0] < len(toks) else (None, None)

    def take(kind=None, value=None):
        k, v = peek()
        if k is None or (kind and k != kind) or (value is not None and v != value):
            raise ValueError("cfg: expected %s %s, found %r" % (kind, value, (k, v)))
        pos[0] += 1
        return v

    def value():
        k, v = peek()
        if k == "val":
            return take()
        if (k, v) == ("op", "{"):
            take()
            g = group("}")
            take("op", "}")
            return g
        if k == "op" and v in "[(":
            close = "]" if take() == "[" else ")"
            items = []
            while peek() != ("op", close):
                items.append(value())
                if peek() == ("op", ","):
                    take()
            take("op", close)
            return items
        raise ValueError("cfg: unexpected token %r" % ((k, v),))

    def group(closing):
        g = {}
        while peek()[0] is not None and peek() != ("op", closing):
            name = take("name")
            if peek()[0] != "op" or peek()[1] not in ":=":
                raise ValueError("cfg: expected ':' or '=' after %s" % name)
            take()
            g[name] = value()
            while peek()[0] == "op" and peek()[1] in ";,":
                take()
        return g

    root = group(None)
    if pos[0] != len(toks):
        raise ValueError("cfg: trailing tokens")
    return root


def lookup(cfg, path):
    node = cfg
    for part in re.split(r"::|\.", path):
        if not isinstance(node, dict) or part not in node:
            raise KeyError("cannot find the setting with name " + path)
        node = node[part]
    return node


def apply_overrides(cfg, argv):
    """`--a::b::c value` pairs (as test/test.sh passes them to ROFT-tracker); the value is converted to the type the file
    gives the setting.  `--from` (the configuration file itself) is skipped.  Returns the arguments that are not settings."""
    rest, i = [], 0
    while i < len(argv):
        a = argv[i]
        if a.startswith("--") and i + 1 < len(argv) and a[2:] != "from":
            try:
                old = lookup(cfg, a[2:])
            except KeyError:
                rest.append(a)
                i += 1
                continue
            raw = argv[i + 1]
            if isinstance(old, bool):
                new = raw.strip().lower() == "true"
            elif isinstance(old, list):
                conv = int if old and all(isinstance(x, int) and not isinstance(x, bool) for x in old) else float
                new = [conv(x) for x in raw.replace(",", " ").split()]
                if len(new) != len(old):
                    raise ValueError("%s expects %d values" % (a, len(old)))
            elif isinstance(old, int):
                new = int(raw)
            elif isinstance(old, float):
                new = float(raw)
            else:
                new = raw
            parts = re.split(r"::|\.", a[2:])
            node = cfg
            for part in parts[:-1]:
                node = node[part]
            node[parts[-1]] = new
            i += 2
        elif a == "--from":
            i += 2
        else:
            rest.append(a)
            i += 1
    return rest


# settings the filter consumes (src/roft/src/main.cpp:43-147 -> ROFTFilter.cpp:32-201); every other key of the files is
# data-set plumbing returned in `extras`
FILTER_KEYS = [
    "sample_time",
    "camera_dataset.width", "camera_dataset.height", "camera_dataset.fx", "camera_dataset.fy", "camera_dataset.cx", "camera_dataset.cy",
    "initial_condition.pose.v", "initial_condition.pose.w", "initial_condition.pose.x", "initial_condition.pose.axis_angle",
    "initial_condition.pose.cov_v", "initial_condition.pose.cov_w", "initial_condition.pose.cov_x", "initial_condition.pose.cov_q",
    "initial_condition.velocity.v", "initial_condition.velocity.w", "initial_condition.velocity.cov_v", "initial_condition.velocity.cov_w",
    "kinematic_model.pose.sigma_linear", "kinematic_model.pose.sigma_angular",
    "kinematic_model.velocity.sigma_linear", "kinematic_model.velocity.sigma_angular",
    "measurement_model.pose.cov_v", "measurement_model.pose.cov_w", "measurement_model.pose.cov_x", "measurement_model.pose.cov_q",
    "measurement_model.velocity.cov_flow", "measurement_model.velocity.depth_maximum", "measurement_model.velocity.subsampling_radius",
    "measurement_model.velocity.weight_flow",
    "measurement_model.use_pose", "measurement_model.use_pose_resync", "measurement_model.use_velocity",
    "outlier_rejection.enable", "outlier_rejection.gain",
    "pose_dataset.fps_reduction", "pose_dataset.delay", "pose_dataset.original_fps", "pose_dataset.desired_fps",
    "segmentation_dataset.fps_reduction", "segmentation_dataset.delay", "segmentation_dataset.original_fps",
    "segmentation_dataset.desired_fps", "segmentation_dataset.flow_aided",
    "unscented_transform.alpha", "unscented_transform.beta", "unscented_transform.kappa",
]
DATASET_KEYS = [
    "camera_dataset.path", "camera_dataset.data_prefix", "camera_dataset.rgb_prefix", "camera_dataset.depth_prefix",
    "camera_dataset.data_format", "camera_dataset.rgb_format", "camera_dataset.depth_format", "camera_dataset.heading_zeros",
    "camera_dataset.index_offset",
    "log.enable", "log.enable_segmentation", "log.path",
    "model.name", "model.use_internal_db", "model.internal_db_name", "model.external_path",
    "optical_flow_dataset.path", "optical_flow_dataset.set", "optical_flow_dataset.heading_zeros", "optical_flow_dataset.index_offset",
    "pose_dataset.path", "pose_dataset.skip_rows", "pose_dataset.skip_cols",
    "segmentation_dataset.path", "segmentation_dataset.format", "segmentation_dataset.set", "segmentation_dataset.heading_zeros",
    "segmentation_dataset.index_offset",
]


def frames_between(delay, fps_reduction, original_fps, desired_fps):
    """get_frames_between_iterations() of the source main.cpp builds (:346-381): int(original / desired) for the *Delayed
    data-set sources (DatasetImageSegmentationDelayed.cpp:78; desired = original without fps reduction), unknown (-1)
    for the plain ones."""
    if not (delay or fps_reduction):
        return -1
    if not fps_reduction:
        desired_fps = original_fps
    return int(original_fps / desired_fps)


def to_engine(cfg, flow_type, flow_grid=None, flow_scale=None, max_objects=1, max_batch_frames=1, device=0):
    """(roft_config, roft_object_desc, extras) from a parsed configuration.  The flow format is a property of the flow
    frames, not of the file (DatasetImageOpticalFlow.cpp:46-50): grid = width / flow columns, scale 32 for CV_16SC2."""
    from . import engine as E
    g = lambda k: lookup(cfg, k)
    c = E.default_config(int(g("camera_dataset.width")), int(g("camera_dataset.height")), flow_type, max_objects=max_objects,
                         device=device, max_batch_frames=max_batch_frames)
    c.cam.fx, c.cam.fy, c.cam.cx, c.cam.cy = (float(g("camera_dataset." + k)) for k in ("fx", "fy", "cx", "cy"))
    if flow_grid is not None:
        c.flow_grid = int(flow_grid)
    if flow_scale is not None:
        c.flow_scale = float(flow_scale)
    c.sample_time = float(g("sample_time"))
    c.ut.alpha, c.ut.beta, c.ut.kappa = (float(g("unscented_transform." + k)) for k in ("alpha", "beta", "kappa"))
    c.depth_maximum = float(g("measurement_model.velocity.depth_maximum"))
    c.subsampling_radius = float(g("measurement_model.velocity.subsampling_radius"))
    c.flow_weighting = int(bool(g("measurement_model.velocity.weight_flow")))
    c.use_pose = int(bool(g("measurement_model.use_pose")))
    c.use_pose_resync = int(bool(g("measurement_model.use_pose_resync")))
    c.use_velocity = int(bool(g("measurement_model.use_velocity")))
    c.outlier_rejection = int(bool(g("outlier_rejection.enable")))
    # outlier_rejection.gain reaches ROFTFilter through a `const bool` parameter (ROFTFilter.h:64): any non-zero value
    # is 1 there, and the decision is a ratio test the gain cancels out of -- read, reported, not used
    c.flow_aided_segmentation = int(bool(g("segmentation_dataset.flow_aided")))
    c.mask_frames_between = frames_between(g("segmentation_dataset.delay"), g("segmentation_dataset.fps_reduction"),
                                           float(g("segmentation_dataset.original_fps")), float(g("segmentation_dataset.desired_fps")))
    c.pose_frames_between = max(0, frames_between(g("pose_dataset.delay"), g("pose_dataset.fps_reduction"),
                                                  float(g("pose_dataset.original_fps")), float(g("pose_dataset.desired_fps"))))
    o = E.default_object()
    ax = [float(x) for x in g("initial_condition.pose.axis_angle")]
    n = math.sqrt(ax[0] ** 2 + ax[1] ** 2 + ax[2] ** 2)
    h = 0.5 * ax[3]
    q = [math.cos(h)] + [(math.sin(h) * a / n if n > 0 else 0.0) for a in ax[:3]]   # Quaterniond(AngleAxisd(angle, axis)), main.cpp:291
    mean = list(g("initial_condition.pose.v")) + list(g("initial_condition.pose.w")) + list(g("initial_condition.pose.x")) + q
    cov = sum((list(g("initial_condition.pose.cov_" + k)) for k in ("v", "w", "x", "q")), [])
    for i in range(13):
        o.p_mean0[i] = float(mean[i])
    for i in range(12):
        o.p_cov0_diag[i] = float(cov[i])
    vm = list(g("initial_condition.velocity.v")) + list(g("initial_condition.velocity.w"))
    vc = list(g("initial_condition.velocity.cov_v")) + list(g("initial_condition.velocity.cov_w"))
    vq = list(g("kinematic_model.velocity.sigma_linear")) + list(g("kinematic_model.velocity.sigma_angular"))
    for i in range(6):
        o.v_mean0[i], o.v_cov0_diag[i], o.v_q_diag[i] = float(vm[i]), float(vc[i]), float(vq[i])
    for i in range(3):
        # kinematic_model.pose.sigma_linear is the PSD of the linear acceleration, sigma_angular the variance of the angular
        # velocity (main.cpp:78-79; packed swapped :311-313 and unpacked swapped again ROFTFilter.cpp:89-90)
        o.p_psd_lin_acc[i] = float(g("kinematic_model.pose.sigma_linear")[i])
        o.p_sigma_ang_vel[i] = float(g("kinematic_model.pose.sigma_angular")[i])
        o.p_meas_cov_v[i] = float(g("measurement_model.pose.cov_v")[i])
        o.p_meas_cov_w[i] = float(g("measurement_model.pose.cov_w")[i])
        o.p_meas_cov_x[i] = float(g("measurement_model.pose.cov_x")[i])
        o.p_meas_cov_q[i] = float(g("measurement_model.pose.cov_q")[i])
    o.v_meas_cov_flow[0], o.v_meas_cov_flow[1] = (float(x) for x in g("measurement_model.velocity.cov_flow"))
    extras = {}
    for k in DATASET_KEYS + ["outlier_rejection.gain"]:
        try:
            extras[k] = lookup(cfg, k)
        except KeyError:
            pass
    return c, o, extras


def load(path, argv=(), **kw):
    """File + command-line overrides -> (roft_config, roft_object_desc, extras, remaining arguments)."""
    with open(path) as f:
        cfg = parse_cfg(f.read())
    rest = apply_overrides(cfg, list(argv))
    c, o, extras = to_engine(cfg, **kw)
    return c, o, extras, rest


def default_text(width, height, fx, fy, cx, cy):
    """A configuration file in the reference's format holding the defaults of the ABI (roft_default_config /
    roft_default_object: the filter settings of config/config_fast_ycb.cfg) for the given camera."""
    import ctypes as C
    c, o = L.Config(), L.ObjectDesc()
    L.check(L.lib().roft_default_config(C.byref(c), int(width), int(height), L.FLOW_F32C2))
    L.check(L.lib().roft_default_object(C.byref(o)))
    arr = lambda a: "[" + ", ".join(repr(float(x)) for x in a) + "]"
    b = lambda v: "true" if v else "false"
    fps = lambda n: "fps_reduction = true; delay = true; original_fps = 30.0; desired_fps = %r;" % (30.0 / max(n, 1))
    return "\n".join([
        "sample_time = %r;" % c.sample_time,
        "camera_dataset: { width = %d; height = %d; fx = %r; fy = %r; cx = %r; cy = %r; }" % (width, height, float(fx), float(fy), float(cx), float(cy)),
        "initial_condition: { pose: { v = %s; w = %s; x = %s; axis_angle = [1.0, 0.0, 0.0, 0.0]; cov_v = %s; cov_w = %s; cov_x = %s; cov_q = %s; }"
        % (arr(o.p_mean0[0:3]), arr(o.p_mean0[3:6]), arr(o.p_mean0[6:9]), arr(o.p_cov0_diag[0:3]), arr(o.p_cov0_diag[3:6]),
           arr(o.p_cov0_diag[6:9]), arr(o.p_cov0_diag[9:12])),
        "  velocity: { v = %s; w = %s; cov_v = %s; cov_w = %s; } }" % (arr(o.v_mean0[0:3]), arr(o.v_mean0[3:6]), arr(o.v_cov0_diag[0:3]), arr(o.v_cov0_diag[3:6])),
        "kinematic_model: { pose: { sigma_linear = %s; sigma_angular = %s; } velocity: { sigma_linear = %s; sigma_angular = %s; } }"
        % (arr(o.p_psd_lin_acc), arr(o.p_sigma_ang_vel), arr(o.v_q_diag[0:3]), arr(o.v_q_diag[3:6])),
        "measurement_model: { pose: { cov_v = %s; cov_w = %s; cov_x = %s; cov_q = %s; }" % (arr(o.p_meas_cov_v), arr(o.p_meas_cov_w), arr(o.p_meas_cov_x), arr(o.p_meas_cov_q)),
        "  velocity: { cov_flow = %s; depth_maximum = %r; subsampling_radius = %r; weight_flow = %s; }" % (arr(o.v_meas_cov_flow), c.depth_maximum, c.subsampling_radius, b(c.flow_weighting)),
        "  use_pose = %s; use_pose_resync = %s; use_velocity = %s; }" % (b(c.use_pose), b(c.use_pose_resync), b(c.use_velocity)),
        "outlier_rejection: { enable = %s; gain = 0.01; }" % b(c.outlier_rejection),
        "pose_dataset: { %s }" % fps(c.pose_frames_between),
        "segmentation_dataset: { %s flow_aided = %s; }" % (fps(c.mask_frames_between), b(c.flow_aided_segmentation)),
        "unscented_transform: { alpha = %r; beta = %r; kappa = %r; }" % (c.ut.alpha, c.ut.beta, c.ut.kappa), ""])


def dump_cfg(cfg, indent=0):
    """Nested dict -> text in the reference's format (the inverse of parse_cfg)."""
    pad, out = "    " * indent, []
    for k, v in cfg.items():
        if isinstance(v, dict):
            out += ["%s%s:" % (pad, k), pad + "{", dump_cfg(v, indent + 1).rstrip("\n"), pad + "}"]
        else:
            one = lambda x: ("true" if x else "false") if isinstance(x, bool) else ('"%s"' % x.replace("\\", "\\\\").replace('"', '\\"') if isinstance(x, str) else repr(x))
            out.append("%s%s = %s;" % (pad, k, "[" + ", ".join(one(x) for x in v) + "]" if isinstance(v, list) else one(v)))
    return "\n".join(out) + "\n"


def tracker_text(width, height, fx, fy, cx, cy):
    """A COMPLETE configuration file of ROFT-tracker (every key src/roft/src/main.cpp:43-147 reads, laid out like
    config/config_fast_ycb.cfg): the filter defaults of default_text() plus the data-set, log and model sections with the
    placeholders test/test.sh overrides on the command line."""
    cfg = parse_cfg(default_text(width, height, fx, fy, cx, cy))
    cfg["camera_dataset"].update(path="?", data_prefix="/", rgb_prefix="rgb/", depth_prefix="depth/", data_format="txt", rgb_format="png",
                                 depth_format="float", heading_zeros=0, index_offset=0)
    cfg["log"] = dict(enable=True, enable_segmentation=False, path="?")
    cfg["model"] = dict(name="?", use_internal_db=True, internal_db_name="DOPE", external_path="?")
    cfg["optical_flow_dataset"] = dict(path="?", set="nvof", heading_zeros=0, index_offset=0)
    cfg["pose_dataset"] = dict(dict(path="?", skip_rows=0, skip_cols=0), **cfg["pose_dataset"])
    cfg["segmentation_dataset"] = dict(dict(path="?", format="png", set="mrcnn", heading_zeros=0, index_offset=0), **cfg["segmentation_dataset"])
    return dump_cfg(dict(sorted(cfg.items(), key=lambda kv: (isinstance(kv[1], dict), kv[0]))))


def all_keys(cfg, prefix=""):
    for k, v in cfg.items():
        if isinstance(v, dict):
            yield from all_keys(v, prefix + k + ".")
        else:
            yield prefix + k


__all__ = ["parse_cfg", "apply_overrides", "lookup", "to_engine", "load", "frames_between", "default_text", "tracker_text", "dump_cfg", "FILTER_KEYS", "DATASET_KEYS",
           "all_keys"]
