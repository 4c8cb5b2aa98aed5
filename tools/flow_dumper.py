#!/usr/bin/env python3
"""Optical-flow dumper: the MI355X counterpart of the reference's `ROFT-of-dumper`
(tools/nvof/dumper/src/main.cpp:40-146) with the same synopsis and the same output files.

  flow_dumper.py <dataset_path> <data_format> <rgb_format> <heading_zeros> <index_offset> <camera_width>
                 <camera_height> <nvof_version> <output_path>

Frames `%0<heading_zeros>d.<rgb_format>` are read from <dataset_path>/rgb/ starting at index <index_offset>; their
number is the number of rows of <dataset_path>/data.<data_format> (RobotsIO::Camera log).  For every frame but the
first, the forward flow from the previous frame is written to <output_path>/<index>.float in the `.float` layout of
OpticalFlowUtilities.cpp:77-136:  nvof1 -> CV_16SC2 (S10.5) at grid 4, nvof2 -> CV_32FC2 at grid 1
(ImageOpticalFlowNVOF.cpp:19-80).  The flow itself comes from the HIP pyramidal Lucas-Kanade producer
(roft_flow_producer_*); there is no CPU path.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LOG_NAME = "ROFT-of-dumper"
BATCH = 16


def synopsis():
    sys.stderr.write("Synopsis: " + LOG_NAME + " <dataset_path> <data_format> <rgb_format> <heading_zeros> <index_offset> "
                     "<camera_width> <camera_height> <nvof_version> <output_path>\n\n"
                     "  <nvof_version> to be chosen among \"nvof1\" and \"nvof2\".\n")
    return 1


def main(argv):
    if len(argv) != 10:
        return synopsis()
    dataset_path, data_format, rgb_format = argv[1], argv[2], argv[3]
    try:
        names = ("heading_zeros", "index_offset", "camera_width", "camera_height")
        vals = []
        for i, name in zip(range(4, 8), names):
            if int(argv[i]) < 0:
                raise ValueError
            vals.append(int(argv[i]))
        heading_zeros, index_offset, width, height = vals
    except ValueError:
        sys.stderr.write("Invalid value %s for parameter <%s>.\n" % (argv[i], name))
        return 1
    nvof = argv[8]
    if nvof not in ("nvof1", "nvof2"):
        sys.stderr.write("Invalid <nvof_version> \"%s\"\n" % nvof)
        return 1
    output_path = argv[9]
    if rgb_format != "png":
        sys.stderr.write("only png frames can be decoded here\n")
        return 1

    import numpy as np
    import torch

    from roft_amd import _lib as L
    from roft_amd import io, ops

    L.require_device()
    rgb_dir = os.path.join(dataset_path, "rgb")
    print("Running with:\n    Rgb frames in: %s/%%%dd.%s\n    Rgb frames resolution: %d x %d\n    Output path: %s"
          % (rgb_dir, heading_zeros, rgb_format, width, height, output_path))
    n_frames = len(io.read_data_txt(os.path.join(dataset_path, "data." + data_format))[0])
    os.makedirs(output_path, exist_ok=True)
    ft = L.FLOW_S16C2 if nvof == "nvof1" else L.FLOW_F32C2

    def name(i, ext):
        return ("%0" + str(heading_zeros) + "d.%s") % (i, ext)

    def load(i):
        g = io.rgb_to_gray(io.read_png(os.path.join(rgb_dir, name(i, rgb_format))))
        if g.shape != (height, width):
            raise ValueError("frame %d is %s, expected %d x %d" % (i, g.shape[::-1], width, height))
        return torch.from_numpy(np.ascontiguousarray(g)).cuda()

    fp = ops.FlowProducer(width, height, BATCH, ft)
    oshape = (height, width, 2) if ft == L.FLOW_F32C2 else (height // 4, width // 4, 2)
    out = torch.zeros((BATCH,) + oshape, dtype=torch.float32 if ft == L.FLOW_F32C2 else torch.int16, device="cuda")
    last = load(index_offset) if n_frames > 0 else None
    k = 1
    while k < n_frames:
        m = min(BATCH, n_frames - k)
        frames = [last] + [load(index_offset + k + j) for j in range(m)]
        torch.cuda.synchronize()
        fp.run([frames[j].data_ptr() for j in range(m)], [frames[j + 1].data_ptr() for j in range(m)],
               [out[j].data_ptr() for j in range(m)])
        fp.sync()
        host = out[:m].cpu().numpy()
        for j in range(m):
            io.save_flow(host[j], os.path.join(output_path, name(index_offset + k + j, "float")))
        last = frames[-1]
        k += m
    fp.close()
    print("\nProcessing completed.")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
