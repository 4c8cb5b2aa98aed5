// Drives the whole-filter facade ROFT::ROFTFilter (include/ROFT/Filters.h) the way ROFT-tracker drives
// ROFT::ROFTFilter: one filtering_step() per frame with that frame's depth / flow / mask / pose, then the estimate.
// Input: a small binary stream written by tests/test_facade.py; output: pose (13) + twist (6) per frame.
#include <cstdio>
#include <cstdint>
#include <vector>

#include "ROFT/Filters.h"

template <class T>
static bool rd(FILE* f, T* p, size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[5];   // W, H, n_frames, n_verts, n_tris
    double cam[4], init[13];
    if (!rd(f, hdr, 5) || !rd(f, cam, 4) || !rd(f, init, 13)) return 2;
    const int W = hdr[0], H = hdr[1], n = hdr[2], nv = hdr[3], nt = hdr[4];
    std::vector<float> verts(3 * (size_t)nv);
    std::vector<int32_t> tris(3 * (size_t)nt);
    if (!rd(f, verts.data(), verts.size()) || !rd(f, tris.data(), tris.size())) return 2;
    try {
        roft_config cfg;
        if (roft_default_config(&cfg, W, H, ROFT_FLOW_F32C2) != ROFT_OK) throw std::runtime_error(roft_last_error_string());
        cfg.cam.fx = cam[0]; cfg.cam.fy = cam[1]; cfg.cam.cx = cam[2]; cfg.cam.cy = cam[3];
        cfg.max_objects = 1;
        ROFT::ROFTFilter filter(cfg);
        roft_object_desc obj;
        roft_default_object(&obj);
        for (int i = 0; i < 13; ++i) obj.p_mean0[i] = init[i];
        obj.mesh.verts = verts.data(); obj.mesh.n_verts = nv;
        obj.mesh.tris = tris.data(); obj.mesh.n_tris = nt;
        const int id = filter.add_object(obj);
        FILE* o = std::fopen(argv[2], "wb");
        std::vector<float> depth((size_t)W * H), flow(2 * (size_t)W * H);
        std::vector<uint8_t> mask((size_t)W * H);
        for (int k = 0; k < n; ++k) {
            double dt, pose[7];
            int32_t flags[3];   // has_flow, has_mask, has_pose
            if (!rd(f, &dt, 1) || !rd(f, flags, 3) || !rd(f, depth.data(), depth.size())) return 2;
            if (flags[0] && !rd(f, flow.data(), flow.size())) return 2;
            if (flags[1] && !rd(f, mask.data(), mask.size())) return 2;
            if (flags[2] && !rd(f, pose, 7)) return 2;
            roft_frame_input in{};
            in.dt = dt;
            in.depth = depth.data();
            in.flow = flags[0] ? flow.data() : nullptr;
            in.mask = flags[1] ? mask.data() : nullptr;
            in.pose_valid = flags[2];
            for (int i = 0; i < 3; ++i) in.pose_x[i] = pose[i];
            for (int i = 0; i < 4; ++i) in.pose_q[i] = pose[3 + i];
            in.mem_kind = ROFT_MEM_HOST;
            filter.filtering_step({in});
            double p13[13], tw[6];
            filter.state(id, p13, tw);
            std::fwrite(p13, 8, 13, o);
            std::fwrite(tw, 8, 6, o);
        }
        std::fclose(o);
    } catch (const std::runtime_error& e) {
        std::printf("runtime_error: %s\n", e.what());
        return 3;
    }
    std::fclose(f);
    return 0;
}
