// Drives ROFT::ROFTFilter (include/ROFT/ROFTFilter.h) the way ROFT-tracker drives the reference's class: construct it with
// the reference's constructor signature (sources + parameter vectors, src/roft-lib/include/ROFT/ROFTFilter.h:42-73), boot(),
// run().  Input: a recorded stream written by tests/test_facade.py; output: pose (13) + twist (6) per frame.
#include <cstdio>
#include <fstream>

#include "mem_sources.h"

// a filter that records its estimate after every step (the reference logs the same quantities, ROFTFilter.cpp:386-394)
class RecordingFilter : public ROFT::ROFTFilter {
public:
    using ROFT::ROFTFilter::ROFTFilter;
    std::vector<double> rows;

protected:
    void filtering_step() override
    {
        const long before = frames();
        ROFT::ROFTFilter::filtering_step();
        if (frames() == before) return;
        for (int i = 0; i < 13; ++i) rows.push_back(pose_belief().mean()(i));
        for (int i = 0; i < 6; ++i) rows.push_back(velocity_belief().mean()(i));
    }
};

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    RecordedStream s;
    if (!s.load(argv[1])) return 2;
    try {
        // the mesh goes through a Wavefront OBJ file, as the reference's ModelParameters name one
        const std::string obj_path = std::string(argv[2]) + ".obj";
        {
            std::ofstream o(obj_path);
            o.precision(9);
            for (std::size_t i = 0; i < s.verts.size() / 3; ++i) o << "v " << s.verts[3 * i] << " " << s.verts[3 * i + 1] << " " << s.verts[3 * i + 2] << "\n";
            for (std::size_t i = 0; i < s.tris.size() / 3; ++i)
                o << "f " << s.tris[3 * i] + 1 << "//" << s.tris[3 * i] + 1 << " " << s.tris[3 * i + 1] + 1 << "//" << s.tris[3 * i + 1] + 1 << " "
                  << s.tris[3 * i + 2] + 1 << "//" << s.tris[3 * i + 2] + 1 << "\n";
        }
        ROFT::ModelParameters model;
        model.name("object");
        model.mesh_external_path(obj_path);
        auto camera = std::make_shared<ROFT::CameraMeasurement>(std::make_shared<MemCamera>(s));
        // argv[3]: "pooled" / "rewriter" serve the recording from pinned pool buffers (the filter then hands them over in place),
        // the latter through ONE flow matrix rewritten every frame; default: caller memory (staged by the submit call)
        const std::string mode = argc > 3 ? argv[3] : "";
        std::shared_ptr<RobotsIO::Utils::Segmentation> segmentation;
        std::shared_ptr<ROFT::ImageOpticalFlowSource> flow;
        if (mode == "pooled" || mode == "rewriter") {
            segmentation = std::make_shared<MemSegmentationPooled>(s, 6);
            flow = std::make_shared<MemFlowPooled>(s, mode == "rewriter");
        } else {
            segmentation = std::make_shared<MemSegmentation>(s, 6);
            flow = std::make_shared<MemFlow>(s);
        }
        auto pose = std::make_shared<MemPose>(s, 6);
        // parameter vectors as src/roft/src/main.cpp:286-325 packs the keys of config_fast_ycb.cfg
        Eigen::VectorXd p0(13), p_cov0(12), p_model(6), p_meas(12), v0(6), v_cov0(6), v_model(6), v_meas(2);
        for (int i = 0; i < 13; ++i) p0(i) = s.init[i];
        for (int i = 0; i < 12; ++i) p_cov0(i) = 1e-3;
        for (int i = 0; i < 6; ++i) { p_model(i) = 1.0; v_cov0(i) = 1e-3; v_model(i) = 0.1; }
        for (int i = 0; i < 3; ++i) { p_meas(i) = 0.1; p_meas(3 + i) = 1e-4; p_meas(6 + i) = 1e-3; p_meas(9 + i) = 1e-4; }
        v_meas(0) = v_meas(1) = 1.0;
        RecordingFilter filter(camera, segmentation, flow, pose, model, p0, p_cov0, p_model, p_meas, v0, v_cov0, v_model, v_meas,
                               /* ut */ 1.0, 2.0, 0.0, /* sample_time */ s.frames[0].dt, /* pose_meas */ true, /* pose_resync */ true,
                               /* outlier rejection, gain */ true, true, /* velocity_meas */ true, /* flow_weighting */ true,
                               /* flow_aided_segmentation */ true, /* maximum_depth */ 2.0, /* subsampling_radius */ 35.0,
                               /* enable_log */ false, "", "");
        filter.boot();
        filter.run();     // until the camera runs out of frames
        filter.wait();
        FILE* o = std::fopen(argv[2], "wb");
        std::fwrite(filter.rows.data(), 8, filter.rows.size(), o);
        std::fclose(o);
    } catch (const std::runtime_error& e) {
        std::printf("runtime_error: %s\n", e.what());
        return 3;
    }
    return 0;
}
