// The last stage of ROFT-tracker's main() (src/roft/src/main.cpp:393-424) against the facade: the filter is built with the
// reference's constructor call, the two segmentation probes are attached with set_probe(), the log is switched on with
// enable_log(log_path, "") and the filter is driven with boot() / run() / wait() -- the same statements, over in-memory
// sources instead of the dataset readers.  usage: tracker_tail_check <stream.bin> <log dir>
// Leaves in <log dir>: pose_estimate.txt, velocity_estimate.txt, execution_times.txt (ROFTFilter::log_file_names,
// ROFTFilter.cpp:247-252), pose_measurements.txt, velocity_measurements.txt (CartesianQuaternionMeasurement.cpp:535-539),
// segmentation/<i>.png and segmentation_refined/<i>.png.
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "mem_sources.h"

using namespace ROFT;
using namespace RobotsIO::Utils;

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    RecordedStream s;
    if (!s.load(argv[1])) return 2;
    const std::string log_path = argv[2];
    const bool enable_log = true, enable_log_segmentation = true;
    try {
        const std::string obj_path = log_path + "/mesh.obj";
        {
            std::ofstream o(obj_path);
            o.precision(9);
            for (std::size_t i = 0; i < s.verts.size() / 3; ++i) o << "v " << s.verts[3 * i] << " " << s.verts[3 * i + 1] << " " << s.verts[3 * i + 2] << "\n";
            for (std::size_t i = 0; i < s.tris.size() / 3; ++i) o << "f " << s.tris[3 * i] + 1 << " " << s.tris[3 * i + 1] + 1 << " " << s.tris[3 * i + 2] + 1 << "\n";
        }
        ModelParameters model_parameters;
        model_parameters.name("object");
        model_parameters.mesh_external_path(obj_path);
        auto camera = std::make_shared<CameraMeasurement>(std::make_shared<MemCamera>(s));
        std::shared_ptr<Segmentation> segmentation = std::make_shared<MemSegmentation>(s, 6);
        std::shared_ptr<ImageOpticalFlowSource> flow = std::make_shared<MemFlow>(s);
        std::shared_ptr<Transform> pose = std::make_shared<MemPose>(s, 6);
        Eigen::VectorXd p_initial_condition(13), p_initial_covariance(12), p_model_covariance(6), p_measurement_covariance(12);
        Eigen::VectorXd v_initial_condition(6), v_initial_covariance(6), v_model_covariance(6), v_measurement_covariance(2);
        for (int i = 0; i < 13; ++i) p_initial_condition(i) = s.init[i];
        for (int i = 0; i < 12; ++i) p_initial_covariance(i) = 1e-3;
        for (int i = 0; i < 6; ++i) { p_model_covariance(i) = 1.0; v_initial_covariance(i) = 1e-3; v_model_covariance(i) = 0.1; }
        for (int i = 0; i < 3; ++i) { p_measurement_covariance(i) = 0.1; p_measurement_covariance(3 + i) = 1e-4; p_measurement_covariance(6 + i) = 1e-3; p_measurement_covariance(9 + i) = 1e-4; }
        v_measurement_covariance(0) = v_measurement_covariance(1) = 1.0;
        const double ut_alpha = 1.0, ut_beta = 2.0, ut_kappa = 0.0, sample_time = s.frames[0].dt, depth_maximum = 2.0, subsampling_radius = 35.0;
        const bool use_pose_measurement = true, use_pose_resync = true, outlier_rejection_enable = true, use_velocity_measurement = true,
                   flow_weighting = true, flow_aided_segmentation = true;
        const double outlier_rejection_gain = 0.01;   // (narrowed to bool by the constructor, ROFTFilter.h:64)

        // ---- from here on: main.cpp:393-424
        std::unique_ptr<ROFTFilter> filter = std::make_unique<ROFTFilter>(
            camera, segmentation, flow, pose, model_parameters, p_initial_condition, p_initial_covariance, p_model_covariance,
            p_measurement_covariance, v_initial_condition, v_initial_covariance, v_model_covariance, v_measurement_covariance, ut_alpha,
            ut_beta, ut_kappa, sample_time, use_pose_measurement, use_pose_resync, outlier_rejection_enable, outlier_rejection_gain,
            use_velocity_measurement, flow_weighting, flow_aided_segmentation, depth_maximum, subsampling_radius, enable_log, log_path, "");
        if (enable_log_segmentation) {
            std::unique_ptr<Probe> probe_0 = std::unique_ptr<ImageFileProbe>(new ImageFileProbe(log_path + "/segmentation/", "", "png"));
            filter->set_probe("output_segmentation", std::move(probe_0));
            std::unique_ptr<Probe> probe_1 = std::unique_ptr<ImageFileProbe>(new ImageFileProbe(log_path + "/segmentation_refined/", "", "png"));
            filter->set_probe("output_segmentation_refined", std::move(probe_1));
        }
        if (enable_log) filter->enable_log(log_path, "");
        filter->boot();
        filter->run();
        if (!filter->wait()) return EXIT_FAILURE;
    } catch (const std::runtime_error& e) {
        std::printf("runtime_error: %s\n", e.what());
        return 3;
    }
    return EXIT_SUCCESS;
}
