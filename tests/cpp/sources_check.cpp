// CPU checks of the file-backed sources and of the configuration reader the reference's executable is built on
// (include/compat/ConfigParser.h, include/ROFT/CompatIO.h, DatasetImage*.h).  No device call.
//   sources_check cfg <out.txt> <argv of ROFT-tracker...>     every setting main.cpp reads, one `key=value` per line
//   sources_check png <file.png> <out.bin>                    rows cols channels (3 ints) + the decoded bytes
//   sources_check gray <file.png> <out.bin>                   the same through bgr_to_gray
//   sources_check sched <root> <object> <set> <poses.txt> <frames> <w> <h> <fps> <simulated_fps> <delay 0|1>
//                                                             per frame: mask delivered (value of pixel 0, -1 = none), pose x (nan = none)
//   sources_check queue                                       OpticalFlowQueueHandler: window, region after a stamp, unknown stamp
//   sources_check mesh <name> <set> <external path>           MeshResource: sizes of the internal-data-base and external texts
//   sources_check nvof <root> <w> <h> <1|2> <out.bin>          (GPU) ImageOpticalFlowNVOF(camera, ...) over the camera images: per frame a flag + the flow
//   sources_check obj <file.obj>                              load_obj_mesh: vertices, triangles, index checksum, bounding box
//   sources_check camera <root> <w> <h>                       per frame: index stamp_rgb stamp_depth depth(0,0) depth(h-1,w-1) pose x qw
#include <cmath>
#include <cstdio>
#include <cstring>

#include <ConfigParser.h>
#include <ROFT/DatasetImageOpticalFlow.h>
#include <ROFT/DatasetImageSegmentationDelayed.h>
#include <ROFT/MeshResource.h>
#include <ROFT/OpticalFlowQueueHandler.h>
#include <ROFT/ROFTFilter.h>

static std::string str(const Eigen::VectorXd& v)
{
    std::ostringstream s;
    s.precision(17);
    for (std::size_t i = 0; i < v.size(); ++i) s << (i ? "," : "") << v(i);
    return s.str();
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const std::string mode = argv[1];
    try {
        if (mode == "cfg") {
            std::FILE* out = std::fopen(argv[2], "w");
            ConfigParser conf(argc - 3, argv + 3);   // argv[3] plays the program name
            const char* doubles[] = {"sample_time", "camera_dataset.fx", "camera_dataset.fy", "camera_dataset.cx", "camera_dataset.cy",
                                     "measurement_model.velocity.depth_maximum", "measurement_model.velocity.subsampling_radius", "outlier_rejection.gain",
                                     "pose_dataset.original_fps", "pose_dataset.desired_fps", "segmentation_dataset.original_fps",
                                     "segmentation_dataset.desired_fps", "unscented_transform.alpha", "unscented_transform.beta", "unscented_transform.kappa"};
            const char* ints[] = {"camera_dataset.width", "camera_dataset.height", "camera_dataset.heading_zeros", "camera_dataset.index_offset",
                                  "optical_flow_dataset.heading_zeros", "optical_flow_dataset.index_offset", "pose_dataset.skip_rows", "pose_dataset.skip_cols",
                                  "segmentation_dataset.heading_zeros", "segmentation_dataset.index_offset"};
            const char* bools[] = {"log.enable", "log.enable_segmentation", "measurement_model.velocity.weight_flow", "measurement_model.use_pose",
                                   "measurement_model.use_pose_resync", "measurement_model.use_velocity", "model.use_internal_db", "outlier_rejection.enable",
                                   "pose_dataset.fps_reduction", "pose_dataset.delay", "segmentation_dataset.fps_reduction", "segmentation_dataset.delay",
                                   "segmentation_dataset.flow_aided"};
            const char* strings[] = {"camera_dataset.path", "camera_dataset.data_prefix", "camera_dataset.rgb_prefix", "camera_dataset.depth_prefix",
                                     "camera_dataset.data_format", "camera_dataset.rgb_format", "camera_dataset.depth_format", "log.path", "model.name",
                                     "model.internal_db_name", "model.external_path", "optical_flow_dataset.path", "optical_flow_dataset.set", "pose_dataset.path",
                                     "segmentation_dataset.path", "segmentation_dataset.format", "segmentation_dataset.set"};
            const char* vectors[] = {"initial_condition.pose.v", "initial_condition.pose.w", "initial_condition.pose.x", "initial_condition.pose.axis_angle",
                                     "initial_condition.pose.cov_v", "initial_condition.pose.cov_w", "initial_condition.pose.cov_x", "initial_condition.pose.cov_q",
                                     "initial_condition.velocity.v", "initial_condition.velocity.w", "initial_condition.velocity.cov_v",
                                     "initial_condition.velocity.cov_w", "kinematic_model.pose.sigma_linear", "kinematic_model.pose.sigma_angular",
                                     "kinematic_model.velocity.sigma_linear", "kinematic_model.velocity.sigma_angular", "measurement_model.pose.cov_v",
                                     "measurement_model.pose.cov_w", "measurement_model.pose.cov_x", "measurement_model.pose.cov_q",
                                     "measurement_model.velocity.cov_flow"};
            for (const char* k : doubles) { double v = NAN; conf(k, v); std::fprintf(out, "%s=%.17g\n", k, v); }
            for (const char* k : ints) { int v = -12345; conf(k, v); std::fprintf(out, "%s=%d\n", k, v); }
            for (const char* k : bools) { bool v = false; conf(k, v); std::fprintf(out, "%s=%s\n", k, v ? "true" : "false"); }
            for (const char* k : strings) { std::string v = "<unset>"; conf(k, v); std::fprintf(out, "%s=%s\n", k, v.c_str()); }
            for (const char* k : vectors) { Eigen::VectorXd v; conf(k, v); std::fprintf(out, "%s=%s\n", k, str(v).c_str()); }
            std::fclose(out);
            return 0;
        }
        if (mode == "png" || mode == "gray") {
            cv::Mat m = ROFT::compat::read_png(argv[2]);
            if (mode == "gray") m = ROFT::compat::bgr_to_gray(m);
            std::FILE* out = std::fopen(argv[3], "wb");
            const int hdr[3] = {m.rows, m.cols, m.empty() ? 0 : (int)m.elemSize()};
            std::fwrite(hdr, sizeof(hdr), 1, out);
            if (!m.empty()) std::fwrite(m.data, 1, m.total() * m.elemSize(), out);
            std::fclose(out);
            return 0;
        }
        if (mode == "sched") {
            const std::string root = argv[2], object = argv[3], set = argv[4], poses = argv[5];
            const int frames = std::atoi(argv[6]), w = std::atoi(argv[7]), h = std::atoi(argv[8]);
            const double fps = std::atof(argv[9]), sim = std::atof(argv[10]);
            const bool delay = std::atoi(argv[11]) != 0;
            const std::size_t offset = argc > 12 ? (std::size_t)std::atoi(argv[12]) : 0;   // where the tracker starts (test_ho3d.sh:142-160)
            ROFT::ModelParameters mp;
            mp.name(object);
            ROFT::DatasetImageSegmentationDelayed seg((float)fps, (float)sim, delay, root, "png", w, h, set, mp, 0, offset);
            RobotsIO::Utils::DatasetTransformDelayed tr(fps, sim, delay, poses, offset, 0, 7);
            ROFT::DatasetImageSegmentation plain(root, "png", w, h, set, mp, 0, offset);
            RobotsIO::Utils::DatasetTransform plain_tr(poses, offset, 0, 7);
            std::printf("between %d %d %d %d\n", seg.get_frames_between_iterations(), tr.get_frames_between_iterations(),
                        plain.get_frames_between_iterations(), plain_tr.get_frames_between_iterations());
            for (int k = (int)offset; k < frames; ++k) {
                seg.step_frame();
                plain.step_frame();
                const auto m = seg.segmentation(false);
                const auto pm = plain.segmentation(false);
                const bool got = tr.freeze(false), pgot = plain_tr.freeze(false);
                std::printf("%d %d %.17g %d %.17g\n", k, m.first ? (int)m.second.data[0] : -1, got ? tr.transform().translation()[0] : NAN,
                            pm.first ? (int)pm.second.data[0] : -1, pgot ? plain_tr.transform().translation()[0] : NAN);
            }
            return 0;
        }
        if (mode == "queue") {
            ROFT::OpticalFlowQueueHandler q(4);
            for (int k = 0; k < 7; ++k) {
                cv::Mat m(1, 1, CV_32FC2);
                m.at<float>(0, 0) = (float)k;
                q.add_flow(m, k / 30.0);      // entries 3, 4, 5, 6 survive
            }
            auto show = [&](double stamp) {
                std::printf("%.6f:", stamp);
                for (const cv::Mat& m : q.get_buffer_region(stamp)) std::printf(" %d", (int)m.at<float>(0, 0));
                std::printf("\n");
            };
            show(4 / 30.0); show(4 / 30.0 + 5e-4); show(4 / 30.0 + 2e-3); show(2 / 30.0); show(6 / 30.0); show(3 / 30.0);
            q.clear();
            show(4 / 30.0);
            return 0;
        }
        if (mode == "mesh") {
            ROFT::ModelParameters mp;
            mp.name(argv[2]);
            mp.internal_db_name(argv[3]);
            mp.mesh_external_path(argv[4]);
            mp.use_internal_db(false);
            std::printf("external %zu\n", ROFT::MeshResource(mp).as_string().size());
            mp.use_internal_db(true);
            std::printf("internal %zu\n", ROFT::MeshResource(mp).as_string().size());
            std::printf("named %zu\n", ROFT::MeshResource(argv[2], argv[3]).as_string().size());
            return 0;
        }
        if (mode == "nvof") {
            const int w = std::atoi(argv[3]), h = std::atoi(argv[4]), product = std::atoi(argv[5]);
            auto cam = std::make_shared<ROFT::CameraMeasurement>(std::make_unique<RobotsIO::Camera::DatasetCamera>(
                argv[2], "/", "rgb/", "depth/", "txt", "png", "float", 0, 0, w, h, 1.0, 2.0, 3.0, 4.0));
            std::shared_ptr<ROFT::ImageOpticalFlowSource> flow;
            if (product == 1) flow = std::make_shared<ROFT::ImageOpticalFlowNVOF>(cam, ROFT::ImageOpticalFlowNVOF::NVOFPerformance_1_0::Slow, false);
            else flow = std::make_shared<ROFT::ImageOpticalFlowNVOF>(cam, ROFT::ImageOpticalFlowNVOF::NVOFPerformance_2_0::Slow);
            std::FILE* out = std::fopen(argv[6], "wb");
            while (cam->freeze(ROFT::CameraMeasurementType::RGBD)) {
                flow->step_frame();
                bool valid = false;
                cv::Mat f;
                std::tie(valid, f) = flow->flow(false);
                const int hdr[4] = {(int)valid, f.rows, f.cols, f.type()};
                std::fwrite(hdr, sizeof(hdr), 1, out);
                if (valid) std::fwrite(f.data, 1, f.total() * f.elemSize(), out);
            }
            std::fclose(out);
            return 0;
        }
        if (mode == "obj") {
            std::vector<float> v;
            std::vector<std::int32_t> t;
            ROFT::load_obj_mesh(argv[2], v, t);
            long long sum = 0;
            for (std::size_t i = 0; i < t.size(); ++i) sum += (long long)t[i] * (long long)(i % 7 + 1);
            float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
            for (std::size_t i = 0; i < v.size(); ++i) { lo[i % 3] = std::min(lo[i % 3], v[i]); hi[i % 3] = std::max(hi[i % 3], v[i]); }
            std::printf("%zu %zu %lld %.9g %.9g %.9g %.9g %.9g %.9g\n", v.size() / 3, t.size() / 3, sum, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]);
            return 0;
        }
        if (mode == "camera") {
            const int w = std::atoi(argv[3]), h = std::atoi(argv[4]);
            RobotsIO::Camera::DatasetCamera cam(argv[2], "/", "rgb/", "depth/", "txt", "png", "float", 0, argc > 5 ? (std::size_t)std::atoi(argv[5]) : 0, w, h, 1.0, 2.0, 3.0, 4.0);
            while (cam.step_frame()) {
                const auto d = cam.depth(true);
                const auto p = cam.pose(true);
                const auto rgb = cam.rgb(true);
                std::printf("%d %.17g %.17g %d %.9g %.9g %.17g %.17g %d\n", (int)cam.frame_index(), cam.time_stamp_rgb().second, cam.time_stamp_depth().second,
                            (int)d.first, d.first ? d.second(0, 0) : NAN, d.first ? d.second(h - 1, w - 1) : NAN, p.second.translation()[0],
                            p.second.quaternion()[0], rgb.first ? (int)rgb.second.data[0] : -1);
            }
            return 0;
        }
    } catch (const std::exception& e) {
        std::printf("runtime_error %s\n", e.what());
        return 3;
    }
    return 2;
}
