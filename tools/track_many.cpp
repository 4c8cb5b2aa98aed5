// ROFT-tracker-batch -- what ROFT-tracker (src/roft/src/main.cpp of the reference) does for one object, for MANY objects at
// once on one MI355X: one engine, every object its own sequence directory, all of them advanced by the same launches.
// The reference runs one tracker process per object (main.cpp:421-424); objects do not interact anywhere in
// ROFTFilter::filtering_step, so this is the same computation per object (tests/test_facade.py compares the two).
//
//   ROFT-tracker-batch --from config.cfg [--group::key value ...]        the reference's configuration file and overrides
//                      --log_root DIR [--batch_frames T] [--outlier_bands B] [--device D] [--shard RANK WORLD]
//                      --object SEQUENCE_DIR NAME [MESH.obj] [--object ...]
//
// Several GPUs: one process per GPU, each given the same object list, `--device r --shard r G`: process r tracks the r-th block
// of ceil(n / G) objects (the partition of roft_amd/parallel.py) -- no process talks to another while tracking.
// `--gather ID_FILE` (built with -DROFT_WITH_RCCL): the one exchange a job needs, natively -- at the end every process hands its
// per-object result rows (19 doubles per object-frame: pose 13 | twist 6, what ROFTFilter logs) to an ncclAllGather over
// RCCL / xGMI and process 0 writes the logs of ALL objects; the ncclUniqueId travels through ID_FILE (process 0 writes it, the
// others wait for it: no MPI, no launcher beyond starting the G processes).
//
// Per object: camera / flow / mask / pose sources exactly as main.cpp:327-381 builds them from the configuration, rooted at
// SEQUENCE_DIR (`pose_dataset.path` is taken relative to it), the initial pose = the first row of its pose file
// (test/test.sh:120-123), the mesh = MESH.obj or MeshResource's data base entry NAME.  Output: DIR/NAME/pose_estimate.txt and
// velocity_estimate.txt in the format of the reference's logs.
//
//   g++ -std=c++17 -O2 -I include/compat -I include tools/track_many.cpp -L roft_amd/csrc -lroft_hip -o ROFT-tracker-batch
//   ... -DROFT_WITH_RCCL -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include ... -L/opt/rocm/lib -lrccl -lamdhip64     (with --gather)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <thread>

#ifdef ROFT_WITH_RCCL
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#endif

#include <ConfigParser.h>
#include <ROFT/Filters.h>

using namespace ROFT;

namespace {

struct ObjectArgs {
    std::string root, name, mesh;
};

struct EstimateLog : bfl::Logger {
    std::vector<std::string> log_file_names(const std::string& p, const std::string& n) override
    {
        return {p + "/" + n + "pose_estimate", p + "/" + n + "velocity_estimate"};
    }
};

struct TrackedObject {
    ObjectArgs args;
    std::shared_ptr<CameraMeasurement> camera;
    std::shared_ptr<RobotsIO::Utils::Segmentation> segmentation;
    std::shared_ptr<ImageOpticalFlowSource> flow;
    std::shared_ptr<RobotsIO::Utils::Transform> pose;
    std::vector<float> verts;
    std::vector<std::int32_t> tris;
    double last_stamp = -1.0;
    bool mask_received = false;
};

// what one frame of one object hands to the engine, kept alive until the batch has been submitted
struct FrameBuffers {
    Eigen::MatrixXf depth;
    cv::Mat flow, mask;
};

#ifdef ROFT_WITH_RCCL
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define NCCL_OK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) throw std::runtime_error(std::string(#x) + ": " + ncclGetErrorString(r_)); } while (0)

// The result rows of every process -> every process (ncclAllGather over RCCL): shard sizes first (the block partition may be
// uneven and sequences may differ in length), then the rows in blocks padded to the largest shard.  Returns, per rank, its
// (objects, frames, rows[frames][objects][19]).
struct ShardRows { long n_obj = 0, frames = 0; std::vector<double> rows; };
std::vector<ShardRows> gather_rows(const std::string& id_file, int rank, int world, int device, long n_obj, long frames, const std::vector<double>& rows)
{
    HIP_OK(hipSetDevice(device));
    // The id travels through ID_FILE as (job nonce, ncclUniqueId).  A file left behind by a job that crashed must never be taken for
    // this job's: process 0 removes whatever is there before it creates its id, and the others accept a file only if it carries
    // their job's nonce (ROFT_JOB_NONCE, the same non-zero number for every process of a job: a launcher's pid or start time) or,
    // without a nonce, if it was written no earlier than ten minutes before they started (a leftover of an earlier job is older
    // or, if the job before died just now, is removed by this job's process 0 within its first milliseconds).
    ncclUniqueId id;
    unsigned long long nonce = 0;
    if (const char* jn = std::getenv("ROFT_JOB_NONCE")) nonce = std::strtoull(jn, nullptr, 10);
    if (rank == 0) {
        std::error_code ec;
        std::filesystem::remove(id_file, ec);
        NCCL_OK(ncclGetUniqueId(&id));
        const std::string tmp = id_file + ".tmp";
        std::FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(&nonce, sizeof(nonce), 1, f) != 1 || std::fwrite(&id, sizeof(id), 1, f) != 1) throw std::runtime_error("cannot write " + tmp);
        std::fclose(f);
        std::filesystem::rename(tmp, id_file);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        const auto oldest = std::filesystem::file_time_type::clock::now() - std::chrono::minutes(10);
        for (;;) {
            std::error_code ec;
            const auto written = std::filesystem::last_write_time(id_file, ec);
            if (!ec && (nonce != 0 || written >= oldest)) {
                std::FILE* f = std::fopen(id_file.c_str(), "rb");
                if (f) {
                    unsigned long long theirs = 0;
                    const bool ok = std::fread(&theirs, sizeof(theirs), 1, f) == 1 && std::fread(&id, sizeof(id), 1, f) == 1;
                    std::fclose(f);
                    if (ok && theirs == nonce) break;
                }
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) throw std::runtime_error("no ncclUniqueId of this job in " + id_file + " after 120 s");
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    ncclComm_t comm;
    NCCL_OK(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    // shard sizes
    long long mine[2] = {n_obj, frames};
    long long *d_mine = nullptr, *d_all = nullptr;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_mine), sizeof(mine)));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_all), sizeof(mine) * world));
    HIP_OK(hipMemcpy(d_mine, mine, sizeof(mine), hipMemcpyHostToDevice));
    NCCL_OK(ncclAllGather(d_mine, d_all, 2, ncclInt64, comm, stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<long long> all(2 * (size_t)world);
    HIP_OK(hipMemcpy(all.data(), d_all, sizeof(mine) * world, hipMemcpyDeviceToHost));
    size_t pad = 1;
    for (int r = 0; r < world; ++r) pad = std::max(pad, (size_t)(all[2 * r] * all[2 * r + 1] * 19));
    // rows
    double *d_send = nullptr, *d_recv = nullptr;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_send), sizeof(double) * pad));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_recv), sizeof(double) * pad * world));
    HIP_OK(hipMemset(d_send, 0, sizeof(double) * pad));
    if (!rows.empty()) HIP_OK(hipMemcpy(d_send, rows.data(), sizeof(double) * rows.size(), hipMemcpyHostToDevice));
    NCCL_OK(ncclAllGather(d_send, d_recv, pad, ncclDouble, comm, stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<ShardRows> out((size_t)world);
    for (int r = 0; r < world; ++r) {
        out[r].n_obj = (long)all[2 * r];
        out[r].frames = (long)all[2 * r + 1];
        out[r].rows.resize((size_t)(out[r].n_obj * out[r].frames * 19));
        if (!out[r].rows.empty()) HIP_OK(hipMemcpy(out[r].rows.data(), d_recv + pad * (size_t)r, sizeof(double) * out[r].rows.size(), hipMemcpyDeviceToHost));
    }
    for (void* p : {(void*)d_mine, (void*)d_all, (void*)d_send, (void*)d_recv}) (void)hipFree(p);
    (void)hipStreamDestroy(stream);
    NCCL_OK(ncclCommDestroy(comm));
    return out;
}
#endif

std::string join(const std::string& root, const std::string& rel)
{
    if (!rel.empty() && rel.front() == '/') return rel;
    return root + (root.empty() || root.back() == '/' ? "" : "/") + rel;
}

}  // namespace

int main(int argc, char** argv)
{
    try {
        // ---- split the command line: what is ours, what is the configuration's
        std::vector<ObjectArgs> objects;
        std::string log_root, gather_id_file;
        int batch_frames = 6, device = 0, shard_rank = 0, shard_world = 1, outlier_bands = 0;
        std::vector<char*> cfg_argv = {argv[0]};
        for (int i = 1; i < argc; ++i) {
            const std::string a = argv[i];
            if (a == "--object") {
                if (i + 2 >= argc) throw std::runtime_error("--object takes SEQUENCE_DIR NAME [MESH.obj]");
                ObjectArgs o{argv[i + 1], argv[i + 2], ""};
                i += 2;
                if (i + 1 < argc && std::strncmp(argv[i + 1], "--", 2) != 0) o.mesh = argv[++i];
                objects.push_back(o);
            } else if (a == "--log_root" && i + 1 < argc) log_root = argv[++i];
            else if (a == "--batch_frames" && i + 1 < argc) batch_frames = std::atoi(argv[++i]);
            else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
            else if (a == "--outlier_bands" && i + 1 < argc) outlier_bands = std::atoi(argv[++i]);   // roft_config::outlier_bands_per_alternative
            else if (a == "--shard" && i + 2 < argc) { shard_rank = std::atoi(argv[i + 1]); shard_world = std::atoi(argv[i + 2]); i += 2; }
            else if (a == "--gather" && i + 1 < argc) gather_id_file = argv[++i];
            else cfg_argv.push_back(argv[i]);
        }
        if (objects.empty() || log_root.empty()) throw std::runtime_error("usage: ROFT-tracker-batch --from config.cfg [--group::key value ...] --log_root DIR [--batch_frames T] [--outlier_bands B] [--device D] [--shard RANK WORLD] --object SEQUENCE_DIR NAME [MESH.obj] ...");
        if (shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world) throw std::runtime_error("--shard RANK WORLD: 0 <= RANK < WORLD");
#ifndef ROFT_WITH_RCCL
        if (!gather_id_file.empty()) throw std::runtime_error("--gather needs a build with -DROFT_WITH_RCCL (rccl.h / librccl)");
#endif
        const std::vector<ObjectArgs> all_objects = objects;   // (process 0 of a --gather job writes every object's logs)
        {
            const std::size_t per = (objects.size() + (std::size_t)shard_world - 1) / (std::size_t)shard_world;
            const std::size_t lo = std::min(objects.size(), per * (std::size_t)shard_rank), hi = std::min(objects.size(), lo + per);
            objects = std::vector<ObjectArgs>(objects.begin() + (long)lo, objects.begin() + (long)hi);
            if (objects.empty() && gather_id_file.empty()) { std::printf("tracked 0 objects over 0 frames\n"); return EXIT_SUCCESS; }
            if (objects.empty()) throw std::runtime_error("--gather: every process needs at least one object (fewer processes than objects)");
        }
        ConfigParser conf((int)cfg_argv.size(), cfg_argv.data());

        // ---- the settings main.cpp:43-147 reads
        double sample_time = 1.0 / 30.0; conf("sample_time", sample_time);
        int width = 0, height = 0; conf("camera_dataset.width", width); conf("camera_dataset.height", height);
        double fx = 0, fy = 0, cx = 0, cy = 0;
        conf("camera_dataset.fx", fx); conf("camera_dataset.fy", fy); conf("camera_dataset.cx", cx); conf("camera_dataset.cy", cy);
        std::string data_prefix, rgb_prefix, depth_prefix, data_format, rgb_format, depth_format;
        conf("camera_dataset.data_prefix", data_prefix); conf("camera_dataset.rgb_prefix", rgb_prefix); conf("camera_dataset.depth_prefix", depth_prefix);
        conf("camera_dataset.data_format", data_format); conf("camera_dataset.rgb_format", rgb_format); conf("camera_dataset.depth_format", depth_format);
        int cam_zeros = 0, cam_offset = 0; conf("camera_dataset.heading_zeros", cam_zeros); conf("camera_dataset.index_offset", cam_offset);
        Eigen::VectorXd p_v_0, p_w_0, p_cov_v_0, p_cov_w_0, p_cov_x_0, p_cov_q_0, v_v_0, v_w_0, v_cov_v_0, v_cov_w_0;
        conf("initial_condition.pose.v", p_v_0); conf("initial_condition.pose.w", p_w_0);
        conf("initial_condition.pose.cov_v", p_cov_v_0); conf("initial_condition.pose.cov_w", p_cov_w_0);
        conf("initial_condition.pose.cov_x", p_cov_x_0); conf("initial_condition.pose.cov_q", p_cov_q_0);
        conf("initial_condition.velocity.v", v_v_0); conf("initial_condition.velocity.w", v_w_0);
        conf("initial_condition.velocity.cov_v", v_cov_v_0); conf("initial_condition.velocity.cov_w", v_cov_w_0);
        Eigen::VectorXd psd_lin_acc, sigma_ang_vel, kin_q_v, kin_q_w;
        conf("kinematic_model.pose.sigma_linear", psd_lin_acc); conf("kinematic_model.pose.sigma_angular", sigma_ang_vel);
        conf("kinematic_model.velocity.sigma_linear", kin_q_v); conf("kinematic_model.velocity.sigma_angular", kin_q_w);
        Eigen::VectorXd m_cov_v, m_cov_w, m_cov_x, m_cov_q, m_cov_flow;
        conf("measurement_model.pose.cov_v", m_cov_v); conf("measurement_model.pose.cov_w", m_cov_w);
        conf("measurement_model.pose.cov_x", m_cov_x); conf("measurement_model.pose.cov_q", m_cov_q);
        conf("measurement_model.velocity.cov_flow", m_cov_flow);
        double depth_maximum = 2.0, subsampling_radius = 35.0; bool flow_weighting = true;
        conf("measurement_model.velocity.depth_maximum", depth_maximum); conf("measurement_model.velocity.subsampling_radius", subsampling_radius);
        conf("measurement_model.velocity.weight_flow", flow_weighting);
        bool use_pose = true, use_resync = true, use_velocity = true, outlier_rejection = true, flow_aided = true;
        conf("measurement_model.use_pose", use_pose); conf("measurement_model.use_pose_resync", use_resync);
        conf("measurement_model.use_velocity", use_velocity); conf("outlier_rejection.enable", outlier_rejection);
        conf("segmentation_dataset.flow_aided", flow_aided);
        bool use_internal_db = true; std::string internal_db_name;
        conf("model.use_internal_db", use_internal_db); conf("model.internal_db_name", internal_db_name);
        std::string flow_set; int flow_zeros = 0, flow_offset = 0;
        conf("optical_flow_dataset.set", flow_set); conf("optical_flow_dataset.heading_zeros", flow_zeros); conf("optical_flow_dataset.index_offset", flow_offset);
        std::string pose_path; int pose_skip_rows = 0, pose_skip_cols = 0; bool pose_reduce = true, pose_delay = true; double pose_fps = 30.0, pose_rate = 5.0;
        conf("pose_dataset.path", pose_path); conf("pose_dataset.skip_rows", pose_skip_rows); conf("pose_dataset.skip_cols", pose_skip_cols);
        conf("pose_dataset.fps_reduction", pose_reduce); conf("pose_dataset.delay", pose_delay);
        conf("pose_dataset.original_fps", pose_fps); conf("pose_dataset.desired_fps", pose_rate);
        std::string seg_format, seg_set; int seg_zeros = 0, seg_offset = 0; bool seg_reduce = true, seg_delay = true; double seg_fps = 30.0, seg_rate = 5.0;
        conf("segmentation_dataset.format", seg_format); conf("segmentation_dataset.set", seg_set);
        conf("segmentation_dataset.heading_zeros", seg_zeros); conf("segmentation_dataset.index_offset", seg_offset);
        conf("segmentation_dataset.fps_reduction", seg_reduce); conf("segmentation_dataset.delay", seg_delay);
        conf("segmentation_dataset.original_fps", seg_fps); conf("segmentation_dataset.desired_fps", seg_rate);
        double ut_alpha = 1.0, ut_beta = 2.0, ut_kappa = 0.0;
        conf("unscented_transform.alpha", ut_alpha); conf("unscented_transform.beta", ut_beta); conf("unscented_transform.kappa", ut_kappa);

        // ---- sources and meshes per object (main.cpp:327-386)
        std::vector<TrackedObject> tracked(objects.size());
        for (std::size_t o = 0; o < objects.size(); ++o) {
            TrackedObject& t = tracked[o];
            t.args = objects[o];
            t.camera = std::make_shared<CameraMeasurement>(std::make_unique<RobotsIO::Camera::DatasetCamera>(
                t.args.root, data_prefix, rgb_prefix, depth_prefix, data_format, rgb_format, depth_format, cam_zeros, cam_offset, width, height, fx, cx, fy, cy));
            ModelParameters model;
            model.name(t.args.name);
            model.use_internal_db(t.args.mesh.empty() && use_internal_db);
            model.internal_db_name(internal_db_name);
            model.mesh_external_path(t.args.mesh);
            const std::string poses = join(t.args.root, pose_path);
            if (pose_delay || pose_reduce)
                t.pose = std::make_shared<RobotsIO::Utils::DatasetTransformDelayed>(pose_fps, pose_reduce ? pose_rate : pose_fps, pose_delay, poses, pose_skip_rows, pose_skip_cols, 7);
            else t.pose = std::make_shared<RobotsIO::Utils::DatasetTransform>(poses, pose_skip_rows, pose_skip_cols, 7);
            if (seg_delay || seg_reduce)
                t.segmentation = std::make_shared<DatasetImageSegmentationDelayed>((float)seg_fps, (float)(seg_reduce ? seg_rate : seg_fps), seg_delay, t.args.root, seg_format,
                                                                                   width, height, seg_set, model, seg_zeros, seg_offset);
            else t.segmentation = std::make_shared<DatasetImageSegmentation>(t.args.root, seg_format, width, height, seg_set, model, seg_zeros, seg_offset);
            t.flow = std::make_shared<DatasetImageOpticalFlow>(t.args.root, flow_set, width, height, flow_zeros, flow_offset);
            if (outlier_rejection && use_pose) {
                std::istringstream mesh_text(MeshResource(model).as_string());
                parse_obj_mesh(mesh_text, t.args.name, t.verts, t.tris);
            }
        }

        // ---- one engine for all of them (what ROFTFilter's constructor sets for one, ROFTFilter.h)
        roft_config cfg{};
        const int flow_type = tracked[0].flow->get_matrix_type() == CV_16SC2 ? ROFT_FLOW_S16C2 : ROFT_FLOW_F32C2;
        compat::throw_if(roft_default_config(&cfg, width, height, flow_type), "roft_default_config");
        cfg.cam.fx = fx; cfg.cam.fy = fy; cfg.cam.cx = cx; cfg.cam.cy = cy;
        cfg.flow_grid = (int)tracked[0].flow->get_grid_size();
        cfg.flow_scale = tracked[0].flow->get_scaling_factor();
        for (const TrackedObject& t : tracked)
            if (t.flow->get_matrix_type() != tracked[0].flow->get_matrix_type() || t.flow->get_grid_size() != tracked[0].flow->get_grid_size())
                throw std::runtime_error("all sequences of a batch share one optical-flow format");
        cfg.sample_time = sample_time;
        cfg.ut = roft_ut_params{ut_alpha, ut_beta, ut_kappa};
        cfg.depth_maximum = depth_maximum;
        cfg.subsampling_radius = subsampling_radius;
        cfg.flow_weighting = flow_weighting;
        cfg.use_pose = use_pose; cfg.use_pose_resync = use_resync; cfg.use_velocity = use_velocity;
        cfg.outlier_rejection = outlier_rejection;
        cfg.flow_aided_segmentation = flow_aided;
        cfg.mask_frames_between = tracked[0].segmentation->get_frames_between_iterations();
        cfg.pose_frames_between = std::max(0, tracked[0].pose->get_frames_between_iterations());
        cfg.max_objects = (int)tracked.size();
        cfg.device = device;
        if (batch_frames < 1 || batch_frames > ROFT_MAX_BATCH_FRAMES) throw std::runtime_error("--batch_frames out of range");
        cfg.max_batch_frames = batch_frames;
        cfg.outlier_bands_per_alternative = outlier_bands;   // (0: the engine's choice, which follows the load)
        ROFTFilterBatch engine(cfg);
        for (TrackedObject& t : tracked) {
            roft_object_desc d{};
            compat::throw_if(roft_default_object(&d), "roft_default_object");
            // initial pose: the first row of the object's pose file (test/test.sh:120-123)
            const auto rows = compat::read_rows(join(t.args.root, pose_path), pose_skip_rows, pose_skip_cols, 7);
            if (rows.empty()) throw std::runtime_error("no poses in " + join(t.args.root, pose_path));
            const auto T0 = compat::transform_of(rows[0].data());
            for (int i = 0; i < 3; ++i) { d.p_mean0[i] = p_v_0(i); d.p_mean0[3 + i] = p_w_0(i); d.p_mean0[6 + i] = T0.translation()[i]; }
            for (int i = 0; i < 4; ++i) d.p_mean0[9 + i] = T0.quaternion()[i];
            for (int i = 0; i < 3; ++i) {
                d.p_cov0_diag[i] = p_cov_v_0(i); d.p_cov0_diag[3 + i] = p_cov_w_0(i); d.p_cov0_diag[6 + i] = p_cov_x_0(i); d.p_cov0_diag[9 + i] = p_cov_q_0(i);
                d.v_mean0[i] = v_v_0(i); d.v_mean0[3 + i] = v_w_0(i); d.v_cov0_diag[i] = v_cov_v_0(i); d.v_cov0_diag[3 + i] = v_cov_w_0(i);
                d.v_q_diag[i] = kin_q_v(i); d.v_q_diag[3 + i] = kin_q_w(i);
                d.p_sigma_ang_vel[i] = sigma_ang_vel(i); d.p_psd_lin_acc[i] = psd_lin_acc(i);
                d.p_meas_cov_v[i] = m_cov_v(i); d.p_meas_cov_w[i] = m_cov_w(i); d.p_meas_cov_x[i] = m_cov_x(i); d.p_meas_cov_q[i] = m_cov_q(i);
            }
            d.v_meas_cov_flow[0] = m_cov_flow(0); d.v_meas_cov_flow[1] = m_cov_flow(1);
            d.mesh = roft_mesh{t.verts.data(), (int)(t.verts.size() / 3), t.tris.data(), (int)(t.tris.size() / 3)};
            engine.add_object(d);
        }

        // ---- frames: every source polled as ROFTFilter::filtering_step polls it, T frames handed over at a time
        const int n_obj = (int)tracked.size();
        const int log_capacity = 1 << 16;
        compat::throw_if(roft_engine_enable_log(engine.engine(), log_capacity), "roft_engine_enable_log");
        int frames = 0;
        bool more = true;
        while (more && frames + batch_frames <= log_capacity) {
            std::vector<FrameBuffers> buffers((std::size_t)batch_frames * n_obj);
            std::vector<roft_frame_input> inputs((std::size_t)batch_frames * n_obj);
            int t_in = 0;
            for (; t_in < batch_frames && more; ++t_in) {
                for (int o = 0; o < n_obj && more; ++o) {
                    TrackedObject& t = tracked[o];
                    FrameBuffers& b = buffers[(std::size_t)t_in * n_obj + o];
                    roft_frame_input& in = inputs[(std::size_t)t_in * n_obj + o];
                    in = roft_frame_input{};
                    if (!t.camera->freeze(CameraMeasurementType::RGBD)) { more = false; break; }
                    bool valid = false;
                    bfl::Data cam_data;
                    std::tie(valid, cam_data) = t.camera->measure();
                    b.depth = std::get<2>(*bfl::any::any_cast<CameraMeasurement::CameraMeasurementTuple>(&cam_data));
                    double stamp = 0.0;
                    bool has_stamp = false;
                    std::tie(has_stamp, stamp) = t.camera->camera_time_stamp_rgb();
                    in.dt = (has_stamp && t.last_stamp != -1.0) ? stamp - t.last_stamp : sample_time;
                    if (has_stamp) t.last_stamp = stamp;
                    if (t.flow->is_stepping_required()) t.flow->step_frame();
                    bool valid_flow = false;
                    std::tie(valid_flow, b.flow) = t.flow->flow(false);
                    if (t.segmentation->is_stepping_required()) t.segmentation->step_frame();
                    bool new_mask = false;
                    std::tie(new_mask, b.mask) = t.segmentation->segmentation(false);
                    if (!t.mask_received && !new_mask) throw std::runtime_error("the first frame of " + t.args.root + " delivers no mask");
                    t.mask_received = true;
                    in.depth = b.depth.data();
                    in.flow = valid_flow ? b.flow.data : nullptr;
                    in.mask = new_mask ? b.mask.data : nullptr;
                    in.mem_kind = ROFT_MEM_HOST;
                    if (use_pose && t.pose->freeze(false)) {
                        const auto P = t.pose->transform();
                        in.pose_valid = 1;
                        for (int i = 0; i < 3; ++i) in.pose_x[i] = P.translation()[i];
                        for (int i = 0; i < 4; ++i) in.pose_q[i] = P.quaternion()[i];
                    }
                }
                if (!more) break;
            }
            if (t_in == 0) break;
            inputs.resize((std::size_t)t_in * n_obj);
            engine.filtering_steps(inputs, n_obj, t_in);
            frames += t_in;
        }
        engine.wait();

        // ---- logs in the reference's format (ROFTFilter.cpp:386-394 through bfl::Logger)
        std::vector<double> rows((std::size_t)frames * n_obj * 19);
        compat::throw_if(roft_engine_get_log_rows(engine.engine(), 0, frames, rows.data()), "roft_engine_get_log_rows");
        auto write_logs = [&](const std::vector<double>& rows, int n_obj, int frames, const ObjectArgs* names) {
        for (int o = 0; o < n_obj; ++o) {
            const std::string dir = log_root + "/" + names[o].name;
            std::error_code ec;
            std::filesystem::create_directories(dir, ec);
            if (ec) throw std::runtime_error("cannot create " + dir + ": " + ec.message());
            std::remove((dir + "/pose_estimate.txt").c_str());
            std::remove((dir + "/velocity_estimate.txt").c_str());
            EstimateLog log;
            if (!log.enable_log(dir, "")) throw std::runtime_error("cannot write the logs in " + dir);
            for (int f = 0; f < frames; ++f) {
                const double* r = &rows[((std::size_t)f * n_obj + o) * 19];
                Eigen::VectorXd p(13), v(6);
                for (int i = 0; i < 9; ++i) p(i) = r[i];
                double axis[3], angle;
                ROFTFilter::axis_angle(r + 9, axis, angle);
                for (int i = 0; i < 3; ++i) p(9 + i) = axis[i];
                p(12) = angle;
                for (int i = 0; i < 6; ++i) v(i) = r[13 + i];
                log.logger(p.transpose(), v.transpose());
            }
        }
        };
#ifdef ROFT_WITH_RCCL
        if (!gather_id_file.empty()) {
            const std::vector<ShardRows> shards = gather_rows(gather_id_file, shard_rank, shard_world, device, n_obj, frames, rows);
            long total = 0;
            if (shard_rank == 0) {
                const std::size_t per = (all_objects.size() + (std::size_t)shard_world - 1) / (std::size_t)shard_world;
                for (int r = 0; r < shard_world; ++r) {
                    write_logs(shards[r].rows, (int)shards[r].n_obj, (int)shards[r].frames, all_objects.data() + per * (std::size_t)r);
                    total += shards[r].n_obj;
                }
                std::remove(gather_id_file.c_str());
            }
            std::printf("tracked %d objects over %d frames (process %d of %d; rows all-gathered over RCCL%s)\n", n_obj, frames, shard_rank, shard_world,
                        shard_rank == 0 ? (", logs of " + std::to_string(total) + " objects written").c_str() : "");
            return EXIT_SUCCESS;
        }
#endif
        write_logs(rows, n_obj, frames, objects.data());
        std::printf("tracked %d objects over %d frames\n", n_obj, frames);
        return EXIT_SUCCESS;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ROFT-tracker-batch: %s\n", e.what());
        return EXIT_FAILURE;
    }
}
