// Writes the five log files of the tracker with the bfl::Logger stand-in of include/ROFT/Compat.h (the writer the C++
// facade uses): ROFTFilter::log_file_names (pose_estimate, velocity_estimate, execution_times) and
// CartesianQuaternionMeasurement::log_file_names (pose_measurements, velocity_measurements).  Rows: fixed awkward values.
//   log_writer <dir>
#include <cstdio>

#include "ROFT/Compat.h"

struct FilterLog : bfl::Logger {
    std::vector<std::string> log_file_names(const std::string& p, const std::string& n) override
    {
        return {p + "/" + n + "pose_estimate", p + "/" + n + "velocity_estimate", p + "/" + n + "execution_times"};
    }
};
struct MeasLog : bfl::Logger {
    std::vector<std::string> log_file_names(const std::string& p, const std::string& n) override
    {
        return {p + "/" + n + "pose_measurements", p + "/" + n + "velocity_measurements"};
    }
};

int main(int, char** argv)
{
    FilterLog f;
    MeasLog m;
    if (!f.enable_log(argv[1], "") || !m.enable_log(argv[1], "")) return 1;
    for (int k = 0; k < 4; ++k) {
        Eigen::VectorXd pose(13), vel(6), times(2), pm(7), vm(6);
        for (int i = 0; i < 13; ++i) pose(i) = (i % 3 - 1) * 0.123456789 * (k + 1) + (i == 12 ? 3.0 : 0.0) + (i == 4 ? 1e-7 : 0.0) - (i == 7 ? 12345.678 : 0.0);
        for (int i = 0; i < 6; ++i) { vel(i) = -0.5 * i + 1e-5 * k; vm(i) = vel(i) * 2.0; }
        times(0) = 3 + k; times(1) = 0;
        for (int i = 0; i < 7; ++i) pm(i) = (k == 0) ? (i == 3 ? 1.0 : 0.0) : 0.01 * i - 0.02 * k;
        f.logger(pose.transpose(), vel.transpose(), times.transpose());
        m.logger(pm.transpose(), vm.transpose());
    }
    return 0;
}
