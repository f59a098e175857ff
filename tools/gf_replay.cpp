// gf_replay — `rosrun vins vins_node <config.yaml>` + `rosbag play` without ROS:
//   gf_replay <config.yaml> <dataset dir> [<vio.txt>]
//   gf_replay <config.yaml> --bag <recording.bag> [<vio.txt>]      the recording itself (ROS bag format 2.0, host/rosbag_reader.h); topics = the config's
//                                                                   imu_topic / wheel_topic / image0_topic / image1_topic (parameters.cpp:156-157, :211, :230)
// reads the reference's own YAML configuration (parameters.cpp key names), replays the recorded IMU / wheel / RGB / depth messages of
// <dataset dir> (layout in host/replay_node.h) through FeatureTracker::trackImage and Estimator::processImage on the GPU, and writes the
// trajectory file the reference writes (output_path/vio.txt, TUM format) — to <vio.txt> when given, else to `output_path` of the config.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include "../ground-fusion_amd/host/replay_node.h"

static std::string yaml_string(const std::string& file, const std::string& key) {   // top-level string / scalar of the config, "" if absent
    std::ifstream in(file);
    std::string line;
    while (std::getline(in, line)) {
        if (line.compare(0, key.size() + 1, key + ":") != 0) continue;
        std::string v = line.substr(key.size() + 1);
        const size_t h = v.find(" #");
        if (h != std::string::npos) v = v.substr(0, h);
        const size_t a = v.find_first_not_of(" \t\""), b = v.find_last_not_of(" \t\"\r");
        return a == std::string::npos ? std::string() : v.substr(a, b - a + 1);
    }
    return std::string();
}

int main(int argc, char** argv) {
    const bool from_bag = argc >= 4 && std::string(argv[2]) == "--bag";
    if (argc < 3 || (from_bag && argc < 4)) { fprintf(stderr, "usage: %s <config.yaml> <dataset dir> [<vio.txt>]\n       %s <config.yaml> --bag <recording.bag> [<vio.txt>]\n", argv[0], argv[0]); return 2; }
    const int out_arg = from_bag ? 4 : 3;
    try {
        gf::Estimator estimator;
        estimator.readParameters(argv[1]);
        // `max_solver_time` (a wall-clock cap on ceres::Solve) makes a replay depend on the machine and on what else it is doing: honoured only on request
        if (!getenv("GF_HONOUR_SOLVER_TIME")) estimator.cfg.max_solver_time = 0.0;
        estimator.setParameter();
        const std::string out = argc > out_arg ? argv[out_arg] : yaml_string(argv[1], "output_path") + "/vio.txt";
        estimator.setResultPath(out);
        gf::ReplayNode<gf::Estimator> node(estimator);
        const std::string wr = yaml_string(argv[1], "w_replace");
        node.w_replace = wr.empty() ? 0 : atoi(wr.c_str());
        node.gnss_local_time_diff = estimator.cfg.gnss_enable ? estimator.cfg.gnss_local_time_diff : 0.0;   // rosNodeTest.cpp:703-708
        if (from_bag) node.run_bag(argv[3], yaml_string(argv[1], "imu_topic"), yaml_string(argv[1], "wheel_topic"), yaml_string(argv[1], "image0_topic"), yaml_string(argv[1], "image1_topic"));
        else node.run(argv[2]);
        printf("gf_replay: %ld RGB-D pairs (%ld / %ld unpaired frames thrown), %ld GNSS epochs, solver_flag %d, trajectory in %s\n", node.n_pairs, node.n_thrown0,
               node.n_thrown1, node.n_gnss, (int)estimator.solver_flag, out.c_str());
        if (estimator.cfg.gnss_enable) {   // gnss_result.txt of the reference carries the ECEF / ENU position; here as one closing line
            int gi[8]; double yaw, anc[3], ecef[3], enu[3];
            if (gf_estimator_get_gnss_state(estimator.handle(), gi, nullptr, nullptr, &yaw, anc, ecef, enu) == GF_OK)
                printf("gf_replay: gnss_ready %d, anchor %.4f %.4f %.4f, ecef %.4f %.4f %.4f\n", gi[0], anc[0], anc[1], anc[2], ecef[0], ecef[1], ecef[2]);
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "gf_replay: %s\n", e.what());
        return 1;
    }
    return 0;
}
