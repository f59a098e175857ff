// gf_replay — `rosrun vins vins_node <config.yaml>` + `rosbag play` without ROS:
//   gf_replay <config.yaml> <dataset dir> [<vio.txt>]
//   gf_replay --ranks N <config.yaml> <dataset dir 0> <dataset dir 1> ...   SURVEY.md §8(e) without torch: N processes, rank r on GPU r (mod the device count)
//                                                                   replays the recordings r, r + N, ... (each writes <dir>/vio.txt) and the ranks exchange the newest
//                                                                   pose of their sequences with ncclAllGather over RCCL (gf_comm_*); rank 0 prints all of them
//   gf_replay <config.yaml> --bag <recording.bag> [<vio.txt>]      the recording itself (ROS bag format 2.0, host/rosbag_reader.h); topics = the config's
//                                                                   imu_topic / wheel_topic / image0_topic / image1_topic (parameters.cpp:156-157, :211, :230)
// reads the reference's own YAML configuration (parameters.cpp key names), replays the recorded IMU / wheel / RGB / depth messages of
// <dataset dir> (layout in host/replay_node.h) through FeatureTracker::trackImage and Estimator::processImage on the GPU, and writes the
// trajectory file the reference writes (output_path/vio.txt, TUM format) — to <vio.txt> when given, else to `output_path` of the config.
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

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

static void replay_one(const char* config, const std::string& source, bool from_bag, const std::string& out, gf::Estimator& estimator, bool quiet) {
    estimator.readParameters(config);
    // `max_solver_time` (a wall-clock cap on ceres::Solve) makes a replay depend on the machine and on what else it is doing: honoured only on request
    if (!getenv("GF_HONOUR_SOLVER_TIME")) estimator.cfg.max_solver_time = 0.0;
    estimator.setParameter();
    estimator.setResultPath(out);
    gf::ReplayNode<gf::Estimator> node(estimator);
    const std::string wr = yaml_string(config, "w_replace");
    node.w_replace = wr.empty() ? 0 : atoi(wr.c_str());
    node.gnss_local_time_diff = estimator.cfg.gnss_enable ? estimator.cfg.gnss_local_time_diff : 0.0;   // rosNodeTest.cpp:703-708
    if (from_bag) node.run_bag(source, yaml_string(config, "imu_topic"), yaml_string(config, "wheel_topic"), yaml_string(config, "image0_topic"), yaml_string(config, "image1_topic"));
    else node.run(source);
    if (!quiet)
        printf("gf_replay: %ld RGB-D pairs (%ld / %ld unpaired frames thrown), %ld GNSS epochs, solver_flag %d, trajectory in %s\n", node.n_pairs, node.n_thrown0,
               node.n_thrown1, node.n_gnss, (int)estimator.solver_flag, out.c_str());
}

// one rank of `gf_replay --ranks N`: its share of the recordings, then the pose exchange
static int run_rank(int rank, int world, const char* config, const std::vector<std::string>& dirs, const std::string& idfile) {
    try {
        int ndev = 0;
        if (gf_device_count(&ndev) != GF_OK || ndev < 1) throw std::runtime_error("no HIP device");
        const int device = rank % ndev;
        if (gf_set_device(device) != GF_OK) throw std::runtime_error(gf_last_error());
        gf_pin_thread_to_device_node(device);
        unsigned char id[128];
        if (rank == 0) {
            if (gf_comm_unique_id(id) != GF_OK) throw std::runtime_error(gf_last_error());
            const std::string tmp = idfile + ".tmp";
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(id, 1, 128, f) != 128) throw std::runtime_error("cannot write " + tmp);
            fclose(f);
            if (rename(tmp.c_str(), idfile.c_str()) != 0) throw std::runtime_error("cannot publish " + idfile);
        } else {
            FILE* f = nullptr;
            for (int tries = 0; tries < 1200 && !(f = fopen(idfile.c_str(), "rb")); tries++) usleep(50000);
            if (!f || fread(id, 1, 128, f) != 128) throw std::runtime_error("rank " + std::to_string(rank) + ": no unique id from rank 0");
            fclose(f);
        }
        gf_comm* comm = nullptr;
        if (gf_comm_create(id, world, rank, device, &comm) != GF_OK) throw std::runtime_error(gf_last_error());
        const int rounds = ((int)dirs.size() + world - 1) / world;
        std::vector<double> all((size_t)world * 8);
        for (int j = 0; j < rounds; j++) {
            const int k = j * world + rank;
            double mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // px py pz qx qy qz qw, then 1 when this rank had a sequence in this round
            if (k < (int)dirs.size()) {
                gf::Estimator estimator;
                replay_one(config, dirs[k], false, dirs[k] + "/vio.txt", estimator, true);
                // the state straight from the handle: the class mirrors it after inputImage, and the last frame is processed when the IMU samples behind it arrive
                const int W = estimator.cfg.window_size, N = W + 1;
                std::vector<double> Pa(3 * N), Ra(9 * N), Va(3 * N), Baa(3 * N), Bga(3 * N), Ha(N);
                int info[16]; double extr[32];
                if (gf_estimator_get_state(estimator.handle(), Pa.data(), Ra.data(), Va.data(), Baa.data(), Bga.data(), Ha.data(), info, extr) != GF_OK) throw std::runtime_error(gf_last_error());
                const double* R = Ra.data() + 9 * W;
                const double t = R[0] + R[4] + R[8];
                double q[4] = {0, 0, 0, 1};                   // Eigen::Quaterniond(R) for trace > 0 (a ground vehicle near its start attitude); else left as the identity
                if (t > 0) { const double s = std::sqrt(t + 1.0) * 2.0; q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s; }
                for (int c = 0; c < 3; c++) mine[c] = Pa[3 * W + c];
                for (int c = 0; c < 4; c++) mine[3 + c] = q[c];
                mine[7] = 1.0;
            }
            if (gf_comm_allgather_f64(comm, mine, 8, all.data()) != GF_OK) throw std::runtime_error(gf_last_error());
            if (rank == 0)
                for (int r = 0; r < world; r++)
                    if (all[(size_t)r * 8 + 7] != 0.0)
                        printf("gf_replay: sequence %d (rank %d): newest pose %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n", j * world + r, r, all[r * 8 + 0], all[r * 8 + 1], all[r * 8 + 2],
                               all[r * 8 + 3], all[r * 8 + 4], all[r * 8 + 5], all[r * 8 + 6]);
        }
        gf_comm_destroy(comm);
    } catch (const std::exception& e) {
        fprintf(stderr, "gf_replay rank %d: %s\n", rank, e.what());
        return 1;
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 5 && std::string(argv[1]) == "--ranks") {
        const int world = atoi(argv[2]);
        if (world < 1 || world > 64) { fprintf(stderr, "gf_replay: --ranks must be in 1..64\n"); return 2; }
        std::vector<std::string> dirs(argv + 4, argv + argc);
        char idfile[] = "/tmp/gf_replay_id_XXXXXX";
        const int fd = mkstemp(idfile);
        if (fd < 0) { perror("mkstemp"); return 1; }
        close(fd);
        unlink(idfile);                                      // rank 0 publishes the id under this name
        std::vector<pid_t> kids;
        for (int r = 0; r < world; r++) {                    // fork before anything touches the HIP runtime: every rank initialises its own
            const pid_t p = fork();
            if (p < 0) { perror("fork"); return 1; }
            if (p == 0) { const int rc = run_rank(r, world, argv[3], dirs, idfile); fflush(nullptr); _exit(rc); }   // _exit: the parent's atexit handlers are not this child's

            kids.push_back(p);
        }
        int rc = 0;
        for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1; }
        unlink(idfile);
        return rc;
    }
    const bool from_bag = argc >= 4 && std::string(argv[2]) == "--bag";
    if (argc < 3 || (from_bag && argc < 4)) { fprintf(stderr, "usage: %s <config.yaml> <dataset dir> [<vio.txt>]\n       %s <config.yaml> --bag <recording.bag> [<vio.txt>]\n       %s --ranks N <config.yaml> <dataset dir> ...\n", argv[0], argv[0], argv[0]); return 2; }
    const int out_arg = from_bag ? 4 : 3;
    try {
        gf::Estimator estimator;
        const std::string out = argc > out_arg ? argv[out_arg] : yaml_string(argv[1], "output_path") + "/vio.txt";
        replay_one(argv[1], from_bag ? argv[3] : argv[2], from_bag, out, estimator, false);
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
