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
#include <poll.h>
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

// Eigen::Quaterniond(R) (Eigen/src/Geometry/Quaternion.h, QuaternionBase::operator=(MatrixBase)): the branch on the trace and, for a non-positive trace, on the
// largest diagonal entry -- a ground vehicle that has turned around has trace(R) < 0
static void quat_from_R(const double* R, double* q /* x y z w */) {
    const double t = R[0] + R[4] + R[8];
    if (t > 0) {
        double s = std::sqrt(t + 1.0);
        q[3] = 0.5 * s; s = 0.5 / s;
        q[0] = (R[7] - R[5]) * s; q[1] = (R[2] - R[6]) * s; q[2] = (R[3] - R[1]) * s;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[i] = 0.5 * s; s = 0.5 / s;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * s;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * s;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * s;
    }
}

// one rank of `gf_replay --ranks N`: its share of the recordings, then the pose exchange.  The unique id of the communicator arrives over a pipe the parent opened
// before the fork (rank 0 writes world - 1 copies, every other rank reads its own): no file name in /tmp that somebody else could have created first.
static int run_rank(int rank, int world, const char* config, const std::vector<std::string>& dirs, int id_read_fd, const std::vector<int>& id_write_fds) {
    gf_comm* comm = nullptr;
    int failed = 0;
    try {
        int ndev = 0;
        if (gf_device_count(&ndev) != GF_OK || ndev < 1) throw std::runtime_error("no HIP device");
        const int device = rank % ndev;
        if (gf_set_device(device) != GF_OK) throw std::runtime_error(gf_last_error());
        gf_pin_thread_to_device_node(device);
        unsigned char id[128];
        if (rank == 0) {
            if (gf_comm_unique_id(id) != GF_OK) throw std::runtime_error(gf_last_error());
            for (int fd : id_write_fds) if (write(fd, id, 128) != 128) throw std::runtime_error("cannot hand the unique id to a rank");
        } else {
            size_t got = 0;
            while (got < 128) {   // EOF = rank 0 is gone; and a bound on the wait in case it hangs inside the runtime (GF_REPLAY_ID_TIMEOUT_S, default 120 s)
                struct pollfd pf = {id_read_fd, POLLIN, 0};
                static const int tmo_ms = 1000 * (getenv("GF_REPLAY_ID_TIMEOUT_S") ? std::max(1, atoi(getenv("GF_REPLAY_ID_TIMEOUT_S"))) : 120);
                const int pr = poll(&pf, 1, tmo_ms);
                if (pr == 0) throw std::runtime_error("rank " + std::to_string(rank) + ": no unique id from rank 0 within " + std::to_string(tmo_ms / 1000) + " s");
                const ssize_t n = pr > 0 ? read(id_read_fd, id + got, 128 - got) : -1;
                if (n <= 0) throw std::runtime_error("rank " + std::to_string(rank) + ": no unique id from rank 0 (it failed before the exchange)");
                got += (size_t)n;
            }
        }
        if (gf_comm_create(id, world, rank, device, &comm) != GF_OK) throw std::runtime_error(gf_last_error());
    } catch (const std::exception& e) {
        fprintf(stderr, "gf_replay rank %d: %s\n", rank, e.what());
        return 1;      // before the communicator exists nobody waits for this rank inside a collective (ncclCommInitRank of the others fails or times out with it)
    }
    const int rounds = ((int)dirs.size() + world - 1) / world;
    std::vector<double> all((size_t)world * 8);
    for (int j = 0; j < rounds; j++) {
        const int k = j * world + rank;
        double mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // px py pz qx qy qz qw, then 1: this rank had a sequence in this round, -1: it failed on it
        if (k < (int)dirs.size()) {
            // a rank whose replay throws (missing directory, bad recording, estimator error) still takes part in this and every later exchange: the others would
            // otherwise block inside ncclAllGather for good (round-4 advisor)
            try {
                gf::Estimator estimator;
                replay_one(config, dirs[k], false, dirs[k] + "/vio.txt", estimator, true);
                // the state straight from the handle: the class mirrors it after inputImage, and the last frame is processed when the IMU samples behind it arrive
                const int W = estimator.cfg.window_size, N = W + 1;
                std::vector<double> Pa(3 * N), Ra(9 * N), Va(3 * N), Baa(3 * N), Bga(3 * N), Ha(N);
                int info[16]; double extr[32];
                if (gf_estimator_get_state(estimator.handle(), Pa.data(), Ra.data(), Va.data(), Baa.data(), Bga.data(), Ha.data(), info, extr) != GF_OK) throw std::runtime_error(gf_last_error());
                for (int c = 0; c < 3; c++) mine[c] = Pa[3 * W + c];
                quat_from_R(Ra.data() + 9 * W, mine + 3);
                mine[7] = 1.0;
            } catch (const std::exception& e) {
                fprintf(stderr, "gf_replay rank %d: sequence %d (%s): %s\n", rank, k, dirs[k].c_str(), e.what());
                for (double& v : mine) v = 0.0;
                mine[7] = -1.0; failed = 1;
            }
        }
        if (gf_comm_allgather_f64(comm, mine, 8, all.data()) != GF_OK) { fprintf(stderr, "gf_replay rank %d: %s\n", rank, gf_last_error()); failed = 1; break; }
        if (rank == 0)
            for (int r = 0; r < world; r++) {
                if (all[(size_t)r * 8 + 7] > 0.0)
                    printf("gf_replay: sequence %d (rank %d): newest pose %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n", j * world + r, r, all[r * 8 + 0], all[r * 8 + 1], all[r * 8 + 2],
                           all[r * 8 + 3], all[r * 8 + 4], all[r * 8 + 5], all[r * 8 + 6]);
                else if (all[(size_t)r * 8 + 7] < 0.0) { printf("gf_replay: sequence %d (rank %d): FAILED\n", j * world + r, r); failed = 1; }
            }
    }
    gf_comm_destroy(comm);
    return failed;
}

int main(int argc, char** argv) {
    if (argc >= 5 && std::string(argv[1]) == "--ranks") {
        const int world = atoi(argv[2]);
        if (world < 1 || world > 64) { fprintf(stderr, "gf_replay: --ranks must be in 1..64\n"); return 2; }
        std::vector<std::string> dirs(argv + 4, argv + argc);
        {   // more ranks than devices put two ranks on one GPU, which RCCL refuses ("Duplicate GPU"): say so instead of failing inside ncclCommInitRank
            // (GF_REPLAY_ALLOW_SHARED_DEVICE=1 tries anyway: some RCCL builds accept it)
            const pid_t probe = fork();   // the device count from a child: the parent must not touch the HIP runtime before it forks the ranks
            if (probe == 0) { int n = 0; _exit(gf_device_count(&n) == GF_OK ? std::min(n, 200) : 0); }
            int st = 0; waitpid(probe, &st, 0);
            const int ndev = WIFEXITED(st) ? WEXITSTATUS(st) : 0;
            if (ndev < 1) { fprintf(stderr, "gf_replay: no HIP device\n"); return 1; }
            if (world > ndev && !getenv("GF_REPLAY_ALLOW_SHARED_DEVICE")) { fprintf(stderr, "gf_replay: --ranks %d on %d device(s): one rank per GPU (RCCL refuses two ranks on one device)\n", world, ndev); return 2; }
        }
        std::vector<int> rd(world, -1), wr;
        for (int r = 1; r < world; r++) { int fds[2]; if (pipe(fds) != 0) { perror("pipe"); return 1; } rd[r] = fds[0]; wr.push_back(fds[1]); }
        std::vector<pid_t> kids;
        for (int r = 0; r < world; r++) {                    // fork before anything touches the HIP runtime: every rank initialises its own
            const pid_t p = fork();
            if (p < 0) { perror("fork"); return 1; }
            if (p == 0) {
                // a child keeps only the pipe ends it uses (round-5 advisor): with every inherited write end open in ranks >= 1, a rank 0 that died before it wrote the id
                // left the others in read() for good (no EOF while any copy of the write end is open) and the parent in waitpid behind them
                for (int q = 1; q < world; q++) if (q != r) close(rd[q]);
                if (r != 0) for (int fd : wr) close(fd);
                const int rc = run_rank(r, world, argv[3], dirs, rd[r], r == 0 ? wr : std::vector<int>());
                fflush(nullptr); _exit(rc);   // _exit: the parent's atexit handlers are not this child's; rank 0's write ends close with it, on the error paths too
            }
            kids.push_back(p);
        }
        for (int fd : rd) if (fd >= 0) close(fd);
        for (int fd : wr) close(fd);
        int rc = 0;
        for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1; }
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
            if (gf_estimator_get_gnss_state(estimator.handle(), gi, nullptr, nullptr, &yaw, anc, ecef, enu) == GF_OK) {
                printf("gf_replay: gnss_ready %d, anchor %.4f %.4f %.4f, ecef %.4f %.4f %.4f\n", gi[0], anc[0], anc[1], anc[2], ecef[0], ecef[1], ecef[2]);
                // the same state without the rounding of the line above (hex floats: every bit), for whoever compares it with another pipeline (tests/test_replay_gpu.py)
                printf("gf_replay: gnss_state_bits anchor %a %a %a ecef %a %a %a yaw %a\n", anc[0], anc[1], anc[2], ecef[0], ecef[1], ecef[2], yaw);
            }
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "gf_replay: %s\n", e.what());
        return 1;
    }
    return 0;
}
