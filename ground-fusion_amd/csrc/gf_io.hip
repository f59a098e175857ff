// gf_io.hip — ROS-free I/O around the hot path.  Host-only C++; no kernels, no OpenCV:
//   gf_estimator_cfg_from_yaml   readParameters(std::string), vins_estimator/src/estimator/parameters.cpp:138-558, plus the camera file it
//                                points at (camodocal PinholeCamera::Parameters::readFromYamlFile, camera_models/src/camera_models/PinholeCamera.cc:145-183)
//   gf_tum_append                the trajectory line pubOdometry appends to VINS_RESULT_PATH, utility/visualization.cpp:346-357
//   gf_pgm_read                  raw 8/16-bit frames in place of sensor_msgs::Image + cv_bridge (rosNodeTest.cpp:229-287)
// The YAML reader understands the subset of cv::FileStorage's YAML 1.0 that the shipped config/*/*.yaml files use: top-level `key: scalar`,
// quoted strings, `#` comments, `!!opencv-matrix` nodes (rows / cols / dt / data: [ ... ] over several lines) and one level of nested maps.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/groundfusion_hip.h"
#include "../host/rosbag_reader.h"
#include "gf_dmath.hpp"

namespace gf { int set_err(int code, const char* fmt, ...); }

namespace {

struct YamlMat { int rows = 0, cols = 0; std::vector<double> data; };
struct YamlDoc {
    std::map<std::string, std::string> scalar;   // "key" or "parent.key"
    std::map<std::string, YamlMat> mat;
    // cv::FileNode conversions: a missing node reads as 0 / "" (persistence.cpp: FileNode::operator int/double of an empty node)
    bool has(const std::string& k) const { return scalar.count(k) != 0; }
    double real(const std::string& k) const { auto it = scalar.find(k); return it == scalar.end() ? 0.0 : strtod(it->second.c_str(), nullptr); }
    int integer(const std::string& k) const {      // real -> int goes through cvRound (round half to even)
        auto it = scalar.find(k);
        if (it == scalar.end()) return 0;
        return (int)nearbyint(strtod(it->second.c_str(), nullptr));
    }
    std::string str(const std::string& k) const { auto it = scalar.find(k); return it == scalar.end() ? std::string() : it->second; }
};

std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}
std::string strip_comment(const std::string& s) {
    bool in_s = false, in_d = false;
    for (size_t i = 0; i < s.size(); i++) {
        const char c = s[i];
        if (c == '"' && !in_s) in_d = !in_d;
        else if (c == '\'' && !in_d) in_s = !in_s;
        else if (c == '#' && !in_s && !in_d && (i == 0 || isspace((unsigned char)s[i - 1]))) return s.substr(0, i);
    }
    return s;
}
std::string unquote(const std::string& v) {
    if (v.size() >= 2 && ((v.front() == '"' && v.back() == '"') || (v.front() == '\'' && v.back() == '\''))) return v.substr(1, v.size() - 2);
    return v;
}

bool parse_yaml(const char* path, YamlDoc& doc, std::string& err) {
    std::ifstream in(path);
    if (!in) { err = std::string("cannot open ") + path; return false; }
    std::string line, parent;          // parent: the open nested map / matrix node, "" at top level
    bool parent_is_mat = false;
    std::string pending_key, pending;  // a flow sequence `[ ...` still waiting for its `]`
    int lineno = 0;
    auto finish_seq = [&](const std::string& key, std::string body) -> bool {
        for (char& c : body) if (c == '[' || c == ']' || c == ',') c = ' ';
        std::istringstream ss(body);
        std::string tok;
        std::vector<double> v;
        while (ss >> tok) {
            char* end = nullptr;
            const double x = strtod(tok.c_str(), &end);
            if (end == tok.c_str() || *end) { err = "line " + std::to_string(lineno) + ": bad number '" + tok + "'"; return false; }
            v.push_back(x);
        }
        if (parent_is_mat && key == "data") doc.mat[parent].data = v;
        else { YamlMat m; m.rows = 1; m.cols = (int)v.size(); m.data = v; doc.mat[parent.empty() ? key : parent + "." + key] = m; }
        return true;
    };
    while (std::getline(in, line)) {
        lineno++;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (lineno == 1 && line.rfind("%YAML", 0) == 0) continue;
        line = strip_comment(line);
        if (!pending_key.empty()) {                 // continuation of a flow sequence
            pending += " " + line;
            if (line.find(']') != std::string::npos) { if (!finish_seq(pending_key, pending)) return false; pending_key.clear(); pending.clear(); }
            continue;
        }
        const std::string t = trim(line);
        if (t.empty() || t == "---" || t == "...") continue;
        const size_t indent = line.find_first_not_of(" \t");
        const size_t colon = t.find(':');
        if (colon == std::string::npos) { err = "line " + std::to_string(lineno) + ": expected `key: value`"; return false; }
        const std::string key = trim(t.substr(0, colon));
        std::string val = trim(t.substr(colon + 1));
        if (indent == 0) { parent.clear(); parent_is_mat = false; }
        else if (parent.empty()) { err = "line " + std::to_string(lineno) + ": indented entry without a parent node"; return false; }
        if (indent == 0 && (val.empty() || val.rfind("!!opencv-matrix", 0) == 0)) {
            parent = key; parent_is_mat = !val.empty();
            if (parent_is_mat) doc.mat[key] = YamlMat();
            continue;
        }
        if (!val.empty() && val[0] == '[') {
            if (val.find(']') == std::string::npos) { pending_key = key; pending = val; continue; }
            if (!finish_seq(key, val)) return false;
            continue;
        }
        val = unquote(val);
        if (parent_is_mat) {
            if (key == "rows") doc.mat[parent].rows = atoi(val.c_str());
            else if (key == "cols") doc.mat[parent].cols = atoi(val.c_str());
            else if (key != "dt") { err = "line " + std::to_string(lineno) + ": unexpected key '" + key + "' in an opencv-matrix"; return false; }
            continue;
        }
        doc.scalar[parent.empty() ? key : parent + "." + key] = val;
    }
    if (!pending_key.empty()) { err = "unterminated `[` for key '" + pending_key + "'"; return false; }
    for (auto& kv : doc.mat)
        if (kv.second.rows * kv.second.cols != (int)kv.second.data.size()) { err = "matrix '" + kv.first + "': rows*cols != number of data entries"; return false; }
    return true;
}

// T (4x4) -> R normalised through Eigen::Quaterniond (parameters.cpp:265-272, :386-392), t
bool take_transform(const YamlDoc& doc, const char* key, double* R, double* t, std::string& err) {
    auto it = doc.mat.find(key);
    if (it == doc.mat.end() || it->second.rows != 4 || it->second.cols != 4) { err = std::string("missing 4x4 matrix '") + key + "'"; return false; }
    const std::vector<double>& d = it->second.data;
    gfd::M3 Rm;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rm.m[3 * i + j] = d[4 * i + j]; t[i] = d[4 * i + 3]; }
    Rm = gfd::qmat(gfd::qnormalized(gfd::rot_to_quat(Rm)));
    memcpy(R, Rm.m, sizeof(Rm.m));
    return true;
}

}  // namespace

extern "C" {

int gf_estimator_cfg_from_yaml(const char* config_file, gf_estimator_cfg* c) {
    if (!config_file || !c) return gf::set_err(GF_ERR_INVALID, "null argument");
    YamlDoc y;
    std::string err;
    if (!parse_yaml(config_file, y, err)) return gf::set_err(GF_ERR_INVALID, "%s: %s", config_file, err.c_str());
    // what the build does not carry fails here, loudly, instead of being silently ignored
    const char* unsupported[] = {"use_line", "use_yolo", "plane", "equalize", "use_motion"};
    for (const char* k : unsupported)
        if (y.integer(k) != 0) return gf::set_err(GF_ERR_INVALID, "%s: `%s: %d` is outside the built path (DESIGN.md, out of scope)", config_file, k, y.integer(k));
    if (y.integer("num_of_cam") != 1) return gf::set_err(GF_ERR_INVALID, "%s: num_of_cam must be 1 (RGB-D), got %d", config_file, y.integer("num_of_cam"));
    memset(c, 0, sizeof(*c));
    c->window_size = 10;          // WINDOW_SIZE, parameters.h:24 (compile-time in the reference)
    c->max_features = 512;        // capacity; NUM_OF_F = 1000 (parameters.h:25) only sizes para_Feature, max_cnt bounds what a window holds
    c->max_visual = 4096;
    c->focal_length = 600.0;      // FOCAL_LENGTH, parameters.h:23
    c->init_depth = 5.0;          // INIT_DEPTH, parameters.cpp:478
    c->use_imu = y.integer("imu"); c->use_wheel = y.integer("wheel"); c->depth = y.integer("depth");
    c->use_mcc = y.integer("use_mcc"); c->wdetect = y.integer("wdetect"); c->stationary_detect = y.integer("stationary_detect");
    c->only_initial_with_wheel = y.integer("only_initial_with_wheel");
    c->depth_threshold = y.integer("depth_threshold");                       // an `int` in the reference, parameters.cpp:172
    c->multiple_thread = y.integer("multiple_thread");
    c->num_iterations = y.integer("max_num_iterations");
    c->max_solver_time = y.real("max_solver_time");                          // SOLVER_TIME, parameters.cpp:343
    c->min_parallax_px = y.real("keyframe_parallax");
    if (c->use_imu) { c->acc_n = y.real("acc_n"); c->acc_w = y.real("acc_w"); c->gyr_n = y.real("gyr_n"); c->gyr_w = y.real("gyr_w"); c->g_norm = y.real("g_norm"); }
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(c->ric, I, sizeof(I)); memcpy(c->rio, I, sizeof(I));
    c->sx = c->sy = c->sw = 1.0;
    if (c->use_wheel) {
        c->wheel_vel_n = y.real("wheel_velocity_noise_sigma"); c->wheel_gyr_n = y.real("wheel_gyro_noise_sigma");
        c->sx = y.real("sx"); c->sy = y.real("sy"); c->sw = y.real("sw");
        c->estimate_wheel_extrinsic = y.integer("estimate_wheel_extrinsic");
        if (c->estimate_wheel_extrinsic == 2) return gf::set_err(GF_ERR_INVALID, "%s: estimate_wheel_extrinsic: 2 (no prior) is not built", config_file);
        if (!take_transform(y, "body_T_wheel", c->rio, c->tio, err)) return gf::set_err(GF_ERR_INVALID, "%s: %s", config_file, err.c_str());
        if (c->estimate_wheel_extrinsic) c->extrinsic_type_wheel = y.integer("extrinsic_type_wheel");   // parameters.cpp:278-306 (read only when the wheel extrinsic is estimated)
        c->estimate_wheel_intrinsic = y.integer("estimate_wheel_intrinsic");
    }
    c->estimate_extrinsic = y.integer("estimate_extrinsic");
    if (c->estimate_extrinsic == 2) return gf::set_err(GF_ERR_INVALID, "%s: estimate_extrinsic: 2 (online calibration) is not built", config_file);
    if (!take_transform(y, "body_T_cam0", c->ric, c->tic, err)) return gf::set_err(GF_ERR_INVALID, "%s: %s", config_file, err.c_str());
    if (c->estimate_extrinsic) c->extrinsic_type = y.integer("extrinsic_type");                          // parameters.cpp:392-420
    c->td = y.real("td"); c->estimate_td = y.integer("estimate_td");
    c->td_wheel = y.real("td_wheel"); c->estimate_td_wheel = y.integer("estimate_td_wheel");
    if (!c->use_imu) { c->estimate_extrinsic = 0; c->estimate_td = 0; }      // parameters.cpp:508-513
    c->gnss_enable = y.integer("gnss_enable") != 0;                           // parameters.cpp:519-552 (the reference reads these keys only when enabled)
    c->max_gnss_per_frame = 32;
    {
        auto it = y.mat.find("gnss_iono_default_parameters");
        const bool have = it != y.mat.end() && it->second.data.size() == 8;
        if (have) memcpy(c->gnss_iono, it->second.data.data(), 64);
        else if (c->gnss_enable) return gf::set_err(GF_ERR_INVALID, "%s: gnss_iono_default_parameters must be a 1 x 8 matrix", config_file);
        if (c->gnss_enable && y.integer("gnss_local_online_sync") != 0)
            return gf::set_err(GF_ERR_INVALID, "%s: gnss_local_online_sync: 1 (trigger-message time sync) is not built; give gnss_local_time_diff", config_file);
        c->gnss_local_time_diff = y.real("gnss_local_time_diff");
        c->gnss_elevation_thres = y.real("gnss_elevation_thres"); c->gnss_ddt_sigma = y.real("gnss_ddt_sigma");
        c->gnss_psr_std_thres = y.real("gnss_psr_std_thres"); c->gnss_dopp_std_thres = y.real("gnss_dopp_std_thres");
        c->gnss_track_num_thres = (int)(unsigned)y.real("gnss_track_num_thres");   // static_cast<uint32_t>(double), parameters.cpp:547-548
        if (c->gnss_enable && !(c->gnss_ddt_sigma > 0)) return gf::set_err(GF_ERR_INVALID, "%s: gnss_ddt_sigma must be positive", config_file);
    }
    // front end
    gf_tracker_cfg& t = c->tracker;
    t.height = y.integer("image_height"); t.width = y.integer("image_width");   // ROW / COL
    t.batch = 1;
    t.max_cnt = y.integer("max_cnt"); t.min_dist = y.integer("min_dist"); t.flow_back = y.integer("flow_back");
    t.depth_cam = c->depth;
    c->with_tracker = 1;
    // cam0_calib, relative to the directory of the config file (parameters.cpp:436-443)
    const std::string cf(config_file);
    const size_t pn = cf.find_last_of('/');
    const std::string cam = (pn == std::string::npos ? std::string(".") : cf.substr(0, pn)) + "/" + y.str("cam0_calib");
    YamlDoc k;
    if (!parse_yaml(cam.c_str(), k, err)) return gf::set_err(GF_ERR_INVALID, "cam0_calib: %s", err.c_str());
    if (k.has("model_type") && k.str("model_type") != "PINHOLE") return gf::set_err(GF_ERR_INVALID, "%s: model_type '%s': only PINHOLE is built", cam.c_str(), k.str("model_type").c_str());
    t.k1 = k.real("distortion_parameters.k1"); t.k2 = k.real("distortion_parameters.k2");
    t.p1 = k.real("distortion_parameters.p1"); t.p2 = k.real("distortion_parameters.p2");
    t.fx = k.real("projection_parameters.fx"); t.fy = k.real("projection_parameters.fy");
    t.cx = k.real("projection_parameters.cx"); t.cy = k.real("projection_parameters.cy");
    if (!(t.fx > 0) || !(t.fy > 0)) return gf::set_err(GF_ERR_INVALID, "%s: projection_parameters.fx / fy missing", cam.c_str());
    return GF_OK;
}

int gf_tum_append(const char* path, double t, const double* P, const double* R) {
    if (!path || !P || !R) return gf::set_err(GF_ERR_INVALID, "null argument");
    gfd::M3 Rm;
    memcpy(Rm.m, R, sizeof(Rm.m));
    const gfd::Q4 q = gfd::rot_to_quat(Rm);      // tmp_Q = Quaterniond(estimator.Rs[WINDOW_SIZE]), visualization.cpp:301-302
    FILE* f = fopen(path, "a");
    if (!f) return gf::set_err(GF_ERR_INVALID, "cannot open %s for appending", path);
    fprintf(f, "%.9f %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n", t, P[0], P[1], P[2], q.x, q.y, q.z, q.w);   // ios::fixed, setprecision(9)
    fclose(f);
    return GF_OK;
}

int gf_pgm_read(const char* path, int* width, int* height, int* maxval, void* pixels, size_t cap_bytes) {
    if (!path || !width || !height || !maxval) return gf::set_err(GF_ERR_INVALID, "null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return gf::set_err(GF_ERR_INVALID, "cannot open %s", path);
    int vals[3], nv = 0, ch;
    char magic[3] = {0, 0, 0};
    if (fread(magic, 1, 2, f) != 2 || magic[0] != 'P' || magic[1] != '5') { fclose(f); return gf::set_err(GF_ERR_INVALID, "%s: not a binary PGM (P5)", path); }
    while (nv < 3 && (ch = fgetc(f)) != EOF) {
        if (ch == '#') { while ((ch = fgetc(f)) != EOF && ch != '\n') {} continue; }
        if (isspace(ch)) continue;
        if (!isdigit(ch)) { fclose(f); return gf::set_err(GF_ERR_INVALID, "%s: malformed PGM header", path); }
        int v = 0;
        while (ch != EOF && isdigit(ch)) {
            v = v * 10 + (ch - '0');
            if (v > (1 << 24)) { fclose(f); return gf::set_err(GF_ERR_INVALID, "%s: malformed PGM header (number too large)", path); }   // no header field of a frame comes near: stop before the int overflows
            ch = fgetc(f);
        }
        vals[nv++] = v;          // the single whitespace byte after maxval has just been consumed
    }
    if (nv < 3 || vals[0] <= 0 || vals[1] <= 0 || vals[2] <= 0 || vals[2] > 65535) { fclose(f); return gf::set_err(GF_ERR_INVALID, "%s: malformed PGM header", path); }
    *width = vals[0]; *height = vals[1]; *maxval = vals[2];
    const size_t bpp = vals[2] > 255 ? 2 : 1, need = (size_t)vals[0] * vals[1] * bpp;
    if (!pixels) { fclose(f); return GF_OK; }     // header query
    if (cap_bytes < need) { fclose(f); return gf::set_err(GF_ERR_CAPACITY, "%s needs %zu bytes, caller gave %zu", path, need, cap_bytes); }
    if (fread(pixels, 1, need, f) != need) { fclose(f); return gf::set_err(GF_ERR_INVALID, "%s: truncated pixel data", path); }
    fclose(f);
    if (bpp == 2) {               // PGM stores 16-bit samples most significant byte first; hand back host order (little endian here)
        unsigned char* p = (unsigned char*)pixels;
        for (size_t i = 0; i < need; i += 2) std::swap(p[i], p[i + 1]);
    }
    return GF_OK;
}

// ---------------------------------------------------------------- ROS bag files (host/rosbag_reader.h behind the C-ABI)
struct gf_bag {
    gf::BagReader reader;
    std::vector<gf::BagMessageRef> sel;
    explicit gf_bag(const char* path) : reader(path) {}
};
#define GF_BAG_TRY(body) try { body } catch (const std::exception& e) { return gf::set_err(GF_ERR_INVALID, "%s", e.what()); }

int gf_bag_open(const char* path, gf_bag** out) {
    if (!path || !out) return gf::set_err(GF_ERR_INVALID, "null argument");
    GF_BAG_TRY(*out = new gf_bag(path);)
    return GF_OK;
}
int gf_bag_close(gf_bag* b) { delete b; return GF_OK; }
int gf_bag_connection_count(gf_bag* b) { return b ? (int)b->reader.connections().size() : 0; }
int gf_bag_connection(gf_bag* b, int i, int* conn_id, char* topic, int topic_cap, char* type, int type_cap) {
    if (!b || i < 0 || i >= (int)b->reader.connections().size()) return gf::set_err(GF_ERR_INVALID, "connection index out of range");
    const gf::BagConnection& c = b->reader.connections()[i];
    if (conn_id) *conn_id = (int)c.id;
    if (topic && topic_cap > 0) snprintf(topic, topic_cap, "%s", c.topic.c_str());
    if (type && type_cap > 0) snprintf(type, type_cap, "%s", c.type.c_str());
    return GF_OK;
}
int gf_bag_select(gf_bag* b, const char* const* topics, int n_topics, long long* count) {
    if (!b || (n_topics > 0 && !topics)) return gf::set_err(GF_ERR_INVALID, "null argument");
    std::vector<std::string> t;
    for (int i = 0; i < n_topics; i++) t.push_back(topics[i] ? topics[i] : "");
    GF_BAG_TRY(b->sel = b->reader.select(t);)
    if (count) *count = (long long)b->sel.size();
    return GF_OK;
}
int gf_bag_message(gf_bag* b, long long i, int* conn_id, double* t_record, const unsigned char** data, size_t* len) {
    if (!b || !data || !len || i < 0 || i >= (long long)b->sel.size()) return gf::set_err(GF_ERR_INVALID, "message index out of range");
    const gf::BagMessageRef& m = b->sel[(size_t)i];
    if (conn_id) *conn_id = (int)m.conn;
    if (t_record) *t_record = (double)(m.time_ns / 1000000000ull) + 1e-9 * (double)(m.time_ns % 1000000000ull);
    GF_BAG_TRY(*data = b->reader.payload(m, len);)
    return GF_OK;
}
int gf_ros_decode_imu(const unsigned char* data, size_t len, double* t, double* acc, double* gyr) {
    if (!data || !t || !acc || !gyr) return gf::set_err(GF_ERR_INVALID, "null argument");
    GF_BAG_TRY(gf::ros_decode_imu(data, len, t, acc, gyr);)
    return GF_OK;
}
int gf_ros_decode_odometry(const unsigned char* data, size_t len, double* t, double* linear, double* angular, double* position) {
    if (!data || !t || !linear || !angular) return gf::set_err(GF_ERR_INVALID, "null argument");
    GF_BAG_TRY(gf::ros_decode_odometry(data, len, t, linear, angular, position);)
    return GF_OK;
}
int gf_ros_decode_image(const unsigned char* data, size_t len, int depth, double* t, int* width, int* height, void* pixels, size_t cap_bytes) {
    if (!data || !t || !width || !height) return gf::set_err(GF_ERR_INVALID, "null argument");
    GF_BAG_TRY(
        const gf::RosImage m = gf::ros_image(data, len);
        *t = m.header.stamp(); *width = (int)m.width; *height = (int)m.height;
        if (!pixels) return GF_OK;
        const size_t need = (size_t)m.width * m.height * (depth ? 2 : 1);
        if (cap_bytes < need) return gf::set_err(GF_ERR_CAPACITY, "image needs %zu bytes, caller gave %zu", need, cap_bytes);
        if (depth) { std::vector<uint16_t> o; gf::ros_image_to_mono16(m, o); memcpy(pixels, o.data(), need); }
        else { std::vector<uint8_t> o; gf::ros_image_to_mono8(m, o); memcpy(pixels, o.data(), need); }
    )
    return GF_OK;
}

}  // extern "C"
