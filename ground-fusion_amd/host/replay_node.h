// replay_node.h — the ROS node of the reference without ROS: the callbacks of vins_estimator/src/rosNodeTest.cpp
// driven from recorded messages in time-stamp order, the way `rosbag play` delivers them.
//   imu_callback   rosNodeTest.cpp:567-585     wheel_callback  :81-189 (the w_replace yaw-rate substitution included)
//   img0_callback / img1_callback :59-71       sync_process    :290-470 (RGB / depth pairing within 3 ms, the non-YOLO branch)
// Est is gf::Estimator (estimator.h) in the tool; the unit test substitutes a recorder.  Frames are read from binary PGM files
// (8-bit gray = MONO8, 16-bit = MONO16 depth in mm) when their pair is formed, not when their message is queued.
// run_bag() drives the same callbacks from a ROS bag (format 2.0, host/rosbag_reader.h) the way `rosbag play` would: messages of the four topics in record-time
// order, sensor_msgs/Image converted as getImageFromMsg / getDepthImageFromMsg do (rosNodeTest.cpp:238-288) when their pair is formed.
//
// Dataset directory read by load():   imu.csv  t,ax,ay,az,gx,gy,gz     wheel.csv  t,vx,vy,vz,wx,wy,wz     (nav_msgs/Odometry twist)
//                                     image0.csv  t,file               image1.csv  t,file                 (files relative to the directory)
// optional (gnss_enable: 1):          gnss.csv   t_msg,sat,sys,time,psr,dopp,psr_std,dopp_std,wavelength,sx,sy,sz,vx,vy,vz,svdt,svddt,tgd,pr_uura,dp_uura,tow
//                                                one row per L1 observation (gf_gnss_obs); consecutive rows with the same t_msg are one GnssMeasMsg
//                                                (gnss_meas_callback, rosNodeTest.cpp:213-236), delivered at local time t_msg - gnss_local_time_diff
//                                     gnss_align.csv  t,ax,ay,az,yaw,dt0,dt1,dt2,dt3,ddt   a GNSS-VI alignment result on offer from local time t on
#pragma once
#include <algorithm>
#include <cstdio>
#include <deque>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "estimator.h"
#include "rosbag_reader.h"

namespace gf {

struct ImuMsg { double t; Vec3 linear_acceleration, angular_velocity; };
struct OdomMsg { double t; Vec3 linear, angular; };
struct ImageMsg { double t; std::string file; BagMessageRef ref{}; bool in_bag = false; };
struct GnssMsg { double t; std::vector<gf_gnss_obs> meas; };
struct GnssAlignMsg { double t; Vec3 anc; double yaw, dt[4], ddt; };

template <class Est> class ReplayNode {
  public:
    Est& estimator;
    int w_replace = 0;                 // config key `w_replace` (parameters.cpp:178): take the wheel yaw rate from the IMU's -y gyro axis
    std::deque<ImuMsg> imu_buf;
    std::deque<ImageMsg> img0_buf, img1_buf;
    long n_pairs = 0, n_thrown0 = 0, n_thrown1 = 0, n_gnss = 0;
    double gnss_local_time_diff = 0;   // config key `gnss_local_time_diff`: GNSS messages are merged at their local time

    explicit ReplayNode(Est& e) : estimator(e) {}

    void imu_callback(const ImuMsg& m) {
        imu_buf.push_back(m);
        estimator.inputIMU(m.t, m.linear_acceleration, m.angular_velocity);
    }

    void wheel_callback(const OdomMsg& m) {
        const double t = m.t;
        double rz_f = 0, rz_b = 0, t_f = 0, t_b = 0, final_z = 0;
        while (!imu_buf.empty()) {
            const int Bg_z = 0;        // `int Bg_z = 0.000000001;` in the reference, i.e. 0
            double t_imu = imu_buf.front().t;
            if (t - t_imu < 0.006 && t - t_imu > 0) {
                rz_f = -imu_buf.front().angular_velocity[1];
                t_f = t_imu;
                imu_buf.pop_front();
                if (!imu_buf.empty()) {
                    rz_b = -imu_buf.front().angular_velocity[1];
                    t_imu = imu_buf.front().t;
                    t_b = t_imu;
                    imu_buf.pop_front();
                    final_z = rz_f + (rz_b - rz_f) / (t_b - t_f) * (t - t_f) - Bg_z;
                }
                break;
            }
            imu_buf.pop_front();
        }
        Vec3 gyr = m.angular;
        if (final_z != 0 && w_replace) gyr[2] = final_z;
        estimator.inputWheel(t, m.linear, gyr);
    }

    void gnss_meas_callback(const GnssMsg& m) { estimator.inputGNSS(m.t, m.meas); n_gnss++; }           // rosNodeTest.cpp:213-236 (time_diff_valid: offline value)
    void gnss_align_callback(const GnssAlignMsg& m) { estimator.setGNSSAlignment(m.anc, m.yaw, m.dt, m.ddt); }
    void img0_callback(const ImageMsg& m) { img0_buf.push_back(m); }
    void img1_callback(const ImageMsg& m) { img1_buf.push_back(m); }

    // one polling pass of the sync thread, repeated until it finds nothing to do
    void sync_process(const std::string& dir) {
        while (!img0_buf.empty() && !img1_buf.empty()) {
            const double time0 = img0_buf.front().t, time1 = img1_buf.front().t;
            if (time0 < time1 - 0.003) { img0_buf.pop_front(); n_thrown0++; continue; }
            if (time0 > time1 + 0.003) { img1_buf.pop_front(); n_thrown1++; continue; }
            const double time = time0;
            load_frame(dir, img0_buf.front(), false);
            load_frame(dir, img1_buf.front(), true);
            img0_buf.pop_front(); img1_buf.pop_front();
            if (gw_ != dw_ || gh_ != dh_) throw std::runtime_error("replay: gray and depth frames differ in size");
            GrayImage g; g.data = gray_.data(); g.rows = gh_; g.cols = gw_; g.stride = gw_;
            DepthImage d; d.data = (const uint16_t*)depth_.data(); d.rows = dh_; d.cols = dw_; d.stride = dw_;
            estimator.inputImage(time, g, d);
            n_pairs++;
        }
    }

    // merge the recorded topics by time stamp (ties: GNSS alignment, GNSS, imu, wheel, image0, image1) and run the callbacks
    void run(const std::string& dir) {
        struct Ev { double t; int kind; size_t idx; };
        std::vector<ImuMsg> imu; std::vector<OdomMsg> odom; std::vector<ImageMsg> im0, im1;
        load(dir, imu, odom, im0, im1);
        std::vector<GnssMsg> gn; std::vector<GnssAlignMsg> al;
        load_gnss(dir, gn, al);
        std::vector<Ev> ev;
        for (size_t i = 0; i < al.size(); i++) ev.push_back({al[i].t, -2, i});
        for (size_t i = 0; i < gn.size(); i++) ev.push_back({gn[i].t - gnss_local_time_diff, -1, i});
        for (size_t i = 0; i < imu.size(); i++) ev.push_back({imu[i].t, 0, i});
        for (size_t i = 0; i < odom.size(); i++) ev.push_back({odom[i].t, 1, i});
        for (size_t i = 0; i < im0.size(); i++) ev.push_back({im0[i].t, 2, i});
        for (size_t i = 0; i < im1.size(); i++) ev.push_back({im1[i].t, 3, i});
        std::stable_sort(ev.begin(), ev.end(), [](const Ev& a, const Ev& b) { return a.t < b.t || (a.t == b.t && a.kind < b.kind); });
        for (const Ev& e : ev) {
            if (e.kind == -2) gnss_align_callback(al[e.idx]);
            else if (e.kind == -1) gnss_meas_callback(gn[e.idx]);
            else if (e.kind == 0) imu_callback(imu[e.idx]);
            else if (e.kind == 1) wheel_callback(odom[e.idx]);
            else if (e.kind == 2) { img0_callback(im0[e.idx]); sync_process(dir); }
            else { img1_callback(im1[e.idx]); sync_process(dir); }
        }
    }
    // `rosbag play <bag>` into the four subscribers of rosNodeTest.cpp:678-682: messages in record-time order (ties in file order); the callbacks read the
    // header stamps, as the node does.  GNSS topics carry gnss_comm message types and are not read from the bag.
    void run_bag(const std::string& bag_path, const std::string& imu_topic, const std::string& wheel_topic, const std::string& image0_topic, const std::string& image1_topic) {
        BagReader bag(bag_path);
        bag_ = &bag;
        std::vector<std::string> topics;
        for (const std::string& t : {imu_topic, wheel_topic, image0_topic, image1_topic}) if (!t.empty()) topics.push_back(t);
        std::map<uint32_t, int> kind;   // connection id -> 0 imu, 1 wheel, 2 image0, 3 image1
        for (const BagConnection& c : bag.connections()) {
            const int k = c.topic == imu_topic ? 0 : c.topic == wheel_topic ? 1 : c.topic == image0_topic ? 2 : c.topic == image1_topic ? 3 : -1;
            if (k < 0) continue;
            const char* want = k == 0 ? "sensor_msgs/Imu" : k == 1 ? "nav_msgs/Odometry" : "sensor_msgs/Image";
            if (c.type != want) throw std::runtime_error("replay: topic " + c.topic + " carries " + c.type + ", the node subscribes it as " + want);
            kind[c.id] = k;
        }
        for (const std::string& t : topics) {
            bool found = false;
            for (const BagConnection& c : bag.connections()) found |= c.topic == t;
            if (!found) fprintf(stderr, "replay: topic %s is not in %s\n", t.c_str(), bag_path.c_str());
        }
        try {
            for (const BagMessageRef& m : bag.select(topics)) {
                const int k = kind.at(m.conn);
                size_t len = 0;
                const uint8_t* d = bag.payload(m, &len);
                if (k == 0) { ImuMsg im; double a[3], g[3]; ros_decode_imu(d, len, &im.t, a, g); for (int q = 0; q < 3; q++) { im.linear_acceleration[q] = a[q]; im.angular_velocity[q] = g[q]; } imu_callback(im); }
                else if (k == 1) { OdomMsg om; double l[3], a[3]; ros_decode_odometry(d, len, &om.t, l, a, nullptr); for (int q = 0; q < 3; q++) { om.linear[q] = l[q]; om.angular[q] = a[q]; } wheel_callback(om); }
                else {
                    BagCursor c(d, len);
                    ImageMsg im; im.t = ros_header(c).stamp(); im.ref = m; im.in_bag = true;   // the pixels stay in the bag until the pair is formed
                    if (k == 2) img0_callback(im); else img1_callback(im);
                    sync_process("");
                }
            }
        } catch (...) { bag_ = nullptr; throw; }
        bag_ = nullptr;
    }
    // gnss.csv / gnss_align.csv are optional: a dataset without them replays as before
    static void load_gnss(const std::string& dir, std::vector<GnssMsg>& gn, std::vector<GnssAlignMsg>& al) {
        if (std::ifstream(dir + "/gnss.csv"))
            for (auto& r : rows(dir + "/gnss.csv", 21)) {
                gf_gnss_obs o{};
                const double t = num(r[0]);
                o.sat = (int)num(r[1]); o.sys = (int)num(r[2]); o.time = num(r[3]); o.psr = num(r[4]); o.dopp = num(r[5]); o.psr_std = num(r[6]); o.dopp_std = num(r[7]);
                o.wavelength = num(r[8]);
                for (int k = 0; k < 3; k++) { o.sv_pos[k] = num(r[9 + k]); o.sv_vel[k] = num(r[12 + k]); }
                o.svdt = num(r[15]); o.svddt = num(r[16]); o.tgd = num(r[17]); o.pr_uura = num(r[18]); o.dp_uura = num(r[19]); o.tow = num(r[20]);
                if (gn.empty() || gn.back().t != t) gn.push_back({t, {}});
                gn.back().meas.push_back(o);
            }
        if (std::ifstream(dir + "/gnss_align.csv"))
            for (auto& r : rows(dir + "/gnss_align.csv", 10)) {
                GnssAlignMsg m; m.t = num(r[0]); m.anc = vec(r, 1); m.yaw = num(r[4]);
                for (int k = 0; k < 4; k++) m.dt[k] = num(r[5 + k]);
                m.ddt = num(r[9]);
                al.push_back(m);
            }
    }

    static void load(const std::string& dir, std::vector<ImuMsg>& imu, std::vector<OdomMsg>& odom, std::vector<ImageMsg>& im0, std::vector<ImageMsg>& im1) {
        for (auto& r : rows(dir + "/imu.csv", 7)) imu.push_back({num(r[0]), vec(r, 1), vec(r, 4)});
        for (auto& r : rows(dir + "/wheel.csv", 7)) odom.push_back({num(r[0]), vec(r, 1), vec(r, 4)});
        for (auto& r : rows(dir + "/image0.csv", 2)) im0.push_back({num(r[0]), r[1]});
        for (auto& r : rows(dir + "/image1.csv", 2)) im1.push_back({num(r[0]), r[1]});
    }

  private:
    std::vector<uint8_t> gray_, depth_;
    int gw_ = 0, gh_ = 0, dw_ = 0, dh_ = 0;
    BagReader* bag_ = nullptr;
    std::vector<uint16_t> depth16_;

    void load_frame(const std::string& dir, const ImageMsg& m, bool depth) {
        if (!m.in_bag) { if (depth) load_pgm(dir + "/" + m.file, depth_, sizeof(uint16_t), dw_, dh_); else load_pgm(dir + "/" + m.file, gray_, sizeof(uint8_t), gw_, gh_); return; }
        size_t len = 0;
        const uint8_t* d = bag_->payload(m.ref, &len);
        const RosImage im = ros_image(d, len);
        if (depth) { ros_image_to_mono16(im, depth16_); depth_.resize(depth16_.size() * 2); memcpy(depth_.data(), depth16_.data(), depth_.size()); dw_ = (int)im.width; dh_ = (int)im.height; }
        else { ros_image_to_mono8(im, gray_); gw_ = (int)im.width; gh_ = (int)im.height; }
    }

    static double num(const std::string& s) {
        char* end = nullptr;
        const double v = strtod(s.c_str(), &end);
        if (end == s.c_str()) throw std::runtime_error("replay: bad number '" + s + "'");
        return v;
    }
    static Vec3 vec(const std::vector<std::string>& r, int o) { Vec3 v; v[0] = num(r[o]); v[1] = num(r[o + 1]); v[2] = num(r[o + 2]); return v; }
    static std::vector<std::vector<std::string>> rows(const std::string& path, size_t ncol) {
        std::ifstream in(path);
        if (!in) throw std::runtime_error("replay: cannot open " + path);
        std::vector<std::vector<std::string>> out;
        std::string line;
        while (std::getline(in, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty() || line[0] == '#') continue;
            std::vector<std::string> f;
            std::stringstream ss(line);
            std::string tok;
            while (std::getline(ss, tok, ',')) f.push_back(tok);
            if (f.size() != ncol) throw std::runtime_error("replay: " + path + ": expected " + std::to_string(ncol) + " comma-separated fields: " + line);
            out.push_back(f);
        }
        return out;
    }
    static void load_pgm(const std::string& path, std::vector<uint8_t>& buf, size_t bpp, int& w, int& h) {
        int mv = 0;
        if (gf_pgm_read(path.c_str(), &w, &h, &mv, nullptr, 0) != GF_OK) throw std::runtime_error(std::string("replay: ") + gf_last_error());
        if ((mv > 255 ? 2u : 1u) != bpp) throw std::runtime_error("replay: " + path + ": expected a " + std::to_string(8 * bpp) + "-bit PGM");
        buf.resize((size_t)w * h * bpp);
        if (gf_pgm_read(path.c_str(), &w, &h, &mv, buf.data(), buf.size()) != GF_OK) throw std::runtime_error(std::string("replay: ") + gf_last_error());
    }
};

}  // namespace gf
