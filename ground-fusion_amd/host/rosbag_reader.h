// rosbag_reader.h — a reader for ROS bag files, format 2.0, without ROS (first item: "rosbag / sensor_msgs::Image -> raw frames").
//
// The reference is only ever exercised through `rosbag play` (README.md:146-187): its node subscribes to IMU_TOPIC / WHEEL_TOPIC / IMAGE0_TOPIC /
// IMAGE1_TOPIC (rosNodeTest.cpp:678-682) and converts the image messages with cv_bridge (getImageFromMsg / getDepthImageFromMsg, rosNodeTest.cpp:238-288).
// This file restates what is needed of that chain:
//   * the bag container (ros_comm rosbag_storage, "#ROSBAG V2.0"): records = <u32 header length><header fields><u32 data length><data>, header fields
//     = <u32 length>name=value; op 0x03 bag header, 0x05 chunk (compression none / bz2 / lz4), 0x07 connection, 0x02 message data, 0x04 index data,
//     0x06 chunk info.  Messages are delivered in the order of their record time (the time rosbag received them), ties in file order: what `rosbag play` does;
//   * chunk compression: bz2 through libbz2 (dlopen: the image ships the library without headers), lz4 = the LZ4 frame format roslz4 writes, decoded here;
//   * ROS 1 message serialisation (little endian, strings / arrays with a u32 length) of std_msgs/Header, sensor_msgs/Imu, nav_msgs/Odometry,
//     sensor_msgs/Image;
//   * cv_bridge::toCvCopy(msg, MONO8) for mono8 / 8UC1 / rgb8 / bgr8 / rgba8 / bgra8 sources (OpenCV 4.2 cvtColor *2GRAY on 8-bit data:
//     (R 4899 + G 9617 + B 1868 + 2^13) >> 14) and the MONO16 relabelling of the depth topic (any 16-bit single-channel payload, byte-swapped when the
//     message says big endian).
// Host code, header only, C++17.  gnss_comm's message types are not decoded: GNSS stays on the CSV side files of host/replay_node.h.
#pragma once
#include <dlfcn.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace gf {

struct BagError : std::runtime_error { using std::runtime_error::runtime_error; };

// bounds-checked little-endian cursor over a byte range
struct BagCursor {
    const uint8_t* p; const uint8_t* end;
    BagCursor(const uint8_t* b, size_t n) : p(b), end(b + n) {}
    size_t left() const { return (size_t)(end - p); }
    void need(size_t n, const char* what) const { if (left() < n) throw BagError(std::string("bag: truncated ") + what); }
    uint8_t u8(const char* w = "u8") { need(1, w); return *p++; }
    uint32_t u32(const char* w = "u32") { need(4, w); uint32_t v; memcpy(&v, p, 4); p += 4; return v; }
    uint64_t u64(const char* w = "u64") { need(8, w); uint64_t v; memcpy(&v, p, 8); p += 8; return v; }
    double f64(const char* w = "f64") { need(8, w); double v; memcpy(&v, p, 8); p += 8; return v; }
    const uint8_t* bytes(size_t n, const char* w = "bytes") { need(n, w); const uint8_t* q = p; p += n; return q; }
    std::string str(const char* w = "string") { const uint32_t n = u32(w); const uint8_t* q = bytes(n, w); return std::string((const char*)q, n); }
    void skip(size_t n, const char* w = "field") { need(n, w); p += n; }
};

struct BagConnection { uint32_t id = 0; std::string topic, type, md5sum; };
struct BagMessageRef {
    uint64_t time_ns = 0;    // record time (rosbag's receipt time): the play order
    uint32_t conn = 0;
    uint32_t chunk = 0;      // index into the reader's chunk list
    uint32_t offset = 0;     // of the message RECORD inside the uncompressed chunk
    uint64_t serial = 0;     // file order (chunk, offset) for ties
};

// ---------------------------------------------------------------- LZ4 (frame format 1.5 as roslz4 writes it; block format as published by lz4.org)
inline size_t lz4_block_decode(const uint8_t* src, size_t n, std::vector<uint8_t>& out) {   // appends to `out`; matches may reach back into earlier blocks (linked blocks)
    const uint8_t* ip = src; const uint8_t* const iend = src + n;
    const size_t start = out.size();
    while (ip < iend) {
        const unsigned token = *ip++;
        size_t lit = token >> 4;
        if (lit == 15) { unsigned b; do { if (ip >= iend) throw BagError("bag: lz4 block truncated (literal length)"); b = *ip++; lit += b; } while (b == 255); }
        if ((size_t)(iend - ip) < lit) throw BagError("bag: lz4 block truncated (literals)");
        out.insert(out.end(), ip, ip + lit);
        ip += lit;
        if (ip >= iend) break;   // the last sequence of a block has no match part
        if (iend - ip < 2) throw BagError("bag: lz4 block truncated (offset)");
        const size_t off = (size_t)ip[0] | ((size_t)ip[1] << 8);
        ip += 2;
        size_t ml = token & 15;
        if (ml == 15) { unsigned b; do { if (ip >= iend) throw BagError("bag: lz4 block truncated (match length)"); b = *ip++; ml += b; } while (b == 255); }
        ml += 4;
        if (off == 0 || off > out.size()) throw BagError("bag: lz4 match offset outside the decoded data");
        const size_t from = out.size() - off;
        for (size_t i = 0; i < ml; i++) out.push_back(out[from + i]);   // byte by byte: overlapping copies are the format's run-length idiom
    }
    return out.size() - start;
}
inline void lz4_frame_decode(const uint8_t* src, size_t n, size_t expect, std::vector<uint8_t>& out) {
    BagCursor c(src, n);
    out.clear(); out.reserve(expect);
    if (c.u32("lz4 magic") != 0x184D2204u) throw BagError("bag: lz4 chunk does not start with the LZ4 frame magic");
    const uint8_t flg = c.u8("lz4 FLG");
    c.u8("lz4 BD");
    if ((flg >> 6) != 1) throw BagError("bag: unsupported LZ4 frame version");
    const bool block_checksum = flg & 0x10, content_size = flg & 0x08, content_checksum = flg & 0x04, dict_id = flg & 0x01;
    if (content_size) c.skip(8, "lz4 content size");
    if (dict_id) c.skip(4, "lz4 dictionary id");
    c.u8("lz4 header checksum");
    for (;;) {
        const uint32_t bs = c.u32("lz4 block size");
        if (bs == 0) break;   // end mark
        const uint32_t len = bs & 0x7fffffffu;
        const uint8_t* data = c.bytes(len, "lz4 block");
        if (bs & 0x80000000u) out.insert(out.end(), data, data + len);   // stored uncompressed
        else lz4_block_decode(data, len, out);
        if (block_checksum) c.skip(4, "lz4 block checksum");
    }
    if (content_checksum) c.skip(4, "lz4 content checksum");
    if (out.size() != expect) throw BagError("bag: lz4 chunk decodes to " + std::to_string(out.size()) + " bytes, its header says " + std::to_string(expect));
}
inline void bz2_decode(const uint8_t* src, size_t n, size_t expect, std::vector<uint8_t>& out) {
    typedef int (*Fn)(char*, unsigned*, char*, unsigned, int, int);
    static Fn fn = [] {
        void* h = nullptr;
        for (const char* name : {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so"}) if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        return h ? (Fn)dlsym(h, "BZ2_bzBuffToBuffDecompress") : (Fn) nullptr;
    }();
    if (!fn) throw BagError("bag: a bz2-compressed chunk needs libbz2.so.1.0 (not found); re-record with `rosbag compress --lz4` or `rosbag decompress`");
    out.resize(expect ? expect : 1);
    unsigned dl = (unsigned)expect;
    const int rc = fn((char*)out.data(), &dl, (char*)const_cast<uint8_t*>(src), (unsigned)n, 0, 0);
    if (rc != 0 || dl != expect) throw BagError("bag: bz2 chunk failed to decompress (BZ2 error " + std::to_string(rc) + ")");
    out.resize(expect);
}

class BagReader {
  public:
    explicit BagReader(const std::string& path) : path_(path) {
        f_ = fopen(path.c_str(), "rb");
        if (!f_) throw BagError("bag: cannot open " + path);
        try { scan(); } catch (...) { fclose(f_); f_ = nullptr; throw; }
    }
    ~BagReader() { if (f_) fclose(f_); }
    BagReader(const BagReader&) = delete;
    BagReader& operator=(const BagReader&) = delete;

    const std::vector<BagConnection>& connections() const { return conns_; }
    const BagConnection* connection(uint32_t id) const { for (auto& c : conns_) if (c.id == id) return &c; return nullptr; }
    size_t chunk_count() const { return chunks_.size(); }

    // messages of the listed topics (all topics when the list is empty) in play order: record time, ties in file order
    std::vector<BagMessageRef> select(const std::vector<std::string>& topics) const {
        std::vector<BagMessageRef> out;
        for (const BagMessageRef& m : msgs_) {
            if (!topics.empty()) {
                const BagConnection* c = connection(m.conn);
                if (!c || std::find(topics.begin(), topics.end(), c->topic) == topics.end()) continue;
            }
            out.push_back(m);
        }
        std::stable_sort(out.begin(), out.end(), [](const BagMessageRef& a, const BagMessageRef& b) { return a.time_ns < b.time_ns || (a.time_ns == b.time_ns && a.serial < b.serial); });
        return out;
    }
    // the serialized message (valid until the next call that touches another chunk)
    const uint8_t* payload(const BagMessageRef& m, size_t* len) {
        const std::vector<uint8_t>& c = chunk_data(m.chunk);
        if (m.offset >= c.size()) throw BagError("bag: message offset outside its chunk");
        BagCursor cur(c.data() + m.offset, c.size() - m.offset);
        const uint32_t hl = cur.u32("record header length");
        cur.skip(hl, "record header");
        const uint32_t dl = cur.u32("record data length");
        const uint8_t* d = cur.bytes(dl, "message data");
        *len = dl;
        return d;
    }

  private:
    struct Chunk { long data_pos = 0; uint32_t stored = 0, size = 0; int compression = 0; bool indexed = false; };   // compression 0 none, 1 bz2, 2 lz4
    std::string path_;
    FILE* f_ = nullptr;
    std::vector<BagConnection> conns_;
    std::vector<Chunk> chunks_;
    std::vector<BagMessageRef> msgs_;
    std::vector<uint8_t> cache_, raw_;
    long cached_ = -1;

    static std::map<std::string, std::string> fields(const uint8_t* h, size_t n) {
        std::map<std::string, std::string> out;
        BagCursor c(h, n);
        while (c.left()) {
            const uint32_t fl = c.u32("header field length");
            const uint8_t* f = c.bytes(fl, "header field");
            const uint8_t* eq = (const uint8_t*)memchr(f, '=', fl);
            if (!eq) throw BagError("bag: header field without '='");
            out[std::string((const char*)f, eq - f)] = std::string((const char*)eq + 1, fl - (eq - f) - 1);
        }
        return out;
    }
    static uint32_t f_u32(const std::map<std::string, std::string>& f, const char* k) {
        auto it = f.find(k);
        if (it == f.end() || it->second.size() != 4) throw BagError(std::string("bag: record without a 4-byte field '") + k + "'");
        uint32_t v; memcpy(&v, it->second.data(), 4); return v;
    }
    static uint64_t f_time(const std::map<std::string, std::string>& f, const char* k) {   // ros::Time: u32 secs, u32 nsecs
        auto it = f.find(k);
        if (it == f.end() || it->second.size() != 8) throw BagError(std::string("bag: record without an 8-byte field '") + k + "'");
        uint32_t s, ns; memcpy(&s, it->second.data(), 4); memcpy(&ns, it->second.data() + 4, 4);
        return (uint64_t)s * 1000000000ull + ns;
    }
    static int f_op(const std::map<std::string, std::string>& f) {
        auto it = f.find("op");
        if (it == f.end() || it->second.size() != 1) throw BagError("bag: record without an op field");
        return (uint8_t)it->second[0];
    }
    void add_connection(const std::map<std::string, std::string>& hf, const uint8_t* data, size_t n) {
        const uint32_t id = f_u32(hf, "conn");
        if (connection(id)) return;
        BagConnection c; c.id = id;
        auto t = hf.find("topic");
        if (t != hf.end()) c.topic = t->second;
        const auto df = fields(data, n);   // the connection header
        auto g = [&](const char* k) { auto it = df.find(k); return it == df.end() ? std::string() : it->second; };
        if (c.topic.empty()) c.topic = g("topic");
        c.type = g("type"); c.md5sum = g("md5sum");
        conns_.push_back(c);
    }
    bool read_exact(void* dst, size_t n) { return fread(dst, 1, n, f_) == n; }

    void scan() {
        char magic[13];
        if (!read_exact(magic, 13) || memcmp(magic, "#ROSBAG V2.0\n", 13) != 0) throw BagError("bag: " + path_ + " is not a ROS bag of format 2.0");
        std::vector<uint8_t> hdr, data;
        fseek(f_, 0, SEEK_END);
        const long file_size = ftell(f_);
        fseek(f_, 13, SEEK_SET);
        for (;;) {
            uint32_t hl;
            if (!read_exact(&hl, 4)) break;   // end of file
            hdr.resize(hl);
            if (hl && !read_exact(hdr.data(), hl)) throw BagError("bag: truncated record header");
            uint32_t dl;
            if (!read_exact(&dl, 4)) throw BagError("bag: truncated record (data length)");
            const long data_pos = ftell(f_);
            if (data_pos + (long)dl > file_size) throw BagError("bag: truncated record (its data runs past the end of the file)");   // fseek past the end would succeed
            const auto hf = fields(hdr.data(), hl);
            const int op = f_op(hf);
            if (op == 0x05) {   // chunk: remembered, not read
                Chunk c; c.data_pos = data_pos; c.stored = dl; c.size = f_u32(hf, "size");
                auto it = hf.find("compression");
                const std::string comp = it == hf.end() ? "none" : it->second;
                c.compression = comp == "none" ? 0 : comp == "bz2" ? 1 : comp == "lz4" ? 2 : -1;
                if (c.compression < 0) throw BagError("bag: unknown chunk compression '" + comp + "'");
                chunks_.push_back(c);
                if (fseek(f_, (long)dl, SEEK_CUR) != 0) throw BagError("bag: truncated chunk");
            } else if (op == 0x04 || op == 0x07) {   // index data of the last chunk / a connection record behind the chunks
                data.resize(dl);
                if (dl && !read_exact(data.data(), dl)) throw BagError("bag: truncated record data");
                if (op == 0x07) add_connection(hf, data.data(), dl);
                else {
                    if (chunks_.empty()) throw BagError("bag: index data before any chunk");
                    if (f_u32(hf, "ver") != 1) throw BagError("bag: unsupported index data version");
                    const uint32_t conn = f_u32(hf, "conn"), count = f_u32(hf, "count");
                    BagCursor c(data.data(), dl);
                    for (uint32_t i = 0; i < count; i++) {
                        BagMessageRef m;
                        const uint32_t s = c.u32("index time"), ns = c.u32("index time");
                        m.time_ns = (uint64_t)s * 1000000000ull + ns; m.conn = conn; m.chunk = (uint32_t)(chunks_.size() - 1); m.offset = c.u32("index offset");
                        m.serial = ((uint64_t)m.chunk << 32) | m.offset;
                        msgs_.push_back(m);
                    }
                    chunks_.back().indexed = true;
                }
            } else {   // bag header, chunk info, anything newer: skipped
                if (fseek(f_, (long)dl, SEEK_CUR) != 0) throw BagError("bag: truncated record");
            }
        }
        // chunks without index records (a bag that was not closed properly) and connections only the chunks mention: read the chunks themselves
        bool need_conn = false;
        for (const BagMessageRef& m : msgs_) if (!connection(m.conn)) { need_conn = true; break; }
        for (size_t ci = 0; ci < chunks_.size(); ci++) if (!chunks_[ci].indexed || need_conn) scan_chunk((uint32_t)ci, !chunks_[ci].indexed);
    }
    void scan_chunk(uint32_t ci, bool take_messages) {
        const std::vector<uint8_t>& c = chunk_data(ci);
        BagCursor cur(c.data(), c.size());
        while (cur.left()) {
            const uint32_t off = (uint32_t)(cur.p - c.data());
            const uint32_t hl = cur.u32("record header length");
            const uint8_t* h = cur.bytes(hl, "record header");
            const uint32_t dl = cur.u32("record data length");
            const uint8_t* d = cur.bytes(dl, "record data");
            const auto hf = fields(h, hl);
            const int op = f_op(hf);
            if (op == 0x07) add_connection(hf, d, dl);
            else if (op == 0x02 && take_messages) {
                BagMessageRef m; m.time_ns = f_time(hf, "time"); m.conn = f_u32(hf, "conn"); m.chunk = ci; m.offset = off; m.serial = ((uint64_t)ci << 32) | off;
                msgs_.push_back(m);
            }
        }
    }
    const std::vector<uint8_t>& chunk_data(uint32_t ci) {
        if (cached_ == (long)ci) return cache_;
        if (ci >= chunks_.size()) throw BagError("bag: chunk index out of range");
        const Chunk& c = chunks_[ci];
        raw_.resize(c.stored);
        if (fseek(f_, c.data_pos, SEEK_SET) != 0 || (c.stored && !read_exact(raw_.data(), c.stored))) throw BagError("bag: cannot read chunk " + std::to_string(ci));
        cached_ = -1;
        if (c.compression == 0) { if (c.stored != c.size) throw BagError("bag: uncompressed chunk whose stored and declared sizes differ"); cache_.swap(raw_); }
        else if (c.compression == 1) bz2_decode(raw_.data(), raw_.size(), c.size, cache_);
        else lz4_frame_decode(raw_.data(), raw_.size(), c.size, cache_);
        cached_ = (long)ci;
        return cache_;
    }
};

// ---------------------------------------------------------------- ROS 1 messages the node subscribes to
struct RosHeader { uint32_t seq = 0, sec = 0, nsec = 0; std::string frame_id; double stamp() const { return (double)sec + 1e-9 * (double)nsec; } };   // ros::Time::toSec
inline RosHeader ros_header(BagCursor& c) { RosHeader h; h.seq = c.u32("Header.seq"); h.sec = c.u32("Header.stamp"); h.nsec = c.u32("Header.stamp"); h.frame_id = c.str("Header.frame_id"); return h; }

// sensor_msgs/Imu -> what imu_callback reads (rosNodeTest.cpp:567-585): header.stamp, linear_acceleration, angular_velocity
inline void ros_decode_imu(const uint8_t* d, size_t n, double* t, double acc[3], double gyr[3]) {
    BagCursor c(d, n);
    const RosHeader h = ros_header(c);
    c.skip(8 * (4 + 9), "Imu.orientation");
    for (int k = 0; k < 3; k++) gyr[k] = c.f64("Imu.angular_velocity");
    c.skip(8 * 9, "Imu.angular_velocity_covariance");
    for (int k = 0; k < 3; k++) acc[k] = c.f64("Imu.linear_acceleration");
    c.skip(8 * 9, "Imu.linear_acceleration_covariance");
    *t = h.stamp();
}
// nav_msgs/Odometry -> what wheel_callback reads (rosNodeTest.cpp:81-189): header.stamp, twist.twist.linear / angular (pose.pose.position for the raw-odometry display)
inline void ros_decode_odometry(const uint8_t* d, size_t n, double* t, double lin[3], double ang[3], double pos[3]) {
    BagCursor c(d, n);
    const RosHeader h = ros_header(c);
    c.str("Odometry.child_frame_id");
    for (int k = 0; k < 3; k++) { const double v = c.f64("Odometry.pose.position"); if (pos) pos[k] = v; }
    c.skip(8 * (4 + 36), "Odometry.pose");
    for (int k = 0; k < 3; k++) lin[k] = c.f64("Odometry.twist.linear");
    for (int k = 0; k < 3; k++) ang[k] = c.f64("Odometry.twist.angular");
    c.skip(8 * 36, "Odometry.twist.covariance");
    *t = h.stamp();
}
struct RosImage { RosHeader header; uint32_t height = 0, width = 0, step = 0; std::string encoding; uint8_t is_bigendian = 0; const uint8_t* data = nullptr; uint32_t size = 0; };
inline RosImage ros_image(const uint8_t* d, size_t n) {
    BagCursor c(d, n);
    RosImage m;
    m.header = ros_header(c);
    m.height = c.u32("Image.height"); m.width = c.u32("Image.width"); m.encoding = c.str("Image.encoding"); m.is_bigendian = c.u8("Image.is_bigendian"); m.step = c.u32("Image.step");
    m.size = c.u32("Image.data"); m.data = c.bytes(m.size, "Image.data");
    if ((uint64_t)m.step * m.height > m.size) throw BagError("bag: Image.data shorter than step * height");
    return m;
}
// getImageFromMsg (rosNodeTest.cpp:238-263): 8UC1 is relabelled mono8, everything else goes through cv_bridge::toCvCopy(msg, MONO8).  EQUALIZE (CLAHE) is outside the built path.
inline void ros_image_to_mono8(const RosImage& m, std::vector<uint8_t>& out) {
    const std::string& e = m.encoding;
    int ch = 0, r = 0, g = 1, b = 2;
    if (e == "mono8" || e == "8UC1") ch = 1;
    else if (e == "rgb8") ch = 3;
    else if (e == "bgr8") { ch = 3; r = 2; b = 0; }
    else if (e == "rgba8") ch = 4;
    else if (e == "bgra8") { ch = 4; r = 2; b = 0; }
    else throw BagError("bag: image encoding '" + e + "' -> MONO8 is not built (mono8, 8UC1, rgb8, bgr8, rgba8, bgra8 are)");
    if ((uint64_t)m.width * ch > m.step) throw BagError("bag: Image.step shorter than a row");
    out.resize((size_t)m.width * m.height);
    for (uint32_t y = 0; y < m.height; y++) {
        const uint8_t* s = m.data + (size_t)y * m.step;
        uint8_t* o = out.data() + (size_t)y * m.width;
        if (ch == 1) memcpy(o, s, m.width);
        else for (uint32_t x = 0; x < m.width; x++, s += ch)   // OpenCV 4.2 color_rgb.cpp RGB2Gray<uchar>: CV_DESCALE(b B2Y + g G2Y + r R2Y, 14), B2Y 1868, G2Y 9617, R2Y 4899
            o[x] = (uint8_t)((s[b] * 1868 + s[g] * 9617 + s[r] * 4899 + (1 << 13)) >> 14);
    }
}
// getDepthImageFromMsg (rosNodeTest.cpp:265-286): the payload is relabelled MONO16 whatever the message says: two bytes per pixel, in the message's byte order
inline void ros_image_to_mono16(const RosImage& m, std::vector<uint16_t>& out) {
    if ((uint64_t)m.width * 2 > m.step) throw BagError("bag: depth image '" + m.encoding + "' has fewer than two bytes per pixel (the node reads it as MONO16)");
    out.resize((size_t)m.width * m.height);
    for (uint32_t y = 0; y < m.height; y++) {
        const uint8_t* s = m.data + (size_t)y * m.step;
        uint16_t* o = out.data() + (size_t)y * m.width;
        for (uint32_t x = 0; x < m.width; x++) o[x] = m.is_bigendian ? (uint16_t)((s[2 * x] << 8) | s[2 * x + 1]) : (uint16_t)(s[2 * x] | (s[2 * x + 1] << 8));
    }
}

}  // namespace gf
