// gf_comm.hip — the multi-GPU exchange of the path without torch: "independent sliding windows shard across the GPUs of one node with an
// RCCL gather of poses over xGMI".  One process per GPU; the only collective is ncclAllGather of the newest pose of every resident window (7 doubles each).
//   gf_comm_*            a communicator from a 128-byte unique id that rank 0 creates and hands to the others out of band (file, environment, socket)
//   gf_pose_gather       gf_ba's newest poses -> persistent device buffer -> ncclAllGather on the caller's communicator and stream
//   gf_comm_allgather_f64  the same collective for host buffers (tools/gf_replay --ranks: final poses of the sequences each rank replayed)
//   gf_numa_* / gf_pin_*   host threads of a rank onto the cores of its GPU's NUMA node (8 ranks x (tracker pool + group workers) on one host)
// RCCL is resolved at run time: symbols already in the process win (a torch process carries its own librccl.so -- two copies in one process do not mix),
// else librccl.so.1 is opened.  Nothing here links against RCCL, so the library loads on machines without it and says so when the collective is asked for.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only
#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/groundfusion_hip.h"
#include "gf_comm.hpp"

namespace gf { int set_err(int code, const char* fmt, ...); }

namespace {
struct Rccl {
    decltype(&ncclGetUniqueId) getUniqueId = nullptr;
    decltype(&ncclCommInitRank) commInitRank = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclAllGather) allGather = nullptr;
    decltype(&ncclGetErrorString) errorString = nullptr;
    bool ok = false;
    std::string why;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // GF_RCCL_LIBRARY names the one library to take RCCL from (a site's own build; tests/test_host_code.py points it at a file that does not exist to
        // walk the "this machine has no RCCL" path); otherwise symbols already in the process come first -- a torch process carries its own librccl
        const char* forced = getenv("GF_RCCL_LIBRARY");
        void* h = (!forced && dlsym(RTLD_DEFAULT, "ncclAllGather")) ? RTLD_DEFAULT : nullptr;
        std::string last_err;
        auto try_open = [&](const char* name) { (void)dlerror(); h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (!h) { const char* e = dlerror(); last_err = e ? e : (std::string(name) + " not found"); } return h != nullptr; };   // dlerror() clears the message it returns: read it once
        if (!h) { if (forced) try_open(forced); else for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if (try_open(name)) break; }
        if (!h) { r.why = std::string("RCCL is not available: ") + (last_err.empty() ? "librccl.so.1 not found" : last_err); return; }
        r.getUniqueId = (decltype(r.getUniqueId))dlsym(h, "ncclGetUniqueId");
        r.commInitRank = (decltype(r.commInitRank))dlsym(h, "ncclCommInitRank");
        r.commDestroy = (decltype(r.commDestroy))dlsym(h, "ncclCommDestroy");
        r.allGather = (decltype(r.allGather))dlsym(h, "ncclAllGather");
        r.errorString = (decltype(r.errorString))dlsym(h, "ncclGetErrorString");
        r.ok = r.getUniqueId && r.commInitRank && r.commDestroy && r.allGather && r.errorString;
        if (!r.ok) r.why = "RCCL library lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather / ncclGetErrorString";
    });
    return r;
}
#define NCCLCHK(expr) do { const ncclResult_t rc_ = (expr); if (rc_ != ncclSuccess) return gf::set_err(GF_ERR_HIP, "%s: %s", #expr, rccl().errorString(rc_)); } while (0)
#define HIPCHK_(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) return gf::set_err(GF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)
}  // namespace

namespace gf {
int rccl_allgather_f64(const double* send, double* recv, size_t count, void* comm, void* stream) {
    Rccl& r = rccl();
    if (!r.ok) return set_err(GF_ERR_NO_DEVICE, "%s", r.why.c_str());
    if (!comm) return set_err(GF_ERR_INVALID, "null communicator");
    NCCLCHK(r.allGather(send, recv, count, ncclDouble, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)));
    return GF_OK;
}
}  // namespace gf

struct gf_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int world = 1, rank = 0, device = 0;
    double* d_send = nullptr; double* d_recv = nullptr; size_t cap = 0;   // staging of gf_comm_allgather_f64 (doubles per rank)
};

extern "C" {

int gf_comm_unique_id(unsigned char* id128) {
    if (!id128) return gf::set_err(GF_ERR_INVALID, "null argument");
    Rccl& r = rccl();
    if (!r.ok) return gf::set_err(GF_ERR_NO_DEVICE, "%s", r.why.c_str());
    ncclUniqueId id;
    NCCLCHK(r.getUniqueId(&id));
    static_assert(sizeof(id) == 128, "gf_comm_unique_id hands out NCCL_UNIQUE_ID_BYTES");
    memcpy(id128, &id, 128);
    return GF_OK;
}

int gf_comm_create(const unsigned char* id128, int world, int rank, int device, gf_comm** out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return gf::set_err(GF_ERR_INVALID, "bad argument (world %d, rank %d)", world, rank);
    Rccl& r = rccl();
    if (!r.ok) return gf::set_err(GF_ERR_NO_DEVICE, "%s", r.why.c_str());
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return gf::set_err(GF_ERR_NO_DEVICE, "no HIP device");
    if (device < 0 || device >= n) return gf::set_err(GF_ERR_INVALID, "device %d outside 0..%d", device, n - 1);
    HIPCHK_(hipSetDevice(device));
    gf_comm* c = new gf_comm;
    c->world = world; c->rank = rank; c->device = device;
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    const ncclResult_t rc = r.commInitRank(&c->comm, world, id, rank);
    if (rc != ncclSuccess) { delete c; return gf::set_err(GF_ERR_HIP, "ncclCommInitRank(world %d, rank %d, device %d): %s", world, rank, device, r.errorString(rc)); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { r.commDestroy(c->comm); delete c; return gf::set_err(GF_ERR_HIP, "hipStreamCreate failed"); }
    *out = c;
    return GF_OK;
}

int gf_comm_destroy(gf_comm* c) {
    if (!c) return GF_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->comm) rccl().commDestroy(c->comm);
    delete c;
    return GF_OK;
}

int gf_comm_info(gf_comm* c, int* world, int* rank, int* device, void** nccl_comm, void** stream) {
    if (!c) return gf::set_err(GF_ERR_INVALID, "null communicator");
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    if (device) *device = c->device;
    if (nccl_comm) *nccl_comm = c->comm;
    if (stream) *stream = c->stream;
    return GF_OK;
}

int gf_comm_allgather_f64(gf_comm* c, const double* send_host, int n, double* recv_host) {
    if (!c || !send_host || !recv_host || n < 1) return gf::set_err(GF_ERR_INVALID, "bad argument");
    HIPCHK_(hipSetDevice(c->device));
    if ((size_t)n > c->cap) {
        if (c->d_send) (void)hipFree(c->d_send);
        if (c->d_recv) (void)hipFree(c->d_recv);
        c->d_send = c->d_recv = nullptr; c->cap = 0;
        HIPCHK_(hipMalloc((void**)&c->d_send, (size_t)n * sizeof(double)));
        HIPCHK_(hipMalloc((void**)&c->d_recv, (size_t)n * c->world * sizeof(double)));
        c->cap = (size_t)n;
    }
    HIPCHK_(hipMemcpyAsync(c->d_send, send_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (int rc = gf::rccl_allgather_f64(c->d_send, c->d_recv, (size_t)n, c->comm, c->stream)) return rc;
    HIPCHK_(hipMemcpyAsync(recv_host, c->d_recv, (size_t)n * c->world * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK_(hipStreamSynchronize(c->stream));
    return GF_OK;
}

// ---------------------------------------------------------------- NUMA placement of a rank's host threads
int gf_numa_node_of_device(int device, int* node, char* cpulist, int cap) {
    if (!node) return gf::set_err(GF_ERR_INVALID, "null argument");
    *node = -1;
    if (cpulist && cap > 0) cpulist[0] = 0;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) return gf::set_err(GF_ERR_NO_DEVICE, "device %d has no PCI bus id", device);
    std::string id(bus);
    for (char& ch : id) ch = (char)tolower(ch);
    std::ifstream f("/sys/bus/pci/devices/" + id + "/numa_node");
    int nd = -1;
    if (f) f >> nd;
    *node = nd;        // -1: the platform reports no affinity (single node, or a virtual function): nothing to pin to
    if (nd >= 0 && cpulist && cap > 0) {
        std::ifstream l("/sys/devices/system/node/node" + std::to_string(nd) + "/cpulist");
        std::string s;
        if (l) std::getline(l, s);
        snprintf(cpulist, cap, "%s", s.c_str());
    }
    return GF_OK;
}
}  // extern "C"

namespace gf {
// "0-23,96-119" -> cpu_set_t
static bool parse_cpulist(const std::string& s, cpu_set_t* set) {
    CPU_ZERO(set);
    bool any = false;
    size_t i = 0;
    while (i < s.size()) {
        char* end = nullptr;
        const long a = strtol(s.c_str() + i, &end, 10);
        if (end == s.c_str() + i) break;
        long b = a;
        i = end - s.c_str();
        if (i < s.size() && s[i] == '-') { b = strtol(s.c_str() + i + 1, &end, 10); i = end - s.c_str(); }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, set); any = true; }
        if (i < s.size() && s[i] == ',') i++;
    }
    return any;
}
// the calling thread onto the cores of `device`'s NUMA node, restricted to what the process may use.  Default: only when several ranks share the host
// (LOCAL_WORLD_SIZE > 1) -- a lone rank is better off with every core of the machine (measured: 256 members on one GPU of a two-node host, 128 worker threads:
// +10 % end to end unpinned); GF_NUMA_PIN=1 / 0 forces it on / off.  Returns the node or -1.
int pin_thread_to_device_node(int device) {
    static const bool on = [] {
        if (const char* e = getenv("GF_NUMA_PIN")) return atoi(e) != 0;
        const char* lw = getenv("LOCAL_WORLD_SIZE");
        return lw && atoi(lw) > 1;
    }();
    if (!on) return -1;
    int node = -1; char list[512];
    if (gf_numa_node_of_device(device, &node, list, sizeof list) != GF_OK || node < 0 || !list[0]) return -1;
    cpu_set_t want, have, both;
    if (!parse_cpulist(list, &want)) return -1;
    if (sched_getaffinity(0, sizeof have, &have) != 0) return -1;
    CPU_AND(&both, &want, &have);
    if (CPU_COUNT(&both) == 0) return -1;       // the container's cpuset lies on another node: keep what we have
    return sched_setaffinity(0, sizeof both, &both) == 0 ? node : -1;
}
}  // namespace gf

extern "C" int gf_pin_thread_to_device_node(int device) { return gf::pin_thread_to_device_node(device); }
