// gf_comm.hpp — what the other translation units use of gf_comm.hip
#pragma once
#include <cstddef>
namespace gf {
// ncclAllGather of `count` doubles per rank on the caller's communicator (ncclComm_t) and stream (hipStream_t); RCCL resolved at run time
int rccl_allgather_f64(const double* send, double* recv, size_t count, void* comm, void* stream);
// the calling thread onto the cores of the device's NUMA node (GF_NUMA_PIN=0: off); returns the node or -1
int pin_thread_to_device_node(int device);
}
