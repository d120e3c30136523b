// Communication layer of the element-partitioned solver: neighbour (halo) exchange
// plans per hierarchy level, all-reduce and all-gather of small device buffers.
//
// Two transports behind one interface:
//   * RCCL (product): one process per GPU, ncclSend/ncclRecv groups over xGMI for the
//     halo, ncclAllReduce / ncclAllGather for the scalars; everything stream-ordered.
//   * Team (tests):   N "ranks" are N host threads of ONE process sharing one GPU;
//     the same collectives are emulated with barriers + device-to-device copies.  It
//     exists so that the N-rank algorithm can be verified on a 1-GPU box; it is never
//     used by bench.py or the drivers.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

namespace sns {

struct Plan {                                    // halo plan of one level (counts in nodes, 4 doubles each)
    std::vector<int> nbr;                        // neighbour ranks, same order on both sides of a link
    std::vector<int32_t> send_ptr, recv_ptr;     // host
    std::vector<int32_t> h_send_idx, h_recv_idx; // host copies (hierarchy setup)
    int32_t *send_idx = nullptr, *recv_idx = nullptr;   // device
    double *send_buf = nullptr, *recv_buf = nullptr;    // device, 4 doubles per node
    int32_t n_send() const { return send_ptr.empty() ? 0 : send_ptr.back(); }
    int32_t n_recv() const { return recv_ptr.empty() ? 0 : recv_ptr.back(); }
};

struct Team {                                    // in-process emulation of a communicator
    explicit Team(int n_) : n(n_), pub_buf(n_, nullptr), pub_plan(n_, nullptr), slots(n_) {}
    int n;
    std::mutex m;
    std::condition_variable cv;
    int count = 0;
    long gen = 0;
    std::vector<const double*> pub_buf;          // published send buffers
    std::vector<const Plan*> pub_plan;
    std::vector<std::vector<double>> slots;      // host staging for reductions / gathers
    void barrier();
};

struct Comm {
    ncclComm_t nccl = nullptr;
    Team* team = nullptr;
    int rank = 0, nranks = 1;
    std::deque<Plan> plans;                      // per hierarchy level; plans[0] = assembled operator (stable references)
    bool active() const { return nccl != nullptr || team != nullptr; }
};

// all return 0 or an SNS_E_* code (error text via set_error)
int plan_upload(Plan& p);                        // h_send_idx/h_recv_idx -> device, allocate buffers
void plan_free(Plan& p);
// fill the ghost entries of x (4 doubles per node) from the owning ranks
int comm_exchange(Comm* c, const Plan& p, double* x, hipStream_t s);
int comm_allreduce_sum(Comm* c, double* buf_dev, int count, hipStream_t s);
// every rank contributes `count` doubles; recv_dev gets nranks*count (rank order)
int comm_allgather(Comm* c, const double* send_dev, double* recv_dev, int count, hipStream_t s);

}  // namespace sns
