// Communication layer of the element-partitioned solver: neighbour (halo) exchange
// plans per hierarchy level, all-reduce and all-gather of small device buffers.
//
// Three transports behind one interface:
//   * RCCL (product): one process per GPU, ncclSend/ncclRecv groups over xGMI for the
//     halo, ncclAllReduce / ncclAllGather for the scalars; everything stream-ordered.
//   * Peer (product, round 4): one process per GPU of ONE node, no library in the data path.
//     Every rank owns a fine-grained "window" of device memory that its peers map through
//     HIP IPC; a collective is one or two small kernels that STORE into the peers' windows
//     over xGMI and raise a sequence flag there, and the receiver's kernel waits for the
//     flags (bounded) and reads its own window.  Made for the latency-bound strong split:
//     a halo exchange is put + wait/unpack (2 launches) instead of pack + a send/recv group
//     + unpack, an all-reduce of a few doubles is ONE single-workgroup launch (measured RCCL
//     launch floors on this image: 49 us per send/recv group, 10 us per all-reduce).
//   * Team (tests):   N "ranks" are N host threads of ONE process sharing one GPU;
//     the same collectives are emulated with barriers + device-to-device copies.  It
//     exists so that the N-rank algorithm can be verified on a 1-GPU box; it is never
//     used by bench.py or the drivers.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "sns_peer_dev.h"

#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

namespace sns {

struct Peer {                                    // one rank's end of the direct transport (sns_peer_create / _connect)
    int rank = 0, nranks = 1, device = 0;
    char* base[PEER_MAX_RANKS] = {};             // windows in THIS address space: base[rank] own, the others IPC mappings
    bool mapped[PEER_MAX_RANKS] = {};
    bool connected = false;
    size_t bytes = 0, bump = 0;                  // window size; allocation cursor of the plan area (own window)
    size_t ag_off = 0, ag_doubles = 0;           // all-gather staging: 2 parities x ag_doubles, same offset on every rank
    unsigned long long ar_seq = 0, ag_seq = 0;
    PeerCtl** d_ctl = nullptr;                   // device array [nranks]: the peers' control areas
    double** d_ag = nullptr;                     // device array [nranks]: the peers' staging areas
    unsigned int* d_done = nullptr;              // device counters [nranks] (last-workgroup detection of the put kernels)
    int* err_host = nullptr;                     // pinned, mapped: != 0 once a wait has timed out
    int* err_dev = nullptr;
    long long timeout_ticks = 0;                 // bound of every device-side wait, in wall_clock64() ticks (100 MHz)
};

struct Plan {                                    // halo plan of one level (counts in nodes, 4 doubles each)
    std::vector<int> nbr;                        // neighbour ranks, same order on both sides of a link
    std::vector<int32_t> send_ptr, recv_ptr;     // host
    std::vector<int32_t> h_send_idx, h_recv_idx; // host copies (hierarchy setup)
    int32_t *send_idx = nullptr, *recv_idx = nullptr;   // device
    double *send_buf = nullptr, *recv_buf = nullptr;    // device, 4 doubles per node
    // peer transport: receive buffers (by parity of seq) and arrival flags live in this rank's window; the device arrays hold,
    // per neighbour k, where this rank's data / flag go in THAT rank's window
    mutable unsigned long long seq = 0;         // (advanced by every exchange, also through the const reference the solver holds)
    double* win_recv[2] = {nullptr, nullptr};
    unsigned long long* win_flag = nullptr;
    int32_t *d_send_ptr = nullptr, *d_recv_ptr = nullptr;
    double** d_put = nullptr;                    // [2][nn] remote payload addresses
    unsigned long long** d_rflag = nullptr;      // [nn] remote flag addresses
    unsigned int* d_done = nullptr;
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
    Peer* peer = nullptr;                        // (not owned: sns_peer_destroy)
    int rank = 0, nranks = 1;
    std::deque<Plan> plans;                      // per hierarchy level; plans[0] = assembled operator (stable references)
    bool active() const { return nccl != nullptr || team != nullptr || peer != nullptr; }
};

// all return 0 or an SNS_E_* code (error text via set_error)
int plan_upload(Plan& p);                        // h_send_idx/h_recv_idx -> device, allocate buffers
void plan_free(Plan& p);
// fill the ghost entries of x (4 doubles per node) from the owning ranks
int comm_exchange(Comm* c, const Plan& p, double* x, hipStream_t s);
int comm_allreduce_sum(Comm* c, double* buf_dev, int count, hipStream_t s);
// every rank contributes `count` doubles; recv_dev gets nranks*count (rank order)
int comm_allgather(Comm* c, const double* send_dev, double* recv_dev, int count, hipStream_t s);

// peer transport
int peer_create(int device, int rank, int nranks, size_t window_bytes, Peer** out, char ipc_handle_out[64]);
int peer_connect(Peer* p, const char* handles /* nranks x 64 bytes, rank order */);
int peer_finish_connect(Peer* p);               // (second half of peer_connect: device tables from base[])
int peer_destroy(Peer* p);
// in-process self-test + latency probe: nranks threads, a ring of halo links; us_out = {exchange, all-reduce, all-gather} per round
int peer_selftest(int device, int nranks, int halo_nodes, int reps, double us_out[3]);
// verified all-reduces / all-gathers between the REAL ranks of a connected communicator (collective)
int peer_check_links(Peer* p, int rounds);
// wire a freshly uploaded plan to the neighbours' windows.  `table_allgather(mine, all)` is the caller's host all-gather of
// 3 * nranks doubles per rank (collective: every rank connects the same plan at the same time).
struct PlanOffers {                              // where rank j writes in MY window: payload parity 0 / 1, flag; -1 = no link
    std::vector<double> mine, all;
};
int peer_plan_offer(Comm* c, Plan& p, PlanOffers& t);
int peer_plan_connect(Comm* c, Plan& p, const PlanOffers& t);
int peer_check(Comm* c);                         // SNS_E_COMM once a device-side wait has timed out
PeerArgs peer_next_allreduce(Peer* p);           // arguments of the next all-reduce round (for a kernel that carries it inside)

}  // namespace sns
