// Communication layer of the element-partitioned solver: neighbour (halo) exchange
// plans per hierarchy level, all-reduce and all-gather of small device buffers.
//
// Three transports behind one interface:
//   * RCCL (product): one process per GPU, ncclSend/ncclRecv groups over xGMI for the
//     halo, ncclAllReduce / ncclAllGather for the scalars; everything stream-ordered.
//   * Peer (product, round 4): one process per GPU of ONE node, no library in the data path.
//     Every rank owns a fine-grained "window" of device memory that its peers map through
//     HIP IPC; a collective is one or two small kernels that STORE into the peers' windows
//     over xGMI and raise a sequence flag there, and the receiver waits for the flags
//     (bounded) and reads its own window.  Made for the latency-bound strong split.
//     Round 5: a halo exchange is ONE launch (k_halo_put) -- the pass that consumes the
//     halo waits for the flags itself and reads the ghost entries straight from the
//     window (GhostSrc, sns_peer_dev.h); an all-reduce of a few doubles rides inside the
//     launch that finishes the local reduction; sequence numbers live in device memory,
//     so a whole solver iteration is capturable as a hipGraph.
//   * Team (tests + the kernel-floor measurement of the partitioned path): N "ranks" are
//     N host threads of ONE process sharing one GPU.  Since round 5 it runs the PEER
//     transport's kernels on windows wired inside the one address space; only the device
//     flag waits are replaced by a stream synchronisation + host barrier (the ranks'
//     kernels are serialised on one queue and could not wait for each other).  Never used
//     by bench.py or the drivers.
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
    bool host_sync = false;                      // team transport: no device flags, the host barrier orders the rounds
    size_t bytes = 0, bump = 0, bump0 = 0;       // window size; allocation cursor of the plan area (own window) and its start
    size_t flag_off = 0, flag_bump = 0, flag_end = 0;   // the plans' arrival flags live in an area of their own: a reused plan area may
                                                 // hold stale PAYLOAD, and payload bits must never be read as a round number
    int plans_live = 0;                          // plans carved from the window and not yet released (bump resets at 0)
    long long epoch = 0;                         // a plan's rounds start at epoch << 32; raised above every rank's at each connect
    size_t ag_off = 0, ag_doubles = 0;           // all-gather staging: 2 parities x ag_doubles, same offset on every rank
    unsigned long long* d_seq = nullptr;         // device words: [0] all-reduce rounds done, [1] all-gather rounds done
    PeerCtl** d_ctl = nullptr;                   // device array [nranks]: the peers' control areas
    double** d_ag = nullptr;                     // device array [nranks]: the peers' staging areas
    unsigned int* d_done = nullptr;              // device counters (last-workgroup detection of the all-gather put)
    int* err_host = nullptr;                     // pinned, mapped: != 0 once a wait has timed out
    int* err_dev = nullptr;
    long long timeout_ticks = 0;                 // bound of every device-side wait, in wall_clock64() ticks (100 MHz)
};

struct Plan {                                    // halo plan of one level (counts in nodes, 4 doubles each)
    std::vector<int> nbr;                        // neighbour ranks, same order on both sides of a link
    std::vector<int32_t> send_ptr, recv_ptr;     // host
    std::vector<int32_t> h_send_idx, h_recv_idx; // host copies (hierarchy setup)
    int32_t n_own = 0;                           // owned nodes of the level (the ghost nodes follow them)
    bool identity_recv = false;                  // recv_idx[q] == n_own + q: the receive buffer IS the ghost tail, in order
    int32_t *send_idx = nullptr, *recv_idx = nullptr;   // device
    double *send_buf = nullptr, *recv_buf = nullptr;    // device, 4 doubles per node (RCCL staging)
    // peer / team transport: receive buffers (by parity of the round) and arrival flags live in this rank's window; the device
    // arrays hold, per neighbour k, where this rank's data / flag go in THAT rank's window
    double* win_recv[2] = {nullptr, nullptr};
    unsigned long long* win_flag = nullptr;
    unsigned long long* d_seq = nullptr;         // device word: rounds put so far (k_halo_put advances it)
    int32_t *d_send_ptr = nullptr;
    int32_t *d_sr_ptr = nullptr, *d_sr_dst = nullptr;   // per owned row: its send entries, neighbour k << 27 | slot (PutDst)
    int32_t n_sent_rows = 0;                     // owned rows with at least one send entry
    unsigned int* d_expect = nullptr;            // PutDst::expect (comm_plan_put_groups) for the block slots expect_key / expect_slots
    unsigned int h_expect[2] = {0, 0};
    const int32_t* expect_key = nullptr;
    int32_t expect_slots = 0;
    double** d_put = nullptr;                    // [2][nn] remote payload addresses
    unsigned long long** d_rflag = nullptr;      // [nn] remote flag addresses
    unsigned int* d_done = nullptr;
    Peer* owner = nullptr;                       // the communicator whose window holds win_recv (released in plan_free)
    int32_t n_send() const { return send_ptr.empty() ? 0 : send_ptr.back(); }
    int32_t n_recv() const { return recv_ptr.empty() ? 0 : recv_ptr.back(); }
};

struct Team {                                    // in-process communicator: a barrier + the ranks' peer ends
    explicit Team(int n_) : n(n_), peers((size_t)n_, nullptr), rc((size_t)n_, 0) {}
    ~Team();
    int n;
    std::mutex m;
    std::condition_variable cv;
    int count = 0;
    long gen = 0;
    std::vector<Peer*> peers;                    // one window per rank, wired directly (one address space: no IPC)
    std::vector<int> rc;                         // per-rank verdicts of collective setup steps
    void barrier();
};

struct Comm {
    ncclComm_t nccl = nullptr;
    Team* team = nullptr;
    Peer* peer = nullptr;                        // peer transport: the caller's (sns_peer_destroy); team: the Team's
    int rank = 0, nranks = 1;
    std::deque<Plan> plans;                      // per hierarchy level; plans[0] = assembled operator (stable references)
    bool active() const { return nccl != nullptr || team != nullptr || peer != nullptr; }
    // do the level passes read their ghost entries from the receive windows (peer / team), one put launch per exchange?
    bool windows() const { return peer != nullptr; }
};

// all return 0 or an SNS_E_* code (error text via set_error)
int plan_upload(Plan& p);                        // h_send_idx/h_recv_idx -> device, allocate buffers
void plan_free(Plan& p);
// fill the ghost entries of x (4 doubles per node) from the owning ranks
int comm_exchange(Comm* c, const Plan& p, double* x, hipStream_t s);
// window transports: start an exchange of x's owned values (one launch); the consumer takes comm_ghost_src(c, p) and reads the
// ghost entries from the window.  team: returns after the host barrier (every rank's put has completed).
int comm_put(Comm* c, const Plan& p, const double* x, hipStream_t s);
GhostSrc comm_ghost_src(const Comm* c, const Plan& p);
// the put carried by the kernel that produces the vector: pass comm_put_dst to that kernel, then comm_put_carried (the team's
// host barrier, the peer's error check) in place of comm_put.  A plan that cannot carry it answers with an empty PutDst: the
// caller falls back to comm_put.
int comm_plan_put_groups(Comm* c, Plan& p, const int32_t* blk_rows, int32_t n_slots, hipStream_t s);   // (setup; enables comm_put_dst)
PutDst comm_put_dst(const Comm* c, const Plan& p);
int comm_put_carried(Comm* c, const Plan& p, hipStream_t s);
int comm_allreduce_sum(Comm* c, double* buf_dev, int count, hipStream_t s);
// every rank contributes `count` doubles; recv_dev gets nranks*count (rank order)
int comm_allgather(Comm* c, const double* send_dev, double* recv_dev, int count, hipStream_t s);
// ... or, window transports only: rank r's cnt[r] doubles land at recv_dev + off[r] (dev_off / dev_cnt: device arrays [nranks],
// max_count >= every cnt[r]; one chunk: max_count must fit the staging area) -- the replicated tail's right-hand side arrives
// in the order of its rows, no gather kernel behind it
int comm_allgatherv(Comm* c, const double* send_dev, double* recv_dev, int max_count, const int64_t* dev_off,
                    const int64_t* dev_cnt, hipStream_t s);
// the two halves of an all-gather carried by solver kernels (sns_peer_dev.h: AgPut / AgGet); slot = doubles per rank
AgPut comm_ag_put(const Comm* c, int64_t slot_doubles);
AgGet comm_ag_get(const Comm* c);
// team transport: wait for this rank's stream, then for every rank (no-op otherwise).  The window transports' two-phase
// collectives call it between their halves.
int comm_host_barrier(Comm* c, hipStream_t s);

// peer transport
int peer_create(int device, int rank, int nranks, size_t window_bytes, Peer** out, char ipc_handle_out[64]);
int peer_connect(Peer* p, const char* handles /* nranks x 64 bytes, rank order */);
int peer_finish_connect(Peer* p);               // (second half of peer_connect: device tables from base[])
int peer_close_mappings(Peer* p);               // first half of the teardown: unmap the peers' windows (then synchronise the ranks)
int peer_destroy(Peer* p);
// in-process self-test + latency probe: nranks threads, a ring of halo links; us_out = {exchange, all-reduce, all-gather} per round
int peer_selftest(int device, int nranks, int halo_nodes, int reps, double us_out[3]);
// verified all-reduces / all-gathers / halo rings between the REAL ranks of a connected communicator (collective).
// `table_allgather` moves 3 * nranks doubles per rank between the ranks on the host side (the halo ring's plan offers).
int peer_check_links(Peer* p, int rounds);
// wire a freshly uploaded plan to the neighbours' windows.  The caller all-gathers `mine` (3 * nranks doubles per rank) into
// `all` between the two calls (collective: every rank connects the same plan at the same time).
struct PlanOffers {                              // where rank j writes in MY window: payload parity 0 / 1, flag; -1 = no link
    std::vector<double> mine, all;
};
int peer_plan_offer(Comm* c, Plan& p, PlanOffers& t);
int peer_plan_connect(Comm* c, Plan& p, const PlanOffers& t);
int peer_check(Comm* c);                         // SNS_E_COMM once a device-side wait has timed out
PeerArgs peer_allreduce_args(Peer* p, int phase);   // arguments of an all-reduce round (for a kernel that carries it inside)

// team transport (sns_attach_team): this rank's peer end on windows wired inside the process (collective over the team)
int team_peer(Team* t, int device, int rank, Peer** out);

}  // namespace sns
