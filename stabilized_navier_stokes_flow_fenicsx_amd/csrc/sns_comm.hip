// Transports of sns_comm.h: RCCL over xGMI and the direct peer-window transport (product), the in-process Team (tests).
#include "sns_comm.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>

#include "sns_internal.h"
#include "sns_kernels.h"

namespace sns {

#define CHIP(expr)                                                                                   \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                            \
            return SNS_E_HIP;                                                                        \
        }                                                                                            \
    } while (0)
#define CNCCL(expr)                                                                                  \
    do {                                                                                             \
        ncclResult_t _e = (expr);                                                                    \
        if (_e != ncclSuccess) {                                                                     \
            set_error(std::string(#expr) + ": " + ncclGetErrorString(_e));                           \
            return SNS_E_COMM;                                                                       \
        }                                                                                            \
    } while (0)


// ---- peer transport: device side ------------------------------------------------------------------------------------------
// Memory model (sns_peer_dev.h): payload stores -> system-scope fence -> system-scope release store of the round number; the
// receiver polls the round with system-scope acquire loads, every lane executes a system-scope acquire fence and reads the
// payload with plain loads issued after it.  Every wait is bounded by the communicator's timeout and reports through the mapped
// error word instead of spinning for ever.  Round numbers live in device memory and are advanced by the kernels.
namespace {

__device__ __forceinline__ unsigned long long ld_round(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// halo put: the owned values listed in send_idx go straight into the neighbours' receive buffers of round *seq_dev + 1; the last
// workgroup to finish raises this rank's flag in every neighbour's window (also across links that carry no payload in this
// direction: the flag is what keeps a rank from running two exchanges ahead of a neighbour, see comm_exchange) and stores the
// round.  rflag == nullptr (team transport): no flags, the host barrier orders the rounds.
__global__ __launch_bounds__(256) void k_halo_put(int32_t ns, int nn, const int32_t* __restrict__ send_idx,
                                                  const int32_t* __restrict__ send_ptr, const double* __restrict__ x,
                                                  double* const* __restrict__ put, unsigned long long* const* __restrict__ rflag,
                                                  unsigned long long* __restrict__ seq_dev, unsigned int* __restrict__ done) {
    __shared__ int32_t sp[PEER_MAX_RANKS + 1];
    __shared__ int last;
    const int tid = threadIdx.x;
    const unsigned long long seq = ld_round(seq_dev) + 1ull;
    double* const* __restrict__ dst = put + (size_t)(seq & 1ull) * nn;
    if (tid <= nn) sp[tid] = send_ptr[tid];
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * 256 + tid;        // one lane per half node: 16-byte loads and stores
    const int32_t i = (int32_t)(t >> 1);
    const int hh = (int)(t & 1);
    if (i < ns) {
        int k = 0;
        while (k + 1 < nn && i >= sp[k + 1]) ++k;
        const double2 v = *reinterpret_cast<const double2*>(x + 4 * (int64_t)send_idx[i] + 2 * hh);
        *reinterpret_cast<double2*>(dst[k] + 4 * (int64_t)(i - sp[k]) + 2 * hh) = v;
    }
    if (rflag) __threadfence_system();
    __syncthreads();
    if (tid == 0) last = (atomicAdd(done, 1u) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (last) {
        if (rflag) {
            __threadfence_system();
            if (tid < nn) peer_flag_store(rflag[tid], seq);
        }
        if (tid == 0) {
            *done = 0u;
            __hip_atomic_store(seq_dev, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// halo wait + unpack (the exchanges outside the level passes: setup, estimates, Newton's state): every workgroup waits for all
// neighbours' flags of the round the put in front of it left in *seq_dev, then scatters its part of the receive buffer
__global__ __launch_bounds__(256) void k_peer_wait_unpack(int32_t nr, int nn_wait, const int32_t* __restrict__ recv_idx,
                                                          const double* __restrict__ win0, const double* __restrict__ win1,
                                                          const unsigned long long* __restrict__ flag,
                                                          const unsigned long long* __restrict__ seq_dev, double* __restrict__ x,
                                                          int* err, long long timeout_ticks) {
    const int tid = threadIdx.x;
    const unsigned long long seq = ld_round(seq_dev);
    const double* __restrict__ recv_buf = (seq & 1ull) ? win1 : win0;
    if (nn_wait > 0) {
        if (tid < nn_wait) (void)peer_flag_wait(flag + tid, seq, timeout_ticks, err, 1);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");         // every lane: nothing older than the flags is read below
    }
    const int64_t t = (int64_t)blockIdx.x * 256 + tid;        // one lane per half node
    const int32_t i = (int32_t)(t >> 1);
    if (i < nr) {
        const double2 v = *reinterpret_cast<const double2*>(recv_buf + 2 * t);
        *reinterpret_cast<double2*>(x + 4 * (int64_t)recv_idx[i] + 2 * (t & 1)) = v;
    }
}

// all-reduce (sum) of count <= PEER_AR_MAX doubles, one workgroup: contribution into every rank's slot table, flags, wait for
// everybody's, sum in rank order (the same bits on every rank)
__global__ __launch_bounds__(256) void k_peer_allreduce(double* __restrict__ buf, int count, PeerArgs a) {
    peer_allreduce_block(buf, count, a);
}

// all-gather, put half: workgroup (b, r) copies its share of this rank's doubles into rank r's staging area (slot `rank` of the
// round's parity); the last workgroup of column r raises the flag there.  cnt == nullptr: m doubles from send + 0;
// else cnt[rank] doubles (all-gather of ragged pieces).  Round = *seq_dev + 1 (advanced by the wait half).
__global__ __launch_bounds__(256) void k_peer_ag_put(const double* __restrict__ send, int64_t m_uniform,
                                                     const int64_t* __restrict__ cnt, int64_t slot_doubles,
                                                     const unsigned long long* __restrict__ seq_dev, int rank, int64_t stage_doubles,
                                                     double* const* __restrict__ ag, PeerCtl* const* __restrict__ ctl,
                                                     unsigned int* __restrict__ done, int flags) {
    __shared__ int last;
    const int tid = threadIdx.x, r = blockIdx.y;
    const unsigned long long seq = ld_round(seq_dev) + 1ull;
    const int64_t m = cnt ? cnt[rank] : m_uniform;
    double* __restrict__ dst = ag[r] + (int64_t)(seq & 1ull) * stage_doubles + (int64_t)rank * slot_doubles;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < m; i += (int64_t)gridDim.x * 256) dst[i] = send[i];
    if (!flags) return;
    __threadfence_system();
    __syncthreads();
    if (tid == 0) last = (atomicAdd(done + r, 1u) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (last && tid == 0) {
        __threadfence_system();
        peer_flag_store(&ctl[r]->ag_flag[rank], seq);
        done[r] = 0u;
    }
}
// all-gather, wait half: workgroup (b, r) waits for rank r's flag and copies r's doubles from the own staging area to
// recv + (off ? off[r] : r * count + chunk_off); the last workgroup of the grid stores the round
__global__ __launch_bounds__(256) void k_peer_ag_wait_copy(double* __restrict__ recv, int64_t count, int64_t chunk_off,
                                                           int64_t m_uniform, const int64_t* __restrict__ cnt,
                                                           const int64_t* __restrict__ off, int64_t slot_doubles,
                                                           unsigned long long* __restrict__ seq_dev, int rank, int64_t stage_doubles,
                                                           double* const* __restrict__ ag, PeerCtl* const* __restrict__ ctl,
                                                           unsigned int* __restrict__ done_all, int flags, int* err,
                                                           long long timeout_ticks) {
    __shared__ int last;
    const int tid = threadIdx.x, r = blockIdx.y;
    const unsigned long long seq = ld_round(seq_dev) + 1ull;
    if (flags) {
        if (tid == 0) (void)peer_flag_wait(&ctl[rank]->ag_flag[r], seq, timeout_ticks, err, 3);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    const int64_t m = cnt ? cnt[r] : m_uniform;
    const double* __restrict__ src = ag[rank] + (int64_t)(seq & 1ull) * stage_doubles + (int64_t)r * slot_doubles;
    double* __restrict__ dst = recv + (off ? off[r] : (int64_t)r * count + chunk_off);
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < m; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
    __syncthreads();
    if (tid == 0) last = (atomicAdd(done_all, 1u) == gridDim.x * gridDim.y - 1) ? 1 : 0;
    __syncthreads();
    if (last && tid == 0) {
        *done_all = 0u;
        __hip_atomic_store(seq_dev, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

#define CTRY(expr)                                                                                   \
    do {                                                                                             \
        int _rc = (expr);                                                                            \
        if (_rc != SNS_OK) return _rc;                                                               \
    } while (0)

void Team::barrier() {
    std::unique_lock<std::mutex> lk(m);
    const long g = gen;
    if (++count == n) {
        count = 0;
        ++gen;
        cv.notify_all();
    } else {
        cv.wait(lk, [&] { return gen != g; });
    }
}
Team::~Team() {
    for (Peer* p : peers)
        if (p) (void)peer_destroy(p);
}

int comm_host_barrier(Comm* c, hipStream_t s) {
    if (!c || !c->team) return SNS_OK;
    CHIP(hipStreamSynchronize(s));
    c->team->barrier();
    return SNS_OK;
}

int plan_upload(Plan& p) {
    const size_t ns = (size_t)p.n_send(), nr = (size_t)p.n_recv();
    CHIP(hipMalloc((void**)&p.send_idx, std::max<size_t>(1, ns) * sizeof(int32_t)));
    CHIP(hipMalloc((void**)&p.recv_idx, std::max<size_t>(1, nr) * sizeof(int32_t)));
    if (ns) CHIP(hipMemcpy(p.send_idx, p.h_send_idx.data(), ns * sizeof(int32_t), hipMemcpyHostToDevice));
    if (nr) CHIP(hipMemcpy(p.recv_idx, p.h_recv_idx.data(), nr * sizeof(int32_t), hipMemcpyHostToDevice));
    CHIP(hipMalloc((void**)&p.send_buf, std::max<size_t>(1, 4 * ns) * sizeof(double)));
    CHIP(hipMalloc((void**)&p.recv_buf, std::max<size_t>(1, 4 * nr) * sizeof(double)));
    p.identity_recv = true;
    for (size_t q = 0; q < nr; ++q)
        if (p.h_recv_idx[q] != p.n_own + (int32_t)q) { p.identity_recv = false; break; }
    return SNS_OK;
}

void plan_free(Plan& p) {
    if (p.send_idx) (void)hipFree(p.send_idx);
    if (p.recv_idx) (void)hipFree(p.recv_idx);
    if (p.send_buf) (void)hipFree(p.send_buf);
    if (p.recv_buf) (void)hipFree(p.recv_buf);
    if (p.d_send_ptr) (void)hipFree(p.d_send_ptr);
    if (p.d_sr_ptr) (void)hipFree(p.d_sr_ptr);
    if (p.d_sr_dst) (void)hipFree(p.d_sr_dst);
    p.d_sr_ptr = p.d_sr_dst = nullptr;
    if (p.d_expect) (void)hipFree(p.d_expect);
    p.d_expect = nullptr;
    p.expect_key = nullptr;
    if (p.d_put) (void)hipFree(p.d_put);
    if (p.d_rflag) (void)hipFree(p.d_rflag);
    if (p.d_done) (void)hipFree(p.d_done);
    if (p.d_seq) (void)hipFree(p.d_seq);
    p.send_idx = p.recv_idx = nullptr;
    p.send_buf = p.recv_buf = nullptr;
    p.d_send_ptr = nullptr;
    p.d_put = nullptr;
    p.d_rflag = nullptr;
    p.d_done = nullptr;
    p.d_seq = nullptr;
    if (p.owner && p.win_recv[0]) {
        // the plan area of a window is a bump allocator: it is reclaimed as a whole when the last plan carved from it goes
        // (a communicator reused across problems -- bench legs, test suites -- would exhaust its window otherwise)
        if (--p.owner->plans_live <= 0) {
            p.owner->plans_live = 0;
            p.owner->bump = p.owner->bump0;
            p.owner->flag_bump = p.owner->flag_off;
        }
    }
    p.owner = nullptr;
    p.win_recv[0] = p.win_recv[1] = nullptr;
    p.win_flag = nullptr;
}

// window transports: the put half of an exchange
int comm_put(Comm* c, const Plan& p, const double* x, hipStream_t s) {
    if (!c || !c->peer) { set_error("comm_put: no window transport"); return SNS_E_STATE; }
    const int nn = (int)p.nbr.size();
    if (nn == 0) return comm_host_barrier(c, s);             // (team: every rank passes the barrier of every exchange)
    if (!p.d_put) { set_error("peer transport: halo plan was not connected"); return SNS_E_STATE; }
    CTRY(peer_check(c));
    const int32_t ns = p.n_send();
    const unsigned gp = (unsigned)std::max<int64_t>(1, (2 * (int64_t)ns + 255) / 256);
    hipLaunchKernelGGL(k_halo_put, dim3(gp), dim3(256), 0, s, ns, nn, p.send_idx, p.d_send_ptr, x, p.d_put,
                       c->peer->host_sync ? (unsigned long long* const*)nullptr : p.d_rflag, p.d_seq, p.d_done);
    return comm_host_barrier(c, s);
}

// workgroups of a producing kernel that hold a sent row (PutDst::expect), for its two slot groupings; out[2]: sent rows covered
__global__ void k_count_put_groups(int32_t ns, const int32_t* __restrict__ blk_rows, const int32_t* __restrict__ sr_ptr,
                                   unsigned int* __restrict__ out) {
    const int32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (8 * (int64_t)g >= ns) return;
    unsigned int n8 = 0;
    for (int32_t q = 8 * g; q < std::min(ns, 8 * g + 8); ++q) {
        const int32_t row = blk_rows[q];
        if (row >= 0 && sr_ptr[row + 1] > sr_ptr[row]) ++n8;
    }
    if (n8) {
        atomicAdd(&out[1], 1u);
        atomicAdd(&out[2], n8);
    }
    if (g % 8 == 0) {
        bool any = false;
        for (int32_t q = 8 * g; q < std::min(ns, 8 * g + 64) && !any; ++q) {
            const int32_t row = blk_rows[q];
            any = row >= 0 && sr_ptr[row + 1] > sr_ptr[row];
        }
        if (any) atomicAdd(&out[0], 1u);
    }
}

// (setup, once per hierarchy: blk_rows = the level's block slots, 8 per aggregate block)
int comm_plan_put_groups(Comm* c, Plan& p, const int32_t* blk_rows, int32_t n_slots, hipStream_t s) {
    if (!c || !c->peer || !p.d_sr_ptr || p.nbr.empty()) return SNS_OK;
    if (p.expect_key == blk_rows && p.expect_slots == n_slots) return SNS_OK;
    p.expect_key = nullptr;
    p.h_expect[0] = p.h_expect[1] = 0;
    if (!blk_rows || n_slots <= 0) return SNS_OK;
    if (!p.d_expect) CHIP(hipMalloc((void**)&p.d_expect, 3 * sizeof(unsigned int)));
    CHIP(hipMemsetAsync(p.d_expect, 0, 3 * sizeof(unsigned int), s));
    const int32_t ng = (n_slots + 7) / 8;
    hipLaunchKernelGGL(k_count_put_groups, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, s, n_slots, blk_rows, p.d_sr_ptr, p.d_expect);
    unsigned int e[3] = {0, 0, 0};
    CHIP(hipMemcpyAsync(e, p.d_expect, sizeof(e), hipMemcpyDeviceToHost, s));
    CHIP(hipStreamSynchronize(s));
    if (e[2] != (unsigned int)p.n_sent_rows) return SNS_OK;      // (a sent row outside the blocks: the plan cannot carry its put)
    p.h_expect[0] = e[0];
    p.h_expect[1] = e[1];
    p.expect_key = blk_rows;
    p.expect_slots = n_slots;
    return SNS_OK;
}

// ... the same put carried by the kernel that produces x (PutDst; off when the plan cannot: no neighbours, RCCL): the caller
// launches that kernel and then calls comm_put_carried in place of comm_put
PutDst comm_put_dst(const Comm* c, const Plan& p) {
    PutDst d;
    if (!c || !c->peer || p.nbr.empty() || !p.d_put || !p.d_sr_ptr || !p.expect_key || p.h_expect[0] == 0) return d;
    d.expect = p.d_expect;
    d.sr_ptr = p.d_sr_ptr;
    d.sr_dst = p.d_sr_dst;
    d.put = p.d_put;
    d.rflag = c->peer->host_sync ? (unsigned long long* const*)nullptr : p.d_rflag;
    d.seq = p.d_seq;
    d.done = p.d_done;
    d.nn = (int)p.nbr.size();
    return d;
}
int comm_put_carried(Comm* c, const Plan& p, hipStream_t s) {
    if (!c || !c->peer) { set_error("comm_put_carried: no window transport"); return SNS_E_STATE; }
    if (!p.nbr.empty()) CTRY(peer_check(c));
    return comm_host_barrier(c, s);
}

GhostSrc comm_ghost_src(const Comm* c, const Plan& p) {
    GhostSrc g;
    if (!c || !c->peer || p.nbr.empty() || !p.win_recv[0] || !p.identity_recv) return g;
    g.win[0] = p.win_recv[0] - 4 * (int64_t)p.n_own;
    g.win[1] = p.win_recv[1] - 4 * (int64_t)p.n_own;
    g.n_own = p.n_own;
    g.nn = c->peer->host_sync ? 0 : (int)p.nbr.size();
    g.seq = p.d_seq;
    g.flag = p.win_flag;
    g.err = c->peer->err_dev;
    g.timeout_ticks = c->peer->timeout_ticks;
    return g;
}

int comm_exchange(Comm* c, const Plan& p, double* x, hipStream_t s) {
    if (!c || !c->active() || c->nranks <= 1) return SNS_OK;
    const int nn = (int)p.nbr.size();
    const int32_t ns = p.n_send(), nr = p.n_recv();
    if (c->peer) {
        // Flow control without acknowledgements: the receive buffers are double-buffered by the parity of the plan's round, and
        // this rank can start exchange s + 2 only after it has seen every neighbour's flag s + 1, which that neighbour raised
        // after (stream order) it had consumed exchange s -- the buffer about to be overwritten.
        Peer* pe = c->peer;
        CTRY(comm_put(c, p, x, s));
        if (nn == 0) return SNS_OK;
        const unsigned gu = (unsigned)std::max<int64_t>(1, (2 * (int64_t)nr + 255) / 256);
        hipLaunchKernelGGL(k_peer_wait_unpack, dim3(gu), dim3(256), 0, s, nr, pe->host_sync ? 0 : nn, p.recv_idx, p.win_recv[0],
                           p.win_recv[1], p.win_flag, p.d_seq, x, pe->err_dev, pe->timeout_ticks);
        return SNS_OK;
    }
    if (!c->nccl) { set_error("comm_exchange: no transport"); return SNS_E_STATE; }
    if (ns > 0)
        hipLaunchKernelGGL(k_pack, dim3((unsigned)((4 * (int64_t)ns + 255) / 256)), dim3(256), 0, s, ns, p.send_idx, x,
                           p.send_buf);
    if (nn > 0) {
        CNCCL(ncclGroupStart());
        for (int k = 0; k < nn; ++k) {
            const int32_t s0 = p.send_ptr[k], s1 = p.send_ptr[k + 1];
            const int32_t r0 = p.recv_ptr[k], r1 = p.recv_ptr[k + 1];
            if (s1 > s0)
                CNCCL(ncclSend(p.send_buf + 4 * (int64_t)s0, 4 * (size_t)(s1 - s0), ncclDouble, p.nbr[k], c->nccl, s));
            if (r1 > r0)
                CNCCL(ncclRecv(p.recv_buf + 4 * (int64_t)r0, 4 * (size_t)(r1 - r0), ncclDouble, p.nbr[k], c->nccl, s));
        }
        CNCCL(ncclGroupEnd());
    }
    if (nr > 0)
        hipLaunchKernelGGL(k_unpack, dim3((unsigned)((4 * (int64_t)nr + 255) / 256)), dim3(256), 0, s, nr, p.recv_idx,
                           p.recv_buf, x);
    return SNS_OK;
}

int comm_allreduce_sum(Comm* c, double* buf, int count, hipStream_t s) {
    if (!c || !c->active()) return SNS_OK;
    if (c->peer) {
        Peer* pe = c->peer;
        CTRY(peer_check(c));
        for (int off = 0; off < count; off += PEER_AR_MAX) {
            const int m = std::min(PEER_AR_MAX, count - off);
            if (pe->host_sync) {
                hipLaunchKernelGGL(k_peer_allreduce, dim3(1), dim3(256), 0, s, buf + off, m, peer_allreduce_args(pe, 1));
                CTRY(comm_host_barrier(c, s));
                hipLaunchKernelGGL(k_peer_allreduce, dim3(1), dim3(256), 0, s, buf + off, m, peer_allreduce_args(pe, 2));
            } else {
                hipLaunchKernelGGL(k_peer_allreduce, dim3(1), dim3(256), 0, s, buf + off, m, peer_allreduce_args(pe, 0));
            }
        }
        return SNS_OK;
    }
    if (c->nccl) {                                           // also with one rank: keeps the RCCL path exercised
        CNCCL(ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, c->nccl, s));
        return SNS_OK;
    }
    set_error("comm_allreduce_sum: no transport");
    return SNS_E_STATE;
}

namespace {
// one chunk of an all-gather over the windows: m doubles per rank (uniform) or cnt[r] (ragged, <= slot doubles each)
int peer_allgather_chunk(Comm* c, const double* send, double* recv, int64_t count, int64_t chunk_off, int64_t m,
                         const int64_t* cnt, const int64_t* off, int64_t slot, hipStream_t s) {
    Peer* pe = c->peer;
    const int flags = pe->host_sync ? 0 : 1;
    const unsigned gb = (unsigned)std::min<int64_t>(64, std::max<int64_t>(1, (m + 2047) / 2048));
    hipLaunchKernelGGL(k_peer_ag_put, dim3(gb, pe->nranks), dim3(256), 0, s, send, m, cnt, slot, pe->d_seq + 1, pe->rank,
                       (int64_t)pe->ag_doubles, pe->d_ag, pe->d_ctl, pe->d_done, flags);
    CTRY(comm_host_barrier(c, s));
    hipLaunchKernelGGL(k_peer_ag_wait_copy, dim3(gb, pe->nranks), dim3(256), 0, s, recv, count, chunk_off, m, cnt, off, slot,
                       pe->d_seq + 1, pe->rank, (int64_t)pe->ag_doubles, pe->d_ag, pe->d_ctl, pe->d_done + PEER_MAX_RANKS, flags,
                       pe->err_dev, pe->timeout_ticks);
    return SNS_OK;
}
}  // namespace

AgPut comm_ag_put(const Comm* c, int64_t slot_doubles) {
    AgPut a;
    if (!c || !c->peer) return a;
    const Peer* pe = c->peer;
    a.ag = pe->d_ag;
    a.ctl = pe->d_ctl;
    a.seq = pe->d_seq + 1;
    a.done = pe->d_done + PEER_MAX_RANKS + 1;
    a.rank = pe->rank;
    a.nranks = pe->nranks;
    a.flags = pe->host_sync ? 0 : 1;
    a.stage_doubles = (long long)pe->ag_doubles;
    a.slot_doubles = (long long)slot_doubles;
    return a;
}
AgGet comm_ag_get(const Comm* c) {
    AgGet g;
    if (!c || !c->peer) return g;
    const Peer* pe = c->peer;
    g.stage = reinterpret_cast<const double*>(pe->base[pe->rank] + pe->ag_off);
    g.ctl = reinterpret_cast<const PeerCtl*>(pe->base[pe->rank]);
    g.seq = pe->d_seq + 1;
    g.done = pe->d_done + PEER_MAX_RANKS;
    g.nranks = pe->nranks;
    g.flags = pe->host_sync ? 0 : 1;
    g.stage_doubles = (long long)pe->ag_doubles;
    g.err = pe->err_dev;
    g.timeout_ticks = pe->timeout_ticks;
    return g;
}

int comm_allgather(Comm* c, const double* send, double* recv, int count, hipStream_t s) {
    if (!c || !c->active() || c->nranks <= 1) {
        CHIP(hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, s));
        return SNS_OK;
    }
    if (c->peer) {
        Peer* pe = c->peer;
        CTRY(peer_check(c));
        const int64_t per = (int64_t)(pe->ag_doubles / (size_t)pe->nranks);          // doubles per rank and chunk
        for (int64_t off = 0; off < count; off += per) {
            const int64_t m = std::min<int64_t>(per, count - off);
            CTRY(peer_allgather_chunk(c, send + off, recv, (int64_t)count, off, m, nullptr, nullptr, m, s));
        }
        return SNS_OK;
    }
    if (c->nccl) {
        CNCCL(ncclAllGather(send, recv, count, ncclDouble, c->nccl, s));
        return SNS_OK;
    }
    set_error("comm_allgather: no transport");
    return SNS_E_STATE;
}

int comm_allgatherv(Comm* c, const double* send, double* recv, int max_count, const int64_t* dev_off, const int64_t* dev_cnt,
                    hipStream_t s) {
    if (!c || !c->peer) { set_error("comm_allgatherv: window transports only"); return SNS_E_STATE; }
    Peer* pe = c->peer;
    CTRY(peer_check(c));
    const int64_t per = (int64_t)(pe->ag_doubles / (size_t)pe->nranks);
    if (max_count > per) { set_error("comm_allgatherv: piece larger than the staging area"); return SNS_E_ARG; }
    return peer_allgather_chunk(c, send, recv, 0, 0, (int64_t)max_count, dev_cnt, dev_off, (int64_t)max_count, s);
}

// ---- peer transport: host side ----------------------------------------------------------------------------------------------
namespace {
constexpr size_t PEER_CTL_BYTES = (sizeof(PeerCtl) + 4095) / 4096 * 4096;
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
}  // namespace

int peer_create(int device, int rank, int nranks, size_t window_bytes, Peer** out, char handle_out[64]) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t size");
    if (!out || nranks < 1 || nranks > PEER_MAX_RANKS || rank < 0 || rank >= nranks) {
        set_error("sns_peer_create: bad arguments (at most " + std::to_string(PEER_MAX_RANKS) + " ranks)");
        return SNS_E_ARG;
    }
    if (window_bytes == 0) window_bytes = (size_t)64 << 20;
    if (window_bytes < ((size_t)4 << 20)) { set_error("sns_peer_create: window below 4 MiB"); return SNS_E_ARG; }
    CHIP(hipSetDevice(device));
    std::unique_ptr<Peer> p(new Peer);
    p->rank = rank;
    p->nranks = nranks;
    p->device = device;
    p->bytes = window_bytes;
    void* w = nullptr;
    // fine-grained: stores of another GPU become visible to a running kernel of this one (SNS_PEER_WINDOW=uncached | plain for
    // experiments; "plain" is only right when all ranks share one GPU)
    const char* kind = std::getenv("SNS_PEER_WINDOW");
    if (kind && std::strcmp(kind, "plain") == 0) CHIP(hipMalloc(&w, window_bytes));
    else if (kind && std::strcmp(kind, "uncached") == 0) CHIP(hipExtMallocWithFlags(&w, window_bytes, hipDeviceMallocUncached));
    else CHIP(hipExtMallocWithFlags(&w, window_bytes, hipDeviceMallocFinegrained));
    p->base[rank] = static_cast<char*>(w);
    CHIP(hipMemset(w, 0, window_bytes));
    CHIP(hipDeviceSynchronize());
    // control area | the plans' arrival flags (64 KiB, flags only) | all-gather staging (2 parities, a quarter of the window each
    // at most 32 MiB) | plan area (receive buffers)
    p->flag_off = p->flag_bump = PEER_CTL_BYTES;
    p->flag_end = p->flag_off + ((size_t)64 << 10);
    p->ag_off = p->flag_end;
    const size_t ag_bytes = std::min<size_t>((size_t)32 << 20, window_bytes / 4) / (8 * (size_t)nranks) * (8 * (size_t)nranks);
    p->ag_doubles = ag_bytes / 8;
    p->bump = p->bump0 = align_up(p->ag_off + 2 * ag_bytes, 4096);
    CHIP(hipHostMalloc((void**)&p->err_host, sizeof(int), hipHostMallocMapped));
    *p->err_host = 0;
    CHIP(hipHostGetDevicePointer((void**)&p->err_dev, p->err_host, 0));
    CHIP(hipMalloc((void**)&p->d_done, (PEER_MAX_RANKS + 2) * sizeof(unsigned int)));
    CHIP(hipMemset(p->d_done, 0, (PEER_MAX_RANKS + 2) * sizeof(unsigned int)));
    CHIP(hipMalloc((void**)&p->d_seq, 2 * sizeof(unsigned long long)));
    CHIP(hipMemset(p->d_seq, 0, 2 * sizeof(unsigned long long)));
    double ms = 20000.0;
    if (const char* t = std::getenv("SNS_PEER_TIMEOUT_MS")) ms = std::max(1.0, std::atof(t));
    p->timeout_ticks = (long long)(ms * 1.0e5);                                  // wall_clock64(): 100 MHz
    if (handle_out) {
        hipIpcMemHandle_t hd;
        CHIP(hipIpcGetMemHandle(&hd, w));
        std::memcpy(handle_out, &hd, 64);
    }
    *out = p.release();
    return SNS_OK;
}

int peer_connect(Peer* p, const char* handles) {
    if (!p || !handles) return SNS_E_ARG;
    if (p->connected) { set_error("sns_peer_connect: already connected"); return SNS_E_STATE; }
    CHIP(hipSetDevice(p->device));
    for (int r = 0; r < p->nranks; ++r) {
        if (r == p->rank) continue;
        hipIpcMemHandle_t hd;
        std::memcpy(&hd, handles + (size_t)64 * r, 64);
        void* w = nullptr;
        CHIP(hipIpcOpenMemHandle(&w, hd, hipIpcMemLazyEnablePeerAccess));
        p->base[r] = static_cast<char*>(w);
        p->mapped[r] = true;
    }
    return peer_finish_connect(p);
}

// device tables of the peers' control / staging areas, once base[] is complete
int peer_finish_connect(Peer* p) {
    std::vector<PeerCtl*> ctl((size_t)p->nranks);
    std::vector<double*> ag((size_t)p->nranks);
    for (int r = 0; r < p->nranks; ++r) {
        ctl[(size_t)r] = reinterpret_cast<PeerCtl*>(p->base[r]);
        ag[(size_t)r] = reinterpret_cast<double*>(p->base[r] + p->ag_off);
    }
    CHIP(hipMalloc((void**)&p->d_ctl, ctl.size() * sizeof(PeerCtl*)));
    CHIP(hipMalloc((void**)&p->d_ag, ag.size() * sizeof(double*)));
    CHIP(hipMemcpy(p->d_ctl, ctl.data(), ctl.size() * sizeof(PeerCtl*), hipMemcpyHostToDevice));
    CHIP(hipMemcpy(p->d_ag, ag.data(), ag.size() * sizeof(double*), hipMemcpyHostToDevice));
    p->connected = true;
    return SNS_OK;
}

// Teardown in two phases (the IPC contract leaves freeing a window that a peer still has mapped undefined): every rank closes
// its mappings of the others' windows, the ranks synchronise (caller), then every rank frees its own window.
int peer_close_mappings(Peer* p) {
    if (!p) return SNS_OK;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < p->nranks; ++r)
        if (p->mapped[r] && p->base[r]) {
            (void)hipIpcCloseMemHandle(p->base[r]);
            p->mapped[r] = false;
            p->base[r] = nullptr;
        }
    return SNS_OK;
}

int peer_destroy(Peer* p) {
    if (!p) return SNS_OK;
    (void)peer_close_mappings(p);                            // (no-op after an explicit first phase)
    if (p->base[p->rank]) (void)hipFree(p->base[p->rank]);
    if (p->d_ctl) (void)hipFree(p->d_ctl);
    if (p->d_ag) (void)hipFree(p->d_ag);
    if (p->d_done) (void)hipFree(p->d_done);
    if (p->d_seq) (void)hipFree(p->d_seq);
    if (p->err_host) (void)hipHostFree(p->err_host);
    delete p;
    return SNS_OK;
}

PeerArgs peer_allreduce_args(Peer* p, int phase) {
    PeerArgs a;
    a.seq = p->d_seq;
    a.rank = p->rank;
    a.nranks = p->nranks;
    a.ctl = p->d_ctl;
    a.err = p->err_dev;
    a.timeout_ticks = p->timeout_ticks;
    a.phase = phase;
    return a;
}

int peer_check(Comm* c) {
    if (!c || !c->peer) return SNS_OK;
    const int e = *reinterpret_cast<volatile int*>(c->peer->err_host);
    if (e == 0) return SNS_OK;
    static const char* what[] = {"", "halo exchange", "all-reduce", "all-gather"};
    set_error(std::string("peer transport: rank ") + std::to_string(c->rank) + " gave up waiting in a(n) " +
              what[(e >= 1 && e <= 3) ? e : 0] + " (a peer is late by more than SNS_PEER_TIMEOUT_MS, or gone)");
    return SNS_E_COMM;
}

// Receive buffers and flags of a plan are carved from this rank's window; the offsets a neighbour must write to are offered
// through one host all-gather (3 doubles per rank pair), after which every rank can compute the remote addresses.
int peer_plan_offer(Comm* c, Plan& p, PlanOffers& t) {
    Peer* pe = c->peer;
    const int nr_ranks = pe->nranks, nn = (int)p.nbr.size();
    if (!pe->connected) { set_error("peer transport: sns_peer_connect has not been called"); return SNS_E_STATE; }
    t.mine.assign((size_t)3 * nr_ranks + 1, -1.0);
    const size_t flag_off = align_up(pe->flag_bump, 64);
    const size_t rbytes = align_up(std::max<size_t>(32, (size_t)p.n_recv() * 32), 256);
    const size_t r0 = align_up(pe->bump, 256), r1 = r0 + rbytes, end = r1 + rbytes;
    if (end > pe->bytes || flag_off + (size_t)std::max(1, nn) * 8 > pe->flag_end) {
        set_error("peer transport: window of " + std::to_string(pe->bytes >> 20) + " MiB exhausted (sns_peer_create window_bytes)");
        return SNS_E_COMM;
    }
    pe->bump = end;
    pe->flag_bump = flag_off + (size_t)std::max(1, nn) * 8;
    ++pe->plans_live;
    p.owner = pe;
    char* own = pe->base[pe->rank];
    p.win_flag = reinterpret_cast<unsigned long long*>(own + flag_off);
    p.win_recv[0] = reinterpret_cast<double*>(own + r0);
    p.win_recv[1] = reinterpret_cast<double*>(own + r1);
    // (both areas are reused after a reclaim.  Nothing is cleared: the flag area only ever holds round numbers, and the rounds of a
    // new plan start above every round any rank has used so far, see peer_plan_connect; stale payload is never read before the
    // flags of its round have arrived)
    t.mine[(size_t)3 * nr_ranks] = (double)pe->epoch;
    for (int k = 0; k < nn; ++k) {
        const int j = p.nbr[(size_t)k];
        if (j < 0 || j >= nr_ranks || j == pe->rank) continue;                    // (check_plan_symmetry reports it)
        t.mine[(size_t)3 * j + 0] = (double)(r0 + (size_t)32 * p.recv_ptr[(size_t)k]);
        t.mine[(size_t)3 * j + 1] = (double)(r1 + (size_t)32 * p.recv_ptr[(size_t)k]);
        t.mine[(size_t)3 * j + 2] = (double)(flag_off + (size_t)8 * k);
    }
    return SNS_OK;
}

int peer_plan_connect(Comm* c, Plan& p, const PlanOffers& t) {
    Peer* pe = c->peer;
    const int nr_ranks = pe->nranks, nn = (int)p.nbr.size();
    const size_t LEN = (size_t)3 * nr_ranks + 1;
    if (t.all.size() != LEN * nr_ranks) { set_error("peer transport: bad offer table"); return SNS_E_COMM; }
    if (nn > PEER_MAX_RANKS) { set_error("peer transport: too many neighbours"); return SNS_E_ARG; }
    // first round of this plan: above every round of every earlier plan of any rank (the same number on every rank: the table is)
    double ep = 0.0;
    for (int r = 0; r < nr_ranks; ++r) ep = std::max(ep, t.all[(size_t)r * LEN + (size_t)3 * nr_ranks]);
    pe->epoch = (long long)ep + 1;
    const unsigned long long round0 = (unsigned long long)pe->epoch << 32;
    std::vector<double*> put((size_t)2 * std::max(1, nn), nullptr);
    std::vector<unsigned long long*> rflag((size_t)std::max(1, nn), nullptr);
    for (int k = 0; k < nn; ++k) {
        const int j = p.nbr[(size_t)k];
        const double* o = t.all.data() + (size_t)j * LEN + (size_t)3 * pe->rank;      // what rank j offers to this rank
        if (o[0] < 0.0 || o[1] < 0.0 || o[2] < 0.0) {
            set_error("peer transport: rank " + std::to_string(j) + " does not list rank " + std::to_string(pe->rank) + " as a neighbour");
            return SNS_E_COMM;
        }
        put[(size_t)k] = reinterpret_cast<double*>(pe->base[j] + (size_t)o[0]);
        put[(size_t)nn + k] = reinterpret_cast<double*>(pe->base[j] + (size_t)o[1]);
        rflag[(size_t)k] = reinterpret_cast<unsigned long long*>(pe->base[j] + (size_t)o[2]);
    }
    std::vector<int32_t> sp(p.send_ptr);
    if (sp.empty()) sp.assign(1, 0);
    CHIP(hipMalloc((void**)&p.d_send_ptr, sp.size() * sizeof(int32_t)));
    CHIP(hipMemcpy(p.d_send_ptr, sp.data(), sp.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    {
        // per owned row: where it is sent (the put carried by the producing kernel, comm_put_dst)
        const int32_t no = std::max<int32_t>(p.n_own, 0);
        std::vector<int32_t> rp((size_t)no + 1, 0), rd((size_t)std::max<int32_t>(1, p.n_send()), 0);
        bool ok = nn < 16 && p.h_send_idx.size() >= (size_t)p.n_send();
        for (int k = 0; k < nn && ok; ++k)
            for (int32_t q = p.send_ptr[(size_t)k]; q < p.send_ptr[(size_t)k + 1]; ++q) {
                const int32_t row = p.h_send_idx[(size_t)q];
                if (row < 0 || row >= no || q - p.send_ptr[(size_t)k] >= (1 << 27)) { ok = false; break; }
                ++rp[(size_t)row + 1];
            }
        if (ok) {
            p.n_sent_rows = 0;
            for (int32_t i = 0; i < no; ++i) p.n_sent_rows += rp[(size_t)i + 1] > 0 ? 1 : 0;
            for (int32_t i = 0; i < no; ++i) rp[(size_t)i + 1] += rp[(size_t)i];
            std::vector<int32_t> at(rp.begin(), rp.end() - 1);
            for (int k = 0; k < nn; ++k)
                for (int32_t q = p.send_ptr[(size_t)k]; q < p.send_ptr[(size_t)k + 1]; ++q)
                    rd[(size_t)at[(size_t)p.h_send_idx[(size_t)q]]++] = (int32_t)(((uint32_t)k << 27) | (uint32_t)(q - p.send_ptr[(size_t)k]));
            CHIP(hipMalloc((void**)&p.d_sr_ptr, rp.size() * sizeof(int32_t)));
            CHIP(hipMemcpy(p.d_sr_ptr, rp.data(), rp.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            CHIP(hipMalloc((void**)&p.d_sr_dst, rd.size() * sizeof(int32_t)));
            CHIP(hipMemcpy(p.d_sr_dst, rd.data(), rd.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    }
    CHIP(hipMalloc((void**)&p.d_put, put.size() * sizeof(double*)));
    CHIP(hipMalloc((void**)&p.d_rflag, rflag.size() * sizeof(unsigned long long*)));
    CHIP(hipMemcpy(p.d_put, put.data(), put.size() * sizeof(double*), hipMemcpyHostToDevice));
    CHIP(hipMemcpy(p.d_rflag, rflag.data(), rflag.size() * sizeof(unsigned long long*), hipMemcpyHostToDevice));
    CHIP(hipMalloc((void**)&p.d_done, sizeof(unsigned int)));
    CHIP(hipMemset(p.d_done, 0, sizeof(unsigned int)));
    CHIP(hipMalloc((void**)&p.d_seq, sizeof(unsigned long long)));
    CHIP(hipMemcpy(p.d_seq, &round0, sizeof(unsigned long long), hipMemcpyHostToDevice));
    return SNS_OK;
}

// team transport: this rank's peer end on a window of its own, wired to the other ranks' windows inside the one address space
// (what sns_peer_connect does through HIP IPC between processes).  Collective over the team's threads; a communicator is made
// once per team and rank and shared by every handle attached to the team afterwards.
int team_peer(Team* t, int device, int rank, Peer** out) {
    if (!t || rank < 0 || rank >= t->n || !out) return SNS_E_ARG;
    if (t->n > PEER_MAX_RANKS) { set_error("team transport: at most " + std::to_string(PEER_MAX_RANKS) + " ranks"); return SNS_E_ARG; }
    if (!t->peers[(size_t)rank]) {
        Peer* p = nullptr;
        size_t wb = (size_t)64 << 20;
        if (const char* e = std::getenv("SNS_TEAM_WINDOW_MB")) wb = (size_t)std::max(4, std::atoi(e)) << 20;
        t->rc[(size_t)rank] = peer_create(device, rank, t->n, wb, &p, nullptr);
        t->peers[(size_t)rank] = p;
        const std::string err = sns_last_error();
        t->barrier();                                        // every window exists (or its rank has recorded why not)
        int rc = SNS_OK;
        for (int q = 0; q < t->n; ++q)
            if (t->rc[(size_t)q] != SNS_OK) rc = t->rc[(size_t)q];
        if (rc == SNS_OK) {
            for (int q = 0; q < t->n; ++q) p->base[q] = t->peers[(size_t)q]->base[q];
            p->host_sync = true;
            rc = peer_finish_connect(p);
        } else if (t->rc[(size_t)rank] != SNS_OK) {
            set_error(err);
        } else {
            set_error("team transport: another rank could not allocate its window");
        }
        t->barrier();
        if (rc != SNS_OK) return rc;
    }
    *out = t->peers[(size_t)rank];
    return SNS_OK;
}

// ---- in-process self-test and latency probe of the protocol -------------------------------------------------------------------
// nranks host threads of THIS process, each with its own window, stream and communicator end, wired to each other directly (one
// address space: no IPC): a ring of halo links with `halo_nodes` nodes per direction; per collective `reps` rounds with the payload
// checked every round, then `reps` timed rounds of the collective alone.  The streams run concurrently on the one GPU, so the figures
// are the protocol's launch and flag costs between concurrently running queues -- everything but the xGMI hop.
namespace {
__global__ void k_selftest_fill(int32_t n_nodes, double base, double* __restrict__ x) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < 4 * (int64_t)n_nodes) x[t] = base + (double)t;
}
__global__ void k_selftest_check(int32_t n_nodes, double base, const double* __restrict__ x, int* __restrict__ bad) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < 4 * (int64_t)n_nodes && x[t] != base + (double)t) atomicAdd(bad, 1);
}
// the product's read path: ghost block k of the vector straight from the receive window (GhostReader, as the level passes do);
// x is the vector itself (its tail is NOT read: poisoned by the caller)
__global__ __launch_bounds__(256) void k_selftest_check_ghost(GhostSrc g, int32_t first_node, int32_t n_nodes, double base,
                                                              const double* __restrict__ x, int* __restrict__ bad) {
    GhostReader gr;
    gr.begin(g);
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = t < 4 * (int64_t)n_nodes;
    const int32_t node = first_node + (int32_t)(in ? t >> 2 : 0);
    gr.arrive(g, in && node >= gr.n_own);
    if (in && gr.ptr(x, node)[t & 3] != base + (double)t) atomicAdd(bad, 1);
}

// a ring of halo links: rank r <-> r + 1 and r - 1 (one link with two ranks); the same owned nodes [0, halo) go to every
// neighbour, ghost block k = nodes [halo * (1 + k), halo * (2 + k))
void ring_plan(Plan& plan, int r, int nranks, int halo_nodes) {
    const int nn = nranks == 2 ? 1 : 2;
    plan.nbr.assign(1, (r + 1) % nranks);
    if (nn == 2) plan.nbr.push_back((r + nranks - 1) % nranks);
    std::sort(plan.nbr.begin(), plan.nbr.end());
    plan.send_ptr.assign(1, 0);
    plan.recv_ptr.assign(1, 0);
    plan.n_own = halo_nodes;
    for (int k = 0; k < nn; ++k) {
        for (int i = 0; i < halo_nodes; ++i) {
            plan.h_send_idx.push_back(i);
            plan.h_recv_idx.push_back(halo_nodes * (1 + k) + i);
        }
        plan.send_ptr.push_back((int32_t)plan.h_send_idx.size());
        plan.recv_ptr.push_back((int32_t)plan.h_recv_idx.size());
    }
}
}  // namespace

int peer_selftest(int device, int nranks, int halo_nodes, int reps, double us_out[3]) {
    // (one HIP hardware queue per rank besides the null stream's: with the runtime's default of 4 queues a fourth rank would share one,
    // and a kernel waiting for a flag would sit in front of the kernel that raises it)
    if (nranks < 2 || nranks > 3 || halo_nodes < 1 || reps < 1 || !us_out) {
        set_error("sns_peer_selftest: 2 or 3 ranks, halo_nodes >= 1, reps >= 1");
        return SNS_E_ARG;
    }
    Team team(nranks);                                           // (its barrier only)
    std::vector<Peer*> peers((size_t)nranks, nullptr);
    std::vector<PlanOffers> offers((size_t)nranks);
    std::vector<int> rcs((size_t)nranks, SNS_OK);
    std::vector<std::string> errs((size_t)nranks);
    std::vector<double> us((size_t)3 * nranks, 0.0);
    auto work = [&](int r) {
        auto fail = [&](int rc) { rcs[(size_t)r] = rc; errs[(size_t)r] = sns_last_error(); };
        hipStream_t st = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        Comm c;
        Plan plan;
        double *x = nullptr, *ar = nullptr, *ags = nullptr, *agr = nullptr;
        int* bad = nullptr;
        char hd[64];
        int rc = SNS_OK;
        // every rank passes every barrier, whatever its own state: a failing rank must not strand the others
        if (hipSetDevice(device) != hipSuccess) rc = SNS_E_HIP;
        if (rc == SNS_OK) rc = peer_create(device, r, nranks, (size_t)32 << 20, &peers[(size_t)r], hd);
        if (rc != SNS_OK) fail(rc);
        team.barrier();
        bool all_ok = true;
        for (int q = 0; q < nranks; ++q) all_ok = all_ok && peers[(size_t)q] != nullptr;
        if (!all_ok) return;
        Peer* pe = peers[(size_t)r];
        for (int q = 0; q < nranks; ++q) pe->base[q] = peers[(size_t)q]->base[q];
        if (peer_finish_connect(pe) != SNS_OK) fail(SNS_E_HIP);
        c.peer = pe;
        c.rank = r;
        c.nranks = nranks;
        ring_plan(plan, r, nranks, halo_nodes);
        const int nn = (int)plan.nbr.size();
        const int32_t n_local = halo_nodes * (1 + nn);
        if (rcs[(size_t)r] == SNS_OK && plan_upload(plan) != SNS_OK) fail(SNS_E_HIP);
        if (rcs[(size_t)r] == SNS_OK && peer_plan_offer(&c, plan, offers[(size_t)r]) != SNS_OK) fail(SNS_E_COMM);
        team.barrier();
        for (int q = 0; q < nranks; ++q) all_ok = all_ok && rcs[(size_t)q] == SNS_OK;
        if (!all_ok) return;
        PlanOffers t;
        for (int q = 0; q < nranks; ++q) t.all.insert(t.all.end(), offers[(size_t)q].mine.begin(), offers[(size_t)q].mine.end());
        if (peer_plan_connect(&c, plan, t) != SNS_OK) fail(SNS_E_COMM);
        const int32_t ag_count = 2048;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
            hipEventCreate(&e1) != hipSuccess || hipMalloc((void**)&x, 4 * (size_t)n_local * sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&ar, 8 * sizeof(double)) != hipSuccess || hipMalloc((void**)&ags, ag_count * sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&agr, (size_t)ag_count * nranks * sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&bad, sizeof(int)) != hipSuccess || hipMemset(bad, 0, sizeof(int)) != hipSuccess ||
            hipMemset(x, 0, 4 * (size_t)n_local * sizeof(double)) != hipSuccess)
            fail(SNS_E_HIP);
        team.barrier();
        for (int q = 0; q < nranks; ++q) all_ok = all_ok && rcs[(size_t)q] == SNS_OK;
        if (all_ok) {
            const unsigned gh = (unsigned)((4 * (int64_t)halo_nodes + 255) / 256);
            int hbad = 0;
            const bool verbose = std::getenv("SNS_PEER_SELFTEST_VERBOSE") != nullptr;
            bool go_on = true;
            for (int phase = 0; phase < 3 && go_on; ++phase) {
                if (verbose) { std::fprintf(stderr, "[peer selftest] rank %d: phase %d starts\n", r, phase); std::fflush(stderr); }
                const int n_it = 5 + reps;
                for (int it = 0; it < n_it && rcs[(size_t)r] == SNS_OK; ++it) {
                    const double tag = 1.0e6 * (it + 1);
                    int rc2 = SNS_OK;
                    if (phase == 0) {
                        // owned values = f(rank, round); after the exchange ghost block k must hold f(neighbour k, round): odd rounds
                        // through put + wait / unpack into the vector's tail, even rounds through put + the level passes' read
                        // path (straight from the window, the wait inside the consumer)
                        hipLaunchKernelGGL(k_selftest_fill, dim3(gh), dim3(256), 0, st, halo_nodes, tag + 1.0e3 * r, x);
                        if (it & 1) {
                            rc2 = comm_exchange(&c, plan, x, st);
                            for (int k = 0; k < nn; ++k)
                                hipLaunchKernelGGL(k_selftest_check, dim3(gh), dim3(256), 0, st, halo_nodes, tag + 1.0e3 * plan.nbr[(size_t)k],
                                                   x + 4 * (size_t)halo_nodes * (1 + k), bad);
                        } else {
                            rc2 = comm_put(&c, plan, x, st);
                            const GhostSrc g = comm_ghost_src(&c, plan);
                            for (int k = 0; k < nn; ++k)
                                hipLaunchKernelGGL(k_selftest_check_ghost, dim3(gh), dim3(256), 0, st, g, halo_nodes * (1 + k), halo_nodes,
                                                   tag + 1.0e3 * plan.nbr[(size_t)k], x, bad);
                        }
                    } else if (phase == 1) {
                        hipLaunchKernelGGL(k_selftest_fill, dim3(1), dim3(256), 0, st, 1, tag + (double)r, ar);      // 4 values
                        rc2 = comm_allreduce_sum(&c, ar, 4, st);
                        // sum over ranks of (tag + r + t) = nranks (tag + t) + nranks (nranks - 1) / 2
                        double expect[4], got[4];
                        if (it == n_it - 1) {
                            (void)hipMemcpyAsync(got, ar, sizeof(got), hipMemcpyDeviceToHost, st);
                            (void)hipStreamSynchronize(st);
                            for (int q = 0; q < 4; ++q) {
                                expect[q] = 0.0;
                                for (int rr = 0; rr < nranks; ++rr) expect[q] += tag + (double)rr + (double)q;
                                if (got[q] != expect[q]) ++hbad;
                            }
                        }
                    } else {
                        hipLaunchKernelGGL(k_selftest_fill, dim3((ag_count + 255) / 256), dim3(256), 0, st, ag_count / 4, tag + 1.0e3 * r, ags);
                        rc2 = comm_allgather(&c, ags, agr, ag_count, st);
                        for (int rr = 0; rr < nranks; ++rr)
                            hipLaunchKernelGGL(k_selftest_check, dim3((ag_count + 255) / 256), dim3(256), 0, st, ag_count / 4,
                                               tag + 1.0e3 * rr, agr + (size_t)rr * ag_count, bad);
                    }
                    if (rc2 != SNS_OK) fail(rc2);
                }
                // ... and the same number of rounds of the collective alone, timed
                team.barrier();
                (void)hipEventRecord(e0, st);
                for (int it = 0; it < reps && rcs[(size_t)r] == SNS_OK; ++it) {
                    const int rc2 = phase == 0 ? comm_exchange(&c, plan, x, st)
                                  : phase == 1 ? comm_allreduce_sum(&c, ar, 4, st) : comm_allgather(&c, ags, agr, ag_count, st);
                    if (rc2 != SNS_OK) fail(rc2);
                }
                (void)hipEventRecord(e1, st);
                (void)hipStreamSynchronize(st);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                us[(size_t)3 * r + phase] = 1.0e3 * ms / reps;
                if (rcs[(size_t)r] == SNS_OK && peer_check(&c) != SNS_OK) fail(SNS_E_COMM);
                if (verbose) {
                    std::fprintf(stderr, "[peer selftest] rank %d: phase %d done, rc %d, %.1f us per round\n", r, phase, rcs[(size_t)r],
                                 us[(size_t)3 * r + phase]);
                    std::fflush(stderr);
                }
                // all ranks leave the phase loop together: the verdicts are read between two barriers, while nobody writes one
                team.barrier();
                for (int q = 0; q < nranks; ++q) go_on = go_on && rcs[(size_t)q] == SNS_OK;
                team.barrier();
            }
            // the long forms, three verified rounds each: an all-reduce of 40 doubles (two launches of <= PEER_AR_MAX) and an
            // all-gather of more than three staging chunks with a ragged last one
            if (go_on) {
                const int64_t per = (int64_t)(pe->ag_doubles / (size_t)nranks);
                const int64_t big = 3 * per + 1000;
                double *ar40 = nullptr, *bs = nullptr, *br = nullptr;
                if (hipMalloc((void**)&ar40, 40 * sizeof(double)) != hipSuccess || hipMalloc((void**)&bs, (size_t)big * sizeof(double)) != hipSuccess ||
                    hipMalloc((void**)&br, (size_t)big * nranks * sizeof(double)) != hipSuccess)
                    fail(SNS_E_HIP);
                team.barrier();
                bool ok2 = true;
                for (int q = 0; q < nranks; ++q) ok2 = ok2 && rcs[(size_t)q] == SNS_OK;
                team.barrier();
                for (int it = 0; it < 3 && ok2 && rcs[(size_t)r] == SNS_OK; ++it) {
                    const double tag = 7.0e6 * (it + 1);
                    hipLaunchKernelGGL(k_selftest_fill, dim3(1), dim3(256), 0, st, 10, tag + (double)r, ar40);              // 40 values
                    int rc2 = comm_allreduce_sum(&c, ar40, 40, st);
                    double got[40];
                    (void)hipMemcpyAsync(got, ar40, sizeof(got), hipMemcpyDeviceToHost, st);
                    (void)hipStreamSynchronize(st);
                    for (int q = 0; q < 40; ++q) {
                        double expect = 0.0;
                        for (int rr = 0; rr < nranks; ++rr) expect += tag + (double)rr + (double)q;
                        if (got[q] != expect) ++hbad;
                    }
                    const unsigned gbig = (unsigned)((big + 255) / 256);
                    hipLaunchKernelGGL(k_selftest_fill, dim3(gbig), dim3(256), 0, st, (int32_t)(big / 4), tag + 1.0e3 * r, bs);
                    if (rc2 == SNS_OK) rc2 = comm_allgather(&c, bs, br, (int)(big / 4 * 4), st);
                    for (int rr = 0; rr < nranks; ++rr)
                        hipLaunchKernelGGL(k_selftest_check, dim3(gbig), dim3(256), 0, st, (int32_t)(big / 4), tag + 1.0e3 * rr,
                                           br + (size_t)rr * (size_t)(big / 4 * 4), bad);
                    (void)hipStreamSynchronize(st);
                    if (rc2 != SNS_OK) fail(rc2);
                    else if (peer_check(&c) != SNS_OK) fail(SNS_E_COMM);
                }
                team.barrier();                                          // (buffers of the round are read by nobody else: local frees)
                (void)hipFree(ar40); (void)hipFree(bs); (void)hipFree(br);
            }
            int dbad = 0;
            (void)hipMemcpy(&dbad, bad, sizeof(int), hipMemcpyDeviceToHost);
            dbad += hbad;
            if (dbad != 0 && rcs[(size_t)r] == SNS_OK) {
                set_error("peer self-test: rank " + std::to_string(r) + " received " + std::to_string(dbad) + " wrong values");
                fail(SNS_E_COMM);
            }
        }
        team.barrier();                                              // nobody frees a window a peer may still store into
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        plan_free(plan);
        (void)hipFree(x); (void)hipFree(ar); (void)hipFree(ags); (void)hipFree(agr); (void)hipFree(bad);
    };
    std::vector<std::thread> th;
    for (int r = 0; r < nranks; ++r) th.emplace_back(work, r);
    for (auto& t : th) t.join();
    for (int r = 0; r < nranks; ++r)
        if (peers[(size_t)r]) (void)peer_destroy(peers[(size_t)r]);
    for (int r = 0; r < nranks; ++r)
        if (rcs[(size_t)r] != SNS_OK) {
            set_error(errs[(size_t)r]);
            return rcs[(size_t)r];
        }
    for (int k = 0; k < 3; ++k) {
        us_out[k] = 0.0;
        for (int r = 0; r < nranks; ++r) us_out[k] = std::max(us_out[k], us[(size_t)3 * r + k]);
    }
    return SNS_OK;
}

// Link check of a CONNECTED communicator between its real ranks (collective; sns_peer_check_links): `rounds` all-reduces whose
// contributions depend on rank and round, `rounds` all-gathers of 4096 patterned doubles per rank and -- ADVICE r4 -- `rounds` halo
// exchanges over a ring plan between the real ranks (2048 nodes per direction, patterned payload), read alternately through the
// wait / unpack kernel and through the level passes' own path (GhostSrc: wait inside the consumer, ghost entries straight from the
// window), every value verified.  What it is for: the first contact of the windows with real xGMI links -- a visibility problem (a
// stale flag or payload read) shows up here as SNS_E_COMM with a count, not later as a solve that quietly diverges.
int peer_check_links(Peer* pe, int rounds) {
    if (!pe || !pe->connected || rounds < 1) { set_error("sns_peer_check_links: connected communicator, rounds >= 1"); return SNS_E_ARG; }
    CHIP(hipSetDevice(pe->device));
    Comm c;
    c.peer = pe;
    c.rank = pe->rank;
    c.nranks = pe->nranks;
    const int n = 4096, nr = pe->nranks, halo = 2048;
    hipStream_t st = nullptr;
    double *ar = nullptr, *ags = nullptr, *agr = nullptr, *x = nullptr, *off_s = nullptr, *off_r = nullptr;
    int* bad = nullptr;
    int rc = SNS_OK, wrong = 0;
    Plan plan;
    auto body = [&]() -> int {
        CHIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        CHIP(hipMalloc((void**)&ar, 8 * sizeof(double)));
        CHIP(hipMalloc((void**)&ags, n * sizeof(double)));
        CHIP(hipMalloc((void**)&agr, (size_t)n * nr * sizeof(double)));
        CHIP(hipMalloc((void**)&bad, sizeof(int)));
        CHIP(hipMemset(bad, 0, sizeof(int)));
        for (int it = 0; it < rounds; ++it) {
            const double tag = 3.0e6 * (it + 1);
            hipLaunchKernelGGL(k_selftest_fill, dim3(1), dim3(256), 0, st, 2, tag + (double)pe->rank, ar);             // 8 values
            CTRY(comm_allreduce_sum(&c, ar, 8, st));
            double got[8];
            CHIP(hipMemcpyAsync(got, ar, sizeof(got), hipMemcpyDeviceToHost, st));
            CHIP(hipStreamSynchronize(st));
            CTRY(peer_check(&c));
            for (int q = 0; q < 8; ++q) {
                double expect = 0.0;
                for (int r = 0; r < nr; ++r) expect += tag + (double)r + (double)q;
                if (got[q] != expect) ++wrong;
            }
            hipLaunchKernelGGL(k_selftest_fill, dim3((n + 255) / 256), dim3(256), 0, st, n / 4, tag + 1.0e3 * pe->rank, ags);
            CTRY(comm_allgather(&c, ags, agr, n, st));
            for (int r = 0; r < nr; ++r)
                hipLaunchKernelGGL(k_selftest_check, dim3((n + 255) / 256), dim3(256), 0, st, n / 4, tag + 1.0e3 * r, agr + (size_t)r * n, bad);
        }
        CHIP(hipStreamSynchronize(st));
        CTRY(peer_check(&c));
        if (nr >= 2) {
            // the halo ring; its offers travel through the all-gather that has just been verified
            ring_plan(plan, pe->rank, nr, halo);
            const int nn = (int)plan.nbr.size();
            CTRY(plan_upload(plan));
            PlanOffers t;
            const int rco = peer_plan_offer(&c, plan, t);
            const size_t LEN = (size_t)3 * nr + 1;
            if (rco != SNS_OK) t.mine.assign(LEN, -2.0);
            CHIP(hipMalloc((void**)&off_s, LEN * sizeof(double)));
            CHIP(hipMalloc((void**)&off_r, LEN * nr * sizeof(double)));
            CHIP(hipMemcpy(off_s, t.mine.data(), LEN * sizeof(double), hipMemcpyHostToDevice));
            CTRY(comm_allgather(&c, off_s, off_r, (int)LEN, st));
            CHIP(hipStreamSynchronize(st));
            CTRY(peer_check(&c));
            t.all.resize(LEN * nr);
            CHIP(hipMemcpy(t.all.data(), off_r, t.all.size() * sizeof(double), hipMemcpyDeviceToHost));
            if (rco != SNS_OK) return rco;
            for (double v : t.all)
                if (v == -2.0) { set_error("peer link check: a rank could not place the ring plan in its window"); return SNS_E_COMM; }
            CTRY(peer_plan_connect(&c, plan, t));
            const int32_t n_local = halo * (1 + nn);
            CHIP(hipMalloc((void**)&x, 4 * (size_t)n_local * sizeof(double)));
            const unsigned gh = (unsigned)((4 * (int64_t)halo + 255) / 256);
            for (int it = 0; it < rounds; ++it) {
                const double tag = 5.0e6 * (it + 1);
                CHIP(hipMemsetAsync(x, 0xff, 4 * (size_t)n_local * sizeof(double), st));              // poison (NaN pattern)
                hipLaunchKernelGGL(k_selftest_fill, dim3(gh), dim3(256), 0, st, halo, tag + 1.0e3 * pe->rank, x);
                if (it & 1) {
                    CTRY(comm_exchange(&c, plan, x, st));
                    for (int k = 0; k < nn; ++k)
                        hipLaunchKernelGGL(k_selftest_check, dim3(gh), dim3(256), 0, st, halo, tag + 1.0e3 * plan.nbr[(size_t)k],
                                           x + 4 * (size_t)halo * (1 + k), bad);
                } else {
                    CTRY(comm_put(&c, plan, x, st));
                    const GhostSrc g = comm_ghost_src(&c, plan);
                    for (int k = 0; k < nn; ++k)
                        hipLaunchKernelGGL(k_selftest_check_ghost, dim3(gh), dim3(256), 0, st, g, halo * (1 + k), halo,
                                           tag + 1.0e3 * plan.nbr[(size_t)k], x, bad);
                }
            }
            CHIP(hipStreamSynchronize(st));
            CTRY(peer_check(&c));
        }
        int dbad = 0;
        CHIP(hipMemcpy(&dbad, bad, sizeof(int), hipMemcpyDeviceToHost));
        wrong += dbad;
        return SNS_OK;
    };
    rc = body();
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    // (the ring plan's area goes back to the window; a peer still a round behind only ever writes rounds of THIS plan, which no
    // later plan mistakes for its own -- peer_plan_connect)
    plan_free(plan);
    (void)hipFree(ar); (void)hipFree(ags); (void)hipFree(agr); (void)hipFree(bad); (void)hipFree(x); (void)hipFree(off_s); (void)hipFree(off_r);
    if (rc == SNS_OK && wrong != 0) {
        set_error("peer link check: rank " + std::to_string(pe->rank) + " read " + std::to_string(wrong) + " wrong values in " +
                  std::to_string(rounds) + " rounds (stores of a peer not visible: window memory type / peer access)");
        rc = SNS_E_COMM;
    }
    return rc;
}

}  // namespace sns
