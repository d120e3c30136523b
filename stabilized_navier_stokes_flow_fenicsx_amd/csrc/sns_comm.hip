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
// Memory model: the windows are fine-grained device memory; a sender's payload stores are followed by a system-scope fence and
// a system-scope release store of the sequence flag; a receiver polls the flag with system-scope acquire loads and reads the
// payload with system-scope loads (never through a stale cache line of an earlier exchange).  Every wait is bounded by the
// communicator's timeout and reports through the mapped error word instead of spinning for ever.
namespace {

// halo put: the owned values listed in send_idx go straight into the neighbours' receive buffers; the last workgroup to
// finish raises this rank's flag in every neighbour's window (also across links that carry no payload in this direction: the
// flag is what keeps a rank from running two exchanges ahead of a neighbour, see comm_exchange)
__global__ __launch_bounds__(256) void k_peer_put(int32_t ns, int nn, const int32_t* __restrict__ send_idx,
                                                  const int32_t* __restrict__ send_ptr, const double* __restrict__ x,
                                                  double* const* __restrict__ put, unsigned long long* const* __restrict__ rflag,
                                                  unsigned long long seq, unsigned int* __restrict__ done) {
    __shared__ int32_t sp[PEER_MAX_RANKS + 1];
    __shared__ int last;
    const int tid = threadIdx.x;
    if (tid <= nn) sp[tid] = send_ptr[tid];
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * 256 + tid;        // one lane per half node: 16-byte loads and stores
    const int32_t i = (int32_t)(t >> 1);
    const int hh = (int)(t & 1);
    if (i < ns) {
        int k = 0;
        while (k + 1 < nn && i >= sp[k + 1]) ++k;
        const double2 v = *reinterpret_cast<const double2*>(x + 4 * (int64_t)send_idx[i] + 2 * hh);
        *reinterpret_cast<double2*>(put[k] + 4 * (int64_t)(i - sp[k]) + 2 * hh) = v;
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) last = (atomicAdd(done, 1u) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (last) {
        __threadfence_system();
        if (tid < nn) peer_flag_store(rflag[tid], seq);
        if (tid == 0) *done = 0u;
    }
}

// halo wait + unpack: every workgroup waits for all neighbours' flags, then scatters its part of the receive buffer
__global__ __launch_bounds__(256) void k_peer_wait_unpack(int32_t nr, int nn, const int32_t* __restrict__ recv_idx,
                                                          const double* __restrict__ recv_buf,
                                                          const unsigned long long* __restrict__ flag, unsigned long long seq,
                                                          double* __restrict__ x, int* err, long long timeout_ticks) {
    const int tid = threadIdx.x;
    if (tid < nn) (void)peer_flag_wait(flag + tid, seq, timeout_ticks, err, 1);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");             // every lane: nothing older than the flags is read below
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

// all-gather, put half: workgroup (b, r) copies its share of this rank's m doubles into rank r's staging area; the last
// workgroup of column r raises the flag there
__global__ __launch_bounds__(256) void k_peer_ag_put(const double* __restrict__ send, int64_t m, unsigned long long seq, int rank,
                                                     int64_t stage_doubles, double* const* __restrict__ ag,
                                                     PeerCtl* const* __restrict__ ctl, unsigned int* __restrict__ done) {
    __shared__ int last;
    const int tid = threadIdx.x, r = blockIdx.y;
    double* __restrict__ dst = ag[r] + (int64_t)(seq & 1ull) * stage_doubles + (int64_t)rank * m;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < m; i += (int64_t)gridDim.x * 256) dst[i] = send[i];
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
// all-gather, wait half: workgroup (b, r) waits for rank r's flag and copies r's m doubles from the own staging area
__global__ __launch_bounds__(256) void k_peer_ag_wait_copy(double* __restrict__ recv, int64_t count, int64_t off, int64_t m,
                                                           unsigned long long seq, int rank, int64_t stage_doubles,
                                                           double* const* __restrict__ ag, PeerCtl* const* __restrict__ ctl,
                                                           int* err, long long timeout_ticks) {
    const int tid = threadIdx.x, r = blockIdx.y;
    if (tid == 0) (void)peer_flag_wait(&ctl[rank]->ag_flag[r], seq, timeout_ticks, err, 3);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    const double* __restrict__ src = ag[rank] + (int64_t)(seq & 1ull) * stage_doubles + (int64_t)r * m;
    double* __restrict__ dst = recv + (int64_t)r * count + off;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < m; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
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

int plan_upload(Plan& p) {
    const size_t ns = (size_t)p.n_send(), nr = (size_t)p.n_recv();
    CHIP(hipMalloc((void**)&p.send_idx, std::max<size_t>(1, ns) * sizeof(int32_t)));
    CHIP(hipMalloc((void**)&p.recv_idx, std::max<size_t>(1, nr) * sizeof(int32_t)));
    if (ns) CHIP(hipMemcpy(p.send_idx, p.h_send_idx.data(), ns * sizeof(int32_t), hipMemcpyHostToDevice));
    if (nr) CHIP(hipMemcpy(p.recv_idx, p.h_recv_idx.data(), nr * sizeof(int32_t), hipMemcpyHostToDevice));
    CHIP(hipMalloc((void**)&p.send_buf, std::max<size_t>(1, 4 * ns) * sizeof(double)));
    CHIP(hipMalloc((void**)&p.recv_buf, std::max<size_t>(1, 4 * nr) * sizeof(double)));
    return SNS_OK;
}

void plan_free(Plan& p) {
    if (p.send_idx) (void)hipFree(p.send_idx);
    if (p.recv_idx) (void)hipFree(p.recv_idx);
    if (p.send_buf) (void)hipFree(p.send_buf);
    if (p.recv_buf) (void)hipFree(p.recv_buf);
    if (p.d_send_ptr) (void)hipFree(p.d_send_ptr);
    if (p.d_recv_ptr) (void)hipFree(p.d_recv_ptr);
    if (p.d_put) (void)hipFree(p.d_put);
    if (p.d_rflag) (void)hipFree(p.d_rflag);
    if (p.d_done) (void)hipFree(p.d_done);
    p.send_idx = p.recv_idx = nullptr;
    p.send_buf = p.recv_buf = nullptr;
    p.d_send_ptr = p.d_recv_ptr = nullptr;
    p.d_put = nullptr;
    p.d_rflag = nullptr;
    p.d_done = nullptr;
}

int comm_exchange(Comm* c, const Plan& p, double* x, hipStream_t s) {
    if (!c || !c->active() || c->nranks <= 1) return SNS_OK;
    const int nn = (int)p.nbr.size();
    const int32_t ns = p.n_send(), nr = p.n_recv();
    if (c->peer) {
        // Flow control without acknowledgements: the receive buffers are double-buffered by the parity of the plan's sequence
        // number, and this rank can start exchange s + 2 only after it has seen every neighbour's flag s + 1, which that
        // neighbour raised after (stream order) it had unpacked exchange s -- the buffer about to be overwritten.
        Peer* pe = c->peer;
        if (nn == 0) return SNS_OK;
        if (!p.d_put) { set_error("peer transport: halo plan was not connected"); return SNS_E_STATE; }
        CTRY(peer_check(c));
        const unsigned long long seq = ++p.seq;
        const int par = (int)(seq & 1ull);
        const unsigned gp = (unsigned)std::max<int64_t>(1, (2 * (int64_t)ns + 255) / 256);
        hipLaunchKernelGGL(k_peer_put, dim3(gp), dim3(256), 0, s, ns, nn, p.send_idx, p.d_send_ptr, x, p.d_put + (size_t)par * nn,
                           p.d_rflag, seq, p.d_done);
        const unsigned gu = (unsigned)std::max<int64_t>(1, (2 * (int64_t)nr + 255) / 256);
        hipLaunchKernelGGL(k_peer_wait_unpack, dim3(gu), dim3(256), 0, s, nr, nn, p.recv_idx, p.win_recv[par], p.win_flag, seq, x,
                           pe->err_dev, pe->timeout_ticks);
        return SNS_OK;
    }
    if (ns > 0)
        hipLaunchKernelGGL(k_pack, dim3((unsigned)((4 * (int64_t)ns + 255) / 256)), dim3(256), 0, s, ns, p.send_idx, x,
                           p.send_buf);
    if (c->nccl) {
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
    } else {
        Team* t = c->team;
        CHIP(hipStreamSynchronize(s));                       // my packed data is complete
        t->pub_buf[c->rank] = p.send_buf;
        t->pub_plan[c->rank] = &p;
        t->barrier();
        for (int k = 0; k < nn; ++k) {
            const int peer = p.nbr[k];
            const int32_t r0 = p.recv_ptr[k], r1 = p.recv_ptr[k + 1];
            if (r1 <= r0) continue;
            const Plan* pp = t->pub_plan[peer];
            int kk = -1;
            for (size_t q = 0; q < pp->nbr.size(); ++q)
                if (pp->nbr[q] == c->rank) kk = (int)q;
            if (kk < 0 || pp->send_ptr[kk + 1] - pp->send_ptr[kk] != r1 - r0) {
                set_error("team exchange: plans of rank " + std::to_string(c->rank) + " and " + std::to_string(peer) +
                          " disagree");
                t->barrier();
                return SNS_E_COMM;
            }
            CHIP(hipMemcpyAsync(p.recv_buf + 4 * (int64_t)r0, t->pub_buf[peer] + 4 * (int64_t)pp->send_ptr[kk],
                                4 * (size_t)(r1 - r0) * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        CHIP(hipStreamSynchronize(s));
        t->barrier();                                        // peers may now reuse their send buffers
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
            hipLaunchKernelGGL(k_peer_allreduce, dim3(1), dim3(256), 0, s, buf + off, m, peer_next_allreduce(pe));
        }
        return SNS_OK;
    }
    if (c->nccl) {                                           // also with one rank: keeps the RCCL path exercised
        CNCCL(ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, c->nccl, s));
        return SNS_OK;
    }
    Team* t = c->team;
    std::vector<double>& mine = t->slots[c->rank];
    mine.resize(count);
    CHIP(hipMemcpyAsync(mine.data(), buf, count * sizeof(double), hipMemcpyDeviceToHost, s));
    CHIP(hipStreamSynchronize(s));
    t->barrier();
    std::vector<double> sum(count, 0.0);
    for (int r = 0; r < t->n; ++r)                           // fixed order => identical bits on every rank
        for (int i = 0; i < count; ++i) sum[i] += t->slots[r][i];
    t->barrier();
    CHIP(hipMemcpyAsync(buf, sum.data(), count * sizeof(double), hipMemcpyHostToDevice, s));
    CHIP(hipStreamSynchronize(s));                           // `sum` is a stack buffer
    return SNS_OK;
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
            const unsigned long long seq = ++pe->ag_seq;
            const unsigned gb = (unsigned)std::min<int64_t>(64, std::max<int64_t>(1, (m + 2047) / 2048));
            hipLaunchKernelGGL(k_peer_ag_put, dim3(gb, pe->nranks), dim3(256), 0, s, send + off, m, seq, pe->rank,
                               (int64_t)pe->ag_doubles, pe->d_ag, pe->d_ctl, pe->d_done);
            hipLaunchKernelGGL(k_peer_ag_wait_copy, dim3(gb, pe->nranks), dim3(256), 0, s, recv, (int64_t)count, off, m, seq,
                               pe->rank, (int64_t)pe->ag_doubles, pe->d_ag, pe->d_ctl, pe->err_dev, pe->timeout_ticks);
        }
        return SNS_OK;
    }
    if (c->nccl) {
        CNCCL(ncclAllGather(send, recv, count, ncclDouble, c->nccl, s));
        return SNS_OK;
    }
    Team* t = c->team;
    std::vector<double>& mine = t->slots[c->rank];
    mine.resize(count);
    CHIP(hipMemcpyAsync(mine.data(), send, count * sizeof(double), hipMemcpyDeviceToHost, s));
    CHIP(hipStreamSynchronize(s));
    t->barrier();
    std::vector<double> all((size_t)count * t->n);
    for (int r = 0; r < t->n; ++r) std::memcpy(all.data() + (size_t)r * count, t->slots[r].data(), count * sizeof(double));
    t->barrier();
    CHIP(hipMemcpyAsync(recv, all.data(), all.size() * sizeof(double), hipMemcpyHostToDevice, s));
    CHIP(hipStreamSynchronize(s));
    return SNS_OK;
}

// ---- peer transport: host side ----------------------------------------------------------------------------------------------
namespace {
constexpr size_t PEER_CTL_BYTES = (sizeof(PeerCtl) + 4095) / 4096 * 4096;
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
}  // namespace

int peer_create(int device, int rank, int nranks, size_t window_bytes, Peer** out, char handle_out[64]) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t size");
    if (!out || !handle_out || nranks < 1 || nranks > PEER_MAX_RANKS || rank < 0 || rank >= nranks) {
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
    // control area | all-gather staging (2 parities, a quarter of the window each at most 32 MiB) | plan area
    p->ag_off = PEER_CTL_BYTES;
    const size_t ag_bytes = std::min<size_t>((size_t)32 << 20, window_bytes / 4) / (8 * (size_t)nranks) * (8 * (size_t)nranks);
    p->ag_doubles = ag_bytes / 8;
    p->bump = align_up(p->ag_off + 2 * ag_bytes, 4096);
    CHIP(hipHostMalloc((void**)&p->err_host, sizeof(int), hipHostMallocMapped));
    *p->err_host = 0;
    CHIP(hipHostGetDevicePointer((void**)&p->err_dev, p->err_host, 0));
    CHIP(hipMalloc((void**)&p->d_done, PEER_MAX_RANKS * sizeof(unsigned int)));
    CHIP(hipMemset(p->d_done, 0, PEER_MAX_RANKS * sizeof(unsigned int)));
    double ms = 20000.0;
    if (const char* t = std::getenv("SNS_PEER_TIMEOUT_MS")) ms = std::max(1.0, std::atof(t));
    p->timeout_ticks = (long long)(ms * 1.0e5);                                  // wall_clock64(): 100 MHz
    hipIpcMemHandle_t hd;
    CHIP(hipIpcGetMemHandle(&hd, w));
    std::memcpy(handle_out, &hd, 64);
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

int peer_destroy(Peer* p) {
    if (!p) return SNS_OK;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < p->nranks; ++r)
        if (p->mapped[r] && p->base[r]) (void)hipIpcCloseMemHandle(p->base[r]);
    if (p->base[p->rank]) (void)hipFree(p->base[p->rank]);
    if (p->d_ctl) (void)hipFree(p->d_ctl);
    if (p->d_ag) (void)hipFree(p->d_ag);
    if (p->d_done) (void)hipFree(p->d_done);
    if (p->err_host) (void)hipHostFree(p->err_host);
    delete p;
    return SNS_OK;
}

PeerArgs peer_next_allreduce(Peer* p) {
    PeerArgs a;
    a.seq = ++p->ar_seq;
    a.rank = p->rank;
    a.nranks = p->nranks;
    a.ctl = p->d_ctl;
    a.err = p->err_dev;
    a.timeout_ticks = p->timeout_ticks;
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
    t.mine.assign((size_t)3 * nr_ranks, -1.0);
    const size_t flag_off = align_up(pe->bump, 256);
    const size_t rbytes = align_up(std::max<size_t>(32, (size_t)p.n_recv() * 32), 256);
    const size_t r0 = align_up(flag_off + (size_t)std::max(1, nn) * 8, 256), r1 = r0 + rbytes, end = r1 + rbytes;
    if (end > pe->bytes) {
        set_error("peer transport: window of " + std::to_string(pe->bytes >> 20) + " MiB exhausted (sns_peer_create window_bytes)");
        return SNS_E_COMM;
    }
    pe->bump = end;
    char* own = pe->base[pe->rank];
    p.win_flag = reinterpret_cast<unsigned long long*>(own + flag_off);
    p.win_recv[0] = reinterpret_cast<double*>(own + r0);
    p.win_recv[1] = reinterpret_cast<double*>(own + r1);
    p.seq = 0;
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
    if (t.all.size() != (size_t)3 * nr_ranks * nr_ranks) { set_error("peer transport: bad offer table"); return SNS_E_COMM; }
    if (nn > PEER_MAX_RANKS) { set_error("peer transport: too many neighbours"); return SNS_E_ARG; }
    std::vector<double*> put((size_t)2 * std::max(1, nn), nullptr);
    std::vector<unsigned long long*> rflag((size_t)std::max(1, nn), nullptr);
    for (int k = 0; k < nn; ++k) {
        const int j = p.nbr[(size_t)k];
        const double* o = t.all.data() + ((size_t)j * nr_ranks + pe->rank) * 3;   // what rank j offers to this rank
        if (o[0] < 0.0 || o[1] < 0.0 || o[2] < 0.0) {
            set_error("peer transport: rank " + std::to_string(j) + " does not list rank " + std::to_string(pe->rank) + " as a neighbour");
            return SNS_E_COMM;
        }
        put[(size_t)k] = reinterpret_cast<double*>(pe->base[j] + (size_t)o[0]);
        put[(size_t)nn + k] = reinterpret_cast<double*>(pe->base[j] + (size_t)o[1]);
        rflag[(size_t)k] = reinterpret_cast<unsigned long long*>(pe->base[j] + (size_t)o[2]);
    }
    std::vector<int32_t> sp(p.send_ptr), rp(p.recv_ptr);
    if (sp.empty()) sp.assign(1, 0);
    if (rp.empty()) rp.assign(1, 0);
    CHIP(hipMalloc((void**)&p.d_send_ptr, sp.size() * sizeof(int32_t)));
    CHIP(hipMalloc((void**)&p.d_recv_ptr, rp.size() * sizeof(int32_t)));
    CHIP(hipMemcpy(p.d_send_ptr, sp.data(), sp.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    CHIP(hipMemcpy(p.d_recv_ptr, rp.data(), rp.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    CHIP(hipMalloc((void**)&p.d_put, put.size() * sizeof(double*)));
    CHIP(hipMalloc((void**)&p.d_rflag, rflag.size() * sizeof(unsigned long long*)));
    CHIP(hipMemcpy(p.d_put, put.data(), put.size() * sizeof(double*), hipMemcpyHostToDevice));
    CHIP(hipMemcpy(p.d_rflag, rflag.data(), rflag.size() * sizeof(unsigned long long*), hipMemcpyHostToDevice));
    CHIP(hipMalloc((void**)&p.d_done, sizeof(unsigned int)));
    CHIP(hipMemset(p.d_done, 0, sizeof(unsigned int)));
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
        const int nn = nranks == 2 ? 1 : 2;
        plan.nbr.assign(1, (r + 1) % nranks);
        if (nn == 2) plan.nbr.push_back((r + nranks - 1) % nranks);
        std::sort(plan.nbr.begin(), plan.nbr.end());
        plan.send_ptr.assign(1, 0);
        plan.recv_ptr.assign(1, 0);
        for (int k = 0; k < nn; ++k) {
            for (int i = 0; i < halo_nodes; ++i) {
                plan.h_send_idx.push_back(i);                                    // the same owned nodes go to every neighbour
                plan.h_recv_idx.push_back(halo_nodes * (1 + k) + i);             // ghost block k
            }
            plan.send_ptr.push_back((int32_t)plan.h_send_idx.size());
            plan.recv_ptr.push_back((int32_t)plan.h_recv_idx.size());
        }
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
                        // owned values = f(rank, round); after the exchange ghost block k must hold f(neighbour k, round)
                        hipLaunchKernelGGL(k_selftest_fill, dim3(gh), dim3(256), 0, st, halo_nodes, tag + 1.0e3 * r, x);
                        rc2 = comm_exchange(&c, plan, x, st);
                        for (int k = 0; k < nn; ++k)
                            hipLaunchKernelGGL(k_selftest_check, dim3(gh), dim3(256), 0, st, halo_nodes, tag + 1.0e3 * plan.nbr[(size_t)k],
                                               x + 4 * (size_t)halo_nodes * (1 + k), bad);
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
// contributions depend on rank and round and `rounds` all-gathers of 4096 patterned doubles per rank, every value verified.  What it
// is for: the first contact of the windows with real xGMI links -- a visibility problem (a stale flag or payload read) shows up here
// as SNS_E_COMM with a count, not later as a solve that quietly diverges.
int peer_check_links(Peer* pe, int rounds) {
    if (!pe || !pe->connected || rounds < 1) { set_error("sns_peer_check_links: connected communicator, rounds >= 1"); return SNS_E_ARG; }
    CHIP(hipSetDevice(pe->device));
    Comm c;
    c.peer = pe;
    c.rank = pe->rank;
    c.nranks = pe->nranks;
    const int n = 4096, nr = pe->nranks;
    hipStream_t st = nullptr;
    double *ar = nullptr, *ags = nullptr, *agr = nullptr;
    int* bad = nullptr;
    int rc = SNS_OK, wrong = 0;
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
        int dbad = 0;
        CHIP(hipMemcpy(&dbad, bad, sizeof(int), hipMemcpyDeviceToHost));
        wrong += dbad;
        return SNS_OK;
    };
    rc = body();
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    (void)hipFree(ar); (void)hipFree(ags); (void)hipFree(agr); (void)hipFree(bad);
    if (rc == SNS_OK && wrong != 0) {
        set_error("peer link check: rank " + std::to_string(pe->rank) + " read " + std::to_string(wrong) + " wrong values in " +
                  std::to_string(rounds) + " rounds (stores of a peer not visible: window memory type / peer access)");
        rc = SNS_E_COMM;
    }
    return rc;
}

}  // namespace sns
