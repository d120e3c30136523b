// Transports of sns_comm.h: RCCL over xGMI (product) and the in-process Team (tests).
#include "sns_comm.h"

#include <cstring>

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
    p.send_idx = p.recv_idx = nullptr;
    p.send_buf = p.recv_buf = nullptr;
}

int comm_exchange(Comm* c, const Plan& p, double* x, hipStream_t s) {
    if (!c || !c->active() || c->nranks <= 1) return SNS_OK;
    const int nn = (int)p.nbr.size();
    const int32_t ns = p.n_send(), nr = p.n_recv();
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

}  // namespace sns
