// Device side of the peer-window transport (sns_comm.h): the control area every rank exposes to its peers and the
// store / flag / bounded-wait primitives the collective kernels are made of.  Header-only so that a solver kernel can carry a
// collective inside it (k_reduce_final_bicg_peer: final reduction stage + all-reduce + scalar update in one launch).
//
// Memory model: the windows are fine-grained device memory.  A sender's payload stores are followed by a system-scope fence and
// a system-scope release store of a sequence number; a receiver polls that number with system-scope acquire loads and reads the
// payload with system-scope loads, never through a cache line left from an earlier round.  Every wait is bounded and reports
// through a host-mapped error word instead of spinning for ever.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sns {

constexpr int PEER_MAX_RANKS = 16;               // ranks of a peer communicator (one node)
constexpr int PEER_AR_MAX = 32;                  // doubles per all-reduce launch

// Control area at the start of every window -- same layout on every rank, slot [r] written by rank r only.
struct PeerCtl {
    unsigned long long ar_flag[PEER_MAX_RANKS];              // sequence number of rank r's latest all-reduce contribution
    unsigned long long ag_flag[PEER_MAX_RANKS];              // ... all-gather contribution
    double ar_slot[2][PEER_MAX_RANKS][PEER_AR_MAX];          // contributions, by parity of the sequence number
};

struct PeerArgs {                                // what a kernel needs to take part in an all-reduce
    unsigned long long seq;
    int rank, nranks;
    PeerCtl* const* ctl;                         // device array [nranks]: every rank's control area (own included)
    int* err;
    long long timeout_ticks;                     // wall_clock64() ticks (100 MHz)
};

__device__ __forceinline__ void peer_flag_store(unsigned long long* f, unsigned long long v) {
    __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool peer_flag_wait(const unsigned long long* f, unsigned long long v, long long timeout_ticks,
                                               int* err, int code) {
    if (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= v) return true;
    const long long t0 = (long long)wall_clock64();
    for (;;) {
        if (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= v) return true;
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return false;   // someone gave up already
        if ((long long)wall_clock64() - t0 > timeout_ticks) {
            __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}
__device__ __forceinline__ double peer_sys_load(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Sum v[0..count) over the ranks, result back in v (workgroup-shared or global memory; count <= PEER_AR_MAX).  Called by ALL
// threads of one workgroup of >= max(nranks, count) threads.  Contribution into every rank's slot table, flags, wait for
// everybody's, sum in rank order: the same bits on every rank.  Slots are double-buffered by the parity of seq: a rank can start
// round s + 2 only after round s + 1 completed, i.e. after every rank has contributed to s + 1, which each did after reading s.
__device__ __forceinline__ void peer_allreduce_block(double* v, int count, const PeerArgs& a) {
    const int tid = threadIdx.x, nth = blockDim.x, par = (int)(a.seq & 1ull);
    for (int idx = tid; idx < a.nranks * count; idx += nth) {
        const int r = idx / count, i = idx - r * count;
        a.ctl[r]->ar_slot[par][a.rank][i] = v[i];
    }
    __threadfence_system();
    __syncthreads();
    if (tid < a.nranks) {
        peer_flag_store(&a.ctl[tid]->ar_flag[a.rank], a.seq);
        (void)peer_flag_wait(&a.ctl[a.rank]->ar_flag[tid], a.seq, a.timeout_ticks, a.err, 2);
    }
    __syncthreads();
    if (tid < count) {
        double s = 0.0;
        for (int r = 0; r < a.nranks; ++r) s += peer_sys_load(&a.ctl[a.rank]->ar_slot[par][r][tid]);
        v[tid] = s;
    }
    __syncthreads();
}

}  // namespace sns
