// Device side of the peer-window transport (sns_comm.h): the control area every rank exposes to its peers, the
// store / flag / bounded-wait primitives the collective kernels are made of, and the descriptor a solver kernel takes to read the
// ghost entries of its input vector STRAIGHT from the receive window (GhostSrc).  Header-only so that a solver kernel can carry a
// collective inside it (k_reduce_final_bicg_peer: final reduction stage + all-reduce + scalar update in one launch; the level
// passes of a partitioned handle: wait for the neighbours' halo inside the pass itself).
//
// Memory model: the windows are fine-grained device memory.  A sender's payload stores are followed by a system-scope fence and
// a system-scope release store of a sequence number.  A receiver polls that number with system-scope acquire loads, then EVERY
// lane of the wave executes a system-scope acquire fence (which invalidates whatever the caches of this GPU still hold of the
// window from an earlier round) and reads the payload with plain loads issued after the fence.  Nothing reads a receive buffer of
// round s before it has seen the flags of ALL its neighbours for round s, so no cache line of that buffer can be (re)filled
// between the fence and the data's arrival.  Every wait is bounded and reports through a host-mapped error word instead of
// spinning for ever.
//
// Sequence numbers live in DEVICE memory and are advanced by the kernels themselves (the put raises its plan's round, the
// consumer reads it): no kernel argument changes from one round to the next, so a whole solver iteration can be a hipGraph.
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
    unsigned long long* seq;                     // device word of THIS rank: rounds completed; the kernel runs round *seq + 1
    int rank, nranks;
    PeerCtl* const* ctl;                         // device array [nranks]: every rank's control area (own included)
    int* err;
    long long timeout_ticks;                     // wall_clock64() ticks (100 MHz)
    int phase;                                   // 0: the whole collective in this launch (flags + waits);  in-process team
                                                 // transport (the ranks' kernels are serialised on one queue, the host barrier
                                                 // stands in for the flags): 1 = contribute only, 2 = sum only
};

// Where a pass over a partitioned level finds the ghost entries of its input vector: node j >= n_own of the vector lives at
// win[parity] + 4 * j (the pointers are pre-offset by -4 * n_own doubles: the receive buffer holds the ghost nodes in their
// local order), parity = *seq & 1 with *seq the plan's round as left by the put kernel in front of this pass.  nn > 0: the wave
// that meets its first ghost column waits (bounded) for the nn neighbours' arrival flags of that round.  win[0] == nullptr: no
// window -- the ghost entries sit in the vector's own tail (serial handles, RCCL).
struct GhostSrc {
    const double* win[2] = {nullptr, nullptr};
    int32_t n_own = 0x7fffffff;
    int nn = 0;
    const unsigned long long* seq = nullptr;
    const unsigned long long* flag = nullptr;
    int* err = nullptr;
    long long timeout_ticks = 0;
};

// The two halves of an all-gather carried by solver kernels (the right-hand side of a partitioned run's replicated tail): the
// kernel that PRODUCES a rank's piece stores it into every rank's staging area and raises the flags (AgPut), the kernel that
// CONSUMES the gathered vector waits for every rank's flag and reads the pieces from its own staging area (AgGet) -- no
// all-gather launch of its own.  Same staging areas, flags and round counter as the stand-alone all-gather (sns_comm.hip).
struct AgPut {
    double* const* ag = nullptr;                 // device array [nranks]: the ranks' staging areas (nullptr: off)
    PeerCtl* const* ctl = nullptr;
    const unsigned long long* seq = nullptr;     // round = *seq + 1 (stored by the consumer)
    unsigned int* done = nullptr;
    int rank = 0, nranks = 1, flags = 0;         // flags 0 (team transport): the host barrier orders the halves
    long long stage_doubles = 0, slot_doubles = 0;
};
struct AgGet {
    const double* stage = nullptr;               // this rank's staging area (nullptr: off)
    const PeerCtl* ctl = nullptr;                // this rank's control area
    unsigned long long* seq = nullptr;
    unsigned int* done = nullptr;
    int nranks = 1, flags = 0;
    long long stage_doubles = 0;
    int* err = nullptr;
    long long timeout_ticks = 0;
};

// The put half of a halo exchange carried by the kernel that PRODUCES the vector (round 5): every lane that writes entry (row, c)
// of the vector also stores it into the receive windows of the neighbours the row is sent to (sr_ptr / sr_dst: per owned row its
// send entries, neighbour k << 27 | slot in k's segment), and the last workgroup to finish raises the flags and stores the round --
// what k_halo_put does in a launch of its own, minus the launch.  sr_ptr == nullptr: off.
struct PutDst {
    const int32_t* sr_ptr = nullptr;
    const int32_t* sr_dst = nullptr;
    double* const* put = nullptr;                // [2][nn] remote payload addresses, by parity of the round
    unsigned long long* const* rflag = nullptr;  // [nn] remote flags (nullptr, team transport: the host barrier orders the rounds)
    unsigned long long* seq = nullptr;           // the plan's round (device word): this kernel runs round *seq + 1 and stores it
    unsigned int* done = nullptr;
    // workgroups of the producing kernel that hold a sent row, for its two slot groupings (64 / 8 block slots per workgroup): ONLY
    // those count themselves -- every workgroup of a 3400-workgroup pass adding to one counter costs ~30 us of same-address atomics
    const unsigned int* expect = nullptr;
    int nn = 0;
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

// The ghost side of a pass, per wave: `begin` reads the round once (uniform), `ptr` selects the source of a column's x block,
// `arrive` is called (by all active lanes of the wave, any control flow) before the first ghost entry is read.
struct GhostReader {
    const double* xg;                            // pre-offset receive buffer of this round (nullptr: no window)
    int32_t n_own;
    unsigned long long sq;
    bool waited;
    __device__ __forceinline__ void begin(const GhostSrc& g) {
        n_own = g.n_own;
        xg = nullptr;
        sq = 0ull;
        waited = true;
        if (g.win[0]) {
            sq = *g.seq;
            xg = g.win[sq & 1ull];
            waited = g.nn == 0;
        }
    }
    __device__ __forceinline__ const double* ptr(const double* x, int32_t col) const {
        return (col >= n_own ? xg : x) + 4 * (int64_t)col;
    }
    // any: does some lane of this wave need a ghost entry now?  (lanes vote with their own column ids)
    __device__ __forceinline__ void arrive(const GhostSrc& g, bool mine_is_ghost) {
        if (waited) return;
        if (!__any(mine_is_ghost)) return;
        // (every active lane polls every flag -- same address, one request per wave: the lanes that are active here may be any)
        for (int k = 0; k < g.nn; ++k) (void)peer_flag_wait(g.flag + k, sq, g.timeout_ticks, g.err, 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");         // every lane: nothing older than the flags is read below
        waited = true;
    }
};

__device__ __forceinline__ unsigned long long put_begin(const PutDst& pd) {
    return pd.sr_ptr ? __hip_atomic_load(pd.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull : 0ull;
}
// entry (row, c) of the produced vector -> the neighbours' receive windows; returns whether anything was stored
__device__ __forceinline__ bool put_store(const PutDst& pd, unsigned long long seq, int32_t row, int c, double val) {
    const int32_t e0 = pd.sr_ptr[row], e1 = pd.sr_ptr[row + 1];
    for (int32_t e = e0; e < e1; ++e) {
        const int32_t d = pd.sr_dst[e];
        pd.put[(size_t)(seq & 1ull) * pd.nn + (d >> 27)][4 * (int64_t)(d & 0x7ffffff) + c] = val;
    }
    return e1 > e0;
}
// called by ALL threads of every workgroup after their stores (s_last: a workgroup-shared int)
// (grouping: 0 = 64 block slots per workgroup, 1 = 8)
__device__ __forceinline__ void put_finish(const PutDst& pd, unsigned long long seq, bool stored, int* s_last, int grouping) {
    if (!__syncthreads_or(stored ? 1 : 0)) return;         // (uniform over the workgroup)
    if (pd.rflag) __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *s_last = (atomicAdd(pd.done, 1u) + 1u == pd.expect[grouping]) ? 1 : 0;
    __syncthreads();
    if (*s_last) {
        if (pd.rflag) {
            __threadfence_system();
            if ((int)threadIdx.x < pd.nn) peer_flag_store(pd.rflag[threadIdx.x], seq);
        }
        if (threadIdx.x == 0) {
            *pd.done = 0u;
            __hip_atomic_store(pd.seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Sum v[0..count) over the ranks, result back in v (workgroup-shared or global memory; count <= PEER_AR_MAX).  Called by ALL
// threads of one workgroup of >= max(nranks, count) threads.  Contribution into every rank's slot table, flags, wait for
// everybody's, sum in rank order: the same bits on every rank.  Slots are double-buffered by the parity of the round: a rank can
// start round s + 2 only after round s + 1 completed, i.e. after every rank has contributed to s + 1, which each did after reading s.
// a.phase 1 / 2 (team transport): the two halves as separate launches with a host barrier between them.
__device__ __forceinline__ void peer_allreduce_block(double* v, int count, const PeerArgs& a) {
    const int tid = threadIdx.x, nth = blockDim.x;
    const unsigned long long seq = *a.seq + 1ull;
    const int par = (int)(seq & 1ull);
    if (a.phase != 2) {
        for (int idx = tid; idx < a.nranks * count; idx += nth) {
            const int r = idx / count, i = idx - r * count;
            a.ctl[r]->ar_slot[par][a.rank][i] = v[i];
        }
    }
    if (a.phase == 1) return;
    if (a.phase == 0) {
        __threadfence_system();
        __syncthreads();
        if (tid < a.nranks) {
            peer_flag_store(&a.ctl[tid]->ar_flag[a.rank], seq);
            (void)peer_flag_wait(&a.ctl[a.rank]->ar_flag[tid], seq, a.timeout_ticks, a.err, 2);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    if (tid < count) {
        double s = 0.0;
        for (int r = 0; r < a.nranks; ++r) s += peer_sys_load(&a.ctl[a.rank]->ar_slot[par][r][tid]);
        v[tid] = s;
    }
    __syncthreads();
    if (tid == 0) *a.seq = seq;
}

}  // namespace sns
