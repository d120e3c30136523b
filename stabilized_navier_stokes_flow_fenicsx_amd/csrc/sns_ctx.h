// Shared internals of the C-ABI layer (round 5: csrc/sns_api.hip split into setup / cycle / Krylov / C-ABI translation units):
// the context behind an sns_handle, the error / allocation helpers, the launch helpers of the level passes, the handle-side
// predicates of the cycle (which gather the facts and ask csrc/sns_policy.h for every number) and the functions the translation
// units call across each other.  Not installed; the public surface is include/sns.h.
#pragma once
#include <rccl/rccl.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <vector>

#include "sns_comm.h"
#include "sns_harness.h"
#include "sns_internal.h"
#include "sns_kernels.h"
#include "sns_policy.h"

namespace sns {

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) {                                                                           \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + ":" +         \
                      std::to_string(__LINE__));                                                          \
            return SNS_E_HIP;                                                                             \
        }                                                                                                 \
    } while (0)

#define NCCL_TRY(expr)                                                                                    \
    do {                                                                                                  \
        ncclResult_t _e = (expr);                                                                         \
        if (_e != ncclSuccess) {                                                                          \
            set_error(std::string(#expr) + ": " + ncclGetErrorString(_e));                                \
            return SNS_E_COMM;                                                                            \
        }                                                                                                 \
    } while (0)

#define SNS_TRY(expr)                                                                                     \
    do {                                                                                                  \
        int _r = (expr);                                                                                  \
        if (_r != SNS_OK) return _r;                                                                      \
    } while (0)

template <class T>
static inline int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIP_TRY(hipMalloc((void**)p, count * sizeof(T)));
    return SNS_OK;
}

template <class T>
static inline int dev_upload(T** p, const std::vector<T>& v, hipStream_t) {
    SNS_TRY(dev_alloc(p, v.size()));
    if (!v.empty()) HIP_TRY(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return SNS_OK;
}

}  // namespace sns

using namespace sns;

struct sns_ctx {
    sns_options opt;
    int device = 0;
    hipStream_t stream = nullptr;
    // mesh (dim 3: tets; dim 2: triangles in a stride-4 connectivity, z component a Dirichlet dof)
    int dim = 3;
    int32_t n = 0, n_owned = 0;
    int64_t n_global_fine = 0;                   // fine-level rows over all ranks (set when the hierarchy is built)
    int64_t n_global_l1 = 0;                     // level-1 rows over all ranks (the sweep schedule must be the same on every rank)
    int64_t E = 0;
    int32_t* tets = nullptr;
    double* pts = nullptr;
    uint8_t* bc_mask = nullptr;
    double* bc_val = nullptr;
    // assembly maps
    int64_t *nt_ptr = nullptr, *c_ptr = nullptr;
    int32_t *nt_idx = nullptr, *c_idx = nullptr;
    int32_t* od_order = nullptr;       // off-diagonal slots, locally sorted by contribution count (scratch-free assembly)
    double* gext = nullptr;            // Dirichlet data extended by zero (the state the Stokes lifting term is taken at)
    int64_t n_od = 0;
    double *Ke = nullptr, *Fe = nullptr;
    // operator hierarchy; levels[0] is the assembled fine operator.  A deque: references to a level stay valid
    // while coarser levels are appended (a vector reallocation under a live Level& once handed a kernel dangling
    // pointers)
    std::deque<Level> levels;
    std::vector<int32_t*> slot_row;              // per level
    std::vector<uint8_t*> empty_c;               // per level (coarse side), level l -> empty flags of level l+1
    std::vector<double*> pong;                   // per level smoother ping-pong buffer
    int* d_piv = nullptr;
    int* d_sing = nullptr;
    FormVariant fv;                              // sns_set_form_variant (diagnostic; default = the reference's form)
    bool has_matrix = false, pc_ready = false;
    int pc_setups = 0;
    // hipGraph of the launch-bound coarse part of the V-cycle (levels >= graph_level; serial runs only)
    hipStream_t cap_stream = nullptr;
    hipStream_t gj_stream = nullptr;              // second stream of the dense coarsest level's elimination (bulk updates beside the pivot chain)
    hipGraphExec_t coarse_graph = nullptr;
    std::vector<double> graph_sig;                // (omega per level, nu, nu_coarse, f32) the graph was captured with
    bool graph_disabled = false;
    int matrix_form = -1;
    int est_form = -1;                           // form of the matrix the levels' spectral estimates were last taken from
    double est_re = 0.0;                         // ... and its Reynolds number
    // reductions
    double* partial = nullptr;                   // [max(65536*8, n/32)]
    double* partial2 = nullptr;                  // second stage of long reductions
    double* d_scal = nullptr;                    // [256]
    double* h_scal = nullptr;                    // pinned [256]
    // Krylov workspace
    std::vector<double*> kv;                     // allocated vectors (4*n each)
    double* gm_V = nullptr;                      // (m+1) * ld
    double* gm_Z = nullptr;                      // m * ld
    int gm_m = 0;
    double* d_h = nullptr;                       // device Hessenberg column scratch [3*(m+2)]
    // Newton workspace
    double *nw_F = nullptr, *nw_y = nullptr, *nw_w = nullptr, *nw_t = nullptr;
    sns_timings tm{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_it = nullptr;
    // debug counters of the last Krylov solve (sns_get_counters): host syncs, all-reduces, halo exchanges
    int64_t ctr_host_syncs = 0, ctr_allreduce = 0, ctr_exchange = 0;
    int64_t last_ctr[3] = {0, 0, 0};                 // snapshot at the end of the last Krylov solve
    int bnd_dot_blocks = 0;
    int dot_partials = 0;                            // partial sums the last fused SpMV+dot pass left in h->partial
    // multi-GPU, level 0: owned rows with at least one ghost column (the only rows that must wait for the halo)
    int32_t* bnd_rows = nullptr;
    uint8_t* bnd_flag = nullptr;
    int32_t n_bnd = 0;
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_x = nullptr, ev_side = nullptr;
    bool no_overlap = false;
    bool team_overlap = false;                       // SNS_TEAM_OVERLAP: the team transport takes the two-stream path too (tests)
    double* arn_V = nullptr;                          // Arnoldi basis of the damping estimate, 9 vectors of the largest level >= ... asked for
    size_t arn_cap = 0;
    // the put half of a halo exchange carried by the kernel that produced the vector (PutDst, round 5): first_put_carried -- the
    // Krylov kernel that ran the fine level's first sweep put it too; child_put_carried -- the restriction put the next level's first
    // sweep; put_pending -- the last kernel of a window cycle put its result (the vector named), the next exchange of exactly that
    // vector is comm_put_carried; pc_then_op -- the caller of pc_apply promises that an operator application of the result follows
    // (only then may the fine level's last kernel carry the put: a round nobody consumes would void the windows' flow control)
    bool fuse_puts = std::getenv("SNS_NO_CARRIED_PUT") == nullptr;      // (A/B switch of the harnesses; not an option)
    bool first_put_carried = false, child_put_carried = false, pc_then_op = false;
    const double* put_pending = nullptr;
    bool first_sweep_done = false;                   // the V-cycle's fine-level first sweep was done by the Krylov kernel that wrote its input
    bool r3_estimates = false;                       // SNS_R3_SPECTRAL_ESTIMATE (tests of the retry path): round 3's policy -- spectral
                                                     // estimates every 4th setup whatever the operator (first Jacobians on the Stokes estimate)
    double damping_backoff = 1.0;                    // < 1 after a failed AMG-preconditioned solve: all level dampings scaled (krylov())
    int64_t ctr_retries = 0;                         // damping retries since sns_reset_timings
    int last_first_reason = 0;                       // reason of the FIRST attempt of the last solve (0 = no retry happened)
    std::unique_ptr<Comm> comm;
    // distributed coarsest level: global dense inverse, replicated on every rank
    int cg_maxn = 0;                              // padded owned coarsest nodes per rank
    int cg_N = 0;                                 // 4 * nranks * cg_maxn (0 = not used)
    std::vector<int> cg_counts;                   // owned coarsest nodes of every rank
    // multi-GPU: replicated tail of the hierarchy.  levels[rep_level] is a copy of the GLOBAL operator of level
    // rep_level-1 held by every rank (all-gathered values); it and everything below is cycled redundantly on every
    // rank without any exchange.  0 = none.
    int rep_level = 0;
    int32_t rep_maxn = 0, rep_NG = 0, rep_off = 0;
    int64_t rep_maxnz = 0;
    int32_t* rep_valmap = nullptr;                // [nranks*maxnz] gathered slot -> slot of the replicated level (-1: padding)
    int32_t* rep_rowmap = nullptr;                // [NG] row of the replicated level -> gathered row (rank*maxn + i)
    double *rep_vsend = nullptr, *rep_vrecv = nullptr, *rep_bsend = nullptr, *rep_brecv = nullptr;
    int64_t *rep_doff = nullptr, *rep_dcnt = nullptr;   // [nranks] doubles: where rank r's right-hand side goes in the replicated level's b, and how much
    int32_t* cg_colmap = nullptr;                 // local coarsest node -> global (padded) node id
    double *cg_rows = nullptr, *cg_full = nullptr, *cg_send = nullptr, *cg_recv = nullptr;
    std::vector<std::vector<int32_t>> ghost_gid;  // per level: (owner rank, owner-local id) of each ghost node
    std::vector<std::vector<int32_t>> ghost_own;
    std::unique_ptr<HostPattern> pattern;      // kept until the (lazy) hierarchy build
    std::vector<double> host_pts;              // ... with the node coordinates (3 per node): the aggregation's strength filter on anisotropic meshes
    // optional per-launch timing of the fine-level SpMV family
    bool time_kernels = false;
    std::vector<std::array<hipEvent_t, 2>> ev_pool;
    std::vector<int> ev_mode;
    size_t ev_used = 0;
    double kt_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};            // modes 0..3 = SpmvMode, 4 = fused post-sweep on M = A P
    int64_t kt_calls[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};


// ---- across the translation units ------------------------------------------------------------------------------------------------
namespace sns {
// csrc/sns_setup.hip: symbolic hierarchy, assembly driver, numeric setup of the preconditioner
int alloc_level_vectors(Level& L);
int upload_pattern(Level& L, const HostPattern& P, int32_t** slot_row, hipStream_t s);
int global_sum(sns_ctx* h, double* v, int count);
int host_allgather(sns_ctx* h, const std::vector<double>& mine, std::vector<double>& all);
int check_plan_symmetry(sns_ctx* h, const Plan& p, int level);
int connect_plan(sns_ctx* h, Plan& p);
int build_hierarchy(sns_ctx* h, const HostPattern& fine);
int assemble(sns_ctx* h, int form, const double* w, double* F, bool want_matrix);
int timed_assemble(sns_ctx* h, int form, const double* w, double* F, bool want_matrix);
int pc_setup(sns_ctx* h);
int get_vec(sns_ctx* h, size_t k, double** out);
// csrc/sns_cycle.hip: the V-cycle, the preconditioner and operator applications
int vcycle(sns_ctx* h, int l, const double* b, double* x);
int coarse_cycle(sns_ctx* h, int l, const double* b, double* x);
int pc_apply(sns_ctx* h, const double* r, double* z);
int op_apply(sns_ctx* h, double* x, double* y);
int op_apply_dot(sns_ctx* h, double* x, double* y, const double* dotw);
int op_residual(sns_ctx* h, double* x, const double* b, double* r);
// csrc/sns_krylov.hip: BiCGStab / TFQMR / FGMRES and the solve driver with the damping retry
int krylov(sns_ctx* h, const double* b, double* x, int* its, int* reason, double* rnorm);
int norm2(sns_ctx* h, const double* x, double* out);
int dot(sns_ctx* h, const double* x, const double* y, double* out);
}  // namespace sns

// ---- small helpers, launch helpers and the cycle's predicates (internal linkage: every translation unit gets its own) ----------
namespace {

// C-ABI layer (include/sns.h): context, assembly driver, operator hierarchy,
// Krylov (BiCGStab / FGMRES) and Newton drivers.  Host code only launches
// kernels from sns_kernels.hip and moves scalars; there is no CPU compute path.
#include <hip/hip_runtime.h>

inline int vec_grid(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 2048); }

inline int64_t ld_of(const sns_ctx* h) { return 4 * (int64_t)h->n; }

inline int64_t nred_of(const sns_ctx* h) { return 4 * (int64_t)h->n_owned; }


inline int sync_stream(sns_ctx* h) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SNS_OK;
}


inline void time_begin(sns_ctx* h, int mode, hipStream_t st = nullptr) {
    if (!h->time_kernels) return;
    if (h->ev_used == h->ev_pool.size()) {
        std::array<hipEvent_t, 2> p;
        (void)hipEventCreate(&p[0]);
        (void)hipEventCreate(&p[1]);
        h->ev_pool.push_back(p);
        h->ev_mode.push_back(0);
    }
    h->ev_mode[h->ev_used] = mode;
    (void)hipEventRecord(h->ev_pool[h->ev_used][0], st ? st : h->stream);
}

inline void time_end(sns_ctx* h, hipStream_t st = nullptr) {
    if (!h->time_kernels) return;
    (void)hipEventRecord(h->ev_pool[h->ev_used][1], st ? st : h->stream);
    ++h->ev_used;
}

// resolve recorded event pairs (stream must be idle)
inline void time_collect(sns_ctx* h) {
    for (size_t i = 0; i < h->ev_used; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, h->ev_pool[i][0], h->ev_pool[i][1]) == hipSuccess) {
            h->kt_ms[h->ev_mode[i]] += ms;
            h->kt_calls[h->ev_mode[i]]++;
        }
    }
    h->ev_used = 0;
}


// finish a two-stage reduction locally: partial[nblocks][nred] -> dst_dev[0..nred)
inline void reduce_local(sns_ctx* h, int nblocks, int nred, double* dst_dev) {
    if (nblocks > 8192 && nred <= 8) {
        // the fused SpMV+dot leaves one partial per 32 rows (54 k at 10 M tets): a single workgroup needs ~40 us
        // for that, 2048-wide chunks on many CUs first ~5 us
        const int nchunks = (nblocks + 2047) / 2048;
        if (nchunks <= 4096) {
            hipLaunchKernelGGL(k_reduce_chunks, dim3(nchunks, nred), dim3(256), 0, h->stream, nblocks, nred, h->partial,
                               h->partial2);
            hipLaunchKernelGGL(k_reduce_final, dim3(nred), dim3(256), 0, h->stream, nchunks, nred, h->partial2, dst_dev);
            return;
        }
    }
    hipLaunchKernelGGL(k_reduce_final, dim3(nred), dim3(256), 0, h->stream, nblocks, nred, h->partial, dst_dev);
}

// sum `count` device doubles over the ranks (no-op without a communicator)
inline int allreduce(sns_ctx* h, double* buf_dev, int count) {
    if (h->comm && h->comm->active()) ++h->ctr_allreduce;
    return comm_allreduce_sum(h->comm.get(), buf_dev, count, h->stream);
}

inline int reduce_to(sns_ctx* h, int nblocks, int nred, double* dst_dev);

// BiCGStab's two reductions with the scalar update they feed (WHICH 1: alpha, 2: omega & co, k_reduce_final_bicg): without a
// communicator the last reduction stage and the update are one launch; with one, the all-reduce sits between them
template <int WHICH>
int reduce_bicg(sns_ctx* h, int nblocks, double* red, double* sc) {
    constexpr int NRED = WHICH == 1 ? 1 : 5;
    Peer* pe = (h->comm && h->comm->active()) ? h->comm->peer : nullptr;
    if (h->comm && h->comm->active() && !pe) {
        SNS_TRY(reduce_to(h, nblocks, NRED, red));
        if (WHICH == 1) hipLaunchKernelGGL(k_bicg_alpha, dim3(1), dim3(64), 0, h->stream, sc, red);
        else hipLaunchKernelGGL(k_bicg_omega, dim3(1), dim3(64), 0, h->stream, sc, red);
        return SNS_OK;
    }
    const double* src = h->partial;
    int nb = nblocks;
    if (nblocks > 8192) {                        // (as reduce_local: 2048-wide chunks on many CUs first; one workgroup over 27 k
                                                 // partials -- the slab share -- was measured at 29 us against 4.6 + 4.8 for the two stages)
        const int nchunks = (nblocks + 2047) / 2048;
        if (nchunks <= 4096) {
            hipLaunchKernelGGL(k_reduce_chunks, dim3(nchunks, NRED), dim3(256), 0, h->stream, nblocks, NRED, h->partial, h->partial2);
            src = h->partial2;
            nb = nchunks;
        }
    }
    if (pe) {                                    // peer windows: the all-reduce rides inside the same single-workgroup launch
        SNS_TRY(peer_check(h->comm.get()));
        ++h->ctr_allreduce;
        if (pe->host_sync) {                     // team: reduce + contribute | host barrier | sum + scalar update
            hipLaunchKernelGGL((k_reduce_final_bicg_peer<WHICH>), dim3(1), dim3(256), 0, h->stream, nb, src, red, sc,
                               peer_allreduce_args(pe, 1));
            SNS_TRY(comm_host_barrier(h->comm.get(), h->stream));
            hipLaunchKernelGGL((k_reduce_final_bicg_peer<WHICH>), dim3(1), dim3(256), 0, h->stream, nb, src, red, sc,
                               peer_allreduce_args(pe, 2));
            return SNS_OK;
        }
        hipLaunchKernelGGL((k_reduce_final_bicg_peer<WHICH>), dim3(1), dim3(256), 0, h->stream, nb, src, red, sc,
                           peer_allreduce_args(pe, 0));
        return SNS_OK;
    }
    hipLaunchKernelGGL((k_reduce_final_bicg<WHICH>), dim3(1), dim3(256), 0, h->stream, nb, src, red, sc);
    return SNS_OK;
}

inline int reduce_to(sns_ctx* h, int nblocks, int nred, double* dst_dev) {
    reduce_local(h, nblocks, nred, dst_dev);
    return allreduce(h, dst_dev, nred);
}

// ... and bring `count` doubles starting at src_dev to the host (synchronises the stream)
inline int fetch(sns_ctx* h, const double* src_dev, int count, double* out) {
    HIP_TRY(hipMemcpyAsync(h->h_scal, src_dev, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    ++h->ctr_host_syncs;
    SNS_TRY(peer_check(h->comm.get()));          // (peer transport: a collective behind this result may have given up waiting)
    std::memcpy(out, h->h_scal, count * sizeof(double));
    return SNS_OK;
}


// fill the ghost tail of a level-l vector from the owning ranks
inline int exchange_level(sns_ctx* h, int l, double* x) {
    Comm* c = h->comm.get();
    if (!c || !c->active() || c->nranks <= 1 || (size_t)l >= c->plans.size()) return SNS_OK;
    ++h->ctr_exchange;
    return comm_exchange(c, c->plans[l], x, h->stream);
}

inline int halo_exchange(sns_ctx* h, double* x) { return exchange_level(h, 0, x); }

// Per-launch timing of the level-0 SpMV family (bench.py roofline leg): event pairs are
// recorded around every fine-level launch while h->time_kernels is set and resolved after
// the solve has synchronised.


// Multi-GPU, level 0: a pass is either over every row (split 0), over the interior rows only (1: rows with a ghost
// column, flagged in h->bnd_flag, are skipped) or over the boundary rows listed in h->bnd_rows (2).
// Window transports (round 5): 3 = every row in one launch, the ghost entries read straight from the receive window `gs`.
struct Split {
    int mode = 0;
    hipStream_t stream = nullptr;       // nullptr = the handle's stream
    int partial_off = 0;
    GhostSrc gs;
};


// y = A_l x (or fused variants).  rows = number of block rows computed.
template <int MODE>
void launch_spmv(sns_ctx* h, const Level& L, int32_t rows, const double* x, double* y, const double* b,
                 double omega, const double* dotw, Split sp = Split()) {
    hipStream_t st = sp.stream ? sp.stream : h->stream;
    const bool fine = (&L == &h->levels[0]);
    if (sp.mode == 2) rows = h->n_bnd;
    const int grid = (rows + 31) / 32;
    if (grid == 0) return;
    if (fine && sp.mode == 1) {
        time_begin(h, MODE, st);                      // multi-GPU: the interior pass is the bulk of a split launch
        hipLaunchKernelGGL((k_spmv<MODE, 1, 1, 1>), dim3(grid), dim3(256), 0, st, rows, L.rowptr, L.colind, L.vals, x, y,
                           b, L.dinv, omega, dotw, h->partial, (const int32_t*)nullptr, h->bnd_flag, sp.partial_off, GhostSrc());
        time_end(h, st);
    } else if (fine && sp.mode == 2) {
        hipLaunchKernelGGL((k_spmv<MODE, 1, 1, 2>), dim3(grid), dim3(256), 0, st, rows, L.rowptr, L.colind, L.vals, x, y,
                           b, L.dinv, omega, dotw, h->partial, h->bnd_rows, (const uint8_t*)nullptr, sp.partial_off, GhostSrc());
    } else if (fine && sp.mode == 3) {
        time_begin(h, MODE, st);
        hipLaunchKernelGGL((k_spmv<MODE, 1, 1, 3>), dim3(grid), dim3(256), 0, st, rows, L.rowptr, L.colind, L.vals, x, y,
                           b, L.dinv, omega, dotw, h->partial, (const int32_t*)nullptr, (const uint8_t*)nullptr, 0, sp.gs);
        time_end(h, st);
    } else if (fine) {
        time_begin(h, MODE);
#ifdef SNS_HARNESS                                     // in-solver A/B of the stepped loop (harness build only)
        if constexpr (MODE == SPMV_AX || MODE == SPMV_AX_DOT) {
            if (std::getenv("SNS_FP64_STEPPED")) {
                hipLaunchKernelGGL((k_spmv<MODE, 1, 3, 0>), dim3(grid), dim3(256), 0, st, rows, L.rowptr, L.colind, L.vals,
                                   x, y, b, L.dinv, omega, dotw, h->partial, (const int32_t*)nullptr, (const uint8_t*)nullptr, 0, GhostSrc());
                time_end(h);
                return;
            }
        }
#endif
        hipLaunchKernelGGL((k_spmv<MODE, 1, 1, 0>), dim3(grid), dim3(256), 0, st, rows, L.rowptr, L.colind, L.vals,
                           x, y, b, L.dinv, omega, dotw, h->partial, (const int32_t*)nullptr, (const uint8_t*)nullptr, 0, GhostSrc());
        time_end(h);
    } else if constexpr (MODE != SPMV_AX_DOT) {
        hipLaunchKernelGGL((k_spmv<MODE, 0, 0, 0>), dim3(grid), dim3(256), 0, st, rows, L.rowptr, L.colind, L.vals,
                           x, y, b, L.dinv, omega, dotw, h->partial, (const int32_t*)nullptr, (const uint8_t*)nullptr, 0, GhostSrc());
    }
}


// Preconditioner passes (Jacobi sweep, residual) of the AMG cycle on the low-precision copy of the level matrix
// (amg_f32_matrix: 1 = fp32, 2 = fp16 with row scales).
template <int MODE, int FINE, int SPLIT, int FMT>
void launch_lp(sns_ctx* h, const Level& L, int32_t rows, hipStream_t st, const double* x, double* y, const double* b,
               double omega, const GhostSrc& gs = GhostSrc()) {
    const int grid = (rows + 63) / 64;
    if (grid == 0) return;
    const void* vals = FMT == 2 ? (const void*)L.vals16 : (const void*)L.vals32;
#ifdef SNS_HARNESS                                         // in-solver A/B of the stepped loop (harness build only)
    if constexpr (FMT == 2 && FINE == 1 && SPLIT == 0) {
        if (std::getenv("SNS_LP_STEPPED")) {
            hipLaunchKernelGGL((k_spmv_lp<MODE, 1, 0, 2, 0>), dim3(grid), dim3(256), 0, st, rows, L.rowptr, L.colind, vals,
                               L.scale16, x, y, b, L.dinv32, omega, (const int32_t*)nullptr, (const uint8_t*)nullptr, GhostSrc());
            return;
        }
    }
#endif
    hipLaunchKernelGGL((k_spmv_lp<MODE, FINE, SPLIT, FMT, 1>), dim3(grid), dim3(256), 0, st, rows, L.rowptr, L.colind, vals,
                       L.scale16, x, y, b, L.dinv32, omega, SPLIT == 2 ? h->bnd_rows : (const int32_t*)nullptr,
                       SPLIT == 1 ? h->bnd_flag : (const uint8_t*)nullptr, gs);
}

template <int MODE, int FMT>
void launch_lp_fmt(sns_ctx* h, const Level& L, int32_t rows, const double* x, double* y, const double* b, double omega,
                   Split sp) {
    hipStream_t st = sp.stream ? sp.stream : h->stream;
    const bool fine = (&L == &h->levels[0]);
    if (fine && sp.mode == 1) {
        time_begin(h, MODE, st);
        launch_lp<MODE, 1, 1, FMT>(h, L, rows, st, x, y, b, omega);
        time_end(h, st);
    } else if (fine && sp.mode == 2) {
        launch_lp<MODE, 1, 2, FMT>(h, L, h->n_bnd, st, x, y, b, omega);
    } else if (fine && sp.mode == 3) {
        time_begin(h, MODE, st);
        launch_lp<MODE, 1, 3, FMT>(h, L, rows, st, x, y, b, omega, sp.gs);
        time_end(h, st);
    } else if (fine) {
        time_begin(h, MODE);
        launch_lp<MODE, 1, 0, FMT>(h, L, rows, st, x, y, b, omega);
        time_end(h);
    } else {
        launch_lp<MODE, 0, 0, FMT>(h, L, rows, st, x, y, b, omega);
    }
}

inline int lp_format(const sns_ctx* h, const Level& L) {
    if (h->opt.amg_f32_matrix == 2 && L.vals16) return 2;
    if (h->opt.amg_f32_matrix && L.vals32) return 1;
    return 0;
}

template <int MODE>
void launch_pc_spmv(sns_ctx* h, const Level& L, int32_t rows, const double* x, double* y, const double* b,
                    double omega, Split sp = Split()) {
    const int fmt = lp_format(h, L);
    if (fmt == 2) launch_lp_fmt<MODE, 2>(h, L, rows, x, y, b, omega, sp);
    else if (fmt == 1) launch_lp_fmt<MODE, 1>(h, L, rows, x, y, b, omega, sp);
    else launch_spmv<MODE>(h, L, rows, x, y, b, omega, nullptr, sp);
}


// Do the level-0 passes of this handle read their ghost entries straight from the receive window (halo_windows)?
inline bool fine_windows(const sns_ctx* h) {
    const Comm* c = h->comm.get();
    return c && c->windows() && c->nranks > 1 && h->opt.halo_windows && !h->team_overlap && !c->plans.empty() &&
           c->plans[0].identity_recv && c->plans[0].win_recv[0] != nullptr;
}

// Level-0 pass whose input needs a halo exchange first (multi-GPU): the exchange of xe's ghost tail runs on the
// handle's stream (every RCCL call stays on ONE stream, in program order) while the interior rows -- the rows
// without a ghost column, i.e. nearly all of them -- are computed on a second stream; the few boundary rows follow
// once the halo has been unpacked.  `pc` selects the preconditioner flavour of the kernel (fp32 matrix copy).
// Without a transport, with a single rank or with SNS_NO_OVERLAP set this is exchange + one full pass.
template <int MODE>
int exchange_and_spmv(sns_ctx* h, double* xe, const double* x, double* y, const double* b, double omega,
                      const double* dotw, bool pc) {
    Level& L = h->levels[0];
    const int32_t rows = h->n_owned;
    Comm* c = h->comm.get();
    const bool dist = c && c->active() && c->nranks > 1;
    h->bnd_dot_blocks = 0;
    auto pass = [&](Split sp) {
        if constexpr (MODE == SPMV_B_MINUS_AX || MODE == SPMV_JACOBI) {
            if (pc) { launch_pc_spmv<MODE>(h, L, rows, x, y, b, omega, sp); return; }
        }
        launch_spmv<MODE>(h, L, rows, x, y, b, omega, dotw, sp);
    };
    if (dist && fine_windows(h) && xe == x) {
        // window transports: ONE put launch; the pass reads the ghost entries from the receive window and its boundary waves
        // wait for the neighbours' flags themselves -- no unpack, no boundary launch, no second stream
        ++h->ctr_exchange;
        if (h->put_pending == xe) SNS_TRY(comm_put_carried(c, c->plans[0], h->stream));    // (the cycle's last kernel put it)
        else SNS_TRY(comm_put(c, c->plans[0], xe, h->stream));
        h->put_pending = nullptr;
        Split s3;
        s3.mode = 3;
        s3.gs = comm_ghost_src(c, c->plans[0]);
        pass(s3);
        h->dot_partials = (rows + 31) / 32;              // (one per workgroup in this form of the pass)
        return SNS_OK;
    }
    h->dot_partials = 4 * ((rows + 31) / 32);            // one per wave ...

    if (!dist || !h->bnd_flag || h->no_overlap || !h->opt.halo_overlap) {
        SNS_TRY(halo_exchange(h, xe));
        pass(Split());
        return SNS_OK;
    }
    const int gs = (rows + 31) / 32;                 // partial sums exist in the fp64 AX_DOT pass only
    Split s1, s2;
    s1.mode = 1;
    s2.mode = 2;
    s2.partial_off = gs;
    if (MODE == SPMV_AX_DOT) { h->bnd_dot_blocks = (h->n_bnd + 31) / 32; h->dot_partials += 4 * h->bnd_dot_blocks; }   // ... of both launches
    if (c->nccl || (c->peer && !c->team) || h->team_overlap) {
        // (team transport with SNS_TEAM_OVERLAP=1: the same two-stream choreography -- interior pass on the side
        // stream, event joins, per-launch timing events on that stream -- over the emulated exchange, so that the
        // stream dependencies of the production path are exercised on a 1-GPU box)
        if (!h->side_stream) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            HIP_TRY(hipStreamCreateWithPriority(&h->side_stream, hipStreamNonBlocking, lo));
            HIP_TRY(hipEventCreateWithFlags(&h->ev_x, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(h->ev_x, h->stream));                 // x (owned part) is ready
        HIP_TRY(hipStreamWaitEvent(h->side_stream, h->ev_x, 0));
        s1.stream = h->side_stream;
        pass(s1);                                                    // interior rows, concurrent with the halo
        HIP_TRY(hipEventRecord(h->ev_side, h->side_stream));
        SNS_TRY(halo_exchange(h, xe));                               // pack, ncclSend/Recv group, unpack
        pass(s2);                                                    // boundary rows
        HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_side, 0));       // y complete for whatever comes next
    } else {
        // team transport (tests, default): the exchange synchronises the host anyway; same two passes, one stream
        SNS_TRY(halo_exchange(h, xe));
        pass(s1);
        pass(s2);
    }
    return SNS_OK;
}


// Aggregate-block Jacobi smoother (amg_block_smooth, csrc/sns_block.hip), symbolic part: the member rows of every aggregate of
// level L padded to 8 slots.  Levels whose aggregates can have more than 8 members (amg_agg_size > 8) keep the nodal blocks.
// Aggregate blocks on the FINE level: always with amg_block_smooth = 2; with 1 on a partitioned handle whose share of the fine level is
// at most amg_block_fine_rows rows per rank -- the latency-bound strong split, where 20 % fewer iterations (and collectives) outweigh
// the inverse blocks' bytes.  Global counts only: every rank answers alike.
inline bool fine_blocks_wanted(const sns_ctx* h) {
    const Comm* c = h->comm.get();
    return policy::fine_blocks(h->opt, (c && c->active()) ? c->nranks : 1, h->n_global_fine);
}

// Is level l smoothed with the aggregate blocks?  (options only, no device state: every rank of a partitioned run must answer alike)
inline bool block_active(const sns_ctx* h, int l) {
    if (l < 0 || l + 1 >= (int)h->levels.size()) return false;                 // the coarsest level is solved or point-smoothed
    const Level& L = h->levels[l];
    if (!L.blk_rows) return false;
    if (l == 0 && !fine_blocks_wanted(h)) return false;
    if (h->rep_level > 0 && l == h->rep_level - 1) return false;               // only the source of the replicated copy
    // (rows per rank, the same figure on every rank: policy::blocks_allowed)
    const bool replicated = h->rep_level > 0 && l >= h->rep_level;
    const int nr = (h->comm && h->comm->active() && !replicated) ? std::max(1, h->comm->nranks) : 1;
    return policy::blocks_allowed(h->opt, L.n_global, nr);
}


// one smoothing sweep y = x + w S (b - A x) of level l: S = the aggregates' inverse blocks where block_active, else the nodal D^-1
inline void launch_sweep(sns_ctx* h, int l, const Level& L, int32_t rows, const double* x, double* y, const double* b, double omega) {
    if (block_active(h, l) && L.binv32) {
        const int32_t ns = 8 * L.n_blk;
        const unsigned grid = (unsigned)((ns + 63) / 64);
        if (grid == 0) return;
        if (L.binv_fmt == 2)
            hipLaunchKernelGGL((k_bsweep<2, 0>), dim3(grid), dim3(256), 0, h->stream, ns, L.blk_rows, L.rowptr, L.colind,
                               (const void*)L.vals16, L.scale16, (const void*)L.binv32, x, y, b, omega, GhostSrc(), PutDst());
        else
            hipLaunchKernelGGL((k_bsweep<1, 0>), dim3(grid), dim3(256), 0, h->stream, ns, L.blk_rows, L.rowptr, L.colind,
                               (const void*)L.vals32, (const float*)nullptr, (const void*)L.binv32, x, y, b, omega, GhostSrc(), PutDst());
        return;
    }
    launch_pc_spmv<SPMV_JACOBI>(h, L, rows, x, y, b, omega);
}

// first sweep of a cycle from the zero guess, z = w S b (omega = 1: S b alone, the spectral estimate's operator)
inline void launch_first_sweep(sns_ctx* h, int l, const Level& L, int32_t rows, const double* b, double omega, double* z,
                               PutDst* pd = nullptr) {
    if (rows <= 0) return;
    const int g4 = (int)((4 * (int64_t)rows + 255) / 256);
    if (block_active(h, l) && L.binv32) {
        const int32_t ns = 8 * L.n_blk;
        if (L.binv_fmt == 2)
            hipLaunchKernelGGL((k_bfirst<2>), dim3((unsigned)((ns + 63) / 64)), dim3(256), 0, h->stream, ns, L.blk_rows,
                               (const void*)L.binv32, b, omega, z, pd ? *pd : PutDst());
        else
            hipLaunchKernelGGL((k_bfirst<1>), dim3((unsigned)((ns + 63) / 64)), dim3(256), 0, h->stream, ns, L.blk_rows,
                               (const void*)L.binv32, b, omega, z, pd ? *pd : PutDst());
        return;
    }
    if (pd) *pd = PutDst();                              // (the nodal first sweeps do not carry a put)
    if (false) {
    } else if (lp_format(h, L) != 0 && L.dinv32) {
        hipLaunchKernelGGL(k_bjacobi32, dim3(g4), dim3(256), 0, h->stream, rows, L.dinv32, b, omega, z);
    } else {
        hipLaunchKernelGGL(k_bjacobi, dim3(g4), dim3(256), 0, h->stream, rows, L.dinv, b, omega, z);
    }
}


// rows at or below which a level >= 1 ends the hierarchy (it is solved directly)
inline int coarsest_rows(const sns_options& o) { return policy::coarsest_rows(o); }


// |lambda|max of Dinv*A on level l by a few power iterations (device resident; one host sync).
// The damped block-Jacobi smoother x += w Dinv (b - A x) needs w*|lambda|max < 2; on the reference's
// operator the fixed w = 0.9 already diverges at 10 M tets, so w is capped per level at the smoothing-optimal 4/(3 |lambda|max).  (Measured cliff on the coarse
// levels of the 10 M-tet Jacobian: w = 0.80 converges in 45 iterations, w >= 0.82 overflows, although the
// dominant mode itself is still damped there -- the offending mode is not the one of largest modulus.)
// (rank-local row count: with the option on, a level's ranks must all fall on the same side of the threshold -- the slab / RCB
// partitions are balanced to a few rows; off (0, the default) no rank ever takes this path, empty ranks included)
inline bool level_sx(const sns_ctx* h, const Level& L) {
    return L.xg && h->opt.amg_sweep_exchange_rows > 0 && L.n_owned <= h->opt.amg_sweep_exchange_rows;
}

// sweeps per level: the fine level is the expensive one (1 sweep); level 1 and 2 are cheap and are where
// plain aggregation needs the smoothing (4 and 6); levels >= 3 are launch-bound (2).  Measured on the
// 10 M-tet Jacobian: (1,4,6,2) 40-42 its / 180-186 ms; (1,4,4,4) 45 / 204; (2,2,2,2) 54 / 323.
// Large problems (amg_nu_scale_with_size): the plain-aggregation V-cycle loses convergence with its depth, and on a big mesh the
// levels >= 2 cost next to nothing -- measured on one GPU (profiles/r3_deep_sweeps.txt): 81 M tets 73 / 82 -> 53 / 57 iterations and
// 1743 -> 1303 ms per Newton step with 10 + 10 sweeps on level 2 and 8 + 8 below instead of 6 + 6 and 2 + 2; 24 M tets 52 / 55 -> 45 / 49
// with 8 + 8 and 4 + 4; at 10 M tets the extra latency-bound passes cost what they save, so the schedule follows the GLOBAL fine size.
inline int level_nu(const sns_ctx* h, int l) {
    const int ll = (h->rep_level > 0 && l >= h->rep_level) ? l - 1 : l;      // the replicated copy is not a new level
    // (global counts: every rank must arrive at the same schedule -- levels with exchanged sweeps are collective)
    const int nlev = policy::depth_equivalent(h->opt, (int)h->levels.size() - (h->rep_level > 0 ? 1 : 0),
                                              h->levels.back().dense_gj != nullptr, h->levels.back().n);
    const bool small_aggregates = h->n_global_l1 > 0 && (double)h->n_global_fine < 6.0 * (double)h->n_global_l1;
    const policy::ExtraSweeps e = policy::extra_sweeps(h->opt, h->n_global_fine, nlev, small_aggregates);
    return policy::level_nu(h->opt, ll, block_active(h, l), e);
}

// one exchange after the coarse-grid correction makes a SINGLE post-smoothing sweep the exact global block-Jacobi
// sweep; with several sweeps the ghost values would be frozen while the owned ones move, which measurably hurts
// the Stokes operator (8 slabs of the 10 M-tet duct: 47 -> 65 iterations) -- so only where nu = 1 (the fine level)
inline bool level_px(const sns_ctx* h, int l, const Level& L) {
    return L.xg && !level_sx(h, L) && h->opt.amg_post_exchange && level_nu(h, l) == 1 && (l == 0 || !block_active(h, l));
}

// Partitioned level l >= 1 cycled with EXACT global sweeps over a window transport (amg_exact_sweeps, round 5): every sweep is
// preceded by one put launch and reads its ghost entries from the receive window, the coarse-grid correction sits inside the
// first post-sweep (M = A P), residual + restriction stay one launch -- the single-GPU cycle, distributed.  Options and the
// hierarchy's global structure only: every rank answers alike.
inline bool level_exact(const sns_ctx* h, int l) {
    const Comm* c = h->comm.get();
    if (!c || !c->windows() || c->nranks <= 1 || !h->opt.halo_windows || !h->opt.amg_exact_sweeps || h->team_overlap) return false;
    if (l < 1 || l + 1 >= (int)h->levels.size() || (size_t)l >= c->plans.size()) return false;
    const Level& L = h->levels[l];
    if (!L.xg || (h->rep_level > 0 && l >= h->rep_level - 1)) return false;     // partitioned AND cycled (not the replicated tail's source)
    if (level_sx(h, L) || !block_active(h, l) || !L.ap_rowptr) return false;
    if (!h->opt.amg_fused_post || h->opt.amg_fuse_restrict == 0 || h->opt.pc_type != SNS_PC_AMG) return false;
    if (!c->plans[l].identity_recv || !c->plans[l].win_recv[0]) return false;
    const bool rep_src = h->rep_level > 0 && l + 1 == h->rep_level - 1;
    if (rep_src) return L.ap_colind_rep != nullptr;                             // xc straight from the replicated solution
    return (size_t)(l + 1) < c->plans.size() && c->plans[l + 1].identity_recv && c->plans[l + 1].win_recv[0] != nullptr;
}

inline bool uses_ghosts_in_sweeps(const sns_ctx* h, int l, const Level& L) {
    return level_sx(h, L) || level_px(h, l, L) || level_exact(h, l);
}

// Stability limit of the smoother damping on level l from the dominant Ritz values of S A (S = the level's smoother blocks,
// nodal or aggregate): M = 8 Arnoldi steps from the deterministic start vector of the power iteration (classical Gram-Schmidt
// with one re-orthogonalisation, the FGMRES kernels; one host read per step), eigenvalues of the 8 x 8 Hessenberg matrix on the
// host (sns_host_hessenberg_eigs).  |1 - w theta| < 1 needs w < 2 Re(theta) / |theta|^2: *limit = the minimum over the Ritz
// values with |theta| >= 0.5 |theta|max (those a few Arnoldi steps have converged to).  The power iteration above sees the
// modulus only; on a convection-dominated coarse level the dominant eigenvalues are complex, and a level that runs 1 + 6
// sweeps amplifies a damping above the limit seven times per cycle (oracle/experiments/r4_damping.py).
inline void level_sweeps(const sns_ctx* h, int l, int& nu_pre, int& nu_post);

// sweeps before / after the coarse-grid correction on level l (the first pre-sweep is omega D^-1 b)
inline void level_sweeps(const sns_ctx* h, int l, int& nu_pre, int& nu_post) {
    const int ll = (h->rep_level > 0 && l >= h->rep_level) ? l - 1 : l;
    // rank-local sweeps: a partitioned handle (any level: the rule of rounds 2-4) unless the level's sweeps are the exact global
    // ones (level_exact: then it IS the single-GPU cycle)
    const bool part = h->comm && h->comm->active() && h->comm->nranks > 1;
    const bool exact = part && level_exact(h, l);
    const policy::Sweeps s = policy::level_sweeps(h->opt, ll, block_active(h, l), part && !exact, level_nu(h, l), exact);
    nu_pre = s.pre;
    nu_post = s.post;
}

// Does the restriction from level l also do level l + 1's first sweep (k_restrict with dinv32_c)?  Only where that sweep is
// the plain rank-local w Dc^-1 bc of a smoothed level on its fp32 D^-1 copy: not the dense coarsest level, not the level whose
// cycle is the all-gather into the replicated tail (nor that tail's first level, whose right-hand side comes from the gather),
// not a partitioned level whose sweeps exchange ghost values, not the experimental fine-cycle shapes.
inline bool restrict_fuses_first(const sns_ctx* h, int l) {
    const int nl = (int)h->levels.size();
    const int c = l + 1;
    if (l < 0 || c + 1 >= nl) return false;
#ifdef SNS_HARNESS
    if (std::getenv("SNS_NO_RESTRICT_FUSE")) return false;
#endif
    if (h->rep_level > 0 && (c == h->rep_level - 1 || l == h->rep_level - 1)) return false;
    const Level& C = h->levels[c];
    // (a partitioned coarse level qualifies too: its first sweep starts from zero and is rank-local by construction -- owned right-hand
    // side, owned rows of the start buffer, the ghost tail stays as it is --, unless its sweeps exchange ghost values, whose damping
    // and buffers follow the exchanging code path)
    if ((C.xg || C.n != C.n_owned) && level_sx(h, C)) return false;
    if (block_active(h, c)) return C.binv32 != nullptr;    // k_restrict_blk: restriction in the order of the coarse aggregates
    return lp_format(h, C) != 0 && C.dinv32 != nullptr;
}


// The buffer a smoothed level's cycle starts from (its first sweep z = w D^-1 b is written there; after
// nu_pre - 1 + nu_post ping-pong swaps the result must sit in x): the ONE place that knows the parity rule -- vcycle() and the
// restriction of the level above (which writes that first sweep when restrict_fuses_first says so) both ask here.
inline double* cycle_start_buffer(sns_ctx* h, int l, double* x) {
    int nu_pre = 1, nu_post = 1;
    level_sweeps(h, l, nu_pre, nu_post);
    return ((nu_pre - 1 + nu_post) & 1) ? h->pong[l] : x;
}


// Does level l take the fused coarse-grid correction + first post-smoothing sweep (k_post_lp / k_bpost over M = A P)?  Serial levels
// always (given M and a low-precision format); a partitioned fine level when its single post-sweep is the exact global one (px).
// One place for the rule: vcycle() and the callers that choose the cycle's buffers ask here.
inline bool level_fused_post(const sns_ctx* h, int l) {
    if (l < 0 || l + 1 >= (int)h->levels.size()) return false;
    const Level& L = h->levels[l];
    int nu_pre = 1, nu_post = 1;
    level_sweeps(h, l, nu_pre, nu_post);
    const int fmt_l = lp_format(h, L);
    const bool have_m = h->opt.amg_fused_post && L.ap_rowptr && fmt_l != 0 && L.dinv32 &&
                        (fmt_l == 2 ? L.ap_vals16 != nullptr : L.ap_vals32 != nullptr) && nu_post >= 1 && !level_sx(h, L);
    return have_m && (!L.xg || (l == 0 && level_px(h, l, L) && level_nu(h, l) == 1));
}

// A partitioned fine level in that mode never READS the ghost tails of its cycle buffers with the "ghosts are zero" assumption (no
// rank-local sweep runs there: the first sweep starts from zero, the post-sweep goes over M): the halo of the residual can land in
// the iterate's own tail, the tails need no clearing, and the cycle can run in the caller's vector.
inline bool fine_tails_unused(const sns_ctx* h) {
    return h->levels.size() >= 2 && h->levels[0].xg && !(h->rep_level == 1) && level_fused_post(h, 0);
}


// First level (>= 1) small enough that its kernels are launch-bound rather than bandwidth-bound: it and everything
// below run as one graph.  10 M tets: level 2 (36 k rows; level 1 has 218 k rows = 46 us per sweep); 1 M tets: level 1.
inline int serial_graph_level(const sns_ctx* h) {
    int max_rows = policy::GRAPH_MAX_ROWS;
#ifdef SNS_HARNESS
    if (std::getenv("SNS_GRAPH_ROWS")) max_rows = std::atoi(std::getenv("SNS_GRAPH_ROWS"));
#endif
    for (int l = 1; l < (int)h->levels.size(); ++l)
        if (h->levels[l].n <= max_rows) return l;
    return 0;
}


// one aggregate-block sweep of a partitioned level with the ghost entries of x from the level's receive window
inline void launch_sweep_windows(sns_ctx* h, const Level& L, const double* x, double* y, const double* b, double omega, const GhostSrc& gs,
                                 const PutDst& pd = PutDst()) {
    const int32_t ns = 8 * L.n_blk;
    const unsigned grid = (unsigned)((ns + 63) / 64);
    if (grid == 0) return;
    if (L.binv_fmt == 2)
        hipLaunchKernelGGL((k_bsweep<2, 1>), dim3(grid), dim3(256), 0, h->stream, ns, L.blk_rows, L.rowptr, L.colind,
                           (const void*)L.vals16, L.scale16, (const void*)L.binv32, x, y, b, omega, gs, pd);
    else
        hipLaunchKernelGGL((k_bsweep<1, 1>), dim3(grid), dim3(256), 0, h->stream, ns, L.blk_rows, L.rowptr, L.colind,
                           (const void*)L.vals32, (const float*)nullptr, (const void*)L.binv32, x, y, b, omega, gs, pd);
}


// Do both halves of the all-gather of the replicated tail's right-hand side ride in solver kernels (AgPut in the residual +
// restriction of the level above the source, AgGet in the tail's first sweep: k_bfirst_gather)?  The level above the source must
// run the window cycle with the fused residual + restriction (a level >= 1), the tail's first level must take aggregate blocks,
// and a rank's piece must fit the staging area.
inline bool level_exact(const sns_ctx* h, int l);
inline bool rep_gather_first(const sns_ctx* h) {
    const Comm* c = h->comm.get();
    if (!c || !c->windows() || c->nranks <= 1 || !h->opt.halo_windows || h->rep_level < 3) return false;
    if (!level_exact(h, h->rep_level - 2)) return false;
    const Level& C = h->levels[h->rep_level];
    if (!block_active(h, h->rep_level) || !C.binv32 || C.n_blk <= 0 || !h->rep_rowmap) return false;
    return (size_t)4 * h->rep_maxn * (size_t)c->nranks <= c->peer->ag_doubles;
}

// Does level l run the window form of the cycle (vcycle_windows)?  The fine level: its passes read the receive window and its
// single post-sweep is the fused exact one; a level >= 1: level_exact.
inline bool level_windows(const sns_ctx* h, int l) {
    if (l == 0) {
        const Comm* c = h->comm.get();
        return fine_windows(h) && fine_tails_unused(h) && c->plans.size() > 1 && c->plans[1].identity_recv &&
               c->plans[1].win_recv[0] != nullptr;
    }
    return level_exact(h, l);
}

// the put of the fine level's first sweep when a Krylov kernel runs that sweep (k_bfirst_bicg): empty unless the fine level runs
// the window form of the cycle
inline PutDst first_sweep_put(const sns_ctx* h) {
    const Comm* c = h->comm.get();
    if (!h->fuse_puts || !c || !c->peer || !level_windows(h, 0)) return PutDst();
    return comm_put_dst(c, c->plans[0]);
}


}  // namespace
