// C-ABI layer (include/sns.h): context, assembly driver, operator hierarchy,
// Krylov (BiCGStab / FGMRES) and Newton drivers.  Host code only launches
// kernels from sns_kernels.hip and moves scalars; there is no CPU compute path.
#include <hip/hip_runtime.h>
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

namespace sns {

static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }

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
static int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIP_TRY(hipMalloc((void**)p, count * sizeof(T)));
    return SNS_OK;
}
template <class T>
static int dev_upload(T** p, const std::vector<T>& v, hipStream_t) {
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

namespace {

inline int vec_grid(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 2048); }
inline int64_t ld_of(const sns_ctx* h) { return 4 * (int64_t)h->n; }
inline int64_t nred_of(const sns_ctx* h) { return 4 * (int64_t)h->n_owned; }

int sync_stream(sns_ctx* h) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SNS_OK;
}

void time_begin(sns_ctx* h, int mode, hipStream_t st = nullptr) {
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
void time_end(sns_ctx* h, hipStream_t st = nullptr) {
    if (!h->time_kernels) return;
    (void)hipEventRecord(h->ev_pool[h->ev_used][1], st ? st : h->stream);
    ++h->ev_used;
}
// resolve recorded event pairs (stream must be idle)
void time_collect(sns_ctx* h) {
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
void reduce_local(sns_ctx* h, int nblocks, int nred, double* dst_dev) {
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
int allreduce(sns_ctx* h, double* buf_dev, int count) {
    if (h->comm && h->comm->active()) ++h->ctr_allreduce;
    return comm_allreduce_sum(h->comm.get(), buf_dev, count, h->stream);
}
int reduce_to(sns_ctx* h, int nblocks, int nred, double* dst_dev);
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
int reduce_to(sns_ctx* h, int nblocks, int nred, double* dst_dev) {
    reduce_local(h, nblocks, nred, dst_dev);
    return allreduce(h, dst_dev, nred);
}
// ... and bring `count` doubles starting at src_dev to the host (synchronises the stream)
int fetch(sns_ctx* h, const double* src_dev, int count, double* out) {
    HIP_TRY(hipMemcpyAsync(h->h_scal, src_dev, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    ++h->ctr_host_syncs;
    SNS_TRY(peer_check(h->comm.get()));          // (peer transport: a collective behind this result may have given up waiting)
    std::memcpy(out, h->h_scal, count * sizeof(double));
    return SNS_OK;
}

// fill the ghost tail of a level-l vector from the owning ranks
int exchange_level(sns_ctx* h, int l, double* x) {
    Comm* c = h->comm.get();
    if (!c || !c->active() || c->nranks <= 1 || (size_t)l >= c->plans.size()) return SNS_OK;
    ++h->ctr_exchange;
    return comm_exchange(c, c->plans[l], x, h->stream);
}
int halo_exchange(sns_ctx* h, double* x) { return exchange_level(h, 0, x); }

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
        SNS_TRY(comm_put(c, c->plans[0], xe, h->stream));
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

int alloc_level_vectors(Level& L) {
    const size_t nd = 4 * (size_t)L.n;
    SNS_TRY(dev_alloc(&L.x, nd));
    SNS_TRY(dev_alloc(&L.b, nd));
    SNS_TRY(dev_alloc(&L.r, nd));
    HIP_TRY(hipMemset(L.x, 0, nd * sizeof(double)));
    HIP_TRY(hipMemset(L.b, 0, nd * sizeof(double)));
    HIP_TRY(hipMemset(L.r, 0, nd * sizeof(double)));
    return SNS_OK;
}

int upload_pattern(Level& L, const HostPattern& P, int32_t** slot_row, hipStream_t s) {
    L.n = P.n;
    L.nnzb = P.nnzb;
    SNS_TRY(dev_upload(&L.rowptr, P.rowptr, s));
    SNS_TRY(dev_upload(&L.colind, P.colind, s));
    SNS_TRY(dev_upload(&L.diag, P.diag, s));
    SNS_TRY(dev_alloc(&L.vals, (size_t)P.nnzb * 16));
    SNS_TRY(dev_alloc(&L.dinv, (size_t)P.n * 16));
    SNS_TRY(dev_alloc(slot_row, (size_t)P.nnzb));
    hipLaunchKernelGGL(k_fill_slot_row, dim3((P.n + 255) / 256), dim3(256), 0, s, P.n, L.rowptr, *slot_row);
    return SNS_OK;
}

// global sums of a few host doubles (collective; identity without a communicator)
int global_sum(sns_ctx* h, double* v, int count) {
    Comm* c = h->comm.get();
    if (!c || !c->active() || c->nranks <= 1) return SNS_OK;
    HIP_TRY(hipMemcpy(h->d_scal + 64, v, count * sizeof(double), hipMemcpyHostToDevice));
    SNS_TRY(comm_allreduce_sum(c, h->d_scal + 64, count, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    SNS_TRY(peer_check(c));
    HIP_TRY(hipMemcpy(v, h->d_scal + 64, count * sizeof(double), hipMemcpyDeviceToHost));
    return SNS_OK;
}

// host-side all-gather of `mine` (same length on every rank) through the communicator
int host_allgather(sns_ctx* h, const std::vector<double>& mine, std::vector<double>& all) {
    Comm* c = h->comm.get();
    const size_t len = mine.size();
    double *ds = nullptr, *dr = nullptr;
    auto body = [&]() -> int {
        SNS_TRY(dev_alloc(&ds, std::max<size_t>(1, len)));
        SNS_TRY(dev_alloc(&dr, std::max<size_t>(1, len * c->nranks)));
        HIP_TRY(hipMemcpy(ds, mine.data(), len * sizeof(double), hipMemcpyHostToDevice));
        SNS_TRY(comm_allgather(c, ds, dr, (int)len, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        SNS_TRY(peer_check(c));
        all.resize(len * c->nranks);
        HIP_TRY(hipMemcpy(all.data(), dr, all.size() * sizeof(double), hipMemcpyDeviceToHost));
        return SNS_OK;
    };
    const int rc = body();
    if (ds) (void)hipFree(ds);
    if (dr) (void)hipFree(dr);
    return rc;
}

// Every link of a halo plan must be posted by BOTH ends with matching counts -- rank a sends s nodes to b <=> b receives s nodes
// from a, zero included: comm_exchange posts no ncclSend / ncclRecv for an empty direction, so the peer must not post the matching
// call either.  An asymmetric plan deadlocks RCCL where the team transport only reports an error, so every level's plan is checked
// when it is made (hierarchy build, collective): one all-gather of 2 * nranks counts per rank, and every rank reaches the same
// verdict from the same table, i.e. a bad plan ends the run on all ranks instead of hanging some of them.
int check_plan_symmetry(sns_ctx* h, const Plan& p, int level) {
    Comm* c = h->comm.get();
    if (!c || !c->active() || c->nranks <= 1) return SNS_OK;
    const int nr = c->nranks;
    std::vector<double> mine((size_t)2 * nr, 0.0), all;
    bool bad_peer = p.send_ptr.size() != p.nbr.size() + 1 || p.recv_ptr.size() != p.nbr.size() + 1;
    for (size_t k = 0; k < p.nbr.size() && !bad_peer; ++k) {
        const int peer = p.nbr[k];
        if (peer < 0 || peer >= nr || peer == c->rank) { bad_peer = true; break; }
        mine[(size_t)peer] += (double)(p.send_ptr[k + 1] - p.send_ptr[k]);
        mine[(size_t)nr + peer] += (double)(p.recv_ptr[k + 1] - p.recv_ptr[k]);
    }
    if (bad_peer) mine[(size_t)c->rank] = -1.0;                 // (a rank never sends to itself: the slot doubles as the error flag)
    SNS_TRY(host_allgather(h, mine, all));
    for (int a = 0; a < nr; ++a) {
        if (all[(size_t)a * 2 * nr + a] != 0.0) {
            set_error("halo plan of level " + std::to_string(level) + ": rank " + std::to_string(a) + " lists an invalid neighbour");
            return SNS_E_COMM;
        }
        for (int b = 0; b < nr; ++b) {
            const double sent = all[(size_t)a * 2 * nr + b], expected = all[(size_t)b * 2 * nr + nr + a];
            if (sent != expected) {
                set_error("halo plan of level " + std::to_string(level) + " is asymmetric: rank " + std::to_string(a) + " sends " +
                          std::to_string((long long)sent) + " nodes to rank " + std::to_string(b) + ", which expects " +
                          std::to_string((long long)expected));
                return SNS_E_COMM;
            }
        }
    }
    return SNS_OK;
}

// peer transport: wire an uploaded plan to the neighbours' windows (collective, like check_plan_symmetry before it)
int connect_plan(sns_ctx* h, Plan& p) {
    Comm* c = h->comm.get();
    if (!c || !c->peer) return SNS_OK;
    PlanOffers t;
    const int rc = peer_plan_offer(c, p, t);
    if (rc != SNS_OK) t.mine.assign((size_t)3 * c->nranks + 1, -2.0);  // (still take part in the all-gather: the peers must not hang)
    SNS_TRY(host_allgather(h, t.mine, t.all));
    if (rc != SNS_OK) return rc;
    for (double v : t.all)
        if (v == -2.0) { set_error("peer transport: a rank could not place the plan in its window"); return SNS_E_COMM; }
    return peer_plan_connect(c, p, t);
}

int append_level(sns_ctx* h, const HostPattern& P, int32_t n_owned, bool with_xg) {
    h->levels.emplace_back();
    h->slot_row.push_back(nullptr);
    h->empty_c.push_back(nullptr);
    h->pong.push_back(nullptr);
    Level& C = h->levels.back();
    SNS_TRY(upload_pattern(C, P, &h->slot_row.back(), h->stream));
    C.n_owned = n_owned;
    C.n_global = n_owned;                        // (append_level serves the replicated tail: every rank holds all rows)
    SNS_TRY(alloc_level_vectors(C));
    SNS_TRY(dev_alloc(&h->pong.back(), 4 * (size_t)std::max(1, C.n)));
    HIP_TRY(hipMemset(h->pong.back(), 0, 4 * (size_t)std::max(1, C.n) * sizeof(double)));
    if (with_xg) {
        SNS_TRY(dev_alloc(&C.xg, 4 * (size_t)std::max(1, C.n)));
        HIP_TRY(hipMemset(C.xg, 0, 4 * (size_t)std::max(1, C.n) * sizeof(double)));
    }
    return SNS_OK;
}

// Aggregate-block Jacobi smoother (amg_block_smooth, csrc/sns_block.hip), symbolic part: the member rows of every aggregate of
// level L padded to 8 slots.  Levels whose aggregates can have more than 8 members (amg_agg_size > 8) keep the nodal blocks.
// Aggregate blocks on the FINE level: always with amg_block_smooth = 2; with 1 on a partitioned handle whose share of the fine level is
// at most amg_block_fine_rows rows per rank -- the latency-bound strong split, where 20 % fewer iterations (and collectives) outweigh
// the inverse blocks' bytes.  Global counts only: every rank answers alike.
inline bool fine_blocks_wanted(const sns_ctx* h) {
    if (h->opt.amg_block_smooth >= 2) return true;
    if (h->opt.amg_block_smooth < 1 || h->opt.amg_block_fine_rows <= 0) return false;
    const Comm* c = h->comm.get();
    if (!c || !c->active() || c->nranks < 2 || h->n_global_fine <= 0) return false;
    return h->n_global_fine <= (int64_t)h->opt.amg_block_fine_rows * c->nranks;
}
int upload_block_rows(sns_ctx* h, int l, Level& L, const std::vector<int32_t>& m_ptr, const std::vector<int32_t>& m_idx,
                      int32_t nc_owned) {
    const int mode = h->opt.amg_block_smooth;
    if (mode <= 0 || (l == 0 && !fine_blocks_wanted(h)) || nc_owned < 0) return SNS_OK;
    // blocks = aggregates; an aggregate of more than 8 nodes (a leftover node joined a full neighbour) is split in member order
    std::vector<int32_t> rows, of((size_t)std::max(1, L.n), -1);
    rows.reserve((size_t)8 * std::max(1, nc_owned));
    int32_t nb = 0;
    for (int32_t G = 0; G < nc_owned; ++G) {
        const int32_t k0 = m_ptr[(size_t)G], k1 = m_ptr[(size_t)G + 1];
        for (int32_t k = k0; k < k1; k += 8) {
            for (int32_t q = 0; q < 8; ++q) {
                const int32_t node = (k + q < k1) ? m_idx[(size_t)k + q] : -1;
                rows.push_back(node);
                if (node >= 0) of[(size_t)node] = nb;
            }
            ++nb;
        }
    }
    if (rows.empty()) rows.assign(8, -1);
    L.n_blk = nb;
    SNS_TRY(dev_upload(&L.blk_rows, rows, h->stream));
    SNS_TRY(dev_upload(&L.blk_of, of, h->stream));
    return SNS_OK;
}
// Is level l smoothed with the aggregate blocks?  (options only, no device state: every rank of a partitioned run must answer alike)
inline bool block_active(const sns_ctx* h, int l) {
    if (h->opt.amg_block_smooth <= 0 || h->opt.amg_f32_matrix == 0 || h->opt.pc_type != SNS_PC_AMG) return false;
    if (l < 0 || l + 1 >= (int)h->levels.size()) return false;                 // the coarsest level is solved or point-smoothed
    const Level& L = h->levels[l];
    if (!L.blk_rows) return false;
    if (l == 0 && !fine_blocks_wanted(h)) return false;
    if (h->rep_level > 0 && l == h->rep_level - 1) return false;               // only the source of the replicated copy
    // latency-bound levels only (rows per rank, the same figure on every rank)
    const bool replicated = h->rep_level > 0 && l >= h->rep_level;
    const int nr = (h->comm && h->comm->active() && !replicated) ? std::max(1, h->comm->nranks) : 1;
    if (h->opt.amg_block_max_rows > 0 && L.n_global > (int64_t)h->opt.amg_block_max_rows * nr) return false;
    return true;
}

// one smoothing sweep y = x + w S (b - A x) of level l: S = the aggregates' inverse blocks where block_active, else the nodal D^-1
void launch_sweep(sns_ctx* h, int l, const Level& L, int32_t rows, const double* x, double* y, const double* b, double omega) {
    if (block_active(h, l) && L.binv32) {
        const int32_t ns = 8 * L.n_blk;
        const unsigned grid = (unsigned)((ns + 63) / 64);
        if (grid == 0) return;
        if (L.binv_fmt == 2)
            hipLaunchKernelGGL((k_bsweep<2, 0>), dim3(grid), dim3(256), 0, h->stream, ns, L.blk_rows, L.rowptr, L.colind,
                               (const void*)L.vals16, L.scale16, (const void*)L.binv32, x, y, b, omega, GhostSrc());
        else
            hipLaunchKernelGGL((k_bsweep<1, 0>), dim3(grid), dim3(256), 0, h->stream, ns, L.blk_rows, L.rowptr, L.colind,
                               (const void*)L.vals32, (const float*)nullptr, (const void*)L.binv32, x, y, b, omega, GhostSrc());
        return;
    }
    launch_pc_spmv<SPMV_JACOBI>(h, L, rows, x, y, b, omega);
}
// first sweep of a cycle from the zero guess, z = w S b (omega = 1: S b alone, the spectral estimate's operator)
void launch_first_sweep(sns_ctx* h, int l, const Level& L, int32_t rows, const double* b, double omega, double* z) {
    if (rows <= 0) return;
    const int g4 = (int)((4 * (int64_t)rows + 255) / 256);
    if (block_active(h, l) && L.binv32) {
        const int32_t ns = 8 * L.n_blk;
        if (L.binv_fmt == 2)
            hipLaunchKernelGGL((k_bfirst<2>), dim3((unsigned)((ns + 63) / 64)), dim3(256), 0, h->stream, ns, L.blk_rows,
                               (const void*)L.binv32, b, omega, z);
        else
            hipLaunchKernelGGL((k_bfirst<1>), dim3((unsigned)((ns + 63) / 64)), dim3(256), 0, h->stream, ns, L.blk_rows,
                               (const void*)L.binv32, b, omega, z);
    } else if (lp_format(h, L) != 0 && L.dinv32) {
        hipLaunchKernelGGL(k_bjacobi32, dim3(g4), dim3(256), 0, h->stream, rows, L.dinv32, b, omega, z);
    } else {
        hipLaunchKernelGGL(k_bjacobi, dim3(g4), dim3(256), 0, h->stream, rows, L.dinv, b, omega, z);
    }
}

// symbolic part of M = A P of a level (fused first post-smoothing sweep, k_post_lp): pattern + gather lists -> device
// (a rank WITHOUT owned rows uploads the empty pattern all the same: whether a level takes the fused post-sweep -- one level-(l+1)
// exchange -- or the prolongation + level-l halo is decided from these arrays, and every rank must take the same branch)
int upload_ap(sns_ctx* h, Level& L, const HostPattern& fine, int32_t n_rows, const std::vector<int32_t>& agg_all) {
    if (n_rows < 0) n_rows = 0;
    HostAP M;
    try {
        build_ap_pattern(fine, n_rows, agg_all, M);
    } catch (const std::exception& e) {
        set_error(e.what());
        return SNS_E_MESH;
    }
    L.ap_nnz = M.nnz;
    SNS_TRY(dev_upload(&L.ap_rowptr, M.rowptr, h->stream));
    SNS_TRY(dev_upload(&L.ap_colind, M.colind, h->stream));
    SNS_TRY(dev_upload(&L.ap_ptr, M.ap_ptr, h->stream));
    SNS_TRY(dev_upload(&L.ap_idx, M.ap_idx, h->stream));
#ifdef SNS_HARNESS
    if (std::getenv("SNS_AP_GENERIC")) std::fill(M.nib.begin(), M.nib.end(), ~0ull);      // A/B: every row through the one-block-per-step loops
#endif
    SNS_TRY(dev_upload(&L.ap_nib, M.nib, h->stream));
    return SNS_OK;
}

// The coarsest level's direct solve: <= max(amg_coarse_size, 40) nodes take the one-workgroup inverse with partial pivoting of
// rounds 1-3, up to amg_dense_rows nodes the blocked Gauss-Jordan inverse on the matrix cores (csrc/sns_dense.hip); a larger
// last level (amg_max_levels reached) is smoothed.
int alloc_coarsest_solver(sns_ctx* h, Level& last) {
    const sns_options& o = h->opt;
    if (last.n <= std::max(o.amg_coarse_size, 40)) {
        const size_t N = 4 * (size_t)last.n;
        SNS_TRY(dev_alloc(&last.dense_inv, N * N));
        SNS_TRY(dev_alloc(&h->d_piv, N));
    } else if (last.n <= o.amg_dense_rows) {
        const int Np = (4 * last.n + 63) / 64 * 64;
        last.dense_np = Np;
        SNS_TRY(dev_alloc(&last.dense_gj, (size_t)Np * Np));
        SNS_TRY(dev_alloc(&last.dense_work, dense_gj_work_doubles(Np)));
        SNS_TRY(dev_alloc(&last.dense_x32, (size_t)Np * Np));
    }
    return SNS_OK;
}
// rows at or below which a level >= 1 ends the hierarchy (it is solved directly)
inline int coarsest_rows(const sns_options& o) { return std::max(o.amg_coarse_size, std::min(o.amg_dense_rows, 4096)); }

// Multi-GPU: from level R on, every rank holds the GLOBAL operator (values all-gathered at every numeric setup)
// and cycles the rest of the hierarchy redundantly: no exchanges below R, and the smoothing there is the exact
// global block-Jacobi instead of a rank-local one (thin partitions lose their convergence on the deep levels
// otherwise).  `cur` is the local pattern of level R (owned rows, local column ids), collective over the ranks.
int build_replicated_tail(sns_ctx* h, int R, const HostPattern& cur, int32_t n_owned) {
    const sns_options& o = h->opt;
    Comm* c = h->comm.get();
    const int nr = c->nranks, me = c->rank;
    std::vector<double> cnt((size_t)2 * nr, 0.0);
    cnt[me] = (double)n_owned;
    cnt[nr + me] = (double)cur.rowptr[n_owned];
    SNS_TRY(global_sum(h, cnt.data(), 2 * nr));
    std::vector<int64_t> off((size_t)nr + 1, 0);
    int32_t maxn = 1;
    int64_t maxnz = 1;
    for (int r = 0; r < nr; ++r) {
        off[r + 1] = off[r] + (int64_t)cnt[r];
        maxn = std::max(maxn, (int32_t)cnt[r]);
        maxnz = std::max(maxnz, (int64_t)cnt[nr + r]);
    }
    const int32_t NG = (int32_t)off[nr];
    const std::vector<int32_t>& g_own = h->ghost_own[R];
    const std::vector<int32_t>& g_gid = h->ghost_gid[R];
    // [0, maxn): row lengths; [maxn, maxn + maxnz): global column ids of my slots
    std::vector<double> mine((size_t)maxn + (size_t)maxnz, -1.0), all;
    for (int32_t i = 0; i < maxn; ++i) mine[i] = i < n_owned ? (double)(cur.rowptr[i + 1] - cur.rowptr[i]) : 0.0;
    for (int32_t sidx = 0; sidx < cur.rowptr[n_owned]; ++sidx) {
        const int32_t j = cur.colind[sidx];
        int64_t gj;
        if (j < n_owned) gj = off[me] + j;
        else {
            const size_t q = (size_t)(j - n_owned);
            if (q >= g_own.size()) { set_error("replicated tail: ghost column without an owner record"); return SNS_E_STATE; }
            gj = off[g_own[q]] + g_gid[q];
        }
        mine[(size_t)maxn + sidx] = (double)gj;
    }
    SNS_TRY(host_allgather(h, mine, all));
    HostPattern G;
    G.n = NG;
    G.rowptr.assign((size_t)NG + 1, 0);
    std::vector<int32_t> valmap((size_t)nr * maxnz, -1), rowmap((size_t)std::max(1, NG), 0);
    const size_t LEN = mine.size();
    for (int r = 0; r < nr; ++r)
        for (int32_t i = 0; i < (int32_t)cnt[r]; ++i) {
            G.rowptr[(size_t)off[r] + i + 1] = (int32_t)all[r * LEN + i];
            rowmap[(size_t)off[r] + i] = r * maxn + i;
        }
    for (int32_t g = 0; g < NG; ++g) G.rowptr[g + 1] += G.rowptr[g];
    G.nnzb = G.rowptr[NG];
    G.colind.resize((size_t)G.nnzb);
    G.diag.assign((size_t)NG, 0);
    std::vector<std::pair<int32_t, int32_t>> ent;
    for (int r = 0; r < nr; ++r) {
        int64_t src = 0;
        for (int32_t i = 0; i < (int32_t)cnt[r]; ++i) {
            const int32_t g = (int32_t)off[r] + i;
            const int32_t len = (int32_t)all[r * LEN + i];
            ent.clear();
            for (int32_t k = 0; k < len; ++k, ++src)
                ent.emplace_back((int32_t)all[r * LEN + maxn + src], (int32_t)(r * maxnz + src));
            std::sort(ent.begin(), ent.end());
            bool has_diag = false;
            for (int32_t k = 0; k < len; ++k) {
                const int32_t slot = G.rowptr[g] + k;
                if (ent[k].first < 0 || ent[k].first >= NG || (k > 0 && ent[k].first == ent[k - 1].first)) {
                    set_error("replicated tail: inconsistent global pattern");
                    return SNS_E_STATE;
                }
                G.colind[slot] = ent[k].first;
                valmap[ent[k].second] = slot;
                if (ent[k].first == g) { G.diag[g] = slot; has_diag = true; }
            }
            if (!has_diag) { set_error("replicated tail: row without a diagonal block"); return SNS_E_STATE; }
        }
    }
    h->rep_level = (int)h->levels.size();
    h->rep_maxn = maxn;
    h->rep_maxnz = maxnz;
    h->rep_NG = NG;
    h->rep_off = (int32_t)off[me];
    {
        // window transports: the right-hand sides go straight to their rows of the replicated level (comm_allgatherv) ...
        std::vector<int64_t> doff((size_t)nr), dcnt((size_t)nr);
        for (int r = 0; r < nr; ++r) { doff[(size_t)r] = 4 * off[r]; dcnt[(size_t)r] = 4 * (int64_t)cnt[r]; }
        SNS_TRY(dev_upload(&h->rep_doff, doff, h->stream));
        SNS_TRY(dev_upload(&h->rep_dcnt, dcnt, h->stream));
        // ... and the level above the source reads the coarse solution of its fused correction + post-sweep straight from the
        // replicated solution: the columns of its M = A P (local ids of level R: owned, then ghosts) in the replicated level's ids
        Level& A = h->levels[R - 1];
        if (R >= 2 && A.ap_colind && A.ap_nnz > 0) {
            std::vector<int32_t> col((size_t)A.ap_nnz);
            HIP_TRY(hipMemcpy(col.data(), A.ap_colind, col.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            for (auto& j : col) {
                if (j < n_owned) j = (int32_t)off[me] + j;
                else {
                    const size_t q = (size_t)(j - n_owned);
                    if (q >= g_own.size()) { set_error("replicated tail: ghost aggregate without an owner record"); return SNS_E_STATE; }
                    j = (int32_t)off[g_own[q]] + g_gid[q];
                }
            }
            SNS_TRY(dev_upload(&A.ap_colind_rep, col, h->stream));
        }
    }
    SNS_TRY(append_level(h, G, NG, false));
    h->ghost_own.emplace_back();
    h->ghost_gid.emplace_back();
    SNS_TRY(dev_upload(&h->rep_valmap, valmap, h->stream));
    SNS_TRY(dev_upload(&h->rep_rowmap, rowmap, h->stream));
    SNS_TRY(dev_alloc(&h->rep_vsend, (size_t)maxnz * 16));
    SNS_TRY(dev_alloc(&h->rep_vrecv, (size_t)maxnz * 16 * nr));
    SNS_TRY(dev_alloc(&h->rep_bsend, (size_t)maxn * 4));
    SNS_TRY(dev_alloc(&h->rep_brecv, (size_t)maxn * 4 * nr));
    HIP_TRY(hipMemset(h->rep_vsend, 0, (size_t)maxnz * 16 * sizeof(double)));
    HIP_TRY(hipMemset(h->rep_bsend, 0, (size_t)maxn * 4 * sizeof(double)));
    // plain serial aggregation below (identical on every rank: same input, deterministic code)
    HostPattern curp = std::move(G);
    int32_t n_own = NG;
    for (int l = h->rep_level; (int)h->levels.size() < o.amg_max_levels + 1; ++l) {
        if (n_own <= coarsest_rows(o)) break;
        std::vector<int32_t> agg;
        int32_t nc = 0;
        aggregate_nodes(curp, n_own, std::min(255, std::max(2, o.amg_agg_size)), agg, nc);
        if (nc >= n_own || nc == 0) break;
        HostAggregation A;
        build_coarse_from_agg(curp, n_own, agg, nc, nc, A);
        {
            Level& L = h->levels[l];
            L.nc = nc;
            SNS_TRY(dev_upload(&L.agg, A.agg, h->stream));
            SNS_TRY(dev_upload(&L.m_ptr, A.m_ptr, h->stream));
            SNS_TRY(dev_upload(&L.m_idx, A.m_idx, h->stream));
            SNS_TRY(dev_upload(&L.r_ptr, A.r_ptr, h->stream));
            SNS_TRY(dev_upload(&L.r_idx, A.r_idx, h->stream));
            SNS_TRY(upload_block_rows(h, l, L, A.m_ptr, A.m_idx, nc));
            SNS_TRY(upload_ap(h, L, curp, n_own, A.agg));
        }
        SNS_TRY(append_level(h, A.coarse, nc, false));
        h->ghost_own.emplace_back();
        h->ghost_gid.emplace_back();
        curp = std::move(A.coarse);
        n_own = nc;
    }
    SNS_TRY(alloc_coarsest_solver(h, h->levels.back()));
    h->tm.amg_levels = (int)h->levels.size() - 1;
    return SNS_OK;
}

// Build the aggregation hierarchy (symbolic, once per mesh; collective over the ranks).
// Aggregates never cross ranks, but the Galerkin operators keep every cross-rank coupling:
// a ghost fine node's aggregate becomes a ghost coarse node, and each level gets its own
// halo plan derived from the finer one.  With one rank this is plain serial aggregation.
int build_hierarchy(sns_ctx* h, const HostPattern& fine) {
    const sns_options& o = h->opt;
    Comm* c = h->comm.get();
    const bool dist = c && c->active() && c->nranks > 1;
    HostPattern cur = fine;
    int32_t n_owned = h->n_owned;
    std::vector<double> cur_pts = h->dim == 3 ? std::move(h->host_pts) : std::vector<double>();   // coordinates of `cur`'s nodes (coarse: centroids)
    h->host_pts = std::vector<double>();
    {
        double ng[1] = {(double)h->n_owned};
        SNS_TRY(global_sum(h, ng, 1));
        h->n_global_fine = (int64_t)ng[0];
        h->n_global_l1 = 0;
        h->levels[0].n_global = h->n_global_fine;
    }
    h->ghost_gid.assign(1, {});
    h->ghost_own.assign(1, {});
    const int per_rank_coarse = dist ? std::max(1, o.amg_coarse_size / c->nranks) : o.amg_coarse_size;
    if (dist && !h->levels[0].xg) {
        SNS_TRY(dev_alloc(&h->levels[0].xg, 4 * (size_t)h->levels[0].n));
        HIP_TRY(hipMemset(h->levels[0].xg, 0, 4 * (size_t)h->levels[0].n * sizeof(double)));
    }
    for (int l = 0; l + 1 < o.amg_max_levels; ++l) {
        if (dist && l >= 1 && o.amg_replicate_rows > 0) {
            double g[1] = {(double)n_owned};
            SNS_TRY(global_sum(h, g, 1));
            // the replicated level must fit the scratch vectors sized by the local fine level
            double fits[1] = {g[0] <= (double)h->n_owned ? 0.0 : 1.0};
            SNS_TRY(global_sum(h, fits, 1));
            if (g[0] <= (double)o.amg_replicate_rows && g[0] > (double)std::max(o.amg_coarse_size, 40) && fits[0] == 0.0)
                return build_replicated_tail(h, l, cur, n_owned);
        }
        double flag[1] = {n_owned > per_rank_coarse ? 1.0 : 0.0};
        SNS_TRY(global_sum(h, flag, 1));
        if (flag[0] == 0.0) break;
        if (!dist && l >= 1 && n_owned <= coarsest_rows(o)) break;       // serial: this level is solved directly
        std::vector<int32_t> agg;
        int32_t nc_owned = 0;
        aggregate_nodes(cur, n_owned, std::min(255, std::max(2, o.amg_agg_size)), agg, nc_owned,
                        cur_pts.size() == (size_t)3 * cur.n ? cur_pts.data() : nullptr);
        double prog[2] = {(double)n_owned, (double)nc_owned};
        SNS_TRY(global_sum(h, prog, 2));
        if (prog[1] >= prog[0] || prog[1] == 0.0) break;      // no progress anywhere
        if (l == 0) h->n_global_l1 = (int64_t)prog[1];
        int32_t nc_total = nc_owned;
        Plan cplan;
        std::vector<int32_t> g_own, g_gid;                     // ghost coarse nodes: owner rank, owner-local id
        Level& L = h->levels[l];
        if (dist) {
            const Plan& p = c->plans[l];
            std::vector<double> ids((size_t)4 * cur.n, -1.0);
            for (int32_t i = 0; i < n_owned; ++i) ids[(size_t)4 * i] = (double)agg[i];
            HIP_TRY(hipMemcpy(L.xg, ids.data(), ids.size() * sizeof(double), hipMemcpyHostToDevice));
            SNS_TRY(comm_exchange(c, p, L.xg, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            HIP_TRY(hipMemcpy(ids.data(), L.xg, ids.size() * sizeof(double), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemset(L.xg, 0, ids.size() * sizeof(double)));
            cplan.nbr = p.nbr;
            cplan.n_own = nc_owned;
            cplan.send_ptr.assign(1, 0);
            cplan.recv_ptr.assign(1, 0);
            for (size_t k = 0; k < p.nbr.size(); ++k) {
                std::vector<int32_t> u;
                for (int32_t q = p.recv_ptr[k]; q < p.recv_ptr[k + 1]; ++q) {
                    const int32_t rid = (int32_t)ids[(size_t)4 * p.h_recv_idx[q]];
                    if (rid < 0) { set_error("hierarchy: ghost node without an aggregate on its owner"); return SNS_E_COMM; }
                    u.push_back(rid);
                }
                std::sort(u.begin(), u.end());
                u.erase(std::unique(u.begin(), u.end()), u.end());
                for (int32_t q = p.recv_ptr[k]; q < p.recv_ptr[k + 1]; ++q) {
                    const int32_t gnode = p.h_recv_idx[q];
                    const int32_t rid = (int32_t)ids[(size_t)4 * gnode];
                    agg[gnode] = nc_total + (int32_t)(std::lower_bound(u.begin(), u.end(), rid) - u.begin());
                }
                for (size_t q = 0; q < u.size(); ++q) {
                    cplan.h_recv_idx.push_back(nc_total + (int32_t)q);
                    g_own.push_back(p.nbr[k]);
                    g_gid.push_back(u[q]);
                }
                nc_total += (int32_t)u.size();
                cplan.recv_ptr.push_back((int32_t)cplan.h_recv_idx.size());
                std::vector<int32_t> sset;
                for (int32_t q = p.send_ptr[k]; q < p.send_ptr[k + 1]; ++q) sset.push_back(agg[p.h_send_idx[q]]);
                std::sort(sset.begin(), sset.end());
                sset.erase(std::unique(sset.begin(), sset.end()), sset.end());
                cplan.h_send_idx.insert(cplan.h_send_idx.end(), sset.begin(), sset.end());
                cplan.send_ptr.push_back((int32_t)cplan.h_send_idx.size());
            }
        }
        HostAggregation A;
        build_coarse_from_agg(cur, n_owned, agg, nc_owned, nc_total, A);
        L.nc = nc_owned;
        SNS_TRY(dev_upload(&L.agg, A.agg, h->stream));
        SNS_TRY(dev_upload(&L.m_ptr, A.m_ptr, h->stream));
        SNS_TRY(dev_upload(&L.m_idx, A.m_idx, h->stream));
        SNS_TRY(dev_upload(&L.r_ptr, A.r_ptr, h->stream));
        SNS_TRY(dev_upload(&L.r_idx, A.r_idx, h->stream));
        SNS_TRY(upload_block_rows(h, l, L, A.m_ptr, A.m_idx, nc_owned));
        // M = A P for the fused first post-smoothing sweep: every level of a serial hierarchy; in a partitioned one the fine
        // level only (its single post-sweep is the exact global sweep; the distributed coarse levels smooth rank-locally)
        // ... and, on the window transports, every partitioned level: the exact-sweep cycle (level_exact) takes the fused post-sweep too
        if (!dist || l == 0 || c->windows()) SNS_TRY(upload_ap(h, L, cur, n_owned, A.agg));
        h->levels.emplace_back();
        h->slot_row.push_back(nullptr);
        h->empty_c.push_back(nullptr);
        h->pong.push_back(nullptr);
        Level& C = h->levels.back();
        if (&h->levels[l] != &L) { set_error("internal: level storage moved"); return SNS_E_STATE; }
        SNS_TRY(upload_pattern(C, A.coarse, &h->slot_row.back(), h->stream));
        C.n_owned = nc_owned;
        C.n_global = (int64_t)prog[1];
        SNS_TRY(alloc_level_vectors(C));
        SNS_TRY(dev_alloc(&h->pong.back(), 4 * (size_t)C.n));
        HIP_TRY(hipMemset(h->pong.back(), 0, 4 * (size_t)C.n * sizeof(double)));
        if (dist) {
            SNS_TRY(dev_alloc(&C.xg, 4 * (size_t)C.n));
            HIP_TRY(hipMemset(C.xg, 0, 4 * (size_t)C.n * sizeof(double)));
            SNS_TRY(check_plan_symmetry(h, cplan, l + 1));
            SNS_TRY(plan_upload(cplan));
            SNS_TRY(connect_plan(h, cplan));
            c->plans.push_back(std::move(cplan));
        }
        h->ghost_own.push_back(std::move(g_own));
        h->ghost_gid.push_back(std::move(g_gid));
        if (l == 0) {
            SNS_TRY(dev_alloc(&h->empty_c[0], 4 * (size_t)std::max(1, nc_owned)));
            if (nc_owned > 0)
                hipLaunchKernelGGL(k_empty_coarse, dim3((unsigned)((4 * (int64_t)nc_owned + 255) / 256)), dim3(256), 0,
                                   h->stream, nc_owned, L.m_ptr, L.m_idx, L.free_mask, h->empty_c[0]);
        }
        if (cur_pts.size() == (size_t)3 * cur.n) {
            std::vector<double> cp((size_t)3 * nc_total, 0.0), cnt((size_t)nc_total, 0.0);
            for (int32_t i = 0; i < cur.n; ++i) {
                const int32_t I = A.agg[(size_t)i];
                if (I < 0) continue;
                for (int c3 = 0; c3 < 3; ++c3) cp[3 * (size_t)I + c3] += cur_pts[3 * (size_t)i + c3];
                cnt[(size_t)I] += 1.0;
            }
            for (int32_t I = 0; I < nc_total; ++I)
                if (cnt[(size_t)I] > 0.0) for (int c3 = 0; c3 < 3; ++c3) cp[3 * (size_t)I + c3] /= cnt[(size_t)I];
            cur_pts = std::move(cp);
        } else {
            cur_pts.clear();
        }
        cur = std::move(A.coarse);
        n_owned = nc_owned;
    }
    Level& last = h->levels.back();
    if (h->levels.size() > 1) {
        if (!dist) {
            SNS_TRY(alloc_coarsest_solver(h, last));
        } else {
            // global dense coarsest solve, replicated on every rank: rank r's node i -> padded id r*maxn + i
            std::vector<double> cnt(c->nranks, 0.0);
            cnt[c->rank] = (double)last.n_owned;
            SNS_TRY(global_sum(h, cnt.data(), c->nranks));
            int maxn = 0;
            h->cg_counts.resize(c->nranks);
            for (int r = 0; r < c->nranks; ++r) { h->cg_counts[r] = (int)cnt[r]; maxn = std::max(maxn, (int)cnt[r]); }
            const int N = 4 * c->nranks * std::max(1, maxn);
            if (N <= 640) {
                h->cg_maxn = std::max(1, maxn);
                h->cg_N = N;
                std::vector<int32_t> cmap((size_t)last.n, 0);
                for (int32_t i = 0; i < last.n_owned; ++i) cmap[i] = c->rank * h->cg_maxn + i;
                const auto& go = h->ghost_own.back();
                const auto& gg = h->ghost_gid.back();
                for (size_t q = 0; q < go.size(); ++q) cmap[(size_t)last.n_owned + q] = go[q] * h->cg_maxn + gg[q];
                SNS_TRY(dev_upload(&h->cg_colmap, cmap, h->stream));
                SNS_TRY(dev_alloc(&h->cg_rows, (size_t)4 * h->cg_maxn * N));
                SNS_TRY(dev_alloc(&h->cg_full, (size_t)N * N));
                SNS_TRY(dev_alloc(&h->cg_send, (size_t)4 * h->cg_maxn));
                SNS_TRY(dev_alloc(&h->cg_recv, (size_t)N));
                SNS_TRY(dev_alloc(&h->d_piv, (size_t)N));
            }
        }
    }
    h->tm.amg_levels = (int)h->levels.size();
    return SNS_OK;
}

int get_vec(sns_ctx* h, size_t k, double** out);

// 2-D handles (sns_create_2d): triangle P1-P1, Stokes with (stokes_viscosity, stokes_beta) and the UGN-stabilised
// NS form of LidDrivenNavierStokesFlow.py:123-143 / DFG_2D_Validation.py:141-163.  Always the scratch-free path:
// every BSR block by its owner lane, residual-only evaluations by one lane per triangle + the node gather.
int assemble2d(sns_ctx* h, int form, const double* w, double* F, bool want_matrix) {
    Level& L = h->levels[0];
    const unsigned go = (unsigned)((h->n_od + 255) / 256);
    const unsigned gd = (unsigned)((4 * (int64_t)h->n_owned + 255) / 256);
    const int64_t ndof = 4 * (int64_t)h->n;
    const int gv = vec_grid(ndof);
    if (h->E == 0) { set_error("empty mesh"); return SNS_E_ARG; }
    if (form == SNS_FORM_STOKES) {
        const double nu_s = h->opt.stokes_viscosity, beta = h->opt.stokes_beta;
        const double* state = h->gext;            // w == NULL: the system of LinearProblem(a, L, bcs), F(0) = lifting
        if (w) {                                  // linear residual at w: state = w with the Dirichlet data imposed
            double* tmp = nullptr;
            SNS_TRY(get_vec(h, 13, &tmp));
            HIP_TRY(hipMemcpyAsync(tmp, w, ndof * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            hipLaunchKernelGGL(k_snap_bc, dim3(gv), dim3(256), 0, h->stream, ndof, h->bc_mask, h->bc_val, 1e300, tmp);
            state = tmp;
        }
        if (want_matrix)
            hipLaunchKernelGGL((k_fused_offdiag<SNS_FORM_STOKES_2D, false>), dim3(go), dim3(256), 0, h->stream, h->n_od,
                               h->od_order, h->c_ptr, h->c_idx, h->slot_row[0], L.colind, h->tets, h->pts, state,
                               h->bc_mask, nu_s, beta, L.vals);
        if (want_matrix || F)
            hipLaunchKernelGGL((k_fused_diag<SNS_FORM_STOKES_2D, false>), dim3(gd), dim3(256), 0, h->stream, h->n_owned,
                               L.diag, h->c_ptr, h->c_idx, h->tets, h->pts, state, h->bc_mask, h->bc_val, nu_s, beta,
                               want_matrix ? L.vals : (double*)nullptr, F);
        if (w && F) hipLaunchKernelGGL(k_bc_residual, dim3(gv), dim3(256), 0, h->stream, ndof, h->bc_mask, h->bc_val, w, F);
    } else {
        const double nu = 1.0 / h->opt.reynolds;
        bool lifted = false;
        if (F) {
            hipLaunchKernelGGL(k_count_bc_violations, dim3(gv), dim3(256), 0, h->stream, ndof, h->bc_mask, h->bc_val, w,
                               h->partial);
            reduce_local(h, gv, 1, h->d_scal + 60);
            double nviol = 1.0;
            SNS_TRY(fetch(h, h->d_scal + 60, 1, &nviol));
            lifted = nviol != 0.0;
        }
        if (want_matrix) {
            hipLaunchKernelGGL((k_fused_offdiag<SNS_FORM_UGN_2D, false>), dim3(go), dim3(256), 0, h->stream, h->n_od,
                               h->od_order, h->c_ptr, h->c_idx, h->slot_row[0], L.colind, h->tets, h->pts, w, h->bc_mask,
                               nu, 0.0, L.vals);
            hipLaunchKernelGGL((k_fused_diag<SNS_FORM_UGN_2D, false>), dim3(gd), dim3(256), 0, h->stream, h->n_owned,
                               L.diag, h->c_ptr, h->c_idx, h->tets, h->pts, w, h->bc_mask, h->bc_val, nu, 0.0, L.vals, F);
        } else {
            if (!h->Fe) SNS_TRY(dev_alloc(&h->Fe, (size_t)h->E * 16));
            hipLaunchKernelGGL(k_residual_tri, dim3((unsigned)((h->E + 255) / 256)), dim3(256), 0, h->stream, h->E,
                               h->tets, h->pts, w, nu, h->Fe);
            hipLaunchKernelGGL(k_gather_residual, dim3(gd), dim3(256), 0, h->stream, h->n_owned, h->nt_ptr, h->nt_idx,
                               h->bc_mask, h->bc_val, w, h->Fe, F);
        }
        if (lifted && F) {                       // F += A0[:,B] (g - x_B)   (apply_lifting)
            double* dl = nullptr;
            SNS_TRY(get_vec(h, 13, &dl));
            hipLaunchKernelGGL(k_bc_defect, dim3(gv), dim3(256), 0, h->stream, ndof, h->bc_mask, h->bc_val, w, dl);
            hipLaunchKernelGGL((k_fused_lift<SNS_FORM_UGN_2D, false>), dim3(gd), dim3(256), 0, h->stream, h->n_owned,
                               L.diag, h->c_ptr, h->c_idx, h->tets, h->pts, w, h->bc_mask, dl, nu, F);
        }
    }
    if (want_matrix) {
        h->has_matrix = true;
        h->pc_ready = false;
        h->matrix_form = form;
    }
    HIP_TRY(hipGetLastError());
    return SNS_OK;
}

int assemble(sns_ctx* h, int form, const double* w, double* F, bool want_matrix) {
    if (form != SNS_FORM_STOKES && form != SNS_FORM_NS) { set_error("bad form"); return SNS_E_ARG; }
    if (form == SNS_FORM_NS && !w) { set_error("NS form needs a state vector"); return SNS_E_ARG; }
    if (h->dim == 2) return assemble2d(h, form, w, F, want_matrix);
    const int grid = (int)((h->E + EL_TETS_PER_BLOCK - 1) / EL_TETS_PER_BLOCK);
    const double nu = 1.0 / h->opt.reynolds;
    bool fast_residual = false;
    // (a perturbed form -- sns_set_form_variant -- exists in the staged element kernel only: Jacobian AND residual go through it)
    const bool variant = !h->fv.is_default();
    const bool try_fused = want_matrix && h->opt.assembly_fused && form == SNS_FORM_NS && h->E > 0 && !variant;
    if (((!want_matrix && F) || try_fused) && form == SNS_FORM_NS && h->E > 0) {
        // residual only: if the state satisfies the Dirichlet data there is no lifting term (:65) and the
        // one-lane-per-tet kernel applies; otherwise the general fused kernel computes the lifted blocks
        const int64_t ndof = 4 * (int64_t)h->n;
        const int gv = vec_grid(ndof);
        hipLaunchKernelGGL(k_count_bc_violations, dim3(gv), dim3(256), 0, h->stream, ndof, h->bc_mask, h->bc_val, w,
                           h->partial);
        reduce_local(h, gv, 1, h->d_scal + 60);
        double nviol = 1.0;
        SNS_TRY(fetch(h, h->d_scal + 60, 1, &nviol));
        fast_residual = (nviol == 0.0) && !variant;
    }
    Level& L = h->levels[0];
    if (form == SNS_FORM_STOKES && !w && want_matrix && h->opt.assembly_fused && h->E > 0) {
        // the Stokes system of solve_stokes_problem (:197-218): constant element blocks, right-hand side F(0) =
        // lifting A0[:,B] g (row a of A0 applied to the Dirichlet data extended by zero), F_B = -g
        const unsigned go = (unsigned)((h->n_od + 255) / 256);
        const unsigned gd = (unsigned)((4 * (int64_t)h->n_owned + 255) / 256);
        hipLaunchKernelGGL((k_fused_offdiag<SNS_FORM_STOKES, false>), dim3(go), dim3(256), 0, h->stream, h->n_od,
                           h->od_order, h->c_ptr, h->c_idx, h->slot_row[0], L.colind, h->tets, h->pts, h->gext,
                           h->bc_mask, nu, 0.0, L.vals);
        hipLaunchKernelGGL((k_fused_diag<SNS_FORM_STOKES, false>), dim3(gd), dim3(256), 0, h->stream, h->n_owned, L.diag,
                           h->c_ptr, h->c_idx, h->tets, h->pts, h->gext, h->bc_mask, h->bc_val, nu, 0.0, L.vals, F);
        h->has_matrix = true;
        h->pc_ready = false;
        h->matrix_form = form;
        HIP_TRY(hipGetLastError());
        return SNS_OK;
    }
    if (try_fused) {
        // scratch-free path: every BSR block (and every node residual) is computed by the lanes that own it; a
        // state that violates its Dirichlet data adds the lifting term in a third pass over the boundary tets
        const unsigned go = (unsigned)((h->n_od + 255) / 256);
        const unsigned gd = (unsigned)((4 * (int64_t)h->n_owned + 255) / 256);
        if (!h->opt.corrected_convection) {
            hipLaunchKernelGGL((k_fused_offdiag<SNS_FORM_NS, false>), dim3(go), dim3(256), 0, h->stream, h->n_od, h->od_order, h->c_ptr, h->c_idx,
                               h->slot_row[0], L.colind, h->tets, h->pts, w, h->bc_mask, nu, 0.0, L.vals);
            hipLaunchKernelGGL((k_fused_diag<SNS_FORM_NS, false>), dim3(gd), dim3(256), 0, h->stream, h->n_owned, L.diag, h->c_ptr,
                               h->c_idx, h->tets, h->pts, w, h->bc_mask, h->bc_val, nu, 0.0, L.vals, F);
        } else {
            hipLaunchKernelGGL((k_fused_offdiag<SNS_FORM_NS, true>), dim3(go), dim3(256), 0, h->stream, h->n_od, h->od_order, h->c_ptr, h->c_idx,
                               h->slot_row[0], L.colind, h->tets, h->pts, w, h->bc_mask, nu, 0.0, L.vals);
            hipLaunchKernelGGL((k_fused_diag<SNS_FORM_NS, true>), dim3(gd), dim3(256), 0, h->stream, h->n_owned, L.diag, h->c_ptr,
                               h->c_idx, h->tets, h->pts, w, h->bc_mask, h->bc_val, nu, 0.0, L.vals, F);
        }
        if (!fast_residual && F) {
            double* dl = nullptr;
            SNS_TRY(get_vec(h, 13, &dl));
            const int64_t ndof = 4 * (int64_t)h->n;
            hipLaunchKernelGGL(k_bc_defect, dim3(vec_grid(ndof)), dim3(256), 0, h->stream, ndof, h->bc_mask, h->bc_val, w, dl);
            if (!h->opt.corrected_convection)
                hipLaunchKernelGGL((k_fused_lift<SNS_FORM_NS, false>), dim3(gd), dim3(256), 0, h->stream, h->n_owned, L.diag, h->c_ptr,
                                   h->c_idx, h->tets, h->pts, w, h->bc_mask, dl, nu, F);
            else
                hipLaunchKernelGGL((k_fused_lift<SNS_FORM_NS, true>), dim3(gd), dim3(256), 0, h->stream, h->n_owned, L.diag, h->c_ptr,
                                   h->c_idx, h->tets, h->pts, w, h->bc_mask, dl, nu, F);
        }
        h->has_matrix = true;
        h->pc_ready = false;
        h->matrix_form = form;
        HIP_TRY(hipGetLastError());
        return SNS_OK;
    }
    if (want_matrix && !h->Ke) SNS_TRY(dev_alloc(&h->Ke, (size_t)h->E * 256));
    if (!h->Fe) SNS_TRY(dev_alloc(&h->Fe, (size_t)h->E * 16));
    double* Fe = F ? h->Fe : nullptr;
    if (fast_residual) {
        const unsigned gt = (unsigned)((h->E + 255) / 256);
        if (!h->opt.corrected_convection)
            hipLaunchKernelGGL((k_residual_tet<false>), dim3(gt), dim3(256), 0, h->stream, h->E, h->tets, h->pts, w, nu, h->Fe);
        else
            hipLaunchKernelGGL((k_residual_tet<true>), dim3(gt), dim3(256), 0, h->stream, h->E, h->tets, h->pts, w, nu, h->Fe);
    } else if (grid > 0) {
        if (form == SNS_FORM_STOKES)
            hipLaunchKernelGGL((k_element<SNS_FORM_STOKES, false>), dim3(grid), dim3(256), 0, h->stream, h->E, h->tets,
                               h->pts, w, h->bc_mask, h->bc_val, nu, want_matrix ? 1 : 0, h->Ke, Fe, h->fv);
        else if (!h->opt.corrected_convection)
            hipLaunchKernelGGL((k_element<SNS_FORM_NS, false>), dim3(grid), dim3(256), 0, h->stream, h->E, h->tets,
                               h->pts, w, h->bc_mask, h->bc_val, nu, want_matrix ? 1 : 0, h->Ke, Fe, h->fv);
        else
            hipLaunchKernelGGL((k_element<SNS_FORM_NS, true>), dim3(grid), dim3(256), 0, h->stream, h->E, h->tets,
                               h->pts, w, h->bc_mask, h->bc_val, nu, want_matrix ? 1 : 0, h->Ke, Fe, h->fv);
    }
    if (want_matrix) {
        const int64_t nth = L.nnzb * 8;
        hipLaunchKernelGGL(k_gather_matrix, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, h->stream, L.nnzb,
                           h->c_ptr, h->c_idx, h->slot_row[0], L.colind, h->bc_mask, h->Ke, L.vals);
        h->has_matrix = true;
        h->pc_ready = false;
        h->matrix_form = form;
    }
    if (F) {
        const int64_t nth = 4 * (int64_t)h->n_owned;
        hipLaunchKernelGGL(k_gather_residual, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, h->stream,
                           h->n_owned, h->nt_ptr, h->nt_idx, h->bc_mask, h->bc_val, w, h->Fe, F);
    }
    HIP_TRY(hipGetLastError());
    return SNS_OK;
}

// ---- preconditioner -----------------------------------------------------------
int get_vec(sns_ctx* h, size_t k, double** out);
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
    int add_l2 = 0, add_deep = 0;
    if (h->opt.amg_nu_scale_with_size) {
        // an unstructured mesh: the greedy aggregation reaches ~4.6 nodes per aggregate on a Delaunay mesh where a Kuhn box
        // gives 7.7-8.0 (sns_get_hierarchy), so that its hierarchy is as deep at 0.4 M rows as the structured one at 1.7 M and
        // its coarse operators are denser -- it gains from the first tier of extra sweeps already: config 4u (5 M-tet
        // body-centred Delaunay channel, 7 levels) 71 -> 58 iterations per Newton step and 145-148 -> 131-134 ms, where
        // the structured 10 M-tet duct (7 levels as well) pays +3 % for 43.5 -> 43.0 (scripts/gpu_r3_tierA.py)
        // (depth as rounds 1-3 counted it: a hierarchy that ends in the dense level of round 4 would have gone on for
        // ~log5(rows / amg_coarse_size) more levels)
        int nlev = (int)h->levels.size() - (h->rep_level > 0 ? 1 : 0);
        if (h->levels.back().dense_gj && h->levels.back().n > h->opt.amg_coarse_size)
            nlev += (int)std::ceil(std::log((double)h->levels.back().n / std::max(1, h->opt.amg_coarse_size)) / std::log(5.0));
        // (global counts: every rank must arrive at the same schedule -- levels with exchanged sweeps are collective)
        const bool small_aggregates = h->n_global_l1 > 0 && (double)h->n_global_fine < 6.0 * (double)h->n_global_l1;
        if (h->n_global_fine >= 20000000) { add_l2 = 6; add_deep = 10; }       // 192 M tets: 63 / 71 -> 55 / 66, -10 % time
        else if (h->n_global_fine >= 8000000) { add_l2 = 4; add_deep = 6; }
        else if (h->n_global_fine >= 2500000 || (nlev >= 7 && small_aggregates)) { add_l2 = 2; add_deep = 2; }
    }
    if (block_active(h, l) && ll >= 1) {
        // aggregate blocks: one sweep is worth about two nodal-block sweeps (level 1 of a single-GPU handle: 1 + amg_bnu_l1, see
        // level_sweeps); the size-scaled extra sweeps are halved likewise
        if (ll >= 3) return std::max(1, h->opt.amg_bnu_deep) + (add_deep + 1) / 2;
        if (ll == 2) return std::max(1, h->opt.amg_bnu_l2) + (add_l2 + 1) / 2;
        return std::max(1, h->opt.amg_bnu_l2);
    }
    int nu = std::max(1, h->opt.amg_nu);
    if (ll >= 3 && h->opt.amg_nu_deep > 0) nu = h->opt.amg_nu_deep + add_deep;
    else if (ll == 2 && h->opt.amg_nu_l2 > 0) nu = h->opt.amg_nu_l2 + add_l2;
    else if (ll >= 1 && h->opt.amg_nu_coarse > 0) nu = h->opt.amg_nu_coarse;
    return nu;
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
int estimate_lambda_max(sns_ctx* h, int l, double* out) {
    Level& L = h->levels[l];
    const int32_t rows = L.n_owned;
    const int64_t nd = 4 * (int64_t)rows;
    const int g = vec_grid(nd), g4 = (int)((nd + 255) / 256);
    double* x = h->pong[l];
    double* y = L.r;
    double* z = L.x;
    // deterministic start vector with all frequencies: x_i = 1 + (i*2654435761 mod 1024)/1024 via axpby on an iota is
    // overkill; use b of the last solve if any, else the diagonal-inverse row sums: simplest robust choice = all ones
    if (rows > 0) hipLaunchKernelGGL(k_fill_pattern, dim3(g), dim3(256), 0, h->stream, nd, x);
    double* zero = nullptr;
    if (lp_format(h, L) != 0) {
        SNS_TRY(get_vec(h, 13, &zero));                  // level sizes never exceed the fine level
        if (nd > 0) HIP_TRY(hipMemsetAsync(zero, 0, nd * sizeof(double), h->stream));
    }
    double lam = 0.0;
    const int iters = 12;
    // distributed levels whose sweeps see exchanged ghost values are damped for the GLOBAL operator; purely
    // rank-local sweeps (ghost values zero) for the rank-local one
    const bool glob = uses_ghosts_in_sweeps(h, l, L);
    for (int it = 0; it < iters; ++it) {
        if (glob) SNS_TRY(exchange_level(h, l, x));
        // the spectrum of the matrix the sweeps actually read: with a low-precision copy y = 0 - A~ x (the sign does not
        // matter to ||Dinv A x||), half the bytes of the fp64 pass
        if (lp_format(h, L) != 0 && zero) launch_pc_spmv<SPMV_B_MINUS_AX>(h, L, rows, x, y, zero, 0.0);
        else launch_spmv<SPMV_AX>(h, L, rows, x, y, nullptr, 0.0, nullptr);
        if (rows > 0) {
            if (block_active(h, l) && L.binv32) launch_first_sweep(h, l, L, rows, y, 1.0, z);      // the smoother's own blocks
            else hipLaunchKernelGGL(k_bjacobi, dim3(g4), dim3(256), 0, h->stream, rows, L.dinv, y, 1.0, z);
            hipLaunchKernelGGL(k_dot2, dim3(g), dim3(256), 0, h->stream, nd, x, z, h->partial);   // (x.z, z.z)
        }
        if (glob) SNS_TRY(reduce_to(h, g, 2, h->d_scal + 16 + 2 * it));
        else reduce_local(h, g, 2, h->d_scal + 16 + 2 * it);
        // normalise with the device-side norm: x = z / ||z||  (scale read on device)
        if (rows > 0)
            hipLaunchKernelGGL(k_scale_by_rsqrt, dim3(g), dim3(256), 0, h->stream, nd, h->d_scal + 16 + 2 * it + 1, z, x);
    }
    std::vector<double> v(2 * iters);
    SNS_TRY(fetch(h, h->d_scal + 16, 2 * iters, v.data()));
    // x was normalised each step, so ||z|| of the last steps estimates |lambda|max; take the max of the tail
    for (int it = iters - 3; it < iters; ++it) lam = std::max(lam, std::sqrt(v[2 * it + 1]));
    *out = lam;
    return SNS_OK;
}

// Stability limit of the smoother damping on level l from the dominant Ritz values of S A (S = the level's smoother blocks,
// nodal or aggregate): M = 8 Arnoldi steps from the deterministic start vector of the power iteration (classical Gram-Schmidt
// with one re-orthogonalisation, the FGMRES kernels; one host read per step), eigenvalues of the 8 x 8 Hessenberg matrix on the
// host (sns_host_hessenberg_eigs).  |1 - w theta| < 1 needs w < 2 Re(theta) / |theta|^2: *limit = the minimum over the Ritz
// values with |theta| >= 0.5 |theta|max (those a few Arnoldi steps have converged to).  The power iteration above sees the
// modulus only; on a convection-dominated coarse level the dominant eigenvalues are complex, and a level that runs 1 + 6
// sweeps amplifies a damping above the limit seven times per cycle (oracle/experiments/r4_damping.py).
inline void level_sweeps(const sns_ctx* h, int l, int& nu_pre, int& nu_post);
int arnoldi_ritz(sns_ctx* h, int l, double* theta_max, double* limit) {
    constexpr int M = 8;
    Level& L = h->levels[l];
    const int32_t rows = L.n_owned;
    const int64_t nd = 4 * (int64_t)rows;
    *theta_max = 0.0;
    *limit = 1e30;
    // levels whose sweeps see exchanged ghost values (level_exact): the GLOBAL operator's Ritz values -- one exchange per Arnoldi
    // step, the dots summed over the ranks; collective, so every rank goes through it whatever its row count.  (Levels that
    // exchange per sweep by amg_sweep_exchange_rows / the fine level's single post-sweep: not estimated, as in round 4.)
    const bool glob = level_exact(h, l);
    if (!glob && (rows <= 0 || uses_ghosts_in_sweeps(h, l, L))) return SNS_OK;
    const int g = std::max(1, vec_grid(nd));           // (a rank without rows on a collective level still launches: empty loops, zero partials)
    auto reduce = [&](int nred, double* dst) -> int {
        if (glob) return reduce_to(h, g, nred, dst);
        reduce_local(h, g, nred, dst);
        return SNS_OK;
    };
    if (h->arn_cap < (size_t)(M + 1) * nd) {
        if (h->arn_V) (void)hipFree(h->arn_V);
        h->arn_V = nullptr;
        SNS_TRY(dev_alloc(&h->arn_V, (size_t)(M + 1) * nd));
        h->arn_cap = (size_t)(M + 1) * nd;
    }
    double* V = h->arn_V;
    double* y = L.r;
    double* zero = nullptr;
    double* xin = nullptr;                              // the SpMV input needs the level's full length (ghost tail = 0)
    SNS_TRY(get_vec(h, 13, &zero));
    SNS_TRY(get_vec(h, 12, &xin));
    HIP_TRY(hipMemsetAsync(zero, 0, nd * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(xin, 0, 4 * (size_t)L.n * sizeof(double), h->stream));
    double* sc = h->d_scal + 192;                      // [0, 8) pass-1 coefficients, [8, 16) pass 2, [16, 18) (w.w, w.w)
    hipLaunchKernelGGL(k_fill_pattern, dim3(g), dim3(256), 0, h->stream, nd, V);
    hipLaunchKernelGGL(k_dot2, dim3(g), dim3(256), 0, h->stream, nd, V, V, h->partial);
    SNS_TRY(reduce(2, sc + 16));
    hipLaunchKernelGGL(k_scale_by_rsqrt, dim3(g), dim3(256), 0, h->stream, nd, sc + 17, V, V);
    std::vector<double> H((size_t)M * M, 0.0);
    const bool lp = lp_format(h, L) != 0;
    int m_done = 0;
    for (int j = 0; j < M; ++j) {
        double* vj = V + (size_t)j * nd;
        double* w = V + (size_t)(j + 1) * nd;
        HIP_TRY(hipMemcpyAsync(xin, vj, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        if (glob) SNS_TRY(exchange_level(h, l, xin));
        // y = -A~ v (the copy the sweeps read) resp. + A v; w = S A v
        if (lp) launch_pc_spmv<SPMV_B_MINUS_AX>(h, L, rows, xin, y, zero, 0.0);
        else launch_spmv<SPMV_AX>(h, L, rows, xin, y, nullptr, 0.0, nullptr);
        launch_first_sweep(h, l, L, rows, y, lp ? -1.0 : 1.0, w);
        for (int pass = 0; pass < 2; ++pass) {
            hipLaunchKernelGGL(k_multi_dot8, dim3(g), dim3(256), 0, h->stream, nd, j + 1, V, nd, w, h->partial);
            SNS_TRY(reduce(8, sc + 8 * pass));
            hipLaunchKernelGGL(k_multi_axpy8, dim3(g), dim3(256), 0, h->stream, nd, j + 1, V, nd, sc + 8 * pass, -1.0, w,
                               (double*)nullptr);
        }
        hipLaunchKernelGGL(k_dot2, dim3(g), dim3(256), 0, h->stream, nd, w, w, h->partial);
        SNS_TRY(reduce(2, sc + 16));
        double v[18];
        SNS_TRY(fetch(h, sc, 18, v));
        for (int k = 0; k <= j; ++k) H[(size_t)k * M + j] = v[k] + v[8 + k];
        m_done = j + 1;
        const double wn = std::sqrt(std::max(0.0, v[16]));
        if (!(wn > 1e-12) || j + 1 == M) break;        // invariant subspace (tiny levels) or done
        H[(size_t)(j + 1) * M + j] = wn;
        hipLaunchKernelGGL(k_scale_by_rsqrt, dim3(g), dim3(256), 0, h->stream, nd, sc + 17, w, w);
    }
    std::vector<double> Hm((size_t)m_done * m_done), re((size_t)m_done), im((size_t)m_done);
    for (int i = 0; i < m_done; ++i)
        for (int j = 0; j < m_done; ++j) Hm[(size_t)i * m_done + j] = H[(size_t)i * M + j];
    if (sns_host_hessenberg_eigs(m_done, Hm.data(), re.data(), im.data()) != SNS_OK) return SNS_OK;
    double tmax = 0.0;
    for (int i = 0; i < m_done; ++i) tmax = std::max(tmax, std::hypot(re[i], im[i]));
    double lim = 1e30;
    for (int i = 0; i < m_done; ++i) {
        const double a2 = re[i] * re[i] + im[i] * im[i];
        if (std::sqrt(a2) < 0.5 * tmax || !(a2 > 0.0)) continue;
        lim = std::min(lim, 2.0 * std::max(re[i], 0.0) / a2);
    }
    *theta_max = tmax;
    *limit = lim;
    return SNS_OK;
}

// Growth factor per sweep of the damped block-Jacobi iteration matrix G_w = I - w Dinv A on the
// dominant mode of Dinv A (left in pong[l] by estimate_lambda_max).  |lambda|max alone does not bound
// the stable damping of a NON-symmetric operator (|1 - w lambda| < 1 needs w < 2 Re(lambda)/|lambda|^2):
// on the 10 M-tet Jacobian w = 0.8 converges and w = 0.85 on the coarse levels breaks BiCGStab down.
int jacobi_growth(sns_ctx* h, int l, double omega, double* growth) {
    Level& L = h->levels[l];
    const int32_t rows = L.n_owned;
    const int64_t nd = 4 * (int64_t)rows;
    const int g = vec_grid(nd);
    double* x0 = h->pong[l];
    double* xa = L.x;
    double* xb = L.r;
    double* zero = nullptr;
    SNS_TRY(get_vec(h, 13, &zero));                      // level sizes never exceed the fine level
    if (nd > 0) HIP_TRY(hipMemsetAsync(zero, 0, nd * sizeof(double), h->stream));
    // keep x0 intact (it seeds later trials): first sweep x0 -> xa, then ping-pong xa <-> xb
    const bool glob = uses_ghosts_in_sweeps(h, l, L);
    if (glob) SNS_TRY(exchange_level(h, l, x0));
    launch_sweep(h, l, L, rows, x0, xa, zero, omega);
    double* cur = xa;
    double* oth = xb;
    const int sweeps = 6;
    for (int s = 1; s < sweeps; ++s) {
        if (s == 2 || s == sweeps - 1) {
            if (rows > 0) hipLaunchKernelGGL(k_dot2, dim3(g), dim3(256), 0, h->stream, nd, cur, cur, h->partial);
            if (glob) SNS_TRY(reduce_to(h, g, 2, h->d_scal + 48 + (s == 2 ? 0 : 2)));
            else reduce_local(h, g, 2, h->d_scal + 48 + (s == 2 ? 0 : 2));
        }
        if (glob) SNS_TRY(exchange_level(h, l, cur));
        launch_sweep(h, l, L, rows, cur, oth, zero, omega);
        std::swap(cur, oth);
    }
    double v[4];
    SNS_TRY(fetch(h, h->d_scal + 48, 4, v));             // ||x_2||^2, ||x_{sweeps-1}||^2
    *growth = (v[0] > 0.0) ? std::pow(v[2] / v[0], 0.5 / (double)(sweeps - 1 - 2)) : 0.0;
    // scratch vectors: only [0, nd) was written; ghost tails stay untouched
    return SNS_OK;
}

int pc_setup(sns_ctx* h) {
    if (!h->has_matrix) { set_error("pc_setup before a matrix was assembled"); return SNS_E_STATE; }
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    const int nl = (h->opt.pc_type == SNS_PC_AMG) ? (int)h->levels.size() : 1;
    bool any_block = false;
    const bool new_operator = !h->r3_estimates && (h->matrix_form != h->est_form || h->opt.reynolds != h->est_re);
    for (int l = 0; l < nl; ++l) {
        Level& L = h->levels[l];
        const int32_t rows = L.n_owned;
        if (h->rep_level > 0 && l == h->rep_level - 1) {
            // level R is only the source of the replicated copy: all-gather my rows' blocks, scatter them into place
            Level& C = h->levels[h->rep_level];
            if (L.nnzb > 0)
                HIP_TRY(hipMemcpyAsync(h->rep_vsend, L.vals, (size_t)L.nnzb * 16 * sizeof(double), hipMemcpyDeviceToDevice,
                                       h->stream));
            SNS_TRY(comm_allgather(h->comm.get(), h->rep_vsend, h->rep_vrecv, (int)(h->rep_maxnz * 16), h->stream));
            const int64_t nsrc = h->rep_maxnz * h->comm->nranks;
            hipLaunchKernelGGL(k_scatter_blocks, dim3((unsigned)((nsrc * 8 + 255) / 256)), dim3(256), 0, h->stream, nsrc,
                               h->rep_valmap, h->rep_vrecv, C.vals);
            continue;
        }
        if (rows > 0)
            hipLaunchKernelGGL(k_dinv, dim3((rows + 255) / 256), dim3(256), 0, h->stream, rows, L.diag, L.vals, L.dinv);
        if (block_active(h, l)) {
            // the aggregates' inverse diagonal blocks, from the fp64 operator (what the nodal D^-1 is to the point smoother)
            // ... in the format of the level's matrix copy (fp32, or fp16 with row scales: half the bytes of a block sweep's extra stream)
            const int bf = h->opt.amg_f32_matrix == 2 ? 2 : 1;
            if (L.binv32 && L.binv_fmt != bf) { (void)hipFree(L.binv32); L.binv32 = nullptr; }
            if (!L.binv32) {
                uint8_t* pb = nullptr;
                SNS_TRY(dev_alloc(&pb, binv_bytes_per_block(bf) * (size_t)std::max(1, L.n_blk)));
                L.binv32 = pb;
                L.binv_fmt = bf;
            }
            if (L.n_blk > 0) {
                if (bf == 2)
                    hipLaunchKernelGGL((k_binv<2>), dim3((unsigned)((L.n_blk + 7) / 8)), dim3(256), 0, h->stream, L.n_blk, L.blk_rows,
                                       L.blk_of, L.rowptr, L.colind, L.vals, L.binv32, h->d_sing);
                else
                    hipLaunchKernelGGL((k_binv<1>), dim3((unsigned)((L.n_blk + 7) / 8)), dim3(256), 0, h->stream, L.n_blk, L.blk_rows,
                                       L.blk_of, L.rowptr, L.colind, L.vals, L.binv32, h->d_sing);
            }
            any_block = true;
        }
        L.omega = h->opt.amg_omega * h->damping_backoff;
        const bool direct = (L.dense_inv || L.dense_gj || h->cg_N > 0) && l + 1 == nl && nl > 1;   // solved, not smoothed
        if (h->opt.pc_type == SNS_PC_AMG && h->opt.amg_f32_matrix && !direct) {
            if (!L.dinv32) SNS_TRY(dev_alloc(&L.dinv32, (size_t)16 * std::max(1, L.n)));
            if (rows > 0)
                hipLaunchKernelGGL(k_cvt_f32, dim3(vec_grid(16 * (int64_t)rows)), dim3(256), 0, h->stream, 16 * (int64_t)rows,
                                   L.dinv, L.dinv32);
            if (h->opt.amg_f32_matrix == 2) {
                if (!L.vals16) {
                    uint2* v16 = nullptr;
                    SNS_TRY(dev_alloc(&v16, (size_t)L.nnzb * 4));
                    L.vals16 = v16;
                    SNS_TRY(dev_alloc(&L.scale16, (size_t)4 * std::max(1, L.n)));
                }
                // ONE pass over the level's fp64 operator writes its fp16 copy and, where the level has one, the fp16 copy of
                // M = A P for the fused post-smoothing sweep (k_lp_copies16)
                const bool with_m = l + 1 < nl && L.ap_rowptr && L.ap_nib && h->opt.amg_fused_post;
                if (with_m && !L.ap_vals16) {
                    uint2* v16 = nullptr;
                    SNS_TRY(dev_alloc(&v16, (size_t)L.ap_nnz * 4));
                    L.ap_vals16 = v16;
                    SNS_TRY(dev_alloc(&L.ap_scale16, (size_t)4 * std::max(1, L.n)));
                }
                if (rows > 0) {
                    const unsigned grid = (unsigned)((rows + 31) / 32);
                    if (with_m)
                        hipLaunchKernelGGL((k_lp_copies16<1>), dim3(grid), dim3(128), 0, h->stream, rows, L.rowptr, L.vals,
                                           (uint2*)L.vals16, L.scale16, L.ap_rowptr, L.ap_colind, L.ap_ptr, L.ap_idx, L.ap_nib, L.agg,
                                           L.free_mask, (uint2*)L.ap_vals16, L.ap_scale16);
                    else
                        hipLaunchKernelGGL((k_lp_copies16<0>), dim3(grid), dim3(128), 0, h->stream, rows, L.rowptr, L.vals,
                                           (uint2*)L.vals16, L.scale16, (const int32_t*)nullptr, (const int32_t*)nullptr,
                                           (const int32_t*)nullptr, (const int32_t*)nullptr, (const uint64_t*)nullptr,
                                           (const int32_t*)nullptr, (const uint8_t*)nullptr, (uint2*)nullptr, (float*)nullptr);
                }
            }
            bool want32 = h->opt.amg_f32_matrix != 2;
#ifdef SNS_HARNESS
            if (std::getenv("SNS_BOTH_LP")) want32 = true;       // the fp16-vs-fp32 A/B needs both copies
#endif
            if (want32) {
                if (!L.vals32) SNS_TRY(dev_alloc(&L.vals32, (size_t)L.nnzb * 16));
                if (L.nnzb > 0)
                    hipLaunchKernelGGL(k_cvt_f32, dim3(vec_grid(L.nnzb * 16)), dim3(256), 0, h->stream, L.nnzb * 16, L.vals,
                                       L.vals32);
            }
        }
        if (h->opt.pc_type == SNS_PC_AMG && !direct) {
            // the spectrum moves little between the Jacobians of one Newton sequence: re-estimate every 4th setup
            double lam = L.lambda_max;
            // (collective when the level's sweeps use exchanged ghost values: every rank takes part, rows or not)
            // ... but not from the Stokes operator to a Jacobian (or to another Reynolds number): round 3 took the first three
            // Jacobians' damping from the Stokes solve's estimate, which is what let level 1 of the jittered 120 x 30 x 30 duct
            // run at w = 0.72 where its own spectrum allows 0.48 (tests/test_gpu_parity.py::test_damping_backoff_...)
            if ((rows > 0 || uses_ghosts_in_sweeps(h, l, L)) && (!(lam > 0.0) || (h->pc_setups & 3) == 0 || new_operator))
                SNS_TRY(estimate_lambda_max(h, l, &lam));
            const bool fresh = !(L.lambda_max > 0.0) || (h->pc_setups & 3) == 0 || new_operator;
            L.lambda_max = lam;
            if (lam > 0.0) L.omega = std::min(h->opt.amg_omega, (4.0 / 3.0) / lam) * h->damping_backoff;
            // levels that run 3 or more sweeps per cycle: the stability limit of the dominant (complex) Ritz values as well
            if (h->opt.amg_ritz_limit && l + 1 < nl) {
                int a = 1, b = 1;
                level_sweeps(h, l, a, b);
                if (a + b >= 3) {
                    if (fresh && (rows > 0 || level_exact(h, l))) {
                        double tmax = 0.0, lim = 1e30;
                        SNS_TRY(arnoldi_ritz(h, l, &tmax, &lim));
                        L.ritz_limit = lim;
                        if (h->opt.monitor)
                            std::printf("    AMG level %d: Ritz |theta|max %.4f, damping limit 2 Re/|theta|^2 = %.4f\n", l, tmax, lim);
                    }
                    if (L.ritz_limit > 0.0) L.omega = std::min(L.omega, L.ritz_limit * h->damping_backoff);
                }
            }
            // the growth check of rounds 1-3 (back off until a sweep contracts the dominant mode by >= 10 %): amg_growth_check
            // 2 = on every level (round 3), 1 = only on levels that run >= 3 sweeps per cycle, 0 = never.  A level with two sweeps per
            // cycle (the fine level, V(1,1)) does not compound an amplified mode, and backing its damping off for the sake of a few
            // complex outliers weakens the smoothing of everything else (prototype: 72 iterations at w0 = 0.46, 84 at 0.24)
            bool check_growth = h->opt.amg_growth_check >= 2;
            if (h->opt.amg_growth_check == 1 && l + 1 < nl) {
                int a = 1, b = 1;
                level_sweeps(h, l, a, b);
                check_growth = a + b >= 3;
            }
            if (fresh && lam > 0.0 && !check_growth) L.omega_checked = 0.0;
            if (fresh && lam > 0.0 && check_growth) {
                // verify the damping on the dominant mode; back off until a sweep contracts it by >= 10 %
                for (int trial = 0; trial < 6; ++trial) {
                    double gr = 0.0;
                    SNS_TRY(jacobi_growth(h, l, L.omega, &gr));
                    if (h->opt.monitor) std::printf("    AMG level %d: omega %.4f growth/sweep on dominant mode %.4f\n", l, L.omega, gr);
                    if (gr < 0.9) break;
                    L.omega *= 0.9;
                }
                L.omega_checked = L.omega;
            } else if (L.omega_checked > 0.0) {
                L.omega = std::min(L.omega, L.omega_checked);
            }
            if (h->opt.monitor) std::printf("    AMG level %d: n %d |lambda|max(Dinv A) %.4f omega %.4f\n", l, rows, lam, L.omega);
        }
        if (l + 1 < nl && L.ap_rowptr && h->opt.amg_fused_post && h->opt.pc_type == SNS_PC_AMG && lp_format(h, L) != 0 &&
            rows > 0) {
            // numeric part of M = A P, straight into the level's low-precision format (no fp64 copy of M)
            const unsigned gq = (unsigned)((4 * (int64_t)rows + 255) / 256);
            if (lp_format(h, L) == 2) {
                // (written together with the fp16 copy of A above)
            } else {
                if (!L.ap_vals32) SNS_TRY(dev_alloc(&L.ap_vals32, (size_t)L.ap_nnz * 16));
                hipLaunchKernelGGL(k_ap_cvt32, dim3(gq), dim3(256), 0, h->stream, rows, L.ap_rowptr, L.ap_colind, L.ap_ptr,
                                   L.ap_idx, L.vals, L.agg, L.free_mask, (float4*)L.ap_vals32);
            }
        }
        if (l + 1 < nl) {
            Level& C = h->levels[l + 1];
            const int64_t nth = C.nnzb * 8;
            hipLaunchKernelGGL(k_galerkin, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, h->stream, C.nnzb,
                               L.r_ptr, L.r_idx, L.vals, h->slot_row[l + 1], C.colind,
                               (l == 0) ? h->empty_c[0] : (const uint8_t*)nullptr, L.m_ptr, C.vals);
        } else if (h->cg_N > 0 && nl > 1) {
            // distributed coarsest level: my rows of the GLOBAL dense matrix -> all-gather -> replicated inverse
            const int N = h->cg_N, mr = 4 * h->cg_maxn;
            HIP_TRY(hipMemsetAsync(h->cg_rows, 0, (size_t)mr * N * sizeof(double), h->stream));
            const int64_t nth = L.nnzb * 16;
            if (nth > 0)
                hipLaunchKernelGGL(k_bsr_to_dense_map, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, h->stream,
                                   L.n_owned, L.rowptr, L.colind, L.vals, h->cg_colmap, N, h->cg_rows);
            // padding rows (ranks with fewer nodes than maxn) get a unit diagonal
            hipLaunchKernelGGL(k_pad_identity, dim3(1), dim3(256), 0, h->stream, 4 * L.n_owned, mr,
                               h->comm->rank * mr, N, h->cg_rows);
            SNS_TRY(comm_allgather(h->comm.get(), h->cg_rows, h->cg_full, mr * N, h->stream));
            hipLaunchKernelGGL(k_dense_inverse, dim3(1), dim3(1024), 0, h->stream, N, h->cg_full, h->d_piv, h->d_sing);
        } else if (L.dense_gj && nl > 1) {
            // blocked Gauss-Jordan inverse on the fp64 matrix cores, then its fp32 copy for the cycle's matvec
            const int N = 4 * L.n, Np = L.dense_np;
            HIP_TRY(hipMemsetAsync(L.dense_gj, 0, (size_t)Np * Np * sizeof(double), h->stream));
            const int64_t nth = L.nnzb * 16;
            hipLaunchKernelGGL(k_bsr_to_dense_ld, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, h->stream, L.nnzb,
                               h->slot_row[l], L.colind, L.vals, Np, L.dense_gj);
            if (Np > N) hipLaunchKernelGGL(k_dense_pad_diag, dim3((Np - N + 255) / 256), dim3(256), 0, h->stream, N, Np, L.dense_gj);
            // (the two-stream schedule of dense_gj_inverse is opt-in: measured, it is SLOWER -- 3.9 against 2.4 ms at N = 1900 --
            // because the bulk update's 900 workgroups fill the chip and the pivot chain's few workgroups queue behind them)
            if (!h->gj_stream && std::getenv("SNS_GJ_TWO_STREAMS") &&
                hipStreamCreateWithFlags(&h->gj_stream, hipStreamNonBlocking) != hipSuccess) h->gj_stream = nullptr;
            dense_gj_inverse(h->stream, h->gj_stream, Np, L.dense_gj, L.dense_work, h->d_sing);
            const int64_t nn = (int64_t)Np * Np;
            hipLaunchKernelGGL(k_dense_to_f32, dim3((unsigned)((nn / 4 + 255) / 256)), dim3(256), 0, h->stream, nn, L.dense_gj,
                               L.dense_x32);
        } else if (L.dense_inv && nl > 1) {
            const int N = 4 * L.n;
            HIP_TRY(hipMemsetAsync(L.dense_inv, 0, (size_t)N * N * sizeof(double), h->stream));
            const int64_t nth = L.nnzb * 16;
            hipLaunchKernelGGL(k_bsr_to_dense, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, h->stream, L.n,
                               L.rowptr, L.colind, L.vals, L.dense_inv);
            hipLaunchKernelGGL(k_dense_inverse, dim3(1), dim3(1024), 0, h->stream, N, L.dense_inv, h->d_piv,
                               h->d_sing);
        }
    }
    ++h->pc_setups;
    h->est_form = h->matrix_form;
    h->est_re = h->opt.reynolds;
    const bool check_sing = nl > 1 && (h->levels[nl - 1].dense_gj != nullptr || any_block);
    int* h_sing = reinterpret_cast<int*>(h->h_scal + 768);
    *h_sing = 0;
    if (check_sing) HIP_TRY(hipMemcpyAsync(h_sing, h->d_sing, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->tm.pc_setup_ms += ms;
    HIP_TRY(hipGetLastError());
    if (check_sing) {
        // The aggregate blocks' inverses are rank-local: one rank alone returning an error here would leave the others in the
        // Krylov loop's collectives (an RCCL hang, a peer time-out).  The verdict is taken over all ranks.
        double bad[1] = {*h_sing != 0 ? 1.0 : 0.0};
        SNS_TRY(global_sum(h, bad, 1));
        if (bad[0] != 0.0) {
            // (the elimination runs without pivoting across its 64 x 64 blocks: see csrc/sns_dense.hip for why that is safe on this
            // operator class; if it ever is not, say so instead of preconditioning with garbage -- there is no fallback hierarchy)
            HIP_TRY(hipMemset(h->d_sing, 0, sizeof(int)));
            set_error("AMG setup: a dense inverse (coarsest level or an aggregate block" +
                      std::string(*h_sing != 0 ? "" : ", on another rank") + ") met a zero or non-finite pivot; set "
                      "amg_dense_rows = 0 / amg_block_smooth = 0");
            return SNS_E_STATE;
        }
    }
    h->pc_ready = true;
    return SNS_OK;
}

// sweeps before / after the coarse-grid correction on level l (the first pre-sweep is omega D^-1 b)
inline void level_sweeps(const sns_ctx* h, int l, int& nu_pre, int& nu_post) {
    const int nu = level_nu(h, l);
    nu_pre = nu_post = nu;
    const int ll = (h->rep_level > 0 && l >= h->rep_level) ? l - 1 : l;
    if (ll == 1) {
        // level 1: asymmetric sweep counts (amg_nu_l1_pre / amg_nu_l1_post).  Automatic (both options 0): a single-GPU handle
        // runs 1 + (nu + 2) sweeps -- post-smoothing is the more valuable half under a piecewise-constant prolongation, 1 + 6
        // needs the iterations of 4 + 4 with one level-1 pass less (-3 % per Newton step on the 10 M-tet duct, neutral
        // elsewhere); a partitioned handle keeps nu + nu, its post-sweeps being rank-local (1 + 6 costs 8-11 % more
        // iterations there)
        // (... unless its sweeps are the exact global ones: level_exact -- then it IS the single-GPU cycle)
        const bool partitioned = h->comm && h->comm->active() && h->comm->nranks > 1 && !level_exact(h, l);
        if (block_active(h, l)) {
            if (!partitioned) { nu_pre = 1; nu_post = std::max(1, h->opt.amg_bnu_l1); }
        } else if (h->opt.amg_nu_l1_pre == 0 && h->opt.amg_nu_l1_post == 0 && !partitioned && nu >= 2) {
            nu_pre = 1;
            nu_post = nu + 2;
        }
        if (h->opt.amg_nu_l1_pre > 0) nu_pre = h->opt.amg_nu_l1_pre;
        if (h->opt.amg_nu_l1_post > 0) nu_post = h->opt.amg_nu_l1_post;
    }
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
    if (l == 0 && h->opt.amg_fine_cycle != 0) return false;
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
    return h->levels.size() >= 2 && h->levels[0].xg && h->opt.amg_fine_cycle == 0 && !(h->rep_level == 1) && level_fused_post(h, 0);
}

int vcycle(sns_ctx* h, int l, const double* b, double* x);
// First level (>= 1) small enough that its kernels are launch-bound rather than bandwidth-bound: it and everything
// below run as one graph.  10 M tets: level 2 (36 k rows; level 1 has 218 k rows = 46 us per sweep); 1 M tets: level 1.
inline int serial_graph_level(const sns_ctx* h) {
    int max_rows = 150000;
#ifdef SNS_HARNESS
    if (std::getenv("SNS_GRAPH_ROWS")) max_rows = std::atoi(std::getenv("SNS_GRAPH_ROWS"));
#endif
    for (int l = 1; l < (int)h->levels.size(); ++l)
        if (h->levels[l].n <= max_rows) return l;
    return 0;
}

// Coarse part of the cycle (the graph level and below) as ONE hipGraph launch.  Captured on a private
// stream (the caller's stream may be the legacy default stream, which cannot be captured), re-captured when
// the per-level damping or the cycle shape changed.  Distributed runs keep direct launches (the exchange
// inside the cycle is a host-driven RCCL group).  Any capture failure disables the graph for good.
int coarse_cycle(sns_ctx* h, int l, const double* b, double* x) {
    const bool dist = h->comm && h->comm->active() && h->comm->nranks > 1;
    // distributed runs: only the replicated tail is free of exchanges and can be captured
    const int gl = dist ? h->rep_level : serial_graph_level(h);
    if (gl <= 0 || l != gl || h->graph_disabled || (int)h->levels.size() <= gl + 1) return vcycle(h, l, b, x);
    std::vector<double> sig;
    for (auto& L : h->levels) sig.push_back(L.omega);
    sig.push_back(h->opt.amg_nu); sig.push_back(h->opt.amg_nu_coarse); sig.push_back(h->opt.amg_nu_deep);
    sig.push_back(h->opt.amg_nu_l2);
    sig.push_back(h->opt.amg_nu_l1_pre); sig.push_back(h->opt.amg_nu_l1_post);
    sig.push_back(h->opt.amg_f32_matrix);
    sig.push_back(h->opt.amg_fused_post);
    sig.push_back(h->opt.amg_nu_scale_with_size);
    sig.push_back(h->opt.amg_fine_cycle);
    sig.push_back(h->opt.amg_block_smooth); sig.push_back(h->opt.amg_bnu_l1); sig.push_back(h->opt.amg_bnu_l2);
    sig.push_back(h->opt.amg_bnu_deep); sig.push_back(h->opt.amg_block_max_rows); sig.push_back(h->opt.amg_block_fine_rows);
    sig.push_back(h->opt.amg_fuse_restrict);
    sig.push_back(restrict_fuses_first(h, gl - 1) ? 1.0 : 0.0);
    sig.push_back(gl);
    if (!h->coarse_graph || sig != h->graph_sig) {
        if (h->coarse_graph) { (void)hipGraphExecDestroy(h->coarse_graph); h->coarse_graph = nullptr; }
        if (!h->cap_stream && hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) != hipSuccess) {
            h->graph_disabled = true;
            return vcycle(h, l, b, x);
        }
        HIP_TRY(hipStreamSynchronize(h->stream));          // capture must not race with pending work on the buffers
        hipStream_t user = h->stream;
        hipGraph_t graph = nullptr;
        bool ok = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            h->stream = h->cap_stream;
            const int rc = vcycle(h, l, b, x);
            h->stream = user;
            ok = (hipStreamEndCapture(h->cap_stream, &graph) == hipSuccess) && rc == SNS_OK && graph;
        }
        if (ok) ok = hipGraphInstantiate(&h->coarse_graph, graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) (void)hipGraphDestroy(graph);
        if (!ok) {
            (void)hipGetLastError();
            h->coarse_graph = nullptr;
            h->graph_disabled = true;
            return vcycle(h, l, b, x);
        }
        h->graph_sig = sig;
    }
    HIP_TRY(hipGraphLaunch(h->coarse_graph, h->stream));
    return SNS_OK;
}

// one aggregate-block sweep of a partitioned level with the ghost entries of x from the level's receive window
void launch_sweep_windows(sns_ctx* h, const Level& L, const double* x, double* y, const double* b, double omega, const GhostSrc& gs) {
    const int32_t ns = 8 * L.n_blk;
    const unsigned grid = (unsigned)((ns + 63) / 64);
    if (grid == 0) return;
    if (L.binv_fmt == 2)
        hipLaunchKernelGGL((k_bsweep<2, 1>), dim3(grid), dim3(256), 0, h->stream, ns, L.blk_rows, L.rowptr, L.colind,
                           (const void*)L.vals16, L.scale16, (const void*)L.binv32, x, y, b, omega, gs);
    else
        hipLaunchKernelGGL((k_bsweep<1, 1>), dim3(grid), dim3(256), 0, h->stream, ns, L.blk_rows, L.rowptr, L.colind,
                           (const void*)L.vals32, (const float*)nullptr, (const void*)L.binv32, x, y, b, omega, gs);
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

// The V-cycle of a PARTITIONED level over a window transport (peer windows / the in-process team; round 5).  Every exchange is one
// put launch (comm_put) and the pass behind it reads the ghost entries from the level's receive window, its boundary waves waiting
// for the neighbours themselves: no staging copy, no unpack, no split pass.  Level 0: first sweep | put, residual | restriction
// (+ level 1's first sweep) | coarse | put of level 1's solution, fused correction + post-sweep.  Level >= 1 (level_exact): the
// single-GPU schedule with exact global sweeps -- [put, sweep]* | put, residual + restriction (+ next first sweep) in one launch |
// coarse | fused correction + first post-sweep (the coarse solution read straight from the replicated tail where that is the next
// level, else after a put of it) | [put, sweep]*.
int vcycle_windows(sns_ctx* h, int l, const double* b, double* x) {
    Level& L = h->levels[l];
    Level& C = h->levels[l + 1];
    Comm* c = h->comm.get();
    const Plan& P = c->plans[l];
    const int32_t rows = L.n_owned;
    const double om = L.omega;
    int nu_pre = 1, nu_post = 1;
    level_sweeps(h, l, nu_pre, nu_post);
    double* cur = cycle_start_buffer(h, l, x);
    double* oth = (cur == x) ? h->pong[l] : x;
    if (rows > 0 && !(l > 0 && restrict_fuses_first(h, l - 1)) && !(l == 0 && h->first_sweep_done))
        launch_first_sweep(h, l, L, rows, b, om, cur);
    if (l == 0) h->first_sweep_done = false;
    for (int s = 1; s < nu_pre; ++s) {                     // (level >= 1 only: the fine level runs one sweep per half cycle)
        ++h->ctr_exchange;
        SNS_TRY(comm_put(c, P, cur, h->stream));
        if (rows > 0) launch_sweep_windows(h, L, cur, oth, b, om, comm_ghost_src(c, P));
        std::swap(cur, oth);
    }
    const bool rep_src = h->rep_level > 0 && l + 1 == h->rep_level - 1;
    double* cb = rep_src ? h->rep_bsend : C.b;
    const double* cx = rep_src ? h->levels[h->rep_level].x + 4 * (size_t)h->rep_off : C.x;
    const bool fuse = restrict_fuses_first(h, l);
    const float* dc = fuse ? C.dinv32 : nullptr;
    double* zc = fuse ? cycle_start_buffer(h, l + 1, C.x) : nullptr;
    const int fmt = lp_format(h, L);
    // residual (+ restriction): the true residual needs the neighbours' iterate
    ++h->ctr_exchange;
    SNS_TRY(comm_put(c, P, cur, h->stream));
    const GhostSrc gs = comm_ghost_src(c, P);
    if (l == 0) {
        Split s3;
        s3.mode = 3;
        s3.gs = gs;
        if (rows > 0) launch_pc_spmv<SPMV_B_MINUS_AX>(h, L, rows, cur, L.r, b, 0.0, s3);
        if (C.n_owned > 0) {
            if (fuse && block_active(h, 1)) {
                const int32_t ns = 8 * C.n_blk;
                if (C.binv_fmt == 2)
                    hipLaunchKernelGGL((k_restrict_blk<2>), dim3((unsigned)((ns + 63) / 64)), dim3(256), 0, h->stream, ns, C.blk_rows,
                                       L.m_ptr, L.m_idx, L.free_mask, L.r, cb, (const void*)C.binv32, C.omega, zc);
                else
                    hipLaunchKernelGGL((k_restrict_blk<1>), dim3((unsigned)((ns + 63) / 64)), dim3(256), 0, h->stream, ns, C.blk_rows,
                                       L.m_ptr, L.m_idx, L.free_mask, L.r, cb, (const void*)C.binv32, C.omega, zc);
            } else {
                hipLaunchKernelGGL(k_restrict, dim3((unsigned)((4 * (int64_t)C.n_owned + 255) / 256)), dim3(256), 0, h->stream,
                                   C.n_owned, L.m_ptr, L.m_idx, L.free_mask, L.r, cb, dc, C.omega, zc);
            }
        }
    } else if (rows > 0 && C.n_owned > 0) {
        const int mode = !fuse ? 0 : (block_active(h, l + 1) ? 2 : 1);
        const int32_t* slots = mode == 2 ? C.blk_rows : nullptr;
        const int32_t n_slots = mode == 2 ? 8 * C.n_blk : C.n_owned;
        const unsigned grid = (unsigned)((n_slots + 7) / 8);
        const void* vals = fmt == 2 ? (const void*)L.vals16 : (const void*)L.vals32;
        const float* sc16 = fmt == 2 ? L.scale16 : nullptr;
#define SNS_RRW(F, M)                                                                                                              \
    hipLaunchKernelGGL((k_resid_restrict<F, M, 1>), dim3(grid), dim3(256), 0, h->stream, C.n_owned, n_slots, slots, L.m_ptr, L.m_idx, \
                       L.free_mask, L.rowptr, L.colind, vals, sc16, (const double*)cur, b, L.r, cb, dc, (const void*)C.binv32, C.omega, zc, gs)
        if (fmt == 2) { if (mode == 2) SNS_RRW(2, 2); else if (mode == 1) SNS_RRW(2, 1); else SNS_RRW(2, 0); }
        else          { if (mode == 2) SNS_RRW(1, 2); else if (mode == 1) SNS_RRW(1, 1); else SNS_RRW(1, 0); }
#undef SNS_RRW
    }
    SNS_TRY(coarse_cycle(h, l + 1, cb, rep_src ? nullptr : C.x));
    // fused coarse-grid correction + first post-smoothing sweep over M = A P; the ghost aggregates' part of the coarse solution:
    // level >= 1 above the replicated tail reads every entry from the replicated solution (M's columns renumbered into its ids,
    // ap_colind_rep), else one put of the coarse level's solution and the window behind it
    GhostSrc gc;
    const double* xc = cx;
    const int32_t* apc = L.ap_colind;
    if (rep_src && l >= 1) {
        xc = h->levels[h->rep_level].x;
        apc = L.ap_colind_rep;
    } else {
        ++h->ctr_exchange;
        SNS_TRY(comm_put(c, c->plans[l + 1], cx, h->stream));
        gc = comm_ghost_src(c, c->plans[l + 1]);
    }
    if (rows > 0) {
        if (l == 0) time_begin(h, 4);
        if (block_active(h, l) && L.binv32) {
            const int32_t ns = 8 * L.n_blk;
            const unsigned gb = (unsigned)((ns + 63) / 64);
            const void* mv = fmt == 2 ? (const void*)L.ap_vals16 : (const void*)L.ap_vals32;
            const float* ms = fmt == 2 ? L.ap_scale16 : nullptr;
#define SNS_BPW(F, G)                                                                                                          \
    hipLaunchKernelGGL((k_bpost<F, G>), dim3(gb), dim3(256), 0, h->stream, ns, L.blk_rows, L.ap_rowptr, apc, mv, ms,             \
                       (const void*)L.binv32, xc, cx, (const double*)cur, (const double*)L.r, om, L.agg, L.free_mask, oth, gc)
            if (gc.win[0]) { if (fmt == 2) SNS_BPW(2, 1); else SNS_BPW(1, 1); }
            else           { if (fmt == 2) SNS_BPW(2, 0); else SNS_BPW(1, 0); }
#undef SNS_BPW
        } else {
            const int grid = (rows + 63) / 64;             // (nodal blocks: the fine level only, see level_exact)
            if (fmt == 2)
                hipLaunchKernelGGL((k_post_lp<2, 2>), dim3(grid), dim3(256), 0, h->stream, rows, L.ap_rowptr, L.ap_colind, L.ap_vals16,
                                   L.ap_scale16, xc, (const double*)cur, (const double*)L.r, L.dinv32, om, L.agg, L.free_mask, oth, gc);
            else
                hipLaunchKernelGGL((k_post_lp<1, 2>), dim3(grid), dim3(256), 0, h->stream, rows, L.ap_rowptr, L.ap_colind,
                                   (const void*)L.ap_vals32, (const float*)nullptr, xc, (const double*)cur, (const double*)L.r, L.dinv32,
                                   om, L.agg, L.free_mask, oth, gc);
        }
        if (l == 0) time_end(h);
    }
    std::swap(cur, oth);
    for (int s = 1; s < nu_post; ++s) {
        ++h->ctr_exchange;
        SNS_TRY(comm_put(c, P, cur, h->stream));
        if (rows > 0) launch_sweep_windows(h, L, cur, oth, b, om, comm_ghost_src(c, P));
        std::swap(cur, oth);
    }
    // cur == x by construction of the start buffer
    return SNS_OK;
}

// V-cycle on level l: x <- approx A_l^-1 b  (x overwritten; zero initial guess)
int vcycle(sns_ctx* h, int l, const double* b, double* x) {
    Level& L = h->levels[l];
    const int32_t rows = L.n_owned;
    const bool last = (l + 1 == (int)h->levels.size());
    const double om = L.omega;
    const int g4 = (int)((4 * (int64_t)rows + 255) / 256);
    if (h->rep_level > 0 && l == h->rep_level - 1) {
        // all-gather the right-hand side, cycle the replicated tail, keep my rows of the result
        Level& C = h->levels[h->rep_level];
        if (rows > 0 && b != h->rep_bsend)                   // (vcycle of the level above restricts straight into rep_bsend)
            HIP_TRY(hipMemcpyAsync(h->rep_bsend, b, 4 * (size_t)rows * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        Comm* cm = h->comm.get();
        if (cm->windows() && h->opt.halo_windows && h->rep_doff &&
            (size_t)4 * h->rep_maxn * (size_t)cm->nranks <= cm->peer->ag_doubles) {
            // (every rank's rows land where the replicated level keeps them: no gather kernel behind the all-gather)
            SNS_TRY(comm_allgatherv(cm, h->rep_bsend, C.b, 4 * h->rep_maxn, h->rep_doff, h->rep_dcnt, h->stream));
        } else {
            SNS_TRY(comm_allgather(cm, h->rep_bsend, h->rep_brecv, 4 * h->rep_maxn, h->stream));
            hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((4 * (int64_t)h->rep_NG + 255) / 256)), dim3(256), 0, h->stream,
                               h->rep_NG, h->rep_rowmap, h->rep_brecv, C.b);
        }
        SNS_TRY(coarse_cycle(h, h->rep_level, C.b, C.x));
        if (rows > 0 && x)                                   // (x == nullptr: the caller reads its rows of C.x in place)
            HIP_TRY(hipMemcpyAsync(x, C.x + 4 * (size_t)h->rep_off, 4 * (size_t)rows * sizeof(double),
                                   hipMemcpyDeviceToDevice, h->stream));
        return SNS_OK;
    }
    if (last) {
        if (h->cg_N > 0) {
            const int N = h->cg_N, mr = 4 * h->cg_maxn;
            HIP_TRY(hipMemsetAsync(h->cg_send, 0, mr * sizeof(double), h->stream));
            if (rows > 0)
                HIP_TRY(hipMemcpyAsync(h->cg_send, b, 4 * (size_t)rows * sizeof(double), hipMemcpyDeviceToDevice,
                                       h->stream));
            SNS_TRY(comm_allgather(h->comm.get(), h->cg_send, h->cg_recv, mr, h->stream));
            if (rows > 0)
                hipLaunchKernelGGL(k_dense_matvec, dim3((4 * rows + 3) / 4), dim3(256), 0, h->stream, N,
                                   h->cg_full + (size_t)h->comm->rank * mr * N, h->cg_recv, x, 4 * rows);
            return SNS_OK;
        }
        if (L.dense_inv) {
            const int N = 4 * L.n;
            hipLaunchKernelGGL(k_dense_matvec, dim3((N + 3) / 4), dim3(256), 0, h->stream, N, L.dense_inv, b, x, N);
            return SNS_OK;
        }
        if (L.dense_x32) {
            const int N = 4 * L.n;
            hipLaunchKernelGGL(k_dense_matvec32, dim3((N + 3) / 4), dim3(256), 0, h->stream, N, L.dense_np, L.dense_x32, b, x);
            return SNS_OK;
        }
        // coarsest level too large for the dense solve: a fixed number of Jacobi sweeps (still a linear operator)
        double* cur = x;
        double* oth = h->pong[l];
        if (rows == 0) return SNS_OK;
        hipLaunchKernelGGL(k_bjacobi, dim3(g4), dim3(256), 0, h->stream, rows, L.dinv, b, om, cur);
        for (int s = 0; s < 8; ++s) {       // even count: result ends in x
            launch_pc_spmv<SPMV_JACOBI>(h, L, rows, cur, oth, b, om);
            std::swap(cur, oth);
        }
        return SNS_OK;
    }
    if (level_windows(h, l)) return vcycle_windows(h, l, b, x);
    if (l == 0 && !L.xg && h->opt.amg_fine_cycle != 0 && rows > 0) {
        // experimental fine-level cycle shapes (single GPU): 1 = V(0,1): no pre-smoothing, the right-hand side itself
        // is restricted; 2 = V(1,0): no post-smoothing.  One fine-level matrix pass per cycle instead of two.
        Level& C = h->levels[1];
        if (h->opt.amg_fine_cycle == 1) {
            hipLaunchKernelGGL(k_restrict, dim3((unsigned)((4 * (int64_t)C.n_owned + 255) / 256)), dim3(256), 0, h->stream,
                               C.n_owned, L.m_ptr, L.m_idx, L.free_mask, b, C.b, (const float*)nullptr, 0.0, (double*)nullptr);
            SNS_TRY(coarse_cycle(h, 1, C.b, C.x));
            double* tmp = h->pong[0];
            HIP_TRY(hipMemsetAsync(tmp, 0, 4 * (size_t)rows * sizeof(double), h->stream));
            hipLaunchKernelGGL(k_prolong_add, dim3(g4), dim3(256), 0, h->stream, rows, L.agg, L.free_mask, C.x, tmp);
            launch_pc_spmv<SPMV_JACOBI>(h, L, rows, tmp, x, b, om);
        } else {
            hipLaunchKernelGGL(k_bjacobi, dim3(g4), dim3(256), 0, h->stream, rows, L.dinv, b, om, x);
            launch_pc_spmv<SPMV_B_MINUS_AX>(h, L, rows, x, L.r, b, 0.0);
            hipLaunchKernelGGL(k_restrict, dim3((unsigned)((4 * (int64_t)C.n_owned + 255) / 256)), dim3(256), 0, h->stream,
                               C.n_owned, L.m_ptr, L.m_idx, L.free_mask, L.r, C.b, (const float*)nullptr, 0.0, (double*)nullptr);
            SNS_TRY(coarse_cycle(h, 1, C.b, C.x));
            hipLaunchKernelGGL(k_prolong_add, dim3(g4), dim3(256), 0, h->stream, rows, L.agg, L.free_mask, C.x, x);
        }
        return SNS_OK;
    }
    const int nu = level_nu(h, l);
    int nu_pre = nu, nu_post = nu;
    level_sweeps(h, l, nu_pre, nu_post);
    double* cur = cycle_start_buffer(h, l, x);
    double* oth = (cur == x) ? h->pong[l] : x;
    // distributed: on levels with few rows per rank the sweeps see the neighbours' current iterate (one small
    // exchange per sweep); on the big levels they stay rank-local (ghost values zero) and only the residual is exact
    const bool sx = level_sx(h, L);
    // ... and the sweeps AFTER the coarse-grid correction take the neighbours' corrected iterate as (frozen) ghost
    // values: with zero ghosts they would see the whole correction as a residual along the partition interfaces
    const bool px = level_px(h, l, L);
    const size_t ghost4 = 4 * (size_t)(L.n - rows);
    const bool tails_unused = (l == 0) && fine_tails_unused(h);
    if (px && ghost4 > 0 && !tails_unused) {
        HIP_TRY(hipMemsetAsync(cur + 4 * (size_t)rows, 0, ghost4 * sizeof(double), h->stream));
        HIP_TRY(hipMemsetAsync(oth + 4 * (size_t)rows, 0, ghost4 * sizeof(double), h->stream));
    }
    // first sweep from a zero guess: z = omega D^-1 b, with the D^-1 copy the other sweeps of this level read (already done
    // by the restriction kernel of the level above where restrict_fuses_first says so)
    if (rows > 0 && !(l > 0 && restrict_fuses_first(h, l - 1)) && !(l == 0 && h->first_sweep_done))
        launch_first_sweep(h, l, L, rows, b, om, cur);
    if (l == 0) h->first_sweep_done = false;
    for (int s = 1; s < nu_pre; ++s) {
        if (sx) SNS_TRY(exchange_level(h, l, cur));
        launch_sweep(h, l, L, rows, cur, oth, b, om);
        std::swap(cur, oth);
    }
    Level& C = h->levels[l + 1];
    // Below the fine level the residual and the restriction (+ the next level's first sweep) are ONE launch (k_resid_restrict):
    // `xres` is then the vector the residual reads and the pass itself is issued with the restriction further down.
    const int fmt_rr = lp_format(h, L);
    // (amg_fuse_restrict = 2: a single-GPU fine level as well -- its residual kernel is the tuned k_spmv_lp, kept by default)
    const bool rr_level = l >= 1 || (h->opt.amg_fuse_restrict >= 2 && !L.xg && h->opt.amg_fine_cycle == 0);
    const bool rr_fused = rr_level && h->opt.amg_fuse_restrict != 0 && fmt_rr != 0 && rows > 0 && C.n_owned > 0 && L.m_ptr &&
                          (!block_active(h, l + 1) || !restrict_fuses_first(h, l) || C.binv_fmt == fmt_rr);
    const double* xres = cur;
    if (sx) {
        SNS_TRY(exchange_level(h, l, cur));
        if (!rr_fused) launch_pc_spmv<SPMV_B_MINUS_AX>(h, L, rows, cur, L.r, b, 0.0);
    } else if (L.xg && tails_unused) {
        // (fine level, fused post-sweep: the halo lands in the iterate's own ghost tail, no copy into the exchange vector)
        SNS_TRY(exchange_and_spmv<SPMV_B_MINUS_AX>(h, cur, cur, L.r, b, 0.0, nullptr, true));
    } else if (L.xg) {      // true residual needs the neighbours' iterate
        HIP_TRY(hipMemcpyAsync(L.xg, cur, 4 * (size_t)rows * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        if (l == 0) {
            SNS_TRY(exchange_and_spmv<SPMV_B_MINUS_AX>(h, L.xg, L.xg, L.r, b, 0.0, nullptr, true));
        } else {
            SNS_TRY(exchange_level(h, l, L.xg));
            xres = L.xg;
            if (!rr_fused) launch_pc_spmv<SPMV_B_MINUS_AX>(h, L, rows, L.xg, L.r, b, 0.0);
        }
    } else if (!rr_fused) {
        launch_pc_spmv<SPMV_B_MINUS_AX>(h, L, rows, cur, L.r, b, 0.0);
    }
    // the level below is only the source of the replicated tail: its right-hand side is restricted straight into the all-gather's
    // send buffer, and the correction is prolongated straight from this rank's rows of the replicated solution (no copies)
    const bool rep_src = h->rep_level > 0 && l + 1 == h->rep_level - 1;
    double* cb = rep_src ? h->rep_bsend : C.b;
    const double* cx = rep_src ? h->levels[h->rep_level].x + 4 * (size_t)h->rep_off : C.x;
    if (C.n_owned > 0) {
        // the restriction also does the next level's first sweep (z = w Dc^-1 bc into the buffer that level starts from)
        const float* dc = nullptr;
        double* zc = nullptr;
        const bool fuse = restrict_fuses_first(h, l);
        if (fuse) {
            dc = C.dinv32;
            zc = cycle_start_buffer(h, l + 1, C.x);          // (coarse_cycle below is called with x = C.x)
        }
        if (rr_fused) {
            // mode of the coarse level's first sweep: 0 none, 1 nodal D^-1, 2 its aggregate blocks (walked in THEIR order)
            const int mode = !fuse ? 0 : (block_active(h, l + 1) ? 2 : 1);
            const int32_t* slots = mode == 2 ? C.blk_rows : nullptr;
            const int32_t n_slots = mode == 2 ? 8 * C.n_blk : C.n_owned;
            const unsigned grid = (unsigned)((n_slots + 7) / 8);
            const void* vals = fmt_rr == 2 ? (const void*)L.vals16 : (const void*)L.vals32;
            const float* sc16 = fmt_rr == 2 ? L.scale16 : nullptr;
#define SNS_RR(F, M)                                                                                                            \
    hipLaunchKernelGGL((k_resid_restrict<F, M, 0>), dim3(grid), dim3(256), 0, h->stream, C.n_owned, n_slots, slots, L.m_ptr, L.m_idx, \
                       L.free_mask, L.rowptr, L.colind, vals, sc16, xres, b, L.r, cb, dc, (const void*)C.binv32, C.omega, zc, GhostSrc())
            if (l == 0) time_begin(h, SPMV_B_MINUS_AX);                      // (bench.py's per-launch accounting of the fine-level passes)
            if (fmt_rr == 2) { if (mode == 2) SNS_RR(2, 2); else if (mode == 1) SNS_RR(2, 1); else SNS_RR(2, 0); }
            else             { if (mode == 2) SNS_RR(1, 2); else if (mode == 1) SNS_RR(1, 1); else SNS_RR(1, 0); }
            if (l == 0) time_end(h);
#undef SNS_RR
        } else if (fuse && block_active(h, l + 1)) {
            const int32_t ns = 8 * C.n_blk;
            if (C.binv_fmt == 2)
                hipLaunchKernelGGL((k_restrict_blk<2>), dim3((unsigned)((ns + 63) / 64)), dim3(256), 0, h->stream, ns, C.blk_rows,
                                   L.m_ptr, L.m_idx, L.free_mask, L.r, cb, (const void*)C.binv32, C.omega, zc);
            else
                hipLaunchKernelGGL((k_restrict_blk<1>), dim3((unsigned)((ns + 63) / 64)), dim3(256), 0, h->stream, ns, C.blk_rows,
                                   L.m_ptr, L.m_idx, L.free_mask, L.r, cb, (const void*)C.binv32, C.omega, zc);
        } else {
            hipLaunchKernelGGL(k_restrict, dim3((unsigned)((4 * (int64_t)C.n_owned + 255) / 256)), dim3(256), 0, h->stream,
                               C.n_owned, L.m_ptr, L.m_idx, L.free_mask, L.r, cb, dc, C.omega, zc);
        }
    }
    SNS_TRY(coarse_cycle(h, l + 1, cb, rep_src ? nullptr : C.x));
    int s_first = 0;
    // Fused coarse-grid correction + first post-smoothing sweep (k_post_lp): z = (cur + P xc) + om Dinv (r - M xc) with
    // M = A P and r the residual restricted above -- the sweep reads M (0.37x the blocks of A on the fine level) instead
    // of A and the prolongation kernel disappears.  Serial levels always; a distributed fine level when its single
    // post-sweep is the exact global one (px): the ghost aggregates' corrections arrive by ONE level-(l+1) exchange
    // instead of the level-l halo of the corrected iterate.
    const int fmt_l = lp_format(h, L);
    const bool fused_post = level_fused_post(h, l);
    if (fused_post) {
        const double* xc = cx;
        if (L.xg) {                                    // distributed fine level: xc incl. the neighbours' aggregates
            if (C.n_owned > 0)
                HIP_TRY(hipMemcpyAsync(C.xg, cx, 4 * (size_t)C.n_owned * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            SNS_TRY(exchange_level(h, l + 1, C.xg));
            xc = C.xg;
        }
        if (rows > 0) {
            const int grid = (rows + 63) / 64;
            const bool fine = (l == 0);
            if (fine) time_begin(h, 4);
            if (block_active(h, l) && L.binv32) {
                const int32_t ns = 8 * L.n_blk;
                const unsigned gb = (unsigned)((ns + 63) / 64);
                if (fmt_l == 2)
                    hipLaunchKernelGGL((k_bpost<2, 0>), dim3(gb), dim3(256), 0, h->stream, ns, L.blk_rows, L.ap_rowptr, L.ap_colind,
                                       (const void*)L.ap_vals16, L.ap_scale16, (const void*)L.binv32, xc, xc, (const double*)cur,
                                       (const double*)L.r, om, L.agg, L.free_mask, oth, GhostSrc());
                else
                    hipLaunchKernelGGL((k_bpost<1, 0>), dim3(gb), dim3(256), 0, h->stream, ns, L.blk_rows, L.ap_rowptr, L.ap_colind,
                                       (const void*)L.ap_vals32, (const float*)nullptr, (const void*)L.binv32, xc, xc,
                                       (const double*)cur, (const double*)L.r, om, L.agg, L.free_mask, oth, GhostSrc());
            } else if (fmt_l == 2) {
                if (fine)
                    hipLaunchKernelGGL((k_post_lp<2, 1>), dim3(grid), dim3(256), 0, h->stream, rows, L.ap_rowptr, L.ap_colind,
                                       L.ap_vals16, L.ap_scale16, xc, cur, L.r, L.dinv32, om, L.agg, L.free_mask, oth, GhostSrc());
                else
                    hipLaunchKernelGGL((k_post_lp<2, 0>), dim3(grid), dim3(256), 0, h->stream, rows, L.ap_rowptr, L.ap_colind,
                                       L.ap_vals16, L.ap_scale16, xc, cur, L.r, L.dinv32, om, L.agg, L.free_mask, oth, GhostSrc());
            } else {
                if (fine)
                    hipLaunchKernelGGL((k_post_lp<1, 1>), dim3(grid), dim3(256), 0, h->stream, rows, L.ap_rowptr, L.ap_colind,
                                       (const void*)L.ap_vals32, (const float*)nullptr, xc, cur, L.r, L.dinv32, om, L.agg,
                                       L.free_mask, oth, GhostSrc());
                else
                    hipLaunchKernelGGL((k_post_lp<1, 0>), dim3(grid), dim3(256), 0, h->stream, rows, L.ap_rowptr, L.ap_colind,
                                       (const void*)L.ap_vals32, (const float*)nullptr, xc, cur, L.r, L.dinv32, om, L.agg,
                                       L.free_mask, oth, GhostSrc());
            }
            if (fine) time_end(h);
        }
        std::swap(cur, oth);
        s_first = 1;
    } else if (rows > 0) {
        hipLaunchKernelGGL(k_prolong_add, dim3(g4), dim3(256), 0, h->stream, rows, L.agg, L.free_mask, cx, cur);
    }
    if (fused_post) {
        // (the post-sweep is done; a partitioned fine level got its neighbours' corrections through xc)
    } else if (px && l == 0 && nu == 1) {
        // the single post-smoothing sweep of the fine level with the neighbours' corrected iterate: halo of `cur`
        // overlapped with the interior rows of the sweep
        SNS_TRY(exchange_and_spmv<SPMV_JACOBI>(h, cur, cur, oth, b, om, nullptr, true));
        std::swap(cur, oth);
        s_first = 1;
    } else if (px) {
        SNS_TRY(exchange_level(h, l, cur));
        if (ghost4 > 0 && nu > 1)
            HIP_TRY(hipMemcpyAsync(oth + 4 * (size_t)rows, cur + 4 * (size_t)rows, ghost4 * sizeof(double),
                                   hipMemcpyDeviceToDevice, h->stream));
    }
    for (int s = s_first; s < nu_post; ++s) {
        if (sx) SNS_TRY(exchange_level(h, l, cur));
        launch_sweep(h, l, L, rows, cur, oth, b, om);
        std::swap(cur, oth);
    }
    // cur == x by construction of the start buffer
    return SNS_OK;
}

int pc_apply_inner(sns_ctx* h, const double* r, double* z);
// (first_sweep_done is consumed by the cycle this call runs and by nothing else: cleared on every way out)
int pc_apply(sns_ctx* h, const double* r, double* z) {
    const int rc = pc_apply_inner(h, r, z);
    h->first_sweep_done = false;
    return rc;
}
int pc_apply_inner(sns_ctx* h, const double* r, double* z) {
    const int64_t nd = nred_of(h);
    switch (h->opt.pc_type) {
        case SNS_PC_NONE:
            HIP_TRY(hipMemcpyAsync(z, r, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            return SNS_OK;
        case SNS_PC_BJACOBI:
            hipLaunchKernelGGL(k_bjacobi, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, h->stream, h->n_owned,
                               h->levels[0].dinv, r, 1.0, z);
            return SNS_OK;
        case SNS_PC_AMG:
            if (h->n > h->n_owned) {
                // distributed: the per-rank V-cycle must see ZERO ghost values on level 0 (block-Jacobi across
                // ranks, like PETSc's parallel default bjacobi).  z's ghost tail may hold halo data, so cycle
                // in internal buffers whose tails are never written and copy the owned part out.
                // (fine_tails_unused: nothing in the fine level's cycle reads a ghost tail as zero -- no internal buffer, no copy)
                if (fine_tails_unused(h)) return vcycle(h, 0, r, z);
                SNS_TRY(vcycle(h, 0, r, h->levels[0].x));
                HIP_TRY(hipMemcpyAsync(z, h->levels[0].x, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
                return SNS_OK;
            }
            return vcycle(h, 0, r, z);
    }
    set_error("bad pc_type");
    return SNS_E_ARG;
}

// operator apply with halo exchange (x must have room for the ghost tail)
int op_apply(sns_ctx* h, double* x, double* y) {
    SNS_TRY(exchange_and_spmv<SPMV_AX>(h, x, x, y, nullptr, 0.0, nullptr, false));
    h->tm.spmv_calls++;
    return SNS_OK;
}
// y = A x with the per-workgroup partial sums of <dotw, y> left in h->partial (BiCGStab's <rhat, A M p>)
int op_apply_dot(sns_ctx* h, double* x, double* y, const double* dotw) {
    SNS_TRY(exchange_and_spmv<SPMV_AX_DOT>(h, x, x, y, nullptr, 0.0, dotw, false));
    h->tm.spmv_calls++;
    return SNS_OK;
}
int op_residual(sns_ctx* h, double* x, const double* b, double* r) {
    SNS_TRY(exchange_and_spmv<SPMV_B_MINUS_AX>(h, x, x, r, b, 0.0, nullptr, false));
    h->tm.spmv_calls++;
    return SNS_OK;
}

int get_vec(sns_ctx* h, size_t k, double** out) {
    while (h->kv.size() <= k) {
        double* p = nullptr;
        SNS_TRY(dev_alloc(&p, (size_t)ld_of(h)));
        HIP_TRY(hipMemset(p, 0, (size_t)ld_of(h) * sizeof(double)));
        h->kv.push_back(p);
    }
    *out = h->kv[k];
    return SNS_OK;
}

int norm2(sns_ctx* h, const double* x, double* out) {
    const int64_t nd = nred_of(h);
    const int g = vec_grid(nd);
    hipLaunchKernelGGL(k_dot2, dim3(g), dim3(256), 0, h->stream, nd, x, x, h->partial);
    SNS_TRY(reduce_to(h, g, 2, h->d_scal));
    double v[2];
    SNS_TRY(fetch(h, h->d_scal, 2, v));
    *out = std::sqrt(v[0]);
    return SNS_OK;
}
int dot(sns_ctx* h, const double* x, const double* y, double* out) {
    const int64_t nd = nred_of(h);
    const int g = vec_grid(nd);
    hipLaunchKernelGGL(k_dot2, dim3(g), dim3(256), 0, h->stream, nd, x, y, h->partial);
    SNS_TRY(reduce_to(h, g, 2, h->d_scal));
    double v[2];
    SNS_TRY(fetch(h, h->d_scal, 2, v));
    *out = v[0];
    return SNS_OK;
}

// Can the Krylov kernel that writes the preconditioner's input also do the V-cycle's first fine-level sweep z = w D^-1 (input)
// (k_bicg_s_first / k_bicg_xrp_first: one dependent launch and one read of the input less per cycle)?  Returns the buffer the
// cycle of pc_apply(., zdst) starts from, or nullptr.
double* fused_first_sweep_target(sns_ctx* h, double* zdst) {
    if (h->opt.pc_type != SNS_PC_AMG || h->levels.size() < 2 || h->opt.amg_fine_cycle != 0 || !h->pc_ready) return nullptr;
    const Level& L = h->levels[0];
    if (lp_format(h, L) == 0 || !L.dinv32 || L.n_owned <= 0) return nullptr;
    if (block_active(h, 0) && (!L.binv32 || L.n_blk <= 0)) return nullptr;      // (aggregate blocks: k_bfirst_bicg, see fused_vector_kernel)
    if (h->rep_level == 1) return nullptr;                       // level 0 is only the source of the replicated copy
    double* x = (h->n > h->n_owned && !fine_tails_unused(h)) ? h->levels[0].x : zdst;   // (as pc_apply chooses the cycle's vector)
    return cycle_start_buffer(h, 0, x);
}

// ---- BiCGStab (right-preconditioned; the recurrences of oracle/solve.py:bicgstab_bj) ----
// Latency-lean formulation: rho / alpha / omega / beta live on the device (sc[]), the vector kernels read them
// there, and the three reductions of the textbook iteration are two -- <rhat, v>, then ONE pass for
// (t.s, t.t, rhat.s, rhat.t, s.s), from which omega, the next rho and ||r||^2 follow (k_bicg_dots5).  The host
// reads (||r||^2, flags) once per iteration, asynchronously: the copy is enqueued, then the x/r update and the
// FIRST HALF of the next iteration (p, M p, A M p, <rhat, v>, alpha: none of it touches x or r) are enqueued
// behind it, and only then does the host wait for the copy's event -- the GPU never idles on the stopping test.
// A converged claim is confirmed by the explicitly computed ||r|| before the loop is left.
int bicgstab(sns_ctx* h, const double* b, double* x, int* its_out, int* reason_out, double* rnorm_out, int stall_window) {
    const sns_options& o = h->opt;
    const int64_t nd = nred_of(h);
    const int g = vec_grid(nd);
    double *r, *rhat, *p, *v, *s, *t, *ph, *sh;
    SNS_TRY(get_vec(h, 0, &r)); SNS_TRY(get_vec(h, 1, &rhat)); SNS_TRY(get_vec(h, 2, &p));
    SNS_TRY(get_vec(h, 3, &v)); SNS_TRY(get_vec(h, 4, &s)); SNS_TRY(get_vec(h, 5, &t));
    SNS_TRY(get_vec(h, 6, &ph)); SNS_TRY(get_vec(h, 7, &sh));
    double* sc = h->d_scal + 128;                         // device scalar block of this solver
    double* red = h->d_scal + 144;                        // reduction results
    double* hpin = h->h_scal + 512;                       // pinned landing zone of (rr, flags)
    if (!h->ev_it) HIP_TRY(hipEventCreateWithFlags(&h->ev_it, hipEventDisableTiming));
    double bnorm, rn;
    SNS_TRY(norm2(h, b, &bnorm));
    SNS_TRY(op_residual(h, x, b, r));
    // ||r0||^2 stays on the device as the first rho (rhat = r0); the host needs it for the start-up test
    hipLaunchKernelGGL(k_dot2, dim3(g), dim3(256), 0, h->stream, nd, r, r, h->partial);
    SNS_TRY(reduce_to(h, g, 2, red));
    hipLaunchKernelGGL(k_bicg_init, dim3(1), dim3(64), 0, h->stream, sc, red);
    {
        double v0[2];
        SNS_TRY(fetch(h, red, 2, v0));
        rn = std::sqrt(v0[0]);
    }
    const double tol = std::max(o.ksp_rtol * bnorm, o.ksp_atol);
    if (o.monitor) std::printf("  0 KSP Residual norm %.12e\n", rn);
    int its = 0, reason = 0;
    if (!(rn == rn)) reason = SNS_KSP_DIVERGED_NANORINF;
    else if (rn <= tol) reason = (rn <= o.ksp_atol) ? SNS_KSP_CONVERGED_ATOL : SNS_KSP_CONVERGED_RTOL;
    if (!reason && o.ksp_max_it < 1) reason = SNS_KSP_DIVERGED_ITS;
    if (!reason) {
        HIP_TRY(hipMemcpyAsync(rhat, r, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemsetAsync(p, 0, nd * sizeof(double), h->stream));
        HIP_TRY(hipMemsetAsync(v, 0, nd * sizeof(double), h->stream));
        auto first_half = [&](bool p_done) -> int {       // p, ph = M p, v = A ph, alpha
            if (!p_done) hipLaunchKernelGGL(k_bicg_p, dim3(g), dim3(256), 0, h->stream, nd, r, sc, v, p);
            SNS_TRY(pc_apply(h, p, ph));
            SNS_TRY(op_apply_dot(h, ph, v, rhat));        // v = A ph with the fused partial sums of <rhat, v>
            SNS_TRY(reduce_bicg<1>(h, h->dot_partials, red, sc));                  // alpha
            return SNS_OK;
        };
        SNS_TRY(first_half(false));
        const double rn0 = rn;
        double best_rn = rn;
        int best_it = 0;
        for (its = 1;; ++its) {
            if (double* z1 = fused_first_sweep_target(h, sh)) {
                const Level& L0 = h->levels[0];
                if (block_active(h, 0)) {
                    const int32_t ns = 8 * L0.n_blk;
                    const unsigned gb = (unsigned)((ns + 63) / 64);
                    if (L0.binv_fmt == 2)
                        hipLaunchKernelGGL((k_bfirst_bicg<2, 1>), dim3(gb), dim3(256), 0, h->stream, ns, L0.blk_rows, (const void*)L0.binv32,
                                           L0.omega, z1, sc, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, v,
                                           (double*)nullptr, r, (double*)nullptr, s);
                    else
                        hipLaunchKernelGGL((k_bfirst_bicg<1, 1>), dim3(gb), dim3(256), 0, h->stream, ns, L0.blk_rows, (const void*)L0.binv32,
                                           L0.omega, z1, sc, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, v,
                                           (double*)nullptr, r, (double*)nullptr, s);
                } else {
                    hipLaunchKernelGGL(k_bicg_s_first, dim3(g), dim3(256), 0, h->stream, nd, r, sc, v, s, L0.dinv32, L0.omega, z1);
                }
                h->first_sweep_done = true;
            } else {
                hipLaunchKernelGGL(k_bicg_s, dim3(g), dim3(256), 0, h->stream, nd, r, sc, v, s);
            }
            SNS_TRY(pc_apply(h, s, sh));
            SNS_TRY(op_apply(h, sh, t));
            hipLaunchKernelGGL(k_bicg_dots5, dim3(g), dim3(256), 0, h->stream, nd, s, t, rhat, h->partial);
            SNS_TRY(reduce_bicg<2>(h, g, red, sc));                       // omega, next rho / beta, ||r||^2, flags
            HIP_TRY(hipMemcpyAsync(hpin, sc + 4, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipEventRecord(h->ev_it, h->stream));
            // speculative first half of the next iteration, enqueued BEFORE the host looks at this one's result; its p-update
            // rides on the x / r update (k_bicg_xrp)
            const bool spec = its < o.ksp_max_it;
            if (spec) {
                if (double* z1 = fused_first_sweep_target(h, ph)) {
                    const Level& L0 = h->levels[0];
                    if (block_active(h, 0)) {
                        const int32_t ns = 8 * L0.n_blk;
                        const unsigned gb = (unsigned)((ns + 63) / 64);
                        if (L0.binv_fmt == 2)
                            hipLaunchKernelGGL((k_bfirst_bicg<2, 2>), dim3(gb), dim3(256), 0, h->stream, ns, L0.blk_rows,
                                               (const void*)L0.binv32, L0.omega, z1, sc, (const double*)ph, (const double*)sh,
                                               (const double*)t, (const double*)v, x, r, p, s);
                        else
                            hipLaunchKernelGGL((k_bfirst_bicg<1, 2>), dim3(gb), dim3(256), 0, h->stream, ns, L0.blk_rows,
                                               (const void*)L0.binv32, L0.omega, z1, sc, (const double*)ph, (const double*)sh,
                                               (const double*)t, (const double*)v, x, r, p, s);
                    } else {
                        hipLaunchKernelGGL(k_bicg_xrp_first, dim3(g), dim3(256), 0, h->stream, nd, sc, ph, sh, s, t, v, x, r, p,
                                           L0.dinv32, L0.omega, z1);
                    }
                    h->first_sweep_done = true;
                } else {
                    hipLaunchKernelGGL(k_bicg_xrp, dim3(g), dim3(256), 0, h->stream, nd, sc, ph, sh, s, t, v, x, r, p);
                }
                SNS_TRY(first_half(true));
            } else {
                hipLaunchKernelGGL(k_bicg_xr, dim3(g), dim3(256), 0, h->stream, nd, sc, ph, sh, s, t, x, r);
            }
            HIP_TRY(hipEventSynchronize(h->ev_it));
            ++h->ctr_host_syncs;
            SNS_TRY(peer_check(h->comm.get()));
            const double rr = hpin[0];
            const int flags = (int)hpin[1];
            rn = std::sqrt(rr);
            if (o.monitor) std::printf("%3d KSP Residual norm %.12e\n", its, rn);
            if ((flags & 1) || !(rn == rn) || std::isinf(rn)) { reason = SNS_KSP_DIVERGED_NANORINF; break; }
            if (rn <= tol) {
                // the three-term formula can lose digits when ||r|| << ||s||: confirm with the vector itself
                double rtrue;
                SNS_TRY(norm2(h, r, &rtrue));
                if (rtrue <= tol) {
                    rn = rtrue;
                    reason = (rn <= o.ksp_atol) ? SNS_KSP_CONVERGED_ATOL : SNS_KSP_CONVERGED_RTOL;
                    break;
                }
            }
            if (flags & 2) { reason = SNS_KSP_DIVERGED_BREAKDOWN; break; }
            if (its >= o.ksp_max_it) { reason = SNS_KSP_DIVERGED_ITS; break; }
            // stagnation watch of the damping-retry feature (stall_window > 0 only on an attempt that can still be retried):
            // BiCGStab under an over-relaxed smoother often does not break down outright but wanders without ever getting
            // anywhere.  Only REAL stagnation ends the attempt: no new best residual at all for stall_window
            // iterations, or, after stall_window iterations, a best residual still at or above the initial one.  A solve
            // that converges slowly -- BiCGStab plateaus on convection-dominated Jacobians -- keeps setting new bests and is
            // left alone, like PETSc's bcgs would leave it.  The attempt's reason is SNS_KSP_STALLED (not a breakdown).
            if (rn < best_rn) { best_rn = rn; best_it = its; }
            if (stall_window > 0 && (its - best_it >= stall_window || (its >= stall_window && best_rn >= rn0))) {
                reason = SNS_KSP_STALLED;
                break;
            }
            if (flags & 4) { reason = SNS_KSP_DIVERGED_BREAKDOWN; ++its; break; }   // rho == 0 stops the NEXT iteration
        }
        // the stopping test runs on the RECURRENCE residual (as PETSc's bcgs does); what is reported is the true one,
        // ||b - A x|| of the returned iterate, from one more operator pass (0.5 ms of a 145-ms solve at 10 M tets)
        if (reason != SNS_KSP_DIVERGED_NANORINF) {
            SNS_TRY(op_residual(h, x, b, t));
            SNS_TRY(norm2(h, t, &rn));
        }
    }
    *its_out = its;
    *reason_out = reason;
    *rnorm_out = rn;
    return SNS_OK;
}

// ---- TFQMR (Freund 1993) on B = A M^-1: the reference's KSP type ('tfqmr', :77, :199, :282) ----
// Same recurrences as oracle/c/sns_oracle.c:orc_solve(method=1).  The quasi-residual bound
// tau*sqrt(m+1) drives the stopping test (as in PETSc); the true residual is reported at the end.
int tfqmr(sns_ctx* h, const double* b, double* x, int* its_out, int* reason_out, double* rnorm_out) {
    const sns_options& o = h->opt;
    const int64_t nd = nred_of(h);
    const int g = vec_grid(nd);
    double *w, *y1, *y2, *u1, *u2, *d, *v, *xh, *rt, *tmp;
    SNS_TRY(get_vec(h, 0, &w)); SNS_TRY(get_vec(h, 1, &y1)); SNS_TRY(get_vec(h, 2, &y2)); SNS_TRY(get_vec(h, 3, &u1));
    SNS_TRY(get_vec(h, 4, &u2)); SNS_TRY(get_vec(h, 5, &d)); SNS_TRY(get_vec(h, 6, &v)); SNS_TRY(get_vec(h, 7, &xh));
    SNS_TRY(get_vec(h, 8, &rt)); SNS_TRY(get_vec(h, 9, &tmp));
    auto axpby = [&](double a, const double* xx, double bb, double* yy) {
        hipLaunchKernelGGL(k_axpby, dim3(g), dim3(256), 0, h->stream, nd, a, xx, bb, yy);
    };
    auto lin3 = [&](double a, const double* xx, double bb, const double* yy, double c, double* zz) {
        hipLaunchKernelGGL(k_axpbypcz, dim3(g), dim3(256), 0, h->stream, nd, a, xx, bb, yy, c, zz);
    };
    auto applyB = [&](const double* in, double* out) -> int {
        SNS_TRY(pc_apply(h, in, tmp));
        return op_apply(h, tmp, out);
    };
    double bnorm, rn;
    SNS_TRY(norm2(h, b, &bnorm));
    SNS_TRY(op_residual(h, x, b, w));
    SNS_TRY(norm2(h, w, &rn));
    const double tol = std::max(o.ksp_rtol * bnorm, o.ksp_atol);
    if (o.monitor) std::printf("  0 KSP Residual norm %.12e\n", rn);
    int its = 0, reason = 0;
    if (!(rn == rn)) reason = SNS_KSP_DIVERGED_NANORINF;
    else if (rn <= tol) reason = (rn <= o.ksp_atol) ? SNS_KSP_CONVERGED_ATOL : SNS_KSP_CONVERGED_RTOL;
    if (!reason) {
        HIP_TRY(hipMemcpyAsync(y1, w, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(rt, w, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        SNS_TRY(applyB(y1, v));
        HIP_TRY(hipMemcpyAsync(u1, v, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemsetAsync(d, 0, nd * sizeof(double), h->stream));
        HIP_TRY(hipMemsetAsync(xh, 0, nd * sizeof(double), h->stream));
        double tau = rn, theta = 0.0, eta = 0.0, rho = rn * rn;
        bool done = false;
        for (its = 1; its <= o.ksp_max_it && !done; ++its) {
            double sigma;
            SNS_TRY(dot(h, rt, v, &sigma));
            if (sigma == 0.0 || rho == 0.0) { reason = SNS_KSP_DIVERGED_BREAKDOWN; break; }
            const double alpha = rho / sigma;
            lin3(1.0, y1, -alpha, v, 0.0, y2);
            SNS_TRY(applyB(y2, u2));
            for (int m = 0; m < 2; ++m) {
                const double* um = m == 0 ? u1 : u2;
                const double* ym = m == 0 ? y1 : y2;
                axpby(-alpha, um, 1.0, w);
                axpby(1.0, ym, theta * theta * eta / alpha, d);
                double wn;
                SNS_TRY(norm2(h, w, &wn));
                theta = wn / tau;
                const double c = 1.0 / std::sqrt(1.0 + theta * theta);
                tau = tau * theta * c;
                eta = c * c * alpha;
                axpby(eta, d, 1.0, xh);
                rn = tau * std::sqrt((double)(2 * its - 1 + m) + 1.0);
                if (o.monitor) std::printf("%3d.%d KSP Residual bound %.12e\n", its, m, rn);
                if (!(rn == rn)) { reason = SNS_KSP_DIVERGED_NANORINF; done = true; break; }
                if (rn <= tol) { done = true; break; }
            }
            if (done) break;
            double rho_new;
            SNS_TRY(dot(h, rt, w, &rho_new));
            const double beta = rho_new / rho;
            rho = rho_new;
            lin3(1.0, w, beta, y2, 0.0, y1);
            SNS_TRY(applyB(y1, u1));
            lin3(1.0, u1, beta, u2, beta * beta, v);
        }
        if (its > o.ksp_max_it) its = o.ksp_max_it;
        SNS_TRY(pc_apply(h, xh, tmp));
        axpby(1.0, tmp, 1.0, x);
        SNS_TRY(op_residual(h, x, b, w));
        SNS_TRY(norm2(h, w, &rn));
        if (!reason) {
            if (done && rn <= 10.0 * tol) reason = (rn <= o.ksp_atol) ? SNS_KSP_CONVERGED_ATOL : SNS_KSP_CONVERGED_RTOL;
            else reason = SNS_KSP_DIVERGED_ITS;
        }
    }
    *its_out = its;
    *reason_out = reason;
    *rnorm_out = rn;
    return SNS_OK;
}

// ---- FGMRES(m), right preconditioning, classical Gram-Schmidt with one re-orthogonalisation ----
int fgmres(sns_ctx* h, const double* b, double* x, int* its_out, int* reason_out, double* rnorm_out) {
    const sns_options& o = h->opt;
    const int m = std::min(200, std::max(1, o.gmres_restart));
    const int64_t nd = nred_of(h), ld = ld_of(h);
    const int g = vec_grid(nd);
    if (h->gm_m != m) {
        if (h->gm_V) { (void)hipFree(h->gm_V); (void)hipFree(h->gm_Z); (void)hipFree(h->d_h); }
        SNS_TRY(dev_alloc(&h->gm_V, (size_t)(m + 1) * ld));
        SNS_TRY(dev_alloc(&h->gm_Z, (size_t)m * ld));
        SNS_TRY(dev_alloc(&h->d_h, (size_t)2 * (m + 16)));
        HIP_TRY(hipMemset(h->gm_V, 0, (size_t)(m + 1) * ld * sizeof(double)));
        HIP_TRY(hipMemset(h->gm_Z, 0, (size_t)m * ld * sizeof(double)));
        h->gm_m = m;
    }
    double* V = h->gm_V;
    double* Z = h->gm_Z;
    const int S = m + 16;                 // stride of one coefficient block
    double* dh1 = h->d_h;                 // pass-1 coefficients [0, m+8)
    double* dh2 = h->d_h + S;             // pass-2 coefficients [0, m+8), then (w.w, w.w) at [m+8, m+10)
    std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), gv(m + 1), y(m), hcol(2 * (m + 16));
    double bnorm, rn;
    SNS_TRY(norm2(h, b, &bnorm));
    const double tol = std::max(o.ksp_rtol * bnorm, o.ksp_atol);
    int its = 0, reason = 0;
    double* r = V;                         // V[0] doubles as the residual vector
    SNS_TRY(op_residual(h, x, b, r));
    SNS_TRY(norm2(h, r, &rn));
    if (o.monitor) std::printf("  0 KSP Residual norm %.12e\n", rn);
    while (!reason) {
        if (!(rn == rn) || std::isinf(rn)) { reason = SNS_KSP_DIVERGED_NANORINF; break; }
        if (rn <= tol) { reason = (rn <= o.ksp_atol) ? SNS_KSP_CONVERGED_ATOL : SNS_KSP_CONVERGED_RTOL; break; }
        if (its >= o.ksp_max_it) { reason = SNS_KSP_DIVERGED_ITS; break; }
        hipLaunchKernelGGL(k_scale_copy, dim3(g), dim3(256), 0, h->stream, nd, 1.0 / rn, r, V);
        std::fill(gv.begin(), gv.end(), 0.0);
        gv[0] = rn;
        int j = 0;
        double res = rn;
        for (; j < m && its < o.ksp_max_it; ++j) {
            double* vj = V + (size_t)j * ld;
            double* zj = Z + (size_t)j * ld;
            double* w = V + (size_t)(j + 1) * ld;
            SNS_TRY(pc_apply(h, vj, zj));
            SNS_TRY(op_apply(h, zj, w));
            const int nv = j + 1;
            // CGS2 with TWO global reductions per iteration: pass 1 dots; pass 2 dots + (w.w), the new
            // norm follows from ||w - V h2||^2 = w.w - |h2|^2 (V orthonormal).
            for (int pass = 0; pass < 2; ++pass) {
                double* dh = pass == 0 ? dh1 : dh2;
                for (int c0 = 0; c0 < nv; c0 += 8) {
                    const int cn = std::min(8, nv - c0);
                    hipLaunchKernelGGL(k_multi_dot8, dim3(g), dim3(256), 0, h->stream, nd, cn, V + (size_t)c0 * ld, ld,
                                       w, h->partial);
                    reduce_local(h, g, 8, dh + c0);
                }
                if (pass == 1) {
                    hipLaunchKernelGGL(k_dot2, dim3(g), dim3(256), 0, h->stream, nd, w, w, h->partial);
                    reduce_local(h, g, 2, dh + (m + 8));          // k_dot2 emits (x.y, y.y): both are w.w here
                    SNS_TRY(allreduce(h, dh, m + 10));
                } else {
                    SNS_TRY(allreduce(h, dh, nv));
                }
                for (int c0 = 0; c0 < nv; c0 += 8) {
                    const int cn = std::min(8, nv - c0);
                    hipLaunchKernelGGL(k_multi_axpy8, dim3(g), dim3(256), 0, h->stream, nd, cn, V + (size_t)c0 * ld,
                                       ld, dh + c0, -1.0, w, (double*)nullptr);
                }
            }
            // one device->host transfer per iteration: h1[0..nv), h2[0..nv), w.w
            SNS_TRY(fetch(h, h->d_h, 2 * S, hcol.data()));
            double* Hj = &H[(size_t)j * (m + 1)];             // column j
            double h2sq = 0.0;
            for (int k = 0; k < nv; ++k) {
                Hj[k] = hcol[k] + hcol[S + k];
                h2sq += hcol[S + k] * hcol[S + k];
            }
            const double ww = hcol[S + (m + 8)];
            double wn2 = ww - h2sq;
            double wn;
            if (!(wn2 > 1e-6 * ww)) SNS_TRY(norm2(h, w, &wn));   // heavy cancellation: measure it
            else wn = std::sqrt(wn2);
            Hj[nv] = wn;
            if (wn > 0.0) hipLaunchKernelGGL(k_scale_copy, dim3(g), dim3(256), 0, h->stream, nd, 1.0 / wn, w, w);
            for (int k = 0; k < j; ++k) {                      // previous rotations
                const double t0 = cs[k] * Hj[k] + sn[k] * Hj[k + 1];
                Hj[k + 1] = -sn[k] * Hj[k] + cs[k] * Hj[k + 1];
                Hj[k] = t0;
            }
            const double den = std::hypot(Hj[j], Hj[j + 1]);
            cs[j] = den > 0 ? Hj[j] / den : 1.0;
            sn[j] = den > 0 ? Hj[j + 1] / den : 0.0;
            Hj[j] = den;
            Hj[j + 1] = 0.0;
            gv[j + 1] = -sn[j] * gv[j];
            gv[j] = cs[j] * gv[j];
            res = std::fabs(gv[j + 1]);
            ++its;
            if (o.monitor) std::printf("%3d KSP Residual norm %.12e\n", its, res);
            if (res <= tol || wn == 0.0 || !(res == res)) { ++j; break; }
        }
        // y = H^-1 g ; x += Z y
        for (int k = j - 1; k >= 0; --k) {
            double sacc = gv[k];
            for (int q = k + 1; q < j; ++q) sacc -= H[(size_t)q * (m + 1) + k] * y[q];
            y[k] = sacc / H[(size_t)k * (m + 1) + k];
        }
        HIP_TRY(hipMemcpyAsync(dh1, y.data(), j * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));             // y is a stack-lifetime host buffer
        for (int c0 = 0; c0 < j; c0 += 8) {
            const int cn = std::min(8, j - c0);
            hipLaunchKernelGGL(k_multi_axpy8, dim3(g), dim3(256), 0, h->stream, nd, cn, Z + (size_t)c0 * ld, ld,
                               dh1 + c0, 1.0, x, (double*)nullptr);
        }
        SNS_TRY(op_residual(h, x, b, r));
        SNS_TRY(norm2(h, r, &rn));
    }
    *its_out = its;
    *reason_out = reason;
    *rnorm_out = rn;
    return SNS_OK;
}

int krylov(sns_ctx* h, const double* b, double* x, int* its, int* reason, double* rnorm) {
    if (!h->has_matrix) { set_error("krylov_solve before a matrix was assembled"); return SNS_E_STATE; }
    if (!h->pc_ready && h->opt.pc_type != SNS_PC_NONE) SNS_TRY(pc_setup(h));
    h->first_sweep_done = false;                 // (a solve that ended in an error between setting and consuming it must not leak it)
    h->ctr_host_syncs = h->ctr_allreduce = h->ctr_exchange = 0;
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    // A solve that BREAKS DOWN (or produces NaN/Inf) under the AMG preconditioner is retried ONCE, from the same initial
    // guess, with every level's block-Jacobi damping scaled by 0.7 (opt.amg_retry_damping, default on): the damping
    // estimate (|lambda|max of Dinv A + a growth check on the dominant mode) is not a bound for a non-symmetric
    // operator, and at cell Reynolds numbers of 5-10 a slightly over-relaxed smoother is what breaks BiCGStab down
    // (measured: jittered 648 k-tet duct, Re 200: auto damping fails after 218 iterations, 0.7 x converges).  A solve
    // that merely runs out of iterations (DIVERGED_ITS) is NOT retried: like PETSc, the reason is reported and that is
    // it.  Because BiCGStab under an over-relaxed smoother more often STAGNATES than breaks down (the same 648 k-tet case,
    // round 3: it wanders between 0.2 and 70 x ||b|| for as long as it is allowed to), the first attempt also ends -- as a
    // breakdown -- when its best residual has not halved for amg_retry_stall_its (100) iterations.  The smaller damping is kept for the later Jacobians of the handle until sns_set_options is called; the
    // retry count and the current factor are visible through sns_get_counters.  *its is the sum over both attempts
    // (<= 2 ksp_max_it).  Not in the reference; converging solves never see it.
    const bool can_retry = h->opt.pc_type == SNS_PC_AMG && h->opt.amg_retry_damping != 0 && h->damping_backoff > 0.4;
    double* x0 = nullptr;
    if (can_retry) {
        SNS_TRY(get_vec(h, 14, &x0));
        HIP_TRY(hipMemcpyAsync(x0, x, nred_of(h) * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
    int its_total = 0;
    h->last_first_reason = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        int rc;
        if (h->opt.ksp_type == SNS_KSP_BICGSTAB)
            rc = bicgstab(h, b, x, its, reason, rnorm, (can_retry && attempt == 0) ? h->opt.amg_retry_stall_its : 0);
        else if (h->opt.ksp_type == SNS_KSP_FGMRES) rc = fgmres(h, b, x, its, reason, rnorm);
        else if (h->opt.ksp_type == SNS_KSP_TFQMR) rc = tfqmr(h, b, x, its, reason, rnorm);
        else { set_error("bad ksp_type"); return SNS_E_ARG; }
        SNS_TRY(rc);
        its_total += *its;
        const bool retryable = *reason == SNS_KSP_DIVERGED_BREAKDOWN || *reason == SNS_KSP_DIVERGED_NANORINF ||
                               *reason == SNS_KSP_STALLED;
        if (!retryable || !can_retry || attempt == 1) break;
        h->last_first_reason = *reason;
        ++h->ctr_retries;
        h->damping_backoff *= 0.7;
        if (h->opt.monitor)
            std::printf("  KSP failed (reason %d after %d iterations): retrying with the smoother damping scaled by %.2f\n",
                        *reason, *its, h->damping_backoff);
        for (auto& L : h->levels) {
            L.omega *= 0.7;
            if (L.omega_checked > 0.0) L.omega_checked *= 0.7;
        }
        HIP_TRY(hipMemcpyAsync(x, x0, nred_of(h) * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
    *its = its_total;
    h->last_ctr[0] = h->ctr_host_syncs; h->last_ctr[1] = h->ctr_allreduce; h->last_ctr[2] = h->ctr_exchange;
    SNS_TRY(halo_exchange(h, x));                          // leave the solution's ghost tail current
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->tm.krylov_ms += ms;
    h->tm.ksp_its += *its;
    time_collect(h);
    HIP_TRY(hipGetLastError());
    return SNS_OK;
}

int timed_assemble(sns_ctx* h, int form, const double* w, double* F, bool want_matrix) {
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    SNS_TRY(assemble(h, form, w, F, want_matrix));
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->tm.assemble_ms += ms;
    return SNS_OK;
}

}  // namespace

// ============================================================================
// C ABI
// ============================================================================
extern "C" {

void sns_default_options(sns_options* o) {
    o->reynolds = 1.0;
    o->ksp_type = SNS_KSP_BICGSTAB;
    o->pc_type = SNS_PC_AMG;
    o->ksp_rtol = 1e-8;
    o->ksp_atol = 1e-50;
    o->ksp_max_it = 10000;
    o->gmres_restart = 30;
    o->snes_rtol = 1e-8;
    o->snes_atol = 1e-8;
    o->snes_stol = 1e-8;
    o->snes_max_it = 30;
    o->amg_max_levels = 12;
    o->amg_coarse_size = 32;
    o->amg_agg_size = 8;
    o->amg_nu = 1;
    o->amg_omega = 0.8;
    o->monitor = 0;
    o->corrected_convection = 0;
    o->amg_f32_matrix = 2;
    o->amg_nu_coarse = 4;
    o->amg_nu_deep = 2;
    o->amg_nu_l2 = 6;
    o->assembly_fused = 1;
    o->amg_sweep_exchange_rows = 0;
    o->amg_replicate_rows = 65536;
    o->amg_post_exchange = 1;
    o->stokes_viscosity = 1.0;
    o->stokes_beta = 0.2;
    o->amg_fine_cycle = 0;
    o->amg_nu_l1_pre = 0;
    o->amg_nu_l1_post = 0;
    o->amg_retry_damping = 1;
    o->amg_retry_stall_its = 100;
    o->halo_overlap = 1;
    o->amg_fused_post = 1;
    o->amg_nu_scale_with_size = 1;
    o->amg_dense_rows = 512;
    o->amg_block_smooth = 1;
    o->amg_bnu_l1 = 3;
    o->amg_bnu_l2 = 4;
    o->amg_bnu_deep = 2;
    o->amg_ritz_limit = 1;
    o->amg_growth_check = 1;
    o->amg_block_max_rows = 0;
    o->amg_block_fine_rows = 600000;
    o->amg_fuse_restrict = 1;
    o->halo_windows = 1;
    o->amg_exact_sweeps = 1;
}

const char* sns_last_error(void) { return g_err.c_str(); }
const char* sns_version(void) { return "sns 0.1 (gfx950)"; }
int sns_abi_version(void) { return SNS_ABI_VERSION; }
int64_t sns_options_size(void) { return (int64_t)sizeof(sns_options); }

// dim 3: points [n*3], cells [E*4];  dim 2: points [n*2], cells [E*3]
static int create_common(int dim, sns_handle* out, int32_t n_nodes, int64_t n_tets, const double* points_in,
                         const int32_t* cells_in, const uint8_t* bc_mask_in, const double* bc_val_in, int device,
                         const sns_options* opt) {
    if (!out || n_nodes <= 0 || n_tets < 0 || !points_in || !cells_in || !bc_mask_in || !bc_val_in) {
        set_error("sns_create: null or empty input");
        return SNS_E_ARG;
    }
    *out = nullptr;
    const int npe = dim + 1;
    // host validation: vertex ids in range, non-degenerate cells (kernels divide by det J)
    for (int64_t t = 0; t < n_tets; ++t) {
        const int32_t* v = cells_in + npe * t;
        for (int a = 0; a < npe; ++a)
            if (v[a] < 0 || v[a] >= n_nodes) { set_error("cell vertex id out of range"); return SNS_E_MESH; }
        const double* x0 = points_in + dim * (int64_t)v[0];
        double J[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        for (int c = 0; c < dim; ++c)
            for (int i = 0; i < dim; ++i) J[i][c] = points_in[dim * (int64_t)v[c + 1] + i] - x0[i];
        const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) -
                           J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                           J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
        if (!(std::fabs(det) > 0.0)) { set_error("degenerate cell " + std::to_string(t)); return SNS_E_MESH; }
    }
    // device layout is the 3-D one in both cases: points in a stride of 3, cells in a stride of 4 (a triangle repeats
    // its last vertex), 4 dofs per node; a 2-D handle constrains the unused z component to 0
    std::vector<double> pts3;
    std::vector<int32_t> cells4;
    std::vector<uint8_t> mask2;
    std::vector<double> val2;
    const double* points = points_in;
    const int32_t* tets = cells_in;
    const uint8_t* bc_mask = bc_mask_in;
    const double* bc_val = bc_val_in;
    if (dim == 2) {
        pts3.assign((size_t)3 * n_nodes, 0.0);
        for (int32_t i = 0; i < n_nodes; ++i) { pts3[3 * (size_t)i] = points_in[2 * (size_t)i]; pts3[3 * (size_t)i + 1] = points_in[2 * (size_t)i + 1]; }
        cells4.resize((size_t)4 * n_tets);
        for (int64_t t = 0; t < n_tets; ++t) {
            for (int a = 0; a < 3; ++a) cells4[4 * (size_t)t + a] = cells_in[3 * t + a];
            cells4[4 * (size_t)t + 3] = cells_in[3 * t + 2];
        }
        mask2.assign(bc_mask_in, bc_mask_in + (size_t)4 * n_nodes);
        val2.assign(bc_val_in, bc_val_in + (size_t)4 * n_nodes);
        for (int32_t i = 0; i < n_nodes; ++i) { mask2[4 * (size_t)i + 2] = 1; val2[4 * (size_t)i + 2] = 0.0; }
        points = pts3.data(); tets = cells4.data(); bc_mask = mask2.data(); bc_val = val2.data();
    }
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<sns_ctx> h(new sns_ctx);
    if (opt) h->opt = *opt; else sns_default_options(&h->opt);
    h->device = device;
    h->dim = dim;
    h->n = n_nodes;
    h->n_owned = n_nodes;
    h->E = n_tets;
    HostPattern P;
    HostAssemblyMaps M;
    try {
        build_pattern(n_nodes, n_tets, tets, P, M, npe);
    } catch (const std::exception& e) {
        set_error(e.what());
        return SNS_E_MESH;
    }
    SNS_TRY(dev_alloc(&h->tets, (size_t)4 * n_tets));
    HIP_TRY(hipMemcpy(h->tets, tets, (size_t)4 * n_tets * sizeof(int32_t), hipMemcpyHostToDevice));
    SNS_TRY(dev_alloc(&h->pts, (size_t)3 * n_nodes));
    HIP_TRY(hipMemcpy(h->pts, points, (size_t)3 * n_nodes * sizeof(double), hipMemcpyHostToDevice));
    SNS_TRY(dev_alloc(&h->bc_mask, (size_t)4 * n_nodes));
    HIP_TRY(hipMemcpy(h->bc_mask, bc_mask, (size_t)4 * n_nodes, hipMemcpyHostToDevice));
    SNS_TRY(dev_alloc(&h->bc_val, (size_t)4 * n_nodes));
    HIP_TRY(hipMemcpy(h->bc_val, bc_val, (size_t)4 * n_nodes * sizeof(double), hipMemcpyHostToDevice));
    {
        std::vector<double> ge((size_t)4 * n_nodes);
        for (size_t i = 0; i < ge.size(); ++i) ge[i] = bc_mask[i] ? bc_val[i] : 0.0;
        SNS_TRY(dev_upload(&h->gext, ge, nullptr));
    }
    SNS_TRY(dev_upload(&h->nt_ptr, M.nt_ptr, nullptr));
    SNS_TRY(dev_upload(&h->nt_idx, M.nt_idx, nullptr));
    SNS_TRY(dev_upload(&h->c_ptr, M.c_ptr, nullptr));
    SNS_TRY(dev_upload(&h->c_idx, M.c_idx, nullptr));
    {
        // lane -> slot map of the scratch-free assembly: off-diagonal slots only, and inside every window of
        // 8192 consecutive slots ordered by descending contribution count, so that the lanes of a wave loop
        // the same number of times (edge valences differ: 4 or 6 tets on a Kuhn mesh) while their gathers
        // stay within the same neighbourhood of the mesh
        const int64_t nnzb = (int64_t)P.colind.size(), WIN = 8192;
        std::vector<int32_t> order;
        order.reserve((size_t)nnzb);
        std::vector<int32_t> row_of((size_t)nnzb);
        for (int32_t i = 0; i < P.n; ++i)
            for (int32_t q = P.rowptr[i]; q < P.rowptr[i + 1]; ++q) row_of[(size_t)q] = i;
        std::vector<std::vector<int32_t>> bucket;
        for (int64_t s0 = 0; s0 < nnzb; s0 += WIN) {
            const int64_t s1 = std::min(nnzb, s0 + WIN);
            for (auto& b : bucket) b.clear();
            for (int64_t q = s0; q < s1; ++q) {
                if (P.colind[(size_t)q] == row_of[(size_t)q]) continue;
                const size_t cnt = (size_t)(M.c_ptr[(size_t)q + 1] - M.c_ptr[(size_t)q]);
                if (bucket.size() <= cnt) bucket.resize(cnt + 1);
                bucket[cnt].push_back((int32_t)q);
            }
            for (size_t c = bucket.size(); c-- > 0;) order.insert(order.end(), bucket[c].begin(), bucket[c].end());
        }
        h->n_od = (int64_t)order.size();
        SNS_TRY(dev_upload(&h->od_order, order, nullptr));
    }
    h->levels.emplace_back();
    h->slot_row.push_back(nullptr);
    h->empty_c.push_back(nullptr);
    h->pong.push_back(nullptr);
    SNS_TRY(upload_pattern(h->levels[0], P, &h->slot_row[0], nullptr));
    h->levels[0].n_owned = n_nodes;
    SNS_TRY(alloc_level_vectors(h->levels[0]));
    SNS_TRY(dev_alloc(&h->pong[0], 4 * (size_t)n_nodes));
    HIP_TRY(hipMemset(h->pong[0], 0, 4 * (size_t)n_nodes * sizeof(double)));
    {   // free mask of level 0 = !bc
        std::vector<uint8_t> fm((size_t)4 * n_nodes);
        for (size_t i = 0; i < fm.size(); ++i) fm[i] = bc_mask[i] ? 0 : 1;
        SNS_TRY(dev_upload(&h->levels[0].free_mask, fm, nullptr));
    }
    // per-block partial sums: vector kernels use <= 2048 blocks x <= 8 sums, the fused SpMV+dot one block per 32 rows
    SNS_TRY(dev_alloc(&h->partial, std::max<size_t>((size_t)65536 * 8, (size_t)n_nodes / 4 + 512)));
    SNS_TRY(dev_alloc(&h->partial2, (size_t)4096 * 8));
    SNS_TRY(dev_alloc(&h->d_scal, 256));
    SNS_TRY(dev_alloc(&h->d_sing, 1));
    HIP_TRY(hipMemset(h->d_sing, 0, sizeof(int)));
    HIP_TRY(hipHostMalloc((void**)&h->h_scal, 1024 * sizeof(double), hipHostMallocDefault));
    HIP_TRY(hipEventCreate(&h->ev0));
    HIP_TRY(hipEventCreate(&h->ev1));
    // the hierarchy is built lazily (first pc_setup) so that sns_attach_comm can shrink n_owned first
    HIP_TRY(hipDeviceSynchronize());
    h->tm = sns_timings{};
    h->graph_disabled = std::getenv("SNS_NO_GRAPH") != nullptr;
    h->r3_estimates = std::getenv("SNS_R3_SPECTRAL_ESTIMATE") != nullptr;
    h->pattern.reset(new HostPattern(std::move(P)));
    h->host_pts.assign(points, points + (size_t)3 * n_nodes);
    *out = h.release();
    return SNS_OK;
}

int sns_create(sns_handle* out, int32_t n_nodes, int64_t n_tets, const double* points, const int32_t* tets,
               const uint8_t* bc_mask, const double* bc_val, int device, const sns_options* opt) {
    return create_common(3, out, n_nodes, n_tets, points, tets, bc_mask, bc_val, device, opt);
}
int sns_create_2d(sns_handle* out, int32_t n_nodes, int64_t n_tris, const double* points, const int32_t* tris,
                  const uint8_t* bc_mask, const double* bc_val, int device, const sns_options* opt) {
    return create_common(2, out, n_nodes, n_tris, points, tris, bc_mask, bc_val, device, opt);
}

}  // extern "C"

namespace {
int ensure_hierarchy(sns_ctx* h) {
    if (!h->pattern) return SNS_OK;                       // already built
    if (h->opt.pc_type != SNS_PC_AMG) return SNS_OK;      // built when (if) AMG is first asked for
    int rc = build_hierarchy(h, *h->pattern);
    h->pattern.reset();
    return rc;
}
}  // namespace

extern "C" {

int sns_destroy(sns_handle h) {
    if (!h) return SNS_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    auto fr = [](void* p) { if (p) (void)hipFree(p); };
    fr(h->tets); fr(h->pts); fr(h->bc_mask); fr(h->bc_val);
    fr(h->nt_ptr); fr(h->nt_idx); fr(h->c_ptr); fr(h->c_idx); fr(h->od_order); fr(h->gext); fr(h->Ke); fr(h->Fe);
    for (auto& L : h->levels) {
        fr(L.rowptr); fr(L.colind); fr(L.diag); fr(L.vals); fr(L.dinv); fr(L.agg); fr(L.m_ptr); fr(L.m_idx);
        fr(L.r_ptr); fr(L.r_idx); fr(L.free_mask); fr(L.x); fr(L.b); fr(L.r); fr(L.dense_inv); fr(L.dense_gj); fr(L.dense_work); fr(L.dense_x32); fr(L.vals32); fr(L.vals16); fr(L.scale16); fr(L.dinv32);
        fr(L.ap_rowptr); fr(L.ap_colind); fr(L.ap_colind_rep); fr(L.ap_ptr); fr(L.ap_idx); fr(L.ap_nib); fr(L.ap_vals32); fr(L.ap_vals16); fr(L.ap_scale16); fr(L.blk_rows); fr(L.blk_of); fr(L.binv32);
    }
    for (auto p : h->slot_row) fr(p);
    for (auto p : h->empty_c) fr(p);
    for (auto p : h->pong) fr(p);
    for (auto p : h->kv) fr(p);
    for (auto& e : h->ev_pool) { (void)hipEventDestroy(e[0]); (void)hipEventDestroy(e[1]); }
    fr(h->d_piv); fr(h->d_sing); fr(h->rep_valmap); fr(h->rep_rowmap); fr(h->rep_vsend); fr(h->rep_vrecv); fr(h->rep_bsend); fr(h->rep_brecv); fr(h->rep_doff); fr(h->rep_dcnt);
    fr(h->arn_V);
    fr(h->partial); fr(h->partial2); fr(h->d_scal); fr(h->gm_V); fr(h->gm_Z); fr(h->d_h);
    fr(h->nw_F); fr(h->nw_y); fr(h->nw_w); fr(h->nw_t);
    if (h->coarse_graph) (void)hipGraphExecDestroy(h->coarse_graph);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (h->gj_stream) (void)hipStreamDestroy(h->gj_stream);
    if (h->h_scal) (void)hipHostFree(h->h_scal);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_it) (void)hipEventDestroy(h->ev_it);
    fr(h->bnd_rows); fr(h->bnd_flag);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->ev_x) (void)hipEventDestroy(h->ev_x);
    if (h->ev_side) (void)hipEventDestroy(h->ev_side);
    fr(h->cg_colmap); fr(h->cg_rows); fr(h->cg_full); fr(h->cg_send); fr(h->cg_recv);
    for (auto& L : h->levels) fr(L.xg);
    if (h->comm) {
        for (auto& p : h->comm->plans) plan_free(p);
        if (h->comm->nccl) (void)ncclCommDestroy(h->comm->nccl);
    }
    delete h;
    return SNS_OK;
}

int sns_set_stream(sns_handle h, void* s) {
    if (!h) return SNS_E_ARG;
    h->stream = (hipStream_t)s;
    return SNS_OK;
}
int sns_set_options(sns_handle h, const sns_options* o) {
    if (!h || !o) return SNS_E_ARG;
    const bool pc_changed = (o->pc_type != h->opt.pc_type) || (o->amg_f32_matrix != h->opt.amg_f32_matrix) ||
                            (o->amg_fused_post != h->opt.amg_fused_post) || (o->amg_block_smooth != h->opt.amg_block_smooth);
    const bool damping_changed = (o->amg_omega != h->opt.amg_omega);
    const bool sweep_exchange_changed = (o->amg_sweep_exchange_rows != h->opt.amg_sweep_exchange_rows) ||
                                        (o->amg_post_exchange != h->opt.amg_post_exchange) ||
                                        (o->amg_fused_post != h->opt.amg_fused_post) || (o->amg_f32_matrix != h->opt.amg_f32_matrix) ||
                                        (o->amg_fine_cycle != h->opt.amg_fine_cycle) || (o->pc_type != h->opt.pc_type);
    h->opt = *o;
    if (h->damping_backoff != 1.0) {                 // a retry's stronger damping does not outlive an options call
        h->damping_backoff = 1.0;
        h->pc_ready = false;
        for (auto& L : h->levels) { L.lambda_max = 0.0; L.omega_checked = 0.0; L.ritz_limit = 0.0; }
    }
    if (sweep_exchange_changed) {
        // rank-local sweeps rely on ghost tails that are never written (zero); sweeps with exchanges fill them
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (size_t l = 0; l < h->levels.size(); ++l) {
            Level& L = h->levels[l];
            const size_t nb = 4 * (size_t)L.n * sizeof(double);
            if (L.x) HIP_TRY(hipMemset(L.x, 0, nb));
            if (l < h->pong.size() && h->pong[l]) HIP_TRY(hipMemset(h->pong[l], 0, nb));
        }
    }
    if (pc_changed || damping_changed) h->pc_ready = false;
    if (damping_changed || pc_changed)
        for (auto& L : h->levels) { L.lambda_max = 0.0; L.omega_checked = 0.0; L.ritz_limit = 0.0; }   // re-estimate and re-verify
    return SNS_OK;
}
int sns_set_form_variant(sns_handle h, double c_inverse, double lsic_scale, double pspg_sign, int one_point_quadrature) {
    if (!h) return SNS_E_ARG;
    if (h->dim != 3) { set_error("sns_set_form_variant: 3-D handles only"); return SNS_E_ARG; }
    FormVariant fv;
    fv.ci = c_inverse;
    fv.lsic = lsic_scale;
    fv.pspg = pspg_sign;
    if (one_point_quadrature) fv.qa = fv.qb = 0.25;
    h->fv = fv;
    h->has_matrix = false;
    h->pc_ready = false;
    return SNS_OK;
}
int sns_get_options(sns_handle h, sns_options* o) {
    if (!h || !o) return SNS_E_ARG;
    *o = h->opt;
    return SNS_OK;
}
int sns_get_sizes(sns_handle h, int32_t* nl, int32_t* no, int64_t* nt, int64_t* nnzb) {
    if (!h) return SNS_E_ARG;
    if (nl) *nl = h->n;
    if (no) *no = h->n_owned;
    if (nt) *nt = h->E;
    if (nnzb) *nnzb = h->levels[0].nnzb;
    return SNS_OK;
}

int sns_comm_unique_id(char id_out[128]) {
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    NCCL_TRY(ncclGetUniqueId(&id));
    std::memcpy(id_out, &id, 128);
    return SNS_OK;
}

static int attach_common(sns_handle h, int rank, int nranks, const char* uid, Team* team, Peer* peer, int32_t n_owned, int n_nbr,
                         const int32_t* nbr, const int32_t* send_ptr, const int32_t* send_idx,
                         const int32_t* recv_ptr, const int32_t* recv_idx) {
    if (!h || nranks < 1 || rank < 0 || rank >= nranks || n_owned < 0 || n_owned > h->n || n_nbr < 0 ||
        (n_nbr > 0 && (!nbr || !send_ptr || !recv_ptr))) {
        set_error("sns_attach_comm: bad arguments");
        return SNS_E_ARG;
    }
    if (h->has_matrix || !h->pattern) {
        set_error("sns_attach_comm must directly follow sns_create");
        return SNS_E_STATE;
    }
    HIP_TRY(hipSetDevice(h->device));
    h->comm.reset(new Comm);
    Comm& c = *h->comm;
    c.rank = rank;
    c.nranks = nranks;
    c.team = team;
    c.peer = peer;
    if (team) SNS_TRY(team_peer(team, h->device, rank, &c.peer));      // (the team runs the peer transport's kernels, sns_comm.h)
    if (uid) {
        ncclUniqueId id;
        std::memcpy(&id, uid, 128);
        NCCL_TRY(ncclCommInitRank(&c.nccl, nranks, id, rank));
    }   // uid == NULL and no team: local part only, the caller moves ghost values and reduces (tests)
    h->n_owned = n_owned;
    h->levels[0].n_owned = n_owned;
    c.plans.emplace_back();
    Plan& p = c.plans[0];
    p.n_own = n_owned;
    p.nbr.assign(nbr, nbr + n_nbr);
    if (n_nbr > 0) {
        p.send_ptr.assign(send_ptr, send_ptr + n_nbr + 1);
        p.recv_ptr.assign(recv_ptr, recv_ptr + n_nbr + 1);
    } else {
        p.send_ptr.assign(1, 0);
        p.recv_ptr.assign(1, 0);
    }
    const int32_t ns = p.n_send(), nr = p.n_recv();
    for (int32_t i = 0; i < ns; ++i)
        if (send_idx[i] < 0 || send_idx[i] >= n_owned) { set_error("send_idx outside owned range"); return SNS_E_ARG; }
    for (int32_t i = 0; i < nr; ++i)
        if (recv_idx[i] < n_owned || recv_idx[i] >= h->n) { set_error("recv_idx outside ghost range"); return SNS_E_ARG; }
    p.h_send_idx.assign(send_idx, send_idx + ns);
    p.h_recv_idx.assign(recv_idx, recv_idx + nr);
    SNS_TRY(plan_upload(p));
    {
        // owned rows that reference a ghost column: the only rows of a level-0 pass that must wait for the halo
        const HostPattern& P = *h->pattern;
        std::vector<int32_t> rows((size_t)std::max(1, n_owned), 0);
        std::vector<uint8_t> flag((size_t)std::max(1, n_owned), 0);
        int32_t nb = 0;
        SNS_TRY(sns_host_boundary_rows(n_owned, P.rowptr.data(), P.colind.data(), rows.data(), &nb));
        for (int32_t q = 0; q < nb; ++q) flag[rows[q]] = 1;
        h->n_bnd = nb;
        rows.resize((size_t)std::max(1, nb));
        SNS_TRY(dev_upload(&h->bnd_rows, rows, nullptr));
        SNS_TRY(dev_upload(&h->bnd_flag, flag, nullptr));
        h->no_overlap = std::getenv("SNS_NO_OVERLAP") != nullptr;
        h->team_overlap = std::getenv("SNS_TEAM_OVERLAP") != nullptr;
    }
    if (!c.active()) {
        // no transport: the per-rank hierarchy must not reference ghost dofs at all
        std::vector<uint8_t> fm((size_t)4 * h->n);
        HIP_TRY(hipMemcpy(fm.data(), h->levels[0].free_mask, fm.size(), hipMemcpyDeviceToHost));
        for (size_t i = (size_t)4 * n_owned; i < fm.size(); ++i) fm[i] = 0;
        HIP_TRY(hipMemcpy(h->levels[0].free_mask, fm.data(), fm.size(), hipMemcpyHostToDevice));
    }
    // first collective of the communicator: both ends of every link agree on its counts (the coarse levels' plans are
    // checked the same way when the hierarchy derives them)
    SNS_TRY(check_plan_symmetry(h, c.plans[0], 0));
    SNS_TRY(connect_plan(h, c.plans[0]));
    return SNS_OK;
}

int sns_attach_comm(sns_handle h, int rank, int nranks, const char uid[128], int32_t n_owned, int n_nbr,
                    const int32_t* nbr, const int32_t* send_ptr, const int32_t* send_idx, const int32_t* recv_ptr,
                    const int32_t* recv_idx) {
    return attach_common(h, rank, nranks, uid, nullptr, nullptr, n_owned, n_nbr, nbr, send_ptr, send_idx, recv_ptr, recv_idx);
}

int sns_team_create(int nranks, void** team_out) {
    if (nranks < 1 || !team_out) return SNS_E_ARG;
    *team_out = new Team(nranks);
    return SNS_OK;
}
int sns_team_destroy(void* team) {
    delete static_cast<Team*>(team);
    return SNS_OK;
}
int sns_attach_team(sns_handle h, void* team, int rank, int nranks, int32_t n_owned, int n_nbr, const int32_t* nbr,
                    const int32_t* send_ptr, const int32_t* send_idx, const int32_t* recv_ptr,
                    const int32_t* recv_idx) {
    if (!team || static_cast<Team*>(team)->n != nranks) { set_error("sns_attach_team: bad team"); return SNS_E_ARG; }
    return attach_common(h, rank, nranks, nullptr, static_cast<Team*>(team), nullptr, n_owned, n_nbr, nbr, send_ptr, send_idx,
                         recv_ptr, recv_idx);
}

int sns_peer_create(int device, int rank, int nranks, int64_t window_bytes, void** peer_out, char ipc_handle_out[64]) {
    if (!peer_out || window_bytes < 0) return SNS_E_ARG;
    Peer* p = nullptr;
    SNS_TRY(peer_create(device, rank, nranks, (size_t)window_bytes, &p, ipc_handle_out));
    *peer_out = p;
    return SNS_OK;
}
int sns_peer_connect(void* peer, const char* ipc_handles) { return peer_connect(static_cast<Peer*>(peer), ipc_handles); }
int sns_peer_disconnect(void* peer) { return peer_close_mappings(static_cast<Peer*>(peer)); }
int sns_peer_destroy(void* peer) { return peer_destroy(static_cast<Peer*>(peer)); }
int sns_peer_check_links(void* peer, int rounds) { return peer_check_links(static_cast<Peer*>(peer), rounds); }
int sns_peer_selftest(int device, int nranks, int halo_nodes, int reps, double us_out[3]) {
    return peer_selftest(device, nranks, halo_nodes, reps, us_out);
}
int sns_attach_peer(sns_handle h, void* peer, int32_t n_owned, int n_nbr, const int32_t* nbr, const int32_t* send_ptr,
                    const int32_t* send_idx, const int32_t* recv_ptr, const int32_t* recv_idx) {
    Peer* p = static_cast<Peer*>(peer);
    if (!p || !p->connected) { set_error("sns_attach_peer: the peer communicator is not connected"); return SNS_E_ARG; }
    if (h && h->device != p->device) { set_error("sns_attach_peer: handle and window live on different devices"); return SNS_E_ARG; }
    return attach_common(h, p->rank, p->nranks, nullptr, nullptr, p, n_owned, n_nbr, nbr, send_ptr, send_idx, recv_ptr, recv_idx);
}

int sns_residual(sns_handle h, int form, const double* w, double* F) {
    if (!h || !F) return SNS_E_ARG;
    return timed_assemble(h, form, w, F, false);
}
int sns_jacobian(sns_handle h, int form, const double* w, double* F) {
    if (!h) return SNS_E_ARG;
    return timed_assemble(h, form, w, F, true);
}
int sns_spmv(sns_handle h, const double* x, double* y) {
    if (!h || !x || !y) return SNS_E_ARG;
    if (!h->has_matrix) { set_error("spmv before a matrix was assembled"); return SNS_E_STATE; }
    SNS_TRY(op_apply(h, const_cast<double*>(x), y));
    return sync_stream(h);
}
int sns_pc_setup(sns_handle h) {
    if (!h) return SNS_E_ARG;
    SNS_TRY(ensure_hierarchy(h));
    return pc_setup(h);
}
int sns_pc_apply(sns_handle h, const double* r, double* z) {
    if (!h || !r || !z) return SNS_E_ARG;
    if (!h->pc_ready && h->opt.pc_type != SNS_PC_NONE) { set_error("pc_apply before pc_setup"); return SNS_E_STATE; }
    SNS_TRY(pc_apply(h, r, z));
    return sync_stream(h);
}
int sns_krylov_solve(sns_handle h, const double* b, double* x, int* its, int* reason, double* rnorm) {
    if (!h || !b || !x || !its || !reason || !rnorm) return SNS_E_ARG;
    SNS_TRY(ensure_hierarchy(h));
    return krylov(h, b, x, its, reason, rnorm);
}

int sns_stokes_solve(sns_handle h, double* U, int* ksp_its, int* reason, double* rnorm) {
    if (!h || !U || !ksp_its || !reason || !rnorm) return SNS_E_ARG;
    SNS_TRY(ensure_hierarchy(h));
    const int64_t nd = nred_of(h), ld = ld_of(h);
    if (!h->nw_F) {
        SNS_TRY(dev_alloc(&h->nw_F, (size_t)ld)); SNS_TRY(dev_alloc(&h->nw_y, (size_t)ld));
        SNS_TRY(dev_alloc(&h->nw_w, (size_t)ld)); SNS_TRY(dev_alloc(&h->nw_t, (size_t)ld));
        HIP_TRY(hipMemset(h->nw_F, 0, ld * sizeof(double))); HIP_TRY(hipMemset(h->nw_y, 0, ld * sizeof(double)));
        HIP_TRY(hipMemset(h->nw_w, 0, ld * sizeof(double))); HIP_TRY(hipMemset(h->nw_t, 0, ld * sizeof(double)));
    }
    // one Newton step of the linear problem from w = 0:  A U = -F(0),  F(0) = lifting, F_B = -g   (:198-214)
    SNS_TRY(timed_assemble(h, SNS_FORM_STOKES, nullptr, h->nw_F, true));
    hipLaunchKernelGGL(k_scale_copy, dim3(vec_grid(nd)), dim3(256), 0, h->stream, nd, -1.0, h->nw_F, h->nw_F);
    HIP_TRY(hipMemsetAsync(U, 0, nd * sizeof(double), h->stream));
    SNS_TRY(krylov(h, h->nw_F, U, ksp_its, reason, rnorm));
    // the Dirichlet rows are identity rows: the reference's ILU-preconditioned solve returns them exactly, a Krylov
    // method under AMG only to its tolerance -- a converged solve hands back the exact data as well
    if (*reason > 0) {
        hipLaunchKernelGGL(k_snap_bc, dim3(vec_grid(nd)), dim3(256), 0, h->stream, nd, h->bc_mask, h->bc_val,
                           1e300, U);
        SNS_TRY(sync_stream(h));
    }
    return SNS_OK;
}

int sns_newton_solve(sns_handle h, double* w, int* its_out, int* reason_out, int* total_ksp, double* hist,
                     int hist_cap) {
    if (!h || !w || !its_out || !reason_out) return SNS_E_ARG;
    SNS_TRY(ensure_hierarchy(h));
    const sns_options& o = h->opt;
    const int64_t nd = nred_of(h), ld = ld_of(h);
    const int g = vec_grid(nd);
    if (!h->nw_F) {
        SNS_TRY(dev_alloc(&h->nw_F, (size_t)ld)); SNS_TRY(dev_alloc(&h->nw_y, (size_t)ld));
        SNS_TRY(dev_alloc(&h->nw_w, (size_t)ld)); SNS_TRY(dev_alloc(&h->nw_t, (size_t)ld));
        HIP_TRY(hipMemset(h->nw_F, 0, ld * sizeof(double))); HIP_TRY(hipMemset(h->nw_y, 0, ld * sizeof(double)));
        HIP_TRY(hipMemset(h->nw_w, 0, ld * sizeof(double))); HIP_TRY(hipMemset(h->nw_t, 0, ld * sizeof(double)));
    }
    double *F = h->nw_F, *y = h->nw_y, *wn = h->nw_w, *Fn = h->nw_t;
    int ksp_total = 0, nh = 0;
    auto record = [&](double f) { if (hist && nh < hist_cap) hist[nh] = f; ++nh; };
    SNS_TRY(halo_exchange(h, w));
    SNS_TRY(timed_assemble(h, SNS_FORM_NS, w, F, true));
    double f, f0;
    SNS_TRY(norm2(h, F, &f));
    f0 = f;
    record(f);
    if (o.monitor) std::printf("  0 SNES Function norm %.12e\n", f);
    int reason = 0, it = 0;
    if (!(f == f)) reason = SNS_SNES_DIVERGED_FNORM_NAN;
    else if (f < o.snes_atol) reason = SNS_SNES_CONVERGED_FNORM_ABS;
    while (!reason) {
        if (it >= o.snes_max_it) { reason = SNS_SNES_DIVERGED_MAX_IT; break; }
        ++it;
        // J y = F
        HIP_TRY(hipMemsetAsync(y, 0, nd * sizeof(double), h->stream));
        int kits = 0, kreason = 0;
        double krn = 0;
        SNS_TRY(krylov(h, F, y, &kits, &kreason, &krn));
        ksp_total += kits;
        if (kreason < 0) { reason = SNS_SNES_DIVERGED_LINEAR_SOLVE; break; }
        // bt line search (cubic backtracking, alpha 1e-4), x_new = x - lambda y
        double initslope;
        {
            double* Jy = Fn;                                 // borrow
            SNS_TRY(op_apply(h, y, Jy));
            SNS_TRY(dot(h, F, Jy, &initslope));
        }
        if (initslope > 0) initslope = -initslope;
        if (initslope == 0) initslope = -1.0;
        double ynorm;
        SNS_TRY(norm2(h, y, &ynorm));
        const double ls_alpha = 1e-4;
        double lam = 1.0, gn = 0.0;
        auto trial = [&](double l) -> int {
            HIP_TRY(hipMemcpyAsync(wn, w, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            hipLaunchKernelGGL(k_axpby, dim3(g), dim3(256), 0, h->stream, nd, -l, y, 1.0, wn);
            SNS_TRY(halo_exchange(h, wn));
            SNS_TRY(timed_assemble(h, SNS_FORM_NS, wn, Fn, false));
            return norm2(h, Fn, &gn);
        };
        SNS_TRY(trial(lam));
        bool ok = 0.5 * gn * gn <= 0.5 * f * f + lam * ls_alpha * initslope;
        if (!ok && gn == gn) {
            double lamprev = lam, gprev = gn;
            double lamtemp = -initslope / (gn * gn - f * f - 2.0 * lam * initslope);
            lam = std::min(std::max(lamtemp, 0.1 * lam), 0.5 * lam);
            for (int k = 0; k < 40; ++k) {
                SNS_TRY(trial(lam));
                if (0.5 * gn * gn <= 0.5 * f * f + lam * ls_alpha * initslope) { ok = true; break; }
                const double t1 = 0.5 * (gn * gn - f * f) - lam * initslope;
                const double t2 = 0.5 * (gprev * gprev - f * f) - lamprev * initslope;
                const double a = (t1 / (lam * lam) - t2 / (lamprev * lamprev)) / (lam - lamprev);
                const double bq = (-lamprev * t1 / (lam * lam) + lam * t2 / (lamprev * lamprev)) / (lam - lamprev);
                const double d = std::max(bq * bq - 3 * a * initslope, 0.0);
                lamtemp = (a == 0) ? -initslope / (2.0 * bq) : (-bq + std::sqrt(d)) / (3.0 * a);
                lamprev = lam;
                gprev = gn;
                lam = std::min(std::max(lamtemp, 0.1 * lam), 0.5 * lam);
            }
        }
        if (!ok) { reason = (gn == gn) ? SNS_SNES_DIVERGED_LINE_SEARCH : SNS_SNES_DIVERGED_FNORM_NAN; break; }
        HIP_TRY(hipMemcpyAsync(w, wn, ld * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        // Dirichlet dofs within round-off of their data become exact (no lifting term from here on, :65)
        hipLaunchKernelGGL(k_snap_bc, dim3(vec_grid(ld)), dim3(256), 0, h->stream, ld, h->bc_mask, h->bc_val, 1e-12, w);
        double xnorm;
        SNS_TRY(norm2(h, w, &xnorm));
        f = gn;
        record(f);
        if (o.monitor) std::printf("%3d SNES Function norm %.12e  (ksp its %d, lambda %.3g)\n", it, f, kits, lam);
        if (f < o.snes_atol) reason = SNS_SNES_CONVERGED_FNORM_ABS;
        else if (f <= o.snes_rtol * f0) reason = SNS_SNES_CONVERGED_FNORM_RELATIVE;
        else if (lam * ynorm < o.snes_stol * xnorm) reason = SNS_SNES_CONVERGED_SNORM_RELATIVE;
        if (reason) break;
        if (it >= o.snes_max_it) { reason = SNS_SNES_DIVERGED_MAX_IT; break; }   // no Jacobian after the last iteration
        SNS_TRY(timed_assemble(h, SNS_FORM_NS, w, F, true));
    }
    *its_out = it;
    *reason_out = reason;
    if (total_ksp) *total_ksp = ksp_total;
    return SNS_OK;
}

int sns_get_bsr(sns_handle h, int32_t* n_rows, int64_t* nnzb, const int32_t** rowptr, const int32_t** colind,
                const double** vals) {
    if (!h) return SNS_E_ARG;
    const Level& L = h->levels[0];
    if (n_rows) *n_rows = L.n;
    if (nnzb) *nnzb = L.nnzb;
    if (rowptr) *rowptr = L.rowptr;
    if (colind) *colind = L.colind;
    if (vals) *vals = L.vals;
    return SNS_OK;
}
int sns_get_element_scratch(sns_handle h, const double** Ke, const double** Fe) {
    if (!h) return SNS_E_ARG;
    if (Ke) *Ke = h->Ke;
    if (Fe) *Fe = h->Fe;
    return SNS_OK;
}
int sns_export(sns_handle h, int what, void* dst, int64_t nbytes) {
    if (!h || !dst) return SNS_E_ARG;
    const Level& L = h->levels[0];
    const void* src = nullptr;
    int64_t need = 0;
    switch (what) {
        case SNS_EXPORT_ROWPTR: src = L.rowptr; need = ((int64_t)L.n + 1) * 4; break;
        case SNS_EXPORT_COLIND: src = L.colind; need = L.nnzb * 4; break;
        case SNS_EXPORT_VALS: src = L.vals; need = L.nnzb * 16 * 8; break;
        case SNS_EXPORT_KE: src = h->Ke; need = h->E * 256 * 8; break;
        case SNS_EXPORT_FE: src = h->Fe; need = h->E * 16 * 8; break;
        default: set_error("sns_export: unknown array"); return SNS_E_ARG;
    }
    if (!src) { set_error("sns_export: array not produced yet"); return SNS_E_STATE; }
    if (need != nbytes) { set_error("sns_export: size mismatch, need " + std::to_string(need)); return SNS_E_ARG; }
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)need, hipMemcpyDeviceToDevice, h->stream));
    return sync_stream(h);
}
int sns_get_counters(sns_handle h, int64_t out[8]) {
    if (!h || !out) return SNS_E_ARG;
    out[0] = h->last_ctr[0];
    out[1] = h->last_ctr[1];
    out[2] = h->last_ctr[2];
    out[3] = h->tm.ksp_its;
    out[4] = h->ctr_retries;
    out[5] = (int64_t)std::llround(h->damping_backoff * 1e6);
    out[6] = h->last_first_reason;
    out[7] = h->levels[0].ap_nnz;
    return SNS_OK;
}
int sns_comm_info(sns_handle h, int32_t out[4]) {
    if (!h || !out) return SNS_E_ARG;
    out[0] = out[1] = out[3] = 0;
    out[2] = 1;
    const Comm* c = h->comm.get();
    if (!c) return SNS_OK;
    out[0] = c->nccl ? 1 : (c->team ? 2 : (c->peer ? 3 : 0));
    out[1] = c->rank;
    out[2] = c->nranks;
    if (c->nccl) {
        int cnt = 0;
        NCCL_TRY(ncclCommCount(c->nccl, &cnt));
        out[3] = cnt;
    }
    return SNS_OK;
}
int sns_get_hierarchy(sns_handle h, int32_t* nlevels, int64_t rows[16], int64_t blocks[16], int32_t sweeps[16], double omega[16]) {
    if (!h || !nlevels) return SNS_E_ARG;
    const int nl = (int)std::min<size_t>(16, h->levels.size());
    *nlevels = nl;
    for (int l = 0; l < nl; ++l) {
        const Level& L = h->levels[l];
        if (rows) rows[l] = L.n_owned;
        if (blocks) blocks[l] = L.nnzb;
        if (sweeps) sweeps[l] = l + 1 < (int)h->levels.size() ? level_nu(h, l) : 0;      // the coarsest level is a dense inverse
        if (omega) omega[l] = L.omega;
    }
    return SNS_OK;
}
int sns_dense_inverse(int device, int32_t N, const double* A, double* Ainv) {
    if (N <= 0 || !A || !Ainv) return SNS_E_ARG;
    HIP_TRY(hipSetDevice(device));
    const int Np = (N + 63) / 64 * 64;
    double *W = nullptr, *work = nullptr;
    int* sing = nullptr;
    SNS_TRY(dev_alloc(&W, (size_t)Np * Np));
    SNS_TRY(dev_alloc(&work, dense_gj_work_doubles(Np)));
    SNS_TRY(dev_alloc(&sing, 1));
    HIP_TRY(hipMemset(sing, 0, sizeof(int)));
    HIP_TRY(hipMemset(W, 0, (size_t)Np * Np * sizeof(double)));
    HIP_TRY(hipMemcpy2D(W, (size_t)Np * sizeof(double), A, (size_t)N * sizeof(double), (size_t)N * sizeof(double), N,
                        hipMemcpyDeviceToDevice));
    if (Np > N) hipLaunchKernelGGL(k_dense_pad_diag, dim3((Np - N + 255) / 256), dim3(256), 0, nullptr, N, Np, W);
    hipStream_t side = nullptr;
    if (std::getenv("SNS_GJ_TWO_STREAMS")) (void)hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
    HIP_TRY(hipDeviceSynchronize());                      // (the null stream does not order a non-blocking side stream)
    hipStream_t mainst = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&mainst, hipStreamNonBlocking));
    dense_gj_inverse(mainst, side, Np, W, work, sing);
    HIP_TRY(hipStreamSynchronize(mainst));
    if (side) { HIP_TRY(hipStreamSynchronize(side)); (void)hipStreamDestroy(side); }
    (void)hipStreamDestroy(mainst);
    HIP_TRY(hipMemcpy2D(Ainv, (size_t)N * sizeof(double), W, (size_t)Np * sizeof(double), (size_t)N * sizeof(double), N,
                        hipMemcpyDeviceToDevice));
    int hs = 0;
    HIP_TRY(hipMemcpy(&hs, sing, sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(W); (void)hipFree(work); (void)hipFree(sing);
    HIP_TRY(hipGetLastError());
    if (hs) { set_error("sns_dense_inverse: zero or non-finite pivot"); return SNS_E_STATE; }
    return SNS_OK;
}
int sns_get_cycle(sns_handle h, int32_t* nlevels, int32_t kind[16], int32_t nu_pre[16], int32_t nu_post[16]) {
    if (!h || !nlevels || !kind || !nu_pre || !nu_post) return SNS_E_ARG;
    const int nl = (int)std::min<size_t>(16, h->levels.size());
    *nlevels = nl;
    for (int l = 0; l < nl; ++l) {
        const Level& L = h->levels[l];
        nu_pre[l] = nu_post[l] = 0;
        if (l + 1 == (int)h->levels.size() && nl > 1) {
            kind[l] = L.dense_gj ? SNS_LEVEL_DIRECT_BLOCKED : (L.dense_inv || h->cg_N > 0) ? SNS_LEVEL_DIRECT : SNS_LEVEL_SWEEPS_ONLY;
            continue;
        }
        kind[l] = block_active(h, l) ? SNS_LEVEL_AGGREGATE_BLOCKS : SNS_LEVEL_NODAL_BLOCKS;
        int a = 1, b = 1;
        level_sweeps(h, l, a, b);
        nu_pre[l] = a;
        nu_post[l] = b;
    }
    return SNS_OK;
}
int sns_get_timings(sns_handle h, sns_timings* t) {
    if (!h || !t) return SNS_E_ARG;
    *t = h->tm;
    t->amg_levels = (int)h->levels.size() - (h->rep_level > 0 ? 1 : 0);
    return SNS_OK;
}
int sns_reset_timings(sns_handle h) {
    if (!h) return SNS_E_ARG;
    h->tm = sns_timings{};
    h->ctr_retries = 0;
    for (int i = 0; i < 8; ++i) { h->kt_ms[i] = 0; h->kt_calls[i] = 0; }
    return SNS_OK;
}
int sns_time_kernels(sns_handle h, int on) {
    if (!h) return SNS_E_ARG;
    h->time_kernels = on != 0;
    return SNS_OK;
}
int sns_get_kernel_times(sns_handle h, double ms_total[8], int64_t calls[8]) {
    if (!h || !ms_total || !calls) return SNS_E_ARG;
    for (int i = 0; i < 8; ++i) { ms_total[i] = h->kt_ms[i]; calls[i] = h->kt_calls[i]; }
    return SNS_OK;
}

int sns_bench_spmv(sns_handle h, const double* x, double* y, int reps, double* ms_avg) {
    if (!h || !x || !y || reps <= 0 || !ms_avg) return SNS_E_ARG;
    if (!h->has_matrix) { set_error("bench_spmv before a matrix was assembled"); return SNS_E_STATE; }
    launch_spmv<SPMV_AX>(h, h->levels[0], h->n_owned, x, y, nullptr, 0.0, nullptr);   // warm
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < reps; ++i) launch_spmv<SPMV_AX>(h, h->levels[0], h->n_owned, x, y, nullptr, 0.0, nullptr);
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *ms_avg = ms / reps;
    HIP_TRY(hipGetLastError());
    return SNS_OK;
}
#ifdef SNS_HARNESS
// interleaved A/B micro-benchmark of kernel variants on the assembled level-0 operator (methodology:
// variants timed alternately in ONE process).  ms_out[v] = average launch ms of variant v.
//   which 0: fp64 y = Ax, default loads (0) vs non-temporal matrix stream (1, production)
//   which 3: fp64 y = Ax, production (0: first 16 blocks up-front) vs the stepped loop of round 1 / early round 2 (1)
//   which 1: low-precision Jacobi sweep, fp16 row-scaled (0) vs fp32 (1) (needs both copies: SNS_BOTH_LP=1)
SNS_API int sns_bench_variants(sns_handle h, int which, int rounds, int reps, double ms_out[2]) {   // (harness build only: not in sns.h)
    if (!h || !ms_out || rounds <= 0 || reps <= 0) return SNS_E_ARG;
    if (!h->has_matrix) { set_error("bench_variants before a matrix was assembled"); return SNS_E_STATE; }
    Level& L = h->levels[0];
    const int32_t rows = h->n_owned;
    double *x, *y, *b;
    SNS_TRY(get_vec(h, 10, &x)); SNS_TRY(get_vec(h, 11, &y)); SNS_TRY(get_vec(h, 12, &b));
    hipLaunchKernelGGL(k_fill_pattern, dim3(vec_grid(4 * (int64_t)rows)), dim3(256), 0, h->stream, 4 * (int64_t)rows, x);
    if (which == 1 && (!L.vals16 || !L.vals32)) { set_error("which 1 needs both the fp16 and the fp32 copy (SNS_BOTH_LP=1)"); return SNS_E_STATE; }
    const int saved_fmt = h->opt.amg_f32_matrix;
    double tot[2] = {0, 0};
    for (int r = 0; r < rounds; ++r)
        for (int v = 0; v < 2; ++v) {
            HIP_TRY(hipEventRecord(h->ev0, h->stream));
            for (int i = 0; i < reps; ++i) {
                const int grid = (rows + 31) / 32;
                if (which == 3) {
                    if (v) hipLaunchKernelGGL((k_spmv<SPMV_AX, 1, 3, 0>), dim3(grid), dim3(256), 0, h->stream, rows, L.rowptr, L.colind, L.vals, x, y, nullptr, L.dinv, 0.0, nullptr, h->partial, (const int32_t*)nullptr, (const uint8_t*)nullptr, 0);
                    else hipLaunchKernelGGL((k_spmv<SPMV_AX, 1, 1, 0>), dim3(grid), dim3(256), 0, h->stream, rows, L.rowptr, L.colind, L.vals, x, y, nullptr, L.dinv, 0.0, nullptr, h->partial, (const int32_t*)nullptr, (const uint8_t*)nullptr, 0);
                } else if (which == 0) {
                    if (v) hipLaunchKernelGGL((k_spmv<SPMV_AX, 1, 1, 0>), dim3(grid), dim3(256), 0, h->stream, rows, L.rowptr, L.colind, L.vals, x, y, nullptr, L.dinv, 0.0, nullptr, h->partial, (const int32_t*)nullptr, (const uint8_t*)nullptr, 0);
                    else hipLaunchKernelGGL((k_spmv<SPMV_AX, 1, 0, 0>), dim3(grid), dim3(256), 0, h->stream, rows, L.rowptr, L.colind, L.vals, x, y, nullptr, L.dinv, 0.0, nullptr, h->partial, (const int32_t*)nullptr, (const uint8_t*)nullptr, 0);
                } else {
                    h->opt.amg_f32_matrix = v ? 1 : 2;
                    launch_pc_spmv<SPMV_JACOBI>(h, L, rows, x, y, b, 0.7);
                }
            }
            HIP_TRY(hipEventRecord(h->ev1, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
            tot[v] += ms / reps;
        }
    h->opt.amg_f32_matrix = saved_fmt;
    ms_out[0] = tot[0] / rounds;
    ms_out[1] = tot[1] / rounds;
    HIP_TRY(hipGetLastError());
    return SNS_OK;
}
#endif  // SNS_HARNESS
int sns_bench_assemble(sns_handle h, int form, const double* w, double* F, int reps, double* ms_avg) {
    if (!h || reps <= 0 || !ms_avg) return SNS_E_ARG;
    SNS_TRY(assemble(h, form, w, F, true));
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < reps; ++i) SNS_TRY(assemble(h, form, w, F, true));
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *ms_avg = ms / reps;
    return SNS_OK;
}
int sns_bench_collective(sns_handle h, int which, int count, int reps, double* ms_avg) {
    if (!h || reps <= 0 || !ms_avg || count < 0) return SNS_E_ARG;
    Comm* c = h->comm.get();
    if (!c || !c->active()) { set_error("sns_bench_collective: no communicator attached"); return SNS_E_STATE; }
    double *snd = nullptr, *rcv = nullptr;
    if (which == 0) {
        if (c->plans.empty() || !h->levels[0].xg) { set_error("sns_bench_collective: no level-0 halo plan"); return SNS_E_STATE; }
    } else if (which == 1) {
        if (count < 1 || count > 32) { set_error("sns_bench_collective: all-reduce of 1..32 doubles"); return SNS_E_ARG; }
        HIP_TRY(hipMemsetAsync(h->d_scal + 64, 0, 32 * sizeof(double), h->stream));
    } else if (which == 2) {
        SNS_TRY(dev_alloc(&snd, (size_t)std::max(1, count)));
        SNS_TRY(dev_alloc(&rcv, (size_t)std::max(1, count) * c->nranks));
        HIP_TRY(hipMemset(snd, 0, (size_t)std::max(1, count) * sizeof(double)));
    } else {
        return SNS_E_ARG;
    }
    auto one = [&]() -> int {
        if (which == 0) return comm_exchange(c, c->plans[0], h->levels[0].xg, h->stream);
        if (which == 1) return comm_allreduce_sum(c, h->d_scal + 64, count, h->stream);
        return comm_allgather(c, snd, rcv, count, h->stream);
    };
    int rc = SNS_OK;
    for (int i = 0; i < 5 && rc == SNS_OK; ++i) rc = one();
    if (rc == SNS_OK) {
        (void)hipEventRecord(h->ev0, h->stream);
        for (int i = 0; i < reps && rc == SNS_OK; ++i) rc = one();
        (void)hipEventRecord(h->ev1, h->stream);
    }
    (void)hipStreamSynchronize(h->stream);
    if (rc == SNS_OK) rc = peer_check(c);
    if (rc == SNS_OK) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, h->ev0, h->ev1) != hipSuccess) rc = SNS_E_HIP;
        *ms_avg = ms / reps;
    }
    if (which == 0) (void)hipMemset(h->levels[0].xg, 0, 4 * (size_t)h->levels[0].n * sizeof(double));   // (the cycle relies on zero ghosts there)
    if (snd) (void)hipFree(snd);
    if (rcv) (void)hipFree(rcv);
    return rc;
}

}  // extern "C"
