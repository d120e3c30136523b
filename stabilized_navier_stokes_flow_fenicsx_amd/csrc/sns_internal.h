// Internal structures shared by the host setup code, the HIP kernels and the
// C-ABI layer.  Not installed; the public surface is include/sns.h.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "sns.h"

namespace sns {

// ---- host-side symbolic data (built once per mesh, sns_host.cpp) ------------
struct HostPattern {
    int32_t n = 0;                       // block rows (nodes)
    int64_t nnzb = 0;
    std::vector<int32_t> rowptr, colind; // BSR pattern, columns sorted per row
    std::vector<int32_t> diag;           // slot of the diagonal block of each row
};

struct HostAssemblyMaps {
    // node -> incident (tet*4 + a), tet order: gather list for the residual
    std::vector<int64_t> nt_ptr;         // n+1
    std::vector<int32_t> nt_idx;         // 4*E   (tet*4+a fits int32 up to 536M tets)
    // BSR slot -> contributing element blocks (tet*16 + a*4 + b), fixed order
    std::vector<int64_t> c_ptr;          // nnzb+1
    std::vector<int32_t> c_idx;          // 16*E  (tet*16+ab as an UNSIGNED 32-bit pattern: up to 268 M tets)
};

struct HostAggregation {
    int32_t nc = 0;
    std::vector<int32_t> agg;            // fine node -> aggregate
    std::vector<int32_t> m_ptr, m_idx;   // aggregate -> member fine nodes
    HostPattern coarse;                  // coarse BSR pattern
    std::vector<int64_t> r_ptr;          // coarse slot -> fine slots (RAP gather)
    std::vector<int32_t> r_idx;
};

struct HostAP {                          // pattern of M = A P (fine rows x coarse columns) + its gather lists
    int32_t n = 0;
    int64_t nnz = 0;
    std::vector<int32_t> rowptr, colind; // per fine row: the aggregates its columns fall into (sorted)
    std::vector<int32_t> slot_row;       // fine row of each M slot
    std::vector<int32_t> ap_ptr, ap_idx; // M slot -> fine slots summed into it
    std::vector<uint64_t> nib;           // the same relation per fine block of a row: nibble j = row-local M slot of block j (15: the column
                                         // takes no part); ~0 = the row has more than 16 blocks or more than 8 slots (k_lp_copies16's register path)
};
void build_ap_pattern(const HostPattern& F, int32_t n_rows, const std::vector<int32_t>& agg_all, HostAP& M);

void build_pattern(int32_t n_nodes, int64_t n_tets, const int32_t* tets, HostPattern& P,
                   HostAssemblyMaps& M, int npe = 4);
void build_aggregation(const HostPattern& fine, int max_agg, HostAggregation& A);
// same, but only nodes [0, n_active) take part (distributed level 0: owned nodes only)
void build_aggregation_active(const HostPattern& fine, int32_t n_active, int max_agg, HostAggregation& A);
void build_coarse_from_agg(const HostPattern& F, int32_t n_owned_fine, const std::vector<int32_t>& agg_all,
                           int32_t nc_owned, int32_t nc_total, HostAggregation& A);

// ---- device-side level of the operator hierarchy ----------------------------
struct Level {
    int32_t n = 0;                       // local block rows (owned + ghost on level 0)
    int32_t n_owned = 0;                 // rows that are solved for (== n when serial)
    int64_t n_global = 0;                // rows of this level over all ranks (a replicated level: its own rows)
    int64_t nnzb = 0;
    int32_t *rowptr = nullptr, *colind = nullptr, *diag = nullptr;
    double* vals = nullptr;              // nnzb*16, block row-major
    double* dinv = nullptr;              // n*16
    float* vals32 = nullptr;             // fp32 copy of vals for the preconditioner passes (amg_f32_matrix)
    void* vals16 = nullptr;              // fp16 copy, row-scaled (amg_f32_matrix = 2): 4 halfs per block row
    float* scale16 = nullptr;            // its scales, one per dof row
    float* dinv32 = nullptr;             // fp32 copy of dinv for the low-precision Jacobi sweeps
    // to the next coarser level
    int32_t nc = 0;
    int32_t* agg = nullptr;              // n
    int32_t *m_ptr = nullptr, *m_idx = nullptr;
    int64_t* r_ptr = nullptr;
    int32_t* r_idx = nullptr;
    uint8_t* free_mask = nullptr;        // 4*n: 1 where dof takes part in transfer (level 0: !bc), else all 1
    // M = A P of this level for the fused first post-smoothing sweep (k_post_lp): pattern + gather lists (symbolic, once),
    // values per numeric setup straight into the level's low-precision format (k_lp_copies16 / k_ap_cvt32)
    int64_t ap_nnz = 0;
    int32_t *ap_rowptr = nullptr, *ap_colind = nullptr, *ap_ptr = nullptr, *ap_idx = nullptr;
    int32_t* ap_colind_rep = nullptr;    // partitioned level right above the replicated tail's source: M's columns in the ids of the replicated level
    uint64_t* ap_nib = nullptr;          // per row: nibble j = the row-local M slot of block j (15: none); ~0 = row too long for k_lp_copies16's registers
    float* ap_vals32 = nullptr;
    void* ap_vals16 = nullptr;
    float* ap_scale16 = nullptr;
    // aggregate-block Jacobi smoother (amg_block_smooth, csrc/sns_block.hip): member rows of every aggregate padded to 8 slots
    // (-1: none), built with the hierarchy; the aggregates' inverse diagonal blocks (fp32, 1024 floats each) per numeric setup
    int32_t* blk_rows = nullptr;
    int32_t* blk_of = nullptr;           // node -> its smoother block (-1: ghost node)
    int32_t n_blk = 0;                   // smoother blocks (= nc, plus one per aggregate of more than 8 nodes)
    void* binv32 = nullptr;              // (fp32 blocks, or fp16 + row scales: the format of the level's matrix copy, binv_fmt)
    int binv_fmt = 0;
    // work vectors (4*n doubles)
    double *x = nullptr, *b = nullptr, *r = nullptr;
    double* xg = nullptr;                // distributed runs: copy of the iterate whose ghost tail is exchanged
    // coarsest level: dense inverse (4n x 4n), row-major
    double* dense_inv = nullptr;
    // ... or, for a coarsest level of up to amg_dense_rows rows, the blocked Gauss-Jordan inverse (csrc/sns_dense.hip): the
    // Np x Np fp64 matrix the elimination works in (Np = 4n rounded up to a multiple of 64), its workspace, and the finished
    // inverse as fp32 (leading dimension Np) for the cycle's matvec
    int dense_np = 0;
    double *dense_gj = nullptr, *dense_work = nullptr;
    float* dense_x32 = nullptr;
    // block-Jacobi damping actually used on this level (<= amg_omega, limited by 4/(3 |lambda|max(Dinv A)))
    double omega = 0.8;
    double lambda_max = 0.0;
    double omega_checked = 0.0;          // damping that passed the growth test at the last fresh estimate
    double ritz_limit = 0.0;             // stability limit of the damping from the Ritz values of S A (amg_ritz_limit; 0 = not estimated)
};

void aggregate_nodes(const HostPattern& F, int32_t n_active, int max_agg, std::vector<int32_t>& agg, int32_t& nc,
                     const double* pts = nullptr, int* which = nullptr);
void set_error(const std::string& s);

}  // namespace sns
