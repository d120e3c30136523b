// Kernel declarations shared between sns_kernels.hip and sns_api.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "sns.h"
#include "sns_peer_dev.h"

namespace sns {

constexpr int EL_TETS_PER_BLOCK = 16;   // tets per 256-thread workgroup of k_element

// internal form ids of the 2-D handles (the public API keeps SNS_FORM_STOKES / SNS_FORM_NS; sns_create_2d handles
// map them to these)
#define SNS_FORM_STOKES_2D 2
#define SNS_FORM_UGN_2D 3

enum SpmvMode { SPMV_AX = 0, SPMV_B_MINUS_AX = 1, SPMV_JACOBI = 2, SPMV_AX_DOT = 3 };

// Perturbations of the 3-D NS form for the study "what do the reference-held constants tell apart" (sns_set_form_variant; the
// staged element kernel only): C_I of the G-metric tau (:237, 36), a factor on the LSIC viscosity nu_L (:249, 1), the sign of the
// PSPG term (tau res_M, grad q) (:247, +1) and the quadrature points (a, b) of the 4-point rule (:222; a = b = 1/4 collapses
// it to the 1-point centroid rule of the same total weight).  The defaults ARE the reference's form.
struct FormVariant {
    double ci = 36.0, lsic = 1.0, pspg = 1.0;
    double qa = 0.1381966011250105, qb = 0.5854101966249685;
    bool is_default() const { return ci == 36.0 && lsic == 1.0 && pspg == 1.0 && qa == 0.1381966011250105 && qb == 0.5854101966249685; }
};

template <int FORM, bool corrected>
__global__ void k_element(int64_t n_tets, const int32_t* tets, const double* pts, const double* w,
                          const uint8_t* bc_mask, const double* bc_val, double nu, int store_K, double* Ke,
                          double* Fe, FormVariant fv);
template <int FORM, bool corrected>
__global__ void k_fused_offdiag(int64_t n_od, const int32_t* od_order, const int64_t* c_ptr, const int32_t* c_idx, const int32_t* slot_row,
                                const int32_t* colind, const int32_t* tets, const double* pts, const double* w,
                                const uint8_t* bc_mask, double nu, double aux, double* vals);
template <int FORM, bool corrected>
__global__ void k_fused_diag(int32_t n_rows, const int32_t* diag, const int64_t* c_ptr, const int32_t* c_idx,
                             const int32_t* tets, const double* pts, const double* w, const uint8_t* bc_mask,
                             const double* bc_val, double nu, double aux, double* vals, double* F);
template <int FORM, bool corrected>
__global__ void k_fused_lift(int32_t n_rows, const int32_t* diag, const int64_t* c_ptr, const int32_t* c_idx,
                             const int32_t* tets, const double* pts, const double* w, const uint8_t* bc_mask,
                             const double* dl, double nu, double* F);
__global__ void k_bc_defect(int64_t ndof, const uint8_t* bc_mask, const double* bc_val, const double* w, double* dl);
template <bool corrected>
__global__ void k_residual_tet(int64_t n_tets, const int32_t* tets, const double* pts, const double* w, double nu,
                               double* Fe);
__global__ void k_residual_tri(int64_t n_tris, const int32_t* tets, const double* pts, const double* w, double nu,
                               double* Fe);
__global__ void k_bc_residual(int64_t ndof, const uint8_t* bc_mask, const double* bc_val, const double* w, double* F);
__global__ void k_snap_bc(int64_t ndof, const uint8_t* bc_mask, const double* bc_val, double rel_tol, double* w);
__global__ void k_count_bc_violations(int64_t ndof, const uint8_t* bc_mask, const double* bc_val, const double* w,
                                      double* partial);
__global__ void k_gather_matrix(int64_t nnzb, const int64_t* c_ptr, const int32_t* c_idx, const int32_t* slot_row,
                                const int32_t* colind, const uint8_t* bc_mask, const double* Ke, double* vals);
__global__ void k_gather_residual(int32_t n_rows, const int64_t* nt_ptr, const int32_t* nt_idx,
                                  const uint8_t* bc_mask, const double* bc_val, const double* w, const double* Fe,
                                  double* F);
template <int MODE, int FINE, int NT, int SPLIT>
__global__ void k_spmv(int32_t n_rows, const int32_t* rowptr, const int32_t* colind, const double* vals,
                       const double* x, double* y, const double* bvec, const double* dinv, double omega,
                       const double* dotw, double* partial, const int32_t* row_list, const uint8_t* skip,
                       int partial_off, GhostSrc gs);
template <int MODE, int FINE, int SPLIT, int FMT, int UP = 1>
__global__ void k_spmv_lp(int32_t n_rows, const int32_t* rowptr, const int32_t* colind, const void* vals,
                          const float* scale, const double* x, double* y, const double* bvec, const float* dinv32,
                          double omega, const int32_t* row_list, const uint8_t* skip, GhostSrc gs);
template <int FMT, int FINE>
__global__ void k_post_lp(int32_t n_rows, const int32_t* rowptr, const int32_t* colind, const void* vals, const float* scale,
                          const double* xc, const double* x_pre, const double* res1, const float* dinv32, double omega,
                          const int32_t* agg, const uint8_t* free_mask, double* y, GhostSrc gs);
__global__ void k_ap_cvt32(int32_t n_rows, const int32_t* rowptr_m, const int32_t* colind_m, const int32_t* ap_ptr,
                           const int32_t* ap_idx, const double* vals_f, const int32_t* agg, const uint8_t* free_mask, float4* out);
template <int WITH_M>
__global__ void k_lp_copies16(int32_t n_rows, const int32_t* rowptr, const double* vals, uint2* out, float* scale,
                              const int32_t* rowptr_m, const int32_t* colind_m, const int32_t* ap_ptr, const int32_t* ap_idx,
                              const uint64_t* ap_nib, const int32_t* agg, const uint8_t* free_mask, uint2* out_m, float* scale_m);
__global__ void k_cvt_f32(int64_t n, const double* x, float* y);
__global__ void k_dinv(int32_t n, const int32_t* diag, const double* vals, double* dinv);
__global__ void k_bjacobi(int32_t n, const double* dinv, const double* r, double omega, double* z);
__global__ void k_bjacobi32(int32_t n, const float* dinv32, const double* r, double omega, double* z);
__global__ void k_reduce_final(int nblocks, int nred, const double* partial, double* out);
__global__ void k_reduce_chunks(int nblocks, int nred, const double* partial, double* out);
template <int WHICH>
__global__ void k_reduce_final_bicg(int nblocks, const double* partial, double* red_out, double* sc);
template <int WHICH>
__global__ void k_reduce_final_bicg_peer(int nblocks, const double* partial, double* red_out, double* sc, PeerArgs pa);
__global__ void k_dot2(int64_t n, const double* x, const double* y, double* partial);
__global__ void k_axpby(int64_t n, double a, const double* x, double b, double* y);
__global__ void k_axpbypcz(int64_t n, double a, const double* x, double b, const double* y, double c, double* z);
__global__ void k_bicg_p(int64_t n, const double* r, const double* sc, const double* v, double* p);
__global__ void k_bicg_alpha(double* sc, const double* red);
__global__ void k_bicg_s(int64_t n, const double* r, const double* sc, const double* v, double* s);
__global__ void k_bicg_dots5(int64_t n, const double* s, const double* t, const double* rhat, double* partial);
__global__ void k_bicg_omega(double* sc, const double* red);
__global__ void k_bicg_xr(int64_t n, const double* sc, const double* ph, const double* sh, const double* s,
                          const double* t, double* x, double* r);
__global__ void k_bicg_xrp(int64_t n, const double* sc, const double* ph, const double* sh, const double* s,
                           const double* t, const double* v, double* x, double* r, double* p);
__global__ void k_bicg_init(double* sc, const double* rr0);
__global__ void k_bicg_s_first(int64_t n, const double* r, const double* sc, const double* v, double* s, const float* dinv32,
                               double omega_pc, double* z);
__global__ void k_bicg_xrp_first(int64_t n, const double* sc, const double* ph, const double* sh, const double* s, const double* t,
                                 const double* v, double* x, double* r, double* p, const float* dinv32, double omega_pc, double* z);
__global__ void k_multi_dot8(int64_t n, int nv, const double* V, int64_t ldv, const double* w, double* partial);
__global__ void k_multi_axpy8(int64_t n, int nv, const double* V, int64_t ldv, const double* h, double sign,
                              double* w, double* partial);
__global__ void k_scale_copy(int64_t n, double a, const double* x, double* y);
__global__ void k_restrict(int32_t nc, const int32_t* m_ptr, const int32_t* m_idx, const uint8_t* free_mask,
                           const double* r, double* bc, const float* dinv32_c, double omega_c, double* z_c);
__global__ void k_prolong_add(int32_t n, const int32_t* agg, const uint8_t* free_mask, const double* xc, double* x);
__global__ void k_galerkin(int64_t nnzb_c, const int64_t* r_ptr, const int32_t* r_idx, const double* vals_f,
                           const int32_t* slot_row_c, const int32_t* colind_c, const uint8_t* fixed_c,
                           const int32_t* m_ptr, double* vals_c);
__global__ void k_scatter_blocks(int64_t nsrc, const int32_t* valmap, const double* src, double* dst);
__global__ void k_gather_rows(int32_t n, const int32_t* rowmap, const double* src, double* dst);
__global__ void k_empty_coarse(int32_t nc, const int32_t* m_ptr, const int32_t* m_idx, const uint8_t* free_mask,
                               uint8_t* empty_c);
__global__ void k_bsr_to_dense(int32_t n, const int32_t* rowptr, const int32_t* colind, const double* vals,
                               double* D);
__global__ void k_dense_inverse(int N, double* A, int* piv, int* singular);
__global__ void k_dense_matvec(int N, const double* D, const double* x, double* y, int nrows);
__global__ void k_bsr_to_dense_map(int32_t n_rows, const int32_t* rowptr, const int32_t* colind, const double* vals,
                                   const int32_t* colmap, int N, double* D);
__global__ void k_pad_identity(int r0, int r1, int g0, int N, double* D);
// csrc/sns_dense.hip: blocked Gauss-Jordan inverse of the dense coarsest level on the fp64 matrix cores + its fp32 matvec
__global__ void k_bsr_to_dense_ld(int64_t nnzb, const int32_t* slot_row, const int32_t* colind, const double* vals, int Np, double* D);
__global__ void k_dense_pad_diag(int N, int Np, double* D);
__global__ void k_dense_to_f32(int64_t n, const double* A, float* X);
__global__ void k_dense_matvec32(int N, int Np, const float* X, const double* b, double* y);
void dense_gj_inverse(hipStream_t s, hipStream_t side, int Np, double* A, double* work, int* singular);
inline size_t dense_gj_work_doubles(int Np) { return (size_t)4 * 64 * Np + 4 * 4096; }
// csrc/sns_block.hip: aggregate-block Jacobi smoother of the coarse levels (FMT = format of the level's matrix copy AND of the
// aggregates' inverse blocks: 1 fp32, 2 fp16 with row scales)
template <int FMT, int GH = 0>
__global__ void k_bsweep(int32_t n_slots, const int32_t* blk_rows, const int32_t* rowptr, const int32_t* colind, const void* vals,
                         const float* scale, const void* binv, const double* x, double* y, const double* bvec, double omega,
                         GhostSrc gs, PutDst pd);
template <int FMT, int GH = 0>
__global__ void k_bpost(int32_t n_slots, const int32_t* blk_rows, const int32_t* rowptr, const int32_t* colind, const void* vals,
                        const float* scale, const void* binv, const double* xc, const double* xc_own, const double* x_pre,
                        const double* res1, double omega, const int32_t* agg, const uint8_t* free_mask, double* y, GhostSrc gs,
                        PutDst pd);
template <int FMT>
__global__ void k_bfirst(int32_t n_slots, const int32_t* blk_rows, const void* binv, const double* bvec, double omega, double* z,
                         PutDst pd);
template <int FMT, int OP>
__global__ void k_bfirst_bicg(int32_t n_slots, const int32_t* blk_rows, const void* binv, double omega_pc, double* z, const double* sc,
                              const double* ph, const double* sh, const double* t, const double* v, double* x, double* r, double* p,
                              double* s, PutDst pd);
template <int FMT>
__global__ void k_restrict_blk(int32_t n_slots, const int32_t* blk_rows_c, const int32_t* m_ptr, const int32_t* m_idx,
                               const uint8_t* free_mask, const double* r, double* bc, const void* binv_c, double omega_c, double* z_c,
                               PutDst pd);
template <int FMT, int MODE, int GH = 0>
__global__ void k_resid_restrict(int32_t nc, int32_t n_cslots, const int32_t* blk_rows_c, const int32_t* m_ptr, const int32_t* m_idx,
                                 const uint8_t* free_mask, const int32_t* rowptr, const int32_t* colind, const void* vals, const float* scale,
                                 const double* x, const double* b, double* r_out, double* bc, const float* dinv32_c, const void* binv_c,
                                 double omega_c, double* z_c, GhostSrc gs, AgPut agp, PutDst pd);
template <int FMT>
__global__ void k_bfirst_gather(int32_t n_slots, const int32_t* blk_rows, const void* binv, double omega, double* z, double* b_out,
                                const int32_t* rowmap, AgGet ag);
template <int FMT>
__global__ void k_binv(int32_t nblk, const int32_t* blk_rows, const int32_t* blk_of, const int32_t* rowptr, const int32_t* colind,
                       const double* vals, void* binv, int* singular);
inline size_t binv_bytes_per_block(int fmt) { return fmt == 2 ? (size_t)136 * 16 : (size_t)4096; }
__global__ void k_pack(int32_t m, const int32_t* idx, const double* x, double* buf);
__global__ void k_unpack(int32_t m, const int32_t* idx, const double* buf, double* x);
__global__ void k_fill_pattern(int64_t n, double* x);
__global__ void k_scale_by_rsqrt(int64_t n, const double* s2, const double* x, double* y);
__global__ void k_fill_slot_row(int32_t n, const int32_t* rowptr, int32_t* slot_row);

}  // namespace sns
