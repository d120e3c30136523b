// HIP kernels for gfx950 (MI355X, CDNA4): wave64, fp64 vector ALU, LDS staging.
// No MFMA anywhere: every kernel here is HBM-bandwidth bound or fp64-VALU bound
// (DESIGN.md "Kernels").  All kernels are written for wave64 only.
//
// K1  k_element<FORM>        element Jacobian blocks + residual, LDS-staged
//     k_gather_matrix        BSR slot <- sum of element blocks (atomic-free scatter)
//     k_gather_residual      node    <- sum of element residuals, F_B = w_B - g
// K2  k_spmv<MODE>           BSR4 SpMV, 8 lanes per block row, fused epilogues
// K3  k_dinv, (Jacobi sweep = k_spmv<MODE_JACOBI>)
// K4  vector kernels with fused dots (two-stage deterministic reductions)
//     AMG transfer / Galerkin kernels, dense coarse inverse
// K7  halo pack / unpack
#include <hip/hip_runtime.h>

#include <cstdint>

#include "sns_kernels.h"

namespace sns {

// ============================================================================
// K1: element kernel
// ============================================================================
// 16 lanes per tet: lane (a,b) owns the 4x4 block coupling local vertices a,b.
// One wave = 4 tets, one 256-thread workgroup = 16 tets.  Nodal coordinates and
// the nodal state are staged in LDS once per tet; lanes 0..3 of a tet compute
// the geometry and the per-quadrature-point scalars (tau, nu_LSIC, u_q, ...)
// into LDS, then all 16 lanes accumulate their block over the 4 points.  The
// 256 blocks of a workgroup are transposed through LDS so that every store
// instruction writes 1 KiB contiguous.
//
// Math: SURVEY.md Appendix A == oracle/element.py, restating
// NavierStokesChannelFlow.py:160-172 (Stokes) and :220-251 + :46 (NS + exact
// Gateaux derivative).

#define QA 0.1381966011250105
#define QB 0.5854101966249685

constexpr int EL_TPB = 256;
constexpr int EL_TETS = EL_TETS_PER_BLOCK;

struct TetLds {
    double X[12];
    double W[16];
    double GW[16];     // (g - w) on Dirichlet dofs, 0 elsewhere  -> lifting (:65)
    double g[12];      // grad phi_a  [a][j]
    double gu[9];      // grad u      [i][j]
    double guga[12];   // (grad u) g_a [a][i]
    double sc[4];      // wd = |detJ|/24, div u, tr G, h^2 (Stokes)
    double q[4][16];   // per point: u[3], p, tau, nuL, Gu[3], conv[3], s[4]
};

__device__ __forceinline__ double phi_q(int q, int a) { return q == a ? QB : QA; }

template <int FORM, bool corrected>
__global__ __launch_bounds__(EL_TPB, 3) void k_element(int64_t n_tets, const int32_t* __restrict__ tets,
                                                    const double* __restrict__ pts,
                                                    const double* __restrict__ w,
                                                    const uint8_t* __restrict__ bc_mask,
                                                    const double* __restrict__ bc_val, double nu,
                                                    int store_K, double* __restrict__ Ke,
                                                    double* __restrict__ Fe, FormVariant fv) {
    // staging data and the output transpose tile share LDS (the tile is written after a barrier
    // that retires every read of the staging data): 34.8 KB per workgroup -> 4 workgroups per CU
    constexpr size_t SH_BYTES = sizeof(TetLds) * EL_TETS, TILE_BYTES = sizeof(double) * EL_TPB * 17;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[SH_BYTES > TILE_BYTES ? SH_BYTES : TILE_BYTES];
    TetLds* sh = reinterpret_cast<TetLds*>(lds_raw);
    double* tile = reinterpret_cast<double*>(lds_raw);

    const int tid = threadIdx.x;
    const int tl = tid >> 4;            // tet within workgroup
    const int l = tid & 15;             // lane within tet
    const int a = l >> 2, b = l & 3;
    const int64_t t = (int64_t)blockIdx.x * EL_TETS + tl;
    const bool live = t < n_tets;
    TetLds& S = sh[tl];

    // ---- stage nodal data ----------------------------------------------------
    if (live) {
        const int32_t na = tets[4 * t + a];
        const int64_t dof = 4 * (int64_t)na + b;      // lane l <-> local dof 4a+c with c=b
        const double wv = w ? w[dof] : 0.0;
        S.W[l] = wv;
        S.GW[l] = bc_mask[dof] ? (bc_val[dof] - wv) : 0.0;
        if (l < 12) {
            const int32_t nv = tets[4 * t + l / 3];
            S.X[l] = pts[3 * (int64_t)nv + l % 3];
        }
    }
    __syncthreads();

    // ---- geometry + per-point scalars (lanes 0..3 = quadrature points) -------
    if (live && l < 4) {
        const int q = l;
        const double* X = S.X;
        double J[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            J[i][0] = X[3 + i] - X[i];
            J[i][1] = X[6 + i] - X[i];
            J[i][2] = X[9 + i] - X[i];
        }
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
        const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
        const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
        const double id = 1.0 / det;
        double K[3][3];                                   // K = J^-1
        K[0][0] = c00 * id;
        K[1][0] = c01 * id;
        K[2][0] = c02 * id;
        K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
        K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
        K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
        K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
        K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
        K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
        double g[4][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            g[1][j] = K[0][j];
            g[2][j] = K[1][j];
            g[3][j] = K[2][j];
            g[0][j] = -(K[0][j] + K[1][j] + K[2][j]);
        }
        double G[3][3];                                   // G = K^T K   (:235)
        double trG = 0.0, GG = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                G[i][j] = K[0][i] * K[0][j] + K[1][i] * K[1][j] + K[2][i] * K[2][j];
                GG += G[i][j] * G[i][j];
                if (i == j) trG += G[i][j];
            }
        const double* W = S.W;
        double gu[3][3], gp[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            gp[j] = W[3] * g[0][j] + W[7] * g[1][j] + W[11] * g[2][j] + W[15] * g[3][j];
#pragma unroll
            for (int i = 0; i < 3; ++i)
                gu[i][j] = W[i] * g[0][j] + W[4 + i] * g[1][j] + W[8 + i] * g[2][j] + W[12 + i] * g[3][j];
        }
        const double divu = gu[0][0] + gu[1][1] + gu[2][2];
        if (q == 0) {
#pragma unroll
            for (int aa = 0; aa < 4; ++aa)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    S.g[3 * aa + j] = g[aa][j];
                    // (grad u) g_a, or for the corrected form g_a^T (grad u) (the transpose contraction)
                    S.guga[3 * aa + j] = corrected
                        ? (g[aa][0] * gu[0][j] + g[aa][1] * gu[1][j] + g[aa][2] * gu[2][j])
                        : (gu[j][0] * g[aa][0] + gu[j][1] * g[aa][1] + gu[j][2] * g[aa][2]);
                }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) S.gu[3 * i + j] = gu[i][j];
            double h2 = 0.0;                               // CellDiameter^2 (:168)
#pragma unroll
            for (int aa = 0; aa < 4; ++aa)
#pragma unroll
                for (int bb = aa + 1; bb < 4; ++bb) {
                    const double d0 = X[3 * aa] - X[3 * bb], d1 = X[3 * aa + 1] - X[3 * bb + 1],
                                 d2 = X[3 * aa + 2] - X[3 * bb + 2];
                    h2 = fmax(h2, d0 * d0 + d1 * d1 + d2 * d2);
                }
            S.sc[0] = fabs(det) * (1.0 / 24.0);
            S.sc[1] = divu;
            S.sc[2] = trG;
            S.sc[3] = h2;
        }
        if (FORM == SNS_FORM_NS) {
            double u[3], p = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) u[i] = 0.0;
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                const double ph = (q == aa) ? fv.qb : fv.qa;
#pragma unroll
                for (int i = 0; i < 3; ++i) u[i] += ph * W[4 * aa + i];
                p += ph * W[4 * aa + 3];
            }
            double Gu[3], conv[3], r[3];
            double uGu = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                Gu[i] = G[i][0] * u[0] + G[i][1] * u[1] + G[i][2] * u[2];
                uGu += u[i] * Gu[i];
                conv[i] = gu[i][0] * u[0] + gu[i][1] * u[1] + gu[i][2] * u[2];          // (u.grad)u :243
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)                       // res_M :241  (reference: dot(u, grad(u)) = (grad u)^T u)
                r[j] = (corrected ? conv[j] : (gu[0][j] * u[0] + gu[1][j] * u[1] + gu[2][j] * u[2])) + gp[j];
            const double tau = 1.0 / sqrt(uGu + fv.ci * nu * nu * GG);                    // :237-238 (C_I = 36)
            const double nuL = fv.lsic / (trG * tau);                                    // :249
            double* Q = S.q[q];
            Q[0] = u[0]; Q[1] = u[1]; Q[2] = u[2]; Q[3] = p; Q[4] = tau; Q[5] = nuL;
            Q[6] = Gu[0]; Q[7] = Gu[1]; Q[8] = Gu[2];
            Q[9] = conv[0]; Q[10] = conv[1]; Q[11] = conv[2];
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) Q[12 + aa] = r[0] * g[aa][0] + r[1] * g[aa][1] + r[2] * g[aa][2];
        }
    }
    __syncthreads();

    // ---- block (a,b): accumulate over quadrature points -----------------------
    double acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0;
    double Rl[4] = {0.0, 0.0, 0.0, 0.0};                 // residual of local node a (lanes with b==0 only, NS)
    // Jacobian blocks are needed when they are stored, or when this tet has a Dirichlet dof whose
    // value differs from g (lifting term A0[:,B](g - x_B), :65).  Residual-only evaluations of the
    // line search skip them otherwise.
    const unsigned long long lift_mask = __ballot(live && S.GW[l] != 0.0);
    const bool need_blocks = store_K || FORM == SNS_FORM_STOKES || ((lift_mask >> (threadIdx.x & 48)) & 0xFFFFull) != 0;
    if (live) {
        const double ga0 = S.g[3 * a], ga1 = S.g[3 * a + 1], ga2 = S.g[3 * a + 2];
        const double gb0 = S.g[3 * b], gb1 = S.g[3 * b + 1], gb2 = S.g[3 * b + 2];
        const double ga[3] = {ga0, ga1, ga2}, gb[3] = {gb0, gb1, gb2};
        const double gab = ga0 * gb0 + ga1 * gb1 + ga2 * gb2;
        const double wd = S.sc[0], divu = S.sc[1], trG = S.sc[2];
        if (FORM == SNS_FORM_STOKES) {
            const double vol = 4.0 * wd;
            const double muT = 0.2 * S.sc[3];                                            // :169
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                acc[5 * i] = vol * gab;                     // (grad u, grad v)
                acc[4 * i + 3] = -wd * ga[i];               // -(p, div v)   int phi_b = vol/4 = wd
                acc[12 + i] = wd * gb[i];                   // +(div u, q)
            }
            acc[15] = muT * vol * gab;                      // mu_T (grad p, grad q)
        } else {
            const double* gum = S.gu;
            const double gg0 = S.guga[3 * a], gg1 = S.guga[3 * a + 1], gg2 = S.guga[3 * a + 2];
            const double guga_a[3] = {gg0, gg1, gg2};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double* Q = S.q[q];
                const double pa = (q == a) ? fv.qb : fv.qa, pb = (q == b) ? fv.qb : fv.qa;
                const double u[3] = {Q[0], Q[1], Q[2]};
                const double tau = Q[4], nuL = Q[5];
                const double Gu[3] = {Q[6], Q[7], Q[8]};
                const double sa = Q[12 + a];
                const double ugb = u[0] * gb0 + u[1] * gb1 + u[2] * gb2;
                const double uga = u[0] * ga0 + u[1] * ga1 + u[2] * ga2;
                // coefficient of delta_ij; convection phi_a (g_b.u); viscous; SUPG tau s_a phi_b
                // (corrected form: test function (u.grad)v = (u.g_a) e_i, so tau*(r.e_i) pieces differ, see below)
                double cu[3], cg[3];
                const double t3 = tau * tau * tau;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double dtau = -t3 * pb * Gu[j];                         // d tau / d u_(b,j)
                    const double dnuL = fv.lsic * (tau / trG) * pb * Gu[j];       // d nu_L
                    cg[j] = dnuL * divu + nuL * gb[j];
                    if (!corrected) {
                        // d(r.g_a) = u_j g_a.g_b + phi_b ((grad u) g_a)_j
                        cu[j] = dtau * sa + tau * (u[j] * gab + pb * guga_a[j]);
                    } else {
                        cu[j] = dtau;                                              // used differently below
                    }
                }
                if (!need_blocks) {
                } else if (!corrected) {
                    const double A1 = pa * ugb + nu * gab + tau * sa * pb;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            acc[4 * i + j] += pa * pb * gum[3 * i + j] + u[i] * cu[j] + ga[i] * cg[j];
                        acc[5 * i] += A1;
                        acc[4 * i + 3] += -pb * ga[i] + tau * u[i] * gab;        // J[(a,i),(b,p)]
                        acc[12 + i] += pa * gb[i] + fv.pspg * cu[i];             // J[(a,p),(b,j)]
                    }
                    acc[15] += fv.pspg * tau * gab;
                } else {
                    // corrected variant: res_M = (u.grad)u + grad p ; SUPG test = (u.grad)v + grad q
                    //   R[(a,i)] += tau (u.g_a) r_i ; R[(a,p)] += tau (r.g_a)
                    // with r = conv + gp: d r_i/d u_(b,j) = delta_ij (u.g_b) + phi_b gu[i][j]
                    const double r0 = Q[9] + 0.0, r1 = Q[10], r2 = Q[11];
                    // r (without grad p) is conv; s_a already holds r.g_a incl. grad p; recover r_i:
                    // S.q stores conv and s only, so rebuild r_i = conv_i + gp_i via gp = sum_a P_a g_a
                    double gp[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        gp[j] = S.W[3] * S.g[j] + S.W[7] * S.g[3 + j] + S.W[11] * S.g[6 + j] + S.W[15] * S.g[9 + j];
                    const double r[3] = {r0 + gp[0], r1 + gp[1], r2 + gp[2]};
                    const double A1 = pa * ugb + nu * gab + tau * uga * ugb;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            acc[4 * i + j] += pa * pb * gum[3 * i + j]
                                + cu[j] * uga * r[i]                               // d tau
                                + tau * pb * ga[j] * r[i]                          // d (u.g_a)
                                + tau * uga * pb * gum[3 * i + j]                  // d r_i (phi_b gu_ij)
                                + ga[i] * cg[j];
                        acc[5 * i] += A1;
                        acc[4 * i + 3] += -pb * ga[i] + tau * uga * gb[i];       // d r_i / d p_b = g_b[i]
                        // continuity row: phi_a g_b[j] + d(tau r.g_a)
                        acc[12 + i] += pa * gb[i] + fv.pspg * (cu[i] * sa + tau * (ugb * ga[i] + pb * guga_a[i]));
                    }
                    acc[15] += fv.pspg * tau * gab;
                }
                if (b == 0) {                               // residual of node a
                    const double p = Q[3];
                    if (!corrected) {
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                            Rl[i] += Q[9 + i] * pa + nu * guga_a[i] - p * ga[i] + tau * u[i] * sa + nuL * divu * ga[i];
                    } else {
                        double gp[3];
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            gp[j] = S.W[3] * S.g[j] + S.W[7] * S.g[3 + j] + S.W[11] * S.g[6 + j] + S.W[15] * S.g[9 + j];
                        // nu (grad u):(grad v) needs (grad u) g_a regardless of the variant
                        const double vg[3] = {gum[0] * ga0 + gum[1] * ga1 + gum[2] * ga2,
                                              gum[3] * ga0 + gum[4] * ga1 + gum[5] * ga2,
                                              gum[6] * ga0 + gum[7] * ga1 + gum[8] * ga2};
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                            Rl[i] += Q[9 + i] * pa + nu * vg[i] - p * ga[i] + tau * uga * (Q[9 + i] + gp[i]) +
                                     nuL * divu * ga[i];
                    }
                    Rl[3] += pa * divu + fv.pspg * tau * sa;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] *= wd;
#pragma unroll
            for (int c = 0; c < 4; ++c) Rl[c] *= wd;
        }
    }

    // ---- element residual: Fe[a][c] = R[a][c] + sum_b block(a,b) * (g-w)_b   (lifting :65)
    // Stokes: R = sum_b block(a,b) * w_b (linear form), so the same reduction with (w + (g-w)).
    if (Fe) {
        double part[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double s = 0.0;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const double xv = (FORM == SNS_FORM_STOKES) ? (S.W[4 * b + d] + S.GW[4 * b + d]) : S.GW[4 * b + d];
                s += acc[4 * c + d] * xv;
            }
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            part[c] = s + Rl[c];
        }
        if (live && b == 0) {
            double* o = Fe + (4 * t + a) * 4;
            o[0] = part[0]; o[1] = part[1]; o[2] = part[2]; o[3] = part[3];
        }
    }

    // ---- transpose 256 blocks through LDS, store 1 KiB per wave instruction ----
    if (store_K) {
        __syncthreads();                                    // all reads of the staging data are done
#pragma unroll
        for (int e = 0; e < 16; ++e) tile[tid * 17 + e] = acc[e];
        __syncthreads();
        const int64_t base = (int64_t)blockIdx.x * (EL_TETS * 256);
        const int64_t lim = n_tets * 256;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = i * EL_TPB + tid;
            const int64_t gidx = base + idx;
            if (gidx < lim) Ke[gidx] = tile[(idx >> 4) * 17 + (idx & 15)];
        }
    }
}

#define SNS_INST_ELEMENT(F, C)                                                                              \
    template __global__ void k_element<F, C>(int64_t, const int32_t*, const double*, const double*,            \
                                             const uint8_t*, const double*, double, int, double*, double*, FormVariant);
SNS_INST_ELEMENT(SNS_FORM_STOKES, false)
SNS_INST_ELEMENT(SNS_FORM_NS, false)
SNS_INST_ELEMENT(SNS_FORM_NS, true)

// quad-permute a double with DPP moves (no LDS, no memory traffic); CTRL = quad_perm encoding
template <int CTRL>
__device__ __forceinline__ double quad_perm(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// ============================================================================
// K1 (scratch-free variant): every BSR block is produced by the lane(s) that own it.
// A lane walks the block's contribution list (tet, a, b) and recomputes the element
// block on the fly, so nothing is staged in HBM: no 2 KiB/tet scratch write, no gather pass,
// no atomics, fixed summation order (bitwise reproducible).  Costs ~2x the block flops of the
// staged kernel (geometry + per-point scalars are recomputed per contribution) and only applies
// when the state satisfies the Dirichlet data (no lifting term, :65), i.e. every Newton iterate
// after the first update; the staged k_element path handles the rest.
// ============================================================================
template <bool corrected>
__device__ __forceinline__ void tet_block_accumulate(const int4 tv, const double* __restrict__ pts,
                                                     const double* __restrict__ w, double nu, int a, int b,
                                                     bool want_res, double acc[16], double Ra[4]) {
    const int32_t nd[4] = {tv.x, tv.y, tv.z, tv.w};
    double X[4][3], W[4][4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const double* pp = pts + 3 * (int64_t)nd[v];
        X[v][0] = pp[0]; X[v][1] = pp[1]; X[v][2] = pp[2];
        const double2* wp = reinterpret_cast<const double2*>(w + 4 * (int64_t)nd[v]);
        const double2 w0 = wp[0], w1 = wp[1];
        W[v][0] = w0.x; W[v][1] = w0.y; W[v][2] = w1.x; W[v][3] = w1.y;
    }
    double J[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        J[i][0] = X[1][i] - X[0][i];
        J[i][1] = X[2][i] - X[0][i];
        J[i][2] = X[3][i] - X[0][i];
    }
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
    const double id = 1.0 / det;
    double K[3][3];
    K[0][0] = c00 * id; K[1][0] = c01 * id; K[2][0] = c02 * id;
    K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
    K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
    K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
    K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
    K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
    K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
    double g[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        g[1][j] = K[0][j]; g[2][j] = K[1][j]; g[3][j] = K[2][j];
        g[0][j] = -(K[0][j] + K[1][j] + K[2][j]);
    }
    double G[3][3], trG = 0.0, GG = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            G[i][j] = K[0][i] * K[0][j] + K[1][i] * K[1][j] + K[2][i] * K[2][j];
            GG += G[i][j] * G[i][j];
            if (i == j) trG += G[i][j];
        }
    double gu[3][3], gp[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        gp[j] = W[0][3] * g[0][j] + W[1][3] * g[1][j] + W[2][3] * g[2][j] + W[3][3] * g[3][j];
#pragma unroll
        for (int i = 0; i < 3; ++i) gu[i][j] = W[0][i] * g[0][j] + W[1][i] * g[1][j] + W[2][i] * g[2][j] + W[3][i] * g[3][j];
    }
    const double divu = gu[0][0] + gu[1][1] + gu[2][2];
    const double wd = fabs(det) * (1.0 / 24.0);
    const double itrG = 1.0 / trG, nu36GG = 36.0 * nu * nu * GG;
    // runtime-indexed rows of g: select with predication (keeps everything in registers)
    double ga[3], gb[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ga[j] = a == 0 ? g[0][j] : (a == 1 ? g[1][j] : (a == 2 ? g[2][j] : g[3][j]));
        gb[j] = b == 0 ? g[0][j] : (b == 1 ? g[1][j] : (b == 2 ? g[2][j] : g[3][j]));
    }
    const double gab = ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2];
    double guga[3], visc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        visc[j] = gu[j][0] * ga[0] + gu[j][1] * ga[1] + gu[j][2] * ga[2];                   // (grad u) g_a
        guga[j] = corrected ? (ga[0] * gu[0][j] + ga[1] * gu[1][j] + ga[2] * gu[2][j]) : visc[j];
    }
    // the quadrature weight wd is folded into the per-point coefficients, so every term lands directly in the
    // caller's accumulators (no per-contribution block)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double u[3] = {0.0, 0.0, 0.0}, p = 0.0;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const double ph = phi_q(q, v);
            u[0] += ph * W[v][0]; u[1] += ph * W[v][1]; u[2] += ph * W[v][2]; p += ph * W[v][3];
        }
        const double pa = phi_q(q, a), pb = phi_q(q, b);
        double Gu[3], conv[3], r[3], uGu = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            Gu[i] = G[i][0] * u[0] + G[i][1] * u[1] + G[i][2] * u[2];
            uGu += u[i] * Gu[i];
            conv[i] = gu[i][0] * u[0] + gu[i][1] * u[1] + gu[i][2] * u[2];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
            r[j] = (corrected ? conv[j] : (gu[0][j] * u[0] + gu[1][j] * u[1] + gu[2][j] * u[2])) + gp[j];
        // tau = m^-1/2, nu_LSIC = 1/(trG tau) = m tau / trG: one rsqrt per point, no divisions
        const double mq = uGu + nu36GG;
        const double tau = rsqrt(mq);
        const double nuL = mq * tau * itrG;
        const double sa = r[0] * ga[0] + r[1] * ga[1] + r[2] * ga[2];
        const double ugb = u[0] * gb[0] + u[1] * gb[1] + u[2] * gb[2];
        const double uga = u[0] * ga[0] + u[1] * ga[1] + u[2] * ga[2];
        const double tw = wd * tau;                            // weighted tau
        const double t3w = tw * tau * tau;
        const double wpb = wd * pb, wpa = wd * pa;
        double cu[3], cg[3];                                   // both carry the weight
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double dtau = -t3w * pb * Gu[j];
            const double dnuL = (tw * itrG) * pb * Gu[j];
            cg[j] = dnuL * divu + (wd * nuL) * gb[j];
            cu[j] = corrected ? dtau : dtau * sa + tw * (u[j] * gab + pb * guga[j]);
        }
        if (!corrected) {
            const double A1 = wpa * ugb + (wd * nu) * gab + tw * sa * pb;
            const double ppw = wpa * pb, tgw = tw * gab;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[4 * i + j] += ppw * gu[i][j] + u[i] * cu[j] + ga[i] * cg[j];
                acc[5 * i] += A1;
                acc[4 * i + 3] += tgw * u[i] - wpb * ga[i];
                acc[12 + i] += wpa * gb[i] + cu[i];
            }
            acc[15] += tgw;
        } else {
            const double A1 = wpa * ugb + (wd * nu) * gab + tw * uga * ugb;
            const double cgu = wpa * pb + tw * uga * pb;       // coefficient of gu[i][j]
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[4 * i + j] += cgu * gu[i][j] + (cu[j] * uga + tw * pb * ga[j]) * r[i] + ga[i] * cg[j];
                acc[5 * i] += A1;
                acc[4 * i + 3] += tw * uga * gb[i] - wpb * ga[i];
                acc[12 + i] += wpa * gb[i] + cu[i] * sa + tw * (ugb * ga[i] + pb * guga[i]);
            }
            acc[15] += tw * gab;
        }
        if (want_res) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
                Ra[i] += wpa * conv[i] + (wd * nu) * visc[i] - (wd * p) * ga[i] +
                         (corrected ? tw * uga * r[i] : tw * u[i] * sa) + (wd * nuL) * divu * ga[i];
            Ra[3] += wpa * divu + tw * sa;
        }
    }
}

// Stokes form (:160-172): block (a,b) of the constant element matrix and, for the diagonal kernel, row a of
// A0 w (w = the Dirichlet data extended by zero: the lifting term of the right-hand side, :65 with x = 0).
__device__ __forceinline__ void tet_block_accumulate_stokes(const int4 tv, const double* __restrict__ pts,
                                                            const double* __restrict__ w, int a, int b, bool want_res,
                                                            double acc[16], double Ra[4]) {
    const int32_t nd[4] = {tv.x, tv.y, tv.z, tv.w};
    double X[4][3];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const double* pp = pts + 3 * (int64_t)nd[v];
        X[v][0] = pp[0]; X[v][1] = pp[1]; X[v][2] = pp[2];
    }
    double J[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        J[i][0] = X[1][i] - X[0][i];
        J[i][1] = X[2][i] - X[0][i];
        J[i][2] = X[3][i] - X[0][i];
    }
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
    const double id = 1.0 / det;
    double K[3][3];
    K[0][0] = c00 * id; K[1][0] = c01 * id; K[2][0] = c02 * id;
    K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
    K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
    K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
    K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
    K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
    K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
    double g[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        g[1][j] = K[0][j]; g[2][j] = K[1][j]; g[3][j] = K[2][j];
        g[0][j] = -(K[0][j] + K[1][j] + K[2][j]);
    }
    double h2 = 0.0;                                   // CellDiameter^2 (:168): longest edge
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int u = v + 1; u < 4; ++u) {
            const double d0 = X[u][0] - X[v][0], d1 = X[u][1] - X[v][1], d2 = X[u][2] - X[v][2];
            h2 = fmax(h2, d0 * d0 + d1 * d1 + d2 * d2);
        }
    const double wd = fabs(det) * (1.0 / 24.0), vol = 4.0 * wd, muT = 0.2 * h2;       // :169
    double ga[3], gb[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ga[j] = a == 0 ? g[0][j] : (a == 1 ? g[1][j] : (a == 2 ? g[2][j] : g[3][j]));
        gb[j] = b == 0 ? g[0][j] : (b == 1 ? g[1][j] : (b == 2 ? g[2][j] : g[3][j]));
    }
    const double gab = ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        acc[5 * i] += vol * gab;                        // (grad u, grad v)
        acc[4 * i + 3] += -wd * ga[i];                  // -(p, div v)
        acc[12 + i] += wd * gb[i];                      // +(div u, q)
    }
    acc[15] += muT * vol * gab;                         // mu_T (grad p, grad q)
    if (want_res) {
        double gu[3][3], gp[3], psum = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            gp[i] = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) gu[i][j] = 0.0;
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const double* wv = w + 4 * (int64_t)nd[v];
            psum += wv[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                gp[j] += wv[3] * g[v][j];
#pragma unroll
                for (int i = 0; i < 3; ++i) gu[i][j] += wv[i] * g[v][j];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
            Ra[i] += vol * (gu[i][0] * ga[0] + gu[i][1] * ga[1] + gu[i][2] * ga[2]) - wd * ga[i] * psum;
        Ra[3] += wd * (gu[0][0] + gu[1][1] + gu[2][2]) + muT * vol * (gp[0] * ga[0] + gp[1] * ga[1] + gp[2] * ga[2]);
    }
}

// ============================================================================
// 2-D triangle P1-P1 variants (handle created by sns_create_2d).  Same node-blocked layout [ux,uy,uz,p]
// (uz is a homogeneous Dirichlet dof: identity row), tets[] holds 3 vertex ids per cell in a stride of 4.
//   * Stokes   nu_s (grad u, grad v) - (p, div v) + (div u, q) + beta h^2 (grad p, grad q)
//       DFG_2D_Validation.py:107-117 (nu_s = 1, beta = 0.2), LidDrivenNavierStokesFlow.py:96-109 (nu, 1/(12 nu))
//   * NS with the h-based Tezduyar UGN parameters, LidDrivenNavierStokesFlow.py:123-143 ==
//       DFG_2D_Validation.py:141-163:  tau_SUPG = (inv1 + (4 nu / h^2)^2)^-1/2, inv1 = |u| <= 1e-8 ? 0 : (2|u|/h)^2,
//       tau_LSIC = h/2 |u| z, z = Re_UGN <= 3 ? Re_UGN/3 : 1, Re_UGN = |u| h / (2 nu); convection and SUPG test
//       function are the consistent (u.grad)(.) there (dot(u, nabla_grad(.))), and the exact Gateaux derivative
//       differentiates both taus (each branch of the conditionals separately, as ufl.derivative does).
// dx(degree 2) on a triangle: 3-point rule (1/6,1/6), (1/6,2/3), (2/3,1/6), weights 1/6.
// ============================================================================
#define T13 0.33333333333333333
#define T16 0.16666666666666666
#define T23 0.66666666666666663

// phi_v at quadrature point q: the vertex that carries 2/3 is 0, 2, 1 for q = 0, 1, 2
__device__ __forceinline__ double phi_q2(int q, int v) { return v == ((3 - q) % 3) ? T23 : T16; }

struct TriGeom {
    double g[3][2];     // grad phi_v
    double wd;          // |det J| / 6  (= quadrature weight x |det J|)
    double h2;          // CellDiameter^2
};
__device__ __forceinline__ void tri_geometry(const int4 tv, const double* __restrict__ pts, TriGeom& T) {
    const int32_t nd[3] = {tv.x, tv.y, tv.z};
    double X[3][2];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const double* pp = pts + 3 * (int64_t)nd[v];
        X[v][0] = pp[0]; X[v][1] = pp[1];
    }
    const double J00 = X[1][0] - X[0][0], J01 = X[2][0] - X[0][0];
    const double J10 = X[1][1] - X[0][1], J11 = X[2][1] - X[0][1];
    const double det = J00 * J11 - J01 * J10;
    const double id = 1.0 / det;
    // K = J^-1, K[k][j] = dX_k / dx_j ; grad phi_1 = K[0][:], grad phi_2 = K[1][:]
    T.g[1][0] = J11 * id;  T.g[1][1] = -J01 * id;
    T.g[2][0] = -J10 * id; T.g[2][1] = J00 * id;
    T.g[0][0] = -(T.g[1][0] + T.g[2][0]);
    T.g[0][1] = -(T.g[1][1] + T.g[2][1]);
    T.wd = fabs(det) * T16;
    double h2 = 0.0;
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int u = v + 1; u < 3; ++u) {
            const double d0 = X[u][0] - X[v][0], d1 = X[u][1] - X[v][1];
            h2 = fmax(h2, d0 * d0 + d1 * d1);
        }
    T.h2 = h2;
}
__device__ __forceinline__ void tri_state(const int4 tv, const double* __restrict__ w, double W[3][3]) {
    const int32_t nd[3] = {tv.x, tv.y, tv.z};
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const double2* wp = reinterpret_cast<const double2*>(w + 4 * (int64_t)nd[v]);
        const double2 w0 = wp[0], w1 = wp[1];
        W[v][0] = w0.x; W[v][1] = w0.y; W[v][2] = w1.y;       // ux, uy, p
    }
}

// per-point UGN scalars at velocity u (h, h2 of the cell): tau_SUPG, tau_LSIC and the coefficients c with
// d tau = c_tau (u . du), d tau_LSIC = c_L (u . du)
__device__ __forceinline__ void ugn_taus(const double u[2], double h, double h2, double nu, double& tau, double& ctau,
                                         double& tauL, double& cL) {
    const double uu = u[0] * u[0] + u[1] * u[1];
    const double un = sqrt(uu);
    const bool slow = un <= 1e-8;                                  // conditional(le(u_norm, 1e-8), 0, .)
    const double c4 = 4.0 / h2;
    const double i3 = 4.0 * nu / h2;                               // 1 / tau_SUNG3
    const double m = (slow ? 0.0 : c4 * uu) + i3 * i3;
    tau = rsqrt(m);
    ctau = slow ? 0.0 : -tau * tau * tau * c4;
    const double ReU = un * h / (2.0 * nu);
    if (ReU <= 3.0) {                                              // z = Re_UGN / 3
        tauL = 0.5 * h * un * (ReU * T13);
        cL = h2 / (6.0 * nu);
    } else {                                                       // z = 1
        tauL = 0.5 * h * un;
        cL = 0.5 * h / un;
    }
}

__device__ __forceinline__ void tri_block_accumulate_ugn(const int4 tv, const double* __restrict__ pts,
                                                         const double* __restrict__ w, double nu, int a, int b,
                                                         bool want_res, double acc[16], double Ra[4]) {
    TriGeom T;
    tri_geometry(tv, pts, T);
    double W[3][3];
    tri_state(tv, w, W);
    double gu[2][2], gp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        gp[j] = W[0][2] * T.g[0][j] + W[1][2] * T.g[1][j] + W[2][2] * T.g[2][j];
#pragma unroll
        for (int i = 0; i < 2; ++i) gu[i][j] = W[0][i] * T.g[0][j] + W[1][i] * T.g[1][j] + W[2][i] * T.g[2][j];
    }
    const double divu = gu[0][0] + gu[1][1];
    const double wd = T.wd, h2 = T.h2, h = sqrt(h2);
    double ga[2], gb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        ga[j] = a == 0 ? T.g[0][j] : (a == 1 ? T.g[1][j] : T.g[2][j]);
        gb[j] = b == 0 ? T.g[0][j] : (b == 1 ? T.g[1][j] : T.g[2][j]);
    }
    const double gab = ga[0] * gb[0] + ga[1] * gb[1];
    const double visc[2] = {gu[0][0] * ga[0] + gu[0][1] * ga[1], gu[1][0] * ga[0] + gu[1][1] * ga[1]};   // (grad u) g_a
    const double guga[2] = {ga[0] * gu[0][0] + ga[1] * gu[1][0], ga[0] * gu[0][1] + ga[1] * gu[1][1]};   // g_a^T (grad u)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        double u[2] = {0.0, 0.0}, p = 0.0;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const double ph = phi_q2(q, v);
            u[0] += ph * W[v][0]; u[1] += ph * W[v][1]; p += ph * W[v][2];
        }
        const double pa = phi_q2(q, a), pb = phi_q2(q, b);
        double tau, ctau, tauL, cL;
        ugn_taus(u, h, h2, nu, tau, ctau, tauL, cL);
        const double conv[2] = {gu[0][0] * u[0] + gu[0][1] * u[1], gu[1][0] * u[0] + gu[1][1] * u[1]};   // (u.grad)u
        const double r[2] = {conv[0] + gp[0], conv[1] + gp[1]};                                         // res
        const double sa = r[0] * ga[0] + r[1] * ga[1];
        const double ugb = u[0] * gb[0] + u[1] * gb[1];
        const double uga = u[0] * ga[0] + u[1] * ga[1];
        const double tw = wd * tau, wpa = wd * pa, wpb = wd * pb;
        double cu[2], cg[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            cu[j] = wd * ctau * pb * u[j];                                     // weighted d tau / d u_(b,j)
            cg[j] = wd * (cL * pb * u[j] * divu + tauL * gb[j]);               // weighted d (tau_LSIC div u)
        }
        const double A1 = wpa * ugb + (wd * nu) * gab + tw * uga * ugb;
        const double cgu = wpa * pb + tw * uga * pb;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[4 * i + j] += cgu * gu[i][j] + (cu[j] * uga + tw * pb * ga[j]) * r[i] + ga[i] * cg[j];
            acc[5 * i] += A1;
            acc[4 * i + 3] += tw * uga * gb[i] - wpb * ga[i];
            acc[12 + i] += wpa * gb[i] + cu[i] * sa + tw * (ugb * ga[i] + pb * guga[i]);
        }
        acc[15] += tw * gab;
        if (want_res) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                Ra[i] += wpa * conv[i] + (wd * nu) * visc[i] - (wd * p) * ga[i] + tw * uga * r[i] +
                         (wd * tauL) * divu * ga[i];
            Ra[3] += wpa * divu + tw * sa;
        }
    }
}

__device__ __forceinline__ void tri_block_accumulate_stokes(const int4 tv, const double* __restrict__ pts,
                                                            const double* __restrict__ w, double nu_s, double beta,
                                                            int a, int b, bool want_res, double acc[16], double Ra[4]) {
    TriGeom T;
    tri_geometry(tv, pts, T);
    const double wd = T.wd, vol = 3.0 * wd, muT = beta * T.h2;
    double ga[2], gb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        ga[j] = a == 0 ? T.g[0][j] : (a == 1 ? T.g[1][j] : T.g[2][j]);
        gb[j] = b == 0 ? T.g[0][j] : (b == 1 ? T.g[1][j] : T.g[2][j]);
    }
    const double gab = ga[0] * gb[0] + ga[1] * gb[1];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        acc[5 * i] += nu_s * vol * gab;                 // nu_s (grad u, grad v)
        acc[4 * i + 3] += -wd * ga[i];                  // -(p, div v), int phi_b = vol / 3 = wd
        acc[12 + i] += wd * gb[i];                      // +(div u, q)
    }
    acc[15] += muT * vol * gab;
    if (want_res) {
        double W[3][3];
        tri_state(tv, w, W);
        double gu[2][2], gp[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            gp[j] = W[0][2] * T.g[0][j] + W[1][2] * T.g[1][j] + W[2][2] * T.g[2][j];
#pragma unroll
            for (int i = 0; i < 2; ++i) gu[i][j] = W[0][i] * T.g[0][j] + W[1][i] * T.g[1][j] + W[2][i] * T.g[2][j];
        }
        const double psum = W[0][2] + W[1][2] + W[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) Ra[i] += nu_s * vol * (gu[i][0] * ga[0] + gu[i][1] * ga[1]) - wd * ga[i] * psum;
        Ra[3] += wd * (gu[0][0] + gu[1][1]) + muT * vol * (gp[0] * ga[0] + gp[1] * ga[1]);
    }
}

// one interface for the four forms (aux: Stokes 2-D pressure-stabilisation coefficient beta; nu: 1/Re, or the
// Stokes 2-D viscosity)
template <int FORM, bool corrected>
__device__ __forceinline__ void block_accumulate(const int4 tv, const double* __restrict__ pts,
                                                 const double* __restrict__ w, double nu, double aux, int a, int b,
                                                 bool want_res, double acc[16], double Ra[4]) {
    if constexpr (FORM == SNS_FORM_STOKES) tet_block_accumulate_stokes(tv, pts, w, a, b, want_res, acc, Ra);
    else if constexpr (FORM == SNS_FORM_NS) tet_block_accumulate<corrected>(tv, pts, w, nu, a, b, want_res, acc, Ra);
    else if constexpr (FORM == SNS_FORM_STOKES_2D) tri_block_accumulate_stokes(tv, pts, w, nu, aux, a, b, want_res, acc, Ra);
    else tri_block_accumulate_ugn(tv, pts, w, nu, a, b, want_res, acc, Ra);
}
constexpr bool form_is_linear(int form) { return form == SNS_FORM_STOKES || form == SNS_FORM_STOKES_2D; }

// off-diagonal BSR blocks: one lane per slot, slots taken from the host's count-sorted list
template <int FORM, bool corrected>
__global__ __launch_bounds__(256) void k_fused_offdiag(int64_t n_od, const int32_t* __restrict__ od_order,
                                                       const int64_t* __restrict__ c_ptr,
                                                       const int32_t* __restrict__ c_idx,
                                                       const int32_t* __restrict__ slot_row,
                                                       const int32_t* __restrict__ colind,
                                                       const int32_t* __restrict__ tets,
                                                       const double* __restrict__ pts, const double* __restrict__ w,
                                                       const uint8_t* __restrict__ bc_mask, double nu,
                                                       double aux, double* __restrict__ vals) {
    const int64_t lane = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= n_od) return;
    const int64_t s = od_order[lane];
    const int32_t row = slot_row[s], col = colind[s];
    double acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0;
    // the (contribution id -> tet nodes) loads of the NEXT contribution are issued before the current block is
    // computed, so only the coordinate/state gather latency sits on the critical path
    const int64_t k1 = c_ptr[s + 1];
    int64_t k = c_ptr[s];
    uint32_t id = 0;                                 // tet*16 + a*4 + b, unsigned (up to 268 M tets)
    int4 tv = make_int4(0, 0, 0, 0);
    if (k < k1) {
        id = (uint32_t)c_idx[k];
        tv = *reinterpret_cast<const int4*>(tets + 4 * (int64_t)(id >> 4));
    }
    while (k < k1) {
        const uint32_t idc = id;
        const int4 tvc = tv;
        if (++k < k1) {
            id = (uint32_t)c_idx[k];
            tv = *reinterpret_cast<const int4*>(tets + 4 * (int64_t)(id >> 4));
        }
        block_accumulate<FORM, corrected>(tvc, pts, w, nu, aux, (idc >> 2) & 3, idc & 3, false, acc, nullptr);
    }
    const uchar4 mr = *reinterpret_cast<const uchar4*>(bc_mask + 4 * (int64_t)row);
    const uchar4 mc = *reinterpret_cast<const uchar4*>(bc_mask + 4 * (int64_t)col);
    const unsigned char rb[4] = {mr.x, mr.y, mr.z, mr.w}, cb[4] = {mc.x, mc.y, mc.z, mc.w};
    double2* o = reinterpret_cast<double2*>(vals + 16 * s);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double v0 = (rb[c] | cb[0]) ? 0.0 : acc[4 * c], v1 = (rb[c] | cb[1]) ? 0.0 : acc[4 * c + 1];
        const double v2 = (rb[c] | cb[2]) ? 0.0 : acc[4 * c + 2], v3 = (rb[c] | cb[3]) ? 0.0 : acc[4 * c + 3];
        o[2 * c] = make_double2(v0, v1);
        o[2 * c + 1] = make_double2(v2, v3);
    }
}

// diagonal blocks + node residuals: 4 lanes per node share the ~24 incident tets, DPP quad sums in a fixed order
template <int FORM, bool corrected>
__global__ __launch_bounds__(256) void k_fused_diag(int32_t n_rows, const int32_t* __restrict__ diag,
                                                    const int64_t* __restrict__ c_ptr,
                                                    const int32_t* __restrict__ c_idx,
                                                    const int32_t* __restrict__ tets, const double* __restrict__ pts,
                                                    const double* __restrict__ w, const uint8_t* __restrict__ bc_mask,
                                                    const double* __restrict__ bc_val, double nu, double aux,
                                                    double* __restrict__ vals, double* __restrict__ F) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t node = gid >> 2;
    const int q = (int)(gid & 3);
    const bool live = node < n_rows;
    double acc[16], R[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0;
    int64_t s = 0;
    if (live) {
        s = diag[node];
        const int64_t k1 = c_ptr[s + 1];
        int64_t k = c_ptr[s] + q;
        uint32_t id = 0;
        int4 tv = make_int4(0, 0, 0, 0);
        if (k < k1) {
            id = (uint32_t)c_idx[k];
            tv = *reinterpret_cast<const int4*>(tets + 4 * (int64_t)(id >> 4));
        }
        while (k < k1) {
            const int a = (id >> 2) & 3;
            const int4 tvc = tv;
            k += 4;
            if (k < k1) {
                id = (uint32_t)c_idx[k];
                tv = *reinterpret_cast<const int4*>(tets + 4 * (int64_t)(id >> 4));
            }
            block_accumulate<FORM, corrected>(tvc, pts, w, nu, aux, a, a, true, acc, R);
        }
    }
    // quad sums: (l0 + l1) + (l2 + l3), identical on every lane
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        double v = acc[e];
        v += quad_perm<0xB1>(v);          // swap neighbours: [1,0,3,2]
        v += quad_perm<0x4E>(v);          // swap pairs:      [2,3,0,1]
        acc[e] = v;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double v = R[c];
        v += quad_perm<0xB1>(v);
        v += quad_perm<0x4E>(v);
        R[c] = v;
    }
    if (!live) return;
    const uchar4 m4 = *reinterpret_cast<const uchar4*>(bc_mask + 4 * node);
    const unsigned char mb[4] = {m4.x, m4.y, m4.z, m4.w};
    // lane q writes row q of the block and component q of the residual
    double row[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const double v = q == 0 ? acc[d] : (q == 1 ? acc[4 + d] : (q == 2 ? acc[8 + d] : acc[12 + d]));
        const unsigned char rbq = q == 0 ? mb[0] : (q == 1 ? mb[1] : (q == 2 ? mb[2] : mb[3]));
        row[d] = (rbq | mb[d]) ? ((q == d) ? 1.0 : 0.0) : v;
    }
    if (vals) {
        double2* o = reinterpret_cast<double2*>(vals + 16 * s + 4 * q);
        o[0] = make_double2(row[0], row[1]);
        o[1] = make_double2(row[2], row[3]);
    }
    if (F) {
        const int64_t dof = 4 * node + q;
        const double rq = q == 0 ? R[0] : (q == 1 ? R[1] : (q == 2 ? R[2] : R[3]));
        F[dof] = bc_mask[dof] ? ((form_is_linear(FORM) ? 0.0 : w[dof]) - bc_val[dof]) : rq;
    }
}
// Lifting term of a state that violates its Dirichlet data (:65): F_free += A0[:,B] (g - x_B), A0 = the unconstrained
// Jacobian.  Same work split as k_fused_diag (4 lanes per node, DPP quad sums); only tets with a violated Dirichlet
// dof (dl != 0 on one of their nodes) cost anything: their blocks (a,b) are recomputed and applied to dl_b.
template <int FORM, bool corrected>
__global__ __launch_bounds__(256) void k_fused_lift(int32_t n_rows, const int32_t* __restrict__ diag,
                                                    const int64_t* __restrict__ c_ptr,
                                                    const int32_t* __restrict__ c_idx,
                                                    const int32_t* __restrict__ tets, const double* __restrict__ pts,
                                                    const double* __restrict__ w, const uint8_t* __restrict__ bc_mask,
                                                    const double* __restrict__ dl, double nu, double* __restrict__ F) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t node = gid >> 2;
    const int q = (int)(gid & 3);
    const bool live = node < n_rows;
    double R[4] = {0.0, 0.0, 0.0, 0.0};
    if (live) {
        const int64_t s = diag[node];
        for (int64_t k = c_ptr[s] + q; k < c_ptr[s + 1]; k += 4) {
            const uint32_t id = (uint32_t)c_idx[k];
            const int a = (id >> 2) & 3;
            const int4 tv = *reinterpret_cast<const int4*>(tets + 4 * (int64_t)(id >> 4));
            const int32_t nd[4] = {tv.x, tv.y, tv.z, tv.w};
            constexpr int NPE = (FORM == SNS_FORM_UGN_2D || FORM == SNS_FORM_STOKES_2D) ? 3 : 4;
#pragma unroll 1
            for (int b = 0; b < NPE; ++b) {
                const double2* dp = reinterpret_cast<const double2*>(dl + 4 * (int64_t)nd[b]);
                const double2 d01 = dp[0], d23 = dp[1];
                if (d01.x == 0.0 && d01.y == 0.0 && d23.x == 0.0 && d23.y == 0.0) continue;
                double blk[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) blk[e] = 0.0;
                block_accumulate<FORM, corrected>(tv, pts, w, nu, 0.0, a, b, false, blk, nullptr);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    R[c] += blk[4 * c] * d01.x + blk[4 * c + 1] * d01.y + blk[4 * c + 2] * d23.x + blk[4 * c + 3] * d23.y;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double v = R[c];
        v += quad_perm<0xB1>(v);
        v += quad_perm<0x4E>(v);
        R[c] = v;
    }
    if (!live) return;
    const int64_t dof = 4 * node + q;
    const double rq = q == 0 ? R[0] : (q == 1 ? R[1] : (q == 2 ? R[2] : R[3]));
    if (!bc_mask[dof]) F[dof] += rq;
}
#define SNS_INST_LIFT(FM, C)                                                                                        \
    template __global__ void k_fused_lift<FM, C>(int32_t, const int32_t*, const int64_t*, const int32_t*, const int32_t*, \
                                                 const double*, const double*, const uint8_t*, const double*, double, double*);
SNS_INST_LIFT(SNS_FORM_NS, false)
SNS_INST_LIFT(SNS_FORM_NS, true)
SNS_INST_LIFT(SNS_FORM_UGN_2D, false)

// dl = g - w on Dirichlet dofs, 0 elsewhere
__global__ __launch_bounds__(256) void k_bc_defect(int64_t ndof, const uint8_t* __restrict__ bc_mask,
                                                   const double* __restrict__ bc_val, const double* __restrict__ w,
                                                   double* __restrict__ dl) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndof; i += (int64_t)gridDim.x * blockDim.x)
        dl[i] = bc_mask[i] ? (bc_val[i] - w[i]) : 0.0;
}

#define SNS_INST_FUSED(FM, C)                                                                                      \
    template __global__ void k_fused_offdiag<FM, C>(int64_t, const int32_t*, const int64_t*, const int32_t*, const int32_t*, \
                                                const int32_t*, const int32_t*, const double*, const double*,       \
                                                const uint8_t*, double, double, double*);                           \
    template __global__ void k_fused_diag<FM, C>(int32_t, const int32_t*, const int64_t*, const int32_t*,               \
                                             const int32_t*, const double*, const double*, const uint8_t*,          \
                                             const double*, double, double, double*, double*);
SNS_INST_FUSED(SNS_FORM_NS, false)
SNS_INST_FUSED(SNS_FORM_NS, true)
SNS_INST_FUSED(SNS_FORM_STOKES, false)
SNS_INST_FUSED(SNS_FORM_STOKES_2D, false)
SNS_INST_FUSED(SNS_FORM_UGN_2D, false)

// residual-only pass of the 2-D UGN form: one lane per triangle, Fe[16 t + 4 a + c] (same layout as the tet kernel,
// so k_gather_residual serves both)
__global__ __launch_bounds__(256) void k_residual_tri(int64_t n_tris, const int32_t* __restrict__ tets,
                                                      const double* __restrict__ pts, const double* __restrict__ w,
                                                      double nu, double* __restrict__ Fe) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tris) return;
    const int4 tv = *reinterpret_cast<const int4*>(tets + 4 * t);
    TriGeom T;
    tri_geometry(tv, pts, T);
    double W[3][3];
    tri_state(tv, w, W);
    double gu[2][2], gp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        gp[j] = W[0][2] * T.g[0][j] + W[1][2] * T.g[1][j] + W[2][2] * T.g[2][j];
#pragma unroll
        for (int i = 0; i < 2; ++i) gu[i][j] = W[0][i] * T.g[0][j] + W[1][i] * T.g[1][j] + W[2][i] * T.g[2][j];
    }
    const double divu = gu[0][0] + gu[1][1];
    const double wd = T.wd, h2 = T.h2, h = sqrt(h2);
    double R[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) R[a][c] = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        double u[2] = {0.0, 0.0}, p = 0.0;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const double ph = phi_q2(q, v);
            u[0] += ph * W[v][0]; u[1] += ph * W[v][1]; p += ph * W[v][2];
        }
        double tau, ctau, tauL, cL;
        ugn_taus(u, h, h2, nu, tau, ctau, tauL, cL);
        const double conv[2] = {gu[0][0] * u[0] + gu[0][1] * u[1], gu[1][0] * u[0] + gu[1][1] * u[1]};
        const double r[2] = {conv[0] + gp[0], conv[1] + gp[1]};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double pa = phi_q2(q, a);
            const double sa = r[0] * T.g[a][0] + r[1] * T.g[a][1];
            const double uga = u[0] * T.g[a][0] + u[1] * T.g[a][1];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const double visc = gu[i][0] * T.g[a][0] + gu[i][1] * T.g[a][1];
                R[a][i] += conv[i] * pa + nu * visc - p * T.g[a][i] + tau * uga * r[i] + tauL * divu * T.g[a][i];
            }
            R[a][2] += pa * divu + tau * sa;
        }
    }
    double2* o = reinterpret_cast<double2*>(Fe + 16 * t);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        o[2 * a] = make_double2(wd * R[a][0], wd * R[a][1]);
        o[2 * a + 1] = make_double2(0.0, wd * R[a][2]);
    }
    o[6] = make_double2(0.0, 0.0);
    o[7] = make_double2(0.0, 0.0);
}

// F_B = w_B - g on Dirichlet dofs (set_bc(F, bc, x, -1)); other entries untouched
__global__ __launch_bounds__(256) void k_bc_residual(int64_t ndof, const uint8_t* __restrict__ bc_mask,
                                                     const double* __restrict__ bc_val, const double* __restrict__ w,
                                                     double* __restrict__ F) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndof; i += (int64_t)gridDim.x * blockDim.x)
        if (bc_mask[i]) F[i] = w[i] - bc_val[i];
}

// Residual-only element pass for states that already satisfy the Dirichlet data (no lifting term):
// ONE LANE PER TET, every lane busy (the fused kernel keeps 12 of 16 lanes idle in its per-point
// phase).  Used by the line search (F(x - lambda y), :51-67 without the Jacobian).
template <bool corrected>
__global__ __launch_bounds__(256) void k_residual_tet(int64_t n_tets, const int32_t* __restrict__ tets,
                                                      const double* __restrict__ pts,
                                                      const double* __restrict__ w, double nu,
                                                      double* __restrict__ Fe) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tets) return;
    const int4 tv = *reinterpret_cast<const int4*>(tets + 4 * t);
    const int32_t nd[4] = {tv.x, tv.y, tv.z, tv.w};
    double X[4][3], W[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const double* pp = pts + 3 * (int64_t)nd[a];
        X[a][0] = pp[0]; X[a][1] = pp[1]; X[a][2] = pp[2];
        const double2* wp = reinterpret_cast<const double2*>(w + 4 * (int64_t)nd[a]);
        const double2 w0 = wp[0], w1 = wp[1];
        W[a][0] = w0.x; W[a][1] = w0.y; W[a][2] = w1.x; W[a][3] = w1.y;
    }
    double J[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        J[i][0] = X[1][i] - X[0][i];
        J[i][1] = X[2][i] - X[0][i];
        J[i][2] = X[3][i] - X[0][i];
    }
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
    const double id = 1.0 / det;
    double K[3][3];
    K[0][0] = c00 * id; K[1][0] = c01 * id; K[2][0] = c02 * id;
    K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
    K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
    K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
    K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
    K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
    K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
    double g[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        g[1][j] = K[0][j]; g[2][j] = K[1][j]; g[3][j] = K[2][j];
        g[0][j] = -(K[0][j] + K[1][j] + K[2][j]);
    }
    double G[3][3], trG = 0.0, GG = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            G[i][j] = K[0][i] * K[0][j] + K[1][i] * K[1][j] + K[2][i] * K[2][j];
            GG += G[i][j] * G[i][j];
            if (i == j) trG += G[i][j];
        }
    double gu[3][3], gp[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        gp[j] = W[0][3] * g[0][j] + W[1][3] * g[1][j] + W[2][3] * g[2][j] + W[3][3] * g[3][j];
#pragma unroll
        for (int i = 0; i < 3; ++i) gu[i][j] = W[0][i] * g[0][j] + W[1][i] * g[1][j] + W[2][i] * g[2][j] + W[3][i] * g[3][j];
    }
    const double divu = gu[0][0] + gu[1][1] + gu[2][2];
    const double wd = fabs(det) * (1.0 / 24.0);
    double R[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) R[a][c] = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double u[3] = {0.0, 0.0, 0.0}, p = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double ph = phi_q(q, a);
            u[0] += ph * W[a][0]; u[1] += ph * W[a][1]; u[2] += ph * W[a][2]; p += ph * W[a][3];
        }
        double Gu[3], conv[3], r[3], uGu = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            Gu[i] = G[i][0] * u[0] + G[i][1] * u[1] + G[i][2] * u[2];
            uGu += u[i] * Gu[i];
            conv[i] = gu[i][0] * u[0] + gu[i][1] * u[1] + gu[i][2] * u[2];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
            r[j] = (corrected ? conv[j] : (gu[0][j] * u[0] + gu[1][j] * u[1] + gu[2][j] * u[2])) + gp[j];
        const double tau = 1.0 / sqrt(uGu + 36.0 * nu * nu * GG);
        const double nuL = 1.0 / (trG * tau);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double pa = phi_q(q, a);
            const double sa = r[0] * g[a][0] + r[1] * g[a][1] + r[2] * g[a][2];
            const double uga = u[0] * g[a][0] + u[1] * g[a][1] + u[2] * g[a][2];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double visc = gu[i][0] * g[a][0] + gu[i][1] * g[a][1] + gu[i][2] * g[a][2];
                const double supg = corrected ? tau * uga * r[i] : tau * u[i] * sa;
                R[a][i] += conv[i] * pa + nu * visc - p * g[a][i] + supg + nuL * divu * g[a][i];
            }
            R[a][3] += pa * divu + tau * sa;
        }
    }
    double2* o = reinterpret_cast<double2*>(Fe + 16 * t);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        o[2 * a] = make_double2(wd * R[a][0], wd * R[a][1]);
        o[2 * a + 1] = make_double2(wd * R[a][2], wd * R[a][3]);
    }
}
template __global__ void k_residual_tet<false>(int64_t, const int32_t*, const double*, const double*, double, double*);
template __global__ void k_residual_tet<true>(int64_t, const int32_t*, const double*, const double*, double, double*);

// BSR slot <- sum over its contributing element blocks (fixed order => bitwise
// reproducible), Dirichlet rows AND columns zeroed, unit diagonal (:74).
// 8 lanes per slot, lane = two adjacent entries (16-B loads): each contribution is one
// 128-B line; 4 independent accumulators keep 4 lines per group in flight.
__global__ __launch_bounds__(256) void k_gather_matrix(int64_t nnzb, const int64_t* __restrict__ c_ptr,
                                                       const int32_t* __restrict__ c_idx,
                                                       const int32_t* __restrict__ slot_row,
                                                       const int32_t* __restrict__ colind,
                                                       const uint8_t* __restrict__ bc_mask,
                                                       const double* __restrict__ Ke, double* __restrict__ vals) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t s = gid >> 3;
    const int e2 = (int)(gid & 7);                 // entries 2*e2, 2*e2+1 of the 4x4 block
    if (s >= nnzb) return;
    const int64_t k0 = c_ptr[s], k1 = c_ptr[s + 1];
    const double2* __restrict__ K2 = reinterpret_cast<const double2*>(Ke);
    double2 a0 = {0.0, 0.0}, a1 = {0.0, 0.0}, a2 = {0.0, 0.0}, a3 = {0.0, 0.0};
    int64_t k = k0;
    for (; k + 3 < k1; k += 4) {
        const uint32_t i0 = (uint32_t)c_idx[k], i1 = (uint32_t)c_idx[k + 1], i2 = (uint32_t)c_idx[k + 2],
                       i3 = (uint32_t)c_idx[k + 3];
        const double2 v0 = K2[(int64_t)i0 * 8 + e2], v1 = K2[(int64_t)i1 * 8 + e2];
        const double2 v2 = K2[(int64_t)i2 * 8 + e2], v3 = K2[(int64_t)i3 * 8 + e2];
        a0.x += v0.x; a0.y += v0.y; a1.x += v1.x; a1.y += v1.y;
        a2.x += v2.x; a2.y += v2.y; a3.x += v3.x; a3.y += v3.y;
    }
    for (; k < k1; ++k) {
        const double2 v0 = K2[(int64_t)(uint32_t)c_idx[k] * 8 + e2];
        a0.x += v0.x; a0.y += v0.y;
    }
    double vx = (a0.x + a1.x) + (a2.x + a3.x), vy = (a0.y + a1.y) + (a2.y + a3.y);
    const int32_t row = slot_row[s], col = colind[s];
    const int c = e2 >> 1, d0 = (e2 & 1) * 2;
    const bool rb = bc_mask[4 * (int64_t)row + c];
    if (rb | bc_mask[4 * (int64_t)col + d0]) vx = (row == col && c == d0) ? 1.0 : 0.0;
    if (rb | bc_mask[4 * (int64_t)col + d0 + 1]) vy = (row == col && c == d0 + 1) ? 1.0 : 0.0;
    reinterpret_cast<double2*>(vals)[s * 8 + e2] = make_double2(vx, vy);
}

// node residual <- sum of incident element residuals; F_B = w_B - g (:67)
__global__ __launch_bounds__(256) void k_gather_residual(int32_t n_rows, const int64_t* __restrict__ nt_ptr,
                                                         const int32_t* __restrict__ nt_idx,
                                                         const uint8_t* __restrict__ bc_mask,
                                                         const double* __restrict__ bc_val,
                                                         const double* __restrict__ w,
                                                         const double* __restrict__ Fe, double* __restrict__ F) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = gid >> 2;
    const int c = (int)(gid & 3);
    if (i >= n_rows) return;
    double s = 0.0;
    // ~24 incident tets per node: 8 ids first, then their 8 entries (two round trips per 8 instead of a dependent pair per
    // tet); summed in list order as before
    const int64_t k1 = nt_ptr[i + 1];
    int64_t k = nt_ptr[i];
    for (; k + 7 < k1; k += 8) {
        int32_t id[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) id[q] = nt_idx[k + q];
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = Fe[(int64_t)id[q] * 4 + c];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += v[q];
    }
    if (k < k1) {
        int32_t id[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) id[q] = (k + q < k1) ? nt_idx[k + q] : -1;
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = id[q] >= 0 ? Fe[(int64_t)id[q] * 4 + c] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (id[q] >= 0) s += v[q];
    }
    const int64_t dof = 4 * i + c;
    if (bc_mask[dof]) s = (w ? w[dof] : 0.0) - bc_val[dof];
    F[dof] = s;
}

// ============================================================================
// K2: BSR4 SpMV.  8 lanes per block row: lane t -> (r = t>>1, half = t&1) loads
// 16 B (two doubles of block row r) per block, so one 8-lane group streams a
// whole 128-B block per instruction and a wave streams 8 consecutive rows.
// Workgroups are remapped so that each XCD walks one contiguous eighth of the
// matrix (its private L2 then holds the x entries of ITS rows only).
// ============================================================================
// streaming (non-temporal) 16-B loads for data that is read exactly once per pass: keeps the matrix
// stream from evicting the x / b / D^-1 vectors out of L2 and the Infinity Cache
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef double f64x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld_stream(const float4* p) {
    const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ double2 ld_stream(const double2* p) {
    const f64x2_t v = __builtin_nontemporal_load(reinterpret_cast<const f64x2_t*>(p));
    return make_double2(v.x, v.y);
}

__device__ __forceinline__ int xcd_remap(int b, int nb) {
    const int q = nb >> 3, r = nb & 7;
    const int xcd = b & 7, k = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// FINE only tags the instantiation launched on the assembled (level-0) operator so that
// profiler summaries separate it from the small coarse-level launches.
// SPLIT (multi-GPU, level 0): 0 = every row; 1 = interior pass: rows flagged in `skip` (rows with a ghost column)
// are left alone, so the pass can run while the halo is still in flight; 2 = boundary pass over the n_rows rows
// listed in `row_list`, after the halo has arrived.  partial_off: first partial-sum slot of this launch (AX_DOT).
// SPLIT 3 (window transports, round 5): every row in ONE launch -- the ghost entries of x are read straight from the receive
// window (`gs`, sns_peer_dev.h) and a wave that meets a ghost column waits for the neighbours' arrival flags itself: no unpack
// kernel, no boundary launch, no second stream.  Rows without a ghost column never wait.
template <int MODE, int FINE, int NT, int SPLIT>
__global__ __launch_bounds__(256) void k_spmv(int32_t n_rows, const int32_t* __restrict__ rowptr,
                                              const int32_t* __restrict__ colind,
                                              const double* __restrict__ vals, const double* __restrict__ x,
                                              double* __restrict__ y, const double* __restrict__ bvec,
                                              const double* __restrict__ dinv, double omega,
                                              const double* __restrict__ dotw, double* __restrict__ partial,
                                              const int32_t* __restrict__ row_list,
                                              const uint8_t* __restrict__ skip, int partial_off, GhostSrc gs) {
    GhostReader gr;
    if (SPLIT == 3) gr.begin(gs);
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int t = lane & 7;
    const int r = t >> 1, hf = t & 1;
    const int32_t ridx = (blk * 4 + (tid >> 6)) * 8 + (lane >> 3);
    double acc0 = 0.0, acc1 = 0.0;
    bool live = ridx < n_rows;
    int32_t row = ridx;
    if (SPLIT == 2) row = live ? row_list[ridx] : 0;
    if (SPLIT == 1) live = live && !skip[ridx];
    // per-row operands of the epilogue are requested before the block loop (their latency hides behind it)
    double pre_v = 0.0, pre_x = 0.0;
    if (live && hf == 0) {
        if (MODE == SPMV_B_MINUS_AX || MODE == SPMV_JACOBI) pre_v = bvec[4 * (int64_t)row + r];
        if (MODE == SPMV_AX_DOT) pre_v = dotw[4 * (int64_t)row + r];
        if (MODE == SPMV_JACOBI) pre_x = x[4 * (int64_t)row + r];
    }
    {
        // The kernel is bound by VMEM instruction issue, not by bytes, so every step of 4 blocks issues as few
        // loads as possible: ONE index load per quad (lane j fetches colind[k + j]; DPP hands the ids round),
        // FOUR 16-B matrix loads, and TWO 16-B x loads (quad lane j fetches half (j & 1) of the x block of
        // column k + (j >> 1), then of column k + 2 + (j >> 1)); the pair (2*hf, 2*hf+1) a lane needs arrives
        // by DPP.  7 VMEM instructions per step instead of 12.  The last step of a row is masked (zero matrix
        // values, clamped index) instead of a scalar tail: a 15-block row takes 4 memory round trips, not 6.
        const int32_t s = live ? rowptr[row] : 0, e = live ? rowptr[row + 1] : 0;
        const int jq = lane & 3;
        const double2* __restrict__ vp = reinterpret_cast<const double2*>(vals) + ((int64_t)s * 8 + r * 2 + hf);
        int32_t k = s;
        if (NT == 1) {
            // production: the first 16 blocks of the row (all of it on a tet mesh's fine level) are requested UP-FRONT --
            // four index loads, sixteen 16-B matrix loads, eight x loads: three dependent round trips per row instead of
            // two per step of 4 blocks plus two per tail block (A/B: 0.583 -> 0.523 ms at 10 M tets, sns_bench_variants 3;
            // NT == 3 keeps the stepped loop for that harness).  Missing blocks are masked (zero values, own x block).
            const int32_t cnt = e - s;
            const int32_t own = live ? row : 0;
            int32_t c[4];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) c[t4] = (4 * t4 + jq < cnt) ? colind[s + 4 * t4 + jq] : own;
            double2 a[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = (j < cnt) ? ld_stream(vp + 8 * j) : make_double2(0.0, 0.0);
            if (SPLIT == 3) gr.arrive(gs, (c[0] >= gr.n_own) | (c[1] >= gr.n_own) | (c[2] >= gr.n_own) | (c[3] >= gr.n_own));
            double2 gA[4], gB[4];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const int32_t cA = __builtin_amdgcn_mov_dpp(c[t4], 0x50, 0xF, 0xF, true);
                const int32_t cB = __builtin_amdgcn_mov_dpp(c[t4], 0xFA, 0xF, 0xF, true);
                gA[t4] = *reinterpret_cast<const double2*>((SPLIT == 3 ? gr.ptr(x, cA) : x + 4 * (int64_t)cA) + 2 * (jq & 1));
                gB[t4] = *reinterpret_cast<const double2*>((SPLIT == 3 ? gr.ptr(x, cB) : x + 4 * (int64_t)cB) + 2 * (jq & 1));
            }
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                acc0 += a[4 * t4].x * quad_perm<0x44>(gA[t4].x) + a[4 * t4].y * quad_perm<0x44>(gA[t4].y);
                acc1 += a[4 * t4 + 1].x * quad_perm<0xEE>(gA[t4].x) + a[4 * t4 + 1].y * quad_perm<0xEE>(gA[t4].y);
                acc0 += a[4 * t4 + 2].x * quad_perm<0x44>(gB[t4].x) + a[4 * t4 + 2].y * quad_perm<0x44>(gB[t4].y);
                acc1 += a[4 * t4 + 3].x * quad_perm<0xEE>(gB[t4].x) + a[4 * t4 + 3].y * quad_perm<0xEE>(gB[t4].y);
            }
            k = (cnt > 16) ? s + 16 : e;
            vp += 128;
        }
        if (NT == 2) {          // harness variant: r1e loop (4 broadcast index loads, 8-B x loads, scalar tail), nt stream
            for (; k + 3 < e; k += 4) {
                const int32_t c0 = colind[k], c1 = colind[k + 1], c2 = colind[k + 2], c3 = colind[k + 3];
                const double2 a0 = ld_stream(vp), a1 = ld_stream(vp + 8), a2 = ld_stream(vp + 16), a3 = ld_stream(vp + 24);
                const double g0 = x[4 * (int64_t)c0 + jq], g1 = x[4 * (int64_t)c1 + jq];
                const double g2 = x[4 * (int64_t)c2 + jq], g3 = x[4 * (int64_t)c3 + jq];
                acc0 += a0.x * quad_perm<0x88>(g0) + a0.y * quad_perm<0xDD>(g0);
                acc1 += a1.x * quad_perm<0x88>(g1) + a1.y * quad_perm<0xDD>(g1);
                acc0 += a2.x * quad_perm<0x88>(g2) + a2.y * quad_perm<0xDD>(g2);
                acc1 += a3.x * quad_perm<0x88>(g3) + a3.y * quad_perm<0xDD>(g3);
                vp += 32;
            }
        } else {
            for (; k + 3 < e; k += 4) {           // rows longer than 16 blocks (coarse levels); NT 0 / 3: the whole row
                const int32_t cme = colind[k + jq];
                const double2 a0 = (NT ? ld_stream(vp) : vp[0]), a1 = (NT ? ld_stream(vp + 8) : vp[8]),
                              a2 = (NT ? ld_stream(vp + 16) : vp[16]), a3 = (NT ? ld_stream(vp + 24) : vp[24]);
                const int32_t cA = __builtin_amdgcn_mov_dpp(cme, 0x50, 0xF, 0xF, true);      // ids of blocks [0,0,1,1]
                const int32_t cB = __builtin_amdgcn_mov_dpp(cme, 0xFA, 0xF, 0xF, true);      // ids of blocks [2,2,3,3]
                if (SPLIT == 3) gr.arrive(gs, cme >= gr.n_own);
                const double2 gA = *reinterpret_cast<const double2*>((SPLIT == 3 ? gr.ptr(x, cA) : x + 4 * (int64_t)cA) + 2 * (jq & 1));
                const double2 gB = *reinterpret_cast<const double2*>((SPLIT == 3 ? gr.ptr(x, cB) : x + 4 * (int64_t)cB) + 2 * (jq & 1));
                // lane (.., hf) takes its pair from quad lane hf (block 0 / 2) or 2 + hf (block 1 / 3)
                acc0 += a0.x * quad_perm<0x44>(gA.x) + a0.y * quad_perm<0x44>(gA.y);
                acc1 += a1.x * quad_perm<0xEE>(gA.x) + a1.y * quad_perm<0xEE>(gA.y);
                acc0 += a2.x * quad_perm<0x44>(gB.x) + a2.y * quad_perm<0x44>(gB.y);
                acc1 += a3.x * quad_perm<0xEE>(gB.x) + a3.y * quad_perm<0xEE>(gB.y);
                vp += 32;
            }
        }
        for (; k < e; ++k) {
            const double2 a0 = (NT ? ld_stream(vp) : vp[0]);
            const int32_t ck = colind[k];
            if (SPLIT == 3) gr.arrive(gs, ck >= gr.n_own);
            const double g0 = (SPLIT == 3 ? gr.ptr(x, ck) : x + 4 * (int64_t)ck)[jq];
            acc0 += a0.x * quad_perm<0x88>(g0) + a0.y * quad_perm<0xDD>(g0);
            vp += 8;
        }
    }
    double acc = acc0 + acc1;
    acc += __shfl_xor(acc, 1);                      // both halves now hold (A x)[4*row + r]
    if (MODE == SPMV_AX) {
        if (live && hf == 0) y[4 * (int64_t)row + r] = acc;
    } else if (MODE == SPMV_B_MINUS_AX) {
        if (live && hf == 0) y[4 * (int64_t)row + r] = pre_v - acc;
    } else if (MODE == SPMV_JACOBI) {
        // y = x + omega * Dinv (b - A x): residual of row component r lives on lanes (r,*);
        // fetch the 4 components of this row's residual with shuffles inside the 8-lane group.
        const double res = live ? (pre_v - acc) : 0.0;      // only the hf == 0 lanes' residuals are fetched below
        const int gbase = lane & ~7;
        const double r0 = __shfl(res, gbase + 0), r1 = __shfl(res, gbase + 2), r2 = __shfl(res, gbase + 4),
                     r3 = __shfl(res, gbase + 6);
        if (live && hf == 0) {
            const double* D = dinv + 16 * (int64_t)row + 4 * r;
            y[4 * (int64_t)row + r] =
                pre_x + omega * (D[0] * r0 + D[1] * r1 + D[2] * r2 + D[3] * r3);
        }
    } else if (MODE == SPMV_AX_DOT) {
        // y = A x and partial[block] = sum_rows dotw . y   (fused <r^, A p> of BiCGStab)
        double pr = 0.0;
        if (live && hf == 0) {
            y[4 * (int64_t)row + r] = acc;
            pr = acc * pre_v;
        }
        // one partial sum per WAVE (no workgroup barrier at the end of the kernel: a wave retires as soon as its
        // 8 rows are done); the host reduces 4 * gridDim.x entries
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pr += __shfl_xor(pr, o);
        if (SPLIT == 3) {
            // the latency-bound strong split: ONE partial per workgroup (a rank's 217 k rows leave 6.8 k instead of 27 k partials,
            // which the single-workgroup final stage takes without the chunk kernel in front of it)
            __shared__ double wsum[4];
            if (lane == 0) wsum[tid >> 6] = pr;
            __syncthreads();
            if (tid == 0) partial[(int64_t)blockIdx.x + partial_off] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        } else if (lane == 0) {
            partial[4 * ((int64_t)blockIdx.x + partial_off) + (tid >> 6)] = pr;
        }
    }
}

#define SNS_INST_SPMV(M, F, N, S)                                                                                  \
    template __global__ void k_spmv<M, F, N, S>(int32_t, const int32_t*, const int32_t*, const double*, const double*, \
                                                double*, const double*, const double*, double, const double*, double*, \
                                                const int32_t*, const uint8_t*, int, GhostSrc);
SNS_INST_SPMV(SPMV_AX, 1, 1, 0)
#ifdef SNS_HARNESS      // A/B variants of the experiment harness only (make HARNESS=1): not in the shipped library
SNS_INST_SPMV(SPMV_AX, 1, 0, 0)
SNS_INST_SPMV(SPMV_AX, 1, 2, 0)
SNS_INST_SPMV(SPMV_AX, 1, 3, 0)
SNS_INST_SPMV(SPMV_AX_DOT, 1, 3, 0)
#endif
SNS_INST_SPMV(SPMV_B_MINUS_AX, 1, 1, 0)
SNS_INST_SPMV(SPMV_JACOBI, 1, 1, 0)
SNS_INST_SPMV(SPMV_AX_DOT, 1, 1, 0)
SNS_INST_SPMV(SPMV_AX, 0, 0, 0)
SNS_INST_SPMV(SPMV_B_MINUS_AX, 0, 0, 0)
SNS_INST_SPMV(SPMV_JACOBI, 0, 0, 0)
SNS_INST_SPMV(SPMV_AX, 1, 1, 1)
SNS_INST_SPMV(SPMV_AX, 1, 1, 2)
SNS_INST_SPMV(SPMV_B_MINUS_AX, 1, 1, 1)
SNS_INST_SPMV(SPMV_B_MINUS_AX, 1, 1, 2)
SNS_INST_SPMV(SPMV_JACOBI, 1, 1, 1)
SNS_INST_SPMV(SPMV_JACOBI, 1, 1, 2)
SNS_INST_SPMV(SPMV_AX_DOT, 1, 1, 1)
SNS_INST_SPMV(SPMV_AX_DOT, 1, 1, 2)
SNS_INST_SPMV(SPMV_AX, 1, 1, 3)
SNS_INST_SPMV(SPMV_B_MINUS_AX, 1, 1, 3)
SNS_INST_SPMV(SPMV_JACOBI, 1, 1, 3)
SNS_INST_SPMV(SPMV_AX_DOT, 1, 1, 3)

// Preconditioner passes with fp32 MATRIX VALUES (vectors, D^-1 and all arithmetic stay fp64):
// the smoother / residual passes of the AMG cycle read a rounded copy of each level operator,
// 68 B instead of 132 B per block.  4 lanes per block row: lane r loads the whole row r of a block
// as one float4 (16 B) -> 16 block rows per wave, no cross-lane reduction.
// broadcast lane J of every quad (4 consecutive lanes)
template <int J>
__device__ __forceinline__ double quad_bcast(double v) { return quad_perm<J * 0x55>(v); }
template <int J>
__device__ __forceinline__ int quad_bcast_i(int v) { return __builtin_amdgcn_mov_dpp(v, J * 0x55, 0xF, 0xF, true); }

// k_spmv_lp: the preconditioner passes (Jacobi sweep, residual) on a LOW-PRECISION COPY of the level matrix.
//   FMT 1: fp32 values (68 B per block with its index instead of 132 B)
//   FMT 2: fp16 values with one fp32 scale per dof row (row-max normalisation: nothing overflows or underflows
//          the half range), 36 B per block; (A x)_r = scale_r * sum_k half_rk x_k
// Vectors, D^-1 and ALL arithmetic stay fp64, so the cycle remains a fixed linear operator in exact arithmetic of
// a slightly perturbed matrix.  4 lanes per block row (16 block rows per wave).  The kernel is bound by VMEM issue /
// L2 requests rather than bytes, so a step of 4 blocks issues as few, as wide loads as it can:
//   * ONE index load per quad (lane j fetches the id of block k + j),
//   * the x gather as TWO 16-B loads per lane: lane j fetches the WHOLE x block of column (k + j); DPP broadcasts
//     hand the four blocks round (instead of four 8-B loads per lane),
//   * the matrix row of lane r as 16-B loads: fp32: one per block; fp16: one per PAIR of blocks -- the fp16 copy
//     stores the blocks of a row pair-interleaved ([row 0 of blocks k,k+1 | row 1 of k,k+1 | ...], 64 B per pair; an
//     odd last block keeps the plain layout), so two blocks cost one load.
// 7 (fp32) / 5 (fp16) VMEM instructions per lane and step instead of 9.
//   SPLIT as in k_spmv (multi-GPU interior / boundary passes).
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ double lp_dot(const float4 a, const double2 xa, const double2 xb) {
    return (double)a.x * xa.x + (double)a.y * xa.y + (double)a.z * xb.x + (double)a.w * xb.y;
}
__device__ __forceinline__ double lp_dot(const f16x4_t a, const double2 xa, const double2 xb) {
    return (double)(float)a.x * xa.x + (double)(float)a.y * xa.y + (double)(float)a.z * xb.x + (double)(float)a.w * xb.y;
}
template <int J>
__device__ __forceinline__ double2 quad_bcast2(const double2 v) {
    return make_double2(quad_bcast<J>(v.x), quad_bcast<J>(v.y));
}

template <int MODE, int FINE, int SPLIT, int FMT, int UP>
__global__ __launch_bounds__(256) void k_spmv_lp(int32_t n_rows, const int32_t* __restrict__ rowptr,
                                                 const int32_t* __restrict__ colind, const void* __restrict__ vals_v,
                                                 const float* __restrict__ scale, const double* __restrict__ x,
                                                 double* __restrict__ y, const double* __restrict__ bvec,
                                                 const float* __restrict__ dinv, double omega,
                                                 const int32_t* __restrict__ row_list,
                                                 const uint8_t* __restrict__ skip, GhostSrc gs) {
    GhostReader gr;                                               // SPLIT 3: ghost entries from the receive window (see k_spmv)
    if (SPLIT == 3) gr.begin(gs);
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int r = lane & 3;
    const int32_t ridx = (blk * 4 + (tid >> 6)) * 16 + (lane >> 2);
    bool live = ridx < n_rows;
    int32_t row = ridx;
    if (SPLIT == 2) row = live ? row_list[ridx] : 0;
    if (SPLIT == 1) live = live && !skip[ridx];
    double acc0 = 0.0, acc1 = 0.0;
    // every lane of a quad walks the same row, so the loop trip count is quad-uniform (DPP needs all 4 lanes)
    const int32_t s = live ? rowptr[row] : 0, e = live ? rowptr[row + 1] : 0;
    // the row's b, x and D^-1 entries are requested BEFORE the block loop, so their latency hides behind it
    double pre_b = 0.0, pre_x = 0.0, sc = 1.0;
    float4 pre_d = make_float4(0.f, 0.f, 0.f, 0.f);      // row r of the node's D^-1 (fp32 copy: part of the smoother's matrix data)
    if (FMT == 2 && live) sc = (double)scale[4 * (int64_t)row + r];
    if (MODE == SPMV_B_MINUS_AX && live) pre_b = bvec[4 * (int64_t)row + r];
    if (MODE == SPMV_JACOBI && live) {
        pre_b = bvec[4 * (int64_t)row + r];
        pre_x = x[4 * (int64_t)row + r];
        pre_d = *reinterpret_cast<const float4*>(dinv + 16 * (int64_t)row + 4 * r);
    }
    const float4* __restrict__ v32 = reinterpret_cast<const float4*>(vals_v) + ((int64_t)s * 4 + r);       // FMT 1
    const uint4* __restrict__ v16 = reinterpret_cast<const uint4*>(vals_v) + ((int64_t)s * 2 + r);         // FMT 2: pairs
    int32_t k = s;
    if constexpr (FMT == 2 && UP > 0) {
        // The first 16 * UP blocks of a row (UP = 1: all of it on a tet mesh's fine level, 15 blocks) are requested UP-FRONT:
        // the index loads, the 16-B matrix loads, then the x loads -- three dependent round trips for the whole row instead
        // of two per step of 4 blocks.  (UP = 2, 32 blocks for the 27-block rows of the coarse levels, was measured in the
        // solver and is slower: 168 VGPRs, level-1 sweep 40 instead of 36 us.)  The kernel
        // is latency-sensitive (PMC: 20 k DRAM lines in flight against 40 k for the fp64 kernel),
        // not bound by bytes, issue or ALU work.  Missing blocks are masked: zero matrix values, the row's own x block.
        constexpr int NS = 4 * UP;                                // steps of 4 blocks requested up-front
        const int32_t cnt = e - s;
        const int32_t np = cnt >> 1;
        int32_t c[NS];
        const int32_t own = live ? row : 0;                       // dead lanes (row index past the end) gather block 0
#pragma unroll
        for (int t = 0; t < NS; ++t) c[t] = (4 * t + r < cnt) ? colind[s + 4 * t + r] : own;
        uint4 P[2 * NS];
#pragma unroll
        for (int q = 0; q < 2 * NS; ++q) {
            P[q] = make_uint4(0u, 0u, 0u, 0u);
            if (q < np) P[q] = v16[4 * q];
            else if (q == np && (cnt & 1)) {                      // the odd last block (plain layout) as the low half of a pair
                const uint2 o = reinterpret_cast<const uint2*>(vals_v)[(int64_t)(e - 1) * 4 + r];
                P[q].x = o.x; P[q].y = o.y;
            }
        }
        if (SPLIT == 3) {
            bool gh = false;
#pragma unroll
            for (int t = 0; t < NS; ++t) gh = gh | (c[t] >= gr.n_own);
            gr.arrive(gs, gh);
        }
        double2 xa[NS], xb[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const double2* xp = reinterpret_cast<const double2*>(SPLIT == 3 ? gr.ptr(x, c[t]) : x + 4 * (int64_t)c[t]);
            xa[t] = xp[0]; xb[t] = xp[1];
        }
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const f16x8_t h0 = *reinterpret_cast<const f16x8_t*>(&P[2 * t]), h1 = *reinterpret_cast<const f16x8_t*>(&P[2 * t + 1]);
            acc0 += lp_dot(h0.lo, quad_bcast2<0>(xa[t]), quad_bcast2<0>(xb[t]));
            acc1 += lp_dot(h0.hi, quad_bcast2<1>(xa[t]), quad_bcast2<1>(xb[t]));
            acc0 += lp_dot(h1.lo, quad_bcast2<2>(xa[t]), quad_bcast2<2>(xb[t]));
            acc1 += lp_dot(h1.hi, quad_bcast2<3>(xa[t]), quad_bcast2<3>(xb[t]));
        }
        k = (cnt > 4 * NS) ? s + 4 * NS : e;                      // longer rows (unstructured meshes): the loop below
        v16 += 8 * NS;
    }
    // the block ids of the NEXT step are requested one step ahead: index -> x gather is a dependent chain of two
    // memory round trips per step otherwise (matters on the coarse levels, whose rows have ~27 blocks = 7 steps)
    int32_t cnext = (k + 3 < e) ? colind[k + r] : 0;
    for (; k + 3 < e; k += 4) {
        const int32_t cme = cnext;
        if (k + 7 < e) cnext = colind[k + 4 + r];
        if (SPLIT == 3) gr.arrive(gs, cme >= gr.n_own);
        const double2* xp = reinterpret_cast<const double2*>(SPLIT == 3 ? gr.ptr(x, cme) : x + 4 * (int64_t)cme);
        const double2 xa = xp[0], xb = xp[1];                     // the whole x block of column (k + r)
        if (FMT == 1) {
            const float4 a0 = v32[0], a1 = v32[4], a2 = v32[8], a3 = v32[12];
            acc0 += lp_dot(a0, quad_bcast2<0>(xa), quad_bcast2<0>(xb));
            acc1 += lp_dot(a1, quad_bcast2<1>(xa), quad_bcast2<1>(xb));
            acc0 += lp_dot(a2, quad_bcast2<2>(xa), quad_bcast2<2>(xb));
            acc1 += lp_dot(a3, quad_bcast2<3>(xa), quad_bcast2<3>(xb));
            v32 += 16;
        } else {
            const uint4 p0 = v16[0], p1 = v16[4];                 // row r of blocks (k, k+1) and (k+2, k+3)
            const f16x8_t h0 = *reinterpret_cast<const f16x8_t*>(&p0), h1 = *reinterpret_cast<const f16x8_t*>(&p1);
            acc0 += lp_dot(h0.lo, quad_bcast2<0>(xa), quad_bcast2<0>(xb));
            acc1 += lp_dot(h0.hi, quad_bcast2<1>(xa), quad_bcast2<1>(xb));
            acc0 += lp_dot(h1.lo, quad_bcast2<2>(xa), quad_bcast2<2>(xb));
            acc1 += lp_dot(h1.hi, quad_bcast2<3>(xa), quad_bcast2<3>(xb));
            v16 += 8;
        }
    }
    if (k < e) {                                                  // 1..3 blocks left; quad-uniform
        const int32_t left = e - k;
        const int32_t cme = colind[k + (r < left ? r : 0)];
        if (SPLIT == 3) gr.arrive(gs, cme >= gr.n_own);
        const double2* xp = reinterpret_cast<const double2*>(SPLIT == 3 ? gr.ptr(x, cme) : x + 4 * (int64_t)cme);
        const double2 xa = xp[0], xb = xp[1];
        if (FMT == 1) {
            acc0 += lp_dot(v32[0], quad_bcast2<0>(xa), quad_bcast2<0>(xb));
            if (left > 1) acc1 += lp_dot(v32[4], quad_bcast2<1>(xa), quad_bcast2<1>(xb));
            if (left > 2) acc0 += lp_dot(v32[8], quad_bcast2<2>(xa), quad_bcast2<2>(xb));
        } else {
            if (left > 1) {
                const uint4 p0 = v16[0];
                const f16x8_t h0 = *reinterpret_cast<const f16x8_t*>(&p0);
                acc0 += lp_dot(h0.lo, quad_bcast2<0>(xa), quad_bcast2<0>(xb));
                acc1 += lp_dot(h0.hi, quad_bcast2<1>(xa), quad_bcast2<1>(xb));
            }
            if (left & 1) {                                       // the odd last block: plain layout, 8 B per row
                const uint2 q = reinterpret_cast<const uint2*>(vals_v)[(int64_t)(e - 1) * 4 + r];
                const f16x4_t hq = *reinterpret_cast<const f16x4_t*>(&q);
                if (left == 1) acc0 += lp_dot(hq, quad_bcast2<0>(xa), quad_bcast2<0>(xb));
                else acc0 += lp_dot(hq, quad_bcast2<2>(xa), quad_bcast2<2>(xb));
            }
        }
    }
    const double acc = sc * (acc0 + acc1);          // (A x)[4*row + r]
    if (MODE == SPMV_B_MINUS_AX) {
        if (live) y[4 * (int64_t)row + r] = pre_b - acc;
    } else if (MODE == SPMV_JACOBI) {
        const double res = live ? (pre_b - acc) : 0.0;
        const double r0 = quad_bcast<0>(res), r1 = quad_bcast<1>(res), r2 = quad_bcast<2>(res), r3 = quad_bcast<3>(res);
        if (live)
            y[4 * (int64_t)row + r] = pre_x + omega * ((double)pre_d.x * r0 + (double)pre_d.y * r1 + (double)pre_d.z * r2 +
                                                        (double)pre_d.w * r3);
    }
}
#define SNS_INST_LP(M, F, S, T, U)                                                                                 \
    template __global__ void k_spmv_lp<M, F, S, T, U>(int32_t, const int32_t*, const int32_t*, const void*,           \
                                                      const float*, const double*, double*, const double*,           \
                                                      const float*, double, const int32_t*, const uint8_t*, GhostSrc);
#define SNS_INST_LP_FMT(T)                   \
    SNS_INST_LP(SPMV_B_MINUS_AX, 1, 0, T, 1) \
    SNS_INST_LP(SPMV_B_MINUS_AX, 1, 1, T, 1) \
    SNS_INST_LP(SPMV_B_MINUS_AX, 1, 2, T, 1) \
    SNS_INST_LP(SPMV_B_MINUS_AX, 1, 3, T, 1) \
    SNS_INST_LP(SPMV_B_MINUS_AX, 0, 0, T, 1) \
    SNS_INST_LP(SPMV_JACOBI, 1, 0, T, 1)     \
    SNS_INST_LP(SPMV_JACOBI, 1, 1, T, 1)     \
    SNS_INST_LP(SPMV_JACOBI, 1, 2, T, 1)     \
    SNS_INST_LP(SPMV_JACOBI, 1, 3, T, 1)     \
    SNS_INST_LP(SPMV_JACOBI, 0, 0, T, 1)
SNS_INST_LP_FMT(1)
SNS_INST_LP_FMT(2)
#ifdef SNS_HARNESS
SNS_INST_LP(SPMV_B_MINUS_AX, 1, 0, 2, 0) SNS_INST_LP(SPMV_JACOBI, 1, 0, 2, 0)      // in-solver A/B of the stepped loop
#endif

// k_post_lp: the FIRST post-smoothing sweep of a V-cycle level fused with the coarse-grid correction (round 3).
// With x2 = x1 + P xc the sweep z = x2 + w Dinv (b - A x2) equals
//     z = (x1 + P xc) + w Dinv (r1 - M xc),   r1 = b - A x1 (the residual the level restricts anyway),  M = A P,
// and M -- fine rows x COARSE columns, the blocks of a row summed per aggregate (k_ap_sum) -- has 0.37x the blocks of A on
// a tet mesh's fine level (15 neighbours fall into <= 8 aggregates).  So the sweep reads M instead of A and gathers from
// the small coarse vector, and the prolongation kernel disappears (x1 + P xc is the row's own epilogue operand).  Same
// linear operator as before up to rounding; M is held in the same low-precision format as the level matrix (fp16 with
// row scales, pair-interleaved, or fp32).  4 lanes per block row like k_spmv_lp; the first 8 blocks of a row (all of it
// on the fine level) are requested up-front.
// __launch_bounds__(256, 6): 80 instead of 103 VGPRs (fp16), 6 instead of 4 waves per SIMD, no spills -- unlike the full
// sweeps this kernel gains from the occupancy (in-solver A/B at 10 M tets: 169 -> 147 us)
template <int FMT, int FINE>
__global__ __launch_bounds__(256, 6) void k_post_lp(int32_t n_rows, const int32_t* __restrict__ rowptr,
                                                 const int32_t* __restrict__ colind, const void* __restrict__ vals_v,
                                                 const float* __restrict__ scale, const double* __restrict__ xc,
                                                 const double* __restrict__ x_pre, const double* __restrict__ res1,
                                                 const float* __restrict__ dinv, double omega,
                                                 const int32_t* __restrict__ agg, const uint8_t* __restrict__ free_mask,
                                                 double* __restrict__ y, GhostSrc gs) {
    // FINE 2: the fine level of a partitioned handle (window transports) -- the ghost aggregates' entries of xc come straight from
    // the coarse level's receive window, the wave that meets one waits for the neighbours' flags itself (see k_spmv, SPLIT 3)
    constexpr bool GH = FINE == 2;
    GhostReader gr;
    if (GH) gr.begin(gs);
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int r = lane & 3;
    const int32_t row = (blk * 4 + (tid >> 6)) * 16 + (lane >> 2);
    const bool live = row < n_rows;
    const int32_t s = live ? rowptr[row] : 0, e = live ? rowptr[row + 1] : 0;
    double pre_b = 0.0, pre_x = 0.0, sc = 1.0;
    float4 pre_d = make_float4(0.f, 0.f, 0.f, 0.f);
    int32_t I = -1;
    if (live) {
        I = agg[row];
        if (FMT == 2) sc = (double)scale[4 * (int64_t)row + r];
        pre_b = res1[4 * (int64_t)row + r];
        pre_x = x_pre[4 * (int64_t)row + r];
        pre_d = *reinterpret_cast<const float4*>(dinv + 16 * (int64_t)row + 4 * r);
        if (I >= 0 && (!free_mask || free_mask[4 * (int64_t)row + r])) pre_x += xc[4 * (int64_t)I + r];   // (x1 + P xc)_row
    }
    double acc0 = 0.0, acc1 = 0.0;
    const float4* __restrict__ v32 = reinterpret_cast<const float4*>(vals_v) + ((int64_t)s * 4 + r);
    const uint4* __restrict__ v16 = reinterpret_cast<const uint4*>(vals_v) + ((int64_t)s * 2 + r);
    int32_t k = s;
    if constexpr (FMT == 2) {
        constexpr int NS = 2;
        const int32_t cnt = e - s;
        const int32_t np = cnt >> 1;
        int32_t c[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) c[t] = (4 * t + r < cnt) ? colind[s + 4 * t + r] : 0;      // masked: coarse node 0 (in bounds)
        uint4 P[2 * NS];
#pragma unroll
        for (int q = 0; q < 2 * NS; ++q) {
            P[q] = make_uint4(0u, 0u, 0u, 0u);
            if (q < np) P[q] = v16[4 * q];
            else if (q == np && (cnt & 1)) {
                const uint2 o = reinterpret_cast<const uint2*>(vals_v)[(int64_t)(e - 1) * 4 + r];
                P[q].x = o.x; P[q].y = o.y;
            }
        }
        if (GH) gr.arrive(gs, (c[0] >= gr.n_own) | (c[1] >= gr.n_own));
        double2 xa[NS], xb[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const double2* xp = reinterpret_cast<const double2*>(GH ? gr.ptr(xc, c[t]) : xc + 4 * (int64_t)c[t]);
            xa[t] = xp[0]; xb[t] = xp[1];
        }
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const f16x8_t h0 = *reinterpret_cast<const f16x8_t*>(&P[2 * t]), h1 = *reinterpret_cast<const f16x8_t*>(&P[2 * t + 1]);
            acc0 += lp_dot(h0.lo, quad_bcast2<0>(xa[t]), quad_bcast2<0>(xb[t]));
            acc1 += lp_dot(h0.hi, quad_bcast2<1>(xa[t]), quad_bcast2<1>(xb[t]));
            acc0 += lp_dot(h1.lo, quad_bcast2<2>(xa[t]), quad_bcast2<2>(xb[t]));
            acc1 += lp_dot(h1.hi, quad_bcast2<3>(xa[t]), quad_bcast2<3>(xb[t]));
        }
        k = (cnt > 4 * NS) ? s + 4 * NS : e;
        v16 += 8 * NS;
    }
    for (; k + 3 < e; k += 4) {                                   // longer rows (coarse levels, unstructured meshes)
        const int32_t cme = colind[k + r];
        if (GH) gr.arrive(gs, cme >= gr.n_own);
        const double2* xp = reinterpret_cast<const double2*>(GH ? gr.ptr(xc, cme) : xc + 4 * (int64_t)cme);
        const double2 xa = xp[0], xb = xp[1];
        if (FMT == 1) {
            const float4 a0 = v32[0], a1 = v32[4], a2 = v32[8], a3 = v32[12];
            acc0 += lp_dot(a0, quad_bcast2<0>(xa), quad_bcast2<0>(xb));
            acc1 += lp_dot(a1, quad_bcast2<1>(xa), quad_bcast2<1>(xb));
            acc0 += lp_dot(a2, quad_bcast2<2>(xa), quad_bcast2<2>(xb));
            acc1 += lp_dot(a3, quad_bcast2<3>(xa), quad_bcast2<3>(xb));
            v32 += 16;
        } else {
            const uint4 p0 = v16[0], p1 = v16[4];
            const f16x8_t h0 = *reinterpret_cast<const f16x8_t*>(&p0), h1 = *reinterpret_cast<const f16x8_t*>(&p1);
            acc0 += lp_dot(h0.lo, quad_bcast2<0>(xa), quad_bcast2<0>(xb));
            acc1 += lp_dot(h0.hi, quad_bcast2<1>(xa), quad_bcast2<1>(xb));
            acc0 += lp_dot(h1.lo, quad_bcast2<2>(xa), quad_bcast2<2>(xb));
            acc1 += lp_dot(h1.hi, quad_bcast2<3>(xa), quad_bcast2<3>(xb));
            v16 += 8;
        }
    }
    if (k < e) {                                                  // 1..3 blocks left; quad-uniform
        const int32_t left = e - k;
        const int32_t cme = colind[k + (r < left ? r : 0)];
        if (GH) gr.arrive(gs, cme >= gr.n_own);
        const double2* xp = reinterpret_cast<const double2*>(GH ? gr.ptr(xc, cme) : xc + 4 * (int64_t)cme);
        const double2 xa = xp[0], xb = xp[1];
        if (FMT == 1) {
            acc0 += lp_dot(v32[0], quad_bcast2<0>(xa), quad_bcast2<0>(xb));
            if (left > 1) acc1 += lp_dot(v32[4], quad_bcast2<1>(xa), quad_bcast2<1>(xb));
            if (left > 2) acc0 += lp_dot(v32[8], quad_bcast2<2>(xa), quad_bcast2<2>(xb));
        } else {
            if (left > 1) {
                const uint4 p0 = v16[0];
                const f16x8_t h0 = *reinterpret_cast<const f16x8_t*>(&p0);
                acc0 += lp_dot(h0.lo, quad_bcast2<0>(xa), quad_bcast2<0>(xb));
                acc1 += lp_dot(h0.hi, quad_bcast2<1>(xa), quad_bcast2<1>(xb));
            }
            if (left & 1) {
                const uint2 q = reinterpret_cast<const uint2*>(vals_v)[(int64_t)(e - 1) * 4 + r];
                const f16x4_t hq = *reinterpret_cast<const f16x4_t*>(&q);
                if (left == 1) acc0 += lp_dot(hq, quad_bcast2<0>(xa), quad_bcast2<0>(xb));
                else acc0 += lp_dot(hq, quad_bcast2<2>(xa), quad_bcast2<2>(xb));
            }
        }
    }
    const double acc = sc * (acc0 + acc1);                        // (M xc)[4*row + r]
    const double rr = live ? (pre_b - acc) : 0.0;
    const double r0 = quad_bcast<0>(rr), r1 = quad_bcast<1>(rr), r2 = quad_bcast<2>(rr), r3 = quad_bcast<3>(rr);
    if (live)
        y[4 * (int64_t)row + r] = pre_x + omega * ((double)pre_d.x * r0 + (double)pre_d.y * r1 + (double)pre_d.z * r2 +
                                                    (double)pre_d.w * r3);
}
#define SNS_INST_POST(T, F)                                                                                        \
    template __global__ void k_post_lp<T, F>(int32_t, const int32_t*, const int32_t*, const void*, const float*,   \
                                             const double*, const double*, const double*, const float*, double,   \
                                             const int32_t*, const uint8_t*, double*, GhostSrc);
SNS_INST_POST(1, 0) SNS_INST_POST(1, 1) SNS_INST_POST(2, 0) SNS_INST_POST(2, 1) SNS_INST_POST(1, 2) SNS_INST_POST(2, 2)

// M = A P for the fused post-smoothing sweep in fp32 (amg_f32_matrix = 1; the fp16 format is written by k_lp_copies16 below):
// M slot (i, J) <- sum of the fine blocks (i, j), j in aggregate J (gather list ap_ptr / ap_idx, fixed order), no fp64 copy of M
// in between.  4 lanes per fine block row (lane r = dof row r).  Dofs excluded from the transfer (level 0: Dirichlet dofs) have
// zero columns in A except the unit diagonal (:74), so the only thing to take out is that 1.0 where the row's own node is in J.
__global__ __launch_bounds__(256) void k_ap_cvt32(int32_t n_rows, const int32_t* __restrict__ rowptr_m,
                                                  const int32_t* __restrict__ colind_m, const int32_t* __restrict__ ap_ptr,
                                                  const int32_t* __restrict__ ap_idx, const double* __restrict__ vals_f,
                                                  const int32_t* __restrict__ agg, const uint8_t* __restrict__ free_mask,
                                                  float4* __restrict__ out) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = gid >> 2;
    const int r = (int)(gid & 3);
    if (row >= n_rows) return;
    const int32_t s = rowptr_m[row], e = rowptr_m[row + 1];
    const int32_t own = agg[row];
    const bool fixed = free_mask && !free_mask[4 * row + r];
    for (int32_t q = s; q < e; ++q) {
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        for (int32_t k = ap_ptr[q]; k < ap_ptr[q + 1]; ++k) {
            const double* a = vals_f + 16 * (int64_t)ap_idx[k] + 4 * r;
            v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
        }
        if (fixed && colind_m[q] == own) v[r] -= 1.0;
        out[(int64_t)q * 4 + r] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
}

// k_lp_copies16: the fp16 copies of a level in ONE pass over its fp64 operator (round 3): the copy of A the smoother reads -- one
// fp32 scale per dof row (the row's largest |entry|), values pair-interleaved as k_spmv_lp<FMT 2> wants them: 4 lanes per block
// row, lane r owns dof row 4*row + r; block j of a row that has a partner (j ^ 1 within the row) lands in pair j >> 1, half j & 1,
// an odd last block keeps the plain position -- and, if the level has one, the same of M = A P for the fused post-smoothing sweep
// (M slot (i, J) <- sum of the fine blocks (i, j), j in aggregate J; a Dirichlet dof's unit diagonal (:74) taken out where the
// row's own node is in J).  Until round 3 two kernels (k_cvt_h16, k_ap_cvt<2>) read the 2.3 GB fine-level operator once each in
// latency-bound loops of one block per step (1.4 - 1.5 ms each at 10 M tets = 2 TB/s).  Here a row of up to 16 blocks is requested UP-FRONT into registers (32 16-B loads in flight per lane), its
// largest |entry| found, the copy of A written (one 16-B store per pair), and M accumulated from the very same registers: slot t of
// row i sums the blocks whose nibble in ap_nib[i] is t (block j -> bits 4j .. 4j+3; 15 = the column takes no part), in block order
// = the order of the ap_idx gather list, so that both copies are BITWISE what the two kernels produced (checked in the solver: same
// V-cycle, same iterates; a setup without spectral estimates 5.3 -> 3.05 ms at 10 M tets, 7.2 -> 4.9 ms on average).  The accumulators of the <= 8
// M slots of a row live in LDS, 32 B per lane and slot, private to the lane (indexable registers; no barrier).  Rows of more than
// 16 blocks or more than 8 slots (ap_nib = ~0: coarse levels, unstructured meshes) take the one-block-per-step loops.
template <int WITH_M>
__global__ __launch_bounds__(128) void k_lp_copies16(int32_t n_rows, const int32_t* __restrict__ rowptr,
                                                     const double* __restrict__ vals, uint2* __restrict__ out,
                                                     float* __restrict__ scale, const int32_t* __restrict__ rowptr_m,
                                                     const int32_t* __restrict__ colind_m, const int32_t* __restrict__ ap_ptr,
                                                     const int32_t* __restrict__ ap_idx, const uint64_t* __restrict__ ap_nib,
                                                     const int32_t* __restrict__ agg, const uint8_t* __restrict__ free_mask,
                                                     uint2* __restrict__ out_m, float* __restrict__ scale_m) {
    __shared__ double2 acc_lds[WITH_M ? 128 * 8 * 2 : 1];           // [slot][half][lane]
    const int tid = threadIdx.x;
    const int64_t row = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 32 + (tid >> 2);       // each XCD walks one contiguous eighth
    const int r = tid & 3;
    if (row >= n_rows) return;
    const int32_t s = rowptr[row], e = rowptr[row + 1];
    const int32_t cnt = e - s;
    constexpr int NB = 16;
    double2 A[NB][2];
    const bool in_regs = cnt <= NB;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            A[j][0] = A[j][1] = make_double2(0.0, 0.0);
            if (j < cnt) {
                const double2* a = reinterpret_cast<const double2*>(vals + 16 * (int64_t)(s + j) + 4 * r);
                A[j][0] = ld_stream(a); A[j][1] = ld_stream(a + 1);                     // read once: streaming loads
            }
        }
        double m = 0.0;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            m = fmax(m, fmax(fmax(fabs(A[j][0].x), fabs(A[j][0].y)), fmax(fabs(A[j][1].x), fabs(A[j][1].y))));
        const float sc = (float)m;
        const double inv = m > 0.0 ? 1.0 / (double)sc : 0.0;
        scale[4 * row + r] = sc;
#pragma unroll
        for (int p = 0; p < NB / 2; ++p) {
            f16x8_t hv;
            hv[0] = (_Float16)(float)(A[2 * p][0].x * inv); hv[1] = (_Float16)(float)(A[2 * p][0].y * inv);
            hv[2] = (_Float16)(float)(A[2 * p][1].x * inv); hv[3] = (_Float16)(float)(A[2 * p][1].y * inv);
            hv[4] = (_Float16)(float)(A[2 * p + 1][0].x * inv); hv[5] = (_Float16)(float)(A[2 * p + 1][0].y * inv);
            hv[6] = (_Float16)(float)(A[2 * p + 1][1].x * inv); hv[7] = (_Float16)(float)(A[2 * p + 1][1].y * inv);
            const uint4 u = *reinterpret_cast<const uint4*>(&hv);
            if (2 * p + 1 < cnt) *reinterpret_cast<uint4*>(out + ((int64_t)s * 4 + p * 8 + r * 2)) = u;      // a pair: 16 B per lane
            else if (2 * p < cnt) out[(int64_t)(s + 2 * p) * 4 + r] = make_uint2(u.x, u.y);                  // the odd last block
        }
    } else {
        double m = 0.0;
        for (int32_t k = s; k < e; ++k) {
            const double* a = vals + 16 * (int64_t)k + 4 * r;
            m = fmax(m, fmax(fmax(fabs(a[0]), fabs(a[1])), fmax(fabs(a[2]), fabs(a[3]))));
        }
        const float sc = (float)m;
        const double inv = m > 0.0 ? 1.0 / (double)sc : 0.0;
        scale[4 * row + r] = sc;
        for (int32_t k = s; k < e; ++k) {
            const double* a = vals + 16 * (int64_t)k + 4 * r;
            f16x4_t hv;
            hv.x = (_Float16)(float)(a[0] * inv); hv.y = (_Float16)(float)(a[1] * inv);
            hv.z = (_Float16)(float)(a[2] * inv); hv.w = (_Float16)(float)(a[3] * inv);
            const int32_t j = k - s;
            const bool paired = (j | 1) < cnt;
            const int64_t dst = paired ? ((int64_t)s * 4 + (int64_t)(j >> 1) * 8 + r * 2 + (j & 1)) : ((int64_t)k * 4 + r);
            out[dst] = *reinterpret_cast<const uint2*>(&hv);
        }
    }
    if constexpr (WITH_M) {
        const int32_t sm = rowptr_m[row], em = rowptr_m[row + 1];
        const int32_t cm = em - sm;
        const int32_t own = agg[row];
        const bool fixed = free_mask && !free_mask[4 * row + r];
        const uint64_t nib = ap_nib[row];
        if (in_regs && nib != ~0ull) {
            // accumulator (slot t, half h) of lane tid at acc_lds[(2 t + h) * 128 + tid]: consecutive lanes 16 B apart (a lane-major
            // layout puts all 64 lanes of a wave on the same banks: setup 3.21 instead of 3.05 ms at 10 M tets)
            double2* acc = acc_lds + tid;
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t * 128] = make_double2(0.0, 0.0);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int t = (int)((nib >> (4 * j)) & 15);
                if (j < cnt && t < 8) {
                    double2 v0 = acc[(2 * t) * 128], v1 = acc[(2 * t + 1) * 128];
                    v0.x += A[j][0].x; v0.y += A[j][0].y; v1.x += A[j][1].x; v1.y += A[j][1].y;
                    acc[(2 * t) * 128] = v0; acc[(2 * t + 1) * 128] = v1;
                }
            }
            double m = 0.0;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t < cm) {
                    if (fixed && colind_m[sm + t] == own) {                    // the unit diagonal of a Dirichlet dof is not part of M
                        double2 v = acc[(2 * t + (r >> 1)) * 128];
                        if (r & 1) v.y -= 1.0; else v.x -= 1.0;
                        acc[(2 * t + (r >> 1)) * 128] = v;
                    }
                    const double2 v0 = acc[(2 * t) * 128], v1 = acc[(2 * t + 1) * 128];
                    m = fmax(m, fmax(fmax(fabs(v0.x), fabs(v0.y)), fmax(fabs(v1.x), fabs(v1.y))));
                }
            }
            const float sc = (float)m;
            const double inv = m > 0.0 ? 1.0 / (double)sc : 0.0;
            scale_m[4 * row + r] = sc;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (2 * p >= cm) break;
                const double2 a0 = acc[(4 * p) * 128], a1 = acc[(4 * p + 1) * 128];
                f16x8_t hv;
                hv[0] = (_Float16)(float)(a0.x * inv); hv[1] = (_Float16)(float)(a0.y * inv);
                hv[2] = (_Float16)(float)(a1.x * inv); hv[3] = (_Float16)(float)(a1.y * inv);
                if (2 * p + 1 < cm) {
                    const double2 b0 = acc[(4 * p + 2) * 128], b1 = acc[(4 * p + 3) * 128];
                    hv[4] = (_Float16)(float)(b0.x * inv); hv[5] = (_Float16)(float)(b0.y * inv);
                    hv[6] = (_Float16)(float)(b1.x * inv); hv[7] = (_Float16)(float)(b1.y * inv);
                    *reinterpret_cast<uint4*>(out_m + ((int64_t)sm * 4 + p * 8 + r * 2)) = *reinterpret_cast<const uint4*>(&hv);
                } else {
                    const uint4 u = *reinterpret_cast<const uint4*>(&hv);
                    out_m[(int64_t)(sm + 2 * p) * 4 + r] = make_uint2(u.x, u.y);
                }
            }
        } else {
            auto slot_sum = [&](int32_t q, double (&v)[4]) {
                v[0] = v[1] = v[2] = v[3] = 0.0;
                for (int32_t k = ap_ptr[q]; k < ap_ptr[q + 1]; ++k) {
                    const double* a = vals + 16 * (int64_t)ap_idx[k] + 4 * r;
                    v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
                }
                if (fixed && colind_m[q] == own) v[r] -= 1.0;
            };
            double m = 0.0;
            for (int32_t q = sm; q < em; ++q) {
                double v[4];
                slot_sum(q, v);
                m = fmax(m, fmax(fmax(fabs(v[0]), fabs(v[1])), fmax(fabs(v[2]), fabs(v[3]))));
            }
            const float sc = (float)m;
            const double inv = m > 0.0 ? 1.0 / (double)sc : 0.0;
            scale_m[4 * row + r] = sc;
            for (int32_t q = sm; q < em; ++q) {
                double v[4];
                slot_sum(q, v);
                f16x4_t hv;
                hv.x = (_Float16)(float)(v[0] * inv); hv.y = (_Float16)(float)(v[1] * inv);
                hv.z = (_Float16)(float)(v[2] * inv); hv.w = (_Float16)(float)(v[3] * inv);
                const int32_t j = q - sm;
                const bool paired = (j | 1) < cm;
                const int64_t dst = paired ? ((int64_t)sm * 4 + (int64_t)(j >> 1) * 8 + r * 2 + (j & 1)) : ((int64_t)q * 4 + r);
                out_m[dst] = *reinterpret_cast<const uint2*>(&hv);
            }
        }
    }
}
template __global__ void k_lp_copies16<0>(int32_t, const int32_t*, const double*, uint2*, float*, const int32_t*, const int32_t*,
                                          const int32_t*, const int32_t*, const uint64_t*, const int32_t*, const uint8_t*, uint2*, float*);
template __global__ void k_lp_copies16<1>(int32_t, const int32_t*, const double*, uint2*, float*, const int32_t*, const int32_t*,
                                          const int32_t*, const int32_t*, const uint64_t*, const int32_t*, const uint8_t*, uint2*, float*);

__global__ __launch_bounds__(256) void k_cvt_f32(int64_t n, const double* __restrict__ x, float* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = (float)x[i];
}

// ============================================================================
// K3: 4x4 diagonal-block inverse (Gauss-Jordan, partial pivoting), one thread per node
// ============================================================================
__global__ __launch_bounds__(256) void k_dinv(int32_t n, const int32_t* __restrict__ diag,
                                              const double* __restrict__ vals, double* __restrict__ dinv) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double A[4][8];
    const double* src = vals + 16 * (int64_t)diag[i];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            A[r][c] = src[4 * r + c];
            A[r][4 + c] = (r == c) ? 1.0 : 0.0;
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int p = k;
        double best = fabs(A[k][k]);
#pragma unroll
        for (int r = k + 1; r < 4; ++r) {
            const double v = fabs(A[r][k]);
            if (v > best) { best = v; p = r; }
        }
#pragma unroll
        for (int r = k + 1; r < 4; ++r)
            if (r == p) {
#pragma unroll
                for (int c = 0; c < 8; ++c) { const double tmp = A[k][c]; A[k][c] = A[r][c]; A[r][c] = tmp; }
            }
        const double ip = (A[k][k] != 0.0) ? 1.0 / A[k][k] : 0.0;
#pragma unroll
        for (int c = 0; c < 8; ++c) A[k][c] *= ip;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r != k) {
                const double f = A[r][k];
#pragma unroll
                for (int c = 0; c < 8; ++c) A[r][c] -= f * A[k][c];
            }
    }
    double* dst = dinv + 16 * (int64_t)i;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[4 * r + c] = A[r][4 + c];
}

// z = omega * Dinv r   (first Jacobi sweep from a zero guess / plain block-Jacobi PC)
__global__ __launch_bounds__(256) void k_bjacobi(int32_t n, const double* __restrict__ dinv,
                                                 const double* __restrict__ r, double omega,
                                                 double* __restrict__ z) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = gid >> 2;
    const int c = (int)(gid & 3);
    if (i >= n) return;
    const double* D = dinv + 16 * i + 4 * c;
    const double* rr = r + 4 * i;
    z[4 * i + c] = omega * (D[0] * rr[0] + D[1] * rr[1] + D[2] * rr[2] + D[3] * rr[3]);
}

// the same with the fp32 copy of D^-1 the low-precision sweeps use (first sweep of a cycle: 128 instead of 192 B per node)
__global__ __launch_bounds__(256) void k_bjacobi32(int32_t n, const float* __restrict__ dinv,
                                                   const double* __restrict__ r, double omega,
                                                   double* __restrict__ z) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = gid >> 2;
    const int c = (int)(gid & 3);
    if (i >= n) return;
    const float4 D = *reinterpret_cast<const float4*>(dinv + 16 * i + 4 * c);
    const double2* rr = reinterpret_cast<const double2*>(r + 4 * i);
    const double2 r01 = rr[0], r23 = rr[1];
    z[4 * i + c] = omega * ((double)D.x * r01.x + (double)D.y * r01.y + (double)D.z * r23.x + (double)D.w * r23.y);
}

// ============================================================================
// K4: vector kernels.  Reductions are two-stage and deterministic: each block
// writes its partial sums to `partial[blockIdx * nred + k]`; k_reduce_final sums
// them in a fixed order into out[k].
// ============================================================================
constexpr int VEC_TPB = 256;

template <int NRED>
__device__ __forceinline__ void block_reduce_store(double (&v)[NRED], double* partial) {
    __shared__ double red[NRED][VEC_TPB / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NRED; ++k) {
        double s = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[k][wv] = s;
    }
    __syncthreads();
    if (threadIdx.x < NRED) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < VEC_TPB / 64; ++w) s += red[threadIdx.x][w];
        partial[(int64_t)blockIdx.x * NRED + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(256) void k_reduce_final(int nblocks, int nred, const double* __restrict__ partial,
                                                      double* __restrict__ out) {
    // one block; thread k < nred sums column k over blocks (strided by 64 lanes per column)
    __shared__ double red[256];
    const int k = blockIdx.x;                   // one block per reduced quantity
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) s += partial[(int64_t)b * nred + k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = red[0];
}

// BiCGStab on one GPU (no all-reduce between the sum and its use): the final stage of the reduction and the scalar update that
// consumes it in ONE single-workgroup launch instead of k_reduce_final + k_bicg_alpha / k_bicg_omega (a dependent launch less
// per half iteration).  The sums are formed exactly as k_reduce_final forms them (same order, same bits).
__device__ void bicg_alpha_update(double* __restrict__ sc, const double* __restrict__ red);
__device__ void bicg_omega_update(double* __restrict__ sc, const double* __restrict__ red);
template <int WHICH>    // 1: nred = 1, alpha;  2: nred = 5, omega / rho / beta / ||r||^2 / flags
__global__ __launch_bounds__(256) void k_reduce_final_bicg(int nblocks, const double* __restrict__ partial,
                                                           double* __restrict__ red_out, double* __restrict__ sc) {
    constexpr int NRED = WHICH == 1 ? 1 : 5;
    __shared__ double red[NRED][256];
    __shared__ double tot[NRED];
    double s[NRED];
#pragma unroll
    for (int k = 0; k < NRED; ++k) s[k] = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x)
#pragma unroll
        for (int k = 0; k < NRED; ++k) s[k] += partial[(int64_t)b * NRED + k];
#pragma unroll
    for (int k = 0; k < NRED; ++k) red[k][threadIdx.x] = s[k];
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {                 // the tree of k_reduce_final, all NRED columns per barrier
        if (threadIdx.x < o)
#pragma unroll
            for (int k = 0; k < NRED; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NRED; ++k) { tot[k] = red[k][0]; red_out[k] = red[k][0]; }
        if (WHICH == 1) bicg_alpha_update(sc, tot);
        else bicg_omega_update(sc, tot);
    }
}
template __global__ void k_reduce_final_bicg<1>(int, const double*, double*, double*);
template __global__ void k_reduce_final_bicg<2>(int, const double*, double*, double*);

// The same launch for a partitioned run over the peer-window transport: local final stage, all-reduce over the ranks (stores
// into the peers' windows, sns_peer_dev.h), scalar update -- where RCCL needs k_reduce_final + ncclAllReduce + k_bicg_alpha.
template <int WHICH>
__global__ __launch_bounds__(256) void k_reduce_final_bicg_peer(int nblocks, const double* __restrict__ partial,
                                                                double* __restrict__ red_out, double* __restrict__ sc,
                                                                PeerArgs pa) {
    constexpr int NRED = WHICH == 1 ? 1 : 5;
    __shared__ double red[NRED][256];
    __shared__ double tot[8];
    if (pa.phase != 2) {                           // (team transport: phase 1 = reduce + contribute, phase 2 = sum + update)
        double s[NRED];
#pragma unroll
        for (int k = 0; k < NRED; ++k) s[k] = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += blockDim.x)
#pragma unroll
            for (int k = 0; k < NRED; ++k) s[k] += partial[(int64_t)b * NRED + k];
#pragma unroll
        for (int k = 0; k < NRED; ++k) red[k][threadIdx.x] = s[k];
        __syncthreads();
        for (int o = blockDim.x / 2; o > 0; o >>= 1) {
            if (threadIdx.x < o)
#pragma unroll
                for (int k = 0; k < NRED; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x < NRED) tot[threadIdx.x] = red[threadIdx.x][0];
        __syncthreads();
    }
    peer_allreduce_block(tot, NRED, pa);
    if (pa.phase == 1) return;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NRED; ++k) red_out[k] = tot[k];
        if (WHICH == 1) bicg_alpha_update(sc, tot);
        else bicg_omega_update(sc, tot);
    }
}
template __global__ void k_reduce_final_bicg_peer<1>(int, const double*, double*, double*, PeerArgs);
template __global__ void k_reduce_final_bicg_peer<2>(int, const double*, double*, double*, PeerArgs);

// first stage of a long reduction: block (c, k) sums partial[b][k] over the 2048 blocks b of chunk c (fixed order)
__global__ __launch_bounds__(256) void k_reduce_chunks(int nblocks, int nred, const double* __restrict__ partial,
                                                       double* __restrict__ out) {
    __shared__ double red[256];
    const int k = blockIdx.y;
    const int b0 = blockIdx.x * 2048, b1 = min(nblocks, b0 + 2048);
    double s = 0.0;
    for (int b = b0 + threadIdx.x; b < b1; b += blockDim.x) s += partial[(int64_t)b * nred + k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(int64_t)blockIdx.x * nred + k] = red[0];
}

// Dirichlet dofs that already agree with the prescribed value to round-off are set to it exactly, so that the
// state keeps qualifying for the lifting-free assembly paths (which test bitwise equality)
__global__ __launch_bounds__(256) void k_snap_bc(int64_t ndof, const uint8_t* __restrict__ bc_mask,
                                                 const double* __restrict__ bc_val, double rel_tol,
                                                 double* __restrict__ w) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndof; i += (int64_t)gridDim.x * blockDim.x)
        if (bc_mask[i]) {
            const double g = bc_val[i];
            if (fabs(w[i] - g) <= rel_tol * fmax(1.0, fabs(g))) w[i] = g;
        }
}

// partial[block] = number of Dirichlet dofs whose current value differs from the prescribed one
__global__ __launch_bounds__(256) void k_count_bc_violations(int64_t ndof, const uint8_t* __restrict__ bc_mask,
                                                             const double* __restrict__ bc_val,
                                                             const double* __restrict__ w, double* __restrict__ partial) {
    double v[1] = {0.0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndof; i += (int64_t)gridDim.x * blockDim.x)
        if (bc_mask[i] && w[i] != bc_val[i]) v[0] += 1.0;
    block_reduce_store<1>(v, partial);
}

// generic: out[0] = x.y, out[1] = y.y  (n = number of doubles reduced over)
__global__ __launch_bounds__(256) void k_dot2(int64_t n, const double* __restrict__ x,
                                              const double* __restrict__ y, double* __restrict__ partial) {
    double v[2] = {0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double a = x[i], b = y[i];
        v[0] += a * b;
        v[1] += b * b;
    }
    block_reduce_store<2>(v, partial);
}

// y = a*x + b*y
__global__ __launch_bounds__(256) void k_axpby(int64_t n, double a, const double* __restrict__ x, double b,
                                               double* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = a * x[i] + (b == 0.0 ? 0.0 : b * y[i]);
}

// z = a*x + b*y + c*z
__global__ __launch_bounds__(256) void k_axpbypcz(int64_t n, double a, const double* __restrict__ x, double b,
                                                  const double* __restrict__ y, double c,
                                                  double* __restrict__ z) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        z[i] = a * x[i] + b * y[i] + (c == 0.0 ? 0.0 : c * z[i]);
}

// ---- BiCGStab with device-resident scalars ------------------------------------------------------------------
// sc[0] = rho = <rhat, r>, sc[1] = alpha, sc[2] = omega, sc[3] = beta, sc[4] = ||r||^2, sc[5] = flags
// (1: NaN/Inf, 2: omega == 0, 4: rho == 0), sc[6] = <rhat, v>.  The host never needs them to launch the next kernel:
// it reads (sc[4], sc[5]) once per iteration, asynchronously, for the stopping test only.
// p = r + beta * (p - omega * v)
__global__ __launch_bounds__(256) void k_bicg_p(int64_t n, const double* __restrict__ r, const double* __restrict__ sc,
                                                const double* __restrict__ v, double* __restrict__ p) {
    const double beta = sc[3], omega = sc[2];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = r[i] + beta * (p[i] - omega * v[i]);
}
// alpha = rho / <rhat, v>
__device__ void bicg_alpha_update(double* __restrict__ sc, const double* __restrict__ red) {
    sc[6] = red[0];
    sc[1] = sc[0] / red[0];
}
__global__ void k_bicg_alpha(double* __restrict__ sc, const double* __restrict__ red) {
    if (threadIdx.x == 0 && blockIdx.x == 0) bicg_alpha_update(sc, red);
}
// s = r - alpha v
__global__ __launch_bounds__(256) void k_bicg_s(int64_t n, const double* __restrict__ r, const double* __restrict__ sc,
                                                const double* __restrict__ v, double* __restrict__ s) {
    const double alpha = sc[1];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s[i] = r[i] - alpha * v[i];
}
// s = r - alpha v together with the FIRST sweep of the V-cycle that s goes into: z = w D^-1 s on the fine level's fp32 D^-1
// copy (what k_bjacobi32 does in a launch of its own right after: same expression on the same operands, bitwise the same z).
// Thread i = dof i, the 4 lanes of a quad = one node, so the node's s is exchanged by DPP.  n is a multiple of 4.
__global__ __launch_bounds__(256) void k_bicg_s_first(int64_t n, const double* __restrict__ r, const double* __restrict__ sc,
                                                      const double* __restrict__ v, double* __restrict__ s,
                                                      const float* __restrict__ dinv32, double omega_pc, double* __restrict__ z) {
    const double alpha = sc[1];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double sv = r[i] - alpha * v[i];
        s[i] = sv;
        const float4 D = *reinterpret_cast<const float4*>(dinv32 + 4 * i);      // row (i % 4) of node (i / 4)'s block
        const double s0 = quad_bcast<0>(sv), s1 = quad_bcast<1>(sv), s2 = quad_bcast<2>(sv), s3 = quad_bcast<3>(sv);
        z[i] = omega_pc * ((double)D.x * s0 + (double)D.y * s1 + (double)D.z * s2 + (double)D.w * s3);
    }
}
// ONE reduction pass for the second half of an iteration: (t.s, t.t, rhat.s, rhat.t, s.s).  omega = t.s / t.t, and
// with r = s - omega t the next rho and the new residual norm follow without touching r:
//   <rhat, r> = rhat.s - omega rhat.t,   ||r||^2 = s.s - 2 omega t.s + omega^2 t.t
__global__ __launch_bounds__(256) void k_bicg_dots5(int64_t n, const double* __restrict__ s, const double* __restrict__ t,
                                                    const double* __restrict__ rhat, double* __restrict__ partial) {
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double sv = s[i], tv = t[i], hv = rhat[i];
        acc[0] += tv * sv;
        acc[1] += tv * tv;
        acc[2] += hv * sv;
        acc[3] += hv * tv;
        acc[4] += sv * sv;
    }
    block_reduce_store<5>(acc, partial);
}
// omega, next rho, beta of the NEXT iteration, ||r||^2 and the flags; out[0..1] = (||r||^2, flags) for the host
__global__ void k_bicg_omega(double* __restrict__ sc, const double* __restrict__ red) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    bicg_omega_update(sc, red);
}
__device__ void bicg_omega_update(double* __restrict__ sc, const double* __restrict__ red) {
    const double ts = red[0], tt = red[1], hs = red[2], ht = red[3], ss = red[4];
    const double omega = (tt > 0.0) ? ts / tt : 0.0;
    const double rho_new = hs - omega * ht;
    double rr = ss - 2.0 * omega * ts + omega * omega * tt;
    const double rho = sc[0], alpha = sc[1];
    double flags = 0.0;
    if (!(rr == rr) || isinf(rr) || !(alpha == alpha)) flags += 1.0;
    if (rr < 0.0) rr = 0.0;                             // round-off of the three-term formula
    if (omega == 0.0) flags += 2.0;
    if (rho_new == 0.0) flags += 4.0;
    sc[3] = (rho_new / rho) * (alpha / omega);
    sc[0] = rho_new;
    sc[2] = omega;
    sc[4] = rr;
    sc[5] = flags;
}
// x += alpha ph + omega sh ; r = s - omega t
__global__ __launch_bounds__(256) void k_bicg_xr(int64_t n, const double* __restrict__ sc, const double* __restrict__ ph,
                                                 const double* __restrict__ sh, const double* __restrict__ s,
                                                 const double* __restrict__ t, double* __restrict__ x,
                                                 double* __restrict__ r) {
    const double alpha = sc[1], omega = sc[2];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        x[i] += alpha * ph[i] + omega * sh[i];
        r[i] = s[i] - omega * t[i];
    }
}
// the same with the NEXT iteration's p = r + beta (p - omega v) in the same pass (r stays in registers): one launch and one
// read of r less per iteration; beta (sc[3]) is already the next iteration's (k_bicg_omega ran before)
__global__ __launch_bounds__(256) void k_bicg_xrp(int64_t n, const double* __restrict__ sc, const double* __restrict__ ph,
                                                  const double* __restrict__ sh, const double* __restrict__ s,
                                                  const double* __restrict__ t, const double* __restrict__ v,
                                                  double* __restrict__ x, double* __restrict__ r, double* __restrict__ p) {
    const double alpha = sc[1], omega = sc[2], beta = sc[3];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        x[i] += alpha * ph[i] + omega * sh[i];
        const double rn = s[i] - omega * t[i];
        r[i] = rn;
        p[i] = rn + beta * (p[i] - omega * v[i]);
    }
}
// ... and with the first sweep of the V-cycle the new p goes into, z = w D^-1 p (see k_bicg_s_first)
__global__ __launch_bounds__(256) void k_bicg_xrp_first(int64_t n, const double* __restrict__ sc, const double* __restrict__ ph,
                                                        const double* __restrict__ sh, const double* __restrict__ s,
                                                        const double* __restrict__ t, const double* __restrict__ v,
                                                        double* __restrict__ x, double* __restrict__ r, double* __restrict__ p,
                                                        const float* __restrict__ dinv32, double omega_pc, double* __restrict__ z) {
    const double alpha = sc[1], omega = sc[2], beta = sc[3];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        x[i] += alpha * ph[i] + omega * sh[i];
        const double rn = s[i] - omega * t[i];
        r[i] = rn;
        const double pv = rn + beta * (p[i] - omega * v[i]);
        p[i] = pv;
        const float4 D = *reinterpret_cast<const float4*>(dinv32 + 4 * i);
        const double p0 = quad_bcast<0>(pv), p1 = quad_bcast<1>(pv), p2 = quad_bcast<2>(pv), p3 = quad_bcast<3>(pv);
        z[i] = omega_pc * ((double)D.x * p0 + (double)D.y * p1 + (double)D.z * p2 + (double)D.w * p3);
    }
}
// sc[0..7] = (rho, alpha, omega, beta, rr, flags, 0, 0) at the start of a solve: rho = rr0 (rhat = r), beta = 0
__global__ void k_bicg_init(double* __restrict__ sc, const double* __restrict__ rr0) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    sc[0] = rr0[0]; sc[1] = 1.0; sc[2] = 1.0; sc[3] = 0.0; sc[4] = rr0[0]; sc[5] = 0.0; sc[6] = 0.0; sc[7] = 0.0;
}

// GMRES: h[k] = V_k . w for k < nv (chunks of 8 basis vectors per pass over w)
__global__ __launch_bounds__(256) void k_multi_dot8(int64_t n, int nv, const double* __restrict__ V, int64_t ldv,
                                                    const double* __restrict__ w, double* __restrict__ partial) {
    double acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double wv = w[i];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < nv) acc[k] += V[k * ldv + i] * wv;
    }
    block_reduce_store<8>(acc, partial);
}

// GMRES: w -= sum_k h[k] V_k (k < nv <= 8, h on device) ; partial = (w.w) when want_norm
__global__ __launch_bounds__(256) void k_multi_axpy8(int64_t n, int nv, const double* __restrict__ V, int64_t ldv,
                                                     const double* __restrict__ h, double sign,
                                                     double* __restrict__ w, double* __restrict__ partial) {
    double hk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) hk[k] = (k < nv) ? sign * h[k] : 0.0;
    double acc[1] = {0.0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double wv = w[i];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < nv) wv += hk[k] * V[k * ldv + i];
        w[i] = wv;
        acc[0] += wv * wv;
    }
    if (partial) block_reduce_store<1>(acc, partial);
}

__global__ __launch_bounds__(256) void k_scale_copy(int64_t n, double a, const double* x, double* y) {  // x may alias y
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = a * x[i];
}

// ============================================================================
// AMG kernels (plain aggregation, 4 dofs per aggregate)
// ============================================================================
// bc[I] = sum_{i in I} free_i * r[i]     (restriction = P0^T, gather => deterministic)
// With dinv32_c the kernel also does the coarse level's FIRST smoothing sweep from a zero guess, z = omega_c Dc^-1 bc (what
// k_bjacobi32 would do in a launch of its own right after: the four components of bc sit in the four lanes of the quad; same
// expression, bitwise the same z) -- one dependent launch less per level and cycle.
__global__ __launch_bounds__(256) void k_restrict(int32_t nc, const int32_t* __restrict__ m_ptr,
                                                  const int32_t* __restrict__ m_idx,
                                                  const uint8_t* __restrict__ free_mask,
                                                  const double* __restrict__ r, double* __restrict__ bc,
                                                  const float* __restrict__ dinv32_c, double omega_c,
                                                  double* __restrict__ z_c) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t I = gid >> 2;
    const int c = (int)(gid & 3);
    if (I >= nc) return;                               // whole quads leave together (nc * 4 threads carry work)
    double s = 0.0;
    const int32_t k0 = m_ptr[I], k1 = m_ptr[I + 1];
    int32_t k = k0;
    // aggregates have up to 8 members: all member ids first, then all residual entries (two round trips instead of
    // one dependent pair per member)
    for (; k + 7 < k1; k += 8) {
        int32_t m[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) m[q] = m_idx[k + q];
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int64_t d = 4 * (int64_t)m[q] + c;
            v[q] = (!free_mask || free_mask[d]) ? r[d] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s += v[q];
    }
    if (k < k1) {
        int32_t m[8];
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) m[q] = (k + q < k1) ? m_idx[k + q] : -1;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int64_t d = 4 * (int64_t)(m[q] < 0 ? 0 : m[q]) + c;
            v[q] = (m[q] >= 0 && (!free_mask || free_mask[d])) ? r[d] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s += v[q];
    }
    bc[4 * I + c] = s;
    if (dinv32_c) {
        const float4 D = *reinterpret_cast<const float4*>(dinv32_c + 16 * I + 4 * c);
        const double s0 = quad_bcast<0>(s), s1 = quad_bcast<1>(s), s2 = quad_bcast<2>(s), s3 = quad_bcast<3>(s);
        z_c[4 * I + c] = omega_c * ((double)D.x * s0 + (double)D.y * s1 + (double)D.z * s2 + (double)D.w * s3);
    }
}
__global__ __launch_bounds__(256) void k_prolong_add(int32_t n, const int32_t* __restrict__ agg,
                                                     const uint8_t* __restrict__ free_mask,
                                                     const double* __restrict__ xc, double* __restrict__ x) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = gid >> 2;
    const int c = (int)(gid & 3);
    if (i >= n) return;
    const int32_t I = agg[i];
    if (I < 0) return;
    if (free_mask && !free_mask[4 * i + c]) return;
    x[4 * i + c] += xc[4 * (int64_t)I + c];
}

// dst block valmap[s] <- src block s (multi-GPU: all-gathered rows of a level scattered into the replicated copy)
__global__ __launch_bounds__(256) void k_scatter_blocks(int64_t nsrc, const int32_t* __restrict__ valmap,
                                                        const double* __restrict__ src, double* __restrict__ dst) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t s = gid >> 3;
    const int t = (int)(gid & 7);
    if (s >= nsrc) return;
    const int32_t d = valmap[s];
    if (d < 0) return;
    reinterpret_cast<double2*>(dst + (int64_t)d * 16)[t] = reinterpret_cast<const double2*>(src + s * 16)[t];
}

// dst row g <- src row rowmap[g] (4 doubles per row)
__global__ __launch_bounds__(256) void k_gather_rows(int32_t n, const int32_t* __restrict__ rowmap,
                                                     const double* __restrict__ src, double* __restrict__ dst) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t g = gid >> 2;
    const int c = (int)(gid & 3);
    if (g >= n) return;
    dst[4 * g + c] = src[4 * (int64_t)rowmap[g] + c];
}

// Galerkin coarse operator A_c = P0^T A P0 as a gather: coarse slot <- sum of fine
// slots; entries whose fine row/col dof is excluded from the transfer are skipped;
// coarse dofs with no free fine dof get a unit diagonal.
__global__ __launch_bounds__(256) void k_galerkin(int64_t nnzb_c, const int64_t* __restrict__ r_ptr,
                                                  const int32_t* __restrict__ r_idx,
                                                  const double* __restrict__ vals_f,
                                                  const int32_t* __restrict__ slot_row_c,
                                                  const int32_t* __restrict__ colind_c,
                                                  const uint8_t* __restrict__ fixed_c,
                                                  const int32_t* __restrict__ m_ptr,
                                                  double* __restrict__ vals_c) {
    // 8 lanes per coarse slot, one double2 each: a fine block is one 128-B line per step, two steps in flight.
    // Dofs excluded from the transfer (level 0: Dirichlet dofs) need no per-entry test: their rows and columns
    // of the assembled operator are zero except the unit diagonal (:74), so the only thing to take out again is
    // one 1.0 per fixed fine dof on the coarse diagonal (fixed_c = number of fixed fine dofs per coarse dof).
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t s = gid >> 3;
    const int t = (int)(gid & 7);
    if (s >= nnzb_c) return;
    double2 v0 = make_double2(0.0, 0.0), v1 = v0;
    const int64_t k1 = r_ptr[s + 1];
    int64_t k = r_ptr[s];
    // 8 fine blocks per step (a fine-level coarse slot gathers ~8): all ids first, then all blocks -- two round trips instead of
    // one dependent pair per two blocks; the even / odd partial sums keep their order (bitwise the result of the loop below)
    for (; k + 7 < k1; k += 8) {
        int32_t f[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = r_idx[k + q];
        double2 a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = ld_stream(reinterpret_cast<const double2*>(vals_f + (int64_t)f[q] * 16) + t);
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            v0.x += a[q].x; v0.y += a[q].y;
            v1.x += a[q + 1].x; v1.y += a[q + 1].y;
        }
    }
    for (; k + 1 < k1; k += 2) {
        const int32_t f0 = r_idx[k], f1 = r_idx[k + 1];
        const double2 a = reinterpret_cast<const double2*>(vals_f + (int64_t)f0 * 16)[t];
        const double2 b = reinterpret_cast<const double2*>(vals_f + (int64_t)f1 * 16)[t];
        v0.x += a.x; v0.y += a.y;
        v1.x += b.x; v1.y += b.y;
    }
    if (k < k1) {
        const double2 a = reinterpret_cast<const double2*>(vals_f + (int64_t)r_idx[k] * 16)[t];
        v0.x += a.x; v0.y += a.y;
    }
    double2 v = make_double2(v0.x + v1.x, v0.y + v1.y);
    if (fixed_c) {
        const int32_t I = slot_row_c[s];
        if (I == colind_c[s]) {
            const int c = t >> 1;                          // lane t holds entries (c, 2*(t&1)) and (c, 2*(t&1)+1)
            const int nfix = fixed_c[4 * (int64_t)I + c];
            const bool all_fixed = nfix == m_ptr[I + 1] - m_ptr[I];     // no free fine dof: unit diagonal
            if ((t & 1) == (c >> 1)) {
                if (c & 1) v.y = all_fixed ? 1.0 : v.y - (double)nfix;
                else v.x = all_fixed ? 1.0 : v.x - (double)nfix;
            }
        }
    }
    reinterpret_cast<double2*>(vals_c + s * 16)[t] = v;
}

// fixed_c[4I+c] = number of fine dofs of component c in aggregate I that are excluded from the transfer
__global__ __launch_bounds__(256) void k_empty_coarse(int32_t nc, const int32_t* __restrict__ m_ptr,
                                                      const int32_t* __restrict__ m_idx,
                                                      const uint8_t* __restrict__ free_mask,
                                                      uint8_t* __restrict__ empty_c) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t I = gid >> 2;
    const int c = (int)(gid & 3);
    if (I >= nc) return;
    int nfix = 0;
    for (int32_t k = m_ptr[I]; k < m_ptr[I + 1]; ++k) nfix += free_mask[4 * (int64_t)m_idx[k] + c] ? 0 : 1;
    empty_c[4 * I + c] = (uint8_t)nfix;
}

// dense N x N (N = 4n) copy of a small BSR matrix
__global__ __launch_bounds__(256) void k_bsr_to_dense(int32_t n, const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ colind,
                                                      const double* __restrict__ vals, double* __restrict__ D) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t s = gid >> 4;
    const int e = (int)(gid & 15);
    if (s >= rowptr[n]) return;
    // find row by linear search over a tiny matrix
    int32_t row = 0;
    while (rowptr[row + 1] <= s) ++row;
    const int N = 4 * n;
    D[(int64_t)(4 * row + (e >> 2)) * N + 4 * colind[s] + (e & 3)] = vals[s * 16 + e];
}

// In-place Gauss-Jordan inverse with partial pivoting, ONE workgroup, matrix in
// global memory (L2-resident: N <= 160).  Only used on the coarsest AMG level.
__global__ __launch_bounds__(1024) void k_dense_inverse(int N, double* __restrict__ A, int* __restrict__ piv,
                                                        int* __restrict__ singular) {
    __shared__ int s_p;
    __shared__ double s_best[1024];
    __shared__ int s_arg[1024];
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int k = 0; k < N; ++k) {
        // pivot search in column k, rows >= k
        double best = -1.0;
        int arg = k;
        for (int r = k + tid; r < N; r += nt) {
            const double v = fabs(A[(int64_t)r * N + k]);
            if (v > best) { best = v; arg = r; }
        }
        s_best[tid] = best;
        s_arg[tid] = arg;
        __syncthreads();
        for (int o = nt / 2; o > 0; o >>= 1) {
            if (tid < o) {
                if (s_best[tid + o] > s_best[tid] ||
                    (s_best[tid + o] == s_best[tid] && s_arg[tid + o] < s_arg[tid])) {
                    s_best[tid] = s_best[tid + o];
                    s_arg[tid] = s_arg[tid + o];
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            s_p = s_arg[0];
            piv[k] = s_arg[0];
            if (!(s_best[0] > 0.0)) *singular = 1;
        }
        __syncthreads();
        const int p = s_p;
        if (p != k)
            for (int c = tid; c < N; c += nt) {
                const double tmp = A[(int64_t)k * N + c];
                A[(int64_t)k * N + c] = A[(int64_t)p * N + c];
                A[(int64_t)p * N + c] = tmp;
            }
        __threadfence_block();
        __syncthreads();
        const double pivv = A[(int64_t)k * N + k];
        const double ip = (pivv != 0.0) ? 1.0 / pivv : 0.0;
        __syncthreads();
        // scale pivot row; A[k][k] <- 1/pivot
        for (int c = tid; c < N; c += nt) A[(int64_t)k * N + c] = (c == k) ? ip : A[(int64_t)k * N + c] * ip;
        __threadfence_block();
        __syncthreads();
        // eliminate column k from all other rows
        for (int idx = tid; idx < N * N; idx += nt) {
            const int r = idx / N, c = idx - r * N;
            if (r == k) continue;
            const double f = A[(int64_t)r * N + k];
            // every thread of row r needs the OLD A[r][k]; column k itself is rewritten last (below)
            if (c != k) A[(int64_t)r * N + c] -= f * A[(int64_t)k * N + c];
        }
        __threadfence_block();
        __syncthreads();
        for (int r = tid; r < N; r += nt)
            if (r != k) A[(int64_t)r * N + k] = -A[(int64_t)r * N + k] * ip;
        __threadfence_block();
        __syncthreads();
    }
    // undo the row interchanges as column interchanges, in reverse order
    for (int k = N - 1; k >= 0; --k) {
        const int p = piv[k];
        if (p != k)
            for (int r = tid; r < N; r += nt) {
                const double tmp = A[(int64_t)r * N + k];
                A[(int64_t)r * N + k] = A[(int64_t)r * N + p];
                A[(int64_t)r * N + p] = tmp;
            }
        __threadfence_block();
        __syncthreads();
    }
}

// y[0..nrows) = D[0..nrows, :] x, dense with N columns, one wave per row
__global__ __launch_bounds__(256) void k_dense_matvec(int N, const double* __restrict__ D,
                                                      const double* __restrict__ x, double* __restrict__ y,
                                                      int nrows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= nrows) return;
    double s = 0.0;
    for (int c = lane; c < N; c += 64) s += D[(int64_t)row * N + c] * x[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[row] = s;
}

// owned rows of a small distributed BSR matrix -> rows of the global dense matrix (N columns),
// column block of local node j = colmap[j]
__global__ __launch_bounds__(256) void k_bsr_to_dense_map(int32_t n_rows, const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ colind,
                                                          const double* __restrict__ vals,
                                                          const int32_t* __restrict__ colmap, int N,
                                                          double* __restrict__ D) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t s = gid >> 4;
    const int e = (int)(gid & 15);
    if (s >= rowptr[n_rows]) return;
    int32_t row = 0;
    while (rowptr[row + 1] <= s) ++row;
    D[(int64_t)(4 * row + (e >> 2)) * N + 4 * colmap[colind[s]] + (e & 3)] = vals[s * 16 + e];
}

// rows [r0, r1) of a row block whose first row is global row g0: unit diagonal (padding dofs)
__global__ __launch_bounds__(256) void k_pad_identity(int r0, int r1, int g0, int N, double* __restrict__ D) {
    for (int r = r0 + threadIdx.x; r < r1; r += blockDim.x) D[(int64_t)r * N + g0 + r] = 1.0;
}

// ============================================================================
// K7: halo pack / unpack (4 doubles per node)
// ============================================================================
__global__ __launch_bounds__(256) void k_pack(int32_t m, const int32_t* __restrict__ idx,
                                              const double* __restrict__ x, double* __restrict__ buf) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= 4 * (int64_t)m) return;
    buf[gid] = x[4 * (int64_t)idx[gid >> 2] + (gid & 3)];
}
__global__ __launch_bounds__(256) void k_unpack(int32_t m, const int32_t* __restrict__ idx,
                                                const double* __restrict__ buf, double* __restrict__ x) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= 4 * (int64_t)m) return;
    x[4 * (int64_t)idx[gid >> 2] + (gid & 3)] = buf[gid];
}

// deterministic rough start vector for the power iteration (all ones plus a cheap hash ripple)
__global__ __launch_bounds__(256) void k_fill_pattern(int64_t n, double* __restrict__ x) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned hsh = (unsigned)(i * 2654435761u) >> 22;          // 10 bits
        x[i] = 1.0 + ((double)hsh - 512.0) * (1.0 / 1024.0);
    }
}
// y = x / sqrt(*s2)   (s2 = squared norm on the device; avoids a host round trip per power iteration)
__global__ __launch_bounds__(256) void k_scale_by_rsqrt(int64_t n, const double* __restrict__ s2,
                                                        const double* __restrict__ x, double* __restrict__ y) {
    const double s = *s2;
    const double a = s > 0.0 ? 1.0 / sqrt(s) : 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = a * x[i];
}

__global__ __launch_bounds__(256) void k_fill_slot_row(int32_t n, const int32_t* __restrict__ rowptr,
                                                       int32_t* __restrict__ slot_row) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) slot_row[k] = i;
}

}  // namespace sns
