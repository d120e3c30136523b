// Dense coarsest level of the AMG hierarchy (round 4): the first level with at most amg_dense_rows block rows (<= 2048 dofs by
// default) is not smoothed and coarsened further but solved EXACTLY, by one dense matvec with its explicit inverse -- one
// launch in place of the ~15 dependent, purely latency-bound launches of the three deepest levels of the V-cycle (475 / 89 / 19
// rows on the 10 M-tet duct and on its 1/8 slab alike), which do not shrink when the mesh is split over more GPUs.
//
// The inverse is rebuilt at every numeric setup (once per Newton iteration), so it has to be fast: a BLOCKED, in-place
// Gauss-Jordan elimination with 64 x 64 blocks whose rank-64 updates run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64) --
// the one GEMM-shaped piece of this code base: 2 N^3 = 17 GFLOP at N = 2048.  Two launches per block step p:
//   k_gj_panel   R_j   = P^-1 A[p, j]  for every block column j != p      (P = A[p, p] after the updates of steps < p)
//                CpT_i = A[i, p]^T     for every block row i != p          (copies: the update below works in place)
//   k_gj_update  A[i, j] -= A[i, p] R_j;  A[i, p] = -A[i, p] P^-1;  A[p, j] = R_j;  A[p, p] = P^-1,
//                and the workgroup of tile (p+1, p+1) inverts its freshly updated tile in LDS: the next step's P^-1.
// (A two-stream schedule -- bulk update beside the pivot chain -- exists as an experiment: dense_gj_inverse.)
// No pivoting across blocks (and none inside: the diagonal tiles are inverted by a block Gauss-Jordan of their own, see invert64): the level
// operators are Galerkin projections of the stabilised form, whose symmetric part is positive definite (viscous + SUPG/LSIC
// terms on the velocity block, the PSPG Laplacian on the pressure block; Dirichlet and empty coarse dofs are identity rows), so
// every leading principal block is nonsingular and the element growth of the elimination is bounded by the ratio of the skew
// to the symmetric part -- 1e2..1e4 on the convection-dominated coarse levels, harmless in fp64.  A pivot that is zero or not
// finite raises *singular, and the numeric setup returns SNS_E_STATE on EVERY rank (pc_setup sums the flag over the ranks; there is no
// fallback hierarchy: the message names the options -- amg_dense_rows = 0 / amg_block_smooth = 0 -- that take the dense inverses out).
// Operands of a tile product sit in LDS "k-major" ([k][m] and [k][n], row stride LDS_LD doubles), so that the 16x16x4 fragment
// reads -- lane l takes element (k0 + l / 16, 16 w + l % 16) -- are contiguous per 16 lanes.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "sns_kernels.h"

namespace sns {

namespace {

constexpr int GB = 64;              // block size of the elimination
constexpr int LDS_LD = 72;          // row stride of an operand tile in LDS (doubles): two workgroups per CU (2 x 73.7 KB); 80 would be
                                    // free of bank conflicts on the fragment reads (72: two-way on half of a half-wave's lanes) but
                                    // leaves room for one workgroup only
constexpr int INV_LD = 65;          // row stride of the tile being inverted

typedef double d4_t __attribute__((ext_vector_type(4)));

// 64 x 64 tile, row-major with leading dimension ld, global -> LDS (row stride lds_ld); 256 threads, 16-B loads, a wave reads
// two whole rows per instruction
__device__ __forceinline__ void tile_to_lds(double* __restrict__ lds, int lds_ld, const double* __restrict__ g, int64_t ld) {
    const int t = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = q * 256 + t;
        const int r = e >> 5, c = (e & 31) * 2;
        const double2 v = *reinterpret_cast<const double2*>(g + (int64_t)r * ld + c);
        lds[r * lds_ld + c] = v.x;
        lds[r * lds_ld + c + 1] = v.y;
    }
}

// acc[jb] (jb = 0..3) = rows [16 w, 16 w + 16) of At^T B, columns [16 jb, 16 jb + 16); At and B k-major in LDS.
// Element reg of acc[jb] on lane l is (row 16 w + l / 16 + 4 reg, column 16 jb + l % 16) (cdna_hip_programming.md, f64 MFMA map).
__device__ __forceinline__ void tile_mfma(const double* __restrict__ At, const double* __restrict__ B, d4_t (&acc)[4]) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int kk = l >> 4, mm = l & 15;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) acc[jb] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k0 = 0; k0 < GB; k0 += 4) {
        const double a = At[(k0 + kk) * LDS_LD + 16 * w + mm];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const double b = B[(k0 + kk) * LDS_LD + 16 * jb + mm];
            acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[jb], 0, 0, 0);
        }
    }
}

// Inverse of a 64 x 64 tile without pivoting, as a 4 x 4 BLOCK Gauss-Jordan elimination with 16 x 16 blocks: the pivot block is
// inverted by ONE wave in registers (lane l holds column l % 16 of the rows l / 16 + 4 q, i.e. the f64 MFMA C / D layout; the
// pivot row and column travel by ds_bpermute, no barrier: 0.2 us per scalar pivot), the 15 block products of a block step run
// on v_mfma_f64_16x16x4_f64, 4 barriers per block step.  Measured (scripts/r4_micro/gj_tile_bench.hip): the scalar version this
// replaces -- the tile in the registers of 4 waves, pivot row / column through LDS, one barrier per pivot -- took 42 us per tile,
// 23 us of it instruction issue (one wave per SIMD) and 18 us LDS exchange + barrier; the pivot chain of the diagonal tiles is
// the critical path of the whole elimination (30 tiles at N = 1900).
// The tile is in M (row stride INV_LD) on entry, the inverse in M on exit; X: 7 scratch blocks of 16 x SB_LD doubles.
constexpr int SB_LD = 17;

__device__ __forceinline__ void inv16_wave(double (&a)[4], int l, int* __restrict__ singular) {
    const int c = l & 15, g = l >> 4;
#pragma unroll 1
    for (int k = 0; k < 16; ++k) {
        const int kq = k >> 2, kg = k & 3;
        double sel = a[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) sel = (q == kq) ? a[q] : sel;
        const double prow = __shfl(sel, 16 * kg + c);              // M[k][c]
        const double piv = __shfl(sel, 16 * kg + k);               // M[k][k]
        if (l == 0 && !(fabs(piv) > 1e-300 && fabs(piv) < 1e300)) *singular = 1;
        double d = __builtin_amdgcn_rcp(piv);
        d = d * (2.0 - piv * d);
        d = d * (2.0 - piv * d);
        const double pk = (c == k) ? d : prow * d;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double f = __shfl(a[q], 16 * g + k);             // M[g + 4 q][k]
            const double upd = (c == k) ? -f * d : a[q] - f * pk;
            a[q] = (g == kg && q == kq) ? pk : upd;
        }
    }
}
// acc (C layout: row l / 16 + 4 reg, column l % 16) += sgn * A B; A, B: 16 x 16 blocks in LDS with row strides lda / ldb
__device__ __forceinline__ d4_t mma16(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb, d4_t acc,
                                      double sgn, int l) {
    const int c = l & 15, g = l >> 4;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const double a = sgn * A[c * lda + g + 4 * s4];
        const double b = B[(g + 4 * s4) * ldb + c];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ d4_t ld_c(const double* __restrict__ X, int ld, int l) {
    const int c = l & 15, g = l >> 4;
    return d4_t{X[g * ld + c], X[(g + 4) * ld + c], X[(g + 8) * ld + c], X[(g + 12) * ld + c]};
}
__device__ __forceinline__ void st_c(double* __restrict__ X, int ld, d4_t v, int l) {
    const int c = l & 15, g = l >> 4;
    X[g * ld + c] = v[0]; X[(g + 4) * ld + c] = v[1]; X[(g + 8) * ld + c] = v[2]; X[(g + 12) * ld + c] = v[3];
}
__device__ __forceinline__ void invert64(double* __restrict__ M, double* __restrict__ X, int* __restrict__ singular) {
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    double* __restrict__ Dinv = X;
    for (int kb = 0; kb < 4; ++kb) {
        if (w == 0) {                                              // the pivot block: one wave, registers
            const d4_t v = ld_c(M + (16 * kb) * INV_LD + 16 * kb, INV_LD, l);
            double a[4] = {v[0], v[1], v[2], v[3]};
            inv16_wave(a, l, singular);
            st_c(Dinv, SB_LD, d4_t{a[0], a[1], a[2], a[3]}, l);
        }
        __syncthreads();
        // R_j = D^-1 M[kb][j] (3 blocks), Cn_i = -M[i][kb] D^-1 (3 blocks): wave w takes products w and w + 4
        for (int pidx = w; pidx < 6; pidx += 4) {
            const int o = pidx % 3;
            const int idx = o + (o >= kb ? 1 : 0);                 // the o-th block index != kb
            d4_t acc = d4_t{0.0, 0.0, 0.0, 0.0};
            if (pidx < 3) {
                acc = mma16(Dinv, SB_LD, M + (16 * kb) * INV_LD + 16 * idx, INV_LD, acc, 1.0, l);
                st_c(X + (1 + o) * 16 * SB_LD, SB_LD, acc, l);
            } else {
                acc = mma16(M + (16 * idx) * INV_LD + 16 * kb, INV_LD, Dinv, SB_LD, acc, -1.0, l);
                st_c(X + (4 + o) * 16 * SB_LD, SB_LD, acc, l);
            }
        }
        __syncthreads();
        // M[i][j] -= M[i][kb] R_j for i, j != kb: 9 products
        for (int pidx = w; pidx < 9; pidx += 4) {
            const int oi = pidx / 3, oj = pidx % 3;
            const int i = oi + (oi >= kb ? 1 : 0), j = oj + (oj >= kb ? 1 : 0);
            double* __restrict__ Mij = M + (16 * i) * INV_LD + 16 * j;
            d4_t acc = ld_c(Mij, INV_LD, l);
            acc = mma16(M + (16 * i) * INV_LD + 16 * kb, INV_LD, X + (1 + oj) * 16 * SB_LD, SB_LD, acc, -1.0, l);
            st_c(Mij, INV_LD, acc, l);
        }
        __syncthreads();
        // block row kb <- R_j, block column kb <- Cn_i, pivot block <- D^-1
        for (int e = t; e < 7 * 256; e += 256) {
            const int blk = e >> 8, r = (e >> 4) & 15, c = e & 15;
            const double v = X[blk * 16 * SB_LD + r * SB_LD + c];
            if (blk == 0) M[(16 * kb + r) * INV_LD + 16 * kb + c] = v;
            else if (blk < 4) { const int o = blk - 1, j = o + (o >= kb ? 1 : 0); M[(16 * kb + r) * INV_LD + 16 * j + c] = v; }
            else { const int o = blk - 4, i = o + (o >= kb ? 1 : 0); M[(16 * i + r) * INV_LD + 16 * kb + c] = v; }
        }
        __syncthreads();
    }
}

// write the inverse held in M0 as Pinv (row-major 64 x 64) and PinvT
__device__ __forceinline__ void store_pinv(const double* __restrict__ M0, double* __restrict__ Pinv, double* __restrict__ PinvT) {
    const int t = threadIdx.x;
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
        const int e = q * 256 + t;
        const int r = e >> 6, c = e & 63;
        Pinv[e] = M0[r * INV_LD + c];
        PinvT[e] = M0[c * INV_LD + r];
    }
}

}  // namespace

// first pivot block: P^-1 of A[0, 0]
__global__ __launch_bounds__(256) void k_gj_first(int Np, const double* __restrict__ A, double* __restrict__ Pinv,
                                                  double* __restrict__ PinvT, int* __restrict__ singular) {
    __shared__ double M[GB * INV_LD + 7 * 16 * SB_LD];
    tile_to_lds(M, INV_LD, A, Np);
    __syncthreads();
    invert64(M, M + GB * INV_LD, singular);
    store_pinv(M, Pinv, PinvT);
}

// block step p, part 1: workgroups [0, nb): R_j = P^-1 A[p, j]; workgroups [nb, 2 nb): CpT_i = A[i, p]^T
__global__ __launch_bounds__(256) void k_gj_panel(int Np, int p, const double* __restrict__ A, const double* __restrict__ PinvT,
                                                  double* __restrict__ R, double* __restrict__ CpT) {
    __shared__ double S[2 * GB * LDS_LD];
    const int nb = Np / GB;
    const int t = threadIdx.x;
    if ((int)blockIdx.x < nb) {
        const int j = blockIdx.x;
        if (j == p) return;
        double* At = S;
        double* Bt = S + GB * LDS_LD;
        tile_to_lds(At, LDS_LD, PinvT, GB);                                          // At[k][m] = P^-1[m][k]
        tile_to_lds(Bt, LDS_LD, A + (int64_t)p * GB * Np + (int64_t)j * GB, Np);     // B[k][n] = A[p, j]
        __syncthreads();
        d4_t acc[4];
        tile_mfma(At, Bt, acc);
        const int w = t >> 6, l = t & 63;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                R[(int64_t)(16 * w + (l >> 4) + 4 * reg) * Np + (int64_t)j * GB + 16 * jb + (l & 15)] = acc[jb][reg];
    } else {
        const int i = blockIdx.x - nb;
        if (i == p) return;
        tile_to_lds(S, INV_LD, A + (int64_t)i * GB * Np + (int64_t)p * GB, Np);      // S[m][k] = A[i, p]
        __syncthreads();
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int e = q * 256 + t;
            const int k = e >> 6, m = e & 63;
            CpT[(int64_t)k * Np + (int64_t)i * GB + m] = S[m * INV_LD + k];
        }
    }
}

// block step p, part 2: one workgroup per tile (i, j).  `part` splits the step for the two-stream schedule of dense_gj_inverse:
// 1 = only the tiles of block row / column p + 1 (what the NEXT step's panel and pivot need: the critical path), 2 = all the
// others, 0 = every tile.
__global__ __launch_bounds__(256) void k_gj_update(int Np, int p, int part, double* __restrict__ A, const double* __restrict__ Pinv,
                                                   const double* __restrict__ R, const double* __restrict__ CpT,
                                                   double* __restrict__ Pinv_next, double* __restrict__ PinvT_next,
                                                   int* __restrict__ singular) {
    __shared__ double S[2 * GB * LDS_LD];                 // operand tiles; reused by the inversion of the next pivot tile
    const int nb = Np / GB;
    const int i = blockIdx.x / nb, j = blockIdx.x - i * nb;
    if (part != 0 && ((i == p + 1 || j == p + 1) != (part == 1))) return;
    const int t = threadIdx.x;
    double* Aij = A + (int64_t)i * GB * Np + (int64_t)j * GB;
    if (i == p) {                                         // pivot block row: R_j, resp. P^-1 on the diagonal
        const double* src = (j == p) ? Pinv : R + (int64_t)j * GB;
        const int64_t lds = (j == p) ? GB : Np;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = q * 256 + t;
            const int r = e >> 5, c = (e & 31) * 2;
            *reinterpret_cast<double2*>(Aij + (int64_t)r * Np + c) = *reinterpret_cast<const double2*>(src + (int64_t)r * lds + c);
        }
        return;
    }
    double* At = S;
    double* Bt = S + GB * LDS_LD;
    tile_to_lds(At, LDS_LD, CpT + (int64_t)i * GB, Np);                              // At[k][m] = A[i, p][m][k]
    if (j == p) tile_to_lds(Bt, LDS_LD, Pinv, GB);
    else tile_to_lds(Bt, LDS_LD, R + (int64_t)j * GB, Np);
    __syncthreads();
    d4_t acc[4];
    tile_mfma(At, Bt, acc);
    const int w = t >> 6, l = t & 63;
    const bool next_pivot = (i == p + 1 && j == p + 1);
    if (next_pivot) __syncthreads();                      // everyone is done with the operand tiles: S becomes the inversion buffer
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = 16 * w + (l >> 4) + 4 * reg, c = 16 * jb + (l & 15);
            double* dst = Aij + (int64_t)r * Np + c;
            const double v = (j == p) ? -acc[jb][reg] : *dst - acc[jb][reg];
            *dst = v;
            if (next_pivot) S[r * INV_LD + c] = v;
        }
    if (next_pivot) {
        __syncthreads();
        invert64(S, S + GB * INV_LD, singular);
        store_pinv(S, Pinv_next, PinvT_next);
    }
}

// BSR level operator -> dense row-major fp64 matrix with leading dimension Np (zeroed beforehand), 16 lanes per block
__global__ __launch_bounds__(256) void k_bsr_to_dense_ld(int64_t nnzb, const int32_t* __restrict__ slot_row,
                                                         const int32_t* __restrict__ colind, const double* __restrict__ vals,
                                                         int Np, double* __restrict__ D) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t s = gid >> 4;
    const int e = (int)(gid & 15);
    if (s >= nnzb) return;
    D[(int64_t)(4 * slot_row[s] + (e >> 2)) * Np + 4 * colind[s] + (e & 3)] = vals[s * 16 + e];
}
// unit diagonal on the padding dofs [N, Np)
__global__ __launch_bounds__(256) void k_dense_pad_diag(int N, int Np, double* __restrict__ D) {
    const int r = N + blockIdx.x * blockDim.x + threadIdx.x;
    if (r < Np) D[(int64_t)r * Np + r] = 1.0;
}
// the finished inverse as fp32 (same leading dimension): what the cycle's matvec streams
__global__ __launch_bounds__(256) void k_dense_to_f32(int64_t n, const double* __restrict__ A, float* __restrict__ X) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const double2 a = *reinterpret_cast<const double2*>(A + i), b = *reinterpret_cast<const double2*>(A + i + 2);
        *reinterpret_cast<float4*>(X + i) = make_float4((float)a.x, (float)a.y, (float)b.x, (float)b.y);
    } else {
        for (int64_t q = i; q < n; ++q) X[q] = (float)A[q];
    }
}

// y[0..N) = X[0..N, 0..N) b, X fp32 row-major with leading dimension Np (a multiple of 64), one wave per row, fp64 accumulation;
// the whole row and its share of b are requested before the first product.  Columns >= N (padding) are never read from b.
__global__ __launch_bounds__(256) void k_dense_matvec32(int N, int Np, const float* __restrict__ X, const double* __restrict__ b,
                                                        double* __restrict__ y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    const float4* __restrict__ xr = reinterpret_cast<const float4*>(X + (int64_t)row * Np);
    double s0 = 0.0, s1 = 0.0;
    for (int c4 = lane; 4 * c4 < N; c4 += 64) {
        const float4 a = xr[c4];
        const int c = 4 * c4;
        if (c + 3 < N) {
            const double2 b0 = *reinterpret_cast<const double2*>(b + c), b1 = *reinterpret_cast<const double2*>(b + c + 2);
            s0 += (double)a.x * b0.x + (double)a.z * b1.x;
            s1 += (double)a.y * b0.y + (double)a.w * b1.y;
        } else {
            s0 += (double)a.x * b[c];
            if (c + 1 < N) s1 += (double)a.y * b[c + 1];
            if (c + 2 < N) s0 += (double)a.z * b[c + 2];
        }
    }
    double s = s0 + s1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[row] = s;
}

// Enqueue the blocked Gauss-Jordan inverse of the Np x Np matrix A (Np a multiple of 64, in place) on stream s.
// work: 4 * 64 * Np doubles (R and CpT, double-buffered over the steps) + 4 * 4096 doubles (P^-1 and its transpose, likewise).
// The pivot chain -- update of block row / column p + 1, inversion of the next pivot tile, next panel -- is the critical path
// (40 of the 60 us of a step at N = 1900); with a second stream `side` the bulk of every step's update runs beside it
// (EXPERIMENT, off by default, SNS_GJ_TWO_STREAMS=1: measured slower, 3.9 against 2.4 ms at N = 1900 -- the bulk launch's 900
// workgroups occupy every CU and the few workgroups of the chain wait for a slot instead of overtaking them):
//   s:    panel(p) -> [U2(p-1) done] -> U1(p): tiles of row / column p + 1, then the next pivot's inverse
//   side: [panel(p) done] -> U2(p): every other tile
// (U1(p) needs the tiles U2(p-1) wrote; U2(p) needs R / CpT of step p and, through the panel, U1(p-1).)  side == nullptr: one stream.
void dense_gj_inverse(hipStream_t s, hipStream_t side, int Np, double* A, double* work, int* singular) {
    const int nb = Np / GB;
    double* RC = work;                                    // [2][R 64 x Np | CpT 64 x Np]
    double* P = work + (size_t)4 * GB * Np;               // [2][Pinv 4096 | PinvT 4096]
    hipEvent_t e_panel = nullptr, e_bulk = nullptr, e_start = nullptr;
    const bool two = side != nullptr && nb > 2 && hipEventCreateWithFlags(&e_panel, hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&e_bulk, hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&e_start, hipEventDisableTiming) == hipSuccess;
    hipLaunchKernelGGL(k_gj_first, dim3(1), dim3(256), 0, s, Np, A, P, P + 4096, singular);
    if (two) {                                            // whatever `side` did before must not overtake the matrix's producers on s
        (void)hipEventRecord(e_start, s);
        (void)hipStreamWaitEvent(side, e_start, 0);
    }
    for (int p = 0; p < nb; ++p) {
        double* cur = P + (size_t)(p & 1) * 8192;
        double* nxt = P + (size_t)((p + 1) & 1) * 8192;
        double* R = RC + (size_t)(p & 1) * 2 * GB * Np;
        double* CpT = R + (size_t)GB * Np;
        if (nb > 1) hipLaunchKernelGGL(k_gj_panel, dim3(2 * nb), dim3(256), 0, s, Np, p, A, cur + 4096, R, CpT);
        if (!two) {
            hipLaunchKernelGGL(k_gj_update, dim3(nb * nb), dim3(256), 0, s, Np, p, 0, A, cur, R, CpT, nxt, nxt + 4096, singular);
            continue;
        }
        (void)hipEventRecord(e_panel, s);
        (void)hipStreamWaitEvent(side, e_panel, 0);
        hipLaunchKernelGGL(k_gj_update, dim3(nb * nb), dim3(256), 0, side, Np, p, 2, A, cur, R, CpT, nxt, nxt + 4096, singular);
        if (p > 0) (void)hipStreamWaitEvent(s, e_bulk, 0);              // U2(p - 1): recorded below in the previous round
        hipLaunchKernelGGL(k_gj_update, dim3(nb * nb), dim3(256), 0, s, Np, p, 1, A, cur, R, CpT, nxt, nxt + 4096, singular);
        (void)hipEventRecord(e_bulk, side);
    }
    if (two) {
        (void)hipStreamWaitEvent(s, e_bulk, 0);
        (void)hipEventDestroy(e_panel);
        (void)hipEventDestroy(e_bulk);
        (void)hipEventDestroy(e_start);
    }
}

}  // namespace sns
