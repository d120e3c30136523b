// Aggregate-block Jacobi smoother for the coarse levels of the AMG hierarchy (round 4).
//
// The damped point-block (4 x 4 per node) Jacobi sweep is what limits the plain-aggregation V-cycle on this non-symmetric,
// saddle-point-like operator: level 1 needs 1 + 6 sweeps, level 2 6 + 6, and below the fine level every
// sweep is a dependent, latency-bound launch.  Here the block of the Jacobi splitting is a whole AGGREGATE -- the <= 8 nodes
// (32 dofs) that become one node of the next level:
//     x <- x + w B^-1 (b - A x),      B = blockdiag over the aggregates of A,
// one sweep = one launch like before, but it is worth about two point sweeps (oracle/experiments/r4_block_smoother.py: level 1
// 1 + 3 and level 2 2 + 2 sweeps reach the iteration count of 1 + 6 and 6 + 6).
//
// Layout: the 32 lanes of a half-wave take one aggregate -- 4 lanes per member row exactly as in k_spmv_lp (lane 4 q + c:
// member q, dof c), the member rows through the padded list blk_rows[8 G + q] (-1: the aggregate has fewer members).  Each
// quad computes its row's residual with the level's low-precision matrix copy (same stepped loop as k_spmv_lp), the 32
// residual entries meet in LDS and lane j multiplies them with row j of the aggregate's inverse B_G^-1 (fp32, 4 KiB per
// aggregate, stored [k/4][j] as float4 so that a half-wave reads 512 contiguous bytes per instruction; requested before the
// row loop).  Vectors and arithmetic stay fp64: the cycle remains a fixed linear operator.
// B_G^-1 is rebuilt at every numeric setup by k_binv (gather of the aggregate's diagonal block from the fp64 level operator,
// 32 x 32 Gauss-Jordan in LDS without pivoting -- principal submatrices of an operator with positive definite symmetric part).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "sns_kernels.h"

namespace sns {

namespace {

typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

// workgroup b of nb -> the b-th workgroup of ITS XCD's contiguous eighth of the grid (hardware deals consecutive workgroup ids round-robin
// to the 8 XCDs): neighbouring aggregates, whose x gathers overlap, share an L2 (as xcd_remap of sns_kernels.hip)
__device__ __forceinline__ int xcd_remap_b(int b, int nb) {
    const int q = nb >> 3, r = nb & 7;
    const int xcd = b & 7, k = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

template <int CTRL>
__device__ __forceinline__ double quad_perm_b(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int J>
__device__ __forceinline__ double qb(double v) { return quad_perm_b<J * 0x55>(v); }
template <int J>
__device__ __forceinline__ double2 qb2(const double2 v) { return make_double2(qb<J>(v.x), qb<J>(v.y)); }

__device__ __forceinline__ double lp_dot_b(const float4 a, const double2 xa, const double2 xb) {
    return (double)a.x * xa.x + (double)a.y * xa.y + (double)a.z * xb.x + (double)a.w * xb.y;
}
__device__ __forceinline__ double lp_dot_b(const f16x4_t a, const double2 xa, const double2 xb) {
    return (double)(float)a.x * xa.x + (double)(float)a.y * xa.y + (double)(float)a.z * xb.x + (double)(float)a.w * xb.y;
}

// sum_k A_rk x_k for dof row r (= lane & 3) of the block row [s, e) of a low-precision level matrix (FMT 1: fp32 blocks, FMT 2:
// fp16 pair-interleaved blocks; the row scale of FMT 2 is applied by the caller).  The stepped loop of k_spmv_lp: one index load
// per quad and step of 4 blocks (requested one step ahead), the x gather as whole blocks handed round by DPP.  s, e are quad-uniform.
// GH: the ghost columns' x blocks come from the level's receive window (window transports: GhostSrc / GhostReader, sns_peer_dev.h);
// the wave that meets the first ghost column waits for the neighbours' arrival flags.
template <int FMT, int GH = 0>
__device__ __forceinline__ double lp_row_times_x(int32_t s, int32_t e, const int32_t* __restrict__ colind,
                                                 const void* __restrict__ vals_v, const double* __restrict__ x, int r,
                                                 GhostReader* gr = nullptr, const GhostSrc* gs = nullptr) {
    double acc0 = 0.0, acc1 = 0.0;
    const float4* __restrict__ v32 = reinterpret_cast<const float4*>(vals_v) + ((int64_t)s * 4 + r);
    const uint4* __restrict__ v16 = reinterpret_cast<const uint4*>(vals_v) + ((int64_t)s * 2 + r);
    int32_t k = s;
    int32_t cnext = (k + 3 < e) ? colind[k + r] : 0;
    for (; k + 3 < e; k += 4) {
        const int32_t cme = cnext;
        if (k + 7 < e) cnext = colind[k + 4 + r];
        if (GH) gr->arrive(*gs, cme >= gr->n_own);
        const double2* xp = reinterpret_cast<const double2*>(GH ? gr->ptr(x, cme) : x + 4 * (int64_t)cme);
        const double2 xa = xp[0], xb = xp[1];
        if (FMT == 1) {
            const float4 a0 = v32[0], a1 = v32[4], a2 = v32[8], a3 = v32[12];
            acc0 += lp_dot_b(a0, qb2<0>(xa), qb2<0>(xb));
            acc1 += lp_dot_b(a1, qb2<1>(xa), qb2<1>(xb));
            acc0 += lp_dot_b(a2, qb2<2>(xa), qb2<2>(xb));
            acc1 += lp_dot_b(a3, qb2<3>(xa), qb2<3>(xb));
            v32 += 16;
        } else {
            const uint4 p0 = v16[0], p1 = v16[4];
            const f16x8_t h0 = *reinterpret_cast<const f16x8_t*>(&p0), h1 = *reinterpret_cast<const f16x8_t*>(&p1);
            acc0 += lp_dot_b(h0.lo, qb2<0>(xa), qb2<0>(xb));
            acc1 += lp_dot_b(h0.hi, qb2<1>(xa), qb2<1>(xb));
            acc0 += lp_dot_b(h1.lo, qb2<2>(xa), qb2<2>(xb));
            acc1 += lp_dot_b(h1.hi, qb2<3>(xa), qb2<3>(xb));
            v16 += 8;
        }
    }
    if (k < e) {                                                  // 1..3 blocks left; quad-uniform
        const int32_t left = e - k;
        const int32_t cme = colind[k + (r < left ? r : 0)];
        if (GH) gr->arrive(*gs, cme >= gr->n_own);
        const double2* xp = reinterpret_cast<const double2*>(GH ? gr->ptr(x, cme) : x + 4 * (int64_t)cme);
        const double2 xa = xp[0], xb = xp[1];
        if (FMT == 1) {
            acc0 += lp_dot_b(v32[0], qb2<0>(xa), qb2<0>(xb));
            if (left > 1) acc1 += lp_dot_b(v32[4], qb2<1>(xa), qb2<1>(xb));
            if (left > 2) acc0 += lp_dot_b(v32[8], qb2<2>(xa), qb2<2>(xb));
        } else {
            if (left > 1) {
                const uint4 p0 = v16[0];
                const f16x8_t h0 = *reinterpret_cast<const f16x8_t*>(&p0);
                acc0 += lp_dot_b(h0.lo, qb2<0>(xa), qb2<0>(xb));
                acc1 += lp_dot_b(h0.hi, qb2<1>(xa), qb2<1>(xb));
            }
            if (left & 1) {                                       // the odd last block: plain layout, 8 B per row
                const uint2 q = reinterpret_cast<const uint2*>(vals_v)[(int64_t)(e - 1) * 4 + r];
                const f16x4_t hq = *reinterpret_cast<const f16x4_t*>(&q);
                if (left == 1) acc0 += lp_dot_b(hq, qb2<0>(xa), qb2<0>(xb));
                else acc0 += lp_dot_b(hq, qb2<2>(xa), qb2<2>(xb));
            }
        }
    }
    return acc0 + acc1;
}

// Row j of an aggregate's inverse block B_G^-1, in the format of the level's matrix copy: BF 1 = fp32 (4 KiB per aggregate, stored
// [k/4][j] as float4), BF 2 = fp16 with one fp32 scale per row (row-max normalisation like the fp16 matrix copy: 2 KiB + 128 B per
// aggregate, stored [k/8][j] as 8 halfs = 16 B, the 32 scales behind them).  Either way a half-wave reads 512 contiguous bytes per
// load instruction, requested before the row loop.
template <int BF> struct BinvRow;
template <> struct BinvRow<1> {
    float4 v[8];
    __device__ __forceinline__ void load(const void* __restrict__ binv, int64_t G, int j, bool in) {
        const float4* __restrict__ p = reinterpret_cast<const float4*>(binv) + G * 256 + j;
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) v[k4] = in ? p[k4 * 32] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ __forceinline__ double dot(const double* __restrict__ sr) const {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            const double2 r01 = *reinterpret_cast<const double2*>(sr + 4 * k4), r23 = *reinterpret_cast<const double2*>(sr + 4 * k4 + 2);
            a0 += (double)v[k4].x * r01.x + (double)v[k4].z * r23.x;
            a1 += (double)v[k4].y * r01.y + (double)v[k4].w * r23.y;
        }
        return a0 + a1;
    }
};
template <> struct BinvRow<2> {
    uint4 h[4];
    float sc;
    __device__ __forceinline__ void load(const void* __restrict__ binv, int64_t G, int j, bool in) {
        const uint4* __restrict__ p = reinterpret_cast<const uint4*>(binv) + G * 136 + j;          // 128 uint4 of halfs + 8 of scales
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) h[k8] = in ? p[k8 * 32] : make_uint4(0u, 0u, 0u, 0u);
        sc = in ? reinterpret_cast<const float*>(reinterpret_cast<const uint4*>(binv) + G * 136 + 128)[j] : 0.f;
    }
    __device__ __forceinline__ double dot(const double* __restrict__ sr) const {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            const f16x8_t q = *reinterpret_cast<const f16x8_t*>(&h[k8]);
            const double2 r01 = *reinterpret_cast<const double2*>(sr + 8 * k8), r23 = *reinterpret_cast<const double2*>(sr + 8 * k8 + 2);
            const double2 r45 = *reinterpret_cast<const double2*>(sr + 8 * k8 + 4), r67 = *reinterpret_cast<const double2*>(sr + 8 * k8 + 6);
            a0 += (double)(float)q[0] * r01.x + (double)(float)q[2] * r23.x + (double)(float)q[4] * r45.x + (double)(float)q[6] * r67.x;
            a1 += (double)(float)q[1] * r01.y + (double)(float)q[3] * r23.y + (double)(float)q[5] * r45.y + (double)(float)q[7] * r67.y;
        }
        return (double)sc * (a0 + a1);
    }
};
// (B_G^-1 res)[j] for lane j of the half-wave that holds aggregate G's residual (one entry per lane); sr = this half-wave's 32
// doubles of LDS.  Every thread of the workgroup must call it (barrier inside).
template <int BF>
__device__ __forceinline__ double block_apply(const BinvRow<BF>& B, double res, double* __restrict__ sr, int j) {
    sr[j] = res;
    __syncthreads();
    return B.dot(sr);
}

}  // namespace

// one sweep  y = x + w B^-1 (b - A x)  over the aggregates' member rows; n_slots = 8 * number of aggregates
template <int FMT, int GH>
__global__ __launch_bounds__(256) void k_bsweep(int32_t n_slots, const int32_t* __restrict__ blk_rows,
                                                const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                                const void* __restrict__ vals_v, const float* __restrict__ scale,
                                                const void* __restrict__ binv, const double* __restrict__ x,
                                                double* __restrict__ y, const double* __restrict__ bvec, double omega, GhostSrc gs,
                                                PutDst pd) {
    __shared__ double sres[8 * 32];
    __shared__ int s_last;
    const unsigned long long pseq = put_begin(pd);
    GhostReader gr;
    if (GH) gr.begin(gs);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 3, j = lane & 31;
    const int32_t slot = ((int32_t)xcd_remap_b((int)blockIdx.x, (int)gridDim.x) * 4 + (tid >> 6)) * 16 + (lane >> 2);
    const bool in = slot < n_slots;
    const int32_t row = in ? blk_rows[slot] : -1;
    const bool live = row >= 0;
    BinvRow<FMT> Bv;
    Bv.load(binv, slot >> 3, j, in);
    const int32_t s = live ? rowptr[row] : 0, e = live ? rowptr[row + 1] : 0;
    double pre_b = 0.0, pre_x = 0.0, sc = 1.0;
    if (live) {
        if (FMT == 2) sc = (double)scale[4 * (int64_t)row + r];
        pre_b = bvec[4 * (int64_t)row + r];
        pre_x = x[4 * (int64_t)row + r];
    }
    const double acc = sc * lp_row_times_x<FMT, GH>(s, e, colind, vals_v, x, r, &gr, &gs);
    const double z = block_apply(Bv, live ? pre_b - acc : 0.0, sres + 32 * (tid >> 5), j);
    bool stored = false;
    if (live) {
        const double val = pre_x + omega * z;
        y[4 * (int64_t)row + r] = val;
        if (pd.sr_ptr) stored = put_store(pd, pseq, row, r, val);
    }
    if (pd.sr_ptr) put_finish(pd, pseq, stored, &s_last, 0);
}
#define SNS_INST_BSWEEP(F, G)                                                                                                 \
    template __global__ void k_bsweep<F, G>(int32_t, const int32_t*, const int32_t*, const int32_t*, const void*, const float*, \
                                            const void*, const double*, double*, const double*, double, GhostSrc, PutDst);
SNS_INST_BSWEEP(1, 0) SNS_INST_BSWEEP(2, 0) SNS_INST_BSWEEP(1, 1) SNS_INST_BSWEEP(2, 1)
#undef SNS_INST_BSWEEP

// coarse-grid correction + first post-smoothing sweep over M = A P with the aggregate blocks (k_post_lp's algebra):
//     y = (x1 + P xc) + w B^-1 (r1 - M xc)
// GH: xc's ghost aggregates from the coarse level's receive window.  xc_own: where the row's OWN aggregate I is read (the owned
// part of the coarse vector; the same pointer as xc unless the columns of M were renumbered, e.g. into the replicated tail's ids)
template <int FMT, int GH>
__global__ __launch_bounds__(256) void k_bpost(int32_t n_slots, const int32_t* __restrict__ blk_rows,
                                               const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                               const void* __restrict__ vals_v, const float* __restrict__ scale,
                                               const void* __restrict__ binv, const double* __restrict__ xc,
                                               const double* __restrict__ xc_own,
                                               const double* __restrict__ x_pre, const double* __restrict__ res1, double omega,
                                               const int32_t* __restrict__ agg, const uint8_t* __restrict__ free_mask,
                                               double* __restrict__ y, GhostSrc gs, PutDst pd) {
    __shared__ double sres[8 * 32];
    __shared__ int s_last;
    const unsigned long long pseq = put_begin(pd);
    GhostReader gr;
    if (GH) gr.begin(gs);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 3, j = lane & 31;
    const int32_t slot = ((int32_t)xcd_remap_b((int)blockIdx.x, (int)gridDim.x) * 4 + (tid >> 6)) * 16 + (lane >> 2);
    const bool in = slot < n_slots;
    const int32_t row = in ? blk_rows[slot] : -1;
    const bool live = row >= 0;
    BinvRow<FMT> Bv;
    Bv.load(binv, slot >> 3, j, in);
    const int32_t s = live ? rowptr[row] : 0, e = live ? rowptr[row + 1] : 0;
    double pre_b = 0.0, pre_x = 0.0, sc = 1.0;
    if (live) {
        const int32_t I = agg[row];
        if (FMT == 2) sc = (double)scale[4 * (int64_t)row + r];
        pre_b = res1[4 * (int64_t)row + r];
        pre_x = x_pre[4 * (int64_t)row + r];
        if (I >= 0 && (!free_mask || free_mask[4 * (int64_t)row + r])) pre_x += xc_own[4 * (int64_t)I + r];   // (x1 + P xc)_row
    }
    const double acc = sc * lp_row_times_x<FMT, GH>(s, e, colind, vals_v, xc, r, &gr, &gs);
    const double z = block_apply(Bv, live ? pre_b - acc : 0.0, sres + 32 * (tid >> 5), j);
    bool stored = false;
    if (live) {
        const double val = pre_x + omega * z;
        y[4 * (int64_t)row + r] = val;
        if (pd.sr_ptr) stored = put_store(pd, pseq, row, r, val);
    }
    if (pd.sr_ptr) put_finish(pd, pseq, stored, &s_last, 0);
}
#define SNS_INST_BPOST(F, G)                                                                                                  \
    template __global__ void k_bpost<F, G>(int32_t, const int32_t*, const int32_t*, const int32_t*, const void*, const float*, \
                                           const void*, const double*, const double*, const double*, const double*, double,   \
                                           const int32_t*, const uint8_t*, double*, GhostSrc, PutDst);
SNS_INST_BPOST(1, 0) SNS_INST_BPOST(2, 0) SNS_INST_BPOST(1, 1) SNS_INST_BPOST(2, 1)
#undef SNS_INST_BPOST

// z = w B^-1 b: the first sweep of a cycle (zero initial guess); with w = 1 the B^-1 of the spectral estimate
template <int FMT>
__global__ __launch_bounds__(256) void k_bfirst(int32_t n_slots, const int32_t* __restrict__ blk_rows,
                                                const void* __restrict__ binv, const double* __restrict__ bvec, double omega,
                                                double* __restrict__ z, PutDst pd) {
    __shared__ double sres[8 * 32];
    __shared__ int s_last;
    const unsigned long long pseq = put_begin(pd);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 3, j = lane & 31;
    const int32_t slot = ((int32_t)xcd_remap_b((int)blockIdx.x, (int)gridDim.x) * 4 + (tid >> 6)) * 16 + (lane >> 2);
    const bool in = slot < n_slots;
    const int32_t row = in ? blk_rows[slot] : -1;
    const bool live = row >= 0;
    BinvRow<FMT> Bv;
    Bv.load(binv, slot >> 3, j, in);
    const double zz = block_apply(Bv, live ? bvec[4 * (int64_t)row + r] : 0.0, sres + 32 * (tid >> 5), j);
    bool stored = false;
    if (live) {
        z[4 * (int64_t)row + r] = omega * zz;
        if (pd.sr_ptr) stored = put_store(pd, pseq, row, r, omega * zz);
    }
    if (pd.sr_ptr) put_finish(pd, pseq, stored, &s_last, 0);
}
template __global__ void k_bfirst<1>(int32_t, const int32_t*, const void*, const double*, double, double*, PutDst);
template __global__ void k_bfirst<2>(int32_t, const int32_t*, const void*, const double*, double, double*, PutDst);

// First sweep of the REPLICATED tail's first level with the wait half of the all-gather of its right-hand side inside: row g of the
// level is piece rowmap[g] (= rank * maxn + i) of the gathered vector, read from this rank's staging area once every rank's flag
// of the round has arrived; b_out gets the right-hand side for the sweeps that follow.  The last workgroup stores the round.
template <int FMT>
__global__ __launch_bounds__(256) void k_bfirst_gather(int32_t n_slots, const int32_t* __restrict__ blk_rows,
                                                       const void* __restrict__ binv, double omega, double* __restrict__ z,
                                                       double* __restrict__ b_out, const int32_t* __restrict__ rowmap, AgGet ag) {
    __shared__ double sres[8 * 32];
    __shared__ int last;
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 3, j = lane & 31;
    const unsigned long long seq = __hip_atomic_load(ag.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
    if (ag.flags) {
        if (tid < ag.nranks) (void)peer_flag_wait(&ag.ctl->ag_flag[tid], seq, ag.timeout_ticks, ag.err, 3);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    const double* __restrict__ src = ag.stage + (int64_t)(seq & 1ull) * ag.stage_doubles;
    const int32_t slot = ((int32_t)xcd_remap_b((int)blockIdx.x, (int)gridDim.x) * 4 + (tid >> 6)) * 16 + (lane >> 2);
    const bool in = slot < n_slots;
    const int32_t row = in ? blk_rows[slot] : -1;
    const bool live = row >= 0;
    BinvRow<FMT> Bv;
    Bv.load(binv, slot >> 3, j, in);
    double bv = 0.0;
    if (live) {
        bv = src[4 * (int64_t)rowmap[row] + c];
        b_out[4 * (int64_t)row + c] = bv;
    }
    const double zz = block_apply(Bv, bv, sres + 32 * (tid >> 5), j);
    if (live) z[4 * (int64_t)row + c] = omega * zz;
    __syncthreads();
    if (tid == 0) last = (atomicAdd(ag.done, 1u) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (last && tid == 0) {
        *ag.done = 0u;
        __hip_atomic_store(ag.seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template __global__ void k_bfirst_gather<1>(int32_t, const int32_t*, const void*, double, double*, double*, const int32_t*, AgGet);
template __global__ void k_bfirst_gather<2>(int32_t, const int32_t*, const void*, double, double*, double*, const int32_t*, AgGet);

// The FINE level's first sweep (aggregate blocks: partitioned handles of the strong split, amg_block_fine_rows) inside the BiCGStab
// vector kernel that produces the cycle's input -- what k_bicg_s_first / k_bicg_xrp_first are to the nodal blocks: one dependent
// launch and one read of the input less per cycle.  The dofs are walked in the order of the aggregates (every owned node is in
// exactly one), the vector updates are elementwise, so the order does not matter to them.  (z may be the buffer ph or sh was read
// from -- every lane reads its own entry before it writes it; those three carry no __restrict__.)
//   OP 1: s = r - alpha v;                                           z = w B^-1 s
//   OP 2: x += alpha ph + omega sh;  r = s - omega t;  p = r + beta (p - omega v);      z = w B^-1 p
template <int FMT, int OP>
__global__ __launch_bounds__(256) void k_bfirst_bicg(int32_t n_slots, const int32_t* __restrict__ blk_rows,
                                                     const void* __restrict__ binv, double omega_pc, double* z,
                                                     const double* __restrict__ sc, const double* ph, const double* sh,
                                                     const double* __restrict__ t,
                                                     const double* __restrict__ v, double* __restrict__ x, double* __restrict__ r,
                                                     double* __restrict__ p, double* __restrict__ s, PutDst pd) {
    __shared__ double sres[8 * 32];
    __shared__ int s_last;
    const unsigned long long pseq = put_begin(pd);
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 3, j = lane & 31;
    const int32_t slot = ((int32_t)xcd_remap_b((int)blockIdx.x, (int)gridDim.x) * 4 + (tid >> 6)) * 16 + (lane >> 2);
    const bool in = slot < n_slots;
    const int32_t row = in ? blk_rows[slot] : -1;
    const bool live = row >= 0;
    BinvRow<FMT> Bv;
    Bv.load(binv, slot >> 3, j, in);
    double val = 0.0;
    if (live) {
        const int64_t i = 4 * (int64_t)row + c;
        if (OP == 1) {
            val = r[i] - sc[1] * v[i];
            s[i] = val;
        } else {
            const double alpha = sc[1], omega = sc[2], beta = sc[3];
            x[i] += alpha * ph[i] + omega * sh[i];
            const double rn = s[i] - omega * t[i];
            r[i] = rn;
            val = rn + beta * (p[i] - omega * v[i]);
            p[i] = val;
        }
    }
    const double zz = block_apply(Bv, val, sres + 32 * (tid >> 5), j);
    bool stored = false;
    if (live) {
        z[4 * (int64_t)row + c] = omega_pc * zz;
        if (pd.sr_ptr) stored = put_store(pd, pseq, row, c, omega_pc * zz);
    }
    if (pd.sr_ptr) put_finish(pd, pseq, stored, &s_last, 0);
}
#define SNS_INST_BFB(F, O)                                                                                                  \
    template __global__ void k_bfirst_bicg<F, O>(int32_t, const int32_t*, const void*, double, double*, const double*,         \
                                                 const double*, const double*, const double*, const double*, double*, double*, \
                                                 double*, double*, PutDst);
SNS_INST_BFB(1, 1) SNS_INST_BFB(1, 2) SNS_INST_BFB(2, 1) SNS_INST_BFB(2, 2)
#undef SNS_INST_BFB

// Restriction to a level that is smoothed with aggregate blocks, fused with that level's first sweep from the zero guess:
//     bc[I] = sum_{i in I} free_i r[i]   (k_restrict's gather, same member order => same bits),     z = w_c B_c^-1 bc.
// The coarse nodes are walked in the order of THEIR aggregates (blk_rows_c of the coarse level), so that the 32 lanes of a
// half-wave hold one coarse aggregate's right-hand side when it is complete -- one dependent launch less per level and cycle.
template <int FMT>
__global__ __launch_bounds__(256) void k_restrict_blk(int32_t n_slots, const int32_t* __restrict__ blk_rows_c,
                                                      const int32_t* __restrict__ m_ptr, const int32_t* __restrict__ m_idx,
                                                      const uint8_t* __restrict__ free_mask, const double* __restrict__ r,
                                                      double* __restrict__ bc, const void* __restrict__ binv_c, double omega_c,
                                                      double* __restrict__ z_c, PutDst pd) {
    __shared__ double sres[8 * 32];
    __shared__ int s_last;
    const unsigned long long pseq = put_begin(pd);
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 3, j = lane & 31;
    const int32_t slot = ((int32_t)xcd_remap_b((int)blockIdx.x, (int)gridDim.x) * 4 + (tid >> 6)) * 16 + (lane >> 2);
    const bool in = slot < n_slots;
    const int32_t I = in ? blk_rows_c[slot] : -1;
    const bool live = I >= 0;
    BinvRow<FMT> Bv;
    Bv.load(binv_c, slot >> 3, j, in);
    double s = 0.0;
    if (live) {
        const int32_t k0 = m_ptr[I], k1 = m_ptr[I + 1];
        for (int32_t k = k0; k < k1; k += 8) {                     // all member ids of a batch first, then their residual entries
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
        bc[4 * (int64_t)I + c] = s;
    }
    const double zz = block_apply(Bv, live ? s : 0.0, sres + 32 * (tid >> 5), j);
    bool stored = false;
    if (live) {
        z_c[4 * (int64_t)I + c] = omega_c * zz;
        if (pd.sr_ptr) stored = put_store(pd, pseq, I, c, omega_c * zz);       // (the coarse level's plan: its first sweep is put at once)
    }
    if (pd.sr_ptr) put_finish(pd, pseq, stored, &s_last, 0);
}
template __global__ void k_restrict_blk<1>(int32_t, const int32_t*, const int32_t*, const int32_t*, const uint8_t*, const double*,
                                           double*, const void*, double, double*, PutDst);
template __global__ void k_restrict_blk<2>(int32_t, const int32_t*, const int32_t*, const int32_t*, const uint8_t*, const double*,
                                           double*, const void*, double, double*, PutDst);

// Residual + restriction (+ the coarse level's first sweep) of a level >= 1 in ONE launch:
//     r = b - A x,      bc[I] = sum_{i in I} free_i r[i],      z_c = w_c S_c bc   (MODE 0: none, 1: nodal D_c^-1, 2: aggregate blocks B_c^-1)
// in place of k_spmv_lp<B_MINUS_AX> followed by k_restrict / k_restrict_blk -- below the fine level every launch is ~5-8 us of
// latency and the residual is read back by nobody but the restriction (and, later, by the fused post-sweep: r is still written).
// A workgroup takes one smoother block of the COARSE level (8 coarse nodes, slots blk_rows_c[8 G ..]; MODE != 2: 8 consecutive coarse
// nodes), a half-wave one coarse node, a quad one member row of that node -- 64 fine rows per workgroup with the row loop of
// k_spmv_lp / k_bsweep, i.e. the residual pass keeps its parallelism.  The member sums are formed in member order (as k_restrict
// forms them: same bits); nodes of more than 8 members take further rounds.
template <int FMT, int MODE, int GH>
__global__ __launch_bounds__(256) void k_resid_restrict(int32_t nc, int32_t n_cslots, const int32_t* __restrict__ blk_rows_c,
                                                        const int32_t* __restrict__ m_ptr, const int32_t* __restrict__ m_idx,
                                                        const uint8_t* __restrict__ free_mask, const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ colind, const void* __restrict__ vals_v,
                                                        const float* __restrict__ scale, const double* __restrict__ x,
                                                        const double* __restrict__ bvec, double* __restrict__ r_out,
                                                        double* __restrict__ bc, const float* __restrict__ dinv32_c,
                                                        const void* __restrict__ binv_c, double omega_c, double* __restrict__ z_c,
                                                        GhostSrc gs, AgPut agp, PutDst pd) {
    __shared__ double sred[8][8][4];
    __shared__ double sbc[32];
    __shared__ int s_last;
    const unsigned long long pseq = put_begin(pd);
    GhostReader gr;
    if (GH) gr.begin(gs);
    // (agp: the restricted right-hand side also goes into every rank's all-gather staging area -- the level below is the source
    // of a partitioned run's replicated tail -- and the last workgroup raises this rank's flag there)
    bool pstored = false;
    const unsigned long long ag_seq = agp.ag ? __hip_atomic_load(agp.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull : 0ull;
    const int tid = threadIdx.x, hw = tid >> 5, q = (tid & 31) >> 2, c = tid & 3;
    const int32_t G = (int32_t)xcd_remap_b((int)blockIdx.x, (int)gridDim.x);          // the coarse smoother block of this workgroup
    const int32_t slot = G * 8 + hw;
    int32_t I = -1;
    if (slot < n_cslots) I = blk_rows_c ? blk_rows_c[slot] : (slot < nc ? slot : -1);
    BinvRow<FMT> Bv;
    if (MODE == 2) Bv.load(binv_c, (int64_t)G, tid & 31, hw == 0);           // (half-wave 0 applies the coarse block)
    const int32_t k0 = I >= 0 ? m_ptr[I] : 0, k1 = I >= 0 ? m_ptr[I + 1] : 0;
    double s = 0.0;
    for (int32_t kb = k0; __syncthreads_or(kb < k1); kb += 8) {
        const int32_t row = (kb + q < k1) ? m_idx[kb + q] : -1;
        const bool live = row >= 0;
        const int32_t rs = live ? rowptr[row] : 0, re = live ? rowptr[row + 1] : 0;
        double v = 0.0;
        if (live) {
            double sc = 1.0;
            if (FMT == 2) sc = (double)scale[4 * (int64_t)row + c];
            const double pre_b = bvec[4 * (int64_t)row + c];
            const double res = pre_b - sc * lp_row_times_x<FMT, GH>(rs, re, colind, vals_v, x, c, &gr, &gs);
            r_out[4 * (int64_t)row + c] = res;
            v = (!free_mask || free_mask[4 * (int64_t)row + c]) ? res : 0.0;
        }
        sred[hw][q][c] = v;
        __syncthreads();
        if (q == 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t) s += sred[hw][t][c];
        }
        __syncthreads();
    }
    if (q == 0 && I >= 0) {
        bc[4 * (int64_t)I + c] = s;
        if (agp.ag) {
            const int64_t o = (int64_t)(ag_seq & 1ull) * agp.stage_doubles + (int64_t)agp.rank * agp.slot_doubles + 4 * (int64_t)I + c;
            for (int r = 0; r < agp.nranks; ++r) agp.ag[r][o] = s;
        }
    }
    if (agp.ag) {
        __shared__ int ag_last;
        if (agp.flags) __threadfence_system();
        __syncthreads();
        if (tid == 0) ag_last = (atomicAdd(agp.done, 1u) == gridDim.x - 1) ? 1 : 0;
        __syncthreads();
        if (ag_last) {
            if (agp.flags) {
                __threadfence_system();
                if (tid < agp.nranks) peer_flag_store(&agp.ctl[tid]->ag_flag[agp.rank], ag_seq);
            }
            if (tid == 0) *agp.done = 0u;
        }
    }
    if (MODE == 1) {
        if (q == 0 && I >= 0) {                                                      // whole quads: the four sums of the node
            const float4 D = *reinterpret_cast<const float4*>(dinv32_c + 16 * (int64_t)I + 4 * c);
            const double s0 = qb<0>(s), s1 = qb<1>(s), s2 = qb<2>(s), s3 = qb<3>(s);
            z_c[4 * (int64_t)I + c] = omega_c * ((double)D.x * s0 + (double)D.y * s1 + (double)D.z * s2 + (double)D.w * s3);
        }
    } else if (MODE == 2) {
        if (q == 0) sbc[4 * hw + c] = I >= 0 ? s : 0.0;
        __syncthreads();
        if (hw == 0) {
            const int j = tid & 31;
            const double zz = Bv.dot(sbc);
            const int32_t J = (G * 8 + (j >> 2) < n_cslots) ? blk_rows_c[(int64_t)G * 8 + (j >> 2)] : -1;
            if (J >= 0) {
                z_c[4 * (int64_t)J + (j & 3)] = omega_c * zz;
                if (pd.sr_ptr) pstored = put_store(pd, pseq, J, j & 3, omega_c * zz);
            }
        }
    }
    if (pd.sr_ptr) put_finish(pd, pseq, pstored, &s_last, 1);
}
#define SNS_INST_RR(F, M, G)                                                                                                        \
    template __global__ void k_resid_restrict<F, M, G>(int32_t, int32_t, const int32_t*, const int32_t*, const int32_t*, const uint8_t*, \
                                                       const int32_t*, const int32_t*, const void*, const float*, const double*,    \
                                                       const double*, double*, double*, const float*, const void*, double, double*, \
                                                       GhostSrc, AgPut, PutDst);
SNS_INST_RR(1, 0, 0) SNS_INST_RR(1, 1, 0) SNS_INST_RR(1, 2, 0) SNS_INST_RR(2, 0, 0) SNS_INST_RR(2, 1, 0) SNS_INST_RR(2, 2, 0)
SNS_INST_RR(1, 0, 1) SNS_INST_RR(1, 1, 1) SNS_INST_RR(1, 2, 1) SNS_INST_RR(2, 0, 1) SNS_INST_RR(2, 1, 1) SNS_INST_RR(2, 2, 1)
#undef SNS_INST_RR

// B_G^-1 of every smoother block from the level's fp64 operator: 8 blocks per workgroup, 32 lanes each (lane j = column j of the
// 32 x 32 block in LDS).  Member slots a block does not fill keep identity rows / columns.  (Blocks = the aggregates; the rare
// aggregate of more than 8 nodes -- a leftover node joins a full neighbour -- is split into chunks of 8 in member order: the
// smoother's partition need not be the coarsening's.  blk_of[node] = the block of a node, -1 for ghost nodes.)
template <int FMT>
__global__ __launch_bounds__(256) void k_binv(int32_t nblk, const int32_t* __restrict__ blk_rows, const int32_t* __restrict__ blk_of,
                                              const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                              const double* __restrict__ vals, void* __restrict__ binv_v,
                                              int* __restrict__ singular) {
    __shared__ double lds[8 * 32 * 33];
    const int tid = threadIdx.x, ha = tid >> 5, j = tid & 31;
    const int32_t G = (int32_t)blockIdx.x * 8 + ha;
    double* __restrict__ M = lds + ha * 32 * 33;
    for (int i = 0; i < 32; ++i) M[i * 33 + j] = (i == j) ? 1.0 : 0.0;
    __syncthreads();
    if (G < nblk) {
        const int q = j >> 2, sub = j & 3;
        const int32_t row = blk_rows[8 * (int64_t)G + q];
        if (row >= 0) {
            for (int32_t k = rowptr[row] + sub; k < rowptr[row + 1]; k += 4) {
                const int32_t jn = colind[k];
                if (blk_of[jn] != G) continue;
                int lj = -1;
                for (int t = 0; t < 8; ++t)
                    if (blk_rows[8 * (int64_t)G + t] == jn) lj = t;
                if (lj < 0) continue;
                const double* __restrict__ blkv = vals + (int64_t)k * 16;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) M[(4 * q + rr) * 33 + 4 * lj + cc] = blkv[4 * rr + cc];
            }
        }
    }
    __syncthreads();
    // in-place Gauss-Jordan, lane j owns column j; the lanes of a wave run in lockstep, so every row's old M[i][p] is read by all
    // of them before lane p overwrites it
    for (int p = 0; p < 32; ++p) {
        const double piv = M[p * 33 + p];
        if (j == 0 && !(fabs(piv) > 1e-300 && fabs(piv) < 1e300)) *singular = 1;
        const double d = 1.0 / piv;
        const double prc = (j == p) ? d : M[p * 33 + j] * d;
        for (int i = 0; i < 32; ++i) {
            if (i == p) continue;
            const double f = M[i * 33 + p];
            const double cur = M[i * 33 + j];
            M[i * 33 + j] = (j == p) ? -f * d : cur - f * prc;
        }
        M[p * 33 + j] = prc;
        __syncthreads();
    }
    if (G < nblk) {
        if (FMT == 1) {
            float4* __restrict__ binv = reinterpret_cast<float4*>(binv_v);
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4)
                binv[((int64_t)G * 8 + k4) * 32 + j] = make_float4((float)M[j * 33 + 4 * k4], (float)M[j * 33 + 4 * k4 + 1],
                                                                   (float)M[j * 33 + 4 * k4 + 2], (float)M[j * 33 + 4 * k4 + 3]);
        } else {
            // fp16 with the row's largest |entry| as scale (nothing overflows or underflows the half range)
            double mx = 0.0;
            for (int k = 0; k < 32; ++k) mx = fmax(mx, fabs(M[j * 33 + k]));
            const double inv = mx > 0.0 ? 1.0 / mx : 0.0;
            uint4* __restrict__ base = reinterpret_cast<uint4*>(binv_v) + (int64_t)G * 136;
#pragma unroll
            for (int k8 = 0; k8 < 4; ++k8) {
                f16x8_t q;
#pragma unroll
                for (int e = 0; e < 8; ++e) q[e] = (_Float16)(float)(M[j * 33 + 8 * k8 + e] * inv);
                base[k8 * 32 + j] = *reinterpret_cast<const uint4*>(&q);
            }
            reinterpret_cast<float*>(base + 128)[j] = (float)mx;
        }
    }
}
template __global__ void k_binv<1>(int32_t, const int32_t*, const int32_t*, const int32_t*, const int32_t*, const double*, void*, int*);
template __global__ void k_binv<2>(int32_t, const int32_t*, const int32_t*, const int32_t*, const int32_t*, const double*, void*, int*);

}  // namespace sns
