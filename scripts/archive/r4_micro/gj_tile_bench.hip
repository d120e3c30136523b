// Round 4 micro-benchmark (experiment, not product): where does the time of the 64 x 64 diagonal-tile inversion of
// csrc/sns_dense.hip go?  Ablations of the same kernel, one workgroup each, timed with HIP events over many launches.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gj_tile_bench scripts/r4_micro/gj_tile_bench.hip && /tmp/gj_tile_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
constexpr int GB = 64, INV_LD = 65;

template <int VAR>   // 0 full; 1 no reciprocal chain (d = 0.5); 2 no LDS exchange / barrier (garbage arithmetic); 3 barrier only
__global__ __launch_bounds__(256) void k_inv(const double* __restrict__ A, double* __restrict__ out, int reps) {
    __shared__ double M[GB * INV_LD + 256];
    double* X = M + GB * INV_LD;
    const int t = threadIdx.x, c = t & 63, w = t >> 6;
    for (int rep = 0; rep < reps; ++rep) {
        double a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = A[(4 * q + w) * 64 + c];
#pragma unroll 1
        for (int k = 0; k < GB; ++k) {
            double* __restrict__ prow = X + (k & 1) * 128;
            double* __restrict__ pcol = prow + 64;
            const int kq = k >> 2;
            double piv, pr, f[16];
            if (VAR != 2) {
                if (w == (k & 3)) {
                    double sel = a[0];
#pragma unroll
                    for (int q = 1; q < 16; ++q) sel = (q == kq) ? a[q] : sel;
                    prow[c] = sel;
                }
                if (VAR != 3 && c == k) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) pcol[4 * q + w] = a[q];
                }
                __syncthreads();
                piv = prow[k];
                pr = prow[c];
#pragma unroll
                for (int q = 0; q < 16; ++q) f[q] = (VAR == 3) ? 0.001 * q : pcol[4 * q + w];
            } else {
                piv = a[kq & 15] + 2.0;
                pr = a[(kq + 1) & 15];
#pragma unroll
                for (int q = 0; q < 16; ++q) f[q] = a[(q + 1) & 15] * 1e-3;
            }
            double d;
            if (VAR == 1) d = 0.5;
            else {
                d = __builtin_amdgcn_rcp(piv);
                d = d * (2.0 - piv * d);
                d = d * (2.0 - piv * d);
            }
            const double pk = (c == k) ? d : pr * d;
            const bool mine = (w == (k & 3));
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const double upd = (c == k) ? -f[q] * d : a[q] - f[q] * pk;
                a[q] = (mine && q == kq) ? pk : upd;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) out[(4 * q + w) * 64 + c] = a[q];
    }
}

// one WAVE, 16 x 16 tile in registers (4 entries per lane, MFMA C/D layout: column l % 16, rows l / 16 + 4 q), pivot row / column by
// ds_bpermute (no barrier): the building block of a 64 x 64 inverse by 16 x 16 blocks
__global__ __launch_bounds__(64) void k_inv16(const double* __restrict__ A, double* __restrict__ out, int reps) {
    const int l = threadIdx.x, c = l & 15, g = l >> 4;
    for (int rep = 0; rep < reps; ++rep) {
        double a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = A[(g + 4 * q) * 64 + c] + (g + 4 * q == c ? 4.0 : 0.0);
#pragma unroll 1
        for (int k = 0; k < 16; ++k) {
            const int kq = k >> 2, kg = k & 3;
            double sel = a[0];
#pragma unroll
            for (int q = 1; q < 4; ++q) sel = (q == kq) ? a[q] : sel;
            const double prow = __shfl(sel, 16 * kg + c);          // M[k][c]
            const double piv = __shfl(sel, 16 * kg + k);           // M[k][k]
            double d = __builtin_amdgcn_rcp(piv);
            d = d * (2.0 - piv * d);
            d = d * (2.0 - piv * d);
            const double pk = (c == k) ? d : prow * d;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double f = __shfl(a[q], 16 * g + k);         // M[g + 4 q][k]
                const double upd = (c == k) ? -f * d : a[q] - f * pk;
                a[q] = (g == kg && q == kq) ? pk : upd;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) out[(g + 4 * q) * 64 + c] = a[q];
    }
}

// ---- hierarchical variant: the 64 x 64 tile as 4 x 4 blocks of 16 x 16 -- block Gauss-Jordan whose pivot blocks are inverted by ONE
// wave in registers (ds_bpermute exchange, no barrier: k_inv16 above) and whose block products run on v_mfma_f64_16x16x4_f64
typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int HL = 65;      // row stride of the tile in LDS
constexpr int SL = 17;      // row stride of a 16 x 16 scratch block

__device__ __forceinline__ void inv16_wave(double (&a)[4], int l) {
    const int c = l & 15, g = l >> 4;
#pragma unroll 1
    for (int k = 0; k < 16; ++k) {
        const int kq = k >> 2, kg = k & 3;
        double sel = a[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) sel = (q == kq) ? a[q] : sel;
        const double prow = __shfl(sel, 16 * kg + c);
        const double piv = __shfl(sel, 16 * kg + k);
        double d = __builtin_amdgcn_rcp(piv);
        d = d * (2.0 - piv * d);
        d = d * (2.0 - piv * d);
        const double pk = (c == k) ? d : prow * d;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double f = __shfl(a[q], 16 * g + k);
            const double upd = (c == k) ? -f * d : a[q] - f * pk;
            a[q] = (g == kg && q == kq) ? pk : upd;
        }
    }
}
// acc (C layout: row l / 16 + 4 reg, column l % 16) += sgn * A B, A and B 16 x 16 blocks in LDS with row strides lda / ldb
__device__ __forceinline__ d4_t mma16(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb, d4_t acc, double sgn, int l) {
    const int c = l & 15, g = l >> 4;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const double a = sgn * A[c * lda + g + 4 * s4];
        const double b = B[(g + 4 * s4) * ldb + c];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ d4_t ldC(const double* __restrict__ X, int ld, int l) {
    const int c = l & 15, g = l >> 4;
    return d4_t{X[g * ld + c], X[(g + 4) * ld + c], X[(g + 8) * ld + c], X[(g + 12) * ld + c]};
}
__device__ __forceinline__ void stC(double* __restrict__ X, int ld, d4_t v, int l) {
    const int c = l & 15, g = l >> 4;
    X[g * ld + c] = v[0]; X[(g + 4) * ld + c] = v[1]; X[(g + 8) * ld + c] = v[2]; X[(g + 12) * ld + c] = v[3];
}
// M: 64 x 64 tile in LDS (stride HL); S: scratch of 7 blocks of 16 x SL doubles (Dinv, R[3], Cn[3])
__device__ __forceinline__ void invert64_h(double* __restrict__ M, double* __restrict__ S) {
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    double* Dinv = S;
    for (int kb = 0; kb < 4; ++kb) {
        if (w == 0) {                                            // pivot block, one wave, registers
            d4_t v = ldC(M + (16 * kb) * HL + 16 * kb, HL, l);
            double a[4] = {v[0], v[1], v[2], v[3]};
            inv16_wave(a, l);
            stC(Dinv, SL, d4_t{a[0], a[1], a[2], a[3]}, l);
        }
        __syncthreads();
        // R_j = Dinv M[kb][j] (3 blocks), Cn_i = -M[i][kb] Dinv (3 blocks): 6 products, wave w takes products w and w + 4
        for (int pidx = w; pidx < 6; pidx += 4) {
            const int o = pidx % 3;
            const int idx = o + (o >= kb ? 1 : 0);               // the o-th block index != kb
            d4_t acc = d4_t{0.0, 0.0, 0.0, 0.0};
            if (pidx < 3) {
                acc = mma16(Dinv, SL, M + (16 * kb) * HL + 16 * idx, HL, acc, 1.0, l);
                stC(S + (1 + o) * 16 * SL, SL, acc, l);
            } else {
                acc = mma16(M + (16 * idx) * HL + 16 * kb, HL, Dinv, SL, acc, -1.0, l);
                stC(S + (4 + o) * 16 * SL, SL, acc, l);
            }
        }
        __syncthreads();
        // M[i][j] -= M[i][kb] R_j for i, j != kb: 9 products
        for (int pidx = w; pidx < 9; pidx += 4) {
            const int oi = pidx / 3, oj = pidx % 3;
            const int i = oi + (oi >= kb ? 1 : 0), j = oj + (oj >= kb ? 1 : 0);
            double* Mij = M + (16 * i) * HL + 16 * j;
            d4_t acc = ldC(Mij, HL, l);
            acc = mma16(M + (16 * i) * HL + 16 * kb, HL, S + (1 + oj) * 16 * SL, SL, acc, -1.0, l);
            stC(Mij, HL, acc, l);
        }
        __syncthreads();
        // row kb <- R_j, column kb <- Cn_i, pivot block <- Dinv: 7 blocks of 256 entries
        for (int e = t; e < 7 * 256; e += 256) {
            const int blk = e >> 8, r = (e >> 4) & 15, c = e & 15;
            const double v = S[blk * 16 * SL + r * SL + c];
            if (blk == 0) M[(16 * kb + r) * HL + 16 * kb + c] = v;
            else if (blk < 4) { const int o = blk - 1, j = o + (o >= kb ? 1 : 0); M[(16 * kb + r) * HL + 16 * j + c] = v; }
            else { const int o = blk - 4, i = o + (o >= kb ? 1 : 0); M[(16 * i + r) * HL + 16 * kb + c] = v; }
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_inv_h(const double* __restrict__ A, double* __restrict__ out, int reps) {
    __shared__ double M[64 * HL];
    __shared__ double S[7 * 16 * SL];
    const int t = threadIdx.x;
    for (int rep = 0; rep < reps; ++rep) {
        for (int e = t; e < 4096; e += 256) M[(e >> 6) * HL + (e & 63)] = A[e];
        __syncthreads();
        invert64_h(M, S);
        for (int e = t; e < 4096; e += 256) out[e] = M[(e >> 6) * HL + (e & 63)];
        __syncthreads();
    }
}

int main() {
    std::vector<double> h(64 * 64);
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 64; ++j) h[i * 64 + j] = (i == j ? 8.0 : 0.0) + 0.3 * std::sin(1.0 + i * 0.37 + j * 0.91) + (i < j ? 0.5 : -0.5) * 0.4;
    double *A, *O;
    hipMalloc(&A, 64 * 64 * 8); hipMalloc(&O, 64 * 64 * 8);
    hipMemcpy(A, h.data(), 64 * 64 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        launch(1);
        hipDeviceSynchronize();
        float ms1 = 0, msN = 0;
        hipEventRecord(e0); for (int i = 0; i < 50; ++i) launch(1); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms1, e0, e1);
        hipEventRecord(e0); launch(200); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&msN, e0, e1);
        std::printf("%-58s %7.2f us per launch (50 launches), %7.2f us per tile inside one launch of 200\n", name, ms1 * 1e3 / 50, msN * 1e3 / 200);
    };
    run("64x64, 4 waves, registers + LDS pivot exchange (product)", [&](int r) { hipLaunchKernelGGL(k_inv<0>, dim3(1), dim3(256), 0, 0, A, O, r); });
    run("  ... without the reciprocal chain", [&](int r) { hipLaunchKernelGGL(k_inv<1>, dim3(1), dim3(256), 0, 0, A, O, r); });
    run("  ... without LDS exchange and barrier", [&](int r) { hipLaunchKernelGGL(k_inv<2>, dim3(1), dim3(256), 0, 0, A, O, r); });
    run("  ... barrier + pivot row only (no pivot column)", [&](int r) { hipLaunchKernelGGL(k_inv<3>, dim3(1), dim3(256), 0, 0, A, O, r); });
    run("16x16, one wave, ds_bpermute, no barrier", [&](int r) { hipLaunchKernelGGL(k_inv16, dim3(1), dim3(64), 0, 0, A, O, r); });
    run("64x64 as 4x4 blocks of 16: wave-level pivot blocks + MFMA", [&](int r) { hipLaunchKernelGGL(k_inv_h, dim3(1), dim3(256), 0, 0, A, O, r); });
    {
        hipLaunchKernelGGL(k_inv_h, dim3(1), dim3(256), 0, 0, A, O, 1);
        std::vector<double> x(64 * 64);
        hipMemcpy(x.data(), O, 64 * 64 * 8, hipMemcpyDeviceToHost);
        double err = 0;
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) {
                double s2 = 0;
                for (int k = 0; k < 64; ++k) s2 += x[i * 64 + k] * h[k * 64 + j];
                err = std::fmax(err, std::fabs(s2 - (i == j)));
            }
        std::printf("|X A - I| of the hierarchical variant: %.2e\n", err);
    }
    // check k_inv<0> against the identity
    hipLaunchKernelGGL(k_inv<0>, dim3(1), dim3(256), 0, 0, A, O, 1);
    std::vector<double> x(64 * 64);
    hipMemcpy(x.data(), O, 64 * 64 * 8, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 64; ++j) {
            double s = 0;
            for (int k = 0; k < 64; ++k) s += x[i * 64 + k] * h[k * 64 + j];
            err = std::fmax(err, std::fabs(s - (i == j)));
        }
    std::printf("|X A - I| of the product variant: %.2e\n", err);
    return 0;
}
