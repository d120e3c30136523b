#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
#include "sns_internal.h"
namespace sns { void set_error(const std::string&) {} }
using namespace sns;
int main() {
    // Kuhn box mesh nx x ny x nz
    const int nx = 14, ny = 9, nz = 7;
    const int sy = nz + 1, sx = (ny + 1) * (nz + 1), n = (nx + 1) * sx;
    const int perm[6][3] = {{0,1,2},{0,2,1},{1,0,2},{1,2,0},{2,0,1},{2,1,0}};
    std::vector<int32_t> tets;
    for (int i = 0; i < nx; ++i) for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k) {
        const int base = i * sx + j * sy + k;
        for (auto& p : perm) {
            int off[3] = {0, 0, 0}; int v = base; tets.push_back(v);
            for (int s = 0; s < 3; ++s) { off[p[s]] = 1; tets.push_back(base + off[0] * sx + off[1] * sy + off[2]); }
        }
    }
    // shuffle local vertex order of some tets
    std::mt19937 rng(3);
    for (size_t t = 0; t < tets.size() / 4; t += 3) std::shuffle(tets.begin() + 4 * t, tets.begin() + 4 * t + 4, rng);
    const int64_t E = (int64_t)tets.size() / 4;
    HostPattern P; HostAssemblyMaps M;
    build_pattern(n, E, tets.data(), P, M);
    std::printf("n %d E %ld nnzb %ld c_idx %zu\n", n, (long)E, (long)P.nnzb, M.c_idx.size());
    HostPattern cur = P; int32_t nown = n;
    for (int l = 0; l < 6 && nown > 8; ++l) {
        std::vector<int32_t> agg; int32_t nc = 0;
        aggregate_nodes(cur, nown, 8, agg, nc);
        HostAggregation A; build_coarse_from_agg(cur, nown, agg, nc, nc, A);
        std::printf("level %d: %d -> %d nodes, coarse nnzb %ld, r_idx %zu\n", l, nown, nc, (long)A.coarse.nnzb, A.r_idx.size());
        // M = A P pattern of the fused post-smoothing sweep: every fine slot lands in exactly one M slot of its row,
        // the M slot's column is the aggregate of the fine slot's column, columns sorted and unique per row
        HostAP M2; build_ap_pattern(cur, nown, A.agg, M2);
        if (M2.ap_ptr.back() != cur.rowptr[nown] || (int64_t)M2.colind.size() != M2.nnz) { std::printf("ERROR ap sizes\n"); return 1; }
        std::vector<char> seen((size_t)cur.rowptr[nown], 0);
        for (int32_t i = 0; i < nown; ++i)
            for (int32_t s = M2.rowptr[i]; s < M2.rowptr[i + 1]; ++s) {
                if (s > M2.rowptr[i] && M2.colind[s] <= M2.colind[s - 1]) { std::printf("ERROR ap order\n"); return 1; }
                if (M2.slot_row[s] != i) { std::printf("ERROR ap slot_row\n"); return 1; }
                for (int32_t q = M2.ap_ptr[s]; q < M2.ap_ptr[s + 1]; ++q) {
                    const int32_t k = M2.ap_idx[q];
                    if (k < cur.rowptr[i] || k >= cur.rowptr[i + 1] || A.agg[cur.colind[k]] != M2.colind[s] || seen[k]++) {
                        std::printf("ERROR ap gather list\n"); return 1;
                    }
                }
            }
        // block -> slot nibbles of k_lp_copies16: the inverse of the gather lists wherever the row fits the kernel's registers
        if ((int32_t)M2.nib.size() != nown) { std::printf("ERROR nib size\n"); return 1; }
        int64_t in_regs = 0;
        for (int32_t i = 0; i < nown; ++i) {
            const int32_t cnt = cur.rowptr[i + 1] - cur.rowptr[i], cm = M2.rowptr[i + 1] - M2.rowptr[i];
            if (M2.nib[i] == ~0ull) { if (cnt <= 16 && cm <= 8) { std::printf("ERROR nib fallback\n"); return 1; } continue; }
            if (cnt > 16 || cm > 8) { std::printf("ERROR nib range\n"); return 1; }
            ++in_regs;
            for (int32_t j = 0; j < 16; ++j) {
                const int t = (int)((M2.nib[i] >> (4 * j)) & 15);
                if (j >= cnt) { if (t != 15) { std::printf("ERROR nib padding\n"); return 1; } continue; }
                const int32_t J = A.agg[cur.colind[cur.rowptr[i] + j]];
                if (J < 0 ? t != 15 : (t >= cm || M2.colind[M2.rowptr[i] + t] != J)) { std::printf("ERROR nib slot\n"); return 1; }
            }
        }
        std::printf("level %d: %ld of %d rows fit the one-pass copy kernel's registers\n", l, (long)in_regs, nown);
        std::printf("level %d: A*P pattern %ld slots (%.2f of the level's blocks)\n", l, (long)M2.nnz, (double)M2.nnz / cur.rowptr[nown]);
        cur = A.coarse; nown = nc;
    }
    // partial-active aggregation (distributed level 0)
    std::vector<int32_t> agg; int32_t nc = 0;
    aggregate_nodes(P, n / 2, 8, agg, nc);
    HostAggregation A2; build_aggregation_active(P, n / 2, 8, A2);
    std::printf("active aggregation: %d aggregates\n", A2.nc);
    return 0;
}
