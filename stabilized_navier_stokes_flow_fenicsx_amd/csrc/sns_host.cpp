// Host-side symbolic setup: BSR sparsity pattern from the tet connectivity,
// gather maps for the atomic-free assembly, and the aggregation hierarchy.
//
// Replaces what DOLFINx does behind `create_matrix(problem.a)`
// (NavierStokesChannelFlow.py:272: sparsity pattern from the dofmap) and what
// PETSc's MatSetValuesLocal row search does on every assembly (:74): here the
// (tet, a, b) -> block-slot relation is resolved ONCE and inverted, so the
// device never searches and never needs atomics.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <complex>
#include <cstring>
#include <numeric>
#include <stdexcept>

#include "sns_internal.h"
#include "sns_policy.h"

namespace sns {

// npe = vertices per cell (4: tets, 3: triangles); the connectivity is stored in a stride of 4 either way and
// element-block ids stay cell*16 + a*4 + b.
void build_pattern(int32_t n, int64_t E, const int32_t* tets, HostPattern& P, HostAssemblyMaps& M, int npe) {
    // element-block ids tet*16 + a*4 + b are stored as UNSIGNED 32-bit values: up to 2^28 = 268 M tets
    if (E * 16 > (int64_t)UINT32_MAX) throw std::runtime_error("mesh too large for 32-bit element-block ids (268 M tets)");
    // ---- node -> incident (tet, a) ------------------------------------------
    M.nt_ptr.assign((size_t)n + 1, 0);
    for (int64_t t = 0; t < E; ++t)
        for (int a = 0; a < npe; ++a) {
            int32_t v = tets[4 * t + a];
            if (v < 0 || v >= n) throw std::runtime_error("tet vertex id out of range");
            M.nt_ptr[(size_t)v + 1]++;
        }
    for (int32_t i = 0; i < n; ++i) M.nt_ptr[i + 1] += M.nt_ptr[i];
    M.nt_idx.resize((size_t)npe * E);
    {
        std::vector<int64_t> cur(M.nt_ptr.begin(), M.nt_ptr.end() - 1);
        for (int64_t t = 0; t < E; ++t)                       // tet order => deterministic gather order
            for (int a = 0; a < npe; ++a) M.nt_idx[(size_t)cur[tets[4 * t + a]]++] = (int32_t)(4 * t + a);
    }
    // ---- rows: sorted unique neighbour nodes ----------------------------------
    P.n = n;
    P.rowptr.assign((size_t)n + 1, 0);
    std::vector<int32_t> cnt((size_t)n, 0);
#pragma omp parallel
    {
        std::vector<int32_t> tmp;
#pragma omp for schedule(dynamic, 1024)
        for (int32_t i = 0; i < n; ++i) {
            tmp.clear();
            for (int64_t k = M.nt_ptr[i]; k < M.nt_ptr[i + 1]; ++k) {
                const int32_t* tv = tets + 4 * (int64_t)(M.nt_idx[k] >> 2);
                tmp.insert(tmp.end(), tv, tv + npe);
            }
            if (tmp.empty()) tmp.push_back(i);                // isolated node: keep a diagonal
            std::sort(tmp.begin(), tmp.end());
            cnt[i] = (int32_t)(std::unique(tmp.begin(), tmp.end()) - tmp.begin());
        }
    }
    int64_t nnzb = 0;
    for (int32_t i = 0; i < n; ++i) nnzb += cnt[i];
    if (nnzb > (int64_t)INT32_MAX) throw std::runtime_error("nnz blocks exceed int32 (shard the mesh)");
    for (int32_t i = 0; i < n; ++i) P.rowptr[i + 1] = P.rowptr[i] + cnt[i];
    P.nnzb = nnzb;
    P.colind.resize((size_t)nnzb);
    P.diag.resize((size_t)n);
#pragma omp parallel
    {
        std::vector<int32_t> tmp;
#pragma omp for schedule(dynamic, 1024)
        for (int32_t i = 0; i < n; ++i) {
            tmp.clear();
            for (int64_t k = M.nt_ptr[i]; k < M.nt_ptr[i + 1]; ++k) {
                const int32_t* tv = tets + 4 * (int64_t)(M.nt_idx[k] >> 2);
                tmp.insert(tmp.end(), tv, tv + npe);
            }
            if (tmp.empty()) tmp.push_back(i);
            std::sort(tmp.begin(), tmp.end());
            auto e = std::unique(tmp.begin(), tmp.end());
            std::copy(tmp.begin(), e, P.colind.begin() + P.rowptr[i]);
            P.diag[i] = P.rowptr[i] + (int32_t)(std::lower_bound(tmp.begin(), e, i) - tmp.begin());
        }
    }
    // ---- slot -> contributing element blocks -----------------------------------
    // Row i receives block (a,b) of every incident (tet,a); count per slot, then fill in
    // (incident-tet order, b order): rows are independent, so this is parallel AND deterministic.
    M.c_ptr.assign((size_t)nnzb + 1, 0);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int32_t i = 0; i < n; ++i) {
        const int32_t* cb = P.colind.data() + P.rowptr[i];
        const int32_t len = P.rowptr[i + 1] - P.rowptr[i];
        for (int64_t k = M.nt_ptr[i]; k < M.nt_ptr[i + 1]; ++k) {
            const int32_t* tv = tets + 4 * (int64_t)(M.nt_idx[k] >> 2);
            for (int b = 0; b < npe; ++b) {
                int32_t s = P.rowptr[i] + (int32_t)(std::lower_bound(cb, cb + len, tv[b]) - cb);
                M.c_ptr[(size_t)s + 1]++;
            }
        }
    }
    for (int64_t s = 0; s < nnzb; ++s) M.c_ptr[s + 1] += M.c_ptr[s];
    M.c_idx.resize((size_t)npe * npe * E);
#pragma omp parallel
    {
        std::vector<int32_t> fill;
#pragma omp for schedule(dynamic, 1024)
        for (int32_t i = 0; i < n; ++i) {
            const int32_t* cb = P.colind.data() + P.rowptr[i];
            const int32_t len = P.rowptr[i + 1] - P.rowptr[i];
            fill.assign((size_t)len, 0);
            for (int64_t k = M.nt_ptr[i]; k < M.nt_ptr[i + 1]; ++k) {
                const int32_t ta = M.nt_idx[k];
                const int32_t* tv = tets + 4 * (int64_t)(ta >> 2);
                for (int b = 0; b < npe; ++b) {
                    int32_t j = (int32_t)(std::lower_bound(cb, cb + len, tv[b]) - cb);
                    int64_t s = P.rowptr[i] + j;
                    M.c_idx[(size_t)(M.c_ptr[s] + fill[j]++)] =
                        (int32_t)((uint32_t)(ta >> 2) * 16u + (uint32_t)((ta & 3) * 4 + b));
                }
            }
        }
    }
}

// Greedy size-limited aggregation on the node graph (plain aggregation AMG).
// A seed takes up to max_agg-1 still-free neighbours; a free node with no free
// neighbour joins the aggregate of its first aggregated neighbour.  Sequential
// and therefore deterministic; O(nnzb).  Rows >= fine.n_owned ... are handled by
// the caller through `owned` (nodes outside are never aggregated).
// Is the node cloud of a level anisotropic -- cells stretched in one direction, like the extruded nozzle channel whose planes lie
// 2 lc apart over a cross-section of size lc / 2?  Measure: the median over the nodes of (longest / shortest edge)^2.  A Kuhn
// box gives 3 (edge, face and body diagonals), body-centred and Delaunay meshes less, an extrusion with aspect ratio a about
// a^2 + 1.  Above ANISO_ON the aggregation below keeps to the STRONG connections (round 5).
constexpr double ANISO_ON = 6.0, ANISO_KEEP = 4.0;
bool cloud_is_anisotropic(const HostPattern& F, int32_t n_active, const double* pts) {
    if (!pts || n_active < 64) return false;
    std::vector<double> ratio;
    const int32_t step = std::max(1, n_active / 20000);                    // a sample of the nodes is enough
    for (int32_t i = 0; i < n_active; i += step) {
        double lo = 1e300, hi = 0.0;
        for (int32_t k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k) {
            const int32_t j = F.colind[k];
            if (j == i) continue;
            double d2 = 0.0;
            for (int c = 0; c < 3; ++c) { const double d = pts[3 * (size_t)j + c] - pts[3 * (size_t)i + c]; d2 += d * d; }
            lo = std::min(lo, d2);
            hi = std::max(hi, d2);
        }
        if (hi > 0.0 && lo > 0.0) ratio.push_back(hi / lo);
    }
    if (ratio.size() < 16) return false;
    std::nth_element(ratio.begin(), ratio.begin() + ratio.size() / 2, ratio.end());
    return ratio[ratio.size() / 2] > ANISO_ON;
}

namespace {

// how compact a set of aggregates is: total scatter of the nodes about their aggregates' centroids, made independent of the
// NUMBER of aggregates (n clusters of a cloud of fixed volume scatter like n^(-2/3) each): lower = more compact
double scatter_score(const std::vector<int32_t>& agg, int32_t nc, int32_t n_active, const double* pts) {
    if (nc <= 0) return 1e300;
    std::vector<double> c((size_t)3 * nc, 0.0);
    std::vector<int32_t> cnt((size_t)nc, 0);
    for (int32_t i = 0; i < n_active; ++i) {
        const int32_t a = agg[i];
        if (a < 0 || a >= nc) continue;
        for (int k = 0; k < 3; ++k) c[3 * (size_t)a + k] += pts[3 * (size_t)i + k];
        ++cnt[a];
    }
    for (int32_t a = 0; a < nc; ++a)
        if (cnt[a]) for (int k = 0; k < 3; ++k) c[3 * (size_t)a + k] /= cnt[a];
    double w = 0.0;
    for (int32_t i = 0; i < n_active; ++i) {
        const int32_t a = agg[i];
        if (a < 0 || a >= nc) continue;
        for (int k = 0; k < 3; ++k) { const double d = pts[3 * (size_t)i + k] - c[3 * (size_t)a + k]; w += d * d; }
    }
    return w * std::pow((double)nc, 2.0 / 3.0);
}

// Pairwise aggregation (round 5): log2(max_agg) rounds in which every cluster is matched with the CLOSEST adjacent cluster that is
// still unmatched (centroid distance; on an anisotropic cloud only within ANISO_KEEP x its closest neighbour cluster), nodes ->
// pairs -> quadruples -> octets.  Unlike the greedy sweep its aggregates do not depend on how the node numbers run through the
// mesh: the greedy sweep takes the free neighbours AHEAD of its front, which on a Kuhn lattice numbered along the cells' common
// diagonal are exactly the seven other corners of a cube -- and on the same lattice numbered against it a sheared box three nodes
// wide, for 41 % more Krylov iterations (profiles/r5_prism_vs_kuhn.txt).  Leftover single nodes join their closest neighbour cluster.
void aggregate_pairwise(const HostPattern& F, int32_t n_active, int max_agg, const double* pts, bool strong_only,
                        std::vector<int32_t>& agg, int32_t& nc) {
    std::vector<int32_t> of((size_t)F.n, -1);
    for (int32_t i = 0; i < n_active; ++i) of[i] = i;
    int32_t ncl = n_active;
    std::vector<int32_t> ptr, idx, size_, newid, adj;
    std::vector<double> cen;
    std::vector<uint8_t> matched;
    auto members = [&]() {
        ptr.assign((size_t)ncl + 1, 0);
        for (int32_t i = 0; i < n_active; ++i) ++ptr[(size_t)of[i] + 1];
        for (int32_t c = 0; c < ncl; ++c) ptr[(size_t)c + 1] += ptr[c];
        idx.resize((size_t)n_active);
        std::vector<int32_t> at(ptr.begin(), ptr.end() - 1);
        for (int32_t i = 0; i < n_active; ++i) idx[(size_t)at[of[i]]++] = i;
        cen.assign((size_t)3 * ncl, 0.0);
        size_.assign((size_t)ncl, 0);
        for (int32_t c = 0; c < ncl; ++c) {
            size_[c] = ptr[(size_t)c + 1] - ptr[c];
            for (int32_t m = ptr[c]; m < ptr[(size_t)c + 1]; ++m)
                for (int k = 0; k < 3; ++k) cen[3 * (size_t)c + k] += pts[3 * (size_t)idx[m] + k];
            if (size_[c]) for (int k = 0; k < 3; ++k) cen[3 * (size_t)c + k] /= size_[c];
        }
    };
    auto cdist2 = [&](int32_t a, int32_t b) {
        double d2 = 0.0;
        for (int k = 0; k < 3; ++k) { const double d = cen[3 * (size_t)a + k] - cen[3 * (size_t)b + k]; d2 += d * d; }
        return d2;
    };
    auto neighbours = [&](int32_t a) {                          // clusters adjacent to a, ascending, without a
        adj.clear();
        for (int32_t m = ptr[a]; m < ptr[(size_t)a + 1]; ++m) {
            const int32_t i = idx[m];
            for (int32_t k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k) {
                const int32_t j = F.colind[k];
                if (j >= n_active) continue;
                const int32_t b = of[j];
                if (b != a) adj.push_back(b);
            }
        }
        std::sort(adj.begin(), adj.end());
        adj.erase(std::unique(adj.begin(), adj.end()), adj.end());
    };
    for (int round = 1; (1 << round) <= max_agg; ++round) {
        members();
        matched.assign((size_t)ncl, 0);
        newid.assign((size_t)ncl, -1);
        int32_t nnew = 0;
        for (int32_t a = 0; a < ncl; ++a) {
            if (matched[a]) continue;
            neighbours(a);
            double lo = 1e300;
            for (int32_t b : adj) lo = std::min(lo, cdist2(a, b));
            const double lim = strong_only ? ANISO_KEEP * lo * (1.0 + 1e-9) : 1e300;
            int32_t best = -1;
            double bd = 1e300;
            for (int32_t b : adj) {
                if (matched[b] || size_[a] + size_[b] > max_agg) continue;
                const double d2 = cdist2(a, b);
                if (d2 > lim) continue;
                if (best < 0 || d2 < bd * (1.0 - 1e-12)) { best = b; bd = d2; }
            }
            matched[a] = 1;
            newid[a] = nnew;
            if (best >= 0) { matched[best] = 1; newid[best] = nnew; }
            ++nnew;
        }
        for (int32_t i = 0; i < n_active; ++i) of[i] = newid[of[i]];
        ncl = nnew;
    }
    // single nodes left over join the closest adjacent cluster (the greedy sweep's rule for a node without a free neighbour)
    members();
    newid.assign((size_t)ncl, -1);
    std::vector<int32_t> target((size_t)ncl, -1);
    for (int32_t a = 0; a < ncl; ++a) {
        if (size_[a] != 1) continue;
        neighbours(a);
        double lo = 1e300;
        for (int32_t b : adj) lo = std::min(lo, cdist2(a, b));
        const double lim = strong_only ? ANISO_KEEP * lo * (1.0 + 1e-9) : 1e300;
        int32_t best = -1;
        double bd = 1e300;
        for (int32_t b : adj) {
            if (size_[b] < 2 || target[b] >= 0) continue;       // (not another single node, nor one that has moved)
            const double d2 = cdist2(a, b);
            if (d2 > lim) continue;
            if (best < 0 || d2 < bd * (1.0 - 1e-12)) { best = b; bd = d2; }
        }
        target[a] = best;
    }
    int32_t nnew = 0;
    for (int32_t a = 0; a < ncl; ++a)
        if (target[a] < 0) newid[a] = nnew++;
    for (int32_t a = 0; a < ncl; ++a)
        if (target[a] >= 0) newid[a] = newid[target[a]];
    agg.assign((size_t)F.n, -1);
    for (int32_t i = 0; i < n_active; ++i) agg[i] = newid[of[i]];
    nc = nnew;
}

}  // namespace

// Size-limited greedy aggregation.  With `pts` (3 coordinates per node of the level) on an ANISOTROPIC cloud a node only takes
// (or joins) neighbours within 2x its shortest edge -- the strong couplings of a diffusion-dominated operator scale with
// 1 / length^2, the block-Jacobi smoother handles exactly those, and aggregating along them alone is the semi-coarsening such
// meshes need; on every other cloud (all of rounds 1-4's meshes) the filter is off and the aggregates are those of round 1.
static void aggregate_greedy(const HostPattern& F, int32_t n_active, int max_agg, std::vector<int32_t>& agg,
                             int32_t& nc, const double* pts, bool strong_only) {
    agg.assign((size_t)F.n, -1);
    nc = 0;
    auto dist2 = [&](int32_t i, int32_t j) {
        double d2 = 0.0;
        for (int c = 0; c < 3; ++c) { const double d = pts[3 * (size_t)j + c] - pts[3 * (size_t)i + c]; d2 += d * d; }
        return d2;
    };
    for (int32_t i = 0; i < n_active; ++i) {
        if (agg[i] >= 0) continue;
        double lim = 1e300;
        if (strong_only) {
            double lo = 1e300;
            for (int32_t k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k)
                if (F.colind[k] != i) lo = std::min(lo, dist2(i, F.colind[k]));
            lim = ANISO_KEEP * lo * (1.0 + 1e-9);
        }
        int taken = 0;
        int32_t first_agg_nb = -1;
        for (int32_t k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k) {
            int32_t j = F.colind[k];
            if (j == i || j >= n_active) continue;
            if (strong_only && dist2(i, j) > lim) continue;
            if (agg[j] < 0) {
                if (taken < max_agg - 1) { agg[j] = nc; ++taken; }
            } else if (first_agg_nb < 0) first_agg_nb = agg[j];
        }
        if (taken == 0 && first_agg_nb >= 0) { agg[i] = first_agg_nb; continue; }
        agg[i] = nc++;
    }
}

// The aggregation of a level.  Without coordinates: the greedy sweep.  With coordinates: the greedy sweep AND the pairwise
// aggregation; the pairwise one is taken when its aggregates are more compact by more than 5 % (scatter_score) -- on the Kuhn
// lattices numbered along their diagonal the sweep's aggregates are cubes already and stay what they were in rounds 1-4.
// SNS_AGGREGATION=greedy|pairwise forces one (experiments); `which`, if given, reports the choice (0 greedy, 1 pairwise).
void aggregate_nodes(const HostPattern& F, int32_t n_active, int max_agg, std::vector<int32_t>& agg,
                     int32_t& nc, const double* pts, int* which) {
    const bool strong_only = cloud_is_anisotropic(F, n_active, pts);
    aggregate_greedy(F, n_active, max_agg, agg, nc, pts, strong_only);
    if (which) *which = 0;
    const char* force = std::getenv("SNS_AGGREGATION");
    if (!pts || n_active < 64 || max_agg < 2 || (force && force[0] == 'g')) return;
    // (the sweep's aggregates are as compact as cubes already -- the Kuhn lattices numbered along their diagonal, i.e. every
    // structured mesh of rounds 1-4: scatter per node at most 0.85 x (mean squared shortest edge) x (nodes per aggregate / 8)^(2/3),
    // a 2 x 2 x 2 cube has 0.75 --: nothing to gain, and the second aggregation of a 100 M-node level is not free)
    if (!(force && force[0] == 'p')) {
        const int32_t step = std::max(1, n_active / 50000);
        double lo_sum = 0.0;
        int64_t cnt = 0;
        for (int32_t i = 0; i < n_active; i += step) {
            double lo = 1e300;
            for (int32_t k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k) {
                const int32_t j = F.colind[k];
                if (j == i) continue;
                double d2 = 0.0;
                for (int c = 0; c < 3; ++c) { const double d = pts[3 * (size_t)j + c] - pts[3 * (size_t)i + c]; d2 += d * d; }
                lo = std::min(lo, d2);
            }
            if (lo < 1e300) { lo_sum += lo; ++cnt; }
        }
        if (cnt > 0 && nc > 0) {
            const double w_per_node = scatter_score(agg, nc, n_active, pts) / std::pow((double)nc, 2.0 / 3.0) / (double)n_active;
            if (w_per_node <= 0.85 * lo_sum / (double)cnt * std::pow((double)n_active / (double)nc / 8.0, 2.0 / 3.0)) return;
        }
    }
    std::vector<int32_t> agg2;
    int32_t nc2 = 0;
    aggregate_pairwise(F, n_active, max_agg, pts, strong_only, agg2, nc2);
    if (nc2 <= 0 || nc2 >= n_active) return;
    const double sg = scatter_score(agg, nc, n_active, pts), sp = scatter_score(agg2, nc2, n_active, pts);
    if (std::getenv("SNS_AGGREGATION_VERBOSE"))
        std::fprintf(stderr, "[sns] aggregation of %d nodes: greedy %d aggregates, score %.4g; pairwise %d, score %.4g\n", n_active, nc,
                     sg, nc2, sp);
    if ((force && force[0] == 'p') || sp < 0.95 * sg) {
        agg.swap(agg2);
        nc = nc2;
        if (which) *which = 1;
    }
}

// Coarse pattern, member lists and Galerkin gather lists from a complete aggregate map.
//   agg_all[j] : aggregate (LOCAL coarse id) of every local fine node; owned fine nodes map to
//                [0, nc_owned), ghost fine nodes to [nc_owned, nc_total) (aggregates never cross ranks;
//                a ghost node's aggregate is a ghost coarse node); -1 = node takes no part.
// Rows are built for owned aggregates only; ghost coarse rows stay empty.
void build_coarse_from_agg(const HostPattern& F, int32_t n_owned_fine, const std::vector<int32_t>& agg_all,
                           int32_t nc_owned, int32_t nc_total, HostAggregation& A) {
    A.agg = agg_all;
    A.nc = nc_owned;
    const int32_t nc = nc_owned;
    A.m_ptr.assign((size_t)nc + 1, 0);
    for (int32_t i = 0; i < n_owned_fine; ++i) A.m_ptr[A.agg[i] + 1]++;
    for (int32_t I = 0; I < nc; ++I) A.m_ptr[I + 1] += A.m_ptr[I];
    A.m_idx.resize((size_t)n_owned_fine);
    {
        std::vector<int32_t> cur(A.m_ptr.begin(), A.m_ptr.end() - 1);
        for (int32_t i = 0; i < n_owned_fine; ++i) A.m_idx[cur[A.agg[i]]++] = i;
    }
    HostPattern& C = A.coarse;
    C.n = nc_total;
    C.rowptr.assign((size_t)nc_total + 1, 0);
    std::vector<std::vector<int32_t>> rows((size_t)nc);
#pragma omp parallel
    {
        std::vector<int32_t> tmp;
#pragma omp for schedule(dynamic, 256)
        for (int32_t I = 0; I < nc; ++I) {
            tmp.clear();
            for (int32_t m = A.m_ptr[I]; m < A.m_ptr[I + 1]; ++m) {
                int32_t i = A.m_idx[m];
                for (int32_t k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k) {
                    int32_t J = A.agg[F.colind[k]];
                    if (J >= 0) tmp.push_back(J);
                }
            }
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            rows[I] = tmp;
        }
    }
    for (int32_t I = 0; I < nc; ++I) C.rowptr[I + 1] = C.rowptr[I] + (int32_t)rows[I].size();
    for (int32_t I = nc; I < nc_total; ++I) C.rowptr[I + 1] = C.rowptr[I];
    C.nnzb = C.rowptr[nc_total];
    C.colind.resize((size_t)C.nnzb);
    C.diag.assign((size_t)nc_total, 0);
    A.r_ptr.assign((size_t)C.nnzb + 1, 0);
#pragma omp parallel for schedule(dynamic, 256)
    for (int32_t I = 0; I < nc; ++I) {
        std::copy(rows[I].begin(), rows[I].end(), C.colind.begin() + C.rowptr[I]);
        C.diag[I] = C.rowptr[I] + (int32_t)(std::lower_bound(rows[I].begin(), rows[I].end(), I) - rows[I].begin());
        for (int32_t m = A.m_ptr[I]; m < A.m_ptr[I + 1]; ++m) {
            int32_t i = A.m_idx[m];
            for (int32_t k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k) {
                int32_t J = A.agg[F.colind[k]];
                if (J < 0) continue;
                int32_t s = C.rowptr[I] + (int32_t)(std::lower_bound(rows[I].begin(), rows[I].end(), J) - rows[I].begin());
                A.r_ptr[(size_t)s + 1]++;
            }
        }
    }
    for (int64_t s = 0; s < C.nnzb; ++s) A.r_ptr[s + 1] += A.r_ptr[s];
    A.r_idx.resize((size_t)A.r_ptr[C.nnzb]);
#pragma omp parallel
    {
        std::vector<int32_t> fill;
#pragma omp for schedule(dynamic, 256)
        for (int32_t I = 0; I < nc; ++I) {
            fill.assign(rows[I].size(), 0);
            for (int32_t m = A.m_ptr[I]; m < A.m_ptr[I + 1]; ++m) {
                int32_t i = A.m_idx[m];
                for (int32_t k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k) {
                    int32_t J = A.agg[F.colind[k]];
                    if (J < 0) continue;
                    int32_t jj = (int32_t)(std::lower_bound(rows[I].begin(), rows[I].end(), J) - rows[I].begin());
                    int64_t s = C.rowptr[I] + jj;
                    A.r_idx[(size_t)(A.r_ptr[s] + fill[jj]++)] = k;
                }
            }
        }
    }
}

// Pattern of M = A P (fine rows x coarse columns) for the fused post-smoothing sweep (k_post_lp): row i holds one
// block per aggregate its columns fall into; ap_ptr / ap_idx list, per M slot, the fine slots summed into it (fine-slot
// order: deterministic).  Columns whose node takes no part in the transfer (agg < 0) are dropped.
void build_ap_pattern(const HostPattern& F, int32_t n_rows, const std::vector<int32_t>& agg_all, HostAP& M) {
    M.n = n_rows;
    M.rowptr.assign((size_t)n_rows + 1, 0);
    std::vector<int32_t> cnt((size_t)std::max(1, n_rows), 0);
#pragma omp parallel
    {
        std::vector<int32_t> tmp;
#pragma omp for schedule(dynamic, 1024)
        for (int32_t i = 0; i < n_rows; ++i) {
            tmp.clear();
            for (int32_t k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k) {
                const int32_t J = agg_all[F.colind[k]];
                if (J >= 0) tmp.push_back(J);
            }
            std::sort(tmp.begin(), tmp.end());
            cnt[i] = (int32_t)(std::unique(tmp.begin(), tmp.end()) - tmp.begin());
        }
    }
    int64_t nnz = 0;
    for (int32_t i = 0; i < n_rows; ++i) nnz += cnt[i];
    if (nnz > (int64_t)INT32_MAX) throw std::runtime_error("A*P pattern exceeds int32 slots");
    for (int32_t i = 0; i < n_rows; ++i) M.rowptr[i + 1] = M.rowptr[i] + cnt[i];
    M.nnz = nnz;
    M.colind.resize((size_t)nnz);
    M.slot_row.resize((size_t)nnz);
    M.ap_ptr.assign((size_t)nnz + 1, 0);
    M.ap_idx.resize((size_t)(n_rows > 0 ? F.rowptr[n_rows] : 0));
#pragma omp parallel
    {
        std::vector<int32_t> tmp, fill;
#pragma omp for schedule(dynamic, 1024)
        for (int32_t i = 0; i < n_rows; ++i) {
            tmp.clear();
            for (int32_t k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k) {
                const int32_t J = agg_all[F.colind[k]];
                if (J >= 0) tmp.push_back(J);
            }
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            const int32_t m0 = M.rowptr[i];
            std::copy(tmp.begin(), tmp.end(), M.colind.begin() + m0);
            for (size_t q = 0; q < tmp.size(); ++q) M.slot_row[(size_t)m0 + q] = i;
            for (int32_t k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k) {
                const int32_t J = agg_all[F.colind[k]];
                if (J < 0) continue;
                const int32_t q = (int32_t)(std::lower_bound(tmp.begin(), tmp.end(), J) - tmp.begin());
                M.ap_ptr[(size_t)m0 + q + 1]++;
            }
        }
    }
    // ap_ptr currently holds per-slot counts (shifted by one): prefix sum, then fill in fine-slot order
    for (int64_t s = 0; s < nnz; ++s) M.ap_ptr[s + 1] += M.ap_ptr[s];
    M.ap_idx.resize((size_t)M.ap_ptr[nnz]);
#pragma omp parallel
    {
        std::vector<int32_t> fill;
#pragma omp for schedule(dynamic, 1024)
        for (int32_t i = 0; i < n_rows; ++i) {
            const int32_t m0 = M.rowptr[i], len = M.rowptr[i + 1] - m0;
            fill.assign((size_t)len, 0);
            const int32_t* cb = M.colind.data() + m0;
            for (int32_t k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k) {
                const int32_t J = agg_all[F.colind[k]];
                if (J < 0) continue;
                const int32_t q = (int32_t)(std::lower_bound(cb, cb + len, J) - cb);
                M.ap_idx[(size_t)(M.ap_ptr[(size_t)m0 + q] + fill[q]++)] = k;
            }
        }
    }
    // block -> slot nibbles (k_lp_copies16 accumulates M from the row it holds in registers, in block order = gather-list order)
    M.nib.assign((size_t)n_rows, ~0ull);
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < n_rows; ++i) {
        const int32_t s = F.rowptr[i], cnt = F.rowptr[i + 1] - s;
        const int32_t m0 = M.rowptr[i], cm = M.rowptr[i + 1] - m0;
        if (cnt > 16 || cm > 8) continue;
        const int32_t* cb = M.colind.data() + m0;
        uint64_t v = 0;
        for (int32_t j = 0; j < 16; ++j) {
            uint64_t t = 15;
            if (j < cnt) {
                const int32_t J = agg_all[F.colind[s + j]];
                if (J >= 0) t = (uint64_t)(std::lower_bound(cb, cb + cm, J) - cb);
            }
            v |= t << (4 * j);
        }
        M.nib[i] = v;
    }
}

void build_aggregation_active(const HostPattern& F, int32_t n_active, int max_agg, HostAggregation& A) {
    std::vector<int32_t> agg;
    int32_t nc = 0;
    aggregate_nodes(F, n_active, max_agg, agg, nc);
    build_coarse_from_agg(F, n_active, agg, nc, nc, A);
}

void build_aggregation(const HostPattern& fine, int max_agg, HostAggregation& A) {
    build_aggregation_active(fine, fine.n, max_agg, A);
}

}  // namespace sns

// ---- host-only C ABI (include/sns.h) ------------------------------------------------
extern "C" int sns_host_pattern(int32_t n, int64_t E, const int32_t* tets, int64_t* nnzb_out, int32_t* rowptr,
                                int32_t* colind, int64_t* c_ptr, int32_t* c_idx) {
    if (n <= 0 || E < 0 || !tets) { sns::set_error("sns_host_pattern: bad arguments"); return SNS_E_ARG; }
    sns::HostPattern P;
    sns::HostAssemblyMaps M;
    try {
        sns::build_pattern(n, E, tets, P, M, 4);
    } catch (const std::exception& e) {
        sns::set_error(e.what());
        return SNS_E_MESH;
    }
    if (nnzb_out) *nnzb_out = P.nnzb;
    if (rowptr) std::copy(P.rowptr.begin(), P.rowptr.end(), rowptr);
    if (colind) std::copy(P.colind.begin(), P.colind.end(), colind);
    if (c_ptr) std::copy(M.c_ptr.begin(), M.c_ptr.end(), c_ptr);
    if (c_idx) std::copy(M.c_idx.begin(), M.c_idx.end(), c_idx);
    return SNS_OK;
}

extern "C" int sns_host_aggregate_pts(int32_t n, const int32_t* rowptr, const int32_t* colind, int32_t n_active,
                                      int max_agg, const double* pts, int32_t* agg_out, int32_t* n_agg_out, int32_t* which_out);
extern "C" int sns_host_aggregate(int32_t n, const int32_t* rowptr, const int32_t* colind, int32_t n_active,
                                  int max_agg, int32_t* agg_out, int32_t* n_agg_out) {
    return sns_host_aggregate_pts(n, rowptr, colind, n_active, max_agg, nullptr, agg_out, n_agg_out, nullptr);
}

extern "C" int sns_host_aggregate_pts(int32_t n, const int32_t* rowptr, const int32_t* colind, int32_t n_active,
                                      int max_agg, const double* pts, int32_t* agg_out, int32_t* n_agg_out, int32_t* which_out) {
    if (n <= 0 || !rowptr || !colind || !agg_out || n_active < 0 || n_active > n || max_agg < 1) {
        sns::set_error("sns_host_aggregate: bad arguments");
        return SNS_E_ARG;
    }
    sns::HostPattern F;
    F.n = n;
    F.rowptr.assign(rowptr, rowptr + n + 1);
    F.nnzb = rowptr[n];
    F.colind.assign(colind, colind + F.nnzb);
    std::vector<int32_t> agg;
    int32_t nc = 0;
    int which = 0;
    sns::aggregate_nodes(F, n_active, max_agg, agg, nc, pts, &which);
    std::copy(agg.begin(), agg.end(), agg_out);
    if (n_agg_out) *n_agg_out = nc;
    if (which_out) *which_out = which;
    return SNS_OK;
}

// owned rows (i < n_owned) of a local pattern with at least one ghost column (>= n_owned): the boundary rows of
// the interior / boundary split of the multi-GPU SpMV (the interior rows never wait for the halo)
extern "C" int sns_host_boundary_rows(int32_t n_owned, const int32_t* rowptr, const int32_t* colind, int32_t* rows_out,
                                      int32_t* n_out) {
    if (n_owned < 0 || !rowptr || !colind || !rows_out || !n_out) { sns::set_error("sns_host_boundary_rows: bad arguments"); return SNS_E_ARG; }
    int32_t m = 0;
    for (int32_t i = 0; i < n_owned; ++i)
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
            if (colind[k] >= n_owned) { rows_out[m++] = i; break; }
    *n_out = m;
    return SNS_OK;
}

// Eigenvalues of a small real upper-Hessenberg matrix (row-major n x n; entries below the first subdiagonal are ignored): shifted
// QR with Givens rotations in complex arithmetic (Wilkinson shift, deflation, an exceptional shift every 10th sweep).  Serves the
// Ritz values of the short Arnoldi process that bounds the smoother damping (sns_api.hip: arnoldi_ritz); n <= 32.
extern "C" int sns_host_hessenberg_eigs(int n, const double* H, double* re, double* im) {
    if (n <= 0 || n > 32 || !H || !re || !im) return SNS_E_ARG;
    typedef std::complex<double> cd;
    std::vector<cd> A((size_t)n * n, cd(0.0, 0.0));
    for (int i = 0; i < n; ++i)
        for (int j = std::max(0, i - 1); j < n; ++j) A[(size_t)i * n + j] = cd(H[(size_t)i * n + j], 0.0);
    auto at = [&](int i, int j) -> cd& { return A[(size_t)i * n + j]; };
    std::vector<cd> ev((size_t)n), cs((size_t)n), sn((size_t)n);
    int hi = n - 1, iter = 0;
    while (hi >= 0) {
        if (hi == 0) { ev[0] = at(0, 0); break; }
        int lo = hi;
        while (lo > 0) {
            double sc = std::abs(at(lo - 1, lo - 1)) + std::abs(at(lo, lo));
            if (sc == 0.0) sc = 1.0;
            if (std::abs(at(lo, lo - 1)) < 1e-15 * sc) { at(lo, lo - 1) = 0.0; break; }
            --lo;
        }
        if (lo == hi) { ev[(size_t)hi] = at(hi, hi); --hi; iter = 0; continue; }
        if (++iter > 300) {                                  // no convergence (never seen): the diagonal as it stands
            for (int i = lo; i <= hi; ++i) ev[(size_t)i] = at(i, i);
            hi = lo - 1; iter = 0;
            continue;
        }
        const cd a = at(hi - 1, hi - 1), b = at(hi - 1, hi), c = at(hi, hi - 1), d = at(hi, hi);
        const cd tr = a + d, det = a * d - b * c, disc = std::sqrt(tr * tr * 0.25 - det);
        const cd m1 = tr * 0.5 + disc, m2 = tr * 0.5 - disc;
        cd mu = std::abs(m1 - d) < std::abs(m2 - d) ? m1 : m2;
        if (iter % 10 == 0) mu += cd(std::abs(at(hi, hi - 1)), 0.5 * std::abs(at(hi, hi - 1)));      // exceptional shift
        for (int i = lo; i <= hi; ++i) at(i, i) -= mu;
        for (int k = lo; k < hi; ++k) {                      // H - mu I = Q R
            const cd x = at(k, k), y = at(k + 1, k);
            const double r = std::sqrt(std::norm(x) + std::norm(y));
            const cd cc = r == 0.0 ? cd(1.0, 0.0) : x / r, ss = r == 0.0 ? cd(0.0, 0.0) : y / r;
            cs[(size_t)k] = cc; sn[(size_t)k] = ss;
            for (int j = k; j <= hi; ++j) {
                const cd t1 = at(k, j), t2 = at(k + 1, j);
                at(k, j) = std::conj(cc) * t1 + std::conj(ss) * t2;
                at(k + 1, j) = -ss * t1 + cc * t2;
            }
        }
        for (int k = lo; k < hi; ++k) {                      // R Q + mu I
            const cd cc = cs[(size_t)k], ss = sn[(size_t)k];
            for (int i = lo; i <= std::min(k + 1, hi); ++i) {
                const cd t1 = at(i, k), t2 = at(i, k + 1);
                at(i, k) = t1 * cc + t2 * ss;
                at(i, k + 1) = -t1 * std::conj(ss) + t2 * std::conj(cc);
            }
        }
        for (int i = lo; i <= hi; ++i) at(i, i) += mu;
    }
    for (int i = 0; i < n; ++i) { re[i] = ev[(size_t)i].real(); im[i] = ev[(size_t)i].imag(); }
    return SNS_OK;
}

// the hierarchy's policy table on its own (csrc/sns_policy.h: the one place that holds the thresholds and schedules)
extern "C" int sns_host_cycle_policy(const sns_options* opt, int nranks, int windows, int nlevels, const int64_t* rows_global,
                                     int rep_level, int64_t rows_global_l1, const uint8_t* has_blocks, int32_t* kind,
                                     int32_t* nu_pre, int32_t* nu_post, int32_t* exact) {
    if (!opt || nranks < 1 || nlevels < 1 || nlevels > 16 || !rows_global || !kind || !nu_pre || !nu_post || rep_level < 0 ||
        rep_level >= nlevels || rep_level == 1) {
        sns::set_error("sns_host_cycle_policy: bad arguments (1..16 levels, rep_level 0 or in [2, nlevels))");
        return SNS_E_ARG;
    }
    bool hb[16];
    for (int l = 0; l < nlevels; ++l) hb[l] = has_blocks ? has_blocks[l] != 0 : true;
    sns::policy::LevelRow rows[16];
    sns::policy::cycle_table(*opt, nranks, windows != 0, nlevels, rows_global, rep_level, rows_global_l1, hb, rows);
    for (int l = 0; l < nlevels; ++l) {
        kind[l] = rows[l].cycled ? rows[l].kind : -1;
        nu_pre[l] = rows[l].pre;
        nu_post[l] = rows[l].post;
        if (exact) exact[l] = rows[l].exact;
    }
    return SNS_OK;
}
