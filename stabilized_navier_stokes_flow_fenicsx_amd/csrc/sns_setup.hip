// Setup side of the C-ABI layer: the symbolic AMG hierarchy (aggregation, halo plans per level, replicated tail), the assembly
// driver, the spectral estimates of the smoother damping and the numeric setup of the preconditioner.
// (round 5: one of the four translation units csrc/sns_api.hip was split into; shared internals in csrc/sns_ctx.h)
#include "sns_ctx.h"

namespace sns {

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
    const policy::CoarsestKind ck = policy::coarsest_kind(o, last.n);
    if (ck == policy::COARSEST_SMALL_INVERSE) {
        const size_t N = 4 * (size_t)last.n;
        SNS_TRY(dev_alloc(&last.dense_inv, N * N));
        SNS_TRY(dev_alloc(&h->d_piv, N));
    } else if (ck == policy::COARSEST_BLOCKED_INVERSE) {
        const int Np = (4 * last.n + 63) / 64 * 64;
        last.dense_np = Np;
        SNS_TRY(dev_alloc(&last.dense_gj, (size_t)Np * Np));
        SNS_TRY(dev_alloc(&last.dense_work, dense_gj_work_doubles(Np)));
        SNS_TRY(dev_alloc(&last.dense_x32, (size_t)Np * Np));
    }
    return SNS_OK;
}

// Multi-GPU: from level R on, every rank holds the GLOBAL operator (values all-gathered at every numeric setup)
// and cycles the rest of the hierarchy redundantly: no exchanges below R, and the smoothing there is the exact
// global block-Jacobi instead of a rank-local one (thin partitions lose their convergence on the deep levels
// otherwise).  `cur` is the local pattern of level R (owned rows, local column ids), collective over the ranks.
int build_replicated_tail(sns_ctx* h, int R, const HostPattern& cur, int32_t n_owned, const std::vector<double>& cur_pts) {
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
    // the replicated level's node coordinates (all-gathered like its pattern), so that its aggregation sees shapes, not only numbers
    std::vector<double> gpts;
    {
        double have[1] = {cur_pts.size() == (size_t)3 * cur.n ? 0.0 : 1.0};
        SNS_TRY(global_sum(h, have, 1));
        if (have[0] == 0.0) {
            std::vector<double> pm((size_t)3 * maxn, 0.0), pa;
            for (int32_t i = 0; i < n_owned; ++i)
                for (int k = 0; k < 3; ++k) pm[3 * (size_t)i + k] = cur_pts[3 * (size_t)i + k];
            SNS_TRY(host_allgather(h, pm, pa));
            gpts.resize((size_t)3 * NG);
            for (int r = 0; r < nr; ++r)
                for (int32_t i = 0; i < (int32_t)cnt[r]; ++i)
                    for (int k = 0; k < 3; ++k) gpts[3 * ((size_t)off[r] + i) + k] = pa[(size_t)r * pm.size() + 3 * (size_t)i + k];
        }
    }
    // plain serial aggregation below (identical on every rank: same input, deterministic code)
    HostPattern curp = std::move(G);
    int32_t n_own = NG;
    for (int l = h->rep_level; (int)h->levels.size() < o.amg_max_levels + 1; ++l) {
        if (n_own <= coarsest_rows(o)) break;
        std::vector<int32_t> agg;
        int32_t nc = 0;
        aggregate_nodes(curp, n_own, std::min(255, std::max(2, o.amg_agg_size)), agg, nc,
                        gpts.size() == (size_t)3 * n_own ? gpts.data() : nullptr);
        if (nc >= n_own || nc == 0) break;
        if (gpts.size() == (size_t)3 * n_own) {              // the next level's nodes: the aggregates' centroids
            std::vector<double> cp((size_t)3 * nc, 0.0);
            std::vector<int32_t> cn((size_t)nc, 0);
            for (int32_t i = 0; i < n_own; ++i) {
                const int32_t I = agg[i];
                if (I < 0) continue;
                for (int k = 0; k < 3; ++k) cp[3 * (size_t)I + k] += gpts[3 * (size_t)i + k];
                ++cn[I];
            }
            for (int32_t I = 0; I < nc; ++I)
                if (cn[I]) for (int k = 0; k < 3; ++k) cp[3 * (size_t)I + k] /= cn[I];
            gpts = std::move(cp);
        }
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
            if (policy::replicate_from(o, l, (int64_t)g[0], fits[0] == 0.0))
                return build_replicated_tail(h, l, cur, n_owned, cur_pts);
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
            if (N <= policy::DISTRIBUTED_DENSE_MAX_DOFS) {
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
            // the growth check of rounds 1-3 (back off until a sweep contracts the dominant mode by >= 10 %), on levels that run >= 3
            // sweeps per cycle only: a level with two sweeps per cycle (the fine level, V(1,1)) does not compound an amplified mode,
            // and backing its damping off for the sake of a few complex outliers weakens the smoothing of everything else
            // (jittered 120 x 30 x 30 duct, Re 200: 56 iterations this way, 82 with the check on every level, 85 without it; the
            // option that switched between the three, amg_growth_check, was retired in round 5)
            bool check_growth = false;
            if (l + 1 < nl) {
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
    if (h->comm && h->comm->peer && h->opt.pc_type == SNS_PC_AMG)
        for (int l = 0; l < nl && l < (int)h->comm->plans.size(); ++l) {
            const Level& L = h->levels[l];
            if (L.blk_rows && L.n_blk > 0 && L.binv32)        // the puts carried by this level's block kernels (comm_put_dst)
                SNS_TRY(comm_plan_put_groups(h->comm.get(), h->comm->plans[l], L.blk_rows, 8 * L.n_blk, h->stream));
        }
    h->pc_ready = true;
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


}  // namespace sns
