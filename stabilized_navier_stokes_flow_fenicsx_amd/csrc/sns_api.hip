// C-ABI layer (include/sns.h): handle life cycle, communicator attachment, the entry points of the hot path (residual /
// Jacobian / SpMV / preconditioner / Krylov / Stokes / Newton) and the introspection getters.  Host code only launches kernels
// and moves scalars; there is no CPU compute path.  (Round 5: the setup, cycle and Krylov parts live in csrc/sns_setup.hip,
// sns_cycle.hip and sns_krylov.hip; shared internals in csrc/sns_ctx.h; the hierarchy's policy in csrc/sns_policy.h.)
#include "sns_ctx.h"

namespace sns {
static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }

}  // namespace sns

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
                                        (o->pc_type != h->opt.pc_type);
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

