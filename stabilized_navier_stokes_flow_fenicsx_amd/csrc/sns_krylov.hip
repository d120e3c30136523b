// Krylov methods on the device (BiCGStab with device-resident scalars, TFQMR, FGMRES) and the solve driver with the damping retry.
// (round 5: one of the four translation units csrc/sns_api.hip was split into; shared internals in csrc/sns_ctx.h)
#include "sns_ctx.h"

namespace sns {

int norm2(sns_ctx* h, const double* x, double* out) {
    const int64_t nd = nred_of(h);
    const int g = vec_grid(nd);
    hipLaunchKernelGGL(k_dot2, dim3(g), dim3(256), 0, h->stream, nd, x, x, h->partial);
    SNS_TRY(reduce_to(h, g, 2, h->d_scal));
    double v[2];
    SNS_TRY(fetch(h, h->d_scal, 2, v));
    *out = std::sqrt(v[0]);
    return SNS_OK;
}

int dot(sns_ctx* h, const double* x, const double* y, double* out) {
    const int64_t nd = nred_of(h);
    const int g = vec_grid(nd);
    hipLaunchKernelGGL(k_dot2, dim3(g), dim3(256), 0, h->stream, nd, x, y, h->partial);
    SNS_TRY(reduce_to(h, g, 2, h->d_scal));
    double v[2];
    SNS_TRY(fetch(h, h->d_scal, 2, v));
    *out = v[0];
    return SNS_OK;
}


// Can the Krylov kernel that writes the preconditioner's input also do the V-cycle's first fine-level sweep z = w D^-1 (input)
// (k_bicg_s_first / k_bicg_xrp_first: one dependent launch and one read of the input less per cycle)?  Returns the buffer the
// cycle of pc_apply(., zdst) starts from, or nullptr.
double* fused_first_sweep_target(sns_ctx* h, double* zdst) {
    if (h->opt.pc_type != SNS_PC_AMG || h->levels.size() < 2 || !h->pc_ready) return nullptr;
    const Level& L = h->levels[0];
    if (lp_format(h, L) == 0 || !L.dinv32 || L.n_owned <= 0) return nullptr;
    if (block_active(h, 0) && (!L.binv32 || L.n_blk <= 0)) return nullptr;      // (aggregate blocks: k_bfirst_bicg, see fused_vector_kernel)
    if (h->rep_level == 1) return nullptr;                       // level 0 is only the source of the replicated copy
    double* x = (h->n > h->n_owned && !fine_tails_unused(h)) ? h->levels[0].x : zdst;   // (as pc_apply chooses the cycle's vector)
    return cycle_start_buffer(h, 0, x);
}


// ---- BiCGStab (right-preconditioned; the recurrences of oracle/solve.py:bicgstab_bj) ----
// Latency-lean formulation: rho / alpha / omega / beta live on the device (sc[]), the vector kernels read them
// there, and the three reductions of the textbook iteration are two -- <rhat, v>, then ONE pass for
// (t.s, t.t, rhat.s, rhat.t, s.s), from which omega, the next rho and ||r||^2 follow (k_bicg_dots5).  The host
// reads (||r||^2, flags) once per iteration, asynchronously: the copy is enqueued, then the x/r update and the
// FIRST HALF of the next iteration (p, M p, A M p, <rhat, v>, alpha: none of it touches x or r) are enqueued
// behind it, and only then does the host wait for the copy's event -- the GPU never idles on the stopping test.
// A converged claim is confirmed by the explicitly computed ||r|| before the loop is left.
int bicgstab(sns_ctx* h, const double* b, double* x, int* its_out, int* reason_out, double* rnorm_out, int stall_window) {
    const sns_options& o = h->opt;
    const int64_t nd = nred_of(h);
    const int g = vec_grid(nd);
    double *r, *rhat, *p, *v, *s, *t, *ph, *sh;
    SNS_TRY(get_vec(h, 0, &r)); SNS_TRY(get_vec(h, 1, &rhat)); SNS_TRY(get_vec(h, 2, &p));
    SNS_TRY(get_vec(h, 3, &v)); SNS_TRY(get_vec(h, 4, &s)); SNS_TRY(get_vec(h, 5, &t));
    SNS_TRY(get_vec(h, 6, &ph)); SNS_TRY(get_vec(h, 7, &sh));
    double* sc = h->d_scal + 128;                         // device scalar block of this solver
    double* red = h->d_scal + 144;                        // reduction results
    double* hpin = h->h_scal + 512;                       // pinned landing zone of (rr, flags)
    if (!h->ev_it) HIP_TRY(hipEventCreateWithFlags(&h->ev_it, hipEventDisableTiming));
    double bnorm, rn;
    SNS_TRY(norm2(h, b, &bnorm));
    SNS_TRY(op_residual(h, x, b, r));
    // ||r0||^2 stays on the device as the first rho (rhat = r0); the host needs it for the start-up test
    hipLaunchKernelGGL(k_dot2, dim3(g), dim3(256), 0, h->stream, nd, r, r, h->partial);
    SNS_TRY(reduce_to(h, g, 2, red));
    hipLaunchKernelGGL(k_bicg_init, dim3(1), dim3(64), 0, h->stream, sc, red);
    {
        double v0[2];
        SNS_TRY(fetch(h, red, 2, v0));
        rn = std::sqrt(v0[0]);
    }
    const double tol = std::max(o.ksp_rtol * bnorm, o.ksp_atol);
    if (o.monitor) std::printf("  0 KSP Residual norm %.12e\n", rn);
    int its = 0, reason = 0;
    if (!(rn == rn)) reason = SNS_KSP_DIVERGED_NANORINF;
    else if (rn <= tol) reason = (rn <= o.ksp_atol) ? SNS_KSP_CONVERGED_ATOL : SNS_KSP_CONVERGED_RTOL;
    if (!reason && o.ksp_max_it < 1) reason = SNS_KSP_DIVERGED_ITS;
    if (!reason) {
        HIP_TRY(hipMemcpyAsync(rhat, r, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemsetAsync(p, 0, nd * sizeof(double), h->stream));
        HIP_TRY(hipMemsetAsync(v, 0, nd * sizeof(double), h->stream));
        auto first_half = [&](bool p_done) -> int {       // p, ph = M p, v = A ph, alpha
            if (!p_done) hipLaunchKernelGGL(k_bicg_p, dim3(g), dim3(256), 0, h->stream, nd, r, sc, v, p);
            h->pc_then_op = true;
            SNS_TRY(pc_apply(h, p, ph));
            SNS_TRY(op_apply_dot(h, ph, v, rhat));        // v = A ph with the fused partial sums of <rhat, v>
            SNS_TRY(reduce_bicg<1>(h, h->dot_partials, red, sc));                  // alpha
            return SNS_OK;
        };
        SNS_TRY(first_half(false));
        const double rn0 = rn;
        double best_rn = rn;
        int best_it = 0;
        for (its = 1;; ++its) {
            if (double* z1 = fused_first_sweep_target(h, sh)) {
                const Level& L0 = h->levels[0];
                if (block_active(h, 0)) {
                    const int32_t ns = 8 * L0.n_blk;
                    const unsigned gb = (unsigned)((ns + 63) / 64);
                    const PutDst pd0 = first_sweep_put(h);
                    h->first_put_carried = pd0.sr_ptr != nullptr;
                    if (L0.binv_fmt == 2)
                        hipLaunchKernelGGL((k_bfirst_bicg<2, 1>), dim3(gb), dim3(256), 0, h->stream, ns, L0.blk_rows, (const void*)L0.binv32,
                                           L0.omega, z1, sc, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, v,
                                           (double*)nullptr, r, (double*)nullptr, s, pd0);
                    else
                        hipLaunchKernelGGL((k_bfirst_bicg<1, 1>), dim3(gb), dim3(256), 0, h->stream, ns, L0.blk_rows, (const void*)L0.binv32,
                                           L0.omega, z1, sc, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, v,
                                           (double*)nullptr, r, (double*)nullptr, s, pd0);
                } else {
                    hipLaunchKernelGGL(k_bicg_s_first, dim3(g), dim3(256), 0, h->stream, nd, r, sc, v, s, L0.dinv32, L0.omega, z1);
                }
                h->first_sweep_done = true;
            } else {
                hipLaunchKernelGGL(k_bicg_s, dim3(g), dim3(256), 0, h->stream, nd, r, sc, v, s);
            }
            h->pc_then_op = true;
            SNS_TRY(pc_apply(h, s, sh));
            SNS_TRY(op_apply(h, sh, t));
            hipLaunchKernelGGL(k_bicg_dots5, dim3(g), dim3(256), 0, h->stream, nd, s, t, rhat, h->partial);
            SNS_TRY(reduce_bicg<2>(h, g, red, sc));                       // omega, next rho / beta, ||r||^2, flags
            HIP_TRY(hipMemcpyAsync(hpin, sc + 4, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipEventRecord(h->ev_it, h->stream));
            // speculative first half of the next iteration, enqueued BEFORE the host looks at this one's result; its p-update
            // rides on the x / r update (k_bicg_xrp)
            const bool spec = its < o.ksp_max_it;
            if (spec) {
                if (double* z1 = fused_first_sweep_target(h, ph)) {
                    const Level& L0 = h->levels[0];
                    if (block_active(h, 0)) {
                        const int32_t ns = 8 * L0.n_blk;
                        const unsigned gb = (unsigned)((ns + 63) / 64);
                        const PutDst pd0 = first_sweep_put(h);
                        h->first_put_carried = pd0.sr_ptr != nullptr;
                        if (L0.binv_fmt == 2)
                            hipLaunchKernelGGL((k_bfirst_bicg<2, 2>), dim3(gb), dim3(256), 0, h->stream, ns, L0.blk_rows,
                                               (const void*)L0.binv32, L0.omega, z1, sc, (const double*)ph, (const double*)sh,
                                               (const double*)t, (const double*)v, x, r, p, s, pd0);
                        else
                            hipLaunchKernelGGL((k_bfirst_bicg<1, 2>), dim3(gb), dim3(256), 0, h->stream, ns, L0.blk_rows,
                                               (const void*)L0.binv32, L0.omega, z1, sc, (const double*)ph, (const double*)sh,
                                               (const double*)t, (const double*)v, x, r, p, s, pd0);
                    } else {
                        hipLaunchKernelGGL(k_bicg_xrp_first, dim3(g), dim3(256), 0, h->stream, nd, sc, ph, sh, s, t, v, x, r, p,
                                           L0.dinv32, L0.omega, z1);
                    }
                    h->first_sweep_done = true;
                } else {
                    hipLaunchKernelGGL(k_bicg_xrp, dim3(g), dim3(256), 0, h->stream, nd, sc, ph, sh, s, t, v, x, r, p);
                }
                SNS_TRY(first_half(true));
            } else {
                hipLaunchKernelGGL(k_bicg_xr, dim3(g), dim3(256), 0, h->stream, nd, sc, ph, sh, s, t, x, r);
            }
            HIP_TRY(hipEventSynchronize(h->ev_it));
            ++h->ctr_host_syncs;
            SNS_TRY(peer_check(h->comm.get()));
            const double rr = hpin[0];
            const int flags = (int)hpin[1];
            rn = std::sqrt(rr);
            if (o.monitor) std::printf("%3d KSP Residual norm %.12e\n", its, rn);
            if ((flags & 1) || !(rn == rn) || std::isinf(rn)) { reason = SNS_KSP_DIVERGED_NANORINF; break; }
            if (rn <= tol) {
                // the three-term formula can lose digits when ||r|| << ||s||: confirm with the vector itself
                double rtrue;
                SNS_TRY(norm2(h, r, &rtrue));
                if (rtrue <= tol) {
                    rn = rtrue;
                    reason = (rn <= o.ksp_atol) ? SNS_KSP_CONVERGED_ATOL : SNS_KSP_CONVERGED_RTOL;
                    break;
                }
            }
            if (flags & 2) { reason = SNS_KSP_DIVERGED_BREAKDOWN; break; }
            if (its >= o.ksp_max_it) { reason = SNS_KSP_DIVERGED_ITS; break; }
            // stagnation watch of the damping-retry feature (stall_window > 0 only on an attempt that can still be retried):
            // BiCGStab under an over-relaxed smoother often does not break down outright but wanders without ever getting
            // anywhere.  Only REAL stagnation ends the attempt: no new best residual at all for stall_window
            // iterations, or, after stall_window iterations, a best residual still at or above the initial one.  A solve
            // that converges slowly -- BiCGStab plateaus on convection-dominated Jacobians -- keeps setting new bests and is
            // left alone, like PETSc's bcgs would leave it.  The attempt's reason is SNS_KSP_STALLED (not a breakdown).
            if (rn < best_rn) { best_rn = rn; best_it = its; }
            if (stall_window > 0 && (its - best_it >= stall_window || (its >= stall_window && best_rn >= rn0))) {
                reason = SNS_KSP_STALLED;
                break;
            }
            if (flags & 4) { reason = SNS_KSP_DIVERGED_BREAKDOWN; ++its; break; }   // rho == 0 stops the NEXT iteration
        }
        // the stopping test runs on the RECURRENCE residual (as PETSc's bcgs does); what is reported is the true one,
        // ||b - A x|| of the returned iterate, from one more operator pass (0.5 ms of a 145-ms solve at 10 M tets)
        if (reason != SNS_KSP_DIVERGED_NANORINF) {
            SNS_TRY(op_residual(h, x, b, t));
            SNS_TRY(norm2(h, t, &rn));
        }
    }
    *its_out = its;
    *reason_out = reason;
    *rnorm_out = rn;
    return SNS_OK;
}


// ---- TFQMR (Freund 1993) on B = A M^-1: the reference's KSP type ('tfqmr', :77, :199, :282) ----
// Same recurrences as oracle/c/sns_oracle.c:orc_solve(method=1).  The quasi-residual bound
// tau*sqrt(m+1) drives the stopping test (as in PETSc); the true residual is reported at the end.
int tfqmr(sns_ctx* h, const double* b, double* x, int* its_out, int* reason_out, double* rnorm_out) {
    const sns_options& o = h->opt;
    const int64_t nd = nred_of(h);
    const int g = vec_grid(nd);
    double *w, *y1, *y2, *u1, *u2, *d, *v, *xh, *rt, *tmp;
    SNS_TRY(get_vec(h, 0, &w)); SNS_TRY(get_vec(h, 1, &y1)); SNS_TRY(get_vec(h, 2, &y2)); SNS_TRY(get_vec(h, 3, &u1));
    SNS_TRY(get_vec(h, 4, &u2)); SNS_TRY(get_vec(h, 5, &d)); SNS_TRY(get_vec(h, 6, &v)); SNS_TRY(get_vec(h, 7, &xh));
    SNS_TRY(get_vec(h, 8, &rt)); SNS_TRY(get_vec(h, 9, &tmp));
    auto axpby = [&](double a, const double* xx, double bb, double* yy) {
        hipLaunchKernelGGL(k_axpby, dim3(g), dim3(256), 0, h->stream, nd, a, xx, bb, yy);
    };
    auto lin3 = [&](double a, const double* xx, double bb, const double* yy, double c, double* zz) {
        hipLaunchKernelGGL(k_axpbypcz, dim3(g), dim3(256), 0, h->stream, nd, a, xx, bb, yy, c, zz);
    };
    auto applyB = [&](const double* in, double* out) -> int {
        SNS_TRY(pc_apply(h, in, tmp));
        return op_apply(h, tmp, out);
    };
    double bnorm, rn;
    SNS_TRY(norm2(h, b, &bnorm));
    SNS_TRY(op_residual(h, x, b, w));
    SNS_TRY(norm2(h, w, &rn));
    const double tol = std::max(o.ksp_rtol * bnorm, o.ksp_atol);
    if (o.monitor) std::printf("  0 KSP Residual norm %.12e\n", rn);
    int its = 0, reason = 0;
    if (!(rn == rn)) reason = SNS_KSP_DIVERGED_NANORINF;
    else if (rn <= tol) reason = (rn <= o.ksp_atol) ? SNS_KSP_CONVERGED_ATOL : SNS_KSP_CONVERGED_RTOL;
    if (!reason) {
        HIP_TRY(hipMemcpyAsync(y1, w, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(rt, w, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        SNS_TRY(applyB(y1, v));
        HIP_TRY(hipMemcpyAsync(u1, v, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemsetAsync(d, 0, nd * sizeof(double), h->stream));
        HIP_TRY(hipMemsetAsync(xh, 0, nd * sizeof(double), h->stream));
        double tau = rn, theta = 0.0, eta = 0.0, rho = rn * rn;
        bool done = false;
        for (its = 1; its <= o.ksp_max_it && !done; ++its) {
            double sigma;
            SNS_TRY(dot(h, rt, v, &sigma));
            if (sigma == 0.0 || rho == 0.0) { reason = SNS_KSP_DIVERGED_BREAKDOWN; break; }
            const double alpha = rho / sigma;
            lin3(1.0, y1, -alpha, v, 0.0, y2);
            SNS_TRY(applyB(y2, u2));
            for (int m = 0; m < 2; ++m) {
                const double* um = m == 0 ? u1 : u2;
                const double* ym = m == 0 ? y1 : y2;
                axpby(-alpha, um, 1.0, w);
                axpby(1.0, ym, theta * theta * eta / alpha, d);
                double wn;
                SNS_TRY(norm2(h, w, &wn));
                theta = wn / tau;
                const double c = 1.0 / std::sqrt(1.0 + theta * theta);
                tau = tau * theta * c;
                eta = c * c * alpha;
                axpby(eta, d, 1.0, xh);
                rn = tau * std::sqrt((double)(2 * its - 1 + m) + 1.0);
                if (o.monitor) std::printf("%3d.%d KSP Residual bound %.12e\n", its, m, rn);
                if (!(rn == rn)) { reason = SNS_KSP_DIVERGED_NANORINF; done = true; break; }
                if (rn <= tol) { done = true; break; }
            }
            if (done) break;
            double rho_new;
            SNS_TRY(dot(h, rt, w, &rho_new));
            const double beta = rho_new / rho;
            rho = rho_new;
            lin3(1.0, w, beta, y2, 0.0, y1);
            SNS_TRY(applyB(y1, u1));
            lin3(1.0, u1, beta, u2, beta * beta, v);
        }
        if (its > o.ksp_max_it) its = o.ksp_max_it;
        SNS_TRY(pc_apply(h, xh, tmp));
        axpby(1.0, tmp, 1.0, x);
        SNS_TRY(op_residual(h, x, b, w));
        SNS_TRY(norm2(h, w, &rn));
        if (!reason) {
            if (done && rn <= 10.0 * tol) reason = (rn <= o.ksp_atol) ? SNS_KSP_CONVERGED_ATOL : SNS_KSP_CONVERGED_RTOL;
            else reason = SNS_KSP_DIVERGED_ITS;
        }
    }
    *its_out = its;
    *reason_out = reason;
    *rnorm_out = rn;
    return SNS_OK;
}


// ---- FGMRES(m), right preconditioning, classical Gram-Schmidt with one re-orthogonalisation ----
int fgmres(sns_ctx* h, const double* b, double* x, int* its_out, int* reason_out, double* rnorm_out) {
    const sns_options& o = h->opt;
    const int m = std::min(200, std::max(1, o.gmres_restart));
    const int64_t nd = nred_of(h), ld = ld_of(h);
    const int g = vec_grid(nd);
    if (h->gm_m != m) {
        if (h->gm_V) { (void)hipFree(h->gm_V); (void)hipFree(h->gm_Z); (void)hipFree(h->d_h); }
        SNS_TRY(dev_alloc(&h->gm_V, (size_t)(m + 1) * ld));
        SNS_TRY(dev_alloc(&h->gm_Z, (size_t)m * ld));
        SNS_TRY(dev_alloc(&h->d_h, (size_t)2 * (m + 16)));
        HIP_TRY(hipMemset(h->gm_V, 0, (size_t)(m + 1) * ld * sizeof(double)));
        HIP_TRY(hipMemset(h->gm_Z, 0, (size_t)m * ld * sizeof(double)));
        h->gm_m = m;
    }
    double* V = h->gm_V;
    double* Z = h->gm_Z;
    const int S = m + 16;                 // stride of one coefficient block
    double* dh1 = h->d_h;                 // pass-1 coefficients [0, m+8)
    double* dh2 = h->d_h + S;             // pass-2 coefficients [0, m+8), then (w.w, w.w) at [m+8, m+10)
    std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), gv(m + 1), y(m), hcol(2 * (m + 16));
    double bnorm, rn;
    SNS_TRY(norm2(h, b, &bnorm));
    const double tol = std::max(o.ksp_rtol * bnorm, o.ksp_atol);
    int its = 0, reason = 0;
    double* r = V;                         // V[0] doubles as the residual vector
    SNS_TRY(op_residual(h, x, b, r));
    SNS_TRY(norm2(h, r, &rn));
    if (o.monitor) std::printf("  0 KSP Residual norm %.12e\n", rn);
    while (!reason) {
        if (!(rn == rn) || std::isinf(rn)) { reason = SNS_KSP_DIVERGED_NANORINF; break; }
        if (rn <= tol) { reason = (rn <= o.ksp_atol) ? SNS_KSP_CONVERGED_ATOL : SNS_KSP_CONVERGED_RTOL; break; }
        if (its >= o.ksp_max_it) { reason = SNS_KSP_DIVERGED_ITS; break; }
        hipLaunchKernelGGL(k_scale_copy, dim3(g), dim3(256), 0, h->stream, nd, 1.0 / rn, r, V);
        std::fill(gv.begin(), gv.end(), 0.0);
        gv[0] = rn;
        int j = 0;
        double res = rn;
        for (; j < m && its < o.ksp_max_it; ++j) {
            double* vj = V + (size_t)j * ld;
            double* zj = Z + (size_t)j * ld;
            double* w = V + (size_t)(j + 1) * ld;
            SNS_TRY(pc_apply(h, vj, zj));
            SNS_TRY(op_apply(h, zj, w));
            const int nv = j + 1;
            // CGS2 with TWO global reductions per iteration: pass 1 dots; pass 2 dots + (w.w), the new
            // norm follows from ||w - V h2||^2 = w.w - |h2|^2 (V orthonormal).
            for (int pass = 0; pass < 2; ++pass) {
                double* dh = pass == 0 ? dh1 : dh2;
                for (int c0 = 0; c0 < nv; c0 += 8) {
                    const int cn = std::min(8, nv - c0);
                    hipLaunchKernelGGL(k_multi_dot8, dim3(g), dim3(256), 0, h->stream, nd, cn, V + (size_t)c0 * ld, ld,
                                       w, h->partial);
                    reduce_local(h, g, 8, dh + c0);
                }
                if (pass == 1) {
                    hipLaunchKernelGGL(k_dot2, dim3(g), dim3(256), 0, h->stream, nd, w, w, h->partial);
                    reduce_local(h, g, 2, dh + (m + 8));          // k_dot2 emits (x.y, y.y): both are w.w here
                    SNS_TRY(allreduce(h, dh, m + 10));
                } else {
                    SNS_TRY(allreduce(h, dh, nv));
                }
                for (int c0 = 0; c0 < nv; c0 += 8) {
                    const int cn = std::min(8, nv - c0);
                    hipLaunchKernelGGL(k_multi_axpy8, dim3(g), dim3(256), 0, h->stream, nd, cn, V + (size_t)c0 * ld,
                                       ld, dh + c0, -1.0, w, (double*)nullptr);
                }
            }
            // one device->host transfer per iteration: h1[0..nv), h2[0..nv), w.w
            SNS_TRY(fetch(h, h->d_h, 2 * S, hcol.data()));
            double* Hj = &H[(size_t)j * (m + 1)];             // column j
            double h2sq = 0.0;
            for (int k = 0; k < nv; ++k) {
                Hj[k] = hcol[k] + hcol[S + k];
                h2sq += hcol[S + k] * hcol[S + k];
            }
            const double ww = hcol[S + (m + 8)];
            double wn2 = ww - h2sq;
            double wn;
            if (!(wn2 > 1e-6 * ww)) SNS_TRY(norm2(h, w, &wn));   // heavy cancellation: measure it
            else wn = std::sqrt(wn2);
            Hj[nv] = wn;
            if (wn > 0.0) hipLaunchKernelGGL(k_scale_copy, dim3(g), dim3(256), 0, h->stream, nd, 1.0 / wn, w, w);
            for (int k = 0; k < j; ++k) {                      // previous rotations
                const double t0 = cs[k] * Hj[k] + sn[k] * Hj[k + 1];
                Hj[k + 1] = -sn[k] * Hj[k] + cs[k] * Hj[k + 1];
                Hj[k] = t0;
            }
            const double den = std::hypot(Hj[j], Hj[j + 1]);
            cs[j] = den > 0 ? Hj[j] / den : 1.0;
            sn[j] = den > 0 ? Hj[j + 1] / den : 0.0;
            Hj[j] = den;
            Hj[j + 1] = 0.0;
            gv[j + 1] = -sn[j] * gv[j];
            gv[j] = cs[j] * gv[j];
            res = std::fabs(gv[j + 1]);
            ++its;
            if (o.monitor) std::printf("%3d KSP Residual norm %.12e\n", its, res);
            if (res <= tol || wn == 0.0 || !(res == res)) { ++j; break; }
        }
        // y = H^-1 g ; x += Z y
        for (int k = j - 1; k >= 0; --k) {
            double sacc = gv[k];
            for (int q = k + 1; q < j; ++q) sacc -= H[(size_t)q * (m + 1) + k] * y[q];
            y[k] = sacc / H[(size_t)k * (m + 1) + k];
        }
        HIP_TRY(hipMemcpyAsync(dh1, y.data(), j * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));             // y is a stack-lifetime host buffer
        for (int c0 = 0; c0 < j; c0 += 8) {
            const int cn = std::min(8, j - c0);
            hipLaunchKernelGGL(k_multi_axpy8, dim3(g), dim3(256), 0, h->stream, nd, cn, Z + (size_t)c0 * ld, ld,
                               dh1 + c0, 1.0, x, (double*)nullptr);
        }
        SNS_TRY(op_residual(h, x, b, r));
        SNS_TRY(norm2(h, r, &rn));
    }
    *its_out = its;
    *reason_out = reason;
    *rnorm_out = rn;
    return SNS_OK;
}


int krylov(sns_ctx* h, const double* b, double* x, int* its, int* reason, double* rnorm) {
    if (!h->has_matrix) { set_error("krylov_solve before a matrix was assembled"); return SNS_E_STATE; }
    if (!h->pc_ready && h->opt.pc_type != SNS_PC_NONE) SNS_TRY(pc_setup(h));
    h->first_put_carried = h->child_put_carried = h->pc_then_op = false;
    h->put_pending = nullptr;
    h->first_sweep_done = false;                 // (a solve that ended in an error between setting and consuming it must not leak it)
    h->ctr_host_syncs = h->ctr_allreduce = h->ctr_exchange = 0;
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    // A solve that BREAKS DOWN (or produces NaN/Inf) under the AMG preconditioner is retried ONCE, from the same initial
    // guess, with every level's block-Jacobi damping scaled by 0.7 (opt.amg_retry_damping, default on): the damping
    // estimate (|lambda|max of Dinv A + a growth check on the dominant mode) is not a bound for a non-symmetric
    // operator, and at cell Reynolds numbers of 5-10 a slightly over-relaxed smoother is what breaks BiCGStab down
    // (measured: jittered 648 k-tet duct, Re 200: auto damping fails after 218 iterations, 0.7 x converges).  A solve
    // that merely runs out of iterations (DIVERGED_ITS) is NOT retried: like PETSc, the reason is reported and that is
    // it.  Because BiCGStab under an over-relaxed smoother more often STAGNATES than breaks down (the same 648 k-tet case,
    // round 3: it wanders between 0.2 and 70 x ||b|| for as long as it is allowed to), the first attempt also ends -- as a
    // breakdown -- when its best residual has not halved for amg_retry_stall_its (100) iterations.  The smaller damping is kept for the later Jacobians of the handle until sns_set_options is called; the
    // retry count and the current factor are visible through sns_get_counters.  *its is the sum over both attempts
    // (<= 2 ksp_max_it).  Not in the reference; converging solves never see it.
    const bool can_retry = h->opt.pc_type == SNS_PC_AMG && h->opt.amg_retry_damping != 0 && h->damping_backoff > 0.4;
    double* x0 = nullptr;
    if (can_retry) {
        SNS_TRY(get_vec(h, 14, &x0));
        HIP_TRY(hipMemcpyAsync(x0, x, nred_of(h) * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
    int its_total = 0;
    h->last_first_reason = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        int rc;
        if (h->opt.ksp_type == SNS_KSP_BICGSTAB)
            rc = bicgstab(h, b, x, its, reason, rnorm, (can_retry && attempt == 0) ? h->opt.amg_retry_stall_its : 0);
        else if (h->opt.ksp_type == SNS_KSP_FGMRES) rc = fgmres(h, b, x, its, reason, rnorm);
        else if (h->opt.ksp_type == SNS_KSP_TFQMR) rc = tfqmr(h, b, x, its, reason, rnorm);
        else { set_error("bad ksp_type"); return SNS_E_ARG; }
        SNS_TRY(rc);
        its_total += *its;
        const bool retryable = *reason == SNS_KSP_DIVERGED_BREAKDOWN || *reason == SNS_KSP_DIVERGED_NANORINF ||
                               *reason == SNS_KSP_STALLED;
        if (!retryable || !can_retry || attempt == 1) break;
        h->last_first_reason = *reason;
        ++h->ctr_retries;
        h->damping_backoff *= 0.7;
        if (h->opt.monitor)
            std::printf("  KSP failed (reason %d after %d iterations): retrying with the smoother damping scaled by %.2f\n",
                        *reason, *its, h->damping_backoff);
        for (auto& L : h->levels) {
            L.omega *= 0.7;
            if (L.omega_checked > 0.0) L.omega_checked *= 0.7;
        }
        HIP_TRY(hipMemcpyAsync(x, x0, nred_of(h) * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
    *its = its_total;
    h->last_ctr[0] = h->ctr_host_syncs; h->last_ctr[1] = h->ctr_allreduce; h->last_ctr[2] = h->ctr_exchange;
    SNS_TRY(halo_exchange(h, x));                          // leave the solution's ghost tail current
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->tm.krylov_ms += ms;
    h->tm.ksp_its += *its;
    time_collect(h);
    HIP_TRY(hipGetLastError());
    return SNS_OK;
}


}  // namespace sns
