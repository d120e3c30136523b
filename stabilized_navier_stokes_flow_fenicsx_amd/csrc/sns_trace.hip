// Batched particle tracing through a P1 velocity field on a tet mesh.
//
// Replaces NavierStokes/streamtrace.py's per-seed `solve_ivp(RK45, max_step=0.125, t in [0,20])`
// (:208-232 forward, :357-383 reverse; one Python call per seed, thread pool / MPI task farm) by ONE
// kernel launch: one lane per seed, scipy's RK45 (Dormand-Prince 5(4), same tableau, same step-size
// controller, same initial-step heuristic, rtol 1e-3 / atol 1e-6 defaults) with
//   * velocity = P1 interpolation in the containing tet, ZERO outside the mesh   (velfunc :144-158)
//   * point location by walking across faces from the previous tet (the reference queries a
//     bounding-box tree per evaluation)
//   * terminal events: speed < 1e-6 from above (:175-178) and the x-plane crossing, x > 3.7 upward
//     forward / x < 0.13 downward in reverse (:180-188); event points are located on the cubic
//     Hermite interpolant of the step (scipy uses the 4th-order dense output: differences ~1e-6).
// Divergent by nature (adaptive steps, walks); it is latency- not bandwidth-bound: the mesh and the
// field of a 10 M-tet run (0.6 GB) sit in the Infinity Cache.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <string>

#include "sns_internal.h"

namespace sns {

struct TraceMesh {
    const double* pts;      // n x 3
    const int32_t* tets;    // E x 4
    const int32_t* nbr;     // E x 4: tet across the face opposite local vertex a, -1 on the boundary
    const double* vel;      // n x 3 nodal velocity
};

// barycentric coordinates of x in tet t; returns the index of the most negative one (or -1 if inside)
__device__ __forceinline__ int bary(const TraceMesh& M, int32_t t, const double x[3], double lam[4]) {
    const int32_t* tv = M.tets + 4 * (int64_t)t;
    const double* p0 = M.pts + 3 * (int64_t)tv[0];
    const double* p1 = M.pts + 3 * (int64_t)tv[1];
    const double* p2 = M.pts + 3 * (int64_t)tv[2];
    const double* p3 = M.pts + 3 * (int64_t)tv[3];
    const double a0 = p1[0] - p0[0], a1 = p1[1] - p0[1], a2 = p1[2] - p0[2];
    const double b0 = p2[0] - p0[0], b1 = p2[1] - p0[1], b2 = p2[2] - p0[2];
    const double c0 = p3[0] - p0[0], c1 = p3[1] - p0[1], c2 = p3[2] - p0[2];
    const double r0 = x[0] - p0[0], r1 = x[1] - p0[1], r2 = x[2] - p0[2];
    const double bc0 = b1 * c2 - b2 * c1, bc1 = b2 * c0 - b0 * c2, bc2 = b0 * c1 - b1 * c0;
    const double det = a0 * bc0 + a1 * bc1 + a2 * bc2;
    const double id = 1.0 / det;
    lam[1] = (r0 * bc0 + r1 * bc1 + r2 * bc2) * id;
    const double ca0 = c1 * a2 - c2 * a1, ca1 = c2 * a0 - c0 * a2, ca2 = c0 * a1 - c1 * a0;
    lam[2] = (r0 * ca0 + r1 * ca1 + r2 * ca2) * id;
    const double ab0 = a1 * b2 - a2 * b1, ab1 = a2 * b0 - a0 * b2, ab2 = a0 * b1 - a1 * b0;
    lam[3] = (r0 * ab0 + r1 * ab1 + r2 * ab2) * id;
    lam[0] = 1.0 - lam[1] - lam[2] - lam[3];
    int worst = -1;
    double mn = -1e-12;
    for (int a = 0; a < 4; ++a)
        if (lam[a] < mn) { mn = lam[a]; worst = a; }
    return worst;
}

// velocity at x (times sgn); *tet is the walk's starting guess and is updated; outside -> zero (:149-153)
__device__ __forceinline__ bool vel_at(const TraceMesh& M, const double x[3], int32_t* tet, double sgn, double v[3]) {
    int32_t t = *tet;
    double lam[4];
    for (int it = 0; it < 512 && t >= 0; ++it) {
        const int w = bary(M, t, x, lam);
        if (w < 0) {
            const int32_t* tv = M.tets + 4 * (int64_t)t;
            v[0] = v[1] = v[2] = 0.0;
            for (int a = 0; a < 4; ++a) {
                const double* u = M.vel + 3 * (int64_t)tv[a];
                v[0] += lam[a] * u[0]; v[1] += lam[a] * u[1]; v[2] += lam[a] * u[2];
            }
            v[0] *= sgn; v[1] *= sgn; v[2] *= sgn;
            *tet = t;
            return true;
        }
        t = M.nbr[4 * (int64_t)t + w];
    }
    v[0] = v[1] = v[2] = 0.0;
    return false;
}

__device__ __forceinline__ double rms3(const double a[3]) {
    return sqrt((a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * (1.0 / 3.0));
}

// cubic Hermite interpolant of a step (y0,f0) -> (y1,f1) of length h at theta in [0,1]
__device__ __forceinline__ void hermite(const double y0[3], const double f0[3], const double y1[3], const double f1[3],
                                        double h, double th, double out[3]) {
    const double t2 = th * th, t3 = t2 * th;
    const double h00 = 2 * t3 - 3 * t2 + 1, h10 = t3 - 2 * t2 + th, h01 = -2 * t3 + 3 * t2, h11 = t3 - t2;
    for (int i = 0; i < 3; ++i) out[i] = h00 * y0[i] + h10 * h * f0[i] + h01 * y1[i] + h11 * h * f1[i];
}

struct TraceParams {
    double t_end, max_step, rtol, atol, x_stop, speed_min;
    int reverse;      // 0: forward, stop when x crosses x_stop upward; 1: velocity negated, stop when x crosses downward
};

__global__ __launch_bounds__(64) void k_streamtrace(int32_t n_seeds, TraceMesh M, TraceParams P,
                                                    const double* __restrict__ seeds,
                                                    const int32_t* __restrict__ seed_tet, double* __restrict__ out_pos,
                                                    double* __restrict__ out_t, int32_t* __restrict__ out_status,
                                                    int32_t* __restrict__ out_steps) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seeds) return;
    // Dormand-Prince 5(4) as in scipy.integrate.RK45
    const double C[6] = {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0};
    const double A[6][5] = {{0, 0, 0, 0, 0},
                            {1.0 / 5, 0, 0, 0, 0},
                            {3.0 / 40, 9.0 / 40, 0, 0, 0},
                            {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
                            {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
                            {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
    const double B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
    const double Ee[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};
    (void)C;
    const double sgn = P.reverse ? -1.0 : 1.0;
    double y[3] = {seeds[3 * (int64_t)i], seeds[3 * (int64_t)i + 1], seeds[3 * (int64_t)i + 2]};
    int32_t tet = seed_tet[i];
    double f[3];
    vel_at(M, y, &tet, sgn, f);
    double t = 0.0;
    int status = 0, nsteps = 0;      // 0: reached t_end, 1: speed event, 2: plane event, 3: step too small
    // ---- select_initial_step (order 4) ----
    double h_abs;
    {
        double sc[3], q0[3], q1[3];
        for (int k = 0; k < 3; ++k) { sc[k] = P.atol + fabs(y[k]) * P.rtol; q0[k] = y[k] / sc[k]; q1[k] = f[k] / sc[k]; }
        const double d0 = rms3(q0), d1 = rms3(q1);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        h0 = fmin(h0, P.t_end);
        double y1[3], f1[3];
        for (int k = 0; k < 3; ++k) y1[k] = y[k] + h0 * f[k];
        int32_t tt = tet;
        vel_at(M, y1, &tt, sgn, f1);
        double q2[3];
        for (int k = 0; k < 3; ++k) q2[k] = (f1[k] - f[k]) / sc[k];
        const double d2 = rms3(q2) / h0;
        const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : pow(0.01 / fmax(d1, d2), 0.2);
        h_abs = fmin(fmin(100 * h0, h1), fmin(P.t_end, P.max_step));
    }
    double speed_old = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    if (speed_old - P.speed_min <= 0.0 && false) status = 1;      // scipy only reports crossings, not the initial state
    while (status == 0 && t < P.t_end && nsteps < 100000) {
        const double min_step = 10.0 * (nextafter(t, INFINITY) - t);
        if (h_abs > P.max_step) h_abs = P.max_step;
        else if (h_abs < min_step) h_abs = min_step;
        bool accepted = false, rejected = false;
        double y_new[3], f_new[3], h = 0.0, t_new = t;
        int32_t tet_new = tet;
        while (!accepted) {
            if (h_abs < min_step) { status = 3; break; }
            h = h_abs;
            t_new = t + h;
            if (t_new - P.t_end > 0) t_new = P.t_end;
            h = t_new - t;
            h_abs = fabs(h);
            double K[7][3];
            for (int k = 0; k < 3; ++k) K[0][k] = f[k];
            int32_t tw = tet;
            for (int s = 1; s < 6; ++s) {
                double ys[3];
                for (int k = 0; k < 3; ++k) {
                    double dy = 0.0;
                    for (int j = 0; j < s; ++j) dy += A[s][j] * K[j][k];
                    ys[k] = y[k] + h * dy;
                }
                vel_at(M, ys, &tw, sgn, K[s]);
            }
            for (int k = 0; k < 3; ++k) {
                double dy = 0.0;
                for (int j = 0; j < 6; ++j) dy += B[j] * K[j][k];
                y_new[k] = y[k] + h * dy;
            }
            tet_new = tw;
            vel_at(M, y_new, &tet_new, sgn, f_new);
            for (int k = 0; k < 3; ++k) K[6][k] = f_new[k];
            double en = 0.0;
            for (int k = 0; k < 3; ++k) {
                double e = 0.0;
                for (int j = 0; j < 7; ++j) e += Ee[j] * K[j][k];
                const double scale = P.atol + fmax(fabs(y[k]), fabs(y_new[k])) * P.rtol;
                const double q = e * h / scale;
                en += q * q;
            }
            en = sqrt(en * (1.0 / 3.0));
            if (en < 1.0) {
                double factor = (en == 0.0) ? 10.0 : fmin(10.0, 0.9 * pow(en, -0.2));
                if (rejected) factor = fmin(1.0, factor);
                h_abs *= factor;
                accepted = true;
            } else {
                h_abs *= fmax(0.2, 0.9 * pow(en, -0.2));
                rejected = true;
            }
        }
        if (status) break;
        ++nsteps;
        // ---- terminal events on [t, t_new] ----
        const double speed_new = sqrt(f_new[0] * f_new[0] + f_new[1] * f_new[1] + f_new[2] * f_new[2]);
        const double gp_old = y[0] - P.x_stop, gp_new = y_new[0] - P.x_stop;
        const bool ev_plane = P.reverse ? (gp_old >= 0.0 && gp_new <= 0.0) : (gp_old <= 0.0 && gp_new >= 0.0);
        const bool ev_speed = (speed_old - P.speed_min >= 0.0) && (speed_new - P.speed_min <= 0.0);
        double th_plane = 2.0, th_speed = 2.0;
        if (ev_plane) {                        // bisection on the Hermite interpolant of x
            double lo = 0.0, hi = 1.0, yy[3];
            for (int it = 0; it < 60; ++it) {
                const double mid = 0.5 * (lo + hi);
                hermite(y, f, y_new, f_new, h, mid, yy);
                const double g = yy[0] - P.x_stop;
                const bool before = P.reverse ? (g > 0.0) : (g < 0.0);
                if (before) lo = mid; else hi = mid;
            }
            th_plane = 0.5 * (lo + hi);
        }
        if (ev_speed) {                        // bisection on speed(sol(theta)) - speed_min
            double lo = 0.0, hi = 1.0, yy[3], vv[3];
            int32_t tw = tet;
            for (int it = 0; it < 40; ++it) {
                const double mid = 0.5 * (lo + hi);
                hermite(y, f, y_new, f_new, h, mid, yy);
                vel_at(M, yy, &tw, sgn, vv);
                tw = tw < 0 ? tet : tw;
                const double sp = sqrt(vv[0] * vv[0] + vv[1] * vv[1] + vv[2] * vv[2]);
                if (sp - P.speed_min > 0.0) lo = mid; else hi = mid;
            }
            th_speed = 0.5 * (lo + hi);
        }
        if (ev_plane || ev_speed) {
            const double th = fmin(th_plane, th_speed);
            double yy[3];
            hermite(y, f, y_new, f_new, h, th, yy);
            for (int k = 0; k < 3; ++k) y[k] = yy[k];
            t = t + th * h;
            status = (th_speed <= th_plane) ? 1 : 2;
            break;
        }
        for (int k = 0; k < 3; ++k) { y[k] = y_new[k]; f[k] = f_new[k]; }
        t = t_new;
        tet = tet_new < 0 ? tet : tet_new;
        speed_old = speed_new;
    }
    out_pos[3 * (int64_t)i] = y[0];
    out_pos[3 * (int64_t)i + 1] = y[1];
    out_pos[3 * (int64_t)i + 2] = y[2];
    out_t[i] = t;
    out_status[i] = status;
    out_steps[i] = nsteps;
}

}  // namespace sns

extern "C" int sns_streamtrace(int32_t n_nodes, int64_t n_tets, const double* pts_dev, const int32_t* tets_dev,
                               const int32_t* nbr_dev, const double* vel_dev, int32_t n_seeds,
                               const double* seeds_dev, const int32_t* seed_tet_dev, int reverse, double t_end,
                               double max_step, double rtol, double atol, double x_stop, double speed_min,
                               double* pos_out_dev, double* t_out_dev, int32_t* status_out_dev,
                               int32_t* steps_out_dev, void* hip_stream) {
    if (n_nodes <= 0 || n_tets <= 0 || !pts_dev || !tets_dev || !nbr_dev || !vel_dev || n_seeds < 0 || !seeds_dev ||
        !seed_tet_dev || !pos_out_dev || !t_out_dev || !status_out_dev || !steps_out_dev || !(t_end > 0) ||
        !(max_step > 0)) {
        sns::set_error("sns_streamtrace: bad arguments");
        return SNS_E_ARG;
    }
    if (n_seeds == 0) return SNS_OK;
    sns::TraceMesh M{pts_dev, tets_dev, nbr_dev, vel_dev};
    sns::TraceParams P{t_end, max_step, rtol, atol, x_stop, speed_min, reverse ? 1 : 0};
    hipStream_t s = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(sns::k_streamtrace, dim3((n_seeds + 63) / 64), dim3(64), 0, s, n_seeds, M, P, seeds_dev,
                       seed_tet_dev, pos_out_dev, t_out_dev, status_out_dev, steps_out_dev);
    hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        sns::set_error(std::string("sns_streamtrace: ") + hipGetErrorString(e));
        return SNS_E_HIP;
    }
    return SNS_OK;
}
