/*
 * ORACLE (test infrastructure, NOT the product): plain-C restatement of one global-alignment
 * iteration of Align3R's PointCloudOptimizer -- loss, analytic gradients and the Adam update.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Follows (all under /root/reference/):
 *   PointCloudOptimizer.forward            dust3r/cloud_opt/optimizer.py:223-241
 *   get_pw_poses / _get_poses / get_pw_scale /
 *   get_pw_norm_scale_factor / get_adaptors dust3r/cloud_opt/base_opt.py:177-229
 *   get_depthmaps / depth_to_pts3d /
 *   _fast_depthmap_to_pts3d                 dust3r/cloud_opt/optimizer.py:174-200,244-251
 *   geotrf                                  dust3r/utils/geometry.py:40-101
 *   l1_dist / l2_dist / signed_expm1        dust3r/cloud_opt/commons.py:102-120
 *   global_alignment_iter (Adam betas .9,.9) dust3r/cloud_opt/base_opt.py:424-464
 * The unit-quaternion (XYZW) -> rotation closed form stands in for roma.RigidUnitQuat
 * (base_opt.py:188; roma is an unpinned third-party dependency that is not installed here).
 * The gradients are derived by hand (the reference uses autograd); tests/test_oracle_align.py
 * pins loss, gradients and 1/5/50-step trajectories against goldens captured from the
 * reference's own code (tests/golden/align.npz).
 *
 * Arithmetic: element-wise math in float (as the reference), sums accumulated in double.
 * Parallelisation: OpenMP over pixel chunks, image-major (each pixel's depth gradient is
 * complete inside one thread), per-thread accumulators for the small pose parameters.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int E, N, P;
    int use_mono;       /* depth = mono*exp(scalemap)+shift instead of exp(log-depth) */
    int norm_pw_scale;  /* base_opt.py:212-218 */
    int dist_l2;        /* 0: l1 (Euclidean norm), 1: l2 (squared) */
    int train_poses, train_focals, train_pp;
    float base_scale, pw_break, focal_break;
    double total_area_i, total_area_j;
} a3r_oracle_align_cfg;

static void quat_to_R(const float* q, float* R, float* qn_out, float* nrm_out) {
    float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    float x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
    float tx = 2 * x, ty = 2 * y, tz = 2 * z;
    float twx = tx * w, twy = ty * w, twz = tz * w;
    float txx = tx * x, txy = ty * x, txz = tz * x;
    float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
    if (qn_out) { qn_out[0] = x; qn_out[1] = y; qn_out[2] = z; qn_out[3] = w; }
    if (nrm_out) *nrm_out = n;
}

/* dL/dq (un-normalised quaternion) from G = dL/dR, through R(q/|q|) */
static void quat_backward(const float* qn, float nrm, const double* G, double* gq) {
    double x = qn[0], y = qn[1], z = qn[2], w = qn[3];
    double gx = 2 * (y * G[1] + z * G[2] + y * G[3] - 2 * x * G[4] - w * G[5] + z * G[6] + w * G[7] - 2 * x * G[8]);
    double gy = 2 * (-2 * y * G[0] + x * G[1] + w * G[2] + x * G[3] + z * G[5] - w * G[6] + z * G[7] - 2 * y * G[8]);
    double gz = 2 * (-2 * z * G[0] - w * G[1] + x * G[2] + w * G[3] - 2 * z * G[4] + y * G[5] + x * G[6] + y * G[7]);
    double gw = 2 * (-z * G[1] + y * G[2] + z * G[3] - x * G[5] - y * G[6] + x * G[7]);
    double dot = gx * x + gy * y + gz * z + gw * w;
    gq[0] = (gx - dot * x) / nrm; gq[1] = (gy - dot * y) / nrm;
    gq[2] = (gz - dot * z) / nrm; gq[3] = (gw - dot * w) / nrm;
}

static float signed_expm1f(float x) {
    float s = (x > 0) - (x < 0);
    return s * expm1f(fabsf(x));
}
static double signed_expm1_grad(float x) { return x == 0.f ? 0.0 : (double)expf(fabsf(x)); }

/* Builds the per-edge 3x4 transforms [s*R*diag(a) | s*T] and per-image [R | t], focal, pp. */
static void build_transforms(const a3r_oracle_align_cfg* c, const float* pw_poses, const float* pw_adaptors,
                             const float* im_poses, const float* im_focals, const float* im_pp, const float* pp0,
                             float* edge_M /*[E][12]*/, float* edge_s /*[E]*/, float* edge_a /*[E][3]*/,
                             float* img_R /*[N][12]*/, float* img_f /*[N]*/, float* img_pp /*[N][2]*/) {
    double mean_ls = 0;
    for (int e = 0; e < c->E; e++) mean_ls += pw_poses[e * 8 + 7];
    mean_ls /= c->E;
    float nf = c->norm_pw_scale ? expf(logf(c->base_scale) - (float)mean_ls) : 1.f;
    for (int e = 0; e < c->E; e++) {
        const float* p = pw_poses + e * 8;
        float R[9];
        quat_to_R(p, R, NULL, NULL);
        float s = expf(p[7]) * nf;
        float ad[3] = {pw_adaptors[e * 2], pw_adaptors[e * 2], pw_adaptors[e * 2 + 1]};
        if (c->norm_pw_scale) {
            float m = (ad[0] + ad[1] + ad[2]) / 3.f;
            ad[0] -= m; ad[1] -= m; ad[2] -= m;
        }
        for (int k = 0; k < 3; k++) edge_a[e * 3 + k] = expf(ad[k] / c->pw_break);
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 3; k++) edge_M[e * 12 + r * 4 + k] = s * R[r * 3 + k] * edge_a[e * 3 + k];
            edge_M[e * 12 + r * 4 + 3] = s * signed_expm1f(p[4 + r]);
        }
        edge_s[e] = s;
    }
    for (int n = 0; n < c->N; n++) {
        const float* p = im_poses + n * 7;
        float R[9];
        quat_to_R(p, R, NULL, NULL);
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 3; k++) img_R[n * 12 + r * 4 + k] = R[r * 3 + k];
            img_R[n * 12 + r * 4 + 3] = signed_expm1f(p[4 + r]);
        }
        img_f[n] = expf(im_focals[n] / c->focal_break);
        img_pp[n * 2 + 0] = pp0[n * 2 + 0] + 10.f * im_pp[n * 2 + 0];
        img_pp[n * 2 + 1] = pp0[n * 2 + 1] + 10.f * im_pp[n * 2 + 1];
    }
}

/*
 * Loss + gradients.  Inputs are the stacked buffers of optimizer.py:55-71:
 *   ei, ej [E]; pred_i, pred_j [E,P,3]; w_i, w_j [E,P] (= conf_trf(conf), 0 on padding);
 *   imw [N] image widths, imarea [N] = h*w (pixels >= imarea are padding with grid = 0);
 *   mono [N,P] (only if use_mono); pp0 [N,2] = (w/2, h/2).
 * Parameters: pw_poses [E,8], pw_adaptors [E,2] (frozen), depth [N,P] (log-depth or scalemap),
 *   shifts [N], im_poses [N,7], im_focals [N], im_pp [N,2].
 * Outputs: *loss, gradients of the same shapes (g_* may be NULL where not trained).
 */
int a3r_oracle_align_loss_grad(const a3r_oracle_align_cfg* c, const int* ei, const int* ej, const int* imw,
                               const int* imarea, const float* pred_i, const float* pred_j, const float* w_i,
                               const float* w_j, const float* mono, const float* pp0, const float* pw_poses,
                               const float* pw_adaptors, const float* depth, const float* shifts,
                               const float* im_poses, const float* im_focals, const float* im_pp, double* loss,
                               float* g_pw_poses, float* g_depth, float* g_shifts, float* g_im_poses,
                               float* g_im_focals, float* g_im_pp) {
    const int E = c->E, N = c->N, P = c->P;
    float* edge_M = malloc(sizeof(float) * E * 12);
    float* edge_s = malloc(sizeof(float) * E);
    float* edge_a = malloc(sizeof(float) * E * 3);
    float* img_R = malloc(sizeof(float) * N * 12);
    float* img_f = malloc(sizeof(float) * N);
    float* img_pp = malloc(sizeof(float) * N * 2);
    build_transforms(c, pw_poses, pw_adaptors, im_poses, im_focals, im_pp, pp0, edge_M, edge_s, edge_a, img_R,
                     img_f, img_pp);
    /* incidence lists: image n -> (edge, side) */
    int* deg = calloc(N + 1, sizeof(int));
    for (int e = 0; e < E; e++) { deg[ei[e] + 1]++; deg[ej[e] + 1]++; }
    for (int n = 0; n < N; n++) deg[n + 1] += deg[n];
    int* inc = malloc(sizeof(int) * 2 * E);
    int* fill = calloc(N, sizeof(int));
    for (int e = 0; e < E; e++) {
        inc[deg[ei[e]] + fill[ei[e]]++] = e * 2 + 0;
        inc[deg[ej[e]] + fill[ej[e]]++] = e * 2 + 1;
    }
    const double inv_ai = 1.0 / c->total_area_i, inv_aj = 1.0 / c->total_area_j;
    /* global accumulators: per edge A[9], b[3]; per image G[9], bt[3], gf, gpp[2], gshift; loss */
    double* accE = calloc((size_t)E * 12, sizeof(double));
    double* accN = calloc((size_t)N * 16, sizeof(double));
    double total = 0;
    const int CH = 512;
    const int nchunks = (P + CH - 1) / CH;
#pragma omp parallel
    {
        double* lE = calloc((size_t)E * 12, sizeof(double));
        double* lN = calloc((size_t)N * 16, sizeof(double));
        double ltot = 0;
#pragma omp for schedule(dynamic, 4)
        for (int ch = 0; ch < nchunks; ch++) {
            const int p0 = ch * CH, p1 = p0 + CH < P ? p0 + CH : P;
            for (int n = 0; n < N; n++) {
                const float* Rn = img_R + n * 12;
                const float f = img_f[n], ppx = img_pp[n * 2], ppy = img_pp[n * 2 + 1];
                double* aN = lN + n * 16;
                for (int p = p0; p < p1; p++) {
                    float gxy[2] = {0.f, 0.f};
                    if (p < imarea[n]) { gxy[0] = (float)(p % imw[n]); gxy[1] = (float)(p / imw[n]); }
                    float raw = depth[(size_t)n * P + p], d, dd_dparam;
                    if (c->use_mono) {
                        float es = expf(raw), m = mono[(size_t)n * P + p];
                        d = m * es + shifts[n];
                        dd_dparam = m * es;
                    } else {
                        d = expf(raw);
                        dd_dparam = d;
                    }
                    float rel[3] = {d * (gxy[0] - ppx) / f, d * (gxy[1] - ppy) / f, d};
                    float proj[3];
                    for (int r = 0; r < 3; r++)
                        proj[r] = Rn[r * 4] * rel[0] + Rn[r * 4 + 1] * rel[1] + Rn[r * 4 + 2] * rel[2] + Rn[r * 4 + 3];
                    double gp[3] = {0, 0, 0};
                    for (int k = deg[n]; k < deg[n + 1]; k++) {
                        const int e = inc[k] >> 1, side = inc[k] & 1;
                        const float* X = (side ? pred_j : pred_i) + ((size_t)e * P + p) * 3;
                        const float w = (side ? w_j : w_i)[(size_t)e * P + p];
                        const float* M = edge_M + e * 12;
                        float r3[3];
                        for (int r = 0; r < 3; r++)
                            r3[r] = proj[r] - (M[r * 4] * X[0] + M[r * 4 + 1] * X[1] + M[r * 4 + 2] * X[2] + M[r * 4 + 3]);
                        const double inva = side ? inv_aj : inv_ai;
                        double g[3];
                        if (c->dist_l2) {
                            float sq = r3[0] * r3[0] + r3[1] * r3[1] + r3[2] * r3[2];
                            ltot += (double)(sq * w) * inva;
                            for (int r = 0; r < 3; r++) g[r] = 2.0 * w * r3[r] * inva;
                        } else {
                            float rho = sqrtf(r3[0] * r3[0] + r3[1] * r3[1] + r3[2] * r3[2]);
                            ltot += (double)(rho * w) * inva;
                            double cf = rho > 0.f ? (double)w / rho * inva : 0.0;
                            for (int r = 0; r < 3; r++) g[r] = cf * r3[r];
                        }
                        double* aE = lE + (size_t)e * 12;
                        for (int r = 0; r < 3; r++) {
                            gp[r] += g[r];
                            aE[9 + r] += g[r];
                            for (int q = 0; q < 3; q++) aE[r * 3 + q] += g[r] * X[q];
                        }
                    }
                    /* per-image accumulators and the per-pixel depth gradient */
                    double h[3];
                    for (int q = 0; q < 3; q++) h[q] = Rn[q] * gp[0] + Rn[4 + q] * gp[1] + Rn[8 + q] * gp[2];
                    for (int r = 0; r < 3; r++) {
                        aN[9 + r] += gp[r];
                        for (int q = 0; q < 3; q++) aN[r * 3 + q] += gp[r] * rel[q];
                    }
                    double gd = h[0] * (gxy[0] - ppx) / f + h[1] * (gxy[1] - ppy) / f + h[2];
                    aN[12] += -(h[0] * rel[0] + h[1] * rel[1]) / c->focal_break;   /* d/d(im_focals) */
                    aN[13] += -h[0] * d / f * 10.0;                                /* d/d(im_pp.x) */
                    aN[14] += -h[1] * d / f * 10.0;
                    aN[15] += gd;                                                  /* d/d(shift) */
                    if (g_depth) g_depth[(size_t)n * P + p] = (float)(gd * dd_dparam);
                }
            }
        }
#pragma omp critical
        {
            for (size_t i = 0; i < (size_t)E * 12; i++) accE[i] += lE[i];
            for (size_t i = 0; i < (size_t)N * 16; i++) accN[i] += lN[i];
            total += ltot;
        }
        free(lE); free(lN);
    }
    *loss = total;
    /* small-parameter gradients */
    if (g_pw_poses) {
        double sumSs = 0;
        double* Ss = malloc(sizeof(double) * E);
        for (int e = 0; e < E; e++) {
            const float* p = pw_poses + e * 8;
            const double* A = accE + (size_t)e * 12; const double* b = A + 9;
            float R[9], qn[4], nrm;
            quat_to_R(p, R, qn, &nrm);
            const double s = edge_s[e];
            const float* a = edge_a + e * 3;
            double G[9], dLds = 0;
            for (int r = 0; r < 3; r++) {
                for (int q = 0; q < 3; q++) {
                    G[r * 3 + q] = -s * a[q] * A[r * 3 + q];
                    dLds -= (double)R[r * 3 + q] * a[q] * A[r * 3 + q];
                }
                dLds -= (double)signed_expm1f(p[4 + r]) * b[r];
            }
            double gq[4];
            quat_backward(qn, nrm, G, gq);
            for (int k = 0; k < 4; k++) g_pw_poses[e * 8 + k] = (float)gq[k];
            for (int k = 0; k < 3; k++) g_pw_poses[e * 8 + 4 + k] = (float)(-s * b[k] * signed_expm1_grad(p[4 + k]));
            Ss[e] = dLds * s;
            sumSs += Ss[e];
        }
        for (int e = 0; e < E; e++)
            g_pw_poses[e * 8 + 7] = (float)(c->norm_pw_scale ? Ss[e] - sumSs / E : Ss[e]);
        free(Ss);
    }
    for (int n = 0; n < N; n++) {
        const double* A = accN + (size_t)n * 16;
        if (g_im_poses) {
            const float* p = im_poses + n * 7;
            float R[9], qn[4], nrm;
            quat_to_R(p, R, qn, &nrm);
            double gq[4];
            quat_backward(qn, nrm, A, gq);
            for (int k = 0; k < 4; k++) g_im_poses[n * 7 + k] = (float)gq[k];
            for (int k = 0; k < 3; k++) g_im_poses[n * 7 + 4 + k] = (float)(A[9 + k] * signed_expm1_grad(p[4 + k]));
        }
        if (g_im_focals) g_im_focals[n] = (float)A[12];
        if (g_im_pp) { g_im_pp[n * 2] = (float)A[13]; g_im_pp[n * 2 + 1] = (float)A[14]; }
        if (g_shifts) g_shifts[n] = (float)A[15];
    }
    free(edge_M); free(edge_s); free(edge_a); free(img_R); free(img_f); free(img_pp);
    free(deg); free(inc); free(fill); free(accE); free(accN);
    return 0;
}

/* torch.optim.Adam (no amsgrad / weight decay), single tensor: torch/optim/adam.py _single_tensor_adam.
 * step = 1-based step count after increment. */
void a3r_oracle_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                     int step) {
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
#pragma omp parallel for
    for (long i = 0; i < n; i++) {
        float gi = g[i];
        m[i] = m[i] + (gi - m[i]) * (1.f - b1);          /* exp_avg.lerp_(grad, 1-beta1) */
        v[i] = v[i] * b2 + (1.f - b2) * gi * gi;          /* mul_(beta2).addcmul_(g, g, 1-beta2) */
        float denom = sqrtf(v[i]) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (m[i] / denom);
    }
}

/* Forward-only helpers used to pin the pose parameterisation against the goldens. */
void a3r_oracle_pose_matrices(const a3r_oracle_align_cfg* c, const float* pw_poses, const float* pw_adaptors,
                              const float* im_poses, const float* im_focals, const float* im_pp, const float* pp0,
                              float* edge_M, float* img_R, float* img_f, float* img_pp) {
    float* edge_s = malloc(sizeof(float) * c->E);
    float* edge_a = malloc(sizeof(float) * c->E * 3);
    build_transforms(c, pw_poses, pw_adaptors, im_poses, im_focals, im_pp, pp0, edge_M, edge_s, edge_a, img_R, img_f,
                     img_pp);
    free(edge_s); free(edge_a);
}

int a3r_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* =================================================================================================
 * cloud_opt_flow extras (dust3r/cloud_opt_flow/optimizer.py:500-572)
 * ================================================================================================= */

/* relative_pose_loss (optimizer.py:559-572) summed over consecutive image poses:
 *   sum_n ||R_n^T R_{n+1} - I||_F + tw * ||R_n^T (t_{n+1} - t_n)||,  RT = _get_poses(im_poses)
 * Returns the (unweighted) loss and ADDS weight * gradient to g_im_poses [N,7]. */
double a3r_oracle_temporal_loss_grad(int N, const float* im_poses, float translation_weight, float weight,
                                     float* g_im_poses) {
    double total = 0;
    double* GR = calloc((size_t)N * 9, sizeof(double));
    double* Gt = calloc((size_t)N * 3, sizeof(double));
    float* R = malloc(sizeof(float) * N * 9);
    float* T = malloc(sizeof(float) * N * 3);
    float* qn = malloc(sizeof(float) * N * 4);
    float* nr = malloc(sizeof(float) * N);
    for (int n = 0; n < N; n++) {
        quat_to_R(im_poses + n * 7, R + n * 9, qn + n * 4, nr + n);
        for (int k = 0; k < 3; k++) T[n * 3 + k] = signed_expm1f(im_poses[n * 7 + 4 + k]);
    }
    for (int n = 0; n + 1 < N; n++) {
        const float *Ra = R + n * 9, *Rb = R + (n + 1) * 9;
        double M[9], a = 0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += (double)Ra[k * 3 + i] * Rb[k * 3 + j];   /* (Ra^T Rb)_ij */
                M[i * 3 + j] = s - (i == j);
                a += M[i * 3 + j] * M[i * 3 + j];
            }
        a = sqrt(a);
        double d[3], u[3], un = 0;
        for (int k = 0; k < 3; k++) d[k] = (double)T[(n + 1) * 3 + k] - T[n * 3 + k];
        for (int i = 0; i < 3; i++) {
            u[i] = Ra[0 * 3 + i] * d[0] + Ra[1 * 3 + i] * d[1] + Ra[2 * 3 + i] * d[2];
            un += u[i] * u[i];
        }
        un = sqrt(un);
        total += a + translation_weight * un;
        if (a > 0) {      /* dL/dRa = Rb G^T, dL/dRb = Ra G with G = M / a */
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    double sa = 0, sb = 0;
                    for (int k = 0; k < 3; k++) {
                        sa += (double)Rb[i * 3 + k] * M[j * 3 + k] / a;
                        sb += (double)Ra[i * 3 + k] * M[k * 3 + j] / a;
                    }
                    GR[n * 9 + i * 3 + j] += sa;
                    GR[(n + 1) * 9 + i * 3 + j] += sb;
                }
        }
        if (un > 0) {
            double gu[3], Rg[3];
            for (int i = 0; i < 3; i++) gu[i] = translation_weight * u[i] / un;
            for (int i = 0; i < 3; i++) Rg[i] = Ra[i * 3 + 0] * gu[0] + Ra[i * 3 + 1] * gu[1] + Ra[i * 3 + 2] * gu[2];
            for (int i = 0; i < 3; i++) {
                Gt[(n + 1) * 3 + i] += Rg[i];
                Gt[n * 3 + i] -= Rg[i];
                for (int j = 0; j < 3; j++) GR[n * 9 + i * 3 + j] += d[i] * gu[j];   /* u = Ra^T d */
            }
        }
    }
    for (int n = 0; n < N; n++) {
        double gq[4];
        quat_backward(qn + n * 4, nr[n], GR + n * 9, gq);
        for (int k = 0; k < 4; k++) g_im_poses[n * 7 + k] += (float)(weight * gq[k]);
        for (int k = 0; k < 3; k++)
            g_im_poses[n * 7 + 4 + k] += (float)(weight * Gt[n * 3 + k] * signed_expm1_grad(im_poses[n * 7 + 4 + k]));
    }
    free(GR); free(Gt); free(R); free(T); free(qn); free(nr);
    return total;
}

/* Ego-flow loss (optimizer.py:521-541; DepthBasedWarping/warp_by_disp goem_opt.py:195-236; smooth_L1_loss_fn :18-24).
 * Direction 0: source ei -> target ej compared with flow_ij, direction 1: source ej -> target ei with flow_ji.
 *   disp = 1/(depth+1e-6); X = r/disp with r = K_s^-1 (x,y,1); Y = R_t^T (R_s X + T_s - T_t);
 *   n = disp*K_t*Y normalised by (z + 1e-6); flow = n - (x,y); smooth-L1(beta=1) on flow*m vs gt*m with m = ~dynamic,
 *   per element mask (loss < pxl_thre) * m; loss_dir = sum(loss*ppm)/sum(ppm).
 * Pass 1 (scale == NULL): only sums[4] = {S_0, C_0, S_1, C_1}.
 * Pass 2 (scale = {c_0, c_1}): ADDS c_dir * d(sum loss*ppm) to the gradient arrays (depth = log-depth parameter).
 * focals [N] / pp [N,2] are the per-image VALUES (f, cx, cy); g_f [N] is d/df (value), g_ppv [N,2] d/d(cx,cy). */
int a3r_oracle_flow_loss_grad(int E, int N, int H, int W, const int* ei, const int* ej, const float* flow_ij,
                              const float* flow_ji, const unsigned char* dyn, const float* depth_param,
                              const float* im_poses, const float* focals, const float* pp, float pxl_thre,
                              const double* scale, double* sums, float* g_depth, float* g_im_poses, float* g_f,
                              float* g_ppv) {
    const int P = H * W;
    float* R = malloc(sizeof(float) * N * 9);
    float* T = malloc(sizeof(float) * N * 3);
    float* qn = malloc(sizeof(float) * N * 4);
    float* nr = malloc(sizeof(float) * N);
    for (int n = 0; n < N; n++) {
        quat_to_R(im_poses + n * 7, R + n * 9, qn + n * 4, nr + n);
        for (int k = 0; k < 3; k++) T[n * 3 + k] = signed_expm1f(im_poses[n * 7 + 4 + k]);
    }
    double* GR = calloc((size_t)N * 9, sizeof(double));
    double* GT = calloc((size_t)N * 3, sizeof(double));
    double* Gf = calloc(N, sizeof(double));
    double* Gpp = calloc((size_t)N * 2, sizeof(double));
    double* Gd = scale ? calloc((size_t)N * P, sizeof(double)) : NULL;
    sums[0] = sums[1] = sums[2] = sums[3] = 0;
    for (int e = 0; e < E; e++)
        for (int dir = 0; dir < 2; dir++) {
            const int s = dir ? ej[e] : ei[e], t = dir ? ei[e] : ej[e];
            const float* fl = (dir ? flow_ji : flow_ij) + (size_t)e * 2 * P;
            const float *Rs = R + s * 9, *Rt = R + t * 9, *Ts = T + s * 3, *Tt = T + t * 3;
            const double fs = focals[s], ft = focals[t], cxs = pp[s * 2], cys = pp[s * 2 + 1], cxt = pp[t * 2], cyt = pp[t * 2 + 1];
            const double c = scale ? scale[dir] : 0.0;
            double S = 0, C = 0;
            for (int p = 0; p < P; p++) {
                if (dyn[(size_t)s * P + p]) continue;                    /* mask = ~dynamic_mask of the source image */
                const double x = p % W, y = p / W;
                const double dep = exp((double)depth_param[(size_t)s * P + p]);
                const double dp = dep + 1e-6;                            /* 1/disp */
                const double r[3] = {(x - cxs) / fs, (y - cys) / fs, 1.0};
                double Xs[3] = {dp * r[0], dp * r[1], dp}, Pw[3], v[3], Y[3];
                for (int i = 0; i < 3; i++) Pw[i] = Rs[i * 3] * Xs[0] + Rs[i * 3 + 1] * Xs[1] + Rs[i * 3 + 2] * Xs[2] + Ts[i];
                for (int i = 0; i < 3; i++) v[i] = Pw[i] - Tt[i];
                for (int i = 0; i < 3; i++) Y[i] = Rt[0 * 3 + i] * v[0] + Rt[1 * 3 + i] * v[1] + Rt[2 * 3 + i] * v[2];
                const double qx = ft * Y[0] + cxt * Y[2], qy = ft * Y[1] + cyt * Y[2], qz = Y[2] + 1e-6 * dp;
                const double est[2] = {qx / qz - x, qy / qz - y};
                double gn[2] = {0, 0};
                for (int k = 0; k < 2; k++) {
                    const double dlt = est[k] - fl[(size_t)k * P + p], ad = fabs(dlt);
                    const double l = ad < 1.0 ? 0.5 * dlt * dlt : ad - 0.5;
                    if (l < pxl_thre) {
                        S += l; C += 1;
                        gn[k] = ad < 1.0 ? dlt : (dlt > 0 ? 1.0 : -1.0);
                    }
                }
                if (!scale || c == 0.0) continue;
                const double gq[3] = {c * gn[0] / qz, c * gn[1] / qz, -c * (gn[0] * qx + gn[1] * qy) / (qz * qz)};
                const double gY[3] = {ft * gq[0], ft * gq[1], cxt * gq[0] + cyt * gq[1] + gq[2]};
                Gf[t] += gq[0] * Y[0] + gq[1] * Y[1];
                Gpp[t * 2] += gq[0] * Y[2];
                Gpp[t * 2 + 1] += gq[1] * Y[2];
                double gPw[3];
                for (int i = 0; i < 3; i++) gPw[i] = Rt[i * 3] * gY[0] + Rt[i * 3 + 1] * gY[1] + Rt[i * 3 + 2] * gY[2];
                for (int i = 0; i < 3; i++) {
                    GT[t * 3 + i] -= gPw[i];
                    GT[s * 3 + i] += gPw[i];
                    for (int j = 0; j < 3; j++) {
                        GR[t * 9 + i * 3 + j] += v[i] * gY[j];            /* Y = Rt^T v */
                        GR[s * 9 + i * 3 + j] += gPw[i] * Xs[j];          /* Pw = Rs Xs + Ts */
                    }
                }
                double h[3];
                for (int j = 0; j < 3; j++) h[j] = Rs[0 * 3 + j] * gPw[0] + Rs[1 * 3 + j] * gPw[1] + Rs[2 * 3 + j] * gPw[2];
                const double gdp = h[0] * r[0] + h[1] * r[1] + h[2] + 1e-6 * gq[2];
                Gd[(size_t)s * P + p] += gdp * dep;                      /* d/d(log depth) */
                Gf[s] += -(h[0] * Xs[0] + h[1] * Xs[1]) / fs;
                Gpp[s * 2] += -h[0] * dp / fs;
                Gpp[s * 2 + 1] += -h[1] * dp / fs;
            }
            sums[dir * 2] += S;
            sums[dir * 2 + 1] += C;
        }
    if (scale) {
        for (size_t i = 0; i < (size_t)N * P; i++) g_depth[i] += (float)Gd[i];
        for (int n = 0; n < N; n++) {
            double gq[4];
            quat_backward(qn + n * 4, nr[n], GR + n * 9, gq);
            for (int k = 0; k < 4; k++) g_im_poses[n * 7 + k] += (float)gq[k];
            for (int k = 0; k < 3; k++) g_im_poses[n * 7 + 4 + k] += (float)(GT[n * 3 + k] * signed_expm1_grad(im_poses[n * 7 + 4 + k]));
            g_f[n] += (float)Gf[n];
            g_ppv[n * 2] += (float)Gpp[n * 2];
            g_ppv[n * 2 + 1] += (float)Gpp[n * 2 + 1];
        }
        free(Gd);
    }
    free(R); free(T); free(qn); free(nr); free(GR); free(GT); free(Gf); free(Gpp);
    return 0;
}
