// Small dense solvers of the MST initialisation (reference: dust3r/cloud_opt/init_im_poses.py:129-252, 415-482) on the
// device, so that init='mst' runs without LAPACK and without a host round trip per problem:
//   a3r_umeyama_solve   closed-form weighted similarity registration from the 17 raw moments of a3r_umeyama_moments
//                       (the reference calls roma.rigid_points_registration, :415-418): 3x3 SVD by Jacobi, one thread per problem;
//   a3r_pnp_solve       camera pose of an image from its world-space point map with known intrinsics (stands in for
//                       cv2.solvePnPRansac + SQPNP of fast_pnp, :442-482, which is stochastic and absent here -- PARITY UNPINNED):
//                       closed-form start (the similarity carrying the pixel rays onto the world points, a least-squares fit over
//                       all points), then damped Gauss-Newton on the reprojection error with redescending (Cauchy) weights whose
//                       scale is annealed from 40 px to the 5 px inlier threshold -- robust to the gross outliers RANSAC is there
//                       for (tests/test_gpu_ops.py: 40 % outliers); inlier count (< 5 px, in front) for fast_pnp's focal search.
// Everything is float64 inside; problems are batched over blockIdx / threads; no host synchronisation between the rounds.
#include "common.h"
#include <cmath>

namespace a3r {

// cyclic Jacobi eigen-decomposition of the symmetric N x N matrix A (destroyed: its diagonal holds the eigenvalues on return);
// the columns of V are the eigenvectors.  Arrays live wherever the caller put them (LDS for N = 12, registers for N = 3).
template <int N, class Mat>
__device__ inline void jacobi_eigh(Mat& A, Mat& V, int max_sweeps) {
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < max_sweeps; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < N; i++) {
            diag += A[i][i] * A[i][i];
            for (int j = i + 1; j < N; j++) off += A[i][j] * A[i][j];
        }
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (int p = 0; p < N - 1; p++)
            for (int q = p + 1; q < N; q++) {
                const double apq = A[p][q];
                if (apq == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < N; k++) {                      // A <- J^T A J on rows / columns p, q
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < N; k++) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < N; k++) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

struct M3 { double m[3][3]; __device__ double* operator[](int i) { return m[i]; } __device__ const double* operator[](int i) const { return m[i]; } };

__device__ inline double det3(const M3& a) {
    return a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
           a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
}

// A = U diag(S) V^T with det(U) = +1 and S[2] carrying the sign that makes this true (|S| are the singular values, descending):
// V and S^2 from the Jacobi eigen-decomposition of A^T A, u_i = A v_i / s_i for the two leading directions, u_2 = u_0 x u_1.
__device__ inline void svd3(const M3& A, M3& U, double S[3], M3& V) {
    M3 ata, ev;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) ata[i][j] = A[0][i] * A[0][j] + A[1][i] * A[1][j] + A[2][i] * A[2][j];
    jacobi_eigh<3>(ata, ev, 30);
    int idx[3] = {0, 1, 2};                                        // descending eigenvalues
    for (int a = 0; a < 2; a++)
        for (int b = a + 1; b < 3; b++)
            if (ata[idx[b]][idx[b]] > ata[idx[a]][idx[a]]) { const int t = idx[a]; idx[a] = idx[b]; idx[b] = t; }
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) V[r][c] = ev[r][idx[c]];
    double u[3][3];
    for (int c = 0; c < 2; c++) {
        double n = 0.0;
        for (int r = 0; r < 3; r++) { u[c][r] = A[r][0] * V[0][c] + A[r][1] * V[1][c] + A[r][2] * V[2][c]; n += u[c][r] * u[c][r]; }
        n = sqrt(n);
        S[c] = n;
        if (c == 1) {                                              // re-orthogonalise against u_0 (nearly rank-1 inputs)
            const double d = u[1][0] * u[0][0] + u[1][1] * u[0][1] + u[1][2] * u[0][2];
            for (int r = 0; r < 3; r++) u[1][r] -= d * u[0][r];
            n = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
        }
        if (n > 0.0) for (int r = 0; r < 3; r++) u[c][r] /= n;
        else for (int r = 0; r < 3; r++) u[c][r] = r == c ? 1.0 : 0.0;
    }
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    double s2 = 0.0;
    for (int r = 0; r < 3; r++) s2 += u[2][r] * (A[r][0] * V[0][2] + A[r][1] * V[1][2] + A[r][2] * V[2][2]);
    S[2] = s2;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) U[r][c] = u[c][r];
}

// ---- weighted Umeyama from raw moments: out[b] = (s, R row-major [9], T [3]) float32
__global__ __launch_bounds__(64) void umeyama_solve_kernel(const double* __restrict__ partial, int nch, int B, float* __restrict__ out) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double m[17];
    for (int j = 0; j < 17; j++) m[j] = 0.0;
    for (int c = 0; c < nch; c++)
        for (int j = 0; j < 17; j++) m[j] += partial[((size_t)b * nch + c) * 17 + j];
    const double w0 = m[0];
    double xm[3], ym[3];
    for (int i = 0; i < 3; i++) { xm[i] = m[1 + i] / w0; ym[i] = m[4 + i] / w0; }
    const double var_x = m[7] / w0 - (xm[0] * xm[0] + xm[1] * xm[1] + xm[2] * xm[2]);
    M3 cov, U, V;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) cov[i][j] = m[8 + 3 * i + j] / w0 - ym[i] * xm[j];
    double S[3];
    svd3(cov, U, S, V);
    // R = U diag(1, 1, d) V^T with d = det(U V^T) = det(V) here (det(U) = +1); s = (S0 + S1 + d S2) / var_x
    const double d = det3(V) >= 0.0 ? 1.0 : -1.0;
    double R[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i][j] = U[i][0] * V[j][0] + U[i][1] * V[j][1] + d * U[i][2] * V[j][2];
    const double s = (S[0] + S[1] + d * S[2]) / var_x;
    float* o = out + (size_t)b * 13;
    o[0] = (float)s;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) o[1 + 3 * i + j] = (float)R[i][j];
    for (int i = 0; i < 3; i++) o[10 + i] = (float)(ym[i] - s * (R[i][0] * xm[0] + R[i][1] * xm[1] + R[i][2] * xm[2]));
}

// ---- batched PnP with known intrinsics
struct PnpDesc { const float* pts; const unsigned char* msk; int H, W, step, n; float focal, ppx, ppy, pad; };
constexpr int PNP_SUMS = 32;      // phase 0: the 17 Umeyama moments (rays -> world points); phase 1: J^T W J (21) + J^T W e (6);
                                  // both: [27] points used, [28] in front of the camera, [29] inliers in front, [30] inliers behind, [31] truncated squared error
constexpr int PNP_CHUNK = 2048;   // points per block

__device__ inline double pnp_tau(int it) { const double t = 40.0 * exp2(-(double)it); return t > 5.0 ? t : 5.0; }

// phase 0: moments for the closed-form start (pose unused).  phase 1: Gauss-Newton normal equations of the reprojection error at
// pose[b] = world -> camera [R (9, row-major) | t (3)], Cauchy weights 1 / (1 + (e / tau)^2) with tau annealed from 40 px to the
// 5 px inlier threshold (it = iteration number), points behind the camera get weight 0.
__global__ __launch_bounds__(256) void pnp_accumulate_kernel(const PnpDesc* __restrict__ desc, const double* __restrict__ pose, int phase, int it,
                                                             double* __restrict__ partial) {
    __shared__ double red[4][PNP_SUMS];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const PnpDesc d = desc[b];
    double a[PNP_SUMS];
#pragma unroll
    for (int j = 0; j < PNP_SUMS; j++) a[j] = 0.0;
    double R[9], t[3];
    if (phase == 1) {
        for (int j = 0; j < 9; j++) R[j] = pose[(size_t)b * 12 + j];
        for (int j = 0; j < 3; j++) t[j] = pose[(size_t)b * 12 + 9 + j];
    }
    const double tau = pnp_tau(it), f = d.focal;
    for (int i = 0; i < PNP_CHUNK / 256; i++) {
        const int p = chunk * PNP_CHUNK + i * 256 + tid;
        if (p >= d.n) continue;
        const long pix = (long)p * d.step;
        if (!d.msk[pix]) continue;
        const int py = (int)(pix / d.W), px = (int)(pix - (long)py * d.W);
        const double X0 = d.pts[3 * pix], X1 = d.pts[3 * pix + 1], X2 = d.pts[3 * pix + 2];
        const double rx = ((double)px - d.ppx) / f, ry = ((double)py - d.ppy) / f;
        a[27] += 1.0;
        if (phase == 0) {                                           // x = ray (rx, ry, 1), y = world point, weight 1
            a[0] += 1.0;
            a[1] += rx; a[2] += ry; a[3] += 1.0;
            a[4] += X0; a[5] += X1; a[6] += X2;
            a[7] += rx * rx + ry * ry + 1.0;
            a[8] += X0 * rx; a[9] += X0 * ry; a[10] += X0;
            a[11] += X1 * rx; a[12] += X1 * ry; a[13] += X1;
            a[14] += X2 * rx; a[15] += X2 * ry; a[16] += X2;
            continue;
        }
        const double x = R[0] * X0 + R[1] * X1 + R[2] * X2 + t[0], y = R[3] * X0 + R[4] * X1 + R[5] * X2 + t[1];
        const double z = R[6] * X0 + R[7] * X1 + R[8] * X2 + t[2];
        const double zz = fabs(z) > 1e-9 ? z : 1e-9;
        const double ex = (x / zz - rx) * f, ey = (y / zz - ry) * f;
        const double e2 = ex * ex + ey * ey;
        if (z > 0.0) a[28] += 1.0;
        if (e2 < 25.0) a[z > 0.0 ? 29 : 30] += 1.0;
        if (z <= 0.0) continue;
        a[31] += e2 < 25.0 ? e2 : 25.0;                             // truncated squared error: tie-break of the focal search
        const double w = 1.0 / (1.0 + e2 / (tau * tau));
        // left perturbation c' = c + omega x c + tau_t:  d(ex) / d(omega, tau_t), d(ey) / d(omega, tau_t)
        const double iz = f / zz, xz = x / zz, yz = y / zz;
        const double Jx[6] = {-xz * yz * f, (1.0 + xz * xz) * f, -yz * f, iz, 0.0, -xz * iz};
        const double Jy[6] = {-(1.0 + yz * yz) * f, xz * yz * f, xz * f, 0.0, iz, -yz * iz};
        int k = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) {
#pragma unroll
            for (int c = r; c < 6; c++) a[k++] += w * (Jx[r] * Jx[c] + Jy[r] * Jy[c]);
        }
#pragma unroll
        for (int r = 0; r < 6; r++) a[21 + r] += w * (Jx[r] * ex + Jy[r] * ey);
    }
#pragma unroll
    for (int j = 0; j < PNP_SUMS; j++) {
        double v = a[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][j] = v;
    }
    __syncthreads();
    if (tid < PNP_SUMS) partial[((size_t)b * gridDim.x + chunk) * PNP_SUMS + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// one thread per problem: reduce the partials (fixed order), then
//   mode 0: closed-form start -- the similarity (s, Rwc, T) carrying the rays onto the world points (exact for a fronto-parallel
//           plane, a least-squares fit over ALL points otherwise): world -> camera R = Rwc^T, t = -R T;
//   mode 1: one damped Gauss-Newton step, pose <- exp(omega) pose;
//   mode 2: finish: validity, camera-to-world, fast_pnp's score.
__global__ __launch_bounds__(64) void pnp_solve_kernel(const PnpDesc* __restrict__ desc, const double* __restrict__ partial, int nch, int mode, int B,
                                                       double* __restrict__ pose, float* __restrict__ c2w, float* __restrict__ info) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double m[PNP_SUMS];
    for (int j = 0; j < PNP_SUMS; j++) m[j] = 0.0;
    for (int c = 0; c < nch; c++)
        for (int j = 0; j < PNP_SUMS; j++) m[j] += partial[((size_t)b * nch + c) * PNP_SUMS + j];
    double* P = pose + (size_t)b * 12;
    if (mode == 0) {
        const double w0 = m[0] > 0.0 ? m[0] : 1.0;
        double xm[3], ym[3];
        for (int i = 0; i < 3; i++) { xm[i] = m[1 + i] / w0; ym[i] = m[4 + i] / w0; }
        const double var_x = m[7] / w0 - (xm[0] * xm[0] + xm[1] * xm[1] + xm[2] * xm[2]);
        M3 cov, U, V;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) cov[i][j] = m[8 + 3 * i + j] / w0 - ym[i] * xm[j];
        double S[3];
        svd3(cov, U, S, V);
        const double d = det3(V) >= 0.0 ? 1.0 : -1.0;
        double Rwc[3][3], T[3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rwc[i][j] = U[i][0] * V[j][0] + U[i][1] * V[j][1] + d * U[i][2] * V[j][2];
        const double s = (S[0] + S[1] + d * S[2]) / (var_x > 0.0 ? var_x : 1.0);
        for (int i = 0; i < 3; i++) T[i] = ym[i] - s * (Rwc[i][0] * xm[0] + Rwc[i][1] * xm[1] + Rwc[i][2] * xm[2]);
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) P[3 * i + j] = Rwc[j][i];
            P[9 + i] = -(Rwc[0][i] * T[0] + Rwc[1][i] * T[1] + Rwc[2][i] * T[2]);
        }
        return;
    }
    if (mode == 1) {
        // (J^T W J + lambda diag) delta = -J^T W e by Cholesky; a failed factorisation (degenerate problem) leaves the pose alone
        double A[6][6], g[6], L[6][6];
        int k = 0;
        for (int r = 0; r < 6; r++)
            for (int c = r; c < 6; c++) { A[r][c] = m[k]; A[c][r] = m[k]; k++; }
        double tr = 0.0;
        for (int r = 0; r < 6; r++) { g[r] = -m[21 + r]; tr += A[r][r]; }
        for (int r = 0; r < 6; r++) A[r][r] += 1e-9 * tr;
        bool ok = tr > 0.0 && isfinite(tr);
        for (int i = 0; i < 6 && ok; i++)
            for (int j = 0; j <= i; j++) {
                double sum = A[i][j];
                for (int q = 0; q < j; q++) sum -= L[i][q] * L[j][q];
                if (i == j) { if (sum <= 0.0) { ok = false; break; } L[i][i] = sqrt(sum); }
                else L[i][j] = sum / L[j][j];
            }
        if (!ok) return;
        double yv[6], dl[6];
        for (int i = 0; i < 6; i++) { double sum = g[i]; for (int q = 0; q < i; q++) sum -= L[i][q] * yv[q]; yv[i] = sum / L[i][i]; }
        for (int i = 5; i >= 0; i--) { double sum = yv[i]; for (int q = i + 1; q < 6; q++) sum -= L[q][i] * dl[q]; dl[i] = sum / L[i][i]; }
        // dR = exp([omega]x) (Rodrigues); R <- dR R, t <- dR t + tau_t
        const double th2 = dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2], th = sqrt(th2);
        const double ca = th > 1e-12 ? sin(th) / th : 1.0, cb = th > 1e-12 ? (1.0 - cos(th)) / th2 : 0.5;
        const double K[3][3] = {{0.0, -dl[2], dl[1]}, {dl[2], 0.0, -dl[0]}, {-dl[1], dl[0], 0.0}};
        double dR[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double kk = 0.0;
                for (int q = 0; q < 3; q++) kk += K[i][q] * K[q][j];
                dR[i][j] = (i == j ? 1.0 : 0.0) + ca * K[i][j] + cb * kk;
            }
        double Rn[9], tn[3];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) Rn[3 * i + j] = dR[i][0] * P[j] + dR[i][1] * P[3 + j] + dR[i][2] * P[6 + j];
            tn[i] = dR[i][0] * P[9] + dR[i][1] * P[10] + dR[i][2] * P[11] + dl[3 + i];
        }
        for (int j = 0; j < 9; j++) P[j] = Rn[j];
        for (int j = 0; j < 3; j++) P[9 + j] = tn[j];
        return;
    }
    const double n = m[27], front = m[28];
    M3 Rm;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rm[i][j] = P[3 * i + j];
    const bool ok = n >= 6.0 && front >= 0.5 * n && det3(Rm) > 0.0 && isfinite(P[9] + P[10] + P[11]);
    float* o = c2w + (size_t)b * 16;                               // inverse of [R t; 0 1] = [R^T, -R^T t; 0 1]
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) o[4 * i + j] = (float)P[3 * j + i];
        o[4 * i + 3] = (float)-(P[i] * P[9] + P[3 + i] * P[10] + P[6 + i] * P[11]);
    }
    o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
    info[4 * b] = ok ? 1.f : 0.f;
    info[4 * b + 1] = (float)m[29];                                // fast_pnp's score: inliers (< 5 px) in front of the camera
    info[4 * b + 2] = (float)m[31];                                // sum over the points in front of min(e^2, 25 px^2)
    info[4 * b + 3] = desc[b].focal;
}

}  // namespace a3r
using namespace a3r;

extern "C" int a3r_umeyama_solve(const double* partial, int nch, int B, float* out, void* stream) {
    A3R_CHECK_ARG(partial && out && nch > 0 && B > 0, "a3r_umeyama_solve: bad argument");
    hipLaunchKernelGGL(umeyama_solve_kernel, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), partial, nch, B, out);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" size_t a3r_pnp_desc_bytes(void) { return sizeof(PnpDesc); }
extern "C" int a3r_pnp_chunks(int n_max) { return n_max > 0 ? (n_max + PNP_CHUNK - 1) / PNP_CHUNK : 0; }
extern "C" size_t a3r_pnp_work_bytes(int B, int n_max) { return ((size_t)B * a3r_pnp_chunks(n_max) * PNP_SUMS + (size_t)B * 12) * sizeof(double); }

extern "C" int a3r_pnp_solve(const void* desc, int B, int n_max, int iterations, void* work, float* c2w, float* info, void* stream) {
    A3R_CHECK_ARG(desc && work && c2w && info, "a3r_pnp_solve: null pointer");
    A3R_CHECK_ARG(B > 0 && B <= 65535 && n_max > 0 && iterations >= 0 && iterations <= 64, "a3r_pnp_solve: bad shape B=%d n_max=%d iterations=%d", B, n_max, iterations);
    hipStream_t st = as_stream(stream);
    const int nch = a3r_pnp_chunks(n_max);
    double* partial = static_cast<double*>(work);
    double* pose = partial + (size_t)B * nch * PNP_SUMS;
    const PnpDesc* d = static_cast<const PnpDesc*>(desc);
    const dim3 sgrid((B + 63) / 64);
    hipLaunchKernelGGL(pnp_accumulate_kernel, dim3(nch, B), dim3(256), 0, st, d, pose, 0, 0, partial);
    hipLaunchKernelGGL(pnp_solve_kernel, sgrid, dim3(64), 0, st, d, partial, nch, 0, B, pose, c2w, info);
    for (int it = 0; it < iterations; it++) {
        hipLaunchKernelGGL(pnp_accumulate_kernel, dim3(nch, B), dim3(256), 0, st, d, pose, 1, it, partial);
        hipLaunchKernelGGL(pnp_solve_kernel, sgrid, dim3(64), 0, st, d, partial, nch, 1, B, pose, c2w, info);
    }
    // statistics of the final pose (points in front, inliers), then validity / camera-to-world
    hipLaunchKernelGGL(pnp_accumulate_kernel, dim3(nch, B), dim3(256), 0, st, d, pose, 1, iterations, partial);
    hipLaunchKernelGGL(pnp_solve_kernel, sgrid, dim3(64), 0, st, d, partial, nch, 2, B, pose, c2w, info);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}
