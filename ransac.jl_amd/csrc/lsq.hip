// lsq.hip -- least-squares refit of an extracted shape (the step of Schnabel et al. that the
// reference omits: docs/src/ransac.md:163-168, `refit` returns the shape unchanged).  NOT used in
// parity runs; rh_refit stays the reference's scan.
//
// Selection: enabled points compatible with the input shape at 3*eps (and the usual normal test),
// computed once with the refit scan kernel.  Plane: total least squares (centroid + smallest
// eigenvector of the scatter matrix).  Sphere / cylinder / cone: Gauss-Newton on the geometric
// distance.  Every iteration needs the normal equations  A'A, A'r, r'r  of the m <= 6 column
// Jacobian over up to millions of points: a (m+1) x K x (m+1) product, accumulated on the matrix
// cores with v_mfma_f64_16x16x4_f64 (4 points per instruction, rows staged through LDS into the
// A/B fragment layout), reduced over blocks in a fixed order (deterministic), solved on the host.
#include <math.h>
#include <string.h>

#include <vector>

#include "rh_internal.h"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

struct LsqParams {
    double v[16];   // kind-specific: see rows below
};

constexpr int LSQ_COLS = 8;   // Jacobian columns + residual, padded

// one row [J | r] for a point; columns beyond the kind's count stay 0
template <int KIND>
__device__ __forceinline__ void lsq_row(const LsqParams &P, double px, double py, double pz, double row[LSQ_COLS])
{
#pragma unroll
    for (int i = 0; i < LSQ_COLS; i++) row[i] = 0.0;
    if (KIND == RH_PLANE) {   // scatter moments about the reference point: [d, 1]
        row[0] = px - P.v[0]; row[1] = py - P.v[1]; row[2] = pz - P.v[2]; row[3] = 1.0;
        return;
    }
    if (KIND == RH_SPHERE) {  // v[0..2]=centre v[3]=R ; r = |p-c| - R
        const double dx = px - P.v[0], dy = py - P.v[1], dz = pz - P.v[2];
        const double nr = sqrt(dx * dx + dy * dy + dz * dz);
        const double inv = 1.0 / nr;
        row[0] = -dx * inv; row[1] = -dy * inv; row[2] = -dz * inv; row[3] = -1.0;
        row[4] = nr - P.v[3];
        return;
    }
    // cylinder: v[0..2]=c0 v[3..5]=a v[6]=R v[7..9]=e1 v[10..12]=e2
    // cone:     v[0..2]=apex v[3..5]=a v[6]=cos(phi) v[7..9]=e1 v[10..12]=e2 v[13]=sin(phi)
    const double tx = px - P.v[0], ty = py - P.v[1], tz = pz - P.v[2];
    const double ax = P.v[3], ay = P.v[4], az = P.v[5];
    const double h = ax * tx + ay * ty + az * tz;
    const double qx = tx - ax * h, qy = ty - ay * h, qz = tz - az * h;
    const double rho = sqrt(qx * qx + qy * qy + qz * qz);
    const double inv = 1.0 / rho;
    const double ux = qx * inv, uy = qy * inv, uz = qz * inv;
    const double u1 = ux * P.v[7] + uy * P.v[8] + uz * P.v[9];
    const double u2 = ux * P.v[10] + uy * P.v[11] + uz * P.v[12];
    if (KIND == RH_CYLINDER) {
        row[0] = -u1; row[1] = -u2; row[2] = -h * u1; row[3] = -h * u2; row[4] = -1.0;
        row[5] = rho - P.v[6];
        return;
    }
    const double c = P.v[6], s = P.v[13];
    const double t1 = tx * P.v[7] + ty * P.v[8] + tz * P.v[9];
    const double t2 = tx * P.v[10] + ty * P.v[11] + tz * P.v[12];
    row[0] = -ux * c + ax * s; row[1] = -uy * c + ay * s; row[2] = -uz * c + az * s;
    row[3] = -h * u1 * c - t1 * s;
    row[4] = -h * u2 * c - t2 * s;
    row[5] = -rho * s - h * c;
    row[6] = rho * c - h * s;
}

// partials[block][64]: the 8 x 8 corner of  sum_i row_i row_i'  over the block's selected points
template <int KIND>
__global__ void __launch_bounds__(256)
lsq_accumulate_kernel(const double *__restrict__ pts, int64_t stride, const uint64_t *__restrict__ sel, int64_t nwords,
                      const LsqParams P, double *__restrict__ partials)
{
    __shared__ double rows[4][64][LSQ_COLS + 1];   // +1: conflict-free column reads
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave, nwaves = (int64_t)gridDim.x * 4;
    f64x4 acc = { 0.0, 0.0, 0.0, 0.0 };
    const int comp = lane & 15, sub = lane >> 4;   // A[i=comp][k=sub], B[k=sub][j=comp]
    for (int64_t w = wave0; w < nwords; w += nwaves) {
        const uint64_t m = sel[w];
        if (m == 0) continue;
        double row[LSQ_COLS];
        const int64_t i = (w << 6) + lane;
        if ((m >> lane) & 1ULL) {
            lsq_row<KIND>(P, pts[i], pts[stride + i], pts[2 * stride + i], row);
        } else {
#pragma unroll
            for (int q = 0; q < LSQ_COLS; q++) row[q] = 0.0;
        }
#pragma unroll
        for (int q = 0; q < LSQ_COLS; q++) rows[wave][lane][q] = row[q];
        // wave-private LDS: the wave's own writes are visible to it after the waitcnt the compiler inserts
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const double a = comp < LSQ_COLS ? rows[wave][4 * t + sub][comp] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // D[row][col]: col = lane & 15, row = (lane >> 4) + 4 * reg.  Keep rows/cols < 8.
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int drow = sub + 4 * r;
        if (comp < 8) red[wave][drow * 8 + comp] = acc[r];
    }
    __syncthreads();
    if (threadIdx.x < 64)
        partials[(int64_t)blockIdx.x * 64 + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ void lsq_reduce_kernel(const double *__restrict__ partials, int nblocks, double *__restrict__ out)
{
    const int e = threadIdx.x;
    if (e >= 64) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; b++) s += partials[(int64_t)b * 64 + e];   // fixed order: reproducible
    out[e] = s;
}

// ---- tiny host linear algebra ----
bool solve_sym(int m, const double *A, const double *b, double *x)   // Gaussian elimination, partial pivoting
{
    double M[8][9];
    for (int i = 0; i < m; i++) { for (int j = 0; j < m; j++) M[i][j] = A[i * 8 + j]; M[i][m] = b[i]; }
    for (int k = 0; k < m; k++) {
        int piv = k;
        for (int i = k + 1; i < m; i++) if (fabs(M[i][k]) > fabs(M[piv][k])) piv = i;
        if (M[piv][k] == 0.0 || !(M[piv][k] == M[piv][k])) return false;
        if (piv != k) for (int j = 0; j <= m; j++) { const double t = M[k][j]; M[k][j] = M[piv][j]; M[piv][j] = t; }
        for (int i = k + 1; i < m; i++) {
            const double l = M[i][k] / M[k][k];
            for (int j = k; j <= m; j++) M[i][j] -= l * M[k][j];
        }
    }
    for (int i = m - 1; i >= 0; i--) {
        double s = M[i][m];
        for (int j = i + 1; j < m; j++) s -= M[i][j] * x[j];
        x[i] = s / M[i][i];
    }
    return true;
}

void jacobi_eig3(double S[3][3], double evec[3][3], double eval[3])   // symmetric 3x3, cyclic Jacobi
{
    double V[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    for (int sweep = 0; sweep < 60; sweep++) {
        const double off = fabs(S[0][1]) + fabs(S[0][2]) + fabs(S[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (S[p][q] == 0.0) continue;
                const double theta = (S[q][q] - S[p][p]) / (2 * S[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 3; k++) {
                    const double skp = S[k][p], skq = S[k][q];
                    S[k][p] = c * skp - s * skq;
                    S[k][q] = s * skp + c * skq;
                }
                for (int k = 0; k < 3; k++) {
                    const double spk = S[p][k], sqk = S[q][k];
                    S[p][k] = c * spk - s * sqk;
                    S[q][k] = s * spk + c * sqk;
                }
                for (int k = 0; k < 3; k++) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 3; i++) { eval[i] = S[i][i]; for (int k = 0; k < 3; k++) evec[i][k] = V[k][i]; }
}

void unit(double *v) { const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= n; v[1] /= n; v[2] /= n; }

void frame(const double a[3], double e1[3], double e2[3])   // orthonormal e1, e2 perpendicular to a
{
    int k = 0;
    if (fabs(a[1]) < fabs(a[k])) k = 1;
    if (fabs(a[2]) < fabs(a[k])) k = 2;
    double t[3] = { 0, 0, 0 };
    t[k] = 1.0;
    e1[0] = a[1] * t[2] - a[2] * t[1]; e1[1] = a[2] * t[0] - a[0] * t[2]; e1[2] = a[0] * t[1] - a[1] * t[0];
    unit(e1);
    e2[0] = a[1] * e1[2] - a[2] * e1[1]; e2[1] = a[2] * e1[0] - a[0] * e1[2]; e2[2] = a[0] * e1[1] - a[1] * e1[0];
    unit(e2);
}

}  // namespace

extern "C" int rh_refit_lsq(rh_cloud *c, const rh_shape *shape, const rh_params *p, int32_t max_iter, rh_shape *out,
                            int64_t *n_used, double *rms, int32_t *iters_done)
{
    if (c && c->f32) { rh_set_error("rh_refit_lsq: not available on Float32 clouds"); return RH_E_INVALID; }
    if (!c || !shape || !p || !out) { rh_set_error("rh_refit_lsq: NULL argument"); return RH_E_INVALID; }
    if (shape->kind < 0 || shape->kind > 3) { rh_set_error("unknown shape kind %d", shape->kind); return RH_E_INVALID; }
    RH_TRY(rh_validate_params(p));
    RH_HIP(hipSetDevice(c->device));
    RH_TRY(rh_join_batches(c));
    if (max_iter < 1) max_iter = 1;
    const int kind = shape->kind;
    // selection: compatible points within 3 eps (Schnabel et al. 2007, sec. 4.4)
    rh_prep P0;
    rh_prep_host(*shape, &P0);
    c->select_valid = false;
    RH_TRY(rhk_refit_mask(c, P0, kind, 3.0 * p->eps[kind], p->cos_alpha[kind]));

    int64_t cnt = 0;
    {
        RH_TRY(rhk_compact_mask(c, c->refit_mask, c->nwords, nullptr, 0, c->d_total));
        int32_t total = 0;
        RH_HIP(hipMemcpyAsync(&total, c->d_total, sizeof total, hipMemcpyDeviceToHost, c->stream));
        RH_HIP(hipStreamSynchronize(c->stream));
        cnt = total;
        if (cnt < 8) {
            rh_set_error("rh_refit_lsq: only %lld compatible points within 3 eps of the shape", (long long)cnt);
            return RH_E_INVALID;
        }
    }
    const int nblocks = 512;
    double *d_part = nullptr, *d_out = nullptr;
    RH_HIP(hipMalloc((void **)&d_part, sizeof(double) * 64 * nblocks));
    RH_HIP(hipMalloc((void **)&d_out, sizeof(double) * 64));
    auto done = [&](int rc) { (void)hipFree(d_part); (void)hipFree(d_out); return rc; };

    rh_shape cur = *shape;
    double M[64];
    int it = 0;
    double last_rms = 0;
    for (it = 0; it < max_iter; it++) {
        LsqParams L;
        memset(&L, 0, sizeof L);
        double e1[3], e2[3];
        if (kind == RH_PLANE) {
            for (int i = 0; i < 3; i++) L.v[i] = cur.v[i];
        } else if (kind == RH_SPHERE) {
            for (int i = 0; i < 4; i++) L.v[i] = cur.v[i];
        } else if (kind == RH_CYLINDER) {
            for (int i = 0; i < 3; i++) { L.v[i] = cur.v[3 + i]; L.v[3 + i] = cur.v[i]; }
            L.v[6] = cur.v[6];
            frame(&L.v[3], e1, e2);
            for (int i = 0; i < 3; i++) { L.v[7 + i] = e1[i]; L.v[10 + i] = e2[i]; }
        } else {
            for (int i = 0; i < 6; i++) L.v[i] = cur.v[i];
            L.v[6] = cos(cur.v[6] / 2);
            L.v[13] = sin(cur.v[6] / 2);
            frame(&L.v[3], e1, e2);
            for (int i = 0; i < 3; i++) { L.v[7 + i] = e1[i]; L.v[10 + i] = e2[i]; }
        }
#define LSQ_LAUNCH(K) hipLaunchKernelGGL((lsq_accumulate_kernel<K>), dim3(nblocks), dim3(256), 0, c->stream, c->full, c->n_pad, c->refit_mask, c->nwords, L, d_part)
        switch (kind) {
        case RH_PLANE: LSQ_LAUNCH(RH_PLANE); break;
        case RH_SPHERE: LSQ_LAUNCH(RH_SPHERE); break;
        case RH_CYLINDER: LSQ_LAUNCH(RH_CYLINDER); break;
        default: LSQ_LAUNCH(RH_CONE); break;
        }
#undef LSQ_LAUNCH
        hipLaunchKernelGGL(lsq_reduce_kernel, dim3(1), dim3(64), 0, c->stream, d_part, nblocks, d_out);
        hipError_t e = hipMemcpyAsync(M, d_out, sizeof M, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { rh_set_error("rh_refit_lsq: %s", hipGetErrorString(e)); return done(RH_E_NODEVICE); }

        if (kind == RH_PLANE) {
            const double N = M[3 * 8 + 3];
            const double s[3] = { M[0 * 8 + 3], M[1 * 8 + 3], M[2 * 8 + 3] };
            double S[3][3], evec[3][3], eval[3];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) S[i][j] = M[i * 8 + j] - s[i] * s[j] / N;
            jacobi_eig3(S, evec, eval);
            int mn = 0;
            for (int i = 1; i < 3; i++) if (eval[i] < eval[mn]) mn = i;
            double nn[3] = { evec[mn][0], evec[mn][1], evec[mn][2] };
            unit(nn);
            if (nn[0] * shape->v[3] + nn[1] * shape->v[4] + nn[2] * shape->v[5] < 0) for (int i = 0; i < 3; i++) nn[i] = -nn[i];
            for (int i = 0; i < 3; i++) { cur.v[i] = cur.v[i] + s[i] / N; cur.v[3 + i] = nn[i]; }
            last_rms = sqrt(fmax(eval[mn], 0.0) / N);
            it++;
            break;
        }
        const int m = kind == RH_SPHERE ? 4 : (kind == RH_CYLINDER ? 5 : 6);
        double A[64], b[8], x[8];
        double tr = 0;
        for (int i = 0; i < m; i++) tr += M[i * 8 + i];
        for (int i = 0; i < m; i++) {
            for (int j = 0; j < m; j++) A[i * 8 + j] = M[i * 8 + j];
            A[i * 8 + i] += 1e-12 * tr;   // Levenberg damping, keeps rank-deficient directions tame
            b[i] = -M[i * 8 + m];
        }
        const double rr = M[m * 8 + m];
        if (!solve_sym(m, A, b, x)) { rh_set_error("rh_refit_lsq: singular normal equations"); return done(RH_E_INVALID); }
        if (kind == RH_SPHERE) {
            for (int i = 0; i < 4; i++) cur.v[i] += x[i];
        } else if (kind == RH_CYLINDER) {
            for (int i = 0; i < 3; i++) {
                cur.v[3 + i] += x[0] * e1[i] + x[1] * e2[i];
                cur.v[i] += x[2] * e1[i] + x[3] * e2[i];
            }
            unit(&cur.v[0]);
            cur.v[6] += x[4];
        } else {
            for (int i = 0; i < 3; i++) {
                cur.v[i] += x[i];
                cur.v[3 + i] += x[3] * e1[i] + x[4] * e2[i];
            }
            unit(&cur.v[3]);
            cur.v[6] += 2 * x[5];   // opang = 2 phi
        }
        double step = 0;
        for (int i = 0; i < m; i++) step += x[i] * x[i];
        last_rms = rr;   // sum of squares before this step; converted below
        if (sqrt(step) < 1e-11) { it++; break; }
    }
    if (kind != RH_PLANE) last_rms = sqrt(last_rms / (double)cnt);
    rh_shape_finalize(&cur);
    *out = cur;
    if (n_used) *n_used = cnt;
    if (rms) *rms = last_rms;
    if (iters_done) *iters_done = it;
    return done(RH_OK);
}
