// gn_math.h -- the serial tail of one Gauss-Newton iteration (6x6 rank test, 6x6 inverse, SE(3) pseudo-exp,
// pose composition), written once for host and device.  Restates what RegisterPhotoICP::alignFrames360 does at
// RPI.h:4682-4697 through Eigen / MRPT:
//   (hessian + lambda*getDiagonalMatrix(hessian)).rank()   MRPT Eigen plugin = ColPivHouseholderQR::rank()
//   update_pose = -hessian.inverse() * gradient            Eigen fixed 6x6 inverse = PartialPivLU
//   CPose3D::exp(update, pseudo_exponential = true)        t verbatim, R = Rodrigues(w), float64
//   pose_estim_temp = exp(...).cast<float>() * pose_estim  float32 4x4 product
// All matrices are column-major like Eigen.
#pragma once

#if defined(__HIPCC__)
#define GN_HD __host__ __device__
#else
#define GN_HD
#endif

#include <math.h>

namespace gn {

constexpr float kEpsF = 1.1920929e-07f;

// Number of pivots of a column-pivoted Householder QR of the 6x6 matrix M whose magnitude exceeds
// |max pivot| * (epsilon * 6)  (Eigen's default threshold for ColPivHouseholderQR::rank()).
GN_HD inline int rank6(const float* M) {
    float A[6][6];
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) A[r][c] = M[c * 6 + r];
    float maxColSq = 0.f;
    for (int c = 0; c < 6; ++c) {
        float s = 0.f;
        for (int r = 0; r < 6; ++r) s += A[r][c] * A[r][c];
        maxColSq = s > maxColSq ? s : maxColSq;
    }
    const float threshold_helper = maxColSq * (kEpsF * kEpsF) / 6.f;
    float pivots[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float maxpivot = 0.f;
    int nonzero = 6;
    for (int k = 0; k < 6; ++k) {
        int best = k;
        float bestSq = -1.f;
        for (int c = k; c < 6; ++c) {
            float s = 0.f;
            for (int r = k; r < 6; ++r) s += A[r][c] * A[r][c];
            if (s > bestSq) {
                bestSq = s;
                best = c;
            }
        }
        if (bestSq < threshold_helper * (float)(6 - k)) {
            nonzero = k;
            break;
        }
        if (best != k)
            for (int r = 0; r < 6; ++r) {
                float tmp = A[r][k];
                A[r][k] = A[r][best];
                A[r][best] = tmp;
            }
        float tailSq = 0.f;
        for (int r = k + 1; r < 6; ++r) tailSq += A[r][k] * A[r][k];
        const float c0 = A[k][k];
        float beta, tau;
        float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (tailSq == 0.f) {
            tau = 0.f;
            beta = c0;
        } else {
            beta = sqrtf(c0 * c0 + tailSq);
            if (c0 >= 0.f) beta = -beta;
            for (int r = k + 1; r < 6; ++r) v[r] = A[r][k] / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        v[k] = 1.f;
        for (int c = k + 1; c < 6; ++c) {
            float dot = 0.f;
            for (int r = k; r < 6; ++r) dot += v[r] * A[r][c];
            dot *= tau;
            for (int r = k; r < 6; ++r) A[r][c] -= dot * v[r];
        }
        A[k][k] = beta;
        pivots[k] = beta;
        const float ab = fabsf(beta);
        maxpivot = ab > maxpivot ? ab : maxpivot;
    }
    const float thr = maxpivot * (kEpsF * 6.f);
    int rank = 0;
    for (int k = 0; k < nonzero; ++k) rank += (fabsf(pivots[k]) > thr) ? 1 : 0;
    return rank;
}

// inv = M^-1 through LU with partial pivoting, column by column against the identity.
GN_HD inline bool inverse6(const float* M, float* inv) {
    float LU[6][6];
    int perm[6];
    for (int r = 0; r < 6; ++r) {
        perm[r] = r;
        for (int c = 0; c < 6; ++c) LU[r][c] = M[c * 6 + r];
    }
    for (int k = 0; k < 6; ++k) {
        int piv = k;
        float best = fabsf(LU[k][k]);
        for (int r = k + 1; r < 6; ++r) {
            const float a = fabsf(LU[r][k]);
            if (a > best) {
                best = a;
                piv = r;
            }
        }
        if (best == 0.f) return false;
        if (piv != k) {
            for (int c = 0; c < 6; ++c) {
                float tmp = LU[k][c];
                LU[k][c] = LU[piv][c];
                LU[piv][c] = tmp;
            }
            int tp = perm[k];
            perm[k] = perm[piv];
            perm[piv] = tp;
        }
        for (int r = k + 1; r < 6; ++r) {
            LU[r][k] /= LU[k][k];
            for (int c = k + 1; c < 6; ++c) LU[r][c] -= LU[r][k] * LU[k][c];
        }
    }
    for (int col = 0; col < 6; ++col) {
        float y[6];
        for (int r = 0; r < 6; ++r) {
            float s = (perm[r] == col) ? 1.f : 0.f;
            for (int c = 0; c < r; ++c) s -= LU[r][c] * y[c];
            y[r] = s;
        }
        for (int r = 5; r >= 0; --r) {
            float s = y[r];
            for (int c = r + 1; c < 6; ++c) s -= LU[r][c] * y[c];
            y[r] = s / LU[r][r];
        }
        for (int r = 0; r < 6; ++r) inv[col * 6 + r] = y[r];
    }
    return true;
}

// sin(a)/a and (1-cos(a))/a^2 in float64.  On the device small angles (every Gauss-Newton update is one) use the
// Maclaurin series to full double accuracy instead of the library's argument-reduction sin/cos; the result is cast
// to float32 by the caller (RPI.h:4697), far above any difference in the last double bits.
GN_HD inline void sinc_cosc_sq(double x2, double& a, double& b);
GN_HD inline void sinc_cosc(double angle, double& a, double& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (angle < 0.5) {
        sinc_cosc_sq(angle * angle, a, b);
        return;
    }
#endif
    a = sin(angle) / angle;
    b = (1 - cos(angle)) / (angle * angle);
}
// the same for x2 = angle^2 < 0.25 (Maclaurin series, full double accuracy)
GN_HD inline void sinc_cosc_sq(double x2, double& a, double& b) {
    {
        // sin(x)/x = sum (-1)^k x^2k/(2k+1)!,  (1-cos x)/x^2 = sum (-1)^k x^2k/(2k+2)!   (k <= 9: x^18 < 4e-6^... 1e-22 rel)
        double sa = 1.0 / 121645100408832000.0;      // 1/19!
        sa = -sa * x2 + 1.0 / 355687428096000.0;     // 1/17!
        sa = -sa * x2 + 1.0 / 1307674368000.0;       // 1/15!
        sa = -sa * x2 + 1.0 / 6227020800.0;          // 1/13!
        sa = -sa * x2 + 1.0 / 39916800.0;            // 1/11!
        sa = -sa * x2 + 1.0 / 362880.0;              // 1/9!
        sa = -sa * x2 + 1.0 / 5040.0;                // 1/7!
        sa = -sa * x2 + 1.0 / 120.0;                 // 1/5!
        sa = -sa * x2 + 1.0 / 6.0;                   // 1/3!
        sa = -sa * x2 + 1.0;
        double cb = 1.0 / 2432902008176640000.0;     // 1/20!
        cb = -cb * x2 + 1.0 / 6402373705728000.0;    // 1/18!
        cb = -cb * x2 + 1.0 / 20922789888000.0;      // 1/16!
        cb = -cb * x2 + 1.0 / 87178291200.0;         // 1/14!
        cb = -cb * x2 + 1.0 / 479001600.0;           // 1/12!
        cb = -cb * x2 + 1.0 / 3628800.0;             // 1/10!
        cb = -cb * x2 + 1.0 / 40320.0;               // 1/8!
        cb = -cb * x2 + 1.0 / 720.0;                 // 1/6!
        cb = -cb * x2 + 1.0 / 24.0;                  // 1/4!
        cb = -cb * x2 + 0.5;
        a = sa;
        b = cb;
    }
}

// E = [ Rodrigues(v[3..5])  v[0..2] ; 0 0 0 1 ], float64, column-major.
GN_HD inline void se3_pseudo_exp(const double* v, double* E) {
    const double wx = v[3], wy = v[4], wz = v[5];
    const double angle = sqrt(wx * wx + wy * wy + wz * wz);
    double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    if (angle >= 128 * 2.220446049250313e-16) {
        const double W[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
        double a, b;
        sinc_cosc(angle, a, b);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double w2 = 0;
                for (int k = 0; k < 3; ++k) w2 += W[i][k] * W[k][j];
                R[i][j] += a * W[i][j] + b * w2;
            }
    }
    for (int k = 0; k < 16; ++k) E[k] = 0;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) E[j * 4 + i] = R[i][j];
        E[12 + i] = v[i];
    }
    E[15] = 1;
}

// THIRD-PARTY (MRPT 1.x CPose3D::exp(v, pseudo_exponential = false), RPI.h:4358, 4391): full SE(3) exponential,
// t = u + B (w x u) + C (w x (w x u)), small-angle series below theta^2 < 1e-8 / 1e-6.  Column-major 4x4 out.
GN_HD inline void se3_exp(const double* v, double* E) {
    const double ux = v[0], uy = v[1], uz = v[2], wx = v[3], wy = v[4], wz = v[5];
    const double theta_sq = wx * wx + wy * wy + wz * wz;
    const double theta = sqrt(theta_sq);
    const double cx = wy * uz - wz * uy, cy = wz * ux - wx * uz, cz = wx * uy - wy * ux;
    double A, B, tx, ty, tz;
    if (theta_sq < 1e-8) {
        A = 1.0 - theta_sq / 6.0;
        B = 0.5;
        tx = ux + 0.5 * cx; ty = uy + 0.5 * cy; tz = uz + 0.5 * cz;
    } else {
        double C;
        if (theta_sq < 1e-6) {
            C = (1.0 / 6.0) * (1.0 - theta_sq / 20.0);
            A = 1.0 - theta_sq * C;
            B = 0.5 - 0.25 * (1.0 / 6.0) * theta_sq;
        } else {
            const double inv_theta = 1.0 / theta;
            A = sin(theta) * inv_theta;
            B = (1 - cos(theta)) * (inv_theta * inv_theta);
            C = (1 - A) * (inv_theta * inv_theta);
        }
        const double dx = wy * cz - wz * cy, dy = wz * cx - wx * cz, dz = wx * cy - wy * cx;
        tx = ux + B * cx + C * dx; ty = uy + B * cy + C * dy; tz = uz + B * cz + C * dz;
    }
    const double wx2 = wx * wx, wy2 = wy * wy, wz2 = wz * wz;
    for (int k = 0; k < 16; ++k) E[k] = 0.0;
    E[0] = 1.0 - B * (wy2 + wz2);
    E[5] = 1.0 - B * (wx2 + wz2);
    E[10] = 1.0 - B * (wx2 + wy2);
    { const double a = A * wz, b = B * (wx * wy); E[4] = b - a; E[1] = b + a; }      // R(0,1), R(1,0)
    { const double a = A * wy, b = B * (wx * wz); E[8] = b + a; E[2] = b - a; }      // R(0,2), R(2,0)
    { const double a = A * wx, b = B * (wy * wz); E[9] = b - a; E[6] = b + a; }      // R(1,2), R(2,1)
    E[12] = tx; E[13] = ty; E[14] = tz;
    E[15] = 1.0;
}

GN_HD inline void mat4_mul(const float* A, const float* B, float* C) {
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            C[c * 4 + r] = ((A[0 * 4 + r] * B[c * 4 + 0] + A[1 * 4 + r] * B[c * 4 + 1]) + A[2 * 4 + r] * B[c * 4 + 2]) +
                           A[3 * 4 + r] * B[c * 4 + 3];
}

// One step: returns 0, or 1 when the rank test of (H + lambda diag H) fails.
GN_HD inline int step(const float* H, const float* g, float lambda, const float* pose, float* pose_tmp, float* update) {
    float M[36];
    for (int k = 0; k < 36; ++k) M[k] = H[k];
    for (int i = 0; i < 6; ++i) M[i * 6 + i] = H[i * 6 + i] + lambda * H[i * 6 + i];
    if (rank6(M) != 6) return 1;
    float inv[36];
    if (!inverse6(H, inv)) return 1;
    for (int r = 0; r < 6; ++r) {
        float s = 0.f;
        for (int c = 0; c < 6; ++c) s += (-inv[c * 6 + r]) * g[c];
        update[r] = s;
    }
    double ud[6], E[16];
    for (int i = 0; i < 6; ++i) ud[i] = (double)update[i];
    se3_pseudo_exp(ud, E);
    float Ef[16];
    for (int k = 0; k < 16; ++k) Ef[k] = (float)E[k];
    mat4_mul(Ef, pose, pose_tmp);
    return 0;
}

}  // namespace gn
