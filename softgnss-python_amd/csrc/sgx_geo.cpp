// Satellite positions and the least-squares position solution (reference geoFunctions/__init__.py; SURVEY.md
// section 8(f) item 4): scalar fp64 host code, a few kiloflops per measurement - nothing here belongs on the GPU.
// Each function follows the reference's operation order; libm's pow / sin / cos / atan2 stand where numpy's do
// (they may differ by an ulp), np.linalg.lstsq / matrix_rank / inv are replaced by Householder QR and Gauss-Jordan.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "sgx.h"

void sgx_set_error(const char* fmt, ...);

namespace {

// np.remainder for doubles: the result has the sign of the divisor
double py_mod(double a, double b) {
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

// geoFunctions/__init__.py:745-771
double check_t(double t) {
    const double half_week = 302400.0;
    if (t > half_week) return t - 2 * half_week;
    if (t < -half_week) return t + 2 * half_week;
    return t;
}

// geoFunctions/__init__.py:491-523: rotation about Z by the Earth's turn during the signal's travel
void e_r_corr(double traveltime, const double* x, double* out) {
    const double omegatau = 7.292115147e-05 * traveltime;
    const double c = cos(omegatau), s = sin(omegatau);
    out[0] = c * x[0] + s * x[1] + 0.0 * x[2];
    out[1] = -s * x[0] + c * x[1] + 0.0 * x[2];
    out[2] = 0.0 * x[0] + 0.0 * x[1] + 1.0 * x[2];
}

// geoFunctions/__init__.py:892-996 (C. Goad's iteration); latitude/longitude in degrees
void togeod(double a, double finv, double X, double Y, double Z, double* dphi_out, double* dlambda_out, double* h_out) {
    const double tolsq = 1e-10;
    const int maxit = 10;
    const double rtd = 180 / M_PI;
    const double esq = (finv < 1e-20) ? 0.0 : (2 - 1 / finv) / finv;
    const double oneesq = 1 - esq;
    const double P = sqrt(X * X + Y * Y);
    double dlambda = (P > 1e-20) ? atan2(Y, X) * rtd : 0.0;
    if (dlambda < 0) dlambda = dlambda + 360;
    const double r = sqrt(P * P + Z * Z);
    double sinphi = (r > 1e-20) ? Z / r : 0.0;
    double dphi = asin(sinphi);
    if (r < 1e-20) {
        *dphi_out = dphi;          // radians, like the reference's early return
        *dlambda_out = dlambda;
        *h_out = 0.0;
        return;
    }
    double h = r - a * (1 - sinphi * sinphi / finv);
    for (int i = 0; i < maxit; ++i) {
        sinphi = sin(dphi);
        const double cosphi = cos(dphi);
        const double N_phi = a / sqrt(1 - esq * sinphi * sinphi);
        const double dP = P - (N_phi + h) * cosphi;
        const double dZ = Z - (N_phi * oneesq + h) * sinphi;
        h = h + sinphi * dZ + cosphi * dP;
        dphi = dphi + (cosphi * dZ - sinphi * dP) / (N_phi + h);
        if ((dP * dP + dZ * dZ) < tolsq) break;
    }
    *dphi_out = dphi * rtd;
    *dlambda_out = dlambda;
    *h_out = h;
}

// geoFunctions/__init__.py:1003-1064: azimuth / elevation (degrees) and length of dx seen from X
void topocent(const double* X, const double* dx, double* Az, double* El, double* D) {
    const double dtr = M_PI / 180;
    double phi, lambda, h;
    togeod(6378137, 298.257223563, X[0], X[1], X[2], &phi, &lambda, &h);
    const double cl = cos(lambda * dtr), sl = sin(lambda * dtr);
    const double cb = cos(phi * dtr), sb = sin(phi * dtr);
    // F = [[-sl, -sb cl, cb cl], [cl, -sb sl, cb sl], [0, cb, sb]]; local = F^T dx
    const double E = -sl * dx[0] + cl * dx[1] + 0.0 * dx[2];
    const double N = -sb * cl * dx[0] + -sb * sl * dx[1] + cb * dx[2];
    const double U = cb * cl * dx[0] + cb * sl * dx[1] + sb * dx[2];
    const double hor_dis = sqrt(E * E + N * N);
    if (hor_dis < 1e-20) {
        *Az = 0.0;
        *El = 90.0;
    } else {
        *Az = atan2(E, N) / dtr;
        *El = atan2(U, hor_dis) / dtr;
    }
    if (*Az < 0) *Az = *Az + 360;
    *D = sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]);
}

// geoFunctions/__init__.py:1071-1186 (Goad & Goodman 1974): range correction in metres
double tropo(double sinel, double hsta, double p, double tkel, double hum, double hp, double htkel, double hhum) {
    const double a_e = 6378.137, b0 = 7.839257e-05, tlapse = -6.5;
    const double tkhum = tkel + tlapse * (hhum - htkel);
    const double atkel = 7.5 * (tkhum - 273.15) / (237.3 + tkhum - 273.15);
    const double e0 = 0.0611 * hum * pow(10.0, atkel);
    const double tksea = tkel - tlapse * htkel;
    const double em = -978.77 / (2870400.0 * tlapse * 1e-05);
    const double tkelh = tksea + tlapse * hhum;
    const double e0sea = e0 * pow(tksea / tkelh, 4 * em);
    const double tkelp = tksea + tlapse * hp;
    const double psea = p * pow(tksea / tkelp, em);
    if (sinel < 0) sinel = 0;
    double total = 0.0;
    bool done = false;
    double refsea = 7.7624e-05 / tksea;
    double htop = 1.1385e-05 / refsea;
    refsea = refsea * psea;
    double ref = refsea * pow((htop - hsta) / htop, 4);
    for (;;) {
        double rtop = pow(a_e + htop, 2) - pow(a_e + hsta, 2) * (1 - pow(sinel, 2));
        if (rtop < 0) rtop = 0;
        rtop = sqrt(rtop) - (a_e + hsta) * sinel;
        const double a = -sinel / (htop - hsta);
        const double b = -b0 * (1 - pow(sinel, 2)) / (htop - hsta);
        double rn[8];
        for (int i = 0; i < 8; ++i) rn[i] = pow(rtop, (double)(i + 2));
        double alpha[8] = {2 * a,
                           2 * pow(a, 2) + 4 * b / 3,
                           a * (pow(a, 2) + 3 * b),
                           pow(a, 4) / 5 + 2.4 * pow(a, 2) * b + 1.2 * pow(b, 2),
                           2 * a * b * (pow(a, 2) + 3 * b) / 3,
                           pow(b, 2) * (6 * pow(a, 2) + 4 * b) * 0.1428571,
                           0,
                           0};
        if (pow(b, 2) > 1e-35) {
            alpha[6] = a * pow(b, 3) / 2;
            alpha[7] = pow(b, 4) / 9;
        }
        double dr = rtop;
        double dot = 0.0;
        for (int i = 0; i < 8; ++i) dot += alpha[i] * rn[i];
        dr = dr + dot;
        total += dr * ref * 1000;
        if (done) break;
        done = true;
        refsea = (0.3719 / tksea - 1.292e-05) / tksea;
        htop = 1.1385e-05 * (1255.0 / tksea + 0.05) / refsea;
        ref = refsea * e0sea * pow((htop - hsta) / htop, 4);
    }
    return total;
}

// least squares solution of A x = b (m x 4, m >= 4) by Householder QR with column norms checked for rank;
// returns the numerical rank (np.linalg.matrix_rank's tolerance: sigma_max * max(m, 4) * eps, applied to |R_kk|)
int lstsq4(std::vector<double> A, std::vector<double> b, int m, double* x) {
    const int n = 4;
    int perm[4] = {0, 1, 2, 3};
    double rdiag[4] = {0, 0, 0, 0};
    for (int k = 0; k < n; ++k) {
        // pivot: remaining column of largest norm
        int best = k;
        double bestn = -1.0;
        for (int j = k; j < n; ++j) {
            double s = 0.0;
            for (int i = k; i < m; ++i) s += A[(size_t)i * n + j] * A[(size_t)i * n + j];
            if (s > bestn) {
                bestn = s;
                best = j;
            }
        }
        if (best != k) {
            for (int i = 0; i < m; ++i) std::swap(A[(size_t)i * n + k], A[(size_t)i * n + best]);
            std::swap(perm[k], perm[best]);
        }
        double norm = sqrt(bestn);
        if (norm == 0.0) {
            rdiag[k] = 0.0;
            continue;
        }
        if (A[(size_t)k * n + k] > 0) norm = -norm;
        // v = x - norm e1 (stored in column k, rows k..m-1)
        A[(size_t)k * n + k] -= norm;
        double vtv = 0.0;
        for (int i = k; i < m; ++i) vtv += A[(size_t)i * n + k] * A[(size_t)i * n + k];
        for (int j = k + 1; j < n; ++j) {
            double s = 0.0;
            for (int i = k; i < m; ++i) s += A[(size_t)i * n + k] * A[(size_t)i * n + j];
            const double f = 2.0 * s / vtv;
            for (int i = k; i < m; ++i) A[(size_t)i * n + j] -= f * A[(size_t)i * n + k];
        }
        {
            double s = 0.0;
            for (int i = k; i < m; ++i) s += A[(size_t)i * n + k] * b[(size_t)i];
            const double f = 2.0 * s / vtv;
            for (int i = k; i < m; ++i) b[(size_t)i] -= f * A[(size_t)i * n + k];
        }
        rdiag[k] = norm;
    }
    const double tol = fabs(rdiag[0]) * (double)std::max(m, n) * 2.220446049250313e-16;
    int rank = 0;
    for (int k = 0; k < n; ++k) rank += (fabs(rdiag[k]) > tol);
    if (rank < n) return rank;
    double y[4];
    for (int k = n - 1; k >= 0; --k) {
        double s = b[(size_t)k];
        for (int j = k + 1; j < n; ++j) s -= A[(size_t)k * n + j] * y[j];
        y[k] = s / rdiag[k];
    }
    for (int k = 0; k < n; ++k) x[perm[k]] = y[k];
    return rank;
}

bool inv4(const double* M, double* out) {
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = M[i * 4 + j];
            a[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r)
            if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
        if (a[p][c] == 0.0) return false;
        if (p != c)
            for (int j = 0; j < 8; ++j) std::swap(a[p][j], a[c][j]);
        const double d = a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
            if (f != 0.0)
                for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[i * 4 + j] = a[i][4 + j];
    return true;
}

// Clenshaw summations of geoFunctions/__init__.py:84-170
double clsin(const double* ar, int degree, double argument) {
    const double cos_arg = 2 * cos(argument);
    double hr1 = 0, hr = 0;
    for (int t = degree; t > 0; --t) {
        const double hr2 = hr1;
        hr1 = hr;
        hr = ar[t - 1] + cos_arg * hr1 - hr2;
    }
    return hr * sin(argument);
}

void clksin(const double* ar, int degree, double arg_real, double arg_imag, double* re, double* im) {
    const double sin_arg_r = sin(arg_real), cos_arg_r = cos(arg_real);
    const double sinh_arg_i = sinh(arg_imag), cosh_arg_i = cosh(arg_imag);
    double r = 2 * cos_arg_r * cosh_arg_i;
    double i = -2 * sin_arg_r * sinh_arg_i;
    double hr1 = 0, hr = 0, hi1 = 0, hi = 0;
    for (int t = degree; t > 0; --t) {
        const double hr2 = hr1;
        hr1 = hr;
        const double hi2 = hi1;
        hi1 = hi;
        const double z = ar[t - 1] + r * hr1 - i * hi - hr2;
        hi = i * hr1 + r * hi1 - hi2;
        hr = z;
    }
    r = sin_arg_r * cosh_arg_i;
    i = cos_arg_r * sinh_arg_i;
    *re = r * hr - i * hi;
    *im = r * hi + i * hr;
}

}   // namespace

extern "C" int sgx_check_t(double time, double* corr) {
    if (!corr) return SGX_E_ARG;
    *corr = check_t(time);
    return SGX_OK;
}

extern "C" int sgx_e_r_corr(double traveltime, const double* X_sat, double* X_sat_rot) {
    if (!X_sat || !X_sat_rot) return SGX_E_ARG;
    e_r_corr(traveltime, X_sat, X_sat_rot);
    return SGX_OK;
}

extern "C" int sgx_togeod(double a, double finv, double X, double Y, double Z, double* dphi, double* dlambda, double* h) {
    if (!dphi || !dlambda || !h) return SGX_E_ARG;
    togeod(a, finv, X, Y, Z, dphi, dlambda, h);
    return SGX_OK;
}

extern "C" int sgx_topocent(const double* X, const double* dx, double* Az, double* El, double* D) {
    if (!X || !dx || !Az || !El || !D) return SGX_E_ARG;
    topocent(X, dx, Az, El, D);
    return SGX_OK;
}

extern "C" int sgx_tropo(double sinel, double hsta, double p, double tkel, double hum, double hp, double htkel,
                         double hhum, double* ddr) {
    if (!ddr) return SGX_E_ARG;
    *ddr = tropo(sinel, hsta, p, tkel, hum, hp, htkel, hhum);
    return SGX_OK;
}

// geoFunctions/__init__.py:779-885.  eph: [32][27] in the order of sgx_ephemeris, row PRN-1.
extern "C" int sgx_satpos(double transmitTime, const int32_t* prnList, int32_t n, const double* eph,
                          double* satPositions, double* satClkCorr) {
    if (!prnList || !eph || !satPositions || !satClkCorr || n < 0) {
        sgx_set_error("bad argument to sgx_satpos");
        return SGX_E_ARG;
    }
    const double gpsPi = 3.14159265359;
    const double Omegae_dot = 7.2921151467e-05, GM = 3.986005e+14, F = -4.442807633e-10;
    for (int s = 0; s < n; ++s) {
        if (prnList[s] < 1 || prnList[s] > 32) {
            sgx_set_error("IndexError: PRN %d outside 1..32", prnList[s]);
            return SGX_E_RANGE;
        }
        const double* e = eph + (size_t)(prnList[s] - 1) * SGX_EPH_FIELDS;
        const double T_GD = e[3], t_oc = e[5], a_f2 = e[6], a_f1 = e[7], a_f0 = e[8];
        const double C_rs = e[10], deltan = e[11], M_0 = e[12], C_uc = e[13], ecc = e[14], C_us = e[15], sqrtA = e[16],
                     t_oe = e[17], C_ic = e[18], omega_0 = e[19], C_is = e[20], i_0 = e[21], C_rc = e[22],
                     omega = e[23], omegaDot = e[24], iDot = e[26];
        const double dt = check_t(transmitTime - t_oc);
        satClkCorr[s] = (a_f2 * dt + a_f1) * dt + a_f0 - T_GD;
        const double time = transmitTime - satClkCorr[s];
        const double a = sqrtA * sqrtA;
        const double tk = check_t(time - t_oe);
        const double n0 = sqrt(GM / pow(a, 3));
        const double nn = n0 + deltan;
        double M = M_0 + nn * tk;
        M = py_mod(M + 2 * gpsPi, 2 * gpsPi);
        double E = M;
        for (int ii = 0; ii < 10; ++ii) {
            const double E_old = E;
            E = M + ecc * sin(E);
            const double dE = py_mod(E - E_old, 2 * gpsPi);
            if (fabs(dE) < 1e-12) break;
        }
        E = py_mod(E + 2 * gpsPi, 2 * gpsPi);
        const double dtr = F * ecc * sqrtA * sin(E);
        const double nu = atan2(sqrt(1 - pow(ecc, 2)) * sin(E), cos(E) - ecc);
        double phi = nu + omega;
        phi = py_mod(phi, 2 * gpsPi);
        const double u = phi + C_uc * cos(2 * phi) + C_us * sin(2 * phi);
        const double r = a * (1 - ecc * cos(E)) + C_rc * cos(2 * phi) + C_rs * sin(2 * phi);
        const double i = i_0 + iDot * tk + C_ic * cos(2 * phi) + C_is * sin(2 * phi);
        double Omega = omega_0 + (omegaDot - Omegae_dot) * tk - Omegae_dot * t_oe;
        Omega = py_mod(Omega + 2 * gpsPi, 2 * gpsPi);
        satPositions[0 * (size_t)n + s] = cos(u) * r * cos(Omega) - sin(u) * r * cos(i) * sin(Omega);
        satPositions[1 * (size_t)n + s] = cos(u) * r * sin(Omega) + sin(u) * r * cos(i) * cos(Omega);
        satPositions[2 * (size_t)n + s] = sin(u) * r * sin(i);
        satClkCorr[s] = (a_f2 * dt + a_f1) * dt + a_f0 - T_GD + dtr;
    }
    return SGX_OK;
}

// geoFunctions/__init__.py:636-739.  satpos [3][n], obs [n]; pos[4] = X, Y, Z, dt (metres); el/az [n] degrees;
// dop[5] = GDOP PDOP HDOP VDOP TDOP.  *rank_deficient = 1 when the reference returns zeros early.
extern "C" int sgx_least_square_pos(const double* satpos, const double* obs, int32_t n, double c_mps,
                                    int32_t useTropCorr, double* pos, double* el, double* az, double* dop,
                                    int32_t* rank_deficient) {
    if (!satpos || !obs || !pos || !el || !az || !dop || !rank_deficient || n < 1) {
        sgx_set_error("bad argument to sgx_least_square_pos");
        return SGX_E_ARG;
    }
    const int nmbOfIterations = 7;
    const double dtr = M_PI / 180;
    for (int k = 0; k < 4; ++k) pos[k] = 0.0;
    for (int k = 0; k < 5; ++k) dop[k] = 0.0;
    for (int i = 0; i < n; ++i) az[i] = el[i] = 0.0;
    *rank_deficient = 0;
    std::vector<double> A((size_t)n * 4), omc((size_t)n);
    for (int iter = 0; iter < nmbOfIterations; ++iter) {
        for (int i = 0; i < n; ++i) {
            const double Xi[3] = {satpos[0 * (size_t)n + i], satpos[1 * (size_t)n + i], satpos[2 * (size_t)n + i]};
            double Rot_X[3];
            double trop;
            if (iter == 0) {
                Rot_X[0] = Xi[0];
                Rot_X[1] = Xi[1];
                Rot_X[2] = Xi[2];
                trop = 2;
            } else {
                const double rho2 = pow(Xi[0] - pos[0], 2) + pow(Xi[1] - pos[1], 2) + pow(Xi[2] - pos[2], 2);
                const double traveltime = sqrt(rho2) / c_mps;
                e_r_corr(traveltime, Xi, Rot_X);
                const double d[3] = {Rot_X[0] - pos[0], Rot_X[1] - pos[1], Rot_X[2] - pos[2]};
                double dist;
                topocent(pos, d, &az[i], &el[i], &dist);
                trop = useTropCorr ? tropo(sin(el[i] * dtr), 0.0, 1013.0, 293.0, 50.0, 0.0, 0.0, 0.0) : 0.0;
            }
            const double d0 = Rot_X[0] - pos[0], d1 = Rot_X[1] - pos[1], d2 = Rot_X[2] - pos[2];
            omc[(size_t)i] = obs[i] - sqrt(d0 * d0 + d1 * d1 + d2 * d2) - pos[3] - trop;
            A[(size_t)i * 4 + 0] = -d0 / obs[i];
            A[(size_t)i * 4 + 1] = -d1 / obs[i];
            A[(size_t)i * 4 + 2] = -d2 / obs[i];
            A[(size_t)i * 4 + 3] = 1;
        }
        double x[4] = {0, 0, 0, 0};
        const int rank = (n >= 4) ? lstsq4(A, omc, n, x) : n;
        if (rank != 4) {
            for (int k = 0; k < 4; ++k) pos[k] = 0.0;     // "exit gracefully": zeros, el / az / dop as they stand
            *rank_deficient = 1;
            return SGX_OK;
        }
        for (int k = 0; k < 4; ++k) pos[k] = pos[k] + x[k];
    }
    double AtA[16], Q[16];
    for (int r = 0; r < 4; ++r)
        for (int cc = 0; cc < 4; ++cc) {
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += A[(size_t)i * 4 + r] * A[(size_t)i * 4 + cc];
            AtA[r * 4 + cc] = s;
        }
    if (!inv4(AtA, Q)) {
        sgx_set_error("LinAlgError: Singular matrix");
        return SGX_E_RANGE;
    }
    dop[0] = sqrt(Q[0] + Q[5] + Q[10] + Q[15]);
    dop[1] = sqrt(Q[0] + Q[5] + Q[10]);
    dop[2] = sqrt(Q[0] + Q[5]);
    dop[3] = sqrt(Q[10]);
    dop[4] = sqrt(Q[15]);
    return SGX_OK;
}

// geoFunctions/__init__.py:7-77: ellipsoid i (0 International 1924 ... 4 WGS-84); degrees and metres
extern "C" int sgx_cart2geo(double X, double Y, double Z, int32_t i, double* phi_out, double* lambda_out, double* h_out) {
    if (!phi_out || !lambda_out || !h_out || i < 0 || i > 4) {
        sgx_set_error("IndexError: ellipsoid index %d outside 0..4", i);
        return SGX_E_RANGE;
    }
    const double a[5] = {6378388.0, 6378160.0, 6378135.0, 6378137.0, 6378137.0};
    const double f[5] = {1.0 / 297, 1 / 298.247, 1 / 298.26, 1 / 298.257222101, 1 / 298.257223563};
    double lambda = atan2(Y, X);
    const double ex2 = (2 - f[i]) * f[i] / pow(1 - f[i], 2);
    const double c = a[i] * sqrt(1 + ex2);
    double phi = atan(Z / (sqrt(pow(X, 2) + pow(Y, 2)) * (1 - (2 - f[i])) * f[i]));
    double h = 0.1, oldh = 0;
    int iterations = 0;
    while (fabs(h - oldh) > 1e-12) {
        oldh = h;
        const double N = c / sqrt(1 + ex2 * pow(cos(phi), 2));
        phi = atan(Z / (sqrt(pow(X, 2) + pow(Y, 2)) * (1 - (2 - f[i]) * f[i] * N / (N + h))));
        h = sqrt(pow(X, 2) + pow(Y, 2)) / cos(phi) - N;
        iterations += 1;
        if (iterations > 100) break;
    }
    *phi_out = phi * (180 / M_PI);
    *lambda_out = lambda * (180 / M_PI);
    *h_out = h;
    return SGX_OK;
}

// geoFunctions/__init__.py:529-571
extern "C" int sgx_find_utm_zone(double latitude, double longitude, int32_t* zone) {
    if (!zone) return SGX_E_ARG;
    if (longitude > 180 || longitude < -180) {
        sgx_set_error("IOError: Longitude value exceeds limits (-180:180).");
        return SGX_E_RANGE;
    }
    if (latitude > 84 || latitude < -80) {
        sgx_set_error("IOError: Latitude value exceeds limits (-80:84).");
        return SGX_E_RANGE;
    }
    int z = (int)trunc((180 + longitude) / 6) + 1;
    if (latitude > 72) {
        if (0 <= longitude && longitude < 9) z = 31;
        else if (9 <= longitude && longitude < 21) z = 33;
        else if (21 <= longitude && longitude < 33) z = 35;
        else if (33 <= longitude && longitude < 42) z = 37;
    } else if (56 <= latitude && latitude < 64) {
        if (3 <= longitude && longitude < 12) z = 32;
    }
    *zone = z;
    return SGX_OK;
}

// geoFunctions/__init__.py:176-372: ECEF -> UTM easting / northing / height on the International 1924 ellipsoid
extern "C" int sgx_cart2utm(double X, double Y, double Z, int32_t zone, double* E_out, double* N_out, double* U_out) {
    if (!E_out || !N_out || !U_out) return SGX_E_ARG;
    const double a = 6378388.0, f = 1.0 / 297.0;
    const double ex2 = (2 - f) * f / pow(1 - f, 2);
    const double c = a * sqrt(1 + ex2);
    const double vec[3] = {X, Y, Z - 4.5};
    const double alpha = 7.56e-07;
    const double Rv[3] = {1 * vec[0] + -alpha * vec[1] + 0 * vec[2], alpha * vec[0] + 1 * vec[1] + 0 * vec[2],
                          0 * vec[0] + 0 * vec[1] + 1 * vec[2]};
    const double trans[3] = {89.5, 93.8, 127.6};
    const double scale = 0.9999988;
    const double v[3] = {scale * Rv[0] + trans[0], scale * Rv[1] + trans[1], scale * Rv[2] + trans[2]};
    const double L = atan2(v[1], v[0]);
    double N1 = 6395000.0;
    const double nrm = sqrt(v[0] * v[0] + v[1] * v[1]);
    double B = atan2(v[2] / (pow(1 - f, 2) * N1), nrm / N1);
    double U = 0.1, oldU = 0;
    int iterations = 0;
    while (fabs(U - oldU) > 0.0001) {
        oldU = U;
        N1 = c / sqrt(1 + ex2 * pow(cos(B), 2));
        B = atan2(v[2] / (pow(1 - f, 2) * N1 + U), nrm / (N1 + U));
        U = nrm / cos(B) - N1;
        iterations += 1;
        if (iterations > 100) break;
    }
    const double m0 = 0.0004;
    const double n = f / (2 - f);
    const double m = pow(n, 2) * (1.0 / 4.0 + pow(n, 2) / 64);
    const double w = (a * (-n - m0 + m * (1 - m0))) / (1 + n);
    const double Q_n = a + w;
    const double E0 = 500000.0;
    double L0 = (zone - 30) * 6 - 3;
    const double bg[4] = {-0.00337077907, 4.73444769e-06, -8.2991457e-09, 1.5878533e-11};
    const double gtu[4] = {0.000841275991, 7.67306686e-07, 1.2129123e-09, 2.48508228e-12};
    const bool neg_geo = B < 0;
    double Bg_r = fabs(B);
    const double res_clensin = clsin(bg, 4, 2 * Bg_r);
    Bg_r = Bg_r + res_clensin;
    L0 = L0 * M_PI / 180;
    const double Lg_r = L - L0;
    const double cos_BN = cos(Bg_r);
    double Np = atan2(sin(Bg_r), cos(Lg_r) * cos_BN);
    double Ep = atanh(sin(Lg_r) * cos_BN);
    Np *= 2;
    Ep *= 2;
    double dN, dE;
    clksin(gtu, 4, Np, Ep, &dN, &dE);
    Np /= 2;
    Ep /= 2;
    Np += dN;
    Ep += dE;
    double N = Q_n * Np;
    const double E = Q_n * Ep + E0;
    if (neg_geo) N = -N + 20000000;
    *E_out = E;
    *N_out = N;
    *U_out = U;
    return SGX_OK;
}
