#!/bin/bash
# AddressSanitizer + UBSan over the scalar host code of libsgx (csrc/sgx_geo.cpp, csrc/sgx_navhost.cpp): a CPU build of those files alone,
# driven with randomised and degenerate inputs.  GPU sanitizers are not available on this pool; this covers the
# part of the library that never touches the device.   Usage: bash tools/sanitize_host.sh
set -e
cd "$(dirname "$0")/.."
out=${TMPDIR:-/tmp}/sgx_san
mkdir -p "$out"
cat > "$out/driver.cpp" <<'CPP'
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "sgx.h"
void sgx_set_error(const char* fmt, ...) { (void)fmt; }
static double rnd(double a, double b) { return a + (b - a) * (double)rand() / RAND_MAX; }
int sgx_nav_select(const double* I_P, const short* corr, int32_t n_ch, int32_t ms, int32_t search_start,
                   int32_t* firstSubFrame);
static void nav_round(int it) {
    // a +-1 bit stream with preambles every 300 bits at a random phase, random amplitude, truncated at random
    const int ms = 200 + rand() % 9000, nch = 1 + rand() % 3;
    std::vector<double> ip((size_t)nch * ms);
    std::vector<short> corr((size_t)nch * ms);
    static const int pre[8] = {1, -1, -1, -1, 1, -1, 1, 1};
    for (int c = 0; c < nch; ++c) {
        const int phase = rand() % 6000;
        for (int t = 0; t < ms; ++t) {
            const int bit = ((t + phase) / 20) % 300;
            const double v = bit < 8 ? pre[bit] : ((rand() & 1) ? 1 : -1);
            ip[(size_t)c * ms + t] = v * rnd(0.2, 3.0) + rnd(-0.3, 0.3);
        }
        for (int t = 0; t < ms; ++t) {
            int acc = 0;
            for (int k = 0; k < 160; ++k)
                if (t + k < ms) acc += (ip[(size_t)c * ms + t + k] > 0 ? 1 : -1) * pre[k / 20];
            corr[(size_t)c * ms + t] = (short)acc;
        }
    }
    std::vector<int32_t> first(nch);
    sgx_nav_select(ip.data(), corr.data(), nch, ms, it % 5 == 0 ? rand() % ms : 0, first.data());
    uint8_t bits[1501];
    int32_t nb = 0;
    sgx_nav_bits(ip.data(), ms, rand() % ms, bits, &nb);
    double w[32];
    for (auto& v : w) v = (rand() & 1) ? 1.0 : -1.0;
    int32_t st = 0;
    sgx_nav_parity_check(w, &st);
    std::vector<uint8_t> frame(1500);
    for (auto& b : frame) b = rand() & 1;
    double eph[SGX_EPH_FIELDS];
    int64_t tow = 0;
    sgx_ephemeris(frame.data(), it % 7 == 0 ? 1499 : 1500, rand() & 1, eph, &tow);
    std::vector<double> abs_s((size_t)nch * ms), when(4), pr(4);
    for (auto& v : abs_s) v = rnd(0, 1e9);
    for (auto& v : when) v = rnd(-50, ms + 50);
    int32_t list[3] = {0, 1, 2};
    sgx_pseudoranges(abs_s.data(), nch, ms, when.data(), list, rand() % 4, 4, 38192, 68.802, 299792458.0, pr.data());
}

int main() {
    srand(7);
    long calls = 0;
    for (int it = 0; it < 300; ++it) nav_round(it);
    calls += 300 * 5;
    for (int it = 0; it < 20000; ++it) {
        std::vector<double> eph(32 * SGX_EPH_FIELDS);
        for (auto& v : eph) v = 0.0;
        for (int p = 0; p < 32; ++p) {
            double* e = &eph[p * SGX_EPH_FIELDS];
            e[16] = it % 50 == 0 ? 0.0 : rnd(5100, 5200);   // sqrtA (sometimes degenerate)
            e[14] = rnd(0, it % 97 == 0 ? 1.5 : 0.03);      // e
            e[12] = rnd(-4, 4); e[19] = rnd(-4, 4); e[23] = rnd(-4, 4); e[21] = rnd(0.9, 1.0);
            e[5] = e[17] = 100800; e[11] = rnd(4e-9, 5e-9); e[24] = -8e-9;
        }
        int32_t prn[12];
        const int n = 1 + rand() % 12;
        for (int i = 0; i < n; ++i) prn[i] = 1 + rand() % 32;
        std::vector<double> pos(3 * n), clk(n), obs(n), el(n), az(n);
        sgx_satpos(100800 + rnd(-4000, 4000) + (it % 31 == 0 ? 400000 : 0), prn, n, eph.data(), pos.data(), clk.data());
        for (int i = 0; i < n; ++i) obs[i] = it % 41 == 0 ? 0.0 : rnd(1.9e7, 2.6e7);
        double p4[4], dop[5];
        int32_t def = 0;
        sgx_least_square_pos(pos.data(), obs.data(), n, 299792458.0, it & 1, p4, el.data(), az.data(), dop, &def);
        double a, b, c;
        const double X = rnd(-7e6, 7e6), Y = rnd(-7e6, 7e6), Z = rnd(-7e6, 7e6);
        sgx_cart2geo(X, Y, Z, rand() % 5, &a, &b, &c);
        sgx_cart2geo(0, 0, it % 2 ? 6.4e6 : 0.0, 4, &a, &b, &c);
        int32_t zone = 0;
        if (sgx_find_utm_zone(rnd(-90, 90), rnd(-190, 190), &zone) == SGX_OK) sgx_cart2utm(X, Y, Z, zone, &a, &b, &c);
        sgx_togeod(6378137, it % 13 == 0 ? 0.0 : 298.257223563, X, Y, Z, &a, &b, &c);
        sgx_togeod(6378137, 298.257223563, 0, 0, 0, &a, &b, &c);
        const double xs[3] = {X, Y, Z}, dx[3] = {rnd(-3e7, 3e7), rnd(-3e7, 3e7), it % 17 == 0 ? 0.0 : rnd(-3e7, 3e7)};
        double rot[3];
        sgx_e_r_corr(rnd(0, 0.1), xs, rot);
        sgx_topocent(xs, dx, &a, &b, &c);
        sgx_tropo(rnd(-1, 1), rnd(0, 3), rnd(800, 1050), rnd(250, 310), rnd(0, 100), rnd(0, 2), rnd(0, 2), rnd(0, 2), &a);
        sgx_check_t(rnd(-7e5, 7e5), &a);
        calls += 12;
    }
    printf("sanitized host run: %ld calls, no report\n", calls);
    return 0;
}
CPP
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -Iinclude "$out/driver.cpp" softgnss-python_amd/csrc/sgx_geo.cpp softgnss-python_amd/csrc/sgx_navhost.cpp \
    -o "$out/driver" -lm
ASAN_OPTIONS=detect_leaks=1 "$out/driver"
