// Host side of the navigation stage (reference postNavigation.py:27-72, 125-138, 443-631; ephemeris.py): the
// candidate logic of findPreambles on the device's correlation output, navPartyChk, the 20-ms bit integration,
// ephemeris decoding and calculatePseudoranges.  Scalar code with no HIP dependency, so it also builds under
// the CPU sanitizers (tools/sanitize_host.sh).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "sgx.h"

void sgx_set_error(const char* fmt, ...);

#define SGX_CHECK_ARG(cond)                                                       \
    do {                                                                          \
        if (!(cond)) {                                                            \
            sgx_set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__); \
            return SGX_E_ARG;                                                     \
        }                                                                         \
    } while (0)

// numpy's reduction order for reshape(20, -1, order='F').sum(0): the 20 contiguous values of a column go through
// the unrolled pairwise sum (8 accumulators over two rounds, tree combine, then the last 4 in sequence)
static inline double sum20(const double* a) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j] + a[j + 8];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (int i = 16; i < 20; ++i) res += a[i];
    return res;
}

// postNavigation.py:443-521 on +-1 values; flips d1..d24 in place like the reference
static int parity_status(double* ndat) {
    if (ndat[1] != 1)
        for (int i = 2; i < 26; ++i) ndat[i] *= -1;
    static const int rows[6][17] = {{0, 2, 3, 4, 6, 7, 11, 12, 13, 14, 15, 18, 19, 21, 24, -1},
                                    {1, 3, 4, 5, 7, 8, 12, 13, 14, 15, 16, 19, 20, 22, 25, -1},
                                    {0, 2, 4, 5, 6, 8, 9, 13, 14, 15, 16, 17, 20, 21, 23, -1},
                                    {1, 3, 5, 6, 7, 9, 10, 14, 15, 16, 17, 18, 21, 22, 24, -1},
                                    {1, 2, 4, 6, 7, 8, 10, 11, 15, 16, 17, 18, 19, 22, 23, 25, -1},
                                    {0, 4, 6, 7, 9, 10, 11, 12, 14, 16, 20, 23, 24, 25, -1}};
    int ok = 0;
    for (int p = 0; p < 6; ++p) {
        double v = 1.0;
        for (int j = 0; rows[p][j] >= 0; ++j) v *= ndat[rows[p][j]];
        ok += (v == ndat[26 + p]);
    }
    return ok == 6 ? (int)(-1 * ndat[1]) : 0;
}

extern "C" int sgx_nav_parity_check(double* ndat32, int32_t* status) {
    SGX_CHECK_ARG(ndat32 && status);
    *status = parity_status(ndat32);
    return SGX_OK;
}

// postNavigation.py:125-138: 20-ms sums from one bit before the subframe start, 1500 bits after it
extern "C" int sgx_nav_bits(const double* I_P_row, int32_t ms, int32_t subFrameStart, uint8_t* bits,
                            int32_t* n_bits) {
    SGX_CHECK_ARG(I_P_row && bits && n_bits && ms >= 1 && subFrameStart >= 0);
    const int lo = subFrameStart - 20, hi = subFrameStart + 1500 * 20;
    const int a0 = lo < 0 ? (ms + lo > 0 ? ms + lo : 0) : (lo < ms ? lo : ms);
    const int a1 = hi < ms ? hi : ms;
    const int len = a1 > a0 ? a1 - a0 : 0;
    if (len % 20 != 0) {
        sgx_set_error("ValueError: cannot reshape array of size %d into shape (20,newaxis) "
                      "(subframe start %d of a %d ms record, reference postNavigation.py:128-131)",
                      len, subFrameStart, ms);
        return SGX_E_RANGE;
    }
    *n_bits = len / 20;
    for (int w = 0; w < len / 20; ++w) bits[w] = sum20(I_P_row + a0 + 20 * w) > 0 ? 1 : 0;
    return SGX_OK;
}

// postNavigation.py:27-72: relative pseudoranges (metres) at one measurement point per channel
extern "C" int sgx_pseudoranges(const double* absoluteSample, int32_t n_rows, int32_t ms, const double* msOfTheSignal,
                                const int32_t* channelList, int32_t n_list, int32_t numberOfChannels,
                                int64_t samplesPerCode, double startOffset, double c_mps, double* pseudoranges) {
    SGX_CHECK_ARG(absoluteSample && msOfTheSignal && pseudoranges && (channelList || n_list == 0));
    SGX_CHECK_ARG(n_rows >= 0 && ms >= 1 && n_list >= 0 && numberOfChannels >= 1 && samplesPerCode >= 1);
    for (int i = 0; i < numberOfChannels; ++i) pseudoranges[i] = INFINITY;   // travelTime = Inf * ones(...)
    for (int k = 0; k < n_list; ++k) {
        const int ch = channelList[k];
        long long idx = (long long)msOfTheSignal[ch >= 0 && ch < numberOfChannels ? ch : 0];   // np.int(): truncation
        if (ch < 0 || ch >= numberOfChannels || ch >= n_rows) {
            sgx_set_error("IndexError: channel %d outside the %d tracked / %d configured channels", ch, n_rows,
                          numberOfChannels);
            return SGX_E_RANGE;
        }
        if (idx < 0) idx += ms;                                               // Python's negative index
        if (idx < 0 || idx >= ms) {
            sgx_set_error("IndexError: measurement point %lld outside the %d ms of channel %d",
                          (long long)msOfTheSignal[ch], ms, ch);
            return SGX_E_RANGE;
        }
        pseudoranges[ch] = absoluteSample[(size_t)ch * (size_t)ms + (size_t)idx] / (double)samplesPerCode;
    }
    double mn = INFINITY;
    for (int i = 0; i < numberOfChannels; ++i) mn = pseudoranges[i] < mn ? pseudoranges[i] : mn;
    const double minimum = floor(mn);
    for (int i = 0; i < numberOfChannels; ++i)
        pseudoranges[i] = ((pseudoranges[i] - minimum) + startOffset) * c_mps / 1000;   // left to right, as written
    return SGX_OK;
}

// ---- ephemeris.py:60-195: clock and orbit parameters + TOW from five consecutive subframes -------------------------
namespace {
struct BitView {
    const uint8_t* b;   // 300 polarity-corrected bits of one subframe
    // unsigned value of bits [a0, a1) followed by bits [b0, b1) (Python slices; the second may be empty)
    unsigned long long u(int a0, int a1, int b0 = 0, int b1 = 0) const {
        unsigned long long v = 0;
        for (int i = a0; i < a1; ++i) v = (v << 1) | b[i];
        for (int i = b0; i < b1; ++i) v = (v << 1) | b[i];
        return v;
    }
    // twosComp2dec of the same bits (ephemeris.py:7-25)
    long long s(int a0, int a1, int b0 = 0, int b1 = 0) const {
        const int len = (a1 - a0) + (b1 - b0);
        long long v = (long long)u(a0, a1, b0, b1);
        if (b[a0]) v -= 1ll << len;
        return v;
    }
};
}   // namespace

extern "C" int sgx_ephemeris(const uint8_t* bits, int32_t n_bits, uint8_t d30star, double* eph, int64_t* tow) {
    SGX_CHECK_ARG(bits && eph && tow);
    if (n_bits < 1500) {
        sgx_set_error("TypeError: The parameter BITS must contain 1500 bits!");
        return SGX_E_ARG;
    }
    const double gpsPi = 3.1415926535898;                      // ephemeris.py:94
    const double p2m5 = ldexp(1.0, -5), p2m19 = ldexp(1.0, -19), p2m29 = ldexp(1.0, -29), p2m31 = ldexp(1.0, -31),
                 p2m33 = ldexp(1.0, -33), p2m43 = ldexp(1.0, -43), p2m55 = ldexp(1.0, -55);
    bool have[4] = {false, false, false, false};
    uint8_t sf[300];
    uint8_t d30 = d30star ? 1 : 0;
    for (int i = 0; i < 5; ++i) {
        for (int j = 0; j < 10; ++j) {                         // checkPhase (ephemeris.py:30-57): D30* = 1 inverts d1..d24
            for (int k = 0; k < 30; ++k) {
                const uint8_t v = bits[300 * i + 30 * j + k] ? 1 : 0;
                sf[30 * j + k] = (k < 24 && d30) ? (uint8_t)(1 - v) : v;
            }
            d30 = sf[30 * j + 29];
        }
        const BitView w{sf};
        const int id = (int)w.u(49, 52);
        if (id == 1) {
            have[1] = true;
            eph[0] = (double)(w.u(60, 70) + 1024);             // weekNumber
            eph[1] = (double)w.u(72, 76);                      // accuracy
            eph[2] = (double)w.u(76, 82);                      // health
            eph[3] = (double)w.s(195, 204) * p2m31;            // T_GD (9 bits as the reference slices them)
            eph[4] = (double)w.u(82, 84, 196, 204);            // IODC
            eph[5] = (double)(w.u(218, 234) * 16);             // t_oc
            eph[6] = (double)w.s(240, 248) * p2m55;            // a_f2
            eph[7] = (double)w.s(248, 264) * p2m43;            // a_f1
            eph[8] = (double)w.s(270, 292) * p2m31;            // a_f0
        } else if (id == 2) {
            have[2] = true;
            eph[9] = (double)w.u(60, 68);                      // IODE_sf2
            eph[10] = (double)w.s(68, 84) * p2m5;              // C_rs
            eph[11] = (double)w.s(90, 106) * p2m43 * gpsPi;    // deltan
            eph[12] = (double)w.s(106, 114, 120, 144) * p2m31 * gpsPi;   // M_0
            eph[13] = (double)w.s(150, 166) * p2m29;           // C_uc
            eph[14] = (double)w.u(166, 174, 180, 204) * p2m33; // e
            eph[15] = (double)w.s(210, 226) * p2m29;           // C_us
            eph[16] = (double)w.u(226, 234, 240, 264) * p2m19; // sqrtA
            eph[17] = (double)(w.u(270, 286) * 16);            // t_oe
        } else if (id == 3) {
            have[3] = true;
            eph[18] = (double)w.s(60, 76) * p2m29;             // C_ic
            eph[19] = (double)w.s(76, 84, 90, 114) * p2m31 * gpsPi;      // omega_0
            eph[20] = (double)w.s(120, 136) * p2m29;           // C_is
            eph[21] = (double)w.s(136, 144, 150, 174) * p2m31 * gpsPi;   // i_0
            eph[22] = (double)w.s(180, 196) * p2m5;            // C_rc
            eph[23] = (double)w.s(196, 204, 210, 234) * p2m31 * gpsPi;   // omega
            eph[24] = (double)w.s(240, 264) * p2m43 * gpsPi;   // omegaDot
            eph[25] = (double)w.u(270, 278);                   // IODE_sf3
            eph[26] = (double)w.s(278, 292) * p2m43 * gpsPi;   // iDot
        }
        if (i == 4) *tow = (int64_t)w.u(30, 47) * 6 - 30;      // TOW of the first subframe of the block
    }
    if (!have[1] || !have[2] || !have[3]) {
        // the reference then reads a local variable that was never assigned (ephemeris.py:190-193)
        sgx_set_error("UnboundLocalError: subframe %d is not among the five decoded subframes",
                      !have[1] ? 1 : (!have[2] ? 2 : 3));
        return SGX_E_RANGE;
    }
    return SGX_OK;
}

// findPreambles after the correlation (postNavigation.py:583-631): corr[ch][t] = sign correlation with the preamble
int sgx_nav_select(const double* I_P, const short* corr, int32_t n_ch, int32_t ms, int32_t search_start,
                   int32_t* firstSubFrame) {
    for (int ch = 0; ch < n_ch; ++ch) {
        firstSubFrame[ch] = 0;
        const double* ip = I_P + (size_t)ch * ms;
        const short* cc = corr + (size_t)ch * ms;
        std::vector<int> index;
        for (int t = search_start; t < ms; ++t)
            if (abs((int)cc[t]) > 153) index.push_back(t);   // postNavigation.py:583
        for (size_t i = 0; i < index.size(); ++i) {
            bool partner = false;
            for (size_t j = 0; j < index.size() && !partner; ++j) partner = (index[j] - index[i] == 6000);
            if (!partner) continue;
            // Python slice I_P[index-40 : index+1200]: a negative start counts from the end, the stop is clipped
            const int lo = index[i] - 40, hi = index[i] + 20 * 60;
            const int a0 = lo < 0 ? (ms + lo > 0 ? ms + lo : 0) : lo;
            const int a1 = hi < ms ? hi : ms;
            const int len = a1 > a0 ? a1 - a0 : 0;
            if (len % 20 != 0) {   // reshape(20, -1) of a slice cut short by the end of the record
                sgx_set_error("ValueError: cannot reshape array of size %d into shape (20,newaxis) "
                              "(preamble candidate at %d ms, reference postNavigation.py:600-602)", len, index[i]);
                return SGX_E_RANGE;
            }
            const int words = len / 20;
            if (words < 32) {      // navPartyChk indexes past the end of a word shorter than 32 bits
                sgx_set_error("IndexError: %d-bit slice around the preamble candidate at %d ms "
                              "(reference postNavigation.py:443-521, 615)", words, index[i]);
                return SGX_E_RANGE;
            }
            double bits[62];
            for (int w = 0; w < 62 && w < words; ++w) {
                bits[w] = sum20(ip + a0 + 20 * w) > 0 ? 1.0 : -1.0;   // reshape(20,-1,'F').sum(0)
            }
            double w1[32], w2[32];
            memcpy(w1, bits, sizeof(w1));
            // the reference checks views of ONE array: the first check's in-place flip of bits[2:26] is seen
            // by the second check only where the views overlap (bits 30, 31)
            const int s1 = parity_status(w1);
            memcpy(bits, w1, sizeof(w1));
            if (s1 != 0 && words < 62) {
                sgx_set_error("IndexError: %d-bit slice, second word incomplete, candidate at %d ms "
                              "(reference postNavigation.py:615)", words, index[i]);
                return SGX_E_RANGE;
            }
            memcpy(w2, bits + 30, sizeof(w2));
            const int s2 = (s1 != 0) ? parity_status(w2) : 0;
            if (s1 != 0 && s2 != 0) {
                firstSubFrame[ch] = index[i];
                break;
            }
        }
    }
    return SGX_OK;
}

// ---- postNavigation.py:150-290: the loop over measurement epochs ------------------------------------------------------
extern "C" int sgx_post_navigate(const double* absoluteSample, int32_t n_rows, int32_t ms, const int32_t* prn_of_row,
                                 const double* subFrameStart, const int32_t* ready, int32_t n_ready,
                                 int32_t numberOfChannels, const double* eph, int64_t tow, int64_t samplesPerCode,
                                 double startOffset, double c_mps, double navSolPeriod, double elevationMask,
                                 int32_t useTropCorr, int32_t n_meas, double* chan_PRN, double* chan_el, double* chan_az,
                                 double* chan_rawP, double* chan_correctedP, double* DOP, double* sol, int32_t* utmZone,
                                 int32_t* not_enough) {
    SGX_CHECK_ARG(absoluteSample && prn_of_row && subFrameStart && (ready || n_ready == 0) && eph && chan_PRN && chan_el &&
                  chan_az && chan_rawP && chan_correctedP && DOP && sol && utmZone && not_enough);
    SGX_CHECK_ARG(n_rows >= 0 && ms >= 1 && n_ready >= 0 && numberOfChannels >= 1 && numberOfChannels <= 4096 && n_meas >= 0);
    const int nch = numberOfChannels, W = 64;
    if (n_meas > W) {
        sgx_set_error("IndexError: index %d is out of bounds for axis 1 with size %d (measurement epochs)", W, W);
        return SGX_E_RANGE;
    }
    for (int i = 0; i < nch * W; ++i) {
        chan_PRN[i] = 0.0;
        chan_el[i] = chan_az[i] = chan_rawP[i] = chan_correctedP[i] = NAN;
    }
    for (int i = 0; i < 5 * W; ++i) DOP[i] = 0.0;
    for (int i = 0; i < 10 * W; ++i) sol[i] = NAN;
    for (int i = 0; i < W; ++i) not_enough[i] = 0;
    *utmZone = 0;
    std::vector<double> satElev((size_t)nch, INFINITY), when((size_t)nch), raw((size_t)nch);
    std::vector<int32_t> active, prns;
    std::vector<double> satPositions, satClk, obs, el, az;
    double transmitTime = (double)tow;
    for (int m = 0; m < n_meas; ++m) {
        // intersect1d((satElev >= elevationMask).nonzero()[0], readyChnList): sorted, unique
        active.clear();
        for (int ch = 0; ch < nch; ++ch) {
            if (!(satElev[(size_t)ch] >= elevationMask)) continue;
            bool in = false;
            for (int k = 0; k < n_ready; ++k) in = in || ready[k] == ch;
            if (in) active.push_back(ch);
        }
        const int na = (int)active.size();
        prns.resize((size_t)na);
        for (int k = 0; k < na; ++k) {
            if (active[(size_t)k] >= n_rows) {
                sgx_set_error("IndexError: channel %d outside the %d tracked channels", active[(size_t)k], n_rows);
                return SGX_E_RANGE;
            }
            prns[(size_t)k] = prn_of_row[active[(size_t)k]];
            chan_PRN[(size_t)active[(size_t)k] * W + m] = (double)prns[(size_t)k];
        }
        for (int ch = 0; ch < nch; ++ch) when[(size_t)ch] = subFrameStart[ch] + navSolPeriod * m;
        int rc = sgx_pseudoranges(absoluteSample, n_rows, ms, when.data(), active.data(), na, nch, samplesPerCode,
                                  startOffset, c_mps, raw.data());
        if (rc != SGX_OK) return rc;
        for (int ch = 0; ch < nch; ++ch) chan_rawP[(size_t)ch * W + m] = raw[(size_t)ch];
        satPositions.assign((size_t)3 * (size_t)(na > 0 ? na : 1), 0.0);
        satClk.assign((size_t)(na > 0 ? na : 1), 0.0);
        rc = sgx_satpos(transmitTime, prns.data(), na, eph, satPositions.data(), satClk.data());
        if (rc != SGX_OK) return rc;
        if (na > 3) {
            obs.resize((size_t)na);
            el.assign((size_t)na, 0.0);
            az.assign((size_t)na, 0.0);
            for (int k = 0; k < na; ++k) obs[(size_t)k] = raw[(size_t)active[(size_t)k]] + satClk[(size_t)k] * c_mps;
            double pos[4] = {0, 0, 0, 0}, dop[5] = {0, 0, 0, 0, 0};
            int32_t deficient = 0;
            rc = sgx_least_square_pos(satPositions.data(), obs.data(), na, c_mps, useTropCorr, pos, el.data(), az.data(), dop,
                                      &deficient);
            if (rc != SGX_OK) return rc;
            if (deficient) pos[0] = pos[1] = pos[2] = pos[3] = 0.0;   // (the reference's early return: a zero position)
            for (int k = 0; k < na; ++k) {
                chan_el[(size_t)active[(size_t)k] * W + m] = el[(size_t)k];
                chan_az[(size_t)active[(size_t)k] * W + m] = az[(size_t)k];
            }
            for (int i = 0; i < 5; ++i) DOP[i * W + m] = dop[i];
            for (int i = 0; i < 4; ++i) sol[i * W + m] = pos[i];
            for (int ch = 0; ch < nch; ++ch) satElev[(size_t)ch] = chan_el[(size_t)ch * W + m];   // (NaN: not in use)
            for (int k = 0; k < na; ++k)
                chan_correctedP[(size_t)active[(size_t)k] * W + m] = (raw[(size_t)active[(size_t)k]] + satClk[(size_t)k] * c_mps) + pos[3];
            double lat, lon, h, E, N, U;
            rc = sgx_cart2geo(pos[0], pos[1], pos[2], 4, &lat, &lon, &h);
            if (rc != SGX_OK) return rc;
            sol[4 * W + m] = lat;
            sol[5 * W + m] = lon;
            sol[6 * W + m] = h;
            rc = sgx_find_utm_zone(lat, lon, utmZone);
            if (rc != SGX_OK) return rc;
            rc = sgx_cart2utm(pos[0], pos[1], pos[2], *utmZone, &E, &N, &U);
            if (rc != SGX_OK) return rc;
            sol[7 * W + m] = E;
            sol[8 * W + m] = N;
            sol[9 * W + m] = U;
        } else {
            not_enough[m] = 1;
            for (int i = 0; i < 10; ++i) sol[i * W + m] = NAN;
            for (int i = 0; i < 5; ++i) DOP[i * W + m] = 0.0;
            for (int k = 0; k < na; ++k) {
                chan_az[(size_t)active[(size_t)k] * W + m] = NAN;
                chan_el[(size_t)active[(size_t)k] * W + m] = NAN;
            }
        }
        transmitTime += navSolPeriod / 1000;
    }
    return SGX_OK;
}
