// scan_accept.h - the match scan's accept bit without the acos table.
//
// COLMAP's one-way tests (FindBestMatchesOneWayBruteForce) on a row's (best, second) values,
// exactly as one_way_accepts (amc_internal.h) evaluates them with the host-libm table lut[d] = acosf(min(d / 512^2, 1)):
//     reject if best == 0, if lut[best] > max_distance, or if lut[best] >= max_ratio * lut[second]
// The scan only needs a SUPERSET of the accepted rows (resolve_index re-tests every kept row exactly, with the
// table), but it needs it cheaply: two table gathers per row cost 2.8 % of the scan, two device acosf 1-2 %.
// lut is non-increasing, so both tests are thresholds on `best`:
//     test 1   best >= min_best,             min_best = the smallest d with lut[d] <= max_distance   (exact)
//     test 2   best >  bcrit(second),        bcrit(s) = the largest d with lut[d] >= max_ratio * lut[s]
// and bcrit(s) ~ 2^18 cos(max_ratio acos(s 2^-18)), an analytic function of s on [0, 2^18]: a degree-8
// polynomial in t = 2 s 2^-18 - 1 (Chebyshev interpolant, power basis, float Horner with fmaf - the same
// instruction sequence on the host and on the device) follows it to 1e-7.  The kernel keeps a row iff
//     best 2^-18 > P(t) - margin
// and build() PROVES the superset property for the options at hand: it walks all 262,145 values of `second`,
// computes bcrit from the table itself and checks the kernel's own float evaluation against it.  If the proof
// fails (a table that is not monotone, a max_ratio for which the fit is poor) the kernel keeps every row
// with best >= min_best ("trivial": slower, still exact).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define AMC_SA_HD __host__ __device__ __forceinline__
#else
#define AMC_SA_HD inline
#endif

namespace amc {

constexpr int kScanAcceptDegree = 8;

struct ScanAccept {
    float coef[kScanAcceptDegree + 1];  // P(t) = sum coef[k] t^k, t = 2 x - 1, x = min(second 2^-18, 1)
    float margin;
    uint32_t min_best;   // test 1 (0xFFFFFFFF: nothing passes)
    int32_t trivial;     // 1: test 2 not evaluated by the scan (every row with best >= min_best is kept)
};

// true: the row may pass the exact tests (never false for a row one_way_accepts accepts)
AMC_SA_HD bool scan_may_accept(const ScanAccept& a, uint32_t best_v, uint32_t second_v) {
    if (best_v < a.min_best || best_v == 0u) return false;
    if (a.trivial) return true;
    const float x = fminf((float)second_v * 0x1p-18f, 1.0f);
    const float y = (float)best_v * 0x1p-18f;
    const float t = fmaf(2.0f, x, -1.0f);
    float p = a.coef[kScanAcceptDegree];
#pragma unroll
    for (int k = kScanAcceptDegree - 1; k >= 0; --k) p = fmaf(p, t, a.coef[k]);
    return y > p - a.margin;
}

// (host function) lut: kAcosLutSize floats, lut[d] = acosf(min(d 2^-18, 1)) from the host libm
inline ScanAccept build_scan_accept(const float* lut, uint32_t lut_size, float max_ratio, float max_distance) {
    ScanAccept a{};
    a.margin = 16.0f * 0x1p-18f;
    a.trivial = 1;
    a.min_best = 0xFFFFFFFFu;
    const uint32_t last = lut_size - 1;  // 262144
    bool monotone = true;
    for (uint32_t d = 1; d <= last; ++d)
        if (lut[d] > lut[d - 1]) monotone = false;
    if (!monotone) {  // thresholds do not describe the tests: keep everything with best > 0
        a.min_best = 1;
        return a;
    }
    // test 1: lut[min(best, last)] <= max_distance.  (NaN max_distance: nothing is rejected by `>`.)
    if (!(max_distance == max_distance)) {
        a.min_best = 1;
    } else {
        uint32_t lo = 0, hi = last + 1;  // first d with lut[d] <= max_distance, in [0, last + 1]
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (lut[mid] <= max_distance) hi = mid; else lo = mid + 1;
        }
        a.min_best = lo > last ? 0xFFFFFFFFu : (lo < 1 ? 1u : lo);
    }
    // the interpolant of cos(max_ratio acos x) on x in [0, 1]
    if (!(max_ratio == max_ratio) || !isfinite(max_ratio)) return a;
    constexpr int D = kScanAcceptDegree;
    const double pi = 3.14159265358979323846;
    double fx[D + 1], cheb[D + 1];
    for (int k = 0; k <= D; ++k) {
        const double t = cos(pi * (k + 0.5) / (D + 1));
        fx[k] = cos((double)max_ratio * acos(0.5 * (t + 1.0)));
    }
    for (int j = 0; j <= D; ++j) {
        double s = 0.0;
        for (int k = 0; k <= D; ++k) s += fx[k] * cos(pi * j * (k + 0.5) / (D + 1));
        cheb[j] = s * 2.0 / (D + 1);
    }
    cheb[0] *= 0.5;
    // Chebyshev -> power basis: T0 = 1, T1 = t, T(n+1) = 2 t T(n) - T(n-1)
    double Tm[D + 1] = {1.0}, Tc[D + 1] = {0.0, 1.0}, mono[D + 1] = {0.0};
    for (int k = 0; k <= D; ++k) mono[k] = cheb[0] * Tm[k] + (D >= 1 ? cheb[1] * Tc[k] : 0.0);
    for (int n = 2; n <= D; ++n) {
        double Tn[D + 1];
        for (int k = 0; k <= D; ++k) Tn[k] = (k > 0 ? 2.0 * Tc[k - 1] : 0.0) - Tm[k];
        for (int k = 0; k <= D; ++k) {
            mono[k] += cheb[n] * Tn[k];
            Tm[k] = Tc[k];
            Tc[k] = Tn[k];
        }
    }
    for (int k = 0; k <= D; ++k) a.coef[k] = (float)mono[k];
    // The proof.  For every second value s: the exact test 2 rejects best b iff lut[min(b, last)] >= max_ratio * lut[min(s, last)]
    // (one float multiply).  lut is non-increasing, so the rejected b are a prefix 0..bcrit(s); bcrit is non-decreasing
    // in s (the right-hand side shrinks).  Every b > bcrit(s) must be kept by the kernel's evaluation; that evaluation
    // is increasing in b, so b = bcrit(s) + 1 decides.  (b > last all behave like b = last in the exact test.)
    a.trivial = 0;
    uint32_t bcrit_plus1 = 0;  // smallest b the exact test 2 accepts, for the current s (two pointers)
    for (uint32_t s = 0; s <= last; ++s) {
        const float rhs = max_ratio * lut[s];
        while (bcrit_plus1 <= last && lut[bcrit_plus1] >= rhs) ++bcrit_plus1;
        if (bcrit_plus1 > last) break;  // nothing passes test 2 from here on: nothing to keep
        ScanAccept probe = a;
        probe.min_best = 0;
        if (!scan_may_accept(probe, bcrit_plus1 == 0 ? 1u : bcrit_plus1, s)) {
            a.trivial = 1;  // the fit is not a bound for these options
            break;
        }
    }
    // second values above 2^18 clamp to x = 1 on both sides: covered by s = last
    return a;
}

}  // namespace amc
