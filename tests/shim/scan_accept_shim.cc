// Host build of pycolmap_amd/csrc/scan_accept.h for tests/test_scan_accept_cpu.py: the thresholds the match scan's
// accept bit uses, and the kernel's own evaluation of them (the same float instruction sequence).
#include "../../pycolmap_amd/csrc/scan_accept.h"

extern "C" {
// out: coef[9], margin, then (as uint32 bit patterns in the float slots) min_best, trivial
void sa_build(const float* lut, unsigned lut_size, float max_ratio, float max_distance, amc::ScanAccept* out) {
    *out = amc::build_scan_accept(lut, lut_size, max_ratio, max_distance);
}
void sa_eval(const amc::ScanAccept* a, const unsigned* best, const unsigned* second, unsigned n, unsigned char* keep) {
    for (unsigned i = 0; i < n; ++i) keep[i] = amc::scan_may_accept(*a, best[i], second[i]) ? 1 : 0;
}
unsigned sa_sizeof() { return (unsigned)sizeof(amc::ScanAccept); }
}
