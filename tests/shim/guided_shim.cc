// Test-only host build of pycolmap_amd/csrc/guided_region.h: the candidate sets of guided matching's
// candidate-generation kernel (match_guided.hip), enumerated on the CPU exactly as the kernel enumerates them -
// grid of the searched image, per point and grid row the covered cells - so that tests can check the property the
// kernel's exactness rests on: every pairing the float32 filter accepts is a candidate.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../pycolmap_amd/csrc/guided_region.h"

using namespace amc::guided;

extern "C" {

// kind 1 (F) / 2 (H); model: 9 floats (the filter's); kp1 / kp2: n x 2 float32.
// dir 0: rows = image-1 points, candidates among image 2; dir 1: rows = image-2 points, candidates among image 1.
// cand: rows x ncols bytes, 1 = listed.  Returns 1 if the pair takes the candidate kernel (grid_ok), 0 if it would go
// to the dense kernel (cand untouched), -1 if the searched image has no grid.
int shim_guided_candidates(int kind, const float* model, float max_residual, const float* kp1, uint32_t n1,
                           const float* kp2, uint32_t n2, int dir, uint8_t* cand) {
    GridGeom g1, g2;
    std::vector<uint32_t> sidx1, start1, sidx2, start2;
    if (!build_grid(kp1, n1, g1, sidx1, start1) || !build_grid(kp2, n2, g2, sidx2, start2)) return -1;
    const float box1[4] = {g1.x0, g1.y0, g1.bx1, g1.by1}, box2[4] = {g2.x0, g2.y0, g2.bx1, g2.by1};
    double bound[2], minv[9];
    if (!guided_pair_setup(kind, model, max_residual, box1, box2, bound, minv)) return 0;
    double m[9];
    for (int k = 0; k < 9; ++k) m[k] = (double)model[k];
    const GridGeom& G = dir == 0 ? g2 : g1;
    const std::vector<uint32_t>& sidx = dir == 0 ? sidx2 : sidx1;
    const std::vector<uint32_t>& start = dir == 0 ? start2 : start1;
    const float* rows = dir == 0 ? kp1 : kp2;
    const uint32_t nrows = dir == 0 ? n1 : n2, ncols = dir == 0 ? n2 : n1;
    std::memset(cand, 0, (size_t)nrows * ncols);
    for (uint32_t r = 0; r < nrows; ++r) {
        const double px = (double)rows[2 * (size_t)r], py = (double)rows[2 * (size_t)r + 1];
        for (int gy = 0; gy < kGridDim; ++gy) {
            double ylo, yhi, xa, xb;
            grid_row_interval(G.y0, G.ch, gy, ylo, yhi);
            if (!guided_row_region(kind, dir, m, minv, (double)max_residual, bound[dir], px, py, ylo, yhi,
                                   (double)G.cw + (double)G.ch, xa, xb))
                continue;
            int gx0, gx1;
            if (!grid_cells_of(xa, xb, G.x0, G.bx1, G.inv_cw, gx0, gx1)) continue;
            for (uint32_t k = start[gy * kGridDim + gx0]; k < start[gy * kGridDim + gx1 + 1]; ++k)
                cand[(size_t)r * ncols + sidx[k]] = 1;
        }
    }
    return 1;
}

}  // extern "C"
