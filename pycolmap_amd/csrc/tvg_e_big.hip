// tvg_e_big.hip - the essential-matrix kernel for pairs whose two index arrays (4 bytes per match) do not fit a
// workgroup's LDS (more than ~38,000 matches): the same source, built with the arrays in the wave's global workspace
// (tvg_core.h idx_u16).  One wave per workgroup, up to 65,535 matches (16-bit indices).  Own symbols:
// tvg_e_big_kernel, launch_tvg_e_big.
#define AMC_TVG_BIG 1
#include "tvg_e.hip"
