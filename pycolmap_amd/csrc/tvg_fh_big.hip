// tvg_fh_big.hip - the F / H kernel for pairs beyond ~38,000 matches: see tvg_e_big.hip.  Own symbols:
// tvg_fh_big_kernel, launch_tvg_fh_big.
#define AMC_TVG_BIG 1
#include "tvg_fh.hip"
