/* placeholder until the denoise chain (F0-F3) lands */
#include "flx_oracle.h"
int flx_oracle_filter(const flx_frame_params *params, const flx_gbuffers *gbuffers, float *out_rgba, int threads) {
  (void)params; (void)gbuffers; (void)out_rgba; (void)threads;
  return FLX_ERR_INVALID;
}
