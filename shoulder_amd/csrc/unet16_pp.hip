// unet16_pp.hip -- translation unit of the ping-pong level-0 kernels (k_unet16_pp.h): compiled on its own, so that a change to
// these kernels does not recompile the rest of the library.
#include "k_unet16_pp.h"
#include "unet16_pp.h"

namespace sh {

void launch_dec0b_head_pp(int ek, unsigned grid, hipStream_t st, const unsigned short* src, const unsigned short* wgt, const float* bias,
                          const float* head_w, const float* head_b, float* logits, int H, int W, int nimg, const unsigned short* zero_page,
                          unsigned* ticket, const int* tk_tab, int ntk) {
  if (ek == 0) hipLaunchKernelGGL((k_dec0b_head_pp<0>), dim3(grid), dim3(PP_THREADS), 0, st, src, wgt, bias, head_w, head_b, logits, H, W, nimg, zero_page, ticket, tk_tab, ntk);
  else hipLaunchKernelGGL((k_dec0b_head_pp<1>), dim3(grid), dim3(PP_THREADS), 0, st, src, wgt, bias, head_w, head_b, logits, H, W, nimg, zero_page, ticket, tk_tab, ntk);
}

}  // namespace sh

#ifdef PP_STAMP
// diagnostic build only: the stamps of the most recent stamped launch -> out[256][8][PP_NSTAMP]
extern "C" int sh_lab_pp_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sh::pp_stamp), sizeof(unsigned long long) * 256 * 8 * PP_NSTAMP);
}
#endif
