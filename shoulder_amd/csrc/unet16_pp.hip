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

void launch_enc0_pp(int ek, unsigned grid, hipStream_t st, const float* image, const float* w0, const float* b0, const unsigned short* wgt,
                    const float* bias, unsigned short* skip, unsigned short* pooled, int H, int W, int nimg, const double* raw,
                    const unsigned long long* mm, unsigned* ticket, const int* tk_tab, int ntk) {
#define E0_GO(EK, RAW) hipLaunchKernelGGL((k_enc0_pp<EK, RAW>), dim3(grid), dim3(PP_THREADS), 0, st, image, w0, b0, wgt, bias, skip, pooled, H, W, nimg, raw, mm, ticket, tk_tab, ntk)
  if (ek == 0) { if (raw) E0_GO(0, true); else E0_GO(0, false); }
  else { if (raw) E0_GO(1, true); else E0_GO(1, false); }
#undef E0_GO
}

}  // namespace sh

namespace sh {
void launch_dec0a_up_pp(int ek, unsigned grid, hipStream_t st, const unsigned short* skip, const unsigned short* low, const unsigned short* wgt,
                        const float* bias, const unsigned short* wup, const float* upb, unsigned short* dst, int H, int W, int nimg,
                        const unsigned short* zero_page, unsigned* ticket, const int* tk_tab, int ntk) {
  if (ek == 0) hipLaunchKernelGGL((k_dec0a_up_pp<0>), dim3(grid), dim3(PP_THREADS), 0, st, skip, low, wgt, bias, wup, upb, dst, H, W, nimg, zero_page, ticket, tk_tab, ntk);
  else hipLaunchKernelGGL((k_dec0a_up_pp<1>), dim3(grid), dim3(PP_THREADS), 0, st, skip, low, wgt, bias, wup, upb, dst, H, W, nimg, zero_page, ticket, tk_tab, ntk);
}
}  // namespace sh

#ifdef PP_STAMP
// diagnostic build only: the stamps of the most recent stamped launch -> out[3][256][8][PP_NSTAMP]
extern "C" int sh_lab_pp_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sh::pp_stamp), sizeof(unsigned long long) * 3 * 256 * 8 * PP_NSTAMP);
}
#endif
