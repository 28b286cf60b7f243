// unet16_pp.h -- host-side launchers of the kernels in k_unet16_pp.h (their translation unit: unet16_pp.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace sh {

// dec0b + head on the ping-pong kernel; ek: 0 = bf16, 1 = f16.  grid = persistent workgroups (one per CU).
void launch_dec0b_head_pp(int ek, unsigned grid, hipStream_t st, const unsigned short* src, const unsigned short* wgt, const float* bias,
                          const float* head_w, const float* head_b, float* logits, int H, int W, int nimg, const unsigned short* zero_page,
                          unsigned* ticket, const int* tk_tab, int ntk);

// enc0a + enc0b + pool on the ping-pong kernel (image: f32 [nimg][H][W], or raw + mm: the unscaled f64 image and its encoded bounds)
void launch_enc0_pp(int ek, unsigned grid, hipStream_t st, const float* image, const float* w0, const float* b0, const unsigned short* wgt,
                    const float* bias, unsigned short* skip, unsigned short* pooled, int H, int W, int nimg, const double* raw,
                    const unsigned long long* mm, unsigned* ticket, const int* tk_tab, int ntk);

// up0 + dec0a on the ping-pong kernel (32 x 8 tiles: tk_tab over nimg * (W / 32) * (H / 8) items)
void launch_dec0a_up_pp(int ek, unsigned grid, hipStream_t st, const unsigned short* skip, const unsigned short* low, const unsigned short* wgt,
                        const float* bias, const unsigned short* wup, const float* upb, unsigned short* dst, int H, int W, int nimg,
                        const unsigned short* zero_page, unsigned* ticket, const int* tk_tab, int ntk);

}  // namespace sh
