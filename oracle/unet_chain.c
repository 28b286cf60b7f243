/* unet_chain.c -- oracle (TEST INFRASTRUCTURE): float32 fma-chain restatement of the UNet layers.
 *
 * Every output element is one fmaf chain that starts from the bias and runs over (16-channel
 * chunk, tap, channel within the chunk) in order (for Cin <= 16: (tap, cin)), exactly the accumulation order of the product's f32 MFMA kernels
 * (shoulder_amd/csrc/k_unet.h), so the two agree bit for bit.  The network stands in for the
 * reference's missing unetcrf_anp.onnx (src/shoulder/humerus/anatomic_neck.py:62-76); see
 * oracle/unet.py.  Layout: activations NHWC, conv weights [ky][kx][cin][cout].
 * Build: make -C oracle   (gcc -O3 -mfma: fmaf() becomes a hardware fused multiply-add).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

void oc_conv3x3(const float* in, const float* w, const float* b, float* out, int H, int W, int Cin, int Cout, int relu) {
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; ++y) {
    float* acc = (float*)malloc((size_t)W * Cout * sizeof(float));
    for (int x = 0; x < W; ++x) memcpy(acc + (size_t)x * Cout, b, Cout * sizeof(float));
    /* chain order: 16-channel chunk (outer), tap, channel within the chunk (inner) */
    for (int c0 = 0; c0 < Cin; c0 += 16) {
      int c1 = c0 + 16 < Cin ? c0 + 16 : Cin;
      for (int dy = 0; dy < 3; ++dy) {
        int gy = y + dy - 1;
        if (gy < 0 || gy >= H) continue;        /* zero padding: fmaf(0, w, acc) == acc */
        for (int dx = 0; dx < 3; ++dx)
          for (int ci = c0; ci < c1; ++ci) {
            const float* wr = w + (((size_t)dy * 3 + dx) * Cin + ci) * Cout;
            for (int x = 0; x < W; ++x) {
              int gx = x + dx - 1;
              if (gx < 0 || gx >= W) continue;
              float v = in[((size_t)gy * W + gx) * Cin + ci];
              float* a = acc + (size_t)x * Cout;
              for (int co = 0; co < Cout; ++co) a[co] = fmaf(v, wr[co], a[co]);
            }
          }
      }
    }
    float* o = out + (size_t)y * W * Cout;
    for (size_t i = 0; i < (size_t)W * Cout; ++i) o[i] = relu ? fmaxf(acc[i], 0.0f) : acc[i];
    free(acc);
  }
}

void oc_upconv2x2(const float* in, const float* w, const float* b, float* out, int H, int W, int Cin, int Cout) {
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x)
      for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
          float* o = out + (((size_t)(2 * y + dy)) * (2 * W) + (2 * x + dx)) * Cout;
          memcpy(o, b, Cout * sizeof(float));
          for (int ci = 0; ci < Cin; ++ci) {
            float v = in[((size_t)y * W + x) * Cin + ci];
            const float* wr = w + (((size_t)dy * 2 + dx) * Cin + ci) * Cout;
            for (int co = 0; co < Cout; ++co) o[co] = fmaf(v, wr[co], o[co]);
          }
        }
}

void oc_head(const float* in, const float* w, float b, float* out, int H, int W, int C) {
  for (size_t p = 0; p < (size_t)H * W; ++p) {
    float a = b;
    for (int c = 0; c < C; ++c) a = fmaf(in[p * C + c], w[c], a);
    out[p] = a;
  }
}
