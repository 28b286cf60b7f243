// conv_lab.hip -- development bench for the bf16 3x3 conv kernels (not part of the product library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I shoulder_amd/csrc -I include -o tools/conv_lab/conv_lab tools/conv_lab/conv_lab.hip
//   conv_lab H W Cin Cout nimg reps kernel
// Runs one conv layer on random bf16 data, checks it against a naive device kernel, reports TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <cmath>
#include "k_unet_bf16.h"
#include "k_conv_bf16_v2.h"
#include "k_conv_bf16_v3.h"
#include "k_conv_bf16_v4.h"
#include "lab_ablate.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

using namespace sh;

// blocked = 1: src is channel-blocked NC/32HW32 (the product kernels' activation layout); the lab variants v2..v4 keep NHWC.
// The comparison buffer `out` is always [pixel][Cout] f32; k_cmp reads the kernel output through the same layout switch.
__global__ void k_naive(const __bf16* src, const float* wf /*[9][Cin][Cout] f32 (bf16-rounded values)*/, const float* bias, float* out,
                        int H, int W, int Cin, int Cout, int nimg, int relu, int blocked) {
  size_t total = (size_t)nimg * H * W * Cout;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    int co = (int)(e % Cout);
    size_t p = e / Cout;
    int x = (int)(p % W), y = (int)((p / W) % H);
    size_t im = p / ((size_t)H * W);
    float a = bias[co];
    for (int t = 0; t < 9; ++t) {
      int gy = y + t / 3 - 1, gx = x + t % 3 - 1;
      if (gy < 0 || gy >= H || gx < 0 || gx >= W) continue;
      const __bf16* s = src + im * H * W * Cin;
      const size_t pix = (size_t)gy * W + gx;
      const float* w = wf + (size_t)t * Cin * Cout + co;
      for (int c = 0; c < Cin; ++c) a += (float)s[blocked ? act_off((size_t)H * W, pix, c) : pix * Cin + c] * w[(size_t)c * Cout];
    }
    if (relu) a = fmaxf(a, 0.0f);
    out[e] = a;
  }
}

__global__ void k_fill(__bf16* p, size_t n, unsigned seed, float scale) {
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)e * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[e] = (__bf16)(((float)(h & 0xffff) / 32768.0f - 1.0f) * scale);
  }
}

__global__ void k_round_f32(float* p, size_t n) {
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) p[e] = (float)(__bf16)p[e];
}

__global__ void k_cmp(const __bf16* a, const float* r, size_t n, float* maxerr, float* maxref, int blocked, size_t HW, int Cout) {
  float me = 0, mr = 0;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const size_t p = e / Cout, im = p / HW;
    const int co = (int)(e % Cout);
    const size_t ai = blocked ? im * HW * Cout + act_off(HW, p - im * HW, co) : e;
    me = fmaxf(me, fabsf((float)a[ai] - r[e]));
    mr = fmaxf(mr, fabsf(r[e]));
  }
  atomicMax((int*)maxerr, __float_as_int(me));
  atomicMax((int*)maxref, __float_as_int(mr));
}

int main(int argc, char** argv) {
  if (argc < 8) { fprintf(stderr, "usage: conv_lab H W Cin Cout nimg reps kernel [check]\n"); return 1; }
  int H = atoi(argv[1]), W = atoi(argv[2]), Cin = atoi(argv[3]), Cout = atoi(argv[4]), nimg = atoi(argv[5]), reps = atoi(argv[6]);
  std::string kern = argv[7];
  int check = argc > 8 ? atoi(argv[8]) : 1;
  size_t nin = (size_t)nimg * H * W * Cin, nout = (size_t)nimg * H * W * Cout, nw = (size_t)9 * Cin * Cout;
  __bf16 *src, *dst, *wpk; float *wf, *bias, *ref, *stat;
  CK(hipMalloc(&src, nin * 2)); CK(hipMalloc(&dst, nout * 2)); CK(hipMalloc(&wpk, nw * 2));
  CK(hipMalloc(&wf, nw * 4)); CK(hipMalloc(&bias, Cout * 4)); CK(hipMalloc(&stat, 8));
  k_fill<<<2048, 256>>>(src, nin, 17u, getenv("LAB_ZERO") ? 0.0f : 1.0f);
  {
    std::vector<float> hw(nw), hb(Cout);
    srand(5);
    float sc = 1.0f / sqrtf(9.0f * Cin);
    for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2 - 1) * sc;
    for (auto& v : hb) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(wf, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), Cout * 4, hipMemcpyHostToDevice));
  }
  k_round_f32<<<512, 256>>>(wf, nw);
  k_pack_w_bf16<<<1024, 256>>>(wf, wpk, 9, Cin, Cout);
  CK(hipDeviceSynchronize());

  auto launch = [&]() {
    const int tiles = (H / 16) * (W / 16);
    if (kern == "base") {
      if (Cout % 64 == 0) hipLaunchKernelGGL((k_conv_mfma_bf16<9, 4>), dim3(tiles, Cout / 64, nimg), dim3(256), 0, 0, src, (const __bf16*)nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, 1, ConvFuse{});
      else hipLaunchKernelGGL((k_conv_mfma_bf16<9, 2>), dim3(tiles, Cout / 32, nimg), dim3(256), 0, 0, src, (const __bf16*)nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, 1, ConvFuse{});
    } else if (kern.rfind("abl", 0) == 0) {
      int a = atoi(kern.c_str() + 3);
      launch_ablate(a, tiles, src, Cin, wpk, bias, dst, H, W, Cout, nimg);
    } else if (kern == "v2") {
      launch_conv_v2(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v3_000") { launch_conv_v3<0, 0, 0>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v3_100") { launch_conv_v3<1, 0, 0>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v3_110") { launch_conv_v3<1, 1, 0>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v3_111") { launch_conv_v3<1, 1, 1>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v3_101") { launch_conv_v3<1, 0, 1>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v4_000") { launch_conv_v4<0, 0, 0>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v4_100") { launch_conv_v4<1, 0, 0>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v4_110") { launch_conv_v4<1, 1, 0>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v4_111") { launch_conv_v4<1, 1, 1>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v4_101") { launch_conv_v4<1, 0, 1>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v4_010") { launch_conv_v4<0, 1, 0>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v4a1") { launch_conv_v4<1, 0, 0, 1>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v4a2") { launch_conv_v4<1, 0, 0, 2>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v4a3") { launch_conv_v4<1, 0, 0, 3>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v4a4") { launch_conv_v4<1, 0, 0, 4>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v4a5") { launch_conv_v4<1, 0, 0, 5>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v4a6") { launch_conv_v4<1, 0, 0, 6>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v4a7") { launch_conv_v4<1, 0, 0, 7>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else if (kern == "v4a8") { launch_conv_v4<1, 0, 0, 8>(src, nullptr, Cin, 0, wpk, bias, dst, H, W, Cout, nimg, 1, 0);
    } else { fprintf(stderr, "unknown kernel %s\n", kern.c_str()); exit(1); }
  };
  launch();
  CK(hipDeviceSynchronize());
  if (check) {
    CK(hipMalloc(&ref, nout * 4));
    const int blocked = (kern == "base" || kern.rfind("abl", 0) == 0) ? 1 : 0;
    k_naive<<<8192, 256>>>(src, wf, bias, ref, H, W, Cin, Cout, nimg, 1, blocked);
    CK(hipMemset(stat, 0, 8));
    k_cmp<<<1024, 256>>>(dst, ref, nout, stat, stat + 1, blocked, (size_t)H * W, Cout);
    float hs[2];
    CK(hipMemcpy(hs, stat, 8, hipMemcpyDeviceToHost));
    printf("check: max|err| %.5f  max|ref| %.3f  %s\n", hs[0], hs[1], hs[0] <= 0.02f * fmaxf(hs[1], 1.0f) ? "OK" : "MISMATCH");
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) launch();
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  double fl = 2.0 * 9 * Cin * Cout * (double)H * W * nimg;
  { long long dbg[2] = {0, 0}; CK(hipMemcpy(dbg, (const char*)v2_zero_page() + 128, 16, hipMemcpyDeviceToHost));
    if (dbg[1] > 0) printf("  core cycles %lld  wall ticks(100MHz) %lld  -> %.0f MHz, %.1f us\n", dbg[0], dbg[1], dbg[0] / (dbg[1] / 100.0), dbg[1] / 100.0); }
  printf("%s H%d W%d Cin%d Cout%d n%d: %.4f ms  %.1f TFLOP/s (%.3f of 2500)\n", kern.c_str(), H, W, Cin, Cout, nimg, ms, fl / ms * 1e-9, fl / ms * 1e-9 / 2500.0);
  return 0;
}
