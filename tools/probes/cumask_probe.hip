// Which CUs does a stream made by hipExtStreamCreateWithCUMask run on?  Every workgroup records (XCC id, SE id, CU id) of the CU it
// ran on; per mask the distinct CUs per XCC are printed.  Also: can a masked stream and an unmasked one overlap, and does a kernel of
// the masked stream ever land outside its mask?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <map>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_where(unsigned* out, int spin) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
}

static void run(const char* what, const std::vector<unsigned>& mask, unsigned* d_out, int nwg) {
  hipStream_t st;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (unsigned)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", what, hipGetErrorString(e)); return; }
  CHK(hipMemsetAsync(d_out, 0xFF, nwg * 8, st));
  hipLaunchKernelGGL(k_where, dim3(nwg), dim3(256), 0, st, d_out, 20000);
  CHK(hipStreamSynchronize(st));
  std::vector<unsigned> h(2 * nwg);
  CHK(hipMemcpy(h.data(), d_out, nwg * 8, hipMemcpyDeviceToHost));
  std::map<unsigned, std::set<unsigned>> per;      // xcc -> {(se, cu)}
  for (int i = 0; i < nwg; ++i) {
    const unsigned xcc = h[2 * i] & 0xF, hw = h[2 * i + 1];
    const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;      // gfx9 HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
    per[xcc].insert(se << 8 | sh << 4 | cu);
  }
  size_t tot = 0;
  printf("%s:", what);
  for (auto& kv : per) { printf("  xcc%u:%zu", kv.first, kv.second.size()); tot += kv.second.size(); }
  printf("   = %zu CUs\n", tot);
  CHK(hipStreamDestroy(st));
}

int main() {
  hipDeviceProp_t pr; CHK(hipGetDeviceProperties(&pr, 0));
  printf("multiProcessorCount %d\n", pr.multiProcessorCount);
  unsigned* d_out; const int nwg = 8192;
  CHK(hipMalloc(&d_out, nwg * 8));
  run("all 256 bits", std::vector<unsigned>(8, 0xFFFFFFFFu), d_out, nwg);
  run("bits 0..31", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0}, d_out, nwg);
  run("bits 32..255", {0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, d_out, nwg);
  run("bits 0..7", {0xFFu, 0, 0, 0, 0, 0, 0, 0}, d_out, nwg);
  run("bits 0,8,16,..", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}, d_out, nwg);
  run("bits 224..255", {0, 0, 0, 0, 0, 0, 0, 0xFFFFFFFFu}, d_out, nwg);
  return 0;
}
