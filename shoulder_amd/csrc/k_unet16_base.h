// k_unet16_base.h -- element types, fragment layout and LDS-DMA helpers shared by the 16-bit UNet kernels.  No kernels in here: the
// translation units of the library (shoulder_hip.hip, unet16_pp.hip) both include it.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace sh {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Activation layout: channel-blocked "NC/32HW32" -- per image, 32-channel planes of [H][W][32] bf16 (64 B per pixel and
// plane).  A 32-channel chunk of a tile row is then one contiguous run (full 128-B lines for the staging loads and the
// LDS-DMA), where the plain NHWC form made every staging step touch half of each pixel's line and fetched most lines twice
// (PMC: 1.34 GB per launch against 0.65 GB algorithmic).  A 32-channel tensor is plain NHWC either way.
__device__ __host__ inline size_t act_off(size_t HW, size_t pix, int c) { return ((size_t)(c >> 5) * HW + pix) * 32 + (size_t)(c & 31); }

// Kernels are templates on an element-KIND integer (EK: 0 = __bf16, 1 = _Float16) and take their 16-bit tensors as
// `const u16*`: with the element TYPE in the template arguments or the parameter list (mangled DF16b / DF16_) rocprofv3's
// demangler prints the symbols half mangled, and profiles/ is keyed by kernel name.
typedef unsigned short u16;
template <int EK> struct EKT;
template <> struct EKT<0> { typedef __bf16 type; };
template <> struct EKT<1> { typedef _Float16 type; };

// element-type trait: vector types and the 16x16x32 MFMA (A = 8 k-values of 16 rows, B = 8 k-values of 16 columns)
template <typename ET> struct E16;
template <> struct E16<__bf16> {
  typedef __bf16 v8 __attribute__((ext_vector_type(8)));
  typedef __bf16 v4 __attribute__((ext_vector_type(4)));
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  static __device__ inline f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct E16<_Float16> {
  typedef _Float16 v8 __attribute__((ext_vector_type(8)));
  typedef _Float16 v4 __attribute__((ext_vector_type(4)));
  typedef _Float16 v2 __attribute__((ext_vector_type(2)));
  static __device__ inline f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

#define UB_PSTR 32      // bf16 elements per LDS row: 32 channels = 64 B = four 16-B slots, unpadded
// XOR swizzle of the 16-B slot inside a row: slot' = slot ^ ((row >> 1) & 2).  With it the 16 lanes of every
// ds_read_b128 lane group (rows p0 + (lane & 15), slot lane >> 4) hit 16 distinct slots of the 256-B bank row for
// every p0 (checked exhaustively), so fragment reads are bank-conflict free without padding.
#define UB_OFF(row, slot) ((row) * UB_PSTR + (((slot) ^ (((row) >> 1) & 2)) << 3))

#define UD_PW 36                            // halo-tile pitch in pixels (34 used)
#define UD_INROWS (18 * UD_PW)              // 648 halo rows of 64 B: an 18 x 34 halo tile around 16 rows x 32 pixels

typedef const __attribute__((address_space(1))) void* ud_gptr;
typedef __attribute__((address_space(3))) void* ud_lptr;

template <typename ET, typename V4>
__device__ inline void ud_store8(ET* p, V4 v) {      // exactly one vector-memory instruction (counted by s_waitcnt vmcnt)
  asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(__builtin_bit_cast(unsigned long long, v)) : "memory");
}
template <typename ET, typename V8>
__device__ inline void ud_store16(ET* p, V8 v) {      // one dwordx4 store: 8 consecutive channels of a pixel
  // s_nop 1 inside the string: hipcc pads nothing around an asm statement, and a store of more than 64 bits must not have
  // its data registers overwritten in the next two issue slots (cdna_hip_programming.md 5.7; without it the later lanes
  // of the wave stored the NEXT tile's values)
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(p), "v"(__builtin_bit_cast(u32x4, v)) : "memory");
}
__device__ inline void ud_store4(float* p, float v) {
  asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
}

// One LDS-DMA piece (64 lanes x 16 bytes -> 1 KB of LDS at lds_dst + 16 lane) as inline assembly.  Through
// __builtin_amdgcn_global_load_lds hipcc knows that vector memory writes LDS, and its wait-count pass then puts s_waitcnt vmcnt(0)
// in front of the first LDS read behind ANY such load (it cannot tell the buffers apart): a wave drains the tile it has just issued
// before it reads a fragment of the tile that landed long ago.  As assembly the transfers are the kernel's own business: its
// counted waits and barriers order them; the `memory` clobber keeps the compiler's loads and stores on their side of every wait.
__device__ inline void ud_dma16(unsigned lds_dst /*wave-uniform LDS byte address*/, const void* p) {
  // (readfirstlane: the "s" constraint alone does not make hipcc keep a value it takes for divergent in a scalar register)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "v"(p) : "memory");
}
__device__ inline void ud_dma16_s(unsigned lds_dst /*wave-uniform*/, unsigned voff, const void* sbase /*wave-uniform*/) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "v"(voff), "s"(sbase) : "memory");
}

// max of two packed pairs of non-negative 16-bit floats (they order like int16), and the same against the lane that holds the
// neighbouring pixel (lane ^ 1): as assembly -- written with __builtin_amdgcn_mov_dpp in a loop over the four dwords of a pixel,
// hipcc (ROCm 7.2) emitted ONE swap and stored its result four times
__device__ inline unsigned pp_pkmax(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline unsigned pp_pkmax_lane1(unsigned a) {
  unsigned t, r;
  asm("s_nop 1\n\tv_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_pk_max_i16 %1, %2, %0" : "=&v"(t), "=&v"(r) : "v"(a));
  return r;
}
// the same for N dwords at once: all swaps, then all maxima -- the two wait states a DPP read needs behind the write of its source
// (and a read of a DPP result behind it) are then other members of the batch instead of s_nop
template <int N> __device__ inline void pp_pkmax_lane1_n(unsigned (&a)[N]) {
  static_assert(N >= 3, "the batch is its own padding");
  unsigned t[N];
  asm volatile("s_nop 1" ::: );
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(t[i]) : "v"(a[i]));
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(t[i]));
}


// the next work ticket: the returning atomic and its wait as ONE statement inside the caller's branch (left to the compiler, the wait
// for the returned value moves behind the branch's join, where every wave of the workgroup drains its LDS-DMA queue for it)
__device__ inline int ud_take_ticket(unsigned* ticket) {
  unsigned t;
  asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(t) : "v"(ticket), "v"(1u) : "memory");
  return (int)t;
}

}  // namespace sh
