// Mock of <hip/hip_bf16.h> for the host-side kernel tests (see hip_runtime.h in this directory).
#pragma once
#include <cstdint>
#include <cstring>
struct __hip_bfloat16 { uint16_t bits; };
static inline __hip_bfloat16 __float2bfloat16(float f) {   // round to nearest even, like v_cvt_pk_bf16_f32
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return {uint16_t((u >> 16) | 0x40)};
  u += 0x7fffu + ((u >> 16) & 1u);
  return {uint16_t(u >> 16)};
}
static inline float __bfloat162float(__hip_bfloat16 b) {
  uint32_t u = uint32_t(b.bits) << 16; float f; memcpy(&f, &u, 4); return f;
}
