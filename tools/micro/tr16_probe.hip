// gfx950 ds_read_b64_tr_b16 semantics probe: lane l supplies the LDS address of 4 consecutive 16-bit elements; what does it get?
// Hypothesis (16-lane groups): out[lane c][j] = in[lane 4*j + c/4 of the group][c % 4].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* slot, unsigned short* out) {
  __shared__ unsigned short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + slot[threadIdx.x] * 4));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
  int h_slot[64];
  unsigned short h_out[256];
  int *d_slot; unsigned short* d_out;
  hipMalloc(&d_slot, sizeof h_slot); hipMalloc(&d_out, sizeof h_out);
  int bad_total = 0;
  for (int trial = 0; trial < 3; ++trial) {
    for (int l = 0; l < 64; ++l) h_slot[l] = trial == 0 ? l : trial == 1 ? (l * 37 + 11) % 256 : (255 - 3 * l);
    hipMemcpy(d_slot, h_slot, sizeof h_slot, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_slot, d_out);
    hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const int g = l & ~15, c = l & 15;
        const int want = h_slot[g + 4 * j + c / 4] * 4 + c % 4;
        if (h_out[l * 4 + j] != want) ++bad;
      }
    printf("trial %d: %d mismatches against the hypothesis\n", trial, bad);
    if (trial == 0 || bad) {
      for (int l = 0; l < 20; ++l) printf("  lane %2d (slot %3d): %4d %4d %4d %4d\n", l, h_slot[l], h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
    }
    bad_total += bad;
  }
  printf("%s\n", bad_total ? "HYPOTHESIS WRONG" : "hypothesis confirmed");
  return 0;
}
