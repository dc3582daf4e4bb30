// What bounds the recurrence's exchange (profiles/NOTES.md: "h fragments stream out of the L2 at ~0.5 KB per clock and XCD",
// a quarter of what MI355X_MICROARCH.md gives an XCD's L2)?  Hypothesis (round 6): the hand-off image of a group is small and every
// member reads ALL of it, so the requests of 30 CUs fall on the few L2 channels that image maps to.  This probe reproduces
// the read side only: R workgroups per XCD (workgroup b -> XCD b % 8, as the recurrence relies on), 8 waves each, every
// workgroup reads the same N bytes of its XCD's region with `sc1` 16-byte buffer loads (L2-served), the region laid out as pieces of
// S bytes at a stride of D bytes.  Reported: clocks per pass and bytes per clock and XCD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/l2_gather_probe tools/micro/l2_gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int UPT>   // 16-byte units per thread and pass
__global__ __launch_bounds__(512) void gather_kernel(const char* base, long group_bytes, int units, int piece_bytes, long piece_stride,
                                                     int iters, int readers, int aux_plain, unsigned* sink, long long* clk) {
  const int g = blockIdx.x & 7, m = blockIdx.x >> 3;
  if (m >= readers) return;
  const int tid = threadIdx.x;
  unsigned off[UPT];
#pragma unroll
  for (int k = 0; k < UPT; ++k) {
    const int u = tid + 512 * k;
    const long byte = (long)u * 16;
    off[k] = u < units ? (unsigned)((byte / piece_bytes) * piece_stride + byte % piece_bytes) : 0x7ffffff0u;
  }
  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long)g * group_bytes), 0, (int)(group_bytes / 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long)g * group_bytes + group_bytes / 2), 0, (int)(group_bytes / 2), 0x00020000);
  unsigned acc = 0;
  // warm the L2, then line the workgroups up roughly (they start within < 1 us of each other)
#pragma unroll
  for (int k = 0; k < UPT; ++k) acc ^= __builtin_amdgcn_raw_buffer_load_b128(r0, off[k], 0, 16)[0] ^ __builtin_amdgcn_raw_buffer_load_b128(r1, off[k], 0, 16)[1];
  __syncthreads();
  __builtin_amdgcn_s_sleep(127);
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    u32x4 v[UPT];
    if (aux_plain) {
#pragma unroll
      for (int k = 0; k < UPT; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128((it & 1) ? r1 : r0, off[k], 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < UPT; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128((it & 1) ? r1 : r0, off[k], 0, 16);
    }
#pragma unroll
    for (int k = 0; k < UPT; ++k) acc ^= v[k][0] ^ v[k][1] ^ v[k][2] ^ v[k][3];
    asm volatile("" ::: "memory");
    __syncthreads();      // one pass at a time, like a time step
  }
  const long long t1 = clock64();
  if (tid == 0) clk[blockIdx.x] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
  const long group_bytes = 8l << 20;     // two 4 MB halves per XCD (alternating "slots")
  char* base; unsigned* sink; long long* clk;
  hipMalloc(&base, 8 * group_bytes); hipMalloc(&sink, 4); hipMalloc(&clk, 8 * 32 * sizeof(long long));
  hipMemset(base, 1, 8 * group_bytes);
  const int iters = 2000;
  struct Cfg { const char* name; int bytes, piece; long stride; };
  std::vector<Cfg> cfgs = {
      {"production B=32 (1 KB of every 2 KB)", 20 * 1024, 1024, 2048},
      {"production B=16 (512 B of every 2 KB)", 20 * 512, 512, 2048},
      {"production B=64 (2 KB chunks, dense)", 20 * 2048, 2048, 2048},
      {"dense 20 KB", 20 * 1024, 20 * 1024, 20 * 1024},
      {"128 B pieces, stride 128 (dense)", 20 * 1024, 128, 128},
      {"128 B pieces, stride 256", 20 * 1024, 128, 256},
      {"128 B pieces, stride 512", 20 * 1024, 128, 512},
      {"128 B pieces, stride 1 KB", 20 * 1024, 128, 1024},
      {"128 B pieces, stride 2 KB", 20 * 1024, 128, 2048},
      {"128 B pieces, stride 4 KB", 20 * 1024, 128, 4096},
      {"128 B pieces, stride 8 KB", 20 * 1024, 128, 8192},
      {"128 B pieces, stride 16 KB", 20 * 1024, 128, 16384},
      {"128 B pieces, stride 4 KB + 128", 20 * 1024, 128, 4096 + 128},
      {"128 B pieces, stride 4 KB + 256", 20 * 1024, 128, 4096 + 256},
      {"256 B pieces, stride 4 KB + 256", 20 * 1024, 256, 4096 + 256},
      {"1 KB pieces, stride 4 KB + 1 KB", 20 * 1024, 1024, 4096 + 1024},
      {"1 KB pieces, stride 4 KB", 20 * 1024, 1024, 4096},
      {"1 KB pieces, stride 8 KB + 1 KB", 20 * 1024, 1024, 8192 + 1024},
      {"1 KB pieces, stride 16 KB + 1 KB", 20 * 1024, 1024, 16384 + 1024},
      {"1 KB pieces, stride 64 KB + 1 KB", 20 * 1024, 1024, 65536 + 1024},
  };
  printf("%-44s %8s %12s %12s %12s\n", "layout", "readers", "clk/pass", "B/clk/XCD", "B/clk/CU");
  for (int plain = 0; plain < 2; ++plain) {
    if (plain) printf("---- plain loads (L1 allowed) ----\n");
    for (const Cfg& c : cfgs) {
      for (int readers : {1, 4, 15, 30}) {
        if (plain && readers != 30) continue;
        const int units = c.bytes / 16;
        if ((long)(c.bytes / c.piece) * c.stride > group_bytes / 2) continue;
        hipLaunchKernelGGL(gather_kernel<6>, dim3(8 * 30), dim3(512), 0, 0, base, group_bytes, units, c.piece, c.stride, iters, readers, plain, sink, clk);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        std::vector<long long> h(8 * 30);
        hipMemcpy(h.data(), clk, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double mx = 0;
        for (int b = 0; b < 8 * readers; ++b) mx = h[b] > mx ? (double)h[b] : mx;
        const double per = mx / iters;
        printf("%-44s %8d %12.1f %12.1f %12.1f\n", c.name, readers, per, (double)c.bytes * readers / per, (double)c.bytes / per);
      }
    }
  }
  return 0;
}
