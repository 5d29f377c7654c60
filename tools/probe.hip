// Hardware probe (gfx950): prints the lane/element mapping of ds_read_b64_tr_b16
// and checks the MFMA C/D layouts the kernels assume.  Build: hipcc --offload-arch=gfx950 -O2 tools/probe.hip -o tools/bin/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__global__ void tr_probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  // mode 0: lane address = lane*8 bytes (contiguous); mode 1: 16-lane group g reads a [4][16] row-major block:
  // lane (c = l&15, g = l>>4): address of element (row ?, col ?) -- try "lane i -> row i/4?": use addr = g*128 + (l&15)*8
  // mode 2: addr = (l&15)*64*2 + g*8  (each lane its own row of 64 elements, 4 consecutive cols per group)
  uint32_t addr;
  if (mode == 0) addr = l * 8;
  else if (mode == 1) addr = (l >> 4) * 128 + (l & 15) * 8;
  else addr = (l & 15) * 128 + (l >> 4) * 8;
  uint32_t base = (uint32_t)(uintptr_t)lds;
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr) : "memory");
  out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16;
  out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}

__global__ void mfma_probe(float* out) {
  // A[i][k] = (i == k) for k < 16 (identity in the first 16 columns) ; B[k][j] = 100*k + j  => D[i][j] = 100*i + j for i < 16
  const int l = threadIdx.x;
  const int i = l & 31, g = l >> 5;
  bf16x8_t a, b;
  for (int e = 0; e < 8; ++e) {
    const int k = g * 8 + e;
    a[e] = (__bf16)((i == k) ? 1.0f : 0.0f);
    b[e] = (__bf16)(float)(16 * k + (i & 15));   // exact in bf16 (< 256)
  }
  f32x16_t acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[l * 16 + r] = acc[r];
}

int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d, mode);
    std::vector<uint16_t> h(256);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16 mode %d (lane: 4 element indices read)\n", mode);
    for (int l = 0; l < 64; ++l) printf("  l%02d: %4d %4d %4d %4d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l % 4 == 3) ? "\n" : "");
  }
  float* f; hipMalloc(&f, 64 * 16 * 4);
  hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, f);
  std::vector<float> hf(1024);
  hipMemcpy(hf.data(), f, 4096, hipMemcpyDeviceToHost);
  // expected with D layout col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5):  D[row][col] = 16*row + (col&15) for row < 16 else 0
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    const float want = row < 16 ? (float)(16 * row + (col & 15)) : 0.f;
    if (hf[l * 16 + r] != want) { if (bad < 8) printf("mfma mismatch lane %d reg %d got %g want %g\n", l, r, hf[l*16+r], want); ++bad; }
  }
  printf("mfma_f32_32x32x16_bf16 C/D layout + k mapping (lane>>5)*8+e: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
  return 0;
}
