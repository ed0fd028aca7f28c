// ds_read_b64_tr_b16 semantics probe (gfx950): LDS holds element index i at 2-byte slot i; every lane reads with its own address
//   mode 0: all lanes use address 0            mode 1: lane l uses byte address 8 * l        mode 2: lane l uses 8 * (l & 15) + 512 * (l >> 4)
// prints the 4 elements each lane received.   hipcc --offload-arch=gfx950 -O2 -o tools/tr_probe tools/tr_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint16_t* out, int mode) {
  __shared__ uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds;
  if (mode == 1) addr += 8 * l;
  if (mode == 2) addr += 8 * (l & 15) + 512 * (l >> 4);
  if (mode == 3) addr += 32 * (l & 15) + 8 * (l >> 4);      // lane i of a group at row i (row stride 32 B), group g at column block g
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)(v >> (16 * j));
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 4; ++mode) {
    probe<<<1, 64>>>(d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("  lane %2d: %4d %4d %4d %4d", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); if (l % 4 == 3) printf("\n"); }
  }
  return 0;
}
