// Which XCD does workgroup i of a launch land on?  (s_getreg HW_REG_XCC_ID; MI355X: 8 XCDs x 32 CUs.)
//   hipcc --offload-arch=gfx950 -O2 tools/probe/src/xcc_probe.hip -o /tmp/xcc_probe && /tmp/xcc_probe [blocks] [threads]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void probe(unsigned *out, int hold) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
  // keep the workgroup resident for a while so that the whole grid is placed at once
  long long t0 = clock64();
  while (clock64() - t0 < hold) {}
}
int main(int argc, char **argv) {
  int blocks = argc > 1 ? atoi(argv[1]) : 40, threads = argc > 2 ? atoi(argv[2]) : 1024;
  unsigned *d, *h = (unsigned *)malloc(8 * blocks);
  hipMalloc(&d, 8 * blocks);
  for (int rep = 0; rep < 2; ++rep) {
    probe<<<blocks, threads>>>(d, 200000);
    hipMemcpy(h, d, 8 * blocks, hipMemcpyDeviceToHost);
    printf("launch %d: xcc id of blocks 0..%d:", rep, blocks - 1);
    for (int i = 0; i < blocks; ++i) printf(" %u", h[2 * i] & 0xf);
    printf("\n");
  }
  return 0;
}
