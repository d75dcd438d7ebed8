// Which XCD does workgroup (x, y) of a 2-D grid run on? Prints, for a few grid shapes, how often the XCC id
// of a workgroup equals the id of workgroup 0 rotated by its linear dispatch index (y * gridDim.x + x) mod 8
// -- the assumption behind xcd_wg() in pcs_icp.hip. Build: hipcc --offload-arch=gfx950 -O2 -o tools/xcd_probe tools/xcd_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* out) {
  if (threadIdx.x == 0) {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    out[blockIdx.y * gridDim.x + blockIdx.x] = v & 15;
  }
}
int main() {
  const int shapes[][2] = {{8, 1}, {40, 9}, {37, 15}, {64, 36}, {1024, 3}};
  for (auto& sh : shapes) {
    const int n = sh[0] * sh[1];
    int* d;
    hipMalloc(&d, n * sizeof(int));
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(d, 0xff, n * sizeof(int));
      probe<<<dim3(sh[0], sh[1]), 256>>>(d);
      std::vector<int> h(n);
      hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
      int ok = 0;
      for (int i = 0; i < n; ++i) ok += h[i] == ((h[0] + i) & 7);
      printf("grid %4d x %2d rep %d: xcc(block 0) = %d, round-robin in linear order for %d of %d blocks\n", sh[0],
             sh[1], rep, h[0], ok, n);
    }
    hipFree(d);
  }
  return 0;
}
