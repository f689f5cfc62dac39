#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o) { extern __shared__ float s[]; s[threadIdx.x] = 1; __syncthreads(); o[threadIdx.x] = s[(threadIdx.x + 1) & 255]; }
int main() {
  for (int bytes = 31000; bytes <= 33800; bytes += 256) {
    int n = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, bytes);
    printf("%d B dynamic LDS, 256 threads: %d blocks per CU\n", bytes, n);
  }
  for (int bytes : {32768, 32767, 32769, 40960, 40961, 50176, 54613, 54614}) {
    int n = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, bytes);
    printf("%d B: %d\n", bytes, n);
  }
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerMultiprocessor %zu maxSharedMemoryPerMultiProcessor %zu sharedMemPerBlock %zu\n", p.sharedMemPerMultiprocessor, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock);
}
