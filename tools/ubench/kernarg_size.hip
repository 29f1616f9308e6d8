#include <hip/hip_runtime.h>
struct Big { float v[1500]; int n; };
__global__ void k(Big b, float* out) { out[threadIdx.x] = b.v[b.n + threadIdx.x % 7]; }
int main() { Big b{}; b.n = 3; for (int i=0;i<1500;++i) b.v[i]=i; float* d; hipMalloc(&d, 256*4); hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, b, d);
  hipError_t e = hipDeviceSynchronize(); float h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost); printf("%s %f %f\n", hipGetErrorString(e), h[0], h[1]); return 0; }
