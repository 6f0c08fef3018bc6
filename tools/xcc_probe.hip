// Which XCD does workgroup b of a 1-D grid run on?  (speed-only assumption of the LDS GEMM's tile map)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;  // HW_REG_XCC_ID
}
int main() {
    for (int n : {224, 256, 416, 832, 1664}) {
        for (int threads : {256, 512, 1024}) {
            int* d; hipMalloc(&d, n * sizeof(int));
            hipLaunchKernelGGL(probe, dim3(n), dim3(threads), 0, 0, d);
            std::vector<int> h(n); hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
            int ok = 0; for (int b = 0; b < n; ++b) ok += (h[b] == b % 8);
            printf("grid %4d x %4d threads: %d/%d blocks on XCD b%%8; first 16:", n, threads, ok, n);
            for (int b = 0; b < 16; ++b) printf(" %d", h[b]);
            printf("\n");
            hipFree(d);
        }
    }
    return 0;
}
