// Occupancy probe: how many 1024-thread workgroups with S bytes of static LDS share a CU on gfx950?
// (round 6: the 400-workgroup forward's last workgroup enters 5.7 us after the first -- one workgroup per CU?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int LDS_BYTES>
__global__ __launch_bounds__(1024) void spin(unsigned long long* t_in, unsigned long long* t_out, int spin_ticks) {
    __shared__ float buf[LDS_BYTES / 4];
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long w0 = wall_clock64();
    buf[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    while (wall_clock64() - w0 < (unsigned long long)spin_ticks) { buf[(threadIdx.x * 7) & 1023] += 1.f; }
    __syncthreads();
    if (threadIdx.x == 0) { t_in[blockIdx.x] = w0; t_out[blockIdx.x] = wall_clock64() + (unsigned long long)(buf[5] > 1e30f); }
    (void)t0;
}
template <int L> void run(int blocks) {
    int occ = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin<L>, 1024, 0);
    unsigned long long *a, *b;
    hipMalloc(&a, blocks * 8); hipMalloc(&b, blocks * 8);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(spin<L>, dim3(blocks), dim3(1024), 0, 0, a, b, 500); hipDeviceSynchronize(); }
    std::vector<unsigned long long> in(blocks), out(blocks);
    hipMemcpy(in.data(), a, blocks * 8, hipMemcpyDeviceToHost); hipMemcpy(out.data(), b, blocks * 8, hipMemcpyDeviceToHost);
    const unsigned long long t0 = *std::min_element(in.begin(), in.end());
    std::sort(in.begin(), in.end());
    int late = 0; for (auto t : in) late += (t - t0) > 300;      // entered more than 3 us after the first (100 MHz ticks)
    printf("lds %6d B  blocks %4d  occupancy API %d  last entry %.2f us  median entry %.2f us  entered > 3 us late: %d  last exit %.2f us\n", L, blocks, occ,
           (in.back() - t0) / 100.0, (in[blocks / 2] - t0) / 100.0, late, (*std::max_element(out.begin(), out.end()) - t0) / 100.0);
    hipFree(a); hipFree(b);
}
int main() {
    for (int blocks : {256, 400, 512}) { run<65536>(blocks); run<32768>(blocks); run<16384>(blocks); run<4096>(blocks); }
    return 0;
}
