// How fast can the chip start workgroups?  Empty (and nearly empty) kernels over grids of 16k - 262k workgroups of 64 - 1024 threads,
// with and without a static LDS allocation: time per launch -> workgroups per microsecond and cycles per workgroup per XCD.
// Answers whether a kernel of many short-lived workgroups (k_blur: 65k workgroups of ~3000 cycles) is bound by the dispatcher.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NT, int LDS> __global__ __launch_bounds__(NT) void k_empty(int* out, int spin) {
    __shared__ int s[LDS > 0 ? LDS / 4 : 1];
    if (LDS > 0) s[threadIdx.x % (LDS / 4)] = threadIdx.x;
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (out && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = s[0];
}

template <int NT, int LDS> void run(int nwg, int spin) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_empty<NT, LDS>), dim3(nwg), dim3(NT), 0, 0, nullptr, spin);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_empty<NT, LDS>), dim3(nwg), dim3(NT), 0, 0, nullptr, spin);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps;
    printf("threads %4d lds %6d spin %5d  wgs %7d : %8.1f us  %7.1f wg/us  %6.1f cycles/wg/xcd (2.4 GHz, 8 XCD)\n", NT, LDS, spin, nwg, us, nwg / us,
           us * 2400.0 * 8 / nwg);
}

int main() {
    for (int spin : {0, 3000}) {
        for (int nwg : {16384, 65536, 262144}) {
            run<64, 0>(nwg, spin); run<256, 0>(nwg, spin); run<256, 13312>(nwg, spin); run<256, 32768>(nwg, spin); run<512, 0>(nwg, spin); run<1024, 0>(nwg, spin);
        }
    }
    return 0;
}
