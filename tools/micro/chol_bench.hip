// Microbenchmark (testing aid): the LDS substitutions of the NLMPC sub-problem in isolation, one wavefront.
// hipcc --offload-arch=gfx950 -O3 -std=c++20 -I../../include -o chol_bench chol_bench.hip   (measured: ~340-370 cycles per step)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <mpcx/nlmpc_device.hpp>
#include <mpcx/nlmpc_engine.hpp>
using namespace mpcx::engine;

__global__ void kern(const int *np, int reps, double *out, long long *cyc)
{
    const int n = np[threadIdx.x];          // not provably uniform: the compiler keeps it in a VGPR, as the solver's working-set size
    extern __shared__ double smem[];
    double *Lp = smem, *invd = smem + 4000;
    const int lane = threadIdx.x;
    for (int r = 0; r < n; ++r)
        for (int k = lane; k <= r; k += 64) Lp[r * (r + 1) / 2 + k] = k == r ? 2.0 + 0.01 * r : 0.01 * ((r * 7 + k * 3) % 11);
    for (int r = lane; r < n; r += 64) invd[r] = 1.0 / Lp[r * (r + 1) / 2 + r];
    nl_wave_sync();
    const Factor F{Lp, invd, smem + 4200, nullptr, 128, 0};
    double t0 = lane < n ? 1.0 + lane : 0.0, t1 = lane + 64 < n ? 0.5 * lane : 0.0;
    const long long c0 = __builtin_readcyclecounter();
    const long long w0 = wall_clock64();
    for (int it = 0; it < reps; ++it) {
        chol_forward<false>(F, n, t0, t1, lane);
        chol_backward<false>(F, n, t0, t1, lane);
        t0 = t0 * 0.5 + 1.0; t1 = t1 * 0.5 + 1.0;
    }
    const long long c1 = __builtin_readcyclecounter();
    const long long w1 = wall_clock64();
    out[lane] = t0 + t1;
    if (lane == 0) { cyc[0] = c1 - c0; cyc[1] = w1 - w0; }
}

int main()
{
    double *out; long long *cyc; int *np;
    hipMalloc(&np, 64 * 4);
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16);
    for (int n : {8, 16, 30, 60, 100}) {
        const int reps = 200;
        int hn[64]; for (int i = 0; i < 64; ++i) hn[i] = n;
        hipMemcpy(np, hn, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), 40000, 0, np, reps, out, cyc);
        hipDeviceSynchronize();
        long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        std::printf("n %3d: %8.0f s_memtime ticks, %8.1f ns (wall_clock64 at 100 MHz) per forward+backward pair; per step %.1f ticks\n", n,
                    (double)h[0] / reps, (double)h[1] * 10.0 / reps, (double)h[0] / reps / (2 * n));
    }
    return 0;
}
