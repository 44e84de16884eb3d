// Micro-benchmark (tools/micro): what the building blocks of a latency-bound workgroup kernel cost on this chip, in shader-clock cycles --
// a chain of dependent LDS reads, a chain of dependent f64 FMAs, a workgroup barrier with 4 wavefronts, a DPP wave reduction, a
// v_readlane broadcast, a call of a small non-inlined function, a dependent scalar load, a dependent global load (L2 hit), an LDS atomic add.
// hipcc --offload-arch=gfx950 -O3 -o latency tools/micro/latency.hip && ./latency
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ long long now() { return __builtin_readcyclecounter(); }
__device__ __attribute__((noinline)) double small_call(double x, const double *p) { return x * 1.0000001 + p[0]; }

template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL> __device__ __forceinline__ double dpp_d(double v) { return __hiloint2double(dpp_i<CTRL>(__double2hiint(v)), dpp_i<CTRL>(__double2loint(v))); }
__device__ __forceinline__ double lane_d(double v, int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); }
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_d<0xB1>(v); v += dpp_d<0x4E>(v); v += dpp_d<0x141>(v); v += dpp_d<0x140>(v);
    return (lane_d(v, 0) + lane_d(v, 16)) + (lane_d(v, 32) + lane_d(v, 48));
}

__global__ __launch_bounds__(256) void probe(long long *out, const int *gidx, const double *gval, int reps)
{
    extern __shared__ double lds[];
    int *ilds = reinterpret_cast<int *>(lds + 2048);
    const int tid = threadIdx.x;
    for (int k = tid; k < 2048; k += 256) { lds[k] = 1.0 + k * 1e-9; ilds[k] = (k * 7 + 1) & 2047; }
    __syncthreads();
    long long t[12];
    // 1. dependent LDS reads (pointer chase through an int table)
    int p = tid & 63;
    t[0] = now();
    for (int r = 0; r < reps; ++r) p = ilds[p];
    t[1] = now();
    // 2. dependent f64 FMAs
    double x = 1.0 + p * 1e-12;
    for (int r = 0; r < reps; ++r) x = fma(x, 1.0000001, 1e-9);
    t[2] = now();
    // 3. workgroup barriers
    for (int r = 0; r < reps; ++r) __syncthreads();
    t[3] = now();
    // 4. wave reductions (DPP + readlane)
    for (int r = 0; r < reps; ++r) x = wave_sum(x) * 0.015625;
    t[4] = now();
    // 5. calls of a small function
    for (int r = 0; r < reps; ++r) x = small_call(x, lds);
    t[5] = now();
    // 6. dependent global loads (pointer chase, table of 2048 ints: L2 / L1 resident)
    int g = tid & 63;
    for (int r = 0; r < reps; ++r) g = gidx[g];
    t[6] = now();
    // 7. LDS atomic adds on doubles (distinct addresses per lane)
    for (int r = 0; r < reps; ++r) atomicAdd(lds + tid, 1e-9);
    __syncthreads();
    t[7] = now();
    // 8. independent LDS reads, 8 in flight
    double acc = 0;
    for (int r = 0; r < reps; r += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = lds[(tid + 64 * u + r) & 2047];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    t[8] = now();
    // 9. dependent scalar loads (uniform pointer chase)
    int sidx = 5;
    for (int r = 0; r < reps; ++r) sidx = __builtin_amdgcn_readfirstlane(gidx[sidx]);
    t[9] = now();
    // 10. LDS read -> FMA -> LDS write -> barrier (a typical step of a chain across wavefronts)
    for (int r = 0; r < reps; ++r) { const double v = lds[(tid + 1) & 255]; __syncthreads(); lds[tid] = fma(v, 0.5, 0.25); __syncthreads(); }
    t[10] = now();
    if (tid == 0) for (int k = 0; k < 10; ++k) out[blockIdx.x * 16 + k] = t[k + 1] - t[k];
    if (x + acc + p + g + sidx == 12345.678) out[15] = 1;
}

int main()
{
    const int reps = 256, blocks = 512;
    long long *out; int *gidx; double *gval;
    hipMalloc(&out, blocks * 16 * sizeof(long long)); hipMalloc(&gidx, 2048 * sizeof(int)); hipMalloc(&gval, 2048 * sizeof(double));
    int h[2048]; for (int k = 0; k < 2048; ++k) h[k] = (k * 7 + 1) & 2047;
    hipMemcpy(gidx, h, sizeof(h), hipMemcpyHostToDevice);
    const char *names[10] = {"dependent LDS read", "dependent f64 FMA", "workgroup barrier (4 waves)", "wave reduction (DPP + readlane)", "call of a small function",
                             "dependent global load (cached)", "LDS atomic add f64", "LDS read, 8 in flight (per read)", "dependent uniform global load + readfirstlane",
                             "LDS read + 2 barriers + LDS write"};
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int nb = cfg == 0 ? 256 : blocks, ldsb = cfg == 0 ? 150 * 1024 : 70 * 1024;      // one or two workgroups per CU
        hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
        hipLaunchKernelGGL(probe, dim3(nb), dim3(256), ldsb, 0, out, gidx, gval, reps);
        hipLaunchKernelGGL(probe, dim3(nb), dim3(256), ldsb, 0, out, gidx, gval, reps);
        hipDeviceSynchronize();
        long long ho[16 * 512];
        hipMemcpy(ho, out, nb * 16 * sizeof(long long), hipMemcpyDeviceToHost);
        printf("%s workgroup(s) of 4 wavefronts per CU:\n", cfg == 0 ? "one" : "two");
        for (int k = 0; k < 10; ++k) {
            double s = 0; for (int b = 0; b < nb; ++b) s += ho[b * 16 + k];
            printf("  %-48s %7.1f cycles\n", names[k], s / nb / reps);
        }
    }
    return 0;
}
