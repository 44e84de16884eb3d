// Batched zero-order-hold discretisation on the device: what libmpc++'s discretization<>() does on the host
// (reference include/mpc/Utils.hpp:23-47, 63-89): [Ad Bd; 0 I] = exp([[A B]; [0 0]] * Ts).  One model per wavefront,
// the (nx+nu)^2 matrix in LDS, scaling and squaring around a Taylor series (‖M / 2^s‖_1 <= 1/2, 18 terms: the
// truncation error is below 1e-19 relative, the squarings add round-off only).  n = nx + nu <= 48.
#include <hip/hip_runtime.h>

#include <cmath>

namespace mpcx {
namespace {

__device__ __forceinline__ void c2d_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// C = A * B (n x n, row-major, ld = n), lanes over the entries of C
__device__ __forceinline__ void mm(double *C, const double *A, const double *B, int n, int lane, double scale)
{
    for (int e = lane; e < n * n; e += 64) {
        const int i = e / n, j = e - i * n;
        double s = 0;
        for (int k = 0; k < n; ++k) s += A[i * n + k] * B[k * n + j];
        C[e] = s * scale;
    }
    c2d_sync();
}

__global__ __launch_bounds__(64) void c2d_expm(int nx, int nu, int batch, const double *__restrict__ A, const double *__restrict__ Bm,
                                              const double *__restrict__ Ts, int ts_stride, double *__restrict__ Ad, double *__restrict__ Bd)
{
    extern __shared__ double sm[];
    const int lane = threadIdx.x, n = nx + nu;
    double *M = sm, *T = sm + n * n, *S = T + n * n, *W = S + n * n;
    for (int b = blockIdx.x; b < batch; b += gridDim.x) {
        const double ts = Ts[(size_t)b * ts_stride];
        // M = [[A B]; [0 0]] * Ts   (inputs column-major like Eigen's: A[b][j*nx + i])
        for (int e = lane; e < n * n; e += 64) {
            const int i = e / n, j = e - i * n;
            double v = 0;
            if (i < nx) v = j < nx ? A[(size_t)b * nx * nx + (size_t)j * nx + i] : Bm[(size_t)b * nx * nu + (size_t)(j - nx) * nx + i];
            M[e] = v * ts;
        }
        c2d_sync();
        // 1-norm -> number of squarings
        double cs = 0;
        for (int j = lane; j < n; j += 64) { double s = 0; for (int i = 0; i < n; ++i) s += fabs(M[i * n + j]); cs = fmax(cs, s); }
        for (int o = 32; o; o >>= 1) cs = fmax(cs, __shfl_xor(cs, o));
        int sq = 0;
        if (cs > 0.5) { sq = (int)ceil(log2(cs / 0.5)); if (sq > 60) sq = 60; }
        const double sc = ldexp(1.0, -sq);
        for (int e = lane; e < n * n; e += 64) { M[e] *= sc; const int i = e / n, j = e - i * n; S[e] = (i == j ? 1.0 : 0.0) + M[e]; T[e] = M[e]; }
        c2d_sync();
        for (int k = 2; k <= 18; ++k) {              // T_k = T_{k-1} M / k ; S += T_k
            mm(W, T, M, n, lane, 1.0 / k);
            for (int e = lane; e < n * n; e += 64) { T[e] = W[e]; S[e] += W[e]; }
            c2d_sync();
        }
        for (int q = 0; q < sq; ++q) {
            mm(W, S, S, n, lane, 1.0);
            for (int e = lane; e < n * n; e += 64) S[e] = W[e];
            c2d_sync();
        }
        for (int e = lane; e < nx * n; e += 64) {
            const int i = e / n, j = e - i * n;
            if (j < nx) Ad[(size_t)b * nx * nx + (size_t)j * nx + i] = S[e];
            else Bd[(size_t)b * nx * nu + (size_t)(j - nx) * nx + i] = S[e];
        }
        c2d_sync();
    }
}

}  // namespace

int c2d_launch(int nx, int nu, int batch, const double *A, const double *B, const double *Ts, int ts_stride, double *Ad, double *Bd,
               void *stream)
{
    const int n = nx + nu;
    if (n > 48) return -2;
    const size_t lds = (size_t)4 * n * n * sizeof(double);
    int blocks = batch < 4096 ? batch : 4096;
    hipLaunchKernelGGL(c2d_expm, dim3(blocks), dim3(64), lds, reinterpret_cast<hipStream_t>(stream), nx, nu, batch, A, B, Ts, ts_stride, Ad, Bd);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace mpcx
