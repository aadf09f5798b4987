// Microbenchmark: banded single-wave recursion with neighbour exchange over DPP (no LDS, no barrier).
// lane L holds R consecutive states; state s takes alpha[s-d], d = 0..BW, from its own lane or lane L-1.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int N = 4000;

template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// VAR 0: q in registers (constants); 1: q read from LDS each step (ds_read_b128, no sync); 2: VAR 1 + global row stores
template <int R, int BW, int VAR>
__global__ void __launch_bounds__(64) band(double* out, unsigned long long* cyc, double* rows) {
    extern __shared__ double qs[];   // [slots][R*(BW+1)][64] as double2 pieces
    const int lane = threadIdx.x;
    double a[R];
    double q[R][BW + 1];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        a[r] = 1.0 + 1e-3 * (lane * R + r);
#pragma unroll
        for (int d = 0; d <= BW; ++d) q[r][d] = 1.0 / (BW + 1) + 1e-6 * d;
    }
    constexpr int NQ = R * (BW + 1);
    if (VAR >= 1) {
        for (int s = 0; s < 8; ++s)
            for (int i = 0; i < NQ; ++i) qs[(s * NQ + i) * 64 + lane] = 1.0 / (BW + 1) + 1e-6 * (i % (BW + 1));
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N; it += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (VAR >= 1) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int d = 0; d <= BW; ++d) q[r][d] = qs[(k * NQ + r * (BW + 1) + d) * 64 + lane];
            }
            // neighbour values: the last BW states of lane L-1 (wave_shr:1 = 0x138)
            double nb[BW];
#pragma unroll
            for (int j = 0; j < BW; ++j) nb[j] = dpp_d<0x138>(a[R - 1 - j]);   // nb[j] = state (s0 - 1 - j)
            double an[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double v = 0.0;
#pragma unroll
                for (int d = 0; d <= BW; ++d) {
                    const int idx = r - d;   // own lane if >= 0, else neighbour nb[-idx-1]
                    const double src = idx >= 0 ? a[idx] : nb[-idx - 1];
                    v = fma(src, q[r][d], v);
                }
                an[r] = v;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = an[r];
            if (VAR >= 2) {
#pragma unroll
                for (int r = 0; r < R; ++r) rows[((size_t)(it + k) * R + r) * 64 + lane] = a[r];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) s += a[r];
    out[lane] = s;
    if (lane == 0) cyc[0] = t1 - t0;
}

template <int R, int BW, int VAR>
void run(const char* name, double* out, unsigned long long* cyc, double* rows) {
    size_t lds = 8 * R * (BW + 1) * 64 * 8;
    for (int i = 0; i < 2; ++i) {
        hipLaunchKernelGGL((band<R, BW, VAR>), dim3(1), dim3(64), lds, 0, out, cyc, rows);
        CHECK(hipDeviceSynchronize());
    }
    unsigned long long c;
    CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-40s R=%d BW=%d  cycles/step=%7.1f\n", name, R, BW, (double)c / N);
}

int main() {
    double *out, *rows; unsigned long long* cyc;
    CHECK(hipMalloc(&out, 4096)); CHECK(hipMalloc(&cyc, 64)); CHECK(hipMalloc(&rows, (size_t)N * 8 * 64 * 8));
    run<1, 2, 0>("q in regs", out, cyc, rows);
    run<3, 2, 0>("q in regs", out, cyc, rows);
    run<3, 2, 1>("q from LDS", out, cyc, rows);
    run<3, 2, 2>("q from LDS + row stores", out, cyc, rows);
    run<4, 4, 0>("q in regs", out, cyc, rows);
    run<4, 4, 1>("q from LDS", out, cyc, rows);
    run<4, 4, 2>("q from LDS + row stores", out, cyc, rows);
    run<4, 3, 2>("q from LDS + row stores", out, cyc, rows);
    run<8, 4, 2>("q from LDS + row stores", out, cyc, rows);
    return 0;
}
