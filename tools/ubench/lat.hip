// Latency microbenchmarks for the frame-step recursion (one workgroup, dependent chain through LDS).
// Build: hipcc --offload-arch=gfx950 -O3 -o lat lat.hip ; run: ./lat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int N = 4000;

template <int VAR>
__global__ void __launch_bounds__(1024) k(float* out, unsigned long long* cyc, float* gstore) {
    __shared__ double bufd[2][1024];
    float* buf0 = (float*)bufd[0];
    float* buf1 = (float*)bufd[1];
    double* d0 = bufd[0];
    double* d1 = bufd[1];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int src = (tid + 1) % nt;
    buf0[tid] = 1.0f + tid * 1e-3f; buf1[tid] = 0.f;
    if (VAR == 2 || VAR == 5 || VAR == 9 || VAR == 10) { d0[tid] = 1.0 + tid * 1e-3; d1[tid] = 0; }
    __syncthreads();
    float* cur = buf0; float* nxt = buf1;
    double* dc = d0; double* dn = d1;
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N; ++i) {
        if (VAR == 1) {            // LDS read -> add -> LDS write -> barrier
            float v = cur[src]; nxt[tid] = v + 1e-6f; __syncthreads();
        } else if (VAR == 2) {     // same, fp64
            double v = dc[src]; dn[tid] = v + 1e-6; __syncthreads();
        } else if (VAR == 3) {     // + 8 dependent fp32 adds
            float v = cur[src];
#pragma unroll
            for (int j = 0; j < 8; ++j) { v = v + 1e-6f; asm volatile("" : "+v"(v)); }
            nxt[tid] = v; __syncthreads();
        } else if (VAR == 4) {     // + exp2, log2 (dependent)
            float v = cur[src];
            v = __builtin_amdgcn_exp2f(v * 1e-3f); v = __builtin_amdgcn_logf(v) + 1.f;
            nxt[tid] = v; __syncthreads();
        } else if (VAR == 5) {     // fp64: 4 reads, mul + 3 dependent fma
            double a = dc[src], b = dc[tid], c = dc[(tid + 2) % nt], d = dc[(tid + 3) % nt];
            double v = a * 0.25; v = fma(b, 0.25, v); v = fma(c, 0.25, v); v = fma(d, 0.25, v);
            dn[tid] = v; __syncthreads();
        } else if (VAR == 6) {     // single wave, no barrier (LDS in-order)
            float v = cur[src]; nxt[tid] = v + 1e-6f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        } else if (VAR == 7) {     // VAR 1 + a global store per step
            float v = cur[src]; v += 1e-6f; nxt[tid] = v; gstore[(size_t)i * nt + tid] = v; __syncthreads();
        } else if (VAR == 8) {     // log-domain 4 arcs per lane (fp32)
            float a = cur[src], b = cur[tid], c = cur[(tid + 2) % nt], d = cur[(tid + 3) % nt];
            float m = fmaxf(fmaxf(a, b), fmaxf(c, d));
            float s = __builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m) + __builtin_amdgcn_exp2f(c - m) + __builtin_amdgcn_exp2f(d - m);
            nxt[tid] = m + __builtin_amdgcn_logf(s) - 2.0f; __syncthreads();
        } else if (VAR == 9) {     // fp64 tree: 4 reads, 2 mul + 2 fma... (a*p + b*q) + (c*r + d*s)
            double a = dc[src], b = dc[tid], c = dc[(tid + 2) % nt], d = dc[(tid + 3) % nt];
            double v = fma(a, 0.25, b * 0.25) + fma(c, 0.25, d * 0.25);
            dn[tid] = v; __syncthreads();
        } else if (VAR == 10) {    // VAR 5 + a global fp64 store per step
            double a = dc[src], b = dc[tid], c = dc[(tid + 2) % nt], d = dc[(tid + 3) % nt];
            double v = a * 0.25; v = fma(b, 0.25, v); v = fma(c, 0.25, v); v = fma(d, 0.25, v);
            dn[tid] = v; ((double*)gstore)[(size_t)i * nt + tid] = v; __syncthreads();
        } else if (VAR == 11) {    // barrier only
            __syncthreads();
        } else if (VAR == 12) {    // LDS read -> write, no barrier needed (own element): pure LDS round trip
            float v = cur[tid]; cur[tid] = v + 1e-6f;
        } else if (VAR == 13) {    // 8 dependent fp64 fma
            double v = dc[src];
#pragma unroll
            for (int j = 0; j < 8; ++j) { v = fma(v, 0.999, 1e-6); asm volatile("" : "+v"(v)); }
            dn[tid] = v; __syncthreads();
        } else if (VAR == 14) {    // 4 dependent exp2
            float v = cur[src];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v = __builtin_amdgcn_exp2f(v * 1e-3f); asm volatile("" : "+v"(v)); }
            nxt[tid] = v; __syncthreads();
        }
        float* t = cur; cur = nxt; nxt = t;
        double* td = dc; dc = dn; dn = td;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    acc = cur[tid] + (float)dc[tid];
    out[tid] = acc;
    if (tid == 0) cyc[0] = t1 - t0;
}

template <int VAR>
void run(const char* name, int nt, float* out, unsigned long long* cyc, float* gs) {
    hipLaunchKernelGGL(k<VAR>, dim3(1), dim3(nt), 0, 0, out, cyc, gs);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<VAR>, dim3(1), dim3(nt), 0, 0, out, cyc, gs);
    CHECK(hipDeviceSynchronize());
    unsigned long long c;
    CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    // s_memtime counts at 100 MHz on gfx9 (constant clock); report both raw ticks and ns
    printf("%-44s nt=%4d  ticks/step=%8.2f  ns/step=%8.1f\n", name, nt, (double)c / N, (double)c / N * 10.0);
}

int main() {
    float *out, *gs; unsigned long long* cyc;
    CHECK(hipMalloc(&out, 4096 * 4)); CHECK(hipMalloc(&cyc, 64)); CHECK(hipMalloc(&gs, (size_t)N * 1024 * 8));
    int nts[] = {64, 192, 256, 576, 1024};
    for (int nt : nts) {
        run<11>("barrier only", nt, out, cyc, gs);
        run<12>("LDS rd->wr own element (no barrier)", nt, out, cyc, gs);
        run<1>("LDS rd -> add -> wr -> barrier (f32)", nt, out, cyc, gs);
        run<2>("LDS rd -> add -> wr -> barrier (f64)", nt, out, cyc, gs);
        run<3>("  + 8 dependent v_add_f32", nt, out, cyc, gs);
        run<13>("  + 8 dependent v_fma_f64", nt, out, cyc, gs);
        run<14>("  + 4 dependent v_exp_f32 (+mul)", nt, out, cyc, gs);
        run<4>("  + exp2 -> log2", nt, out, cyc, gs);
        run<8>("log-domain 4 arcs/lane f32", nt, out, cyc, gs);
        run<5>("linear 4 arcs/lane f64 fma chain", nt, out, cyc, gs);
        run<9>("linear 4 arcs/lane f64 tree", nt, out, cyc, gs);
        run<7>("f32 step + global store", nt, out, cyc, gs);
        run<10>("linear f64 chain + global f64 store", nt, out, cyc, gs);
        if (nt == 64) run<6>("single wave, waitcnt only (no barrier)", nt, out, cyc, gs);
    }
    return 0;
}
