// dependent-chain latency of fp64 FMA / rcp / 32-bit VALU as a function of how many waves of the workgroup are running (the others wait at a barrier)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(double* out, long long* t, int reps, int active_waves) {
    const int tid = threadIdx.x, wid = tid >> 6;
    double x = 1.0 + tid * 1e-9, y = 0.999;
    int a = tid, b = tid * 3;
    long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    __syncthreads();
    if (wid < active_waves) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < reps; i++) { x = fma(x, y, 1e-9); x = fma(x, y, 1e-9); x = fma(x, y, 1e-9); x = fma(x, y, 1e-9); }
        asm volatile("" : "+v"(x)); t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < reps; i++) { x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); }
        asm volatile("" : "+v"(x)); t2 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < reps; i++) { a = a * 3 + b; b = b ^ (a >> 3); a = a + (b << 1); b = b + 7 * a; }
        asm volatile("" : "+v"(a), "+v"(b)); t3 = __builtin_amdgcn_s_memtime();
    }
    __syncthreads();
    if (tid == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; }
    out[tid] = x + a + b;
}
int main() {
    double* out; long long* t; (void)hipMalloc(&out, 512 * 8); (void)hipMalloc(&t, 16 * 8);
    const int reps = 256;
    for (int aw : {8, 4, 2, 1}) {
        for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, t, reps, aw);
        long long h[3]; (void)hipMemcpy(h, t, 3 * 8, hipMemcpyDeviceToHost);
        printf("active waves %d: dependent fma f64 %.1f  rcp f64 %.1f  int chain (per op, ~6 ops/4 stmts) %.1f cycles\n", aw, (double)h[0] / (reps * 4), (double)h[1] / (reps * 4), (double)h[2] / (reps * 4));
    }
    return 0;
}
