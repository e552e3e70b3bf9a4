// micro-benchmarks of the latencies the dense phase of the IPM is built from (gfx950): dependent fp64 FMA / rcp chains, dependent and independent
// v_mfma_f64_16x16x4, MFMA result -> VALU use, LDS write -> read round trip, s_barrier with 8 waves.  One workgroup; prints cycles (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(double* out, long long* t, int reps) {
    __shared__ double lds[2048];
    const int tid = threadIdx.x;
    double x = 1.0 + tid * 1e-9, y = 0.999;
    long long t0, t1;
    // 1. dependent FMA chain
    __syncthreads(); t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; i++) { x = fma(x, y, 1e-9); x = fma(x, y, 1e-9); x = fma(x, y, 1e-9); x = fma(x, y, 1e-9); }
    asm volatile("" : "+v"(x)); t1 = __builtin_amdgcn_s_memtime(); if (tid == 0) t[0] = (t1 - t0);
    // 2. dependent rcp chain
    __syncthreads(); t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; i++) { x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); }
    asm volatile("" : "+v"(x)); t1 = __builtin_amdgcn_s_memtime(); if (tid == 0) t[1] = (t1 - t0);
    // 3. dependent MFMA chain (same accumulator)
    v4 c = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0}, c4 = {0, 0, 0, 0};
    __syncthreads(); t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; i++) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0);
                                     c = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); }
    asm volatile("" : "+v"(c)); t1 = __builtin_amdgcn_s_memtime(); if (tid == 0) t[2] = (t1 - t0);
    // 4. independent MFMAs (4 accumulators), all 8 waves
    __syncthreads(); t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; i++) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c2, 0, 0, 0);
                                     c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c3, 0, 0, 0); c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c4, 0, 0, 0); }
    asm volatile("" : "+v"(c), "+v"(c2), "+v"(c3), "+v"(c4)); t1 = __builtin_amdgcn_s_memtime(); if (tid == 0) t[3] = (t1 - t0);
    // 5. one MFMA then use of its result by the VALU (latency issue -> readable), wave 0 only measured but all waves run
    __syncthreads(); t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; i++) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); x = fma(c[0], 1e-300, x); x = fma(x, y, 1e-9); x = fma(x, y, 1e-9); x = fma(x, y, 1e-9); }
    asm volatile("" : "+v"(x)); t1 = __builtin_amdgcn_s_memtime(); if (tid == 0) t[4] = (t1 - t0);
    // 6. LDS write -> read round trip (same wave)
    __syncthreads(); t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; i++) { lds[tid] = x; x = lds[(tid + 1) & 511 & ~63 | (tid & 63)]; lds[tid + 512] = x; x = lds[512 + tid]; lds[tid + 1024] = x; x = lds[1024 + tid]; lds[tid + 1536] = x; x = lds[1536 + tid]; }
    asm volatile("" : "+v"(x)); t1 = __builtin_amdgcn_s_memtime(); if (tid == 0) t[5] = (t1 - t0);
    // 7. barriers
    __syncthreads(); t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; i++) { __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); }
    t1 = __builtin_amdgcn_s_memtime(); if (tid == 0) t[6] = (t1 - t0);
    // 8. only wave 0 issues MFMAs (others idle at the barrier): one wave's issue rate on its own
    __syncthreads(); t0 = __builtin_amdgcn_s_memtime();
    if (tid < 64) for (int i = 0; i < reps; i++) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c2, 0, 0, 0);
                                     c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c3, 0, 0, 0); c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c4, 0, 0, 0); }
    asm volatile("" : "+v"(c), "+v"(c2), "+v"(c3), "+v"(c4)); t1 = __builtin_amdgcn_s_memtime(); if (tid == 0) t[7] = (t1 - t0);
    // 9. v_cndmask / int op issue: independent 32-bit ops
    int a = tid, b = tid * 3;
    __syncthreads(); t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; i++) { a = a * 3 + b; b = b ^ (a >> 3); a = a + (b << 1); b = b + 7 * a; }
    asm volatile("" : "+v"(a), "+v"(b)); t1 = __builtin_amdgcn_s_memtime(); if (tid == 0) t[8] = (t1 - t0);
    out[tid] = x + c[0] + c2[1] + c3[2] + c4[3] + a + b;
}
int main() {
    double* out; long long* t; hipMalloc(&out, 512 * 8); hipMalloc(&t, 16 * 8);
    const int reps = 256;
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, t, reps);
    long long h[16]; hipMemcpy(h, t, 16 * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"dependent fma f64", "dependent rcp f64", "dependent mfma f64 16x16x4 (same acc), 8 waves", "independent mfma x4, 8 waves", "mfma -> valu use + 3 fma", "lds write->read round trip", "s_barrier (8 waves)", "independent mfma x4, ONE wave", "dependent int ops"};
    for (int i = 0; i < 9; i++) printf("%-55s %8.1f cycles per op\n", names[i], (double)h[i] / (reps * 4));
    return 0;
}
