// Cost model of ONE wave alone on a SIMD (gfx950): cycles per instruction for the instruction mixes of klt.hip's LK iteration.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_probe tools/probes/issue_probe.hip && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP24(x) REP8(x) REP8(x) REP8(x)
template <int V>
__global__ void __launch_bounds__(64) k(unsigned long long *out, float *sink, int iters) {
    float a = threadIdx.x * 1.0f, b = a + 1.f, c = a + 2.f, f = 0.5f;
    int s0 = iters, s1 = 3;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        if (V == 0) {   // 24 DPP adds, three interleaved dependent streams
            REP8(asm volatile("v_add_f32_dpp %0, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_add_f32_dpp %1, %1, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_add_f32_dpp %2, %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" : "+v"(a), "+v"(b), "+v"(c) : "v"(f));)
        } else if (V == 1) {   // 24 DPP adds, one dependent stream (s_nop 1 between: the hazard)
            REP24(asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\ns_nop 1\n" : "+v"(a) : "v"(f));)
        } else if (V == 2) {   // 24 dependent plain adds
            REP24(asm volatile("v_add_f32 %0, %0, %1\n" : "+v"(a) : "v"(f));)
        } else if (V == 3) {   // 24 independent plain adds (3 streams)
            REP8(asm volatile("v_add_f32 %0, %0, %3\nv_add_f32 %1, %1, %3\nv_add_f32 %2, %2, %3\n" : "+v"(a), "+v"(b), "+v"(c) : "v"(f));)
        } else if (V == 4) {   // 24 dependent scalar adds
            REP24(asm volatile("s_add_u32 %0, %0, %1\n" : "+s"(s0) : "s"(s1));)
        } else if (V == 5) {   // 24 x (v_add; s_nop 0)
            REP24(asm volatile("v_add_f32 %0, %0, %1\ns_nop 0\n" : "+v"(a) : "v"(f));)
        } else if (V == 6) {   // 24 dependent v_mad_i32_i24
            int x = __float_as_int(a);
            REP24(asm volatile("v_mad_i32_i24 %0, %0, %1, %0\n" : "+v"(x) : "v"(s1));)
            a = __int_as_float(x);
        } else if (V == 7) {   // 12 x (v_add ; s_add): VALU / SALU alternating, independent
            REP8(asm volatile("v_add_f32 %0, %0, %3\ns_add_u32 %2, %2, 1\nv_add_f32 %1, %1, %3\n" : "+v"(a), "+v"(b), "+s"(s0) : "v"(f));)
        } else if (V == 8) {   // 24 dependent v_pk_mul_f32
            float2 p = make_float2(a, b);
            REP24(asm volatile("v_pk_mul_f32 %0, %0, %1\n" : "+v"(p) : "v"(make_float2(f, f)));)
            a = p.x; b = p.y;
        } else if (V == 9) {   // v_readlane -> s_nop -> valu use, 8 times (3 instr each)
            REP8(asm volatile("v_readlane_b32 %1, %0, 63\ns_nop 1\nv_add_f32 %0, %1, %0\n" : "+v"(a), "+s"(s0) : );)
        } else if (V == 10) {   // 24 x v_cvt_f32_i32 dependent-ish (cvt is it full rate?)
            REP24(asm volatile("v_cvt_f32_i32 %0, %0\n" : "+v"(a));)
        } else if (V == 11) {   // 24 x v_rndne_f32
            REP24(asm volatile("v_rndne_f32 %0, %0\n" : "+v"(a));)
        } else if (V == 12) {   // 24 x v_floor/v_cmp + branchless: v_cmp_neq_f32 + s_or
            REP24(asm volatile("v_cmp_neq_f32 vcc, %0, %1\n" : : "v"(a), "v"(f) : "vcc");)
        } else if (V == 13) {   // 24 x s_waitcnt lgkmcnt(0) (already satisfied) between v_adds: 48 instr
            REP24(asm volatile("v_add_f32 %0, %0, %1\ns_waitcnt lgkmcnt(0)\n" : "+v"(a) : "v"(f));)
        }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[2 * V] = t1 - t0;
        out[2 * V + 1] = w1 - w0;
    }
    sink[threadIdx.x] = a + b + c + (float) s0;
}
int main() {
    unsigned long long *out;
    float *sink;
    hipMalloc(&out, 64 * 8);
    hipMalloc(&sink, 256);
    const int iters = 20000;
    const char *names[] = {"24 dpp adds, 3 streams", "24 x (dpp add dependent + s_nop 1)", "24 dependent v_add_f32", "24 v_add_f32, 3 streams", "24 dependent s_add_u32",
                           "24 x (v_add_f32 + s_nop 0)", "24 dependent v_mad_i32_i24", "8 x (v_add, s_add, v_add)", "24 dependent v_pk_mul_f32",
                           "8 x (v_readlane, s_nop 1, v_add)", "24 dependent v_cvt_f32_i32", "24 dependent v_rndne_f32", "24 v_cmp_neq_f32", "24 x (v_add_f32 + s_waitcnt)"};
#define RUN(V) hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, out, sink, iters); hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, out, sink, iters);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13)
    hipDeviceSynchronize();
    unsigned long long h[64];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    for (int v = 0; v < 14; v++)
        printf("%-40s %7.1f clk/iter  %6.2f clk/instr(24)   clock %.0f MHz\n", names[v], (double) h[2 * v] / iters, (double) h[2 * v] / iters / 24.0,
               (double) h[2 * v] / ((double) h[2 * v + 1] / 100.0));
    return 0;
}
