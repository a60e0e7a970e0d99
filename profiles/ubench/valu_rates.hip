// Micro-benchmark: sustained issue rate of the vector-ALU instructions the trace kernel is made of, with every
// CU busy (W waves per SIMD).  Cycles come from the shader clock (s_memtime) of one wave; the core clock actually
// held is derived from s_memtime against the 100 MHz s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define N_ITER 2048
#define UNROLL 16

typedef float float2v __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(256) rate(unsigned long long *stamps, double seed, float fseed) {
    double a[8], c = seed * 1.0000001, e = seed * 0.99999;
    float f[8], fc = fseed * 1.0001f, fe = fseed * 0.999f;
    float2v p[8], pc = {fc, fc}, pe = {fe, fe};
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; f[i] = fseed + i + threadIdx.x; p[i] = float2v{f[i], f[i] + 1}; }
    unsigned long long t0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            int i = u & 7;
            if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(e));
            if (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fc), "v"(fe));
            if (OP == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc), "v"(pe));
            if (OP == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fc));
            if (OP == 6) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a[i]), "v"(c) : "vcc");
            if (OP == 7) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(f[i]), "v"(fc) : "vcc");
            if (OP == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if (OP == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if (OP == 10) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[i]) : "v"(fc), "v"(fe));
            if (OP == 11) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fc));
            if (OP == 12) asm volatile("v_cmp_nlt_f32_e64 s[20:21], %0, %1" :: "v"(f[i]), "v"(fc) : "s20", "s21");
            if (OP == 13) asm volatile("v_mov_b32 %0, %1" : "=v"(f[i]) : "v"(fc));
            if (OP == 14) asm volatile("v_fma_f32 %0, %0, %0, %1" : "+v"(f[i]) : "v"(fe));
            if (OP == 15) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(e));
            if (OP == 16) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(f[i]) : "s"(fc));
            if (OP == 17) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "s"(fc), "v"(fe));
        }
    }
    unsigned long long t1 = clock64(), r1 = wall_clock64();
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + f[i] + p[i].x + p[i].y;
    if (s == 12345.678) stamps[2] = (unsigned long long)s;
    if (blockIdx.x == 7 && threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

template <int OP> void run(const char *name, int wavesPerSimd) {
    unsigned long long *st; hipMalloc(&st, 64);
    int blocks = 256 * wavesPerSimd;
    for (int rep = 0; rep < 3; ++rep) rate<OP><<<blocks, 256>>>(st, 1.5, 1.5f);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    double cyclesPerInstrPerWave = (double)h[0] / ((double)N_ITER * UNROLL);
    double ghz = (double)h[0] / ((double)h[1] * 10.0);  // realtime ticks are 10 ns
    printf("%-22s %d waves/SIMD: %6.2f cycles per wave-instruction per SIMD  (clock %.2f GHz)\n", name, wavesPerSimd,
           cyclesPerInstrPerWave / wavesPerSimd, ghz);
    hipFree(st);
}

int main() {
    for (int w = 2; w <= 4; w += 2) {
        run<0>("v_add_f64", w); run<1>("v_mul_f64", w); run<2>("v_fma_f64 (3 regs)", w); run<15>("v_fmac_f64", w);
        run<5>("v_add_f32", w); run<11>("v_mul_f32", w); run<16>("v_sub_f32 (sgpr src)", w);
        run<10>("v_fmac_f32", w); run<3>("v_fma_f32 (3 regs)", w); run<14>("v_fma_f32 (a,a,c)", w); run<17>("v_fma_f32 (sgpr src)", w);
        run<4>("v_pk_fma_f32", w); run<8>("v_pk_add_f32", w); run<9>("v_pk_mul_f32", w);
        run<6>("v_cmp_lt_f64 vcc", w); run<7>("v_cmp_lt_f32 vcc", w); run<12>("v_cmp_nlt_f32 sgpr", w); run<13>("v_mov_b32", w);
    }
    return 0;
}
