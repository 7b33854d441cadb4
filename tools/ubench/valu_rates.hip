// valu_rates.hip -- issue cost (SIMD cycles per wave64 instruction) of the vector instructions the score kernel is made
// of, at 1, 2, 4 and 8 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
enum { FMA64, ADD64, MUL64, MAX64, FMA32, ADD32, MUL32, PKFMA, PKMUL, PKADD, RCP64, RSQ64, SQRT64, RCP32, RSQ32, SQRT32, AND32,
       CMP64, CMP32, CNDMASK, CVT3264, CVT6432, SADD, SAND64, SBCNT, SFF1, MIX_V32_S, MIX_V64_S, MIX_V32_2S, NOPS };
static const char *names[NOPS] = { "v_fma_f64", "v_add_f64", "v_mul_f64", "v_max_f64", "v_fma_f32", "v_add_f32", "v_mul_f32",
    "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_rcp_f32", "v_rsq_f32",
    "v_sqrt_f32", "v_and_b32", "v_cmp_gt_f64", "v_cmp_gt_f32", "v_cndmask_b32", "v_cvt_f32_f64", "v_cvt_f64_f32",
    "s_add_u32", "s_and_b64", "s_bcnt1_b64", "s_ff1_b64", "fma32+s_add", "fma64+s_add", "fma32+2s_add" };

template <int OP> __device__ __forceinline__ void one(double &d, float &f, f2 &p, unsigned &u, unsigned long long &acc, unsigned &su, unsigned &su2, unsigned long long &sq,
                                                      double kd, double md, float kf, float mf, f2 kp, f2 mp)
{
    if constexpr (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(kd), "v"(md));
    if constexpr (OP == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(kd));
    if constexpr (OP == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(kd));
    if constexpr (OP == MAX64) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d) : "v"(kd));
    if constexpr (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(kf), "v"(mf));
    if constexpr (OP == ADD32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f) : "v"(kf));
    if constexpr (OP == MUL32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f) : "v"(kf));
    if constexpr (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(kp), "v"(mp));
    if constexpr (OP == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(kp));
    if constexpr (OP == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(kp));
    if constexpr (OP == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d));
    if constexpr (OP == RSQ64) asm volatile("v_rsq_f64 %0, %0" : "+v"(d));
    if constexpr (OP == SQRT64) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d));
    if constexpr (OP == RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f));
    if constexpr (OP == RSQ32) asm volatile("v_rsq_f32 %0, %0" : "+v"(f));
    if constexpr (OP == SQRT32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f));
    if constexpr (OP == AND32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u) : "v"(0xfffffff7u));
    if constexpr (OP == CMP64) { unsigned long long s; asm volatile("v_cmp_gt_f64 %0, %1, %2" : "=s"(s) : "v"(d), "v"(kd)); acc ^= s; }
    if constexpr (OP == CMP32) { unsigned long long s; asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(s) : "v"(f), "v"(kf)); acc ^= s; }
    if constexpr (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u) : "v"(0x1234u));
    if constexpr (OP == CVT3264) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(d));
    if constexpr (OP == SADD) asm volatile("s_add_u32 %0, %0, 3" : "+s"(su) : : "scc");
    if constexpr (OP == SAND64) asm volatile("s_and_b64 %0, %0, %1" : "+s"(sq) : "s"(0xfffffffffffffff7ull) : "scc");
    if constexpr (OP == SBCNT) asm volatile("s_bcnt1_i32_b64 %0, %1" : "=s"(su) : "s"(sq) : "scc");
    if constexpr (OP == SFF1) asm volatile("s_ff1_i32_b64 %0, %1" : "=s"(su) : "s"(sq));
    if constexpr (OP == MIX_V32_S) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(kf), "v"(mf)); asm volatile("s_add_u32 %0, %0, 3" : "+s"(su) : : "scc"); }
    if constexpr (OP == MIX_V64_S) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(kd), "v"(md)); asm volatile("s_add_u32 %0, %0, 3" : "+s"(su) : : "scc"); }
    if constexpr (OP == MIX_V32_2S) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(kf), "v"(mf)); asm volatile("s_add_u32 %0, %0, 3" : "+s"(su) : : "scc"); asm volatile("s_add_u32 %0, %0, 5" : "+s"(su2) : : "scc"); }
    if constexpr (OP == CVT6432) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(f));
}

template <int OP> __global__ void __launch_bounds__(256) kern(float *out, int iters)
{
    double d[8]; float f[8]; f2 p[8]; unsigned u[8]; unsigned su[8], su2[8]; unsigned long long sq[8];
    unsigned long long acc = threadIdx.x == 999 ? 1 : 0x5555555555555555ull;
#pragma unroll
    for (int j = 0; j < 8; j++) { d[j] = threadIdx.x + j; f[j] = threadIdx.x + j; p[j].x = threadIdx.x + j; p[j].y = j; u[j] = threadIdx.x * 8 + j; su[j] = __builtin_amdgcn_readfirstlane(iters + j); su2[j] = su[j] + 1; sq[j] = ((unsigned long long)su[j] << 32) | 0xfff0fff0u; }
    const double kd = 1.0000001, md = 0.5; const float kf = 1.0000001f, mf = 0.5f; const f2 kp = { 1.0000001f, 1.0000002f }, mp = { 0.5f, 0.25f };
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int j = 0; j < 8; j++) one<OP>(d[j], f[j], p[j], u[j], acc, su[j], su2[j], sq[j], kd, md, kf, mf, kp, mp);
        }
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s += d[j] + f[j] + p[j].x + p[j].y + u[j] + su[j] + su2[j] + (double)sq[j];
    if (s == 12345.0 || acc == 12345ull) out[0] = 1;
}

typedef void (*kern_t)(float *, int);
template <int OP> void fill(kern_t *t) { t[OP] = kern<OP>; if constexpr (OP + 1 < NOPS) fill<OP + 1>(t); }

int main(int argc, char **argv)
{
    float *dbuf;
    CHK(hipMalloc(&dbuf, 4096));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("device %s CUs %d clock %.3f GHz\n", prop.name, cus, ghz);
    kern_t tab[NOPS];
    fill<0>(tab);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int ITERS = 4096;
    printf("%-14s %8s %8s %8s %8s   SIMD cycles per wave64 instruction at w waves per SIMD\n", "instr", "w=1", "w=2", "w=4", "w=8");
    for (int op = (argc > 1 ? atoi(argv[1]) : 0); op < NOPS; op++) {
        printf("%-14s", names[op]);
        for (int w = 1; w <= 8; w *= 2) {
            const int blocks = cus * w;   // 256 threads = one wave per SIMD
            hipLaunchKernelGGL(tab[op], dim3(blocks), dim3(256), 0, 0, dbuf, 64);
            CHK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                CHK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(tab[op], dim3(blocks), dim3(256), 0, 0, dbuf, ITERS);
                CHK(hipEventRecord(e1, 0));
                CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf(" %8.2f", best * 1e-3 * ghz * 1e9 / ((double)w * ITERS * 64.0));
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
