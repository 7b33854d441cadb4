// count_seq.hip -- cost of the per-point bookkeeping of the score kernel's stage 2 (SIMD cycles per POINT, wave64) in the
// forms the compiler emits / could emit, at 1, 2, 4 and 8 waves per SIMD.   hipcc -O3 --offload-arch=gfx950 count_seq.hip -o count_seq
//   cmp_cnd_addc  v_cmp_lt_f32 vcc ; v_cndmask_b32 ; v_cmp_lt_f32 vcc ; v_addc_co_u32      (two points: today's count)
//   cmp_addc      v_cmp_lt_f32 vcc ; v_addc_co_u32 w, vcc, w, w, vcc                       (one point: today's mask word)
//   sub_alignbit  v_sub_f32 x, 0.5, t ; v_alignbit_b32 w, w, x, 31                         (one point: sign bit shifted in)
//   clamp_add     v_fma_f32 s, t, H, -H/2 clamp ; v_add_f32 c, c, s                        (one point: 0/1 as a float)
//   fma6          six v_fma_f32 (the plane classifier's arithmetic, for scale)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
enum { CMP_CND_ADDC, CMP_ADDC, SUB_ALIGNBIT, CLAMP_ADD, FMA6, CNDMASK_ONLY, CMP_ONLY, ADDC_ONLY, ALIGNBIT_ONLY, MIN3_ABS, SUBABS, SUB32, MIN2, MINABS, FMAC, FMANEG, MIN3, BCNT, LSHLOR, MAX2, MUL_E64, FMACLAMP, MOV, MBCNT, READLANE, READFIRST, AND32, LSHL32, ADDU32, ADD3, CNDMASK_SGPR, CVTF64, DPPMOV, NOPS };
static const char *names[NOPS] = { "cmp_cnd_addc/2", "cmp_addc", "sub_alignbit", "clamp_add", "fma6", "cndmask", "cmp_vcc", "addc_vcc", "alignbit", "min3_abs/2", "v_sub_e64_abs", "v_sub_e32", "v_min_e32", "v_min_e64_abs", "v_fmac_e32", "v_fma_neg_abs", "v_min3", "v_bcnt", "v_lshl_or", "v_max_e32", "v_mul_e64_neg", "v_fma_clamp", "v_mov_b32", "v_mbcnt_lo+hi/2", "v_readlane", "v_readfirstlane", "v_and_b32", "v_lshlrev_b32", "v_add_u32", "v_add3_u32", "v_cndmask_sgpr", "v_cvt_f32_f64", "v_mov_dpp" };
static const double per[NOPS] = { 2, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1 };   // points per sequence

template <int OP> __device__ __forceinline__ void one(float &t, float &t2, unsigned &w, float &c, float k, float m)
{
    if constexpr (OP == CMP_CND_ADDC) asm volatile("v_cmp_lt_f32 vcc, 0.5, %1\n v_cndmask_b32 v100, 0, 1, vcc\n v_cmp_lt_f32 vcc, 0.5, %2\n s_nop 0\n v_addc_co_u32 %0, vcc, %0, v100, vcc" : "+v"(w) : "v"(t), "v"(t2) : "vcc", "v100");
    if constexpr (OP == CMP_ADDC) asm volatile("v_cmp_lt_f32 vcc, 0.5, %1\n s_nop 1\n v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(t) : "vcc");
    if constexpr (OP == SUB_ALIGNBIT) asm volatile("v_sub_f32 v100, 0.5, %1\n v_alignbit_b32 %0, %0, v100, 31" : "+v"(w) : "v"(t) : "v100");
    if constexpr (OP == CLAMP_ADD) asm volatile("v_fma_f32 v100, %1, %2, %3 clamp\n v_add_f32 %0, %0, v100" : "+v"(c) : "v"(t), "v"(k), "v"(m) : "v100");
    if constexpr (OP == FMA6) asm volatile("v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %0, %1, %3, %0\n v_fma_f32 %0, %2, %3, %0\n v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %0, %1, %3, %0\n v_fma_f32 %0, %2, %3, %0" : "+v"(c) : "v"(t), "v"(k), "v"(m));
    if constexpr (OP == CNDMASK_ONLY) asm volatile("v_cndmask_b32 %0, 0, 1, vcc" : "=v"(w) : : );
    if constexpr (OP == CMP_ONLY) asm volatile("v_cmp_lt_f32 vcc, 0.5, %0" : : "v"(t) : "vcc");
    if constexpr (OP == ADDC_ONLY) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(w) : : "vcc");
    if constexpr (OP == ALIGNBIT_ONLY) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(w) : "v"(t));
    if constexpr (OP == MIN3_ABS) asm volatile("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(c) : "v"(t), "v"(t2));
    if constexpr (OP == SUBABS) asm volatile("v_sub_f32_e64 %0, %1, |%0|" : "+v"(c) : "v"(t));
    if constexpr (OP == SUB32) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(c) : "v"(t));
    if constexpr (OP == MIN2) asm volatile("v_min_f32_e32 %0, %1, %0" : "+v"(c) : "v"(t));
    if constexpr (OP == MINABS) asm volatile("v_min_f32_e64 %0, |%1|, %0" : "+v"(c) : "v"(t));
    if constexpr (OP == FMAC) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(c) : "v"(t), "v"(t2));
    if constexpr (OP == FMANEG) asm volatile("v_fma_f32 %0, -|%1|, %2, %0" : "+v"(c) : "v"(t), "v"(t2));
    if constexpr (OP == MIN3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(c) : "v"(t), "v"(t2));
    if constexpr (OP == BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(w) : "v"(t));
    if constexpr (OP == LSHLOR) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(w) : "v"(t));
    if constexpr (OP == MAX2) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(c) : "v"(t));
    if constexpr (OP == MUL_E64) asm volatile("v_mul_f32_e64 %0, -%1, %0" : "+v"(c) : "v"(t));
    if constexpr (OP == FMACLAMP) asm volatile("v_fma_f32 %0, %1, %2, %0 clamp" : "+v"(c) : "v"(t), "v"(t2));
    // round 5: the classes tools/isa_account.py had priced by analogy (moves, lane operations, plain 32-bit integer)
    if constexpr (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(w) : "v"(t));
    if constexpr (OP == MBCNT) asm volatile("v_mbcnt_lo_u32_b32 %0, s10, 0\n v_mbcnt_hi_u32_b32 %0, s11, %0" : "=v"(w) : : "s10", "s11");
    if constexpr (OP == READLANE) asm volatile("v_readlane_b32 s10, %0, 3" : : "v"(w) : "s10");
    if constexpr (OP == READFIRST) asm volatile("v_readfirstlane_b32 s10, %0" : : "v"(w) : "s10");
    if constexpr (OP == AND32) asm volatile("v_and_b32 %0, %1, %0" : "+v"(w) : "v"(t));
    if constexpr (OP == LSHL32) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(w));
    if constexpr (OP == ADDU32) asm volatile("v_add_u32 %0, %1, %0" : "+v"(w) : "v"(t));
    if constexpr (OP == ADD3) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(w) : "v"(t), "v"(t2));
    if constexpr (OP == CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[10:11]" : "=v"(w) : "v"(t), "v"(t2) : "s10", "s11");
    if constexpr (OP == CVTF64) asm volatile("v_cvt_f32_f64 %0, v[100:101]" : "=v"(c) : : "v100", "v101");
    if constexpr (OP == DPPMOV) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(w) : "v"(t));
}

template <int OP> __global__ void __launch_bounds__(256) kern(float *out, int iters)
{
    float t[8], t2[8], c[8]; unsigned w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { t[j] = 0.001f * threadIdx.x + j; t2[j] = 0.7f - 0.001f * threadIdx.x; c[j] = 0.f; w[j] = threadIdx.x + j; }
    const float k = 1.2676506e30f, m = -0.5f * 1.2676506e30f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int j = 0; j < 8; j++) one<OP>(t[j], t2[j], w[j], c[j], k, m);
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s += c[j] + (float)w[j];
    if (s == 12345.0f) out[0] = 1;
}

typedef void (*kern_t)(float *, int);
template <int OP> void fill(kern_t *tb) { tb[OP] = kern<OP>; if constexpr (OP + 1 < NOPS) fill<OP + 1>(tb); }

int main()
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
    printf("%-16s %8s %8s %8s %8s   SIMD cycles per point at w waves per SIMD\n", "sequence", "w=1", "w=2", "w=4", "w=8");
    for (int op = 0; op < NOPS; op++) {
        printf("%-16s", names[op]);
        for (int w = 1; w <= 8; w *= 2) {
            const int blocks = cus * w;
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
            printf(" %8.2f", best * 1e-3 * ghz * 1e9 / ((double)w * ITERS * 64.0 * per[op]));
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
