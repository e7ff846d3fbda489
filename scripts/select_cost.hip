// What an f64 select costs on gfx950 (scripts/instr_cost.hip measured 48 clocks for compare + two v_cndmask): the pieces,
// written as inline assembly so that the compiler cannot rearrange them.  Clocks per BLOCK of 8 repetitions / 8, on one SIMD,
// with 1 .. 4 wavefronts per SIMD (2.4 GHz assumed).
//
//   hipcc -O3 --offload-arch=gfx950 scripts/select_cost.hip -o /tmp/select_cost && /tmp/select_cost
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(s) s s s s s s s s

enum { CMP64 = 0, CMP32, CND_VCC, CND_SGPR, CMP64_CND, CMP64_NOP_CND, CMP64S_CND, CMPI32_CND, CMP32_CND, FMA_REF, CMP64_FMA, CLASS64, CND_E64_VCC, CND_VCC_IND, CND_SGPR_IND, CMP_COPY_CND, CND_VCC_1, NK };
static const char* NAMES[NK] = {
    "v_cmp_lt_f64 vcc", "v_cmp_lt_f32 vcc", "v_cndmask x2 (vcc fixed)", "v_cndmask_e64 x2 (sgpr fixed)",
    "cmp_f64 vcc, s_nop 1, 2 cndmask", "cmp_f64 vcc, s_nop 7, 2 cndmask", "cmp_f64 -> sgpr + 2 cnd_e64", "cmp_lt_i32 (hi) + 2 cnd", "cmp_f32 + 2 cndmask",
    "v_fma_f64 (reference)", "cmp_f64 vcc + v_fma_f64", "v_cmp_class_f64 vcc", "v_cndmask_e64 x2 (vcc operand)",
    "cndmask vcc x2, independent dst", "cndmask_e64 sgpr x2, indep. dst", "cmp vcc, s_mov s[20:21], 2 cnd_e64", "ONE v_cndmask (vcc fixed)"};

// x lives in v[10:11], a in v[12:13], b in v[14:15] inside every block (fixed registers: AMDGPU inline assembly has no
// modifier for the halves of a 64-bit operand); the two v_mov pairs in front of the 8 repetitions are counted with them.
#define LOADXAB "v_mov_b64 v[10:11], %0\n v_mov_b64 v[12:13], %1\n v_mov_b64 v[14:15], %2\n"
#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "vcc", "s20", "s21"
template <int K>
__global__ void __launch_bounds__(64) k_sel(double* out, int n, double a, double b) {
    double x = 1.0 + 1e-3 * threadIdx.x;
    unsigned long long m = __ballot(threadIdx.x & 1);
    for (int it = 0; it < n; it++) {
        if (K == CMP64) asm volatile(LOADXAB REP8("v_cmp_lt_f64 vcc, v[10:11], v[12:13]\n") ::"v"(x), "v"(a), "v"(b) : CLOB);
        if (K == CMP32) asm volatile(LOADXAB REP8("v_cmp_lt_f32 vcc, v11, v13\n") ::"v"(x), "v"(a), "v"(b) : CLOB);
        if (K == CLASS64) asm volatile(LOADXAB REP8("v_cmp_class_f64 vcc, v[10:11], 3\n") ::"v"(x), "v"(a), "v"(b) : CLOB);
        if (K == CND_VCC)
            asm volatile(LOADXAB "v_cmp_lt_f64 vcc, v[10:11], v[12:13]\n s_nop 4\n"
                         REP8("v_cndmask_b32 v10, v10, v14, vcc\n v_cndmask_b32 v11, v11, v15, vcc\n") ::"v"(x), "v"(a), "v"(b) : CLOB);
        if (K == CND_SGPR)
            asm volatile(LOADXAB REP8("v_cndmask_b32_e64 v10, v10, v14, %3\n v_cndmask_b32_e64 v11, v11, v15, %3\n")
                         ::"v"(x), "v"(a), "v"(b), "s"(m) : CLOB);
        if (K == CMP64_CND)
            asm volatile(LOADXAB REP8("v_cmp_lt_f64 vcc, v[10:11], v[12:13]\n s_nop 1\n v_cndmask_b32 v10, v10, v14, vcc\n v_cndmask_b32 v11, v11, v15, vcc\n")
                         ::"v"(x), "v"(a), "v"(b) : CLOB);
        if (K == CMP64_NOP_CND)
            asm volatile(LOADXAB REP8("v_cmp_lt_f64 vcc, v[10:11], v[12:13]\n s_nop 7\n v_cndmask_b32 v10, v10, v14, vcc\n v_cndmask_b32 v11, v11, v15, vcc\n")
                         ::"v"(x), "v"(a), "v"(b) : CLOB);
        if (K == CMP64S_CND)
            asm volatile(LOADXAB REP8("v_cmp_lt_f64_e64 s[20:21], v[10:11], v[12:13]\n s_nop 1\n v_cndmask_b32_e64 v10, v10, v14, s[20:21]\n v_cndmask_b32_e64 v11, v11, v15, s[20:21]\n")
                         ::"v"(x), "v"(a), "v"(b) : CLOB);
        if (K == CMPI32_CND)
            asm volatile(LOADXAB REP8("v_cmp_lt_i32 vcc, v11, v13\n s_nop 1\n v_cndmask_b32 v10, v10, v14, vcc\n v_cndmask_b32 v11, v11, v15, vcc\n")
                         ::"v"(x), "v"(a), "v"(b) : CLOB);
        if (K == CMP32_CND)
            asm volatile(LOADXAB REP8("v_cmp_lt_f32 vcc, v11, v13\n s_nop 1\n v_cndmask_b32 v10, v10, v14, vcc\n v_cndmask_b32 v11, v11, v15, vcc\n")
                         ::"v"(x), "v"(a), "v"(b) : CLOB);
        if (K == CND_E64_VCC)
            asm volatile(LOADXAB "v_cmp_lt_f64 vcc, v[10:11], v[12:13]\n s_nop 4\n"
                         REP8("v_cndmask_b32_e64 v10, v10, v14, vcc\n v_cndmask_b32_e64 v11, v11, v15, vcc\n") ::"v"(x), "v"(a), "v"(b) : CLOB);
        if (K == CND_VCC_1)
            asm volatile(LOADXAB "v_cmp_lt_f64 vcc, v[10:11], v[12:13]\n s_nop 4\n"
                         REP8("v_cndmask_b32 v10, v10, v14, vcc\n") ::"v"(x), "v"(a), "v"(b) : CLOB);
        if (K == CND_VCC_IND)
            asm volatile(LOADXAB "v_cmp_lt_f64 vcc, v[10:11], v[12:13]\n s_nop 4\n"
                         "v_cndmask_b32 v16, v10, v14, vcc\n v_cndmask_b32 v17, v11, v15, vcc\n v_cndmask_b32 v18, v10, v14, vcc\n v_cndmask_b32 v19, v11, v15, vcc\n"
                         "v_cndmask_b32 v20, v10, v14, vcc\n v_cndmask_b32 v21, v11, v15, vcc\n v_cndmask_b32 v22, v10, v14, vcc\n v_cndmask_b32 v23, v11, v15, vcc\n"
                         "v_cndmask_b32 v16, v10, v14, vcc\n v_cndmask_b32 v17, v11, v15, vcc\n v_cndmask_b32 v18, v10, v14, vcc\n v_cndmask_b32 v19, v11, v15, vcc\n"
                         "v_cndmask_b32 v20, v10, v14, vcc\n v_cndmask_b32 v21, v11, v15, vcc\n v_cndmask_b32 v22, v10, v14, vcc\n v_cndmask_b32 v23, v11, v15, vcc\n"
                         ::"v"(x), "v"(a), "v"(b) : CLOB, "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
        if (K == CND_SGPR_IND)
            asm volatile(LOADXAB
                         "v_cndmask_b32_e64 v16, v10, v14, %3\n v_cndmask_b32_e64 v17, v11, v15, %3\n v_cndmask_b32_e64 v18, v10, v14, %3\n v_cndmask_b32_e64 v19, v11, v15, %3\n"
                         "v_cndmask_b32_e64 v20, v10, v14, %3\n v_cndmask_b32_e64 v21, v11, v15, %3\n v_cndmask_b32_e64 v22, v10, v14, %3\n v_cndmask_b32_e64 v23, v11, v15, %3\n"
                         "v_cndmask_b32_e64 v16, v10, v14, %3\n v_cndmask_b32_e64 v17, v11, v15, %3\n v_cndmask_b32_e64 v18, v10, v14, %3\n v_cndmask_b32_e64 v19, v11, v15, %3\n"
                         "v_cndmask_b32_e64 v20, v10, v14, %3\n v_cndmask_b32_e64 v21, v11, v15, %3\n v_cndmask_b32_e64 v22, v10, v14, %3\n v_cndmask_b32_e64 v23, v11, v15, %3\n"
                         ::"v"(x), "v"(a), "v"(b), "s"(m) : CLOB, "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
        if (K == CMP_COPY_CND)
            asm volatile(LOADXAB REP8("v_cmp_lt_f64 vcc, v[10:11], v[12:13]\n s_nop 1\n s_mov_b64 s[20:21], vcc\n v_cndmask_b32_e64 v10, v10, v14, s[20:21]\n v_cndmask_b32_e64 v11, v11, v15, s[20:21]\n")
                         ::"v"(x), "v"(a), "v"(b) : CLOB);
        if (K == FMA_REF) asm volatile(LOADXAB REP8("v_fma_f64 v[10:11], v[10:11], v[12:13], v[14:15]\n") ::"v"(x), "v"(a), "v"(b) : CLOB);
        if (K == CMP64_FMA)
            asm volatile(LOADXAB REP8("v_cmp_lt_f64 vcc, v[10:11], v[12:13]\n v_fma_f64 v[10:11], v[10:11], v[12:13], v[14:15]\n")
                         ::"v"(x), "v"(a), "v"(b) : CLOB);
    }
    if (x == 123.456) out[blockIdx.x * 64 + threadIdx.x] = x;
}

template <int K>
static double run(double* d, int waves_per_simd, int n) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 1024 * waves_per_simd;
    hipLaunchKernelGGL((k_sel<K>), dim3(grid), dim3(64), 0, 0, d, 16, 0.999999, 1e-9);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_sel<K>), dim3(grid), dim3(64), 0, 0, d, n, 0.999999, 1e-9);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 * 2.4e9 / ((double)n * 8 * waves_per_simd);
}

template <int K>
static void row(double* d) {
    const int n = 20000;
    printf("%-32s %7.1f %7.1f %7.1f %7.1f\n", NAMES[K], run<K>(d, 1, n), run<K>(d, 2, n), run<K>(d, 3, n), run<K>(d, 4, n));
}

int main() {
    double* d;
    (void)hipMalloc(&d, 1024 * 8 * 64 * sizeof(double));
    for (int i = 0; i < 40; i++) run<FMA_REF>(d, 4, 20000);      // clocks up before anything is timed
    printf("clocks per repetition on one SIMD, all its wavefronts together (2.4 GHz assumed)\n%-32s %7s %7s %7s %7s\n", "", "1 wave", "2", "3", "4");
    row<FMA_REF>(d); row<CMP64>(d); row<CMP32>(d); row<CLASS64>(d); row<CND_VCC>(d); row<CND_SGPR>(d); row<CMP64_CND>(d); row<CMP64_NOP_CND>(d);
    row<CMP64S_CND>(d); row<CMPI32_CND>(d); row<CMP32_CND>(d); row<CMP64_FMA>(d);
    row<CND_E64_VCC>(d); row<CND_VCC_1>(d); row<CND_VCC_IND>(d); row<CND_SGPR_IND>(d); row<CMP_COPY_CND>(d); row<FMA_REF>(d);
    (void)hipFree(d);
    return 0;
}
