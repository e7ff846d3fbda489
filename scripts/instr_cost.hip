// What the f64 instructions of the surface-wave kernels cost on gfx950: clocks per wave-instruction on one SIMD, as a
// dependent chain (latency) and as 2 / 4 / 8 independent chains in one wavefront, and with 1 / 2 / 3 wavefronts per SIMD.
// (DESIGN.md section 4 prices the step against VALU issue; this is where the 4-clock figure and the cost of the
// transcendental seeds come from.)
//
//   hipcc -O3 --offload-arch=gfx950 scripts/instr_cost.hip -o gpurun_out/instr_cost && gpurun_out/instr_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum Op { FMA, MUL, ADD, RCP, RSQ, SQRT, RNDNE, LDEXP, SEL, CVT_RCP32, FMA32, RCP32, DIV, MIN, FREXP };
static const char* NAMES[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_rndne_f64",
                              "add + v_ldexp_f64 (2)", "cmp + 2 cndmask (3)", "cvt rcp_f32 cvt add (4)", "v_fma_f32", "v_rcp_f32",
                              "IEEE division f64 (~24)", "v_min_f64", "v_frexp_mant_f64"};

template <int OP>
__device__ __forceinline__ double step(double x, double a, double b) {
    if (OP == FMA) return __builtin_fma(x, a, b);
    if (OP == MUL) return x * a;
    if (OP == ADD) return x + b;
    if (OP == RCP) return __builtin_amdgcn_rcp(x);
    if (OP == RSQ) return __builtin_amdgcn_rsq(x);
    if (OP == SQRT) return __builtin_amdgcn_sqrt(x);
    if (OP == RNDNE) return __builtin_rint(x) + 0.0 * a;
    if (OP == LDEXP) return __builtin_amdgcn_ldexp(x + b, (int)a);
    if (OP == SEL) return x > a ? b : x;
    if (OP == CVT_RCP32) return (double)__builtin_amdgcn_rcpf((float)x) + b;
    if (OP == DIV) return a / x;
    if (OP == MIN) return __builtin_fmin(x, a);
    if (OP == FREXP) return __builtin_amdgcn_frexp_mant(x);
    return x;
}

template <int OP, int ILP>
__global__ void __launch_bounds__(64) k_chain(double* out, int n, double a, double b) {
    double x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = 1.0 + 1e-3 * (threadIdx.x + 64 * i);
    if (OP == FMA32 || OP == RCP32) {
        float y[ILP];
#pragma unroll
        for (int i = 0; i < ILP; i++) y[i] = (float)x[i];
        const float fa = (float)a, fb = (float)b;
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int i = 0; i < ILP; i++) y[i] = OP == FMA32 ? __builtin_fmaf(y[i], fa, fb) : __builtin_amdgcn_rcpf(y[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < ILP; i++) x[i] = y[i];
    } else {
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int i = 0; i < ILP; i++) x[i] = step<OP>(x[i], a, b);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += x[i];
    if (s == 123.456) out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int OP, int ILP>
static double run(double* d, int waves_per_simd, int n) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 1024 * waves_per_simd;
    hipLaunchKernelGGL((k_chain<OP, ILP>), dim3(grid), dim3(64), 0, 0, d, 16, 0.999999, 1e-9);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_chain<OP, ILP>), dim3(grid), dim3(64), 0, 0, d, n, 0.999999, 1e-9);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // clocks per chain step on one SIMD at 2.4 GHz (all wavefronts of the SIMD together); a step is one instruction except
    // where the row's name counts more
    return ms * 1e-3 * 2.4e9 / ((double)n * 8 * ILP * waves_per_simd);
}

template <int OP>
static void row(double* d) {
    const int n = 20000;
    printf("%-24s", NAMES[OP]);
    printf(" %7.1f", run<OP, 1>(d, 1, n));
    printf(" %7.1f", run<OP, 2>(d, 1, n));
    printf(" %7.1f", run<OP, 4>(d, 1, n));
    printf(" %7.1f", run<OP, 8>(d, 1, n));
    printf("  | %7.1f", run<OP, 1>(d, 2, n));
    printf(" %7.1f", run<OP, 1>(d, 3, n));
    printf(" %7.1f", run<OP, 4>(d, 2, n));
    printf(" %7.1f\n", run<OP, 4>(d, 3, n));
}

int main() {
    double* d;
    (void)hipMalloc(&d, 1024 * 8 * 64 * sizeof(double));
    printf("clocks per wave-instruction on one SIMD (2.4 GHz assumed; 1024 SIMDs, every SIMD loaded alike)\n");
    printf("%-24s %7s %7s %7s %7s  | %7s %7s %7s %7s\n", "", "1w ilp1", "ilp2", "ilp4", "ilp8", "2w ilp1", "3w ilp1", "2w ilp4", "3w ilp4");
    row<FMA>(d); row<MUL>(d); row<ADD>(d); row<MIN>(d); row<RCP>(d); row<RSQ>(d); row<SQRT>(d); row<RNDNE>(d); row<LDEXP>(d);
    row<FREXP>(d); row<SEL>(d); row<CVT_RCP32>(d); row<FMA32>(d); row<RCP32>(d); row<DIV>(d);
    (void)hipFree(d);
    return 0;
}
