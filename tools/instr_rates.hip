// instr_rates.hip -- issue cost of the VALU instruction classes in k_bucket_accumulate's loop on gfx950, relative to a
// plain v_add_u32 (not part of the library).  Every kernel runs 8 independent dependency chains of ONE instruction
// (inline asm, so the compiler cannot substitute another) at 2 waves per SIMD, the occupancy of the accumulation kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/instr_rates tools/instr_rates.hip ; prints one JSON line per class.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int CH = 8;
constexpr int UNROLL = 16;

enum Op { MAD_I64_I32, ASHR_I64, LSHL_ADD_U64, MUL_LO_U32, BFE_I32, ADD_U32, MOV_B32, ALIGNBIT, ASHR_I32, ADD3_U32, MAD_U64_U32, LSHLREV_B64, MAD_I32_I24 };

template <int OP>
__global__ void __launch_bounds__(256) k_rate(unsigned long long* out, int iters) {
    unsigned int a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 77u;
    long long acc[CH];
    int x[CH];
#pragma unroll
    for (int k = 0; k < CH; k++) {
        acc[k] = (long long)a * (k + 3);
        x[k] = (int)(b + k);
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < UNROLL; r++) {
#pragma unroll
            for (int k = 0; k < CH; k++) {
                if (OP == MAD_I64_I32) asm volatile("v_mad_i64_i32 %0, s[2:3], %1, %2, %0" : "+v"(acc[k]) : "v"(x[k]), "v"((int)a) : "s2", "s3");
                if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %0" : "+v"(acc[k]) : "v"(x[k]), "v"((int)a) : "s2", "s3");
                if (OP == ASHR_I64) asm volatile("v_ashrrev_i64 %0, 3, %0" : "+v"(acc[k]));
                if (OP == LSHLREV_B64) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(acc[k]));
                if (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[k]) : "v"(acc[(k + 1) % CH]));
                if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[k]) : "v"((int)(a | 1)));
                if (OP == BFE_I32) asm volatile("v_bfe_i32 %0, %0, 1, 30" : "+v"(x[k]));
                if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[k]) : "v"((int)a));
                if (OP == ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[k]) : "v"((int)a), "v"((int)b));
                if (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(x[k]) : "v"(x[(k + 1) % CH]));
                if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 30" : "+v"(x[k]) : "v"((int)a));
                if (OP == ASHR_I32) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(x[k]));
                if (OP == MAD_I32_I24) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x[k]) : "v"((int)a), "v"((int)b));
            }
        }
    }
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < CH; k++) s ^= (unsigned long long)acc[k] ^ (unsigned int)x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static double run(const char* name, unsigned long long* out, int cus, int wps, double base) {
    const int iters = 1000, grid = cus * wps;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<OP>, dim3(grid), dim3(256), 0, 0, out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_rate<OP>, dim3(grid), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    const double n = (double)grid * 256 * iters * UNROLL * CH;
    const double tps = n / ms / 1e9;
    printf("{\"instr\": \"%s\", \"waves_per_simd\": %d, \"T_lane_ops_per_s\": %.2f, \"cost_vs_v_add_u32\": %.2f}\n", name, wps, tps,
           base > 0 ? base / tps : 1.0);
    fflush(stdout);
    return tps;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long* out;
    CHECK(hipMalloc(&out, sizeof(unsigned long long) * 256 * 8 * 1024));
    for (int wps : {2, 1}) {
        const double base = run<ADD_U32>("v_add_u32", out, cus, wps, 0);
        run<MAD_I64_I32>("v_mad_i64_i32", out, cus, wps, base);
        run<MAD_U64_U32>("v_mad_u64_u32", out, cus, wps, base);
        run<ASHR_I64>("v_ashrrev_i64", out, cus, wps, base);
        run<LSHLREV_B64>("v_lshlrev_b64", out, cus, wps, base);
        run<LSHL_ADD_U64>("v_lshl_add_u64", out, cus, wps, base);
        run<MUL_LO_U32>("v_mul_lo_u32", out, cus, wps, base);
        run<BFE_I32>("v_bfe_i32", out, cus, wps, base);
        run<ADD3_U32>("v_add3_u32", out, cus, wps, base);
        run<MOV_B32>("v_mov_b32", out, cus, wps, base);
        run<ALIGNBIT>("v_alignbit_b32", out, cus, wps, base);
        run<ASHR_I32>("v_ashrrev_i32", out, cus, wps, base);
        run<MAD_I32_I24>("v_mad_i32_i24", out, cus, wps, base);
    }
    return 0;
}
