// Development tool: issue cost (shader clocks per wave-instruction) of the vector instructions the
// path kernel is made of, one SIMD's view, at one wave per SIMD and at four.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_cost.hip -o /tmp/valu_cost && /tmp/valu_cost
// Each case issues 128 independent copies of one instruction per loop trip (8 destinations, 16 times);
// the time is that of the wave that finishes last.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP4(x) x x x x x x x x x x x x x x x x
#define CASE64_3(NAME, INS)                                                                      \
    __global__ void k_##NAME(unsigned long long* out, double seed, int trips) {                  \
        double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        double x = seed * 0.5 + 1.0, y = seed * 0.25 + 3.0;                                      \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        for (int i = 0; i < trips; ++i) {                                                        \
            REP4(asm volatile(INS " %0, %8, %9, %0\n" INS " %1, %8, %9, %1\n" INS " %2, %8, %9, %2\n" INS " %3, %8, %9, %3\n" \
                              INS " %4, %8, %9, %4\n" INS " %5, %8, %9, %5\n" INS " %6, %8, %9, %6\n" INS " %7, %8, %9, %7\n" \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                 \
                              : "v"(x), "v"(y));)                                                 \
        }                                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[1] = 1;                      \
        if ((threadIdx.x & 63) == 0) atomicMax(out, t1 - t0);                                                  \
    }
#define CASE64_2(NAME, INS)                                                                      \
    __global__ void k_##NAME(unsigned long long* out, double seed, int trips) {                  \
        double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        double x = seed * 0.5 + 1.0;                                                             \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        for (int i = 0; i < trips; ++i) {                                                        \
            REP4(asm volatile(INS " %0, %8, %0\n" INS " %1, %8, %1\n" INS " %2, %8, %2\n" INS " %3, %8, %3\n"                 \
                              INS " %4, %8, %4\n" INS " %5, %8, %5\n" INS " %6, %8, %6\n" INS " %7, %8, %7\n"                 \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                 \
                              : "v"(x));)                                                         \
        }                                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[1] = 1;                      \
        if ((threadIdx.x & 63) == 0) atomicMax(out, t1 - t0);                                                  \
    }
#define CASE64_SHIFT(NAME, INS)                                                                      \
    __global__ void k_##NAME(unsigned long long* out, double seed, int trips) {                  \
        double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        unsigned x = (unsigned)seed + 4u;                                                                    \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        for (int i = 0; i < trips; ++i) {                                                        \
            REP4(asm volatile(INS " %0, %8, %0\n" INS " %1, %8, %1\n" INS " %2, %8, %2\n" INS " %3, %8, %3\n"                 \
                              INS " %4, %8, %4\n" INS " %5, %8, %5\n" INS " %6, %8, %6\n" INS " %7, %8, %7\n"                 \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                 \
                              : "v"(x));)                                                         \
        }                                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[1] = 1;                      \
        if ((threadIdx.x & 63) == 0) atomicMax(out, t1 - t0);                                                  \
    }
#define CASE64_1(NAME, INS)                                                                      \
    __global__ void k_##NAME(unsigned long long* out, double seed, int trips) {                  \
        double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        for (int i = 0; i < trips; ++i) {                                                        \
            REP4(asm volatile(INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n"                                 \
                              INS " %4, %4\n" INS " %5, %5\n" INS " %6, %6\n" INS " %7, %7\n"                                 \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)              \
        }                                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[1] = 1;                      \
        if ((threadIdx.x & 63) == 0) atomicMax(out, t1 - t0);                                                  \
    }
#define CASE32_3(NAME, INS)                                                                      \
    __global__ void k_##NAME(unsigned long long* out, double seed, int trips) {                  \
        unsigned a0 = (unsigned)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned x = a0 * 3 + 1, y = a0 * 5 + 3;                                                 \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        for (int i = 0; i < trips; ++i) {                                                        \
            REP4(asm volatile(INS " %0, %8, %9, %0\n" INS " %1, %8, %9, %1\n" INS " %2, %8, %9, %2\n" INS " %3, %8, %9, %3\n" \
                              INS " %4, %8, %9, %4\n" INS " %5, %8, %9, %5\n" INS " %6, %8, %9, %6\n" INS " %7, %8, %9, %7\n" \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                 \
                              : "v"(x), "v"(y));)                                                 \
        }                                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345678u) out[1] = 1;                      \
        if ((threadIdx.x & 63) == 0) atomicMax(out, t1 - t0);                                                  \
    }
#define CASE32_2(NAME, INS)                                                                      \
    __global__ void k_##NAME(unsigned long long* out, double seed, int trips) {                  \
        unsigned a0 = (unsigned)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned x = a0 * 3 + 1;                                                                 \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        for (int i = 0; i < trips; ++i) {                                                        \
            REP4(asm volatile(INS " %0, %8, %0\n" INS " %1, %8, %1\n" INS " %2, %8, %2\n" INS " %3, %8, %3\n"                 \
                              INS " %4, %8, %4\n" INS " %5, %8, %5\n" INS " %6, %8, %6\n" INS " %7, %8, %7\n"                 \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                 \
                              : "v"(x));)                                                         \
        }                                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345678u) out[1] = 1;                      \
        if ((threadIdx.x & 63) == 0) atomicMax(out, t1 - t0);                                                  \
    }
#define CASE32_2T(NAME, INS, TAIL)                                                                      \
    __global__ void k_##NAME(unsigned long long* out, double seed, int trips) {                  \
        unsigned a0 = (unsigned)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned x = a0 * 3 + 1;                                                                 \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        for (int i = 0; i < trips; ++i) {                                                        \
            REP4(asm volatile(INS " %0, %8, %0" TAIL "\n" INS " %1, %8, %1" TAIL "\n" INS " %2, %8, %2" TAIL "\n" INS " %3, %8, %3" TAIL "\n"                 \
                              INS " %4, %8, %4" TAIL "\n" INS " %5, %8, %5" TAIL "\n" INS " %6, %8, %6" TAIL "\n" INS " %7, %8, %7" TAIL "\n"                 \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                 \
                              : "v"(x));)                                                         \
        }                                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345678u) out[1] = 1;                      \
        if ((threadIdx.x & 63) == 0) atomicMax(out, t1 - t0);                                                  \
    }
#define CASE32_1(NAME, INS)                                                                      \
    __global__ void k_##NAME(unsigned long long* out, double seed, int trips) {                  \
        float a0 = (float)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        for (int i = 0; i < trips; ++i) {                                                        \
            REP4(asm volatile(INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n"                                 \
                              INS " %4, %4\n" INS " %5, %5\n" INS " %6, %6\n" INS " %7, %7\n"                                 \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)              \
        }                                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[1] = 1;                     \
        if ((threadIdx.x & 63) == 0) atomicMax(out, t1 - t0);                                                  \
    }
// 64-bit destination, 32-bit sources (v_mad_u64_u32 d, s0, s1, d with its carry-out in vcc; conversions)
#define CASE_MAD64(NAME, INS)                                                                    \
    __global__ void k_##NAME(unsigned long long* out, double seed, int trips) {                  \
        unsigned long long a0 = (unsigned long long)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned x = (unsigned)a0 * 3 + 1, y = (unsigned)a0 * 5 + 3;                             \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        for (int i = 0; i < trips; ++i) {                                                        \
            REP4(asm volatile(INS " %0, vcc, %8, %9, %0\n" INS " %1, vcc, %8, %9, %1\n" INS " %2, vcc, %8, %9, %2\n" INS " %3, vcc, %8, %9, %3\n" \
                              INS " %4, vcc, %8, %9, %4\n" INS " %5, vcc, %8, %9, %5\n" INS " %6, vcc, %8, %9, %6\n" INS " %7, vcc, %8, %9, %7\n" \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                 \
                              : "v"(x), "v"(y) : "vcc");)                                         \
        }                                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345678u) out[1] = 1;                      \
        if ((threadIdx.x & 63) == 0) atomicMax(out, t1 - t0);                                                  \
    }
// compare into vcc
#define CASE_CMP(NAME, INS, T)                                                                   \
    __global__ void k_##NAME(unsigned long long* out, double seed, int trips) {                  \
        T x = (T)seed, y = (T)(seed * 2);                                                        \
        unsigned long long acc = 0;                                                              \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        for (int i = 0; i < trips; ++i) {                                                        \
            REP4(asm volatile(INS " vcc, %0, %1\n" INS " vcc, %0, %1\n" INS " vcc, %0, %1\n" INS " vcc, %0, %1\n"             \
                              INS " vcc, %0, %1\n" INS " vcc, %0, %1\n" INS " vcc, %0, %1\n" INS " vcc, %0, %1\n"             \
                              : : "v"(x), "v"(y) : "vcc");)                                       \
        }                                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        if (acc == 12345678u) out[1] = 1;                                                        \
        if ((threadIdx.x & 63) == 0) atomicMax(out, t1 - t0);                                                  \
    }

CASE32_3(fma_f32, "v_fma_f32")
CASE64_3(pk_fma_f32, "v_pk_fma_f32")
CASE64_3(fma_f64, "v_fma_f64")
CASE64_2(add_f64, "v_add_f64")
CASE64_2(mul_f64, "v_mul_f64")
CASE64_2(max_f64, "v_max_f64")
CASE64_1(rcp_f64, "v_rcp_f64")
CASE64_1(rsq_f64, "v_rsq_f64")
CASE64_1(sqrt_f64, "v_sqrt_f64")
CASE64_1(fract_f64, "v_fract_f64")
CASE64_1(floor_f64, "v_floor_f64")
CASE64_3(div_fmas_f64, "v_div_fmas_f64")
CASE64_3(div_fixup_f64, "v_div_fixup_f64")
CASE64_SHIFT(lshlrev_b64, "v_lshlrev_b64")
CASE64_SHIFT(lshrrev_b64, "v_lshrrev_b64")
CASE32_1(rcp_f32, "v_rcp_f32")
CASE32_1(mov_b32, "v_mov_b32")
CASE32_2(mul_lo_u32, "v_mul_lo_u32")
CASE32_2(mul_hi_u32, "v_mul_hi_u32")
CASE32_2(mul_u32_u24, "v_mul_u32_u24")
CASE32_2(xor_b32, "v_xor_b32")
CASE32_2(add_u32, "v_add_u32")
CASE32_3(max3_f32, "v_max3_f32")
CASE32_3(alignbit, "v_alignbit_b32")
CASE32_3(xad_u32, "v_xad_u32")
CASE32_3(add3_u32, "v_add3_u32")
CASE32_3(mad_u32_u24, "v_mad_u32_u24")
CASE32_3(bfe_u32, "v_bfe_u32")
CASE32_2(cndmask, "v_cndmask_b32")
CASE32_2T(cndmask_vcc, "v_cndmask_b32_e32", ", vcc")
CASE32_2T(cndmask_sgpr, "v_cndmask_b32_e64", ", s[10:11]")
CASE32_2T(cndmask_e64_vcc, "v_cndmask_b32_e64", ", vcc")
CASE32_2T(cmp_cndmask_vcc, "v_cmp_lt_u32_e32 vcc, %8, %0\nv_cndmask_b32_e32", ", vcc")
CASE32_2T(cmp_cndmask_sgpr, "v_cmp_lt_u32_e64 s[10:11], %8, %0\nv_cndmask_b32_e64", ", s[10:11]")
CASE32_2T(cmp_cndmask_e64_vcc, "v_cmp_lt_u32_e64 vcc, %8, %0\nv_cndmask_b32_e64", ", vcc")
CASE32_2(cvt_f32_u32_pair, "v_cvt_pk_u16_u32")
CASE32_2(lshl_or, "v_lshlrev_b32")
CASE32_2(and_b32, "v_and_b32")
CASE32_2(readlane_like, "v_mbcnt_lo_u32_b32")
CASE32_2(bpermute, "ds_bpermute_b32")
CASE_MAD64(mad_u64_u32, "v_mad_u64_u32")
CASE_CMP(cmp_lt_f64, "v_cmp_lt_f64", double)
CASE_CMP(cmp_lt_f32, "v_cmp_lt_f32", float)
CASE_CMP(cmp_lt_u32, "v_cmp_lt_u32", unsigned)

struct Case {
    const char* name;
    void (*fn)(unsigned long long*, double, int);
};
#define ENTRY(NAME) Case{#NAME, k_##NAME}

int main() {
    const std::vector<Case> cases = {
        ENTRY(fma_f32), ENTRY(pk_fma_f32), ENTRY(fma_f64), ENTRY(add_f64), ENTRY(mul_f64), ENTRY(max_f64), ENTRY(rcp_f64), ENTRY(rsq_f64),
        ENTRY(sqrt_f64), ENTRY(fract_f64), ENTRY(floor_f64), ENTRY(div_fmas_f64), ENTRY(div_fixup_f64), ENTRY(lshlrev_b64),
        ENTRY(lshrrev_b64), ENTRY(rcp_f32), ENTRY(mov_b32), ENTRY(mul_lo_u32), ENTRY(mul_hi_u32), ENTRY(mul_u32_u24), ENTRY(xor_b32),
        ENTRY(add_u32), ENTRY(max3_f32), ENTRY(alignbit), ENTRY(xad_u32), ENTRY(add3_u32), ENTRY(mad_u32_u24), ENTRY(bfe_u32),
        ENTRY(cndmask), ENTRY(cndmask_vcc), ENTRY(cndmask_sgpr), ENTRY(cndmask_e64_vcc), ENTRY(cmp_cndmask_vcc), ENTRY(cmp_cndmask_sgpr), ENTRY(cmp_cndmask_e64_vcc), ENTRY(cvt_f32_u32_pair), ENTRY(lshl_or), ENTRY(and_b32), ENTRY(readlane_like), ENTRY(bpermute), ENTRY(mad_u64_u32), ENTRY(cmp_lt_f64), ENTRY(cmp_lt_f32), ENTRY(cmp_lt_u32)};
    unsigned long long* out;
    if (hipMalloc(&out, 16) != hipSuccess) return 1;
    const int trips = 500;
    std::printf("%-16s %10s %10s   (shader clocks per wave-instruction; at 4 waves per SIMD: per instruction of ONE wave, so /4 = the SIMD's)\n",
                "instruction", "1 wave", "4 waves");
    for (const Case& c : cases) {
        double per[2];
        for (int v = 0; v < 2; ++v) {
            const int block = v == 0 ? 256 : 1024;
            unsigned long long h[2] = {0, 0};
            for (int rep = 0; rep < 2; ++rep) {  // the first warms the instruction cache
                if (hipMemset(out, 0, 16) != hipSuccess) return 4;
                hipLaunchKernelGGL(c.fn, dim3(1), dim3(block), 0, 0, out, 3.0, trips);
                if (hipDeviceSynchronize() != hipSuccess) return 2;
            }
            if (hipMemcpy(h, out, 16, hipMemcpyDeviceToHost) != hipSuccess) return 3;
            per[v] = (double)h[0] / (trips * 128.0);
        }
        std::printf("%-16s %10.2f %10.2f\n", c.name, per[0], per[1]);
    }
    (void)hipFree(out);
    return 0;
}
