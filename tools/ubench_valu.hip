// ubench_valu.hip -- instruction issue cost of the f64 operations the traversal uses,
// measured on the device with s_memtime: one wave per SIMD and four waves per SIMD.
// Development tool.   hipcc --offload-arch=gfx950 -O2 -o ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

enum Op { ADD_F64, MUL_F64, FMA_F64, MAX_F64, MIN_F64, CMP_F64, CNDMASK, ADD_U32, MOV_B32, DIV_F64, SQRT_F64, RCP_F64, LDS_B64, LDS2_B64, LDS_B128, N_OPS };
static const char* names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_max_f64", "v_min_f64", "v_cmp_f64+cndmask", "v_cndmask_b32", "v_add_u32", "v_mov_b32",
                              "f64 divide (a/b)", "f64 sqrt", "v_rcp_f64", "ds_read_b64 (random)", "ds_read2_b64 (random)", "ds_read_b128 (random)"};

template <int OP>
__global__ void k(double* out, unsigned long long* cycles, int iters, double seed) {
    __shared__ double lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = seed + i;
    __syncthreads();
    double a[8];
    unsigned u[8];
    for (int j = 0; j < 8; ++j) {
        a[j] = seed + threadIdx.x * 0.001 + j;
        u[j] = threadIdx.x * 7 + j;
    }
    const double b = seed * 1.0000001, c = seed * 0.5;
    unsigned idx = (threadIdx.x * 2654435761u) >> 20;  // pseudo-random LDS index 0..4095
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (OP == ADD_F64) a[j] = a[j] + b;
                if (OP == MUL_F64) a[j] = a[j] * b;
                if (OP == FMA_F64) a[j] = __builtin_fma(a[j], b, c);
                if (OP == MAX_F64) a[j] = fmax(a[j], b + j);
                if (OP == MIN_F64) a[j] = fmin(a[j], b + j);
                if (OP == CMP_F64) a[j] = a[j] < b ? c : a[j];
                if (OP == CNDMASK) u[j] = (u[(j + 1) & 7] & 1) ? u[j] : u[(j + 3) & 7];
                if (OP == ADD_U32) u[j] = u[j] + u[(j + 1) & 7];
                if (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(u[j]) : "v"(u[(j + 1) & 7]));
                if (OP == DIV_F64) a[j] = b / a[j];
                if (OP == SQRT_F64) a[j] = sqrt(a[j]);
                if (OP == RCP_F64) a[j] = __builtin_amdgcn_rcp(a[j]);
                if (OP == LDS_B64) {
                    a[j] += lds[(idx + j * 13) & 4095];
                }
                if (OP == LDS2_B64) {
                    const unsigned k0 = (idx + j * 13) & 4080;
                    a[j] += lds[k0] + lds[k0 + 6];
                }
                if (OP == LDS_B128) {
                    const unsigned k0 = (idx + j * 26) & 4094;
                    const double2 v = *reinterpret_cast<const double2*>(&lds[k0 & ~1u]);
                    a[j] += v.x + v.y;
                }
            }
        }
        if (OP >= LDS_B64) idx = (idx * 5 + 1) & 4095;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int j = 0; j < 8; ++j) s += a[j] + u[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
void run(int waves_per_simd) {
    const int cus = 256, block = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
    const int blocks_per_cu = 256 * waves_per_simd / block;
    const int grid = cus * blocks_per_cu, iters = 2000;
    double* out;
    unsigned long long* cyc;
    const int waves = grid * block / 64;
    hipMalloc(&out, sizeof(double) * grid * block);
    hipMalloc(&cyc, sizeof(unsigned long long) * waves);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(block), 0, 0, out, cyc, 10, 1.5);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(block), 0, 0, out, cyc, iters, 1.5);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(waves);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= waves;
    const double ops = (double)iters * 32;
    std::printf("  %-24s %d waves/SIMD: %7.2f clocks per op per wave (s_memtime), %7.2f ns*SIMD per wave-op (wall %.3f ms)\n", names[OP],
                waves_per_simd, avg / ops, ms * 1e6 / ops / waves_per_simd, ms);
    hipFree(out);
    hipFree(cyc);
}

template <int OP>
void both() {
    run<OP>(1);
    run<OP>(4);
}

int main() {
    both<ADD_F64>();
    both<MUL_F64>();
    both<FMA_F64>();
    both<MAX_F64>();
    both<MIN_F64>();
    both<CMP_F64>();
    both<CNDMASK>();
    both<ADD_U32>();
    both<MOV_B32>();
    both<DIV_F64>();
    both<SQRT_F64>();
    both<RCP_F64>();
    both<LDS_B64>();
    both<LDS2_B64>();
    both<LDS_B128>();
    return 0;
}
