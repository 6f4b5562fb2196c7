// microbench.hip -- VALU issue-rate probes for gfx950 (build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench)
// Measures wall time of kernels made of long unrolled runs of one instruction kind, at 1..8 waves per SIMD,
// and prints cycles per wave-instruction per SIMD (assuming the clock printed by the first probe).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 2000;
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void k_fma(float* out, float a, float b)
{
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = __builtin_fmaf(v[i], a, b);
    }
    float s = 0; for (int i = 0; i < 16; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_pkfma(float* out, float a, float b)
{
    f2 v[16];
    for (int i = 0; i < 16; i++) v[i] = (f2){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    const f2 a2 = (f2){a, a}, b2 = (f2){b, b};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = __builtin_elementwise_fma(v[i], a2, b2);
    }
    f2 s = (f2){0, 0}; for (int i = 0; i < 16; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

__global__ void k_exp(float* out, float a)
{
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 0.001f + i * 0.01f;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.0f + v[i];   // exp + fma
    }
    float s = 0; for (int i = 0; i < 16; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 11 readlanes feeding 11 fmas (record broadcast pattern)
__global__ void k_readlane(float* out, float a)
{
    float src[11], acc[11];
    for (int i = 0; i < 11; i++) { src[i] = threadIdx.x * 0.5f + i; acc[i] = i; }
    for (int it = 0; it < ITERS; it++) {
        const int j = it & 63;
#pragma unroll
        for (int i = 0; i < 11; i++) {
            const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(src[i]), j));
            acc[i] = __builtin_fmaf(acc[i], a, s);
        }
    }
    float s = 0; for (int i = 0; i < 11; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// same 11 fmas without the readlanes (baseline for the probe above)
__global__ void k_fma11(float* out, float a)
{
    float src[11], acc[11];
    for (int i = 0; i < 11; i++) { src[i] = threadIdx.x * 0.5f + i; acc[i] = i; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 11; i++) acc[i] = __builtin_fmaf(acc[i], a, src[i]);
    }
    float s = 0; for (int i = 0; i < 11; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// LDS broadcast: three ds_read_b128 of one record + 11 fmas
__global__ void k_ldsbcast(float* out, float a)
{
    __shared__ float4 rec[64 * 3];
    for (int i = threadIdx.x; i < 192; i += blockDim.x) rec[i] = make_float4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    float acc[12];
    for (int i = 0; i < 12; i++) acc[i] = i;
    for (int it = 0; it < ITERS; it++) {
        const int j = it & 63;
        const float4 x = rec[j * 3], y = rec[j * 3 + 1], z = rec[j * 3 + 2];
        acc[0] = __builtin_fmaf(acc[0], a, x.x); acc[1] = __builtin_fmaf(acc[1], a, x.y);
        acc[2] = __builtin_fmaf(acc[2], a, x.z); acc[3] = __builtin_fmaf(acc[3], a, x.w);
        acc[4] = __builtin_fmaf(acc[4], a, y.x); acc[5] = __builtin_fmaf(acc[5], a, y.y);
        acc[6] = __builtin_fmaf(acc[6], a, y.z); acc[7] = __builtin_fmaf(acc[7], a, y.w);
        acc[8] = __builtin_fmaf(acc[8], a, z.x); acc[9] = __builtin_fmaf(acc[9], a, z.y);
        acc[10] = __builtin_fmaf(acc[10], a, z.z); acc[11] = __builtin_fmaf(acc[11], a, z.w);
    }
    float s = 0; for (int i = 0; i < 12; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// DPP adds (10 interleaved chains x 6 steps), as in the backward's wave reduction
__global__ void k_dpp(float* out)
{
    float v[10];
    for (int i = 0; i < 10; i++) v[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITERS / 4; it++) {
#define STEP(C) "v_add_f32_dpp %0, %0, %0 " C "\n v_add_f32_dpp %1, %1, %1 " C "\n v_add_f32_dpp %2, %2, %2 " C "\n v_add_f32_dpp %3, %3, %3 " C "\n v_add_f32_dpp %4, %4, %4 " C "\n v_add_f32_dpp %5, %5, %5 " C "\n v_add_f32_dpp %6, %6, %6 " C "\n v_add_f32_dpp %7, %7, %7 " C "\n v_add_f32_dpp %8, %8, %8 " C "\n v_add_f32_dpp %9, %9, %9 " C "\n"
        asm volatile("s_nop 1\n" STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                     STEP("row_half_mirror row_mask:0xf bank_mask:0xf") STEP("row_mirror row_mask:0xf bank_mask:0xf")
                     STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") STEP("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]));
        for (int i = 0; i < 10; i++) v[i] *= 0.5f;
    }
    float s = 0; for (int i = 0; i < 10; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// gfx950 lane-swap reduction of 10 values: v_permlane32_swap / v_permlane16_swap halve the number of live registers at
// each of the two upper levels (10 -> 5 -> 3), the four in-row levels run on 3 registers: 8 swaps + 8 adds + 12 DPP adds
// instead of 60 DPP adds.  Row r of (t0, t1, t2) ends up holding the totals of values
//   r=0: v0 v4 v8   r=1: v2 v6 -   r=2: v1 v5 v9   r=3: v3 v7 -
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float swap_add32(float a, float b)
{
    const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float swap_add16(float a, float b)
{
    const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ void wave_sum10_swap(const float (&v)[10], float& t0, float& t1, float& t2)
{
    const float s0 = swap_add32(v[0], v[1]), s1 = swap_add32(v[2], v[3]), s2 = swap_add32(v[4], v[5]),
                s3 = swap_add32(v[6], v[7]), s4 = swap_add32(v[8], v[9]);
    t0 = swap_add16(s0, s1);
    t1 = swap_add16(s2, s3);
    t2 = swap_add16(s4, 0.0f);
#define ROWSTEP(C) "v_add_f32_dpp %0, %0, %0 " C "\n v_add_f32_dpp %1, %1, %1 " C "\n v_add_f32_dpp %2, %2, %2 " C "\n"
    asm volatile("s_nop 1\n" ROWSTEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") "s_nop 0\n"
                 ROWSTEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") "s_nop 0\n"
                 ROWSTEP("row_half_mirror row_mask:0xf bank_mask:0xf") "s_nop 0\n"
                 ROWSTEP("row_mirror row_mask:0xf bank_mask:0xf") "s_nop 1"
                 : "+v"(t0), "+v"(t1), "+v"(t2));
}

__device__ __forceinline__ float wave_sum10_transposed(const float (&v)[10])
{
    const float s0 = swap_add32(v[0], v[1]), s1 = swap_add32(v[2], v[3]), s2 = swap_add32(v[4], v[5]),
                s3 = swap_add32(v[6], v[7]), s4 = swap_add32(v[8], v[9]);
    float t0 = swap_add16(s0, s1), t1 = swap_add16(s2, s3), t2 = swap_add16(s4, 0.0f);
    // DPP bank_mask selects the four 4-lane groups of a row, so the halving continues on the 8- and 4-lane levels
    asm volatile(
        "s_nop 1\n\t"
        // halves of a row: lanes 0-7 keep t0, lanes 8-15 take t1; t2 is reduced in full
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 1\n\t"
        // quads of a half (mirror within 8 lanes): quads 0 and 2 keep (t0 | t1), quads 1 and 3 take t2
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1\n\t"
        // inside the quads
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(t0), "+v"(t1), "+v"(t2));
    return t0;
}

// Same reduction with the cheap halvings first: the two in-row levels that bank_mask can split (8-lane halves, quads
// of a half) run on 10 and 5 registers as DPP adds, the lane swaps then see 3 and 2 registers, the quad levels 1:
// 17 DPP adds + 3 swaps + 3 adds.  Result: slot 4 r + q (row r, quad q) holds
//   r=0: v0 v2 v1 v3   r=1: v8 v8 v9 v9   r=2: v4 v6 v5 v7   r=3: (v8 v8 v9 v9 of the upper half only: unused)
__device__ __forceinline__ float wave_sum10_rowfirst(const float (&v)[10])
{
    float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4], a5 = v[5], a6 = v[6], a7 = v[7], a8 = v[8], a9 = v[9];
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %6, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %8, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %8, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %4, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9));
    const float c0 = swap_add32(a0, a4), c1 = swap_add32(a8, a8);
    float d = swap_add16(c0, c1);
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(d));
    return d;
}

__global__ void k_rowfirst_check(float* out)
{
    float v[10];
    for (int i = 0; i < 10; i++) v[i] = (float)((i + 1) * 1000 + (int)threadIdx.x);
    out[threadIdx.x] = wave_sum10_rowfirst(v);
}

__global__ void k_rowfirst(float* out)
{
    float v[10];
    for (int i = 0; i < 10; i++) v[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITERS / 4; it++) {
        const float t = wave_sum10_rowfirst(v);
        for (int i = 0; i < 10; i++) v[i] = v[i] * 0.5f + t * 1e-6f;
    }
    float s = 0; for (int i = 0; i < 10; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_transposed_check(float* out)
{
    float v[10];
    for (int i = 0; i < 10; i++) v[i] = (float)((i + 1) * 1000 + (int)threadIdx.x);
    out[threadIdx.x] = wave_sum10_transposed(v);
}

__global__ void k_transposed(float* out)
{
    float v[10];
    for (int i = 0; i < 10; i++) v[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITERS / 4; it++) {
        const float t = wave_sum10_transposed(v);
        for (int i = 0; i < 10; i++) v[i] = v[i] * 0.5f + t * 1e-6f;
    }
    float s = 0; for (int i = 0; i < 10; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_swapsum(float* out)
{
    float v[10];
    for (int i = 0; i < 10; i++) v[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITERS / 4; it++) {
        float t0, t1, t2;
        wave_sum10_swap(v, t0, t1, t2);
        for (int i = 0; i < 10; i++) v[i] = v[i] * 0.5f + (i % 3 == 0 ? t0 : (i % 3 == 1 ? t1 : t2)) * 1e-6f;
    }
    float s = 0; for (int i = 0; i < 10; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// correctness of the lane-swap reduction: value k of lane l is (k + 1) * 1000 + l
__global__ void k_swapsum_check(float* out)
{
    float v[10];
    for (int i = 0; i < 10; i++) v[i] = (float)((i + 1) * 1000 + (int)threadIdx.x);
    float t0, t1, t2;
    wave_sum10_swap(v, t0, t1, t2);
    out[threadIdx.x * 3] = t0; out[threadIdx.x * 3 + 1] = t1; out[threadIdx.x * 3 + 2] = t2;
}

// one VALU opcode, 16 independent registers, long unrolled runs (issue rate by instruction class)
#define OPK(NAME, ASM)                                                                                       \
    __global__ void NAME(float* out, float a)                                                                \
    {                                                                                                        \
        float v[16];                                                                                         \
        for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 0.001f + i;                                        \
        for (int it = 0; it < ITERS; it++) {                                                                 \
            _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(v[i]) : "v"(a));          \
        }                                                                                                    \
        float s = 0; for (int i = 0; i < 16; i++) s += v[i];                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                      \
    }
OPK(k_op_mul, "v_mul_f32 %0, %0, %1")
OPK(k_op_add, "v_add_f32 %0, %0, %1")
OPK(k_op_sub, "v_sub_f32 %0, %0, %1")
OPK(k_op_fmac, "v_fmac_f32 %0, %0, %1")
OPK(k_op_min, "v_min_f32 %0, %0, %1")
OPK(k_op_mov, "v_mov_b32 %0, %1")
OPK(k_op_cmp, "v_cmp_gt_f32 vcc, %0, %1")
OPK(k_op_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
OPK(k_op_exp, "v_exp_f32 %0, %0")
OPK(k_op_rcp, "v_rcp_f32 %0, %0")
OPK(k_op_cnd_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
OPK(k_op_cnd_mix, "v_cndmask_b32 %0, %0, %1, vcc\n v_fmac_f32 %0, %0, %1\n v_fmac_f32 %0, %0, %1\n v_fmac_f32 %0, %0, %1")
OPK(k_op_cmp_sgpr, "v_cmp_gt_f32_e64 s[20:21], %0, %1")
OPK(k_op_ashr, "v_ashrrev_i32 %0, 31, %0")
OPK(k_op_bfi, "v_bfi_b32 %0, %0, %1, %0")
OPK(k_op_max, "v_max_f32 %0, %0, %1")
OPK(k_op_med3, "v_med3_f32 %0, %0, %1, %1")
OPK(k_op_addu, "v_add_u32 %0, %0, %1")
OPK(k_op_and, "v_and_b32 %0, %0, %1")
OPK(k_op_sub_clamp, "v_sub_f32_e64 %0, %0, %1 clamp")
OPK(k_op_fma_clamp, "v_fma_f32 %0, %0, %1, %1 clamp")
OPK(k_op_fma, "v_fma_f32 %0, %0, %1, %1")
OPK(k_op_max_clamp, "v_max_f32_e64 %0, %0, %0 clamp")
OPK(k_op_cmp_u32, "v_cmp_ge_u32 vcc, %0, %1")
OPK(k_op_dpp_add, "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf")
OPK(k_op_readfirstlane, "v_readfirstlane_b32 s20, %0")

template <class F>
static double time_ms(F&& launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / 5.0;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    float* out;
    CHECK(hipMalloc(&out, sizeof(float) * cus * 4 * 8 * 64 * 2));
    const double ghz = prop.clockRate / 1e6;
    {
        float* chk;
        CHECK(hipMalloc(&chk, sizeof(float) * 192));
        hipLaunchKernelGGL(k_swapsum_check, dim3(1), dim3(64), 0, 0, chk);
        float h[192];
        CHECK(hipMemcpy(h, chk, sizeof(h), hipMemcpyDeviceToHost));
        const int expect[4][3] = {{0, 4, 8}, {2, 6, -1}, {1, 5, 9}, {3, 7, -1}};
        int bad = 0;
        for (int l = 0; l < 64; l++)
            for (int k = 0; k < 3; k++) {
                const int val = expect[l >> 4][k];
                const float want = val < 0 ? 0.0f : 64.0f * (val + 1) * 1000 + 2016.0f;
                if (h[l * 3 + k] != want) { if (bad < 8) printf("swap-sum mismatch lane %d t%d: got %g want %g\n", l, k, h[l * 3 + k], want); bad++; }
            }
        printf("lane-swap reduction check: %s\n", bad ? "FAILED" : "ok");
        hipLaunchKernelGGL(k_transposed_check, dim3(1), dim3(64), 0, 0, chk);
        CHECK(hipMemcpy(h, chk, sizeof(float) * 64, hipMemcpyDeviceToHost));
        const int at[4][4] = {{0, 8, 4, 8}, {2, -1, 6, -1}, {1, 9, 5, 9}, {3, -1, 7, -1}};
        bad = 0;
        for (int l = 0; l < 64; l++) {
            const int val = at[l >> 4][(l >> 2) & 3];
            const float want = val < 0 ? 0.0f : 64.0f * (val + 1) * 1000 + 2016.0f;
            if (h[l] != want) { if (bad < 8) printf("transposed mismatch lane %d: got %g want %g\n", l, h[l], want); bad++; }
        }
        printf("transposed 10-value reduction check: %s\n", bad ? "FAILED" : "ok");
        hipLaunchKernelGGL(k_rowfirst_check, dim3(1), dim3(64), 0, 0, chk);
        CHECK(hipMemcpy(h, chk, sizeof(float) * 64, hipMemcpyDeviceToHost));
        const int at2[4][4] = {{0, 2, 1, 3}, {8, 8, 9, 9}, {4, 6, 5, 7}, {-2, -2, -2, -2}};
        bad = 0;
        for (int l = 0; l < 64; l++) {
            const int val = at2[l >> 4][(l >> 2) & 3];
            if (val == -2) continue;      // unused slots
            const float want = 64.0f * (val + 1) * 1000 + 2016.0f;
            if (h[l] != want) { if (bad < 8) printf("row-first mismatch lane %d: got %g want %g\n", l, h[l], want); bad++; }
        }
        printf("row-first 10-value reduction check: %s\n", bad ? "FAILED" : "ok");
        hipFree(chk);
    }
    {
        const int blocks = cus * 8;
        struct { const char* name; void (*k)(float*, float); } ops[] = {
            {"v_mul_f32", k_op_mul}, {"v_add_f32", k_op_add}, {"v_sub_f32", k_op_sub}, {"v_fmac_f32", k_op_fmac},
            {"v_min_f32", k_op_min}, {"v_mov_b32", k_op_mov}, {"v_cmp_gt_f32", k_op_cmp}, {"v_cndmask_b32", k_op_cndmask},
            {"v_cndmask e64 sgpr", k_op_cnd_sgpr}, {"cndmask + 3 fmac (x4)", k_op_cnd_mix}, {"v_cmp e64 -> sgpr", k_op_cmp_sgpr},
            {"v_ashrrev_i32", k_op_ashr}, {"v_bfi_b32", k_op_bfi}, {"v_max_f32", k_op_max}, {"v_med3_f32", k_op_med3},
            {"v_exp_f32", k_op_exp}, {"v_rcp_f32", k_op_rcp}, {"v_add_u32", k_op_addu}, {"v_and_b32", k_op_and},
            {"v_fma_f32", k_op_fma}, {"v_sub_f32_e64 clamp", k_op_sub_clamp}, {"v_fma_f32 clamp", k_op_fma_clamp},
            {"v_max_f32_e64 clamp", k_op_max_clamp}, {"v_cmp_ge_u32 (vcc)", k_op_cmp_u32}, {"v_add_f32_dpp row_ror", k_op_dpp_add},
            {"v_readfirstlane_b32", k_op_readfirstlane}};
        for (auto& o : ops) {
            const double ms = time_ms([&] { hipLaunchKernelGGL(o.k, dim3(blocks), dim3(256), 0, 0, out, 1.0001f); });
            printf("issue rate, 8 waves/SIMD: %-22s %6.2f cycles per wave-instruction per SIMD\n", o.name,
                   ms * 1e-3 * ghz * 1e9 / (16.0 * ITERS * 8));
        }
    }
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = cus * wps;     // 256-thread blocks: one wave per SIMD each
        struct { const char* name; double instr; double ms; } rows[12];
        int n = 0;
        rows[n++] = {"v_fma_f32 x16", 16.0 * ITERS, time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); })};
        rows[n++] = {"v_pk_fma_f32 x16", 16.0 * ITERS, time_ms([&] { hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); })};
        rows[n++] = {"v_exp_f32+fma x16", 32.0 * ITERS, time_ms([&] { hipLaunchKernelGGL(k_exp, dim3(blocks), dim3(256), 0, 0, out, 1.0f); })};
        rows[n++] = {"11 fma (baseline)", 11.0 * ITERS, time_ms([&] { hipLaunchKernelGGL(k_fma11, dim3(blocks), dim3(256), 0, 0, out, 1.0001f); })};
        rows[n++] = {"11 readlane + 11 fma", 22.0 * ITERS, time_ms([&] { hipLaunchKernelGGL(k_readlane, dim3(blocks), dim3(256), 0, 0, out, 1.0001f); })};
        rows[n++] = {"3 ds_read_b128 + 12 fma", 12.0 * ITERS, time_ms([&] { hipLaunchKernelGGL(k_ldsbcast, dim3(blocks), dim3(256), 0, 0, out, 1.0001f); })};
        rows[n++] = {"60 dpp add + 10 mul", 70.0 * (ITERS / 4), time_ms([&] { hipLaunchKernelGGL(k_dpp, dim3(blocks), dim3(256), 0, 0, out); })};
        rows[n++] = {"transposed sum (23) + 20", 43.0 * (ITERS / 4), time_ms([&] { hipLaunchKernelGGL(k_transposed, dim3(blocks), dim3(256), 0, 0, out); })};
        rows[n++] = {"row-first sum (23) + 20", 43.0 * (ITERS / 4), time_ms([&] { hipLaunchKernelGGL(k_rowfirst, dim3(blocks), dim3(256), 0, 0, out); })};
        rows[n++] = {"swap-sum (28 instr) + 20", 48.0 * (ITERS / 4), time_ms([&] { hipLaunchKernelGGL(k_swapsum, dim3(blocks), dim3(256), 0, 0, out); })};
        for (int i = 0; i < n; i++) {
            const double cyc = rows[i].ms * 1e-3 * ghz * 1e9;
            printf("waves/SIMD %d  %-26s %8.3f ms  %6.2f cycles per counted VALU instr per SIMD (at %.2f GHz)\n", wps, rows[i].name,
                   rows[i].ms, cyc / (rows[i].instr * wps), ghz);
        }
    }
    hipFree(out);
    return 0;
}
