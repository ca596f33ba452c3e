// valu_issue.hip -- how many cycles does one SIMD of an MI355X CU need per VALU wave-instruction?
//
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/valu_issue profiles/valu_issue.hip && gpurun_out/valu_issue
//
// One workgroup of W x 4 waves per CU (W waves on each of the CU's 4 SIMDs), every wave runs the same unrolled stream of
// N independent instructions of one kind, REPS times; every wave brackets its loop with s_memtime and the host takes the
// workgroup's window (latest end - earliest start), so the figure does not depend on how the SIMD arbitrates between its waves.
// Reported: cycles per wave-instruction PER SIMD = elapsed cycles / (REPS x N x W): the issue cost the tile kernel's
// instruction mix is priced with (bench.py).  The streams use 8 independent register chains, so dependent-issue latency is
// hidden even for one wave.  256 workgroups (one per CU) so that the chip's clock is the loaded clock.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                                      \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

constexpr int N_UNROLL = 64;   // instructions per loop body (8 chains x 8)
constexpr int REPS = 2000;

enum Kind { FMA_F64, ADD_F64, MUL_F64, MIN_F64, ADD_U32, MOV_B32, CNDMASK, CNDMASK3, DPP_MOV, FMA_F32, CMP_F64, READLANE, LSHL_ADD, N_KINDS };
static const char* kNames[N_KINDS] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_min_f64", "v_add_u32", "v_mov_b32", "v_cndmask_b32 (vop2, vcc)", "v_cndmask_b32 (vop3, sgpr pair)",
                                     "v_mov_b32 dpp row_shr:1", "v_fma_f32", "v_cmp_ge_f64 (vcc)", "v_readlane_b32", "v_lshl_add_u32"};

template <int KIND>
__global__ __launch_bounds__(1024) void k_issue(unsigned long long* out, double seed) {
    double d[8];
    unsigned u[8];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = seed + i + threadIdx.x * 1e-9; u[i] = (unsigned)(i + threadIdx.x); f[i] = (float)d[i]; }
    const double c1 = 1.0000001, c2 = 1e-9;
    unsigned long long sel = 0x5555aaaa5555aaaaull;
    asm volatile("s_mov_b64 vcc, %1\n\ts_nop 4" : "+s"(sel) : "s"(sel) : "vcc");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int j = 0; j < N_UNROLL / 8; ++j) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(c1), "v"(c2));
                if (KIND == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c2));
                if (KIND == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c1));
                if (KIND == MIN_F64) asm volatile("v_min_f64 %0, |%0|, %1" : "+v"(d[i]) : "v"(c1));
                if (KIND == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                // (vcc is set once in front of the loop and only read here: declaring it clobbered makes the compiler pad every
                //  instruction with the wait states of a VALU write of vcc)
                if (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == CNDMASK3) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "s"(sel));
                if (KIND == DPP_MOV) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
                if (KIND == CMP_F64) asm volatile("v_cmp_ge_f64 vcc, %0, %1" : : "v"(d[i]), "v"(c1) : "vcc");
                if (KIND == READLANE) { unsigned s; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(u[i])); asm volatile("" : : "s"(s)); }
                if (KIND == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double acc = 0;
    unsigned ua = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc += d[i] + f[i]; ua += u[i]; }
    // every wave reports its own window; the host takes (latest end - earliest start) of the workgroup: the SIMD's busy time
    // whatever the arbitration between its waves (oldest-first would let wave 0 finish long before the others)
    if ((threadIdx.x & 63) == 0) {
        out[2 * (blockIdx.x * 16 + (threadIdx.x >> 6))] = t0;
        out[2 * (blockIdx.x * 16 + (threadIdx.x >> 6)) + 1] = t1;
    }
    if (acc == 12345.678 && ua == 77u) out[0] = 1;  // keep the chains alive
}

template <int KIND>
static double run(int waves_per_simd, unsigned long long* dev, int n_blocks) {
    const int threads = waves_per_simd * 4 * 64;
    hipLaunchKernelGGL(k_issue<KIND>, dim3(n_blocks), dim3(threads), 0, 0, dev, 1.5);
    CHK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_issue<KIND>, dim3(n_blocks), dim3(threads), 0, 0, dev, 1.5);
    CHK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(32 * (size_t)n_blocks);
    CHK(hipMemcpy(h.data(), dev, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    double sum = 0;
    const int n_waves = waves_per_simd * 4;
    for (int b = 0; b < n_blocks; ++b) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < n_waves; ++w) {
            lo = std::min(lo, h[2 * ((size_t)b * 16 + w)]);
            hi = std::max(hi, h[2 * ((size_t)b * 16 + w) + 1]);
        }
        sum += (double)(hi - lo);
    }
    const double cycles = sum / n_blocks;
    return cycles / ((double)REPS * N_UNROLL * waves_per_simd);
}

__global__ void k_clock(unsigned long long* out) {
    // shader cycles (s_memtime) against the 100 MHz s_memrealtime over the same interval: ticks of s_memtime per microsecond
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < 100000ull) r1 = __builtin_amdgcn_s_memrealtime();  // 1 ms
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[0] = t1 - t0; out[1] = r1 - r0;
}

int main() {
    unsigned long long* dev;
    const int n_blocks = 256;
    CHK(hipMalloc(&dev, sizeof(unsigned long long) * 32 * n_blocks));
    CHK(hipMemset(dev, 0, sizeof(unsigned long long) * 32 * n_blocks));
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, dev);
    CHK(hipDeviceSynchronize());
    unsigned long long hc[2];
    CHK(hipMemcpy(hc, dev, sizeof hc, hipMemcpyDeviceToHost));
    const double mhz = (double)hc[0] / ((double)hc[1] / 100.0);
    printf("{\"s_memtime_ticks_per_us\": %.1f, \"note\": \"cycles below are s_memtime ticks; at %.0f ticks/us they are %s\",\n", mhz, mhz,
           mhz > 1000 ? "shader cycles" : "NOT shader cycles (constant clock): scale by shader MHz / this");
    printf(" \"cycles_per_wave_instruction_per_simd\": {\n");
    const int ws[4] = {1, 2, 3, 4};
    for (int k = 0; k < N_KINDS; ++k) {
        printf("  \"%s\": {", kNames[k]);
        for (int wi = 0; wi < 4; ++wi) {
            const int w = ws[wi];
            double c = 0;
            switch (k) {
                case FMA_F64: c = run<FMA_F64>(w, dev, n_blocks); break;
                case ADD_F64: c = run<ADD_F64>(w, dev, n_blocks); break;
                case MUL_F64: c = run<MUL_F64>(w, dev, n_blocks); break;
                case MIN_F64: c = run<MIN_F64>(w, dev, n_blocks); break;
                case ADD_U32: c = run<ADD_U32>(w, dev, n_blocks); break;
                case MOV_B32: c = run<MOV_B32>(w, dev, n_blocks); break;
                case CNDMASK: c = run<CNDMASK>(w, dev, n_blocks); break;
                case CNDMASK3: c = run<CNDMASK3>(w, dev, n_blocks); break;
                case DPP_MOV: c = run<DPP_MOV>(w, dev, n_blocks); break;
                case FMA_F32: c = run<FMA_F32>(w, dev, n_blocks); break;
                case CMP_F64: c = run<CMP_F64>(w, dev, n_blocks); break;
                case READLANE: c = run<READLANE>(w, dev, n_blocks); break;
                case LSHL_ADD: c = run<LSHL_ADD>(w, dev, n_blocks); break;
            }
            printf("\"%d_waves_per_simd\": %.3f%s", w, c, wi < 3 ? ", " : "");
        }
        printf("}%s\n", k + 1 < N_KINDS ? "," : "");
    }
    printf(" }}\n");
    CHK(hipFree(dev));
    return 0;
}
