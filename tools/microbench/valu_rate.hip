// valu_rate.hip -- microbenchmark (not product): issue rate of scalar vs packed fp32 VALU on gfx950.
// Each wave runs ITERS iterations of 8 independent chains; reports cycles per wave-instruction per SIMD
// with 1, 2, 4, 8 waves per SIMD resident.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;

template <int KIND> __global__ void k(float* out, float seed)
{
    float a = seed + threadIdx.x, b = 1.0001f, c = 0.5f;
    if (KIND == 0) { // v_fma_f32
        float x0=a,x1=a+1,x2=a+2,x3=a+3,x4=a+4,x5=a+5,x6=a+6,x7=a+7;
        for (int i = 0; i < ITERS; i++) {
            x0=__builtin_fmaf(x0,b,c); x1=__builtin_fmaf(x1,b,c); x2=__builtin_fmaf(x2,b,c); x3=__builtin_fmaf(x3,b,c);
            x4=__builtin_fmaf(x4,b,c); x5=__builtin_fmaf(x5,b,c); x6=__builtin_fmaf(x6,b,c); x7=__builtin_fmaf(x7,b,c);
        }
        out[blockIdx.x*blockDim.x+threadIdx.x]=x0+x1+x2+x3+x4+x5+x6+x7;
    } else if (KIND == 1) { // v_pk_fma_f32
        f2 bb={b,b}, cc={c,c};
        f2 x0={a,a+8},x1={a+1,a+9},x2={a+2,a+10},x3={a+3,a+11},x4={a+4,a+12},x5={a+5,a+13},x6={a+6,a+14},x7={a+7,a+15};
        for (int i = 0; i < ITERS; i++) {
            x0=__builtin_elementwise_fma(x0,bb,cc); x1=__builtin_elementwise_fma(x1,bb,cc); x2=__builtin_elementwise_fma(x2,bb,cc); x3=__builtin_elementwise_fma(x3,bb,cc);
            x4=__builtin_elementwise_fma(x4,bb,cc); x5=__builtin_elementwise_fma(x5,bb,cc); x6=__builtin_elementwise_fma(x6,bb,cc); x7=__builtin_elementwise_fma(x7,bb,cc);
        }
        f2 s=x0+x1+x2+x3+x4+x5+x6+x7; out[blockIdx.x*blockDim.x+threadIdx.x]=s.x+s.y;
    } else if (KIND == 2) { // v_mul_f32 + v_add_f32 alternating (non-fused)
        float x0=a,x1=a+1,x2=a+2,x3=a+3,x4=a+4,x5=a+5,x6=a+6,x7=a+7;
        for (int i = 0; i < ITERS/2; i++) {
            x0*=b; x1*=b; x2*=b; x3*=b; x4*=b; x5*=b; x6*=b; x7*=b;
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            x0+=c; x1+=c; x2+=c; x3+=c; x4+=c; x5+=c; x6+=c; x7+=c;
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        }
        out[blockIdx.x*blockDim.x+threadIdx.x]=x0+x1+x2+x3+x4+x5+x6+x7;
    } else if (KIND == 4) { // v_rcp_f32 (transcendental unit)
        float x0=a+1,x1=a+2,x2=a+3,x3=a+4,x4=a+5,x5=a+6,x6=a+7,x7=a+8;
        for (int i = 0; i < ITERS; i++) {
            x0=__builtin_amdgcn_rcpf(x0); x1=__builtin_amdgcn_rcpf(x1); x2=__builtin_amdgcn_rcpf(x2); x3=__builtin_amdgcn_rcpf(x3);
            x4=__builtin_amdgcn_rcpf(x4); x5=__builtin_amdgcn_rcpf(x5); x6=__builtin_amdgcn_rcpf(x6); x7=__builtin_amdgcn_rcpf(x7);
        }
        out[blockIdx.x*blockDim.x+threadIdx.x]=x0+x1+x2+x3+x4+x5+x6+x7;
    } else if (KIND == 5) { // v_rsq_f32
        float x0=a+1,x1=a+2,x2=a+3,x3=a+4,x4=a+5,x5=a+6,x6=a+7,x7=a+8;
        for (int i = 0; i < ITERS; i++) {
            x0=__builtin_amdgcn_rsqf(x0); x1=__builtin_amdgcn_rsqf(x1); x2=__builtin_amdgcn_rsqf(x2); x3=__builtin_amdgcn_rsqf(x3);
            x4=__builtin_amdgcn_rsqf(x4); x5=__builtin_amdgcn_rsqf(x5); x6=__builtin_amdgcn_rsqf(x6); x7=__builtin_amdgcn_rsqf(x7);
        }
        out[blockIdx.x*blockDim.x+threadIdx.x]=x0+x1+x2+x3+x4+x5+x6+x7;
    } else if (KIND == 6) { // one v_rcp_f32 per 7 v_fma_f32: does the transcendental unit overlap the main pipe?
        float x0=a+1,x1=a+2,x2=a+3,x3=a+4,x4=a+5,x5=a+6,x6=a+7,x7=a+8;
        for (int i = 0; i < ITERS; i++) {
            x0=__builtin_amdgcn_rcpf(x0); x1=__builtin_fmaf(x1,b,c); x2=__builtin_fmaf(x2,b,c); x3=__builtin_fmaf(x3,b,c);
            x4=__builtin_fmaf(x4,b,c); x5=__builtin_fmaf(x5,b,c); x6=__builtin_fmaf(x6,b,c); x7=__builtin_fmaf(x7,b,c);
        }
        out[blockIdx.x*blockDim.x+threadIdx.x]=x0+x1+x2+x3+x4+x5+x6+x7;
    } else if (KIND >= 7 && KIND <= 14) { // one instruction class per kind, 8 independent chains, via inline asm
        float x0=a+1,x1=a+2,x2=a+3,x3=a+4,x4=a+5,x5=a+6,x6=a+7,x7=a+8;
#define OP8(TXT) asm volatile(TXT " %0, %0, %8\n\t" TXT " %1, %1, %8\n\t" TXT " %2, %2, %8\n\t" TXT " %3, %3, %8\n\t" \
                              TXT " %4, %4, %8\n\t" TXT " %5, %5, %8\n\t" TXT " %6, %6, %8\n\t" TXT " %7, %7, %8" \
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b))
        for (int i = 0; i < ITERS; i++) {
            if (KIND == 7) OP8("v_max_f32");
            if (KIND == 8) OP8("v_and_b32");
            if (KIND == 9) OP8("v_add_u32");
            if (KIND == 10) OP8("v_lshlrev_b32");
            if (KIND == 11) OP8("v_mul_f32");
            if (KIND == 12) { // v_cmp_lt_f32 -> vcc, v_cndmask_b32 (2 instructions per chain step)
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n\tv_cndmask_b32 %0, %0, %8, vcc\n\tv_cmp_lt_f32 vcc, %1, %8\n\tv_cndmask_b32 %1, %1, %8, vcc\n\t"
                             "v_cmp_lt_f32 vcc, %2, %8\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cmp_lt_f32 vcc, %3, %8\n\tv_cndmask_b32 %3, %3, %8, vcc"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b) : "vcc");
            }
            if (KIND == 13) { // v_mov_b32 chain (register copies)
                asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %4\n\t"
                             "v_mov_b32 %4, %5\n\tv_mov_b32 %5, %6\n\tv_mov_b32 %6, %7\n\tv_mov_b32 %7, %0"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            }
            if (KIND == 14) { // v_cvt_u32_f32 / v_cvt_f32_u32 alternating
                asm volatile("v_cvt_u32_f32 %0, %0\n\tv_cvt_f32_u32 %0, %0\n\tv_cvt_u32_f32 %1, %1\n\tv_cvt_f32_u32 %1, %1\n\t"
                             "v_cvt_u32_f32 %2, %2\n\tv_cvt_f32_u32 %2, %2\n\tv_cvt_u32_f32 %3, %3\n\tv_cvt_f32_u32 %3, %3"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            }
        }
#undef OP8
        out[blockIdx.x*blockDim.x+threadIdx.x]=x0+x1+x2+x3+x4+x5+x6+x7;
    } else if (KIND == 3) { // v_pk_mul_f32 / v_pk_add_f32 alternating
        f2 bb={b,b}, cc={c,c};
        f2 x0={a,a+8},x1={a+1,a+9},x2={a+2,a+10},x3={a+3,a+11},x4={a+4,a+12},x5={a+5,a+13},x6={a+6,a+14},x7={a+7,a+15};
        for (int i = 0; i < ITERS/2; i++) {
            x0*=bb; x1*=bb; x2*=bb; x3*=bb; x4*=bb; x5*=bb; x6*=bb; x7*=bb;
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            x0+=cc; x1+=cc; x2+=cc; x3+=cc; x4+=cc; x5+=cc; x6+=cc; x7+=cc;
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        }
        f2 s=x0+x1+x2+x3+x4+x5+x6+x7; out[blockIdx.x*blockDim.x+threadIdx.x]=s.x+s.y;
    }
}

template <int KIND> void run(const char* name, float* d)
{
    for (int wps = 1; wps <= 8; wps *= 2) {
        // 256 CUs x 4 SIMDs x wps waves; blocks of 256 threads = 4 waves = one per SIMD
        const int blocks = 256 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double instr_per_simd = (double)ITERS * 8 * wps; // wave-instructions issued on each SIMD
        const double cycles = ms * 1e-3 * 2.4e9;
        printf("%-28s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction (at 2.4 GHz)\n", name, wps, ms, cycles / instr_per_simd);
    }
}

int main()
{
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    // run-in: an idle MI355X needs tens of milliseconds of work to reach its running clocks
    for (int r = 0; r < 2000; r++) hipLaunchKernelGGL(k<0>, dim3(256 * 8), dim3(256), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    run<0>("v_fma_f32", d);
    run<1>("v_pk_fma_f32", d);
    run<2>("v_mul_f32/v_add_f32", d);
    run<3>("v_pk_mul_f32/v_pk_add_f32", d);
    run<4>("v_rcp_f32", d);
    run<5>("v_rsq_f32", d);
    run<6>("1 v_rcp_f32 : 7 v_fma_f32", d);
    run<11>("v_mul_f32 (asm)", d);
    run<7>("v_max_f32", d);
    run<8>("v_and_b32", d);
    run<9>("v_add_u32", d);
    run<10>("v_lshlrev_b32", d);
    run<12>("v_cmp_lt_f32+v_cndmask_b32", d);
    run<13>("v_mov_b32", d);
    run<14>("v_cvt_u32_f32/v_cvt_f32_u32", d);
    return 0;
}
