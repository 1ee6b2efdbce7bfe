// What one wave, alone on its SIMD, pays per instruction of the kinds the pivot-block chain is made of: cycles (s_memtime)
// per instruction in a DEPENDENT chain and in an INDEPENDENT run of the same instruction.  Prints a table, no assertions.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/chain_probe tools/microbench/chain_probe.hip && /tmp/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 64;           // instructions per timed run (unrolled)
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
__global__ void k(float* out, unsigned long long* cyc, float seed) {
    const int l = threadIdx.x;
    __shared__ float lds[256];
    lds[l] = seed + l;
    __syncthreads();
    float x = seed + 1.f + 0.001f * l, y = 1.0001f, z = 0.5f, a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
    int slot = 0;
    unsigned long long t0, t1;
#define TIMED(name_unused, body) t0 = now(); body; t1 = now(); if (l == 0) cyc[slot] = t1 - t0; ++slot;
    // 0: dependent v_fma
    TIMED(0, asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n\t") : "+v"(x) : "v"(y), "v"(z)));
    // 1: independent v_fma (4 accumulators round robin)
    TIMED(1, asm volatile(REP16("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t")
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y), "v"(z)));
    // 2: dependent v_rcp
    TIMED(2, asm volatile(REP64("v_rcp_f32 %0, %0\n\ts_nop 0\n\t") : "+v"(x)));
    // 3: independent v_rcp
    TIMED(3, asm volatile(REP16("v_rcp_f32 %0, %4\n\tv_rcp_f32 %1, %4\n\tv_rcp_f32 %2, %4\n\tv_rcp_f32 %3, %4\n\t")
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y)));
    // 4: dependent v_readlane -> v_mul with the SGPR (2 instructions per link)
    TIMED(4, asm volatile(REP64("v_readlane_b32 s20, %0, 5\n\ts_nop 3\n\tv_mul_f32 %0, s20, %0\n\t") : "+v"(x) :: "s20"));
    // 5: independent v_readlane (different SGPRs)
    TIMED(5, asm volatile(REP16("v_readlane_b32 s20, %0, 5\n\tv_readlane_b32 s21, %0, 6\n\tv_readlane_b32 s22, %0, 7\n\tv_readlane_b32 s23, %0, 8\n\t")
                          :: "v"(x) : "s20", "s21", "s22", "s23"));
    // 6: dependent DPP (quad_perm broadcast) add
    TIMED(6, asm volatile(REP64("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t") : "+v"(x)));
    // 7: dependent v_permlane32_swap
    TIMED(7, asm volatile(REP64("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\t") : "+v"(x), "+v"(y)));
    // 8: dependent ds_bpermute
    {
        int addr = ((l ^ 16) << 2);
        TIMED(8, asm volatile(REP64("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(x) : "v"(addr)));
    }
    // 9: LDS write -> read round trip (dependent)
    {
        int addr = l * 4;
        TIMED(9, asm volatile(REP64("ds_write_b32 %1, %0\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(x) : "v"(addr)));
    }
    // 10: dependent v_mfma_f32_32x32x2 (same accumulator)
    {
        f32x16 acc;
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        TIMED(10, asm volatile(REP64("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\t") : "+v"(acc) : "v"(y), "v"(z)));
        x += acc[3];
    }
    // 11: dependent v_mfma_f32_4x4x1 (same accumulator)
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        TIMED(11, asm volatile(REP64("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n\t") : "+v"(acc) : "v"(y), "v"(z)));
        x += acc[1];
    }
    // 12: the column step as it stands: readlane -> rcp -> mul -> fma (4 instructions per link, + the hazard nops)
    TIMED(12, asm volatile(REP64("v_readlane_b32 s20, %0, 5\n\ts_nop 1\n\tv_rcp_f32 %1, s20\n\ts_nop 0\n\tv_mul_f32 %1, %0, %1\n\tv_fma_f32 %0, -%1, %2, %0\n\t")
                           : "+v"(x), "+v"(a0) : "v"(z) : "s20"));
    // 13: v_readlane result consumed by v_rcp directly, then readlane of the result (2 instructions per link)
    TIMED(13, asm volatile(REP64("v_readlane_b32 s20, %0, 5\n\ts_nop 1\n\tv_rcp_f32 %0, s20\n\ts_nop 0\n\t") : "+v"(x) :: "s20"));
    // 14: s_nop 0 x 64 (scalar issue rate of the wave)
    TIMED(14, asm volatile(REP64("s_nop 0\n\t")));
    // 15: v_mov independent x 64
    TIMED(15, asm volatile(REP16("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %4\n\tv_mov_b32 %3, %4\n\t")
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y)));
    // 16: v_readfirstlane dependent -> v_mul
    TIMED(16, asm volatile(REP64("v_readfirstlane_b32 s20, %0\n\ts_nop 3\n\tv_mul_f32 %0, s20, %0\n\t") : "+v"(x) :: "s20"));
    // 17: one v_mfma_f32_32x32x2 followed by 11 independent v_fma, x 16: do they overlap inside ONE wave?
    {
        f32x16 acc;
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        TIMED(17, asm volatile(REP16("v_mfma_f32_32x32x2_f32 %0, %5, %6, %0\n\t"
                                     "v_fma_f32 %1, %1, %5, %6\n\tv_fma_f32 %2, %2, %5, %6\n\tv_fma_f32 %3, %3, %5, %6\n\tv_fma_f32 %4, %4, %5, %6\n\t"
                                     "v_fma_f32 %1, %1, %5, %6\n\tv_fma_f32 %2, %2, %5, %6\n\tv_fma_f32 %3, %3, %5, %6\n\tv_fma_f32 %4, %4, %5, %6\n\t"
                                     "v_fma_f32 %1, %1, %5, %6\n\tv_fma_f32 %2, %2, %5, %6\n\tv_fma_f32 %3, %3, %5, %6\n\t")
                               : "+v"(acc), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y), "v"(z)));
        x += acc[3];
    }
    // 18: the same with two accumulators alternating (independent matrix instructions 12 instructions apart)
    {
        f32x16 acc, acc2;
        for (int q = 0; q < 16; ++q) { acc[q] = 0.f; acc2[q] = 1.f; }
        TIMED(18, asm volatile(REP4(REP4("v_mfma_f32_32x32x2_f32 %0, %6, %7, %0\n\t"
                                     "v_fma_f32 %2, %2, %6, %7\n\tv_fma_f32 %3, %3, %6, %7\n\tv_fma_f32 %4, %4, %6, %7\n\tv_fma_f32 %5, %5, %6, %7\n\t"
                                     "v_fma_f32 %2, %2, %6, %7\n\tv_mfma_f32_32x32x2_f32 %1, %6, %7, %1\n\t"
                                     "v_fma_f32 %3, %3, %6, %7\n\tv_fma_f32 %4, %4, %6, %7\n\tv_fma_f32 %5, %5, %6, %7\n\t"
                                     "v_fma_f32 %2, %2, %6, %7\n\tv_fma_f32 %3, %3, %6, %7\n\t"))
                               : "+v"(acc), "+v"(acc2), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y), "v"(z)));
        x += acc[3] + acc2[2];
    }
    // 19: v_mfma_f32_4x4x1 followed by 3 independent v_fma, x 16
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        TIMED(19, asm volatile(REP16("v_mfma_f32_4x4x1_16b_f32 %0, %5, %6, %0\n\t"
                                     "v_fma_f32 %1, %1, %5, %6\n\tv_fma_f32 %2, %2, %5, %6\n\tv_fma_f32 %3, %3, %5, %6\n\t")
                               : "+v"(acc), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y), "v"(z)));
        x += acc[1];
    }
    out[l] = x + a0 + a1 + a2 + a3 + y;
}
int main() {
    float* o; unsigned long long* c;
    (void)hipMalloc(&o, 64 * 4); (void)hipMalloc(&c, 32 * 8); (void)hipMemset(c, 0, 32 * 8);
    for (int rep = 0; rep < 2; ++rep) k<<<1, 64>>>(o, c, 1.5f);
    (void)hipDeviceSynchronize();
    unsigned long long h[32]; (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"v_fma dependent", "v_fma independent", "v_rcp dependent (+nop)", "v_rcp independent",
                           "readlane -> v_mul(s) link (nop 3)", "v_readlane independent", "v_add dpp quad_perm dependent (+nop 1)",
                           "v_permlane32_swap dependent (+nop 1)", "ds_bpermute dependent", "ds_write -> ds_read round trip",
                           "v_mfma 32x32x2 dependent", "v_mfma 4x4x1 dependent", "readlane->rcp->mul->fma link", "readlane->rcp link",
                           "s_nop 0", "v_mov independent", "readfirstlane -> v_mul(s) link (nop 3)",
                           "[mfma 32x32x2 + 11 indep. v_fma] x 16 (per 64: x 4)", "[2 indep. mfma 32x32x2 + 11 v_fma] x 16",
                           "[mfma 4x4x1 + 3 indep. v_fma] x 16"};
    // s_memtime = the shader-clock counter clock64() reads: cycles per 64 links and per link
    for (int i = 0; i < 20; ++i) printf("%-42s %8llu cycles per 64 links  = %7.2f per link\n", names[i], h[i], h[i] / 64.0);
    printf("(cycles of the shader clock)\n");
    return 0;
}
