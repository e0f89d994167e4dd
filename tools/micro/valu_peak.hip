// valu_peak.hip — measures the VALU issue rate of gfx950 for the instruction classes the FAST / describe / pyramid kernels are made of,
// at 1..8 waves per SIMD: wave-instructions per cycle and CU (cycles from s_memtime, wall from HIP events).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_peak.hip -o gpurun_out/valu_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 512
#define OPS_PER_ITER 32

template <int OP>
__global__ __launch_bounds__(64) void k_issue(uint32_t* out, unsigned long long* cyc)
{
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17u + i * 0x01010101u + blockIdx.x;
    const uint32_t c = 0x00010001u + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N_ITER; it++) {
#pragma unroll
        for (int k = 0; k < OPS_PER_ITER; k++) {
            uint32_t& x = r[k & 15];
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 1) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 2) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(x) : "v"(c));
            if (OP == 3) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(x) : "v"(c));
            if (OP == 4) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(r[(k + 1) & 15]));
            if (OP == 5) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(x) : "v"(c));
            if (OP == 6) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
            if (OP == 7) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 9) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(x));
            if (OP == 10) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 11) asm volatile("v_dot2_u32_u16 %0, %0, %1, %0" : "+v"(x) : "v"(c));
            if (OP == 12) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 13) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 14) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 15) asm volatile("v_lshrrev_b32 %0, 2, %0" : "+v"(x));
            if (OP == 16) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xa8" : "+v"(x) : "v"(c), "v"(r[(k + 1) & 15]));
            if (OP == 17) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(x) : "v"(c));
            if (OP == 18) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(r[(k + 1) & 15]));
            if (OP == 19) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(r[(k + 1) & 15]));
            if (OP == 20) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 21) asm volatile("v_add_u32 %0, %0, %1\n v_pk_min_u16 %2, %2, %1" : "+v"(x), "+v"(r[(k + 8) & 15]) : "v"(c));
            if (OP == 22) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 23) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(x) : "v"(c));
            if (OP == 24) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(x) : "v"(c));
            if (OP == 25) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(r[(k + 1) & 15]));
            if (OP == 26) asm volatile("v_min_i32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 27) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 28) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x) : "v"(c));
            if (OP == 29) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 30) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(x) : "v"(c));
            if (OP == 31) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x), "v"(c) : "vcc");
            if (OP == 32) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(x) : "v"(r[(k + 1) & 15]), "v"(c) : "vcc");
            if (OP == 33) asm volatile("v_pk_mad_i16 %0, %0, %1, %1" : "+v"(x) : "v"(c));
            if (OP == 34) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1" : : "v"(x), "v"(c) : "s20", "s21");
            if (OP == 35) asm volatile("s_and_b64 s[20:21], s[20:21], s[22:23]" : : : "s20", "s21", "scc");
            if (OP == 36) asm volatile("v_perm_b32 %0, %0, %1, %1\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[24:25], s[24:25], s[22:23]" : "+v"(x) : "v"(c) : "s20", "s21", "s24", "s25", "scc");
            if (OP == 37) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 38) asm volatile("v_sub_u32 %0, %0, %1\n v_alignbit_b32 %2, %2, %0, 31" : "+v"(x), "+v"(r[(k + 8) & 15]) : "v"(c));
            if (OP == 39) asm volatile("v_readlane_b32 s20, %0, 5" : : "v"(x) : "s20");
            if (OP == 40) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\n s_and_b64 s[24:25], s[24:25], s[20:21]\n s_or_b64 s[26:27], s[26:27], s[24:25]" : : "v"(x), "v"(c) : "s20", "s21", "s24", "s25", "s26", "s27", "scc");
            // round 5: the rest of what k_fast_rows is made of
            if (OP == 41) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(r[(k + 1) & 15]));
            if (OP == 42) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(x));
            if (OP == 43) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(c));      // (vcc is only read: declaring it clobbered makes the compiler put an s_nop between the statements)
            if (OP == 60) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x) : "v"(c));
            if (OP == 44) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 45) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(r[(k + 1) & 15]));
            if (OP == 46) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(r[(k + 1) & 15]));
            if (OP == 47) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(r[(k + 1) & 15]));
            if (OP == 48) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x) : "v"(c));
            if (OP == 49) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(r[(k + 1) & 15]));
            if (OP == 50) asm volatile("v_ffbl_b32 %0, %0" : "+v"(x));
            if (OP == 51) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(x) : "v"(c));
            if (OP == 52) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(x) : "v"(c));
            if (OP == 53) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(x));
            if (OP == 54) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(r[(k + 1) & 15]));
            if (OP == 55) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(x));                     // VOP2 with a 32-bit literal (8-byte encoding)
            if (OP == 56) asm volatile("v_and_b32 %0, 0x3f3f3f3f, %0" : "+v"(x));                  // the scan's mask: a literal too
            if (OP == 57) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(x) : "v"(c));            // the same add in the 8-byte VOP3 encoding: encoding or operation?
            if (OP == 58) asm volatile("v_xor_b32 %0, %0, %1\n v_pk_minimum3_f16 %2, %2, %1, %0" : "+v"(x), "+v"(r[(k + 8) & 15]) : "v"(c));
            if (OP == 59) asm volatile("v_sub_u32 %0, %0, %1\n v_or_b32 %2, %2, %0\n v_and_b32 %2, %2, %1\n v_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x), "+v"(r[(k + 8) & 15]) : "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s ^= r[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP> void run(const char* name, uint32_t* d_out, unsigned long long* d_cyc)
{
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        const int nblk = 256 * 4 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_issue<OP>, dim3(nblk), dim3(64), 0, 0, d_out, d_cyc);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_issue<OP>, dim3(nblk), dim3(64), 0, 0, d_out, d_cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(nblk);
        hipMemcpy(c.data(), d_cyc, nblk * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : c) avg += v; avg /= nblk;
        const double insts = (double)N_ITER * OPS_PER_ITER;
        // per CU: 4*wps waves each issuing `insts` in `avg` cycles (memtime runs at 100 MHz on gfx9: convert with wall time instead)
        const double per_cu_wall = 4.0 * wps * insts / (ms * 1e-3 * 2.4e9);
        // (round 5) also per s_memtime tick — the counter ticks with the shader clock under load (~2.0 GHz here), so this is the rate per REAL cycle:
        // 4 SIMDs x wps waves x insts / ticks of a wave
        printf("%-22s waves/SIMD %d: %.3f ms  -> %.2f wave-instr/cycle/CU @2.4GHz (memtime ticks/wave %.0f = %.2f per tick and CU)\n", name, wps, ms, per_cu_wall, avg, 4.0 * wps * insts / avg);
    }
}

int main()
{
    uint32_t* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_out, 256 * 4 * 8 * 64 * 4); hipMalloc(&d_cyc, 256 * 4 * 8 * 8);      // up to 8 waves per SIMD
    run<0>("v_add_u32", d_out, d_cyc);
    run<1>("v_pk_min_u16", d_out, d_cyc);
    run<2>("v_pk_sub_u16 clamp", d_out, d_cyc);
    run<3>("v_alignbyte_b32", d_out, d_cyc);
    run<4>("v_mov_b32_dpp wave_shr", d_out, d_cyc);
    run<5>("v_dot4_u32_u8", d_out, d_cyc);
    run<6>("v_perm_b32", d_out, d_cyc);
    run<7>("v_min_u32", d_out, d_cyc);
    run<8>("v_mul_lo_u32", d_out, d_cyc);
    run<9>("v_bfe_u32", d_out, d_cyc);
    run<10>("v_pk_max_u16", d_out, d_cyc);
    run<11>("v_dot2_u32_u16", d_out, d_cyc);
    run<12>("v_sub_u32", d_out, d_cyc);
    run<13>("v_and_b32", d_out, d_cyc);
    run<14>("v_or_b32", d_out, d_cyc);
    run<15>("v_lshrrev_b32", d_out, d_cyc);
    run<16>("v_bitop3_b32", d_out, d_cyc);
    run<17>("v_lshl_or_b32", d_out, d_cyc);
    run<18>("v_and_or_b32", d_out, d_cyc);
    run<19>("v_add3_u32", d_out, d_cyc);
    run<20>("v_xor_b32", d_out, d_cyc);
    run<21>("v_add_u32+v_pk_min_u16 (x2)", d_out, d_cyc);
    run<22>("v_pk_add_u16", d_out, d_cyc);
    run<23>("v_sad_u8", d_out, d_cyc);
    run<24>("v_alignbit_b32", d_out, d_cyc);
    run<25>("v_max3_u32", d_out, d_cyc);
    run<26>("v_min_i32", d_out, d_cyc);
    run<27>("v_pk_mul_lo_u16", d_out, d_cyc);
    run<28>("v_mad_u32_u24", d_out, d_cyc);
    run<29>("v_mul_u32_u24", d_out, d_cyc);
    run<30>("v_sub_u32_sdwa byte", d_out, d_cyc);
    run<31>("v_cmp_lt_u32 vcc", d_out, d_cyc);
    run<32>("v_cmp+v_addc (x2)", d_out, d_cyc);
    run<33>("v_pk_mad_i16", d_out, d_cyc);
    run<34>("v_cmp_lt_u32_e64 sgpr", d_out, d_cyc);
    run<35>("s_and_b64 (SALU)", d_out, d_cyc);
    run<36>("v_perm + 2 SALU (x3)", d_out, d_cyc);
    run<37>("v_pk_min_i16", d_out, d_cyc);
    run<38>("v_sub+v_alignbit (x2)", d_out, d_cyc);
    run<39>("v_readlane_b32", d_out, d_cyc);
    run<40>("v_cmp_e64 + 2 SALU (x3)", d_out, d_cyc);
    run<41>("v_mov_b32", d_out, d_cyc);
    run<42>("v_lshlrev_b32", d_out, d_cyc);
    run<43>("v_cndmask_b32", d_out, d_cyc);
    run<44>("v_bcnt_u32_b32", d_out, d_cyc);
    run<45>("v_min3_i32", d_out, d_cyc);
    run<46>("v_pk_minimum3_f16", d_out, d_cyc);
    run<47>("v_pk_maximum3_f16", d_out, d_cyc);
    run<48>("v_lshl_add_u32", d_out, d_cyc);
    run<49>("v_or3_b32", d_out, d_cyc);
    run<50>("v_ffbl_b32", d_out, d_cyc);
    run<51>("v_mbcnt_lo_u32_b32", d_out, d_cyc);
    run<52>("v_mad_i32_i24", d_out, d_cyc);
    run<53>("v_ashrrev_i32", d_out, d_cyc);
    run<54>("v_bfi_b32", d_out, d_cyc);
    run<55>("v_add_u32 literal", d_out, d_cyc);
    run<56>("v_and_b32 literal", d_out, d_cyc);
    run<57>("v_add_u32_e64 (VOP3)", d_out, d_cyc);
    run<58>("v_xor+v_pk_min3_f16 (x2)", d_out, d_cyc);
    run<59>("scan mix 3 plain + 1 dpp (x4)", d_out, d_cyc);
    run<60>("v_cndmask_b32_e64 sgpr", d_out, d_cyc);
    return 0;
}
