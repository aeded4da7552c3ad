// tools/ubench.hip -- instruction-rate microbenchmark for gfx950 (diagnostic tool, not part of the library).
// For each vector instruction: cycles per wave-instruction as seen by one wave (s_memtime) with 1, 2, 4, 8 waves per SIMD,
// independent (16 registers) and dependent (one register chain) streams.  Build: make -C tools; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <string>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// OPI(k): independent form on register r[k]; OPD: dependent form on r[0]
#define DEFK(NAME, ASM_IND, ASM_DEP)                                                                      \
    __global__ __launch_bounds__(1024) void ki_##NAME(uint32_t* out, int iters, unsigned long long* cyc) { \
        uint32_t r[16];                                                                                   \
        for (int k = 0; k < 16; k++) r[k] = threadIdx.x * 2654435761u + k * 40503u + 12345u;              \
        uint32_t c = threadIdx.x | 0x01010101u, e = (threadIdx.x * 7u) | 3u;                              \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                             \
        for (int i = 0; i < iters; i++) {                                                                 \
            _Pragma("unroll") for (int k = 0; k < 16; k++) asm volatile(ASM_IND : "+v"(r[k]) : "v"(c), "v"(e)); \
        }                                                                                                 \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                             \
        uint32_t s = 0;                                                                                   \
        for (int k = 0; k < 16; k++) s ^= r[k];                                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                          \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                  \
    }                                                                                                     \
    __global__ __launch_bounds__(1024) void kd_##NAME(uint32_t* out, int iters, unsigned long long* cyc) { \
        uint32_t r = threadIdx.x * 2654435761u + 12345u;                                                  \
        uint32_t c = threadIdx.x | 0x01010101u, e = (threadIdx.x * 7u) | 3u;                              \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                             \
        for (int i = 0; i < iters; i++) {                                                                 \
            _Pragma("unroll") for (int k = 0; k < 16; k++) asm volatile(ASM_DEP : "+v"(r) : "v"(c), "v"(e)); \
        }                                                                                                 \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                          \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                  \
    }

DEFK(xor, "v_xor_b32 %0, %0, %1", "v_xor_b32 %0, %0, %1")
DEFK(xor4, "v_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %2\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %2", "v_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %2\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %2")
DEFK(bcnt4, "v_bcnt_u32_b32 %0, %1, %0\n\tv_bcnt_u32_b32 %0, %2, %0\n\tv_bcnt_u32_b32 %0, %1, %0\n\tv_bcnt_u32_b32 %0, %2, %0", "v_bcnt_u32_b32 %0, %1, %0\n\tv_bcnt_u32_b32 %0, %2, %0\n\tv_bcnt_u32_b32 %0, %1, %0\n\tv_bcnt_u32_b32 %0, %2, %0")
DEFK(mix, "v_xor_b32 %0, %0, %1\n\tv_bcnt_u32_b32 %0, %2, %0", "v_xor_b32 %0, %0, %1\n\tv_bcnt_u32_b32 %0, %2, %0")
DEFK(bcnt, "v_bcnt_u32_b32 %0, %1, %0", "v_bcnt_u32_b32 %0, %1, %0")
DEFK(bcnt_src, "v_bcnt_u32_b32 %0, %0, %1", "v_bcnt_u32_b32 %0, %0, %1")
DEFK(med3u, "v_med3_u32 %0, %0, %1, %2", "v_med3_u32 %0, %0, %1, %2")
DEFK(minu, "v_min_u32 %0, %0, %1", "v_min_u32 %0, %0, %1")
DEFK(max3i, "v_max3_i32 %0, %0, %1, %2", "v_max3_i32 %0, %0, %1, %2")
DEFK(lshl_or, "v_lshl_or_b32 %0, %0, 3, %1", "v_lshl_or_b32 %0, %0, 3, %1")
DEFK(add_u32, "v_add_u32 %0, %0, %1", "v_add_u32 %0, %0, %1")
DEFK(pk_min_i16, "v_pk_min_i16 %0, %0, %1", "v_pk_min_i16 %0, %0, %1")
DEFK(pk_max_i16, "v_pk_max_i16 %0, %0, %1", "v_pk_max_i16 %0, %0, %1")
DEFK(pk_sub_i16, "v_pk_sub_i16 %0, %0, %1", "v_pk_sub_i16 %0, %0, %1")
DEFK(pk_add_u16, "v_pk_add_u16 %0, %0, %1", "v_pk_add_u16 %0, %0, %1")
DEFK(perm, "v_perm_b32 %0, %0, %1, %2", "v_perm_b32 %0, %0, %1, %2")
DEFK(alignbit, "v_alignbit_b32 %0, %0, %1, 31", "v_alignbit_b32 %0, %0, %1, 31")
DEFK(alignbyte, "v_alignbyte_b32 %0, %0, %1, 1", "v_alignbyte_b32 %0, %0, %1, 1")
DEFK(mad_u24, "v_mad_u32_u24 %0, %0, %1, %2", "v_mad_u32_u24 %0, %0, %1, %2")
DEFK(mul_lo, "v_mul_lo_u32 %0, %0, %1", "v_mul_lo_u32 %0, %0, %1")
DEFK(mul_f32, "v_mul_f32 %0, %0, %1", "v_mul_f32 %0, %0, %1")
DEFK(fma_f32, "v_fma_f32 %0, %0, %1, %2", "v_fma_f32 %0, %0, %1, %2")
DEFK(cvt_i32_f32, "v_cvt_i32_f32 %0, %0", "v_cvt_i32_f32 %0, %0")
DEFK(rndne_f32, "v_rndne_f32 %0, %0", "v_rndne_f32 %0, %0")
DEFK(cvt_f32_ubyte1, "v_cvt_f32_ubyte1 %0, %0", "v_cvt_f32_ubyte1 %0, %0")
DEFK(dot4_u8, "v_dot4_u32_u8 %0, %1, %2, %0", "v_dot4_u32_u8 %0, %1, %2, %0")
DEFK(dot2_u16, "v_dot2_u32_u16 %0, %1, %2, %0", "v_dot2_u32_u16 %0, %1, %2, %0")
DEFK(sad_u8, "v_sad_u8 %0, %1, %2, %0", "v_sad_u8 %0, %1, %2, %0")
DEFK(bfe_u32, "v_bfe_u32 %0, %0, 3, 8", "v_bfe_u32 %0, %0, 3, 8")
DEFK(cmp_cndmask, "v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc", "v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
DEFK(mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0", "v_mbcnt_lo_u32_b32 %0, %1, %0")

DEFK(and, "v_and_b32 %0, %0, %1\n\tv_and_b32 %0, %0, %1\n\tv_and_b32 %0, %0, %1\n\tv_and_b32 %0, %0, %1", "v_and_b32 %0, %0, %1\n\tv_and_b32 %0, %0, %1\n\tv_and_b32 %0, %0, %1\n\tv_and_b32 %0, %0, %1")
DEFK(or, "v_or_b32 %0, %0, %1\n\tv_or_b32 %0, %0, %1\n\tv_or_b32 %0, %0, %1\n\tv_or_b32 %0, %0, %1", "v_or_b32 %0, %0, %1\n\tv_or_b32 %0, %0, %1\n\tv_or_b32 %0, %0, %1\n\tv_or_b32 %0, %0, %1")
DEFK(not, "v_not_b32 %0, %0\n\tv_not_b32 %0, %0\n\tv_not_b32 %0, %0\n\tv_not_b32 %0, %0", "v_not_b32 %0, %0\n\tv_not_b32 %0, %0\n\tv_not_b32 %0, %0\n\tv_not_b32 %0, %0")
DEFK(lshl, "v_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %0, 1, %0", "v_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %0, 1, %0")
DEFK(lshr, "v_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %0, 1, %0", "v_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %0, 1, %0")
DEFK(ashr, "v_ashrrev_i32 %0, 1, %0\n\tv_ashrrev_i32 %0, 1, %0\n\tv_ashrrev_i32 %0, 1, %0\n\tv_ashrrev_i32 %0, 1, %0", "v_ashrrev_i32 %0, 1, %0\n\tv_ashrrev_i32 %0, 1, %0\n\tv_ashrrev_i32 %0, 1, %0\n\tv_ashrrev_i32 %0, 1, %0")
DEFK(sub, "v_sub_u32 %0, %0, %1\n\tv_sub_u32 %0, %0, %1\n\tv_sub_u32 %0, %0, %1\n\tv_sub_u32 %0, %0, %1", "v_sub_u32 %0, %0, %1\n\tv_sub_u32 %0, %0, %1\n\tv_sub_u32 %0, %0, %1\n\tv_sub_u32 %0, %0, %1")
DEFK(max_i32, "v_max_i32 %0, %0, %1\n\tv_max_i32 %0, %0, %1\n\tv_max_i32 %0, %0, %1\n\tv_max_i32 %0, %0, %1", "v_max_i32 %0, %0, %1\n\tv_max_i32 %0, %0, %1\n\tv_max_i32 %0, %0, %1\n\tv_max_i32 %0, %0, %1")
DEFK(max_u32, "v_max_u32 %0, %0, %1\n\tv_max_u32 %0, %0, %1\n\tv_max_u32 %0, %0, %1\n\tv_max_u32 %0, %0, %1", "v_max_u32 %0, %0, %1\n\tv_max_u32 %0, %0, %1\n\tv_max_u32 %0, %0, %1\n\tv_max_u32 %0, %0, %1")
DEFK(min_i32, "v_min_i32 %0, %0, %1\n\tv_min_i32 %0, %0, %1\n\tv_min_i32 %0, %0, %1\n\tv_min_i32 %0, %0, %1", "v_min_i32 %0, %0, %1\n\tv_min_i32 %0, %0, %1\n\tv_min_i32 %0, %0, %1\n\tv_min_i32 %0, %0, %1")
DEFK(cndmask, "v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc", "v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc")
DEFK(mov, "v_mov_b32 %0, %1\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %0, %1", "v_mov_b32 %0, %1\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %0, %1")
DEFK(add3, "v_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %0, %0, %1, %2", "v_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %0, %0, %1, %2")
DEFK(and_or, "v_and_or_b32 %0, %0, %1, %2\n\tv_and_or_b32 %0, %0, %1, %2\n\tv_and_or_b32 %0, %0, %1, %2\n\tv_and_or_b32 %0, %0, %1, %2", "v_and_or_b32 %0, %0, %1, %2\n\tv_and_or_b32 %0, %0, %1, %2\n\tv_and_or_b32 %0, %0, %1, %2\n\tv_and_or_b32 %0, %0, %1, %2")
DEFK(or3, "v_or3_b32 %0, %0, %1, %2\n\tv_or3_b32 %0, %0, %1, %2\n\tv_or3_b32 %0, %0, %1, %2\n\tv_or3_b32 %0, %0, %1, %2", "v_or3_b32 %0, %0, %1, %2\n\tv_or3_b32 %0, %0, %1, %2\n\tv_or3_b32 %0, %0, %1, %2\n\tv_or3_b32 %0, %0, %1, %2")
DEFK(lshl_add, "v_lshl_add_u32 %0, %0, 1, %1\n\tv_lshl_add_u32 %0, %0, 1, %1\n\tv_lshl_add_u32 %0, %0, 1, %1\n\tv_lshl_add_u32 %0, %0, 1, %1", "v_lshl_add_u32 %0, %0, 1, %1\n\tv_lshl_add_u32 %0, %0, 1, %1\n\tv_lshl_add_u32 %0, %0, 1, %1\n\tv_lshl_add_u32 %0, %0, 1, %1")
DEFK(add_lshl, "v_add_lshl_u32 %0, %0, %1, 1\n\tv_add_lshl_u32 %0, %0, %1, 1\n\tv_add_lshl_u32 %0, %0, %1, 1\n\tv_add_lshl_u32 %0, %0, %1, 1", "v_add_lshl_u32 %0, %0, %1, 1\n\tv_add_lshl_u32 %0, %0, %1, 1\n\tv_add_lshl_u32 %0, %0, %1, 1\n\tv_add_lshl_u32 %0, %0, %1, 1")
DEFK(bfi, "v_bfi_b32 %0, %0, %1, %2\n\tv_bfi_b32 %0, %0, %1, %2\n\tv_bfi_b32 %0, %0, %1, %2\n\tv_bfi_b32 %0, %0, %1, %2", "v_bfi_b32 %0, %0, %1, %2\n\tv_bfi_b32 %0, %0, %1, %2\n\tv_bfi_b32 %0, %0, %1, %2\n\tv_bfi_b32 %0, %0, %1, %2")
DEFK(sub_u16, "v_sub_u16 %0, %0, %1\n\tv_sub_u16 %0, %0, %1\n\tv_sub_u16 %0, %0, %1\n\tv_sub_u16 %0, %0, %1", "v_sub_u16 %0, %0, %1\n\tv_sub_u16 %0, %0, %1\n\tv_sub_u16 %0, %0, %1\n\tv_sub_u16 %0, %0, %1")
DEFK(add_u16, "v_add_u16 %0, %0, %1\n\tv_add_u16 %0, %0, %1\n\tv_add_u16 %0, %0, %1\n\tv_add_u16 %0, %0, %1", "v_add_u16 %0, %0, %1\n\tv_add_u16 %0, %0, %1\n\tv_add_u16 %0, %0, %1\n\tv_add_u16 %0, %0, %1")
DEFK(max_u16, "v_max_u16 %0, %0, %1\n\tv_max_u16 %0, %0, %1\n\tv_max_u16 %0, %0, %1\n\tv_max_u16 %0, %0, %1", "v_max_u16 %0, %0, %1\n\tv_max_u16 %0, %0, %1\n\tv_max_u16 %0, %0, %1\n\tv_max_u16 %0, %0, %1")
DEFK(min_i16, "v_min_i16 %0, %0, %1\n\tv_min_i16 %0, %0, %1\n\tv_min_i16 %0, %0, %1\n\tv_min_i16 %0, %0, %1", "v_min_i16 %0, %0, %1\n\tv_min_i16 %0, %0, %1\n\tv_min_i16 %0, %0, %1\n\tv_min_i16 %0, %0, %1")
DEFK(mul_u24, "v_mul_u32_u24 %0, %0, %1\n\tv_mul_u32_u24 %0, %0, %1\n\tv_mul_u32_u24 %0, %0, %1\n\tv_mul_u32_u24 %0, %0, %1", "v_mul_u32_u24 %0, %0, %1\n\tv_mul_u32_u24 %0, %0, %1\n\tv_mul_u32_u24 %0, %0, %1\n\tv_mul_u32_u24 %0, %0, %1")
DEFK(mul_i24, "v_mul_i32_i24 %0, %0, %1\n\tv_mul_i32_i24 %0, %0, %1\n\tv_mul_i32_i24 %0, %0, %1\n\tv_mul_i32_i24 %0, %0, %1", "v_mul_i32_i24 %0, %0, %1\n\tv_mul_i32_i24 %0, %0, %1\n\tv_mul_i32_i24 %0, %0, %1\n\tv_mul_i32_i24 %0, %0, %1")
DEFK(add_f32, "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1", "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1")
DEFK(sub_f32, "v_sub_f32 %0, %0, %1\n\tv_sub_f32 %0, %0, %1\n\tv_sub_f32 %0, %0, %1\n\tv_sub_f32 %0, %0, %1", "v_sub_f32 %0, %0, %1\n\tv_sub_f32 %0, %0, %1\n\tv_sub_f32 %0, %0, %1\n\tv_sub_f32 %0, %0, %1")
DEFK(max_f32, "v_max_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %1", "v_max_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %1")
DEFK(min_f32, "v_min_f32 %0, %0, %1\n\tv_min_f32 %0, %0, %1\n\tv_min_f32 %0, %0, %1\n\tv_min_f32 %0, %0, %1", "v_min_f32 %0, %0, %1\n\tv_min_f32 %0, %0, %1\n\tv_min_f32 %0, %0, %1\n\tv_min_f32 %0, %0, %1")
DEFK(fmac_f32, "v_fmac_f32 %0, %1, %2\n\tv_fmac_f32 %0, %1, %2\n\tv_fmac_f32 %0, %1, %2\n\tv_fmac_f32 %0, %1, %2", "v_fmac_f32 %0, %1, %2\n\tv_fmac_f32 %0, %1, %2\n\tv_fmac_f32 %0, %1, %2\n\tv_fmac_f32 %0, %1, %2")
DEFK(cvt_f32_i32, "v_cvt_f32_i32 %0, %0\n\tv_cvt_f32_i32 %0, %0\n\tv_cvt_f32_i32 %0, %0\n\tv_cvt_f32_i32 %0, %0", "v_cvt_f32_i32 %0, %0\n\tv_cvt_f32_i32 %0, %0\n\tv_cvt_f32_i32 %0, %0\n\tv_cvt_f32_i32 %0, %0")
DEFK(cvt_f32_u32, "v_cvt_f32_u32 %0, %0\n\tv_cvt_f32_u32 %0, %0\n\tv_cvt_f32_u32 %0, %0\n\tv_cvt_f32_u32 %0, %0", "v_cvt_f32_u32 %0, %0\n\tv_cvt_f32_u32 %0, %0\n\tv_cvt_f32_u32 %0, %0\n\tv_cvt_f32_u32 %0, %0")
DEFK(cvt_u32_f32, "v_cvt_u32_f32 %0, %0\n\tv_cvt_u32_f32 %0, %0\n\tv_cvt_u32_f32 %0, %0\n\tv_cvt_u32_f32 %0, %0", "v_cvt_u32_f32 %0, %0\n\tv_cvt_u32_f32 %0, %0\n\tv_cvt_u32_f32 %0, %0\n\tv_cvt_u32_f32 %0, %0")
DEFK(cmp_lt_i32, "v_cmp_lt_i32 vcc, %0, %1\n\tv_cmp_lt_i32 vcc, %0, %1\n\tv_cmp_lt_i32 vcc, %0, %1\n\tv_cmp_lt_i32 vcc, %0, %1", "v_cmp_lt_i32 vcc, %0, %1\n\tv_cmp_lt_i32 vcc, %0, %1\n\tv_cmp_lt_i32 vcc, %0, %1\n\tv_cmp_lt_i32 vcc, %0, %1")
DEFK(cmp_lt_u32, "v_cmp_lt_u32 vcc, %0, %1\n\tv_cmp_lt_u32 vcc, %0, %1\n\tv_cmp_lt_u32 vcc, %0, %1\n\tv_cmp_lt_u32 vcc, %0, %1", "v_cmp_lt_u32 vcc, %0, %1\n\tv_cmp_lt_u32 vcc, %0, %1\n\tv_cmp_lt_u32 vcc, %0, %1\n\tv_cmp_lt_u32 vcc, %0, %1")
DEFK(cmp_lt_f32, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cmp_lt_f32 vcc, %0, %1\n\tv_cmp_lt_f32 vcc, %0, %1\n\tv_cmp_lt_f32 vcc, %0, %1", "v_cmp_lt_f32 vcc, %0, %1\n\tv_cmp_lt_f32 vcc, %0, %1\n\tv_cmp_lt_f32 vcc, %0, %1\n\tv_cmp_lt_f32 vcc, %0, %1")
DEFK(cmp_e64, "v_cmp_lt_i32 s[10:11], %0, %1\n\tv_cmp_lt_i32 s[10:11], %0, %1\n\tv_cmp_lt_i32 s[10:11], %0, %1\n\tv_cmp_lt_i32 s[10:11], %0, %1", "v_cmp_lt_i32 s[10:11], %0, %1\n\tv_cmp_lt_i32 s[10:11], %0, %1\n\tv_cmp_lt_i32 s[10:11], %0, %1\n\tv_cmp_lt_i32 s[10:11], %0, %1")
DEFK(dpp_shr, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEFK(add_dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
DEFK(sdwa_sub, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1", "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1")
DEFK(sdwa_max, "v_max_i32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_max_i32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_max_i32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_max_i32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1", "v_max_i32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_max_i32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_max_i32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_max_i32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1")
DEFK(sdwa_xor, "v_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1", "v_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1")
DEFK(pk_lshl_b16, "v_pk_lshlrev_b16 %0, 1, %0\n\tv_pk_lshlrev_b16 %0, 1, %0\n\tv_pk_lshlrev_b16 %0, 1, %0\n\tv_pk_lshlrev_b16 %0, 1, %0", "v_pk_lshlrev_b16 %0, 1, %0\n\tv_pk_lshlrev_b16 %0, 1, %0\n\tv_pk_lshlrev_b16 %0, 1, %0\n\tv_pk_lshlrev_b16 %0, 1, %0")
DEFK(pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1\n\tv_pk_mul_lo_u16 %0, %0, %1\n\tv_pk_mul_lo_u16 %0, %0, %1\n\tv_pk_mul_lo_u16 %0, %0, %1", "v_pk_mul_lo_u16 %0, %0, %1\n\tv_pk_mul_lo_u16 %0, %0, %1\n\tv_pk_mul_lo_u16 %0, %0, %1\n\tv_pk_mul_lo_u16 %0, %0, %1")
DEFK(pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2\n\tv_pk_mad_u16 %0, %0, %1, %2\n\tv_pk_mad_u16 %0, %0, %1, %2\n\tv_pk_mad_u16 %0, %0, %1, %2", "v_pk_mad_u16 %0, %0, %1, %2\n\tv_pk_mad_u16 %0, %0, %1, %2\n\tv_pk_mad_u16 %0, %0, %1, %2\n\tv_pk_mad_u16 %0, %0, %1, %2")
DEFK(med3_i32, "v_med3_i32 %0, %0, %1, %2\n\tv_med3_i32 %0, %0, %1, %2\n\tv_med3_i32 %0, %0, %1, %2\n\tv_med3_i32 %0, %0, %1, %2", "v_med3_i32 %0, %0, %1, %2\n\tv_med3_i32 %0, %0, %1, %2\n\tv_med3_i32 %0, %0, %1, %2\n\tv_med3_i32 %0, %0, %1, %2")
DEFK(min3_u32, "v_min3_u32 %0, %0, %1, %2\n\tv_min3_u32 %0, %0, %1, %2\n\tv_min3_u32 %0, %0, %1, %2\n\tv_min3_u32 %0, %0, %1, %2", "v_min3_u32 %0, %0, %1, %2\n\tv_min3_u32 %0, %0, %1, %2\n\tv_min3_u32 %0, %0, %1, %2\n\tv_min3_u32 %0, %0, %1, %2")
DEFK(sad_u16, "v_sad_u16 %0, %0, %1, %2\n\tv_sad_u16 %0, %0, %1, %2\n\tv_sad_u16 %0, %0, %1, %2\n\tv_sad_u16 %0, %0, %1, %2", "v_sad_u16 %0, %0, %1, %2\n\tv_sad_u16 %0, %0, %1, %2\n\tv_sad_u16 %0, %0, %1, %2\n\tv_sad_u16 %0, %0, %1, %2")
DEFK(msad_u8, "v_msad_u8 %0, %0, %1, %2\n\tv_msad_u8 %0, %0, %1, %2\n\tv_msad_u8 %0, %0, %1, %2\n\tv_msad_u8 %0, %0, %1, %2", "v_msad_u8 %0, %0, %1, %2\n\tv_msad_u8 %0, %0, %1, %2\n\tv_msad_u8 %0, %0, %1, %2\n\tv_msad_u8 %0, %0, %1, %2")
DEFK(lerp_u8, "v_lerp_u8 %0, %0, %1, %2\n\tv_lerp_u8 %0, %0, %1, %2\n\tv_lerp_u8 %0, %0, %1, %2\n\tv_lerp_u8 %0, %0, %1, %2", "v_lerp_u8 %0, %0, %1, %2\n\tv_lerp_u8 %0, %0, %1, %2\n\tv_lerp_u8 %0, %0, %1, %2\n\tv_lerp_u8 %0, %0, %1, %2")
DEFK(cvt_pk_u8, "v_cvt_pk_u8_f32 %0, %0, %1, %2\n\tv_cvt_pk_u8_f32 %0, %0, %1, %2\n\tv_cvt_pk_u8_f32 %0, %0, %1, %2\n\tv_cvt_pk_u8_f32 %0, %0, %1, %2", "v_cvt_pk_u8_f32 %0, %0, %1, %2\n\tv_cvt_pk_u8_f32 %0, %0, %1, %2\n\tv_cvt_pk_u8_f32 %0, %0, %1, %2\n\tv_cvt_pk_u8_f32 %0, %0, %1, %2")
DEFK(ffbh, "v_ffbh_u32 %0, %0\n\tv_ffbh_u32 %0, %0\n\tv_ffbh_u32 %0, %0\n\tv_ffbh_u32 %0, %0", "v_ffbh_u32 %0, %0\n\tv_ffbh_u32 %0, %0\n\tv_ffbh_u32 %0, %0\n\tv_ffbh_u32 %0, %0")
DEFK(bfrev, "v_bfrev_b32 %0, %0\n\tv_bfrev_b32 %0, %0\n\tv_bfrev_b32 %0, %0\n\tv_bfrev_b32 %0, %0", "v_bfrev_b32 %0, %0\n\tv_bfrev_b32 %0, %0\n\tv_bfrev_b32 %0, %0\n\tv_bfrev_b32 %0, %0")
DEFK(readfirstlane, "v_readfirstlane_b32 s10, %0\n\tv_readfirstlane_b32 s10, %0\n\tv_readfirstlane_b32 s10, %0\n\tv_readfirstlane_b32 s10, %0", "v_readfirstlane_b32 s10, %0\n\tv_readfirstlane_b32 s10, %0\n\tv_readfirstlane_b32 s10, %0\n\tv_readfirstlane_b32 s10, %0")
DEFK(rcp_f32, "v_rcp_f32 %0, %0\n\tv_rcp_f32 %0, %0\n\tv_rcp_f32 %0, %0\n\tv_rcp_f32 %0, %0", "v_rcp_f32 %0, %0\n\tv_rcp_f32 %0, %0\n\tv_rcp_f32 %0, %0\n\tv_rcp_f32 %0, %0")
DEFK(sqrt_f32, "v_sqrt_f32 %0, %0\n\tv_sqrt_f32 %0, %0\n\tv_sqrt_f32 %0, %0\n\tv_sqrt_f32 %0, %0", "v_sqrt_f32 %0, %0\n\tv_sqrt_f32 %0, %0\n\tv_sqrt_f32 %0, %0\n\tv_sqrt_f32 %0, %0")
// 64-bit register forms (f64): separate macro
#define DEFK64(NAME, ASM_IND)                                                                             \
    __global__ __launch_bounds__(1024) void ki_##NAME(uint32_t* out, int iters, unsigned long long* cyc) { \
        double r[16];                                                                                     \
        for (int k = 0; k < 16; k++) r[k] = 1.0 + threadIdx.x * 1e-3 + k * 1e-5;                          \
        double c = 1.0 + 1e-9 * threadIdx.x, e = 1e-12;                                                   \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                             \
        for (int i = 0; i < iters; i++) {                                                                 \
            _Pragma("unroll") for (int k = 0; k < 16; k++) asm volatile(ASM_IND : "+v"(r[k]) : "v"(c), "v"(e)); \
        }                                                                                                 \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                             \
        double s = 0;                                                                                     \
        for (int k = 0; k < 16; k++) s += r[k];                                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s;                                                \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                  \
    }                                                                                                     \
    __global__ __launch_bounds__(1024) void kd_##NAME(uint32_t* out, int iters, unsigned long long* cyc) { \
        double r = 1.0 + threadIdx.x * 1e-3;                                                              \
        double c = 1.0 + 1e-9 * threadIdx.x, e = 1e-12;                                                   \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                             \
        for (int i = 0; i < iters; i++) {                                                                 \
            _Pragma("unroll") for (int k = 0; k < 16; k++) asm volatile(ASM_IND : "+v"(r) : "v"(c), "v"(e)); \
        }                                                                                                 \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r;                                                \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                  \
    }
DEFK64(fma_f64, "v_fma_f64 %0, %0, %1, %2")
DEFK64(mul_f64, "v_mul_f64 %0, %0, %1")
DEFK64(add_f64, "v_add_f64 %0, %0, %2")
DEFK64(min_f64, "v_min_f64 %0, %0, %1")
DEFK64(pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
DEFK64(pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
DEFK64(pk_add_f32, "v_pk_add_f32 %0, %0, %2")

__global__ void k_clock(unsigned long long* o, int iters) {
    uint32_t r = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("v_xor_b32 %0, %0, %0" : "+v"(r));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { o[0] = t1 - t0; o[1] = r1 - r0; o[2] = r; }
}

typedef void (*kern_t)(uint32_t*, int, unsigned long long*);
struct Entry { const char* name; kern_t ind, dep; };
#define E(NAME) {#NAME, ki_##NAME, kd_##NAME}

__global__ __launch_bounds__(1024) void k_census(uint32_t* hw, int iters) {
    uint32_t r = threadIdx.x;
    for (int i = 0; i < iters; i++) asm volatile("v_xor_b32 %0, %0, %0" : "+v"(r));
    uint32_t id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
    uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID bits 0..3
    if ((threadIdx.x & 63) == 0) hw[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = (id & 0xFFFFFFu) | (xcc << 24) | (r & 0);
}

int main(int argc, char** argv) {
    const int iters = 5000;
    std::vector<Entry> es = {E(xor4), E(bcnt4), E(and), E(or), E(not), E(lshl), E(lshr), E(ashr), E(sub), E(max_i32), E(max_u32), E(min_i32), E(cndmask), E(mov), E(add3), E(and_or), E(or3), E(lshl_add), E(add_lshl), E(bfi), E(sub_u16), E(add_u16), E(max_u16), E(min_i16), E(mul_u24), E(mul_i24), E(add_f32), E(sub_f32), E(max_f32), E(min_f32), E(fmac_f32), E(cvt_f32_i32), E(cvt_f32_u32), E(cvt_u32_f32), E(cmp_lt_i32), E(cmp_lt_u32), E(cmp_lt_f32), E(cmp_e64), E(dpp_shr), E(add_dpp), E(sdwa_sub), E(sdwa_max), E(sdwa_xor), E(pk_lshl_b16), E(pk_mul_lo_u16), E(pk_mad_u16), E(med3_i32), E(min3_u32), E(sad_u16), E(msad_u8), E(lerp_u8), E(cvt_pk_u8), E(ffbh), E(bfrev), E(readfirstlane), E(rcp_f32), E(sqrt_f32)};
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    uint32_t* d_out; unsigned long long* d_cyc;
    const int maxw = ncu * 32;
    hipMalloc(&d_out, (size_t)maxw * 64 * 4);
    hipMalloc(&d_cyc, (size_t)maxw * 8);
    std::vector<unsigned long long> h(maxw);
    {   // warm the clocks, then relate s_memtime to s_memrealtime (100 MHz) and to the host's event clock
        for (int i = 0; i < 40; i++) hipLaunchKernelGGL(k_clock, dim3(2048), dim3(256), 0, 0, d_cyc, 40000);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k_clock, dim3(256), dim3(256), 0, 0, d_cyc, 40000);
        hipDeviceSynchronize();
        unsigned long long o[3];
        hipMemcpy(o, d_cyc, sizeof(o), hipMemcpyDeviceToHost);
        printf("clock: s_memtime %llu ticks over %.1f us of s_memrealtime -> %.3f GHz\n", o[0], o[1] / 100.0, o[0] / (o[1] * 10.0));
    }
    // geometry: w <= 4 -> ncu blocks of 256 w threads (one per CU if the dispatcher spreads them); w = 8 -> 2 ncu blocks of 1024
    const int nblk = argc > 1 ? atoi(argv[1]) : ncu;
    auto geom = [&](int w, int& blocks, int& threads) { if (w <= 4) { blocks = nblk; threads = 256 * w; } else { blocks = 2 * nblk; threads = 1024; } };
    for (int w : {1, 2, 4, 8}) {  // census: waves per (xcc, se, cu) and per SIMD
        int blocks, threads; geom(w, blocks, threads);
        hipLaunchKernelGGL(k_census, dim3(blocks), dim3(threads), 0, 0, d_out, 200000);
        hipDeviceSynchronize();
        int nw = blocks * threads / 64;
        std::vector<uint32_t> hw(nw);
        hipMemcpy(hw.data(), d_out, (size_t)nw * 4, hipMemcpyDeviceToHost);
        std::vector<int> percu(8 * 64 * 16, 0), persimd(8 * 64 * 16 * 4, 0);
        for (uint32_t v : hw) {
            int simd = (v >> 4) & 3, cu = (v >> 8) & 15, sh = (v >> 12) & 1, se = (v >> 13) & 7, xcc = (v >> 24) & 15;
            int key = ((xcc * 8 + se) * 2 + sh) * 16 + cu;
            percu[key % percu.size()]++; persimd[(key * 4 + simd) % persimd.size()]++;
        }
        int used = 0, mx = 0, smx = 0, sused = 0;
        for (int c : percu) { used += c > 0; mx = std::max(mx, c); }
        for (int c : persimd) { sused += c > 0; smx = std::max(smx, c); }
        printf("census w=%d: %d waves on %d CUs (max %d per CU), %d SIMDs (max %d per SIMD)\n", w, nw, used, mx, sused, smx);
    }
    printf("device %s, %d CUs; ticks (shader cycles) per wave-instruction seen by a wave, median over waves / the same divided by waves per SIMD; launch wall\n", prop.name, ncu);
    printf("%-16s %-4s", "op", "");
    for (int w : {1, 2, 4, 8}) printf("  w/SIMD=%d [min p10 med p90 max] wall   ", w);
    printf("\n");
    auto run = [&](const char* name, const char* kind, auto launch, double per_trip) {
        printf("%-16s %-4s", name, kind);
        for (int w : {1, 2, 4, 8}) {
            int blocks, threads; geom(w, blocks, threads);
            launch(blocks, threads);  // warm
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            launch(blocks, threads);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int nw = blocks * threads / 64;
            hipMemcpy(h.data(), d_cyc, (size_t)nw * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.begin() + nw);
            double sc = 1.0 / (iters * per_trip);
            printf("  [%5.2f %5.2f %5.2f %5.2f %5.2f] %5.0fus", h[0] * sc, h[nw / 10] * sc, h[nw / 2] * sc, h[nw * 9 / 10] * sc, h[nw - 1] * sc, ms * 1e3);
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
        printf("\n");
    };
    for (auto& e : es) {
        std::string nm(e.name); double per = 16.0 * (nm == "cmp_cndmask" || nm == "mix" ? 2 : nm == "xor" || nm == "bcnt" || nm == "fma_f64" ? 1 : 4);
        run(e.name, "ind", [&](int b, int t) { hipLaunchKernelGGL(e.ind, dim3(b), dim3(t), 0, 0, d_out, iters, d_cyc); }, per);

    }
    return 0;
}
