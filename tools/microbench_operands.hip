// microbench_operands.hip -- does the issue cost of the butterfly's VOP3 instructions depend on WHICH registers they read?
// (tools/microbench_ops.hip prices v_mad_u64_u32 at 4.2 cycles with src0 == src1; inside the butterfly block nine of them cost
// 59 cycles.)  Explicit registers: accumulator pairs v[40:41] .. v[54:55], sources v60 .. v67; a pair at an even register
// covers VGPR banks (r % 4, r % 4 + 1).
//   hipcc --offload-arch=gfx950 -O3 -o microbench_operands tools/microbench_operands.hip && ./microbench_operands [waves_per_simd]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
#define REP16(x) x x x x x x x x x x x x x x x x
#define ITER 2048
#define CLOB "vcc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
             "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "s40", "s41", "s42", "s43"

#define KERNEL(NAME, BODY)                                                                      \
    __global__ void __launch_bounds__(256) NAME(u64 *out)                                       \
    {                                                                                           \
        unsigned b = blockIdx.x + 3;                                                            \
        asm volatile("v_mov_b32 v60, %0\n v_mov_b32 v61, %0\n v_mov_b32 v62, %0\n v_mov_b32 v63, %0\n v_mov_b32 v64, %0\n v_mov_b32 v65, %0\n" \
                     "v_mov_b32 v66, %0\n v_mov_b32 v67, %0\n v_mov_b32 v68, %0\n v_mov_b32 v69, %0\n s_mov_b32 s40, 77\n s_mov_b32 s41, 5\n s_mov_b32 s42, 9\n s_mov_b32 s43, 1\n" ::"v"(b) : CLOB); \
        for (int it = 0; it < ITER; it++) asm volatile(REP16(BODY)::: CLOB);                    \
        unsigned r;                                                                             \
        asm volatile("v_xor_b32 %0, v40, v42\n v_xor_b32 %0, %0, v44\n v_xor_b32 %0, %0, v55" : "=v"(r)::CLOB); \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                \
    }

// 8 independent accumulators per body
#define MAD8(S0a, S1a, S0b, S1b)                                                                     \
    "v_mad_u64_u32 v[40:41], vcc, " S0a ", " S1a ", v[40:41]\n v_mad_u64_u32 v[42:43], vcc, " S0b ", " S1b ", v[42:43]\n" \
    "v_mad_u64_u32 v[44:45], vcc, " S0a ", " S1a ", v[44:45]\n v_mad_u64_u32 v[46:47], vcc, " S0b ", " S1b ", v[46:47]\n" \
    "v_mad_u64_u32 v[48:49], vcc, " S0a ", " S1a ", v[48:49]\n v_mad_u64_u32 v[50:51], vcc, " S0b ", " S1b ", v[50:51]\n" \
    "v_mad_u64_u32 v[52:53], vcc, " S0a ", " S1a ", v[52:53]\n v_mad_u64_u32 v[54:55], vcc, " S0b ", " S1b ", v[54:55]\n"
// accumulators at banks (0,1) take sources S0a, S1a; those at banks (2,3) take S0b, S1b
KERNEL(k_same, MAD8("v60", "v60", "v60", "v60"))        // src0 == src1 (microbench_ops)
KERNEL(k_free, MAD8("v62", "v63", "v60", "v61"))        // four different banks per instruction
KERNEL(k_conf2, MAD8("v60", "v61", "v62", "v63"))       // sources in the accumulator's banks
KERNEL(k_conf_all, MAD8("v60", "v64", "v62", "v66"))    // both sources in the bank of the accumulator's low half
KERNEL(k_src_same_bank, MAD8("v62", "v66", "v60", "v64"))  // the two sources share a bank, different from the accumulator's
KERNEL(k_sgpr, MAD8("v62", "s40", "v60", "s40"))
KERNEL(k_sgpr_conf, MAD8("v60", "s40", "v62", "s40"))
#define MADZ8(S0a, S1a, S0b, S1b)                                                                    \
    "v_mad_u64_u32 v[40:41], vcc, " S0a ", " S1a ", 0\n v_mad_u64_u32 v[42:43], vcc, " S0b ", " S1b ", 0\n" \
    "v_mad_u64_u32 v[44:45], vcc, " S0a ", " S1a ", 0\n v_mad_u64_u32 v[46:47], vcc, " S0b ", " S1b ", 0\n" \
    "v_mad_u64_u32 v[48:49], vcc, " S0a ", " S1a ", 0\n v_mad_u64_u32 v[50:51], vcc, " S0b ", " S1b ", 0\n" \
    "v_mad_u64_u32 v[52:53], vcc, " S0a ", " S1a ", 0\n v_mad_u64_u32 v[54:55], vcc, " S0b ", " S1b ", 0\n"
KERNEL(k_zero_free, MADZ8("v62", "v63", "v60", "v61"))
KERNEL(k_zero_conf, MADZ8("v60", "v64", "v62", "v66"))
KERNEL(k_zero_sgpr, MADZ8("v62", "s40", "v60", "s40"))
// the accumulate-into-a-different-pair form the butterfly uses (dst != src2)
KERNEL(k_dst_other, "v_mad_u64_u32 v[40:41], vcc, v62, v63, v[44:45]\n v_mad_u64_u32 v[42:43], vcc, v60, v61, v[46:47]\n"
                    "v_mad_u64_u32 v[44:45], vcc, v62, v63, v[48:49]\n v_mad_u64_u32 v[46:47], vcc, v60, v61, v[50:51]\n"
                    "v_mad_u64_u32 v[48:49], vcc, v62, v63, v[52:53]\n v_mad_u64_u32 v[50:51], vcc, v60, v61, v[54:55]\n"
                    "v_mad_u64_u32 v[52:53], vcc, v62, v63, v[40:41]\n v_mad_u64_u32 v[54:55], vcc, v60, v61, v[42:43]\n")
// 64-bit add: VGPR + VGPR, VGPR + SGPR pair
KERNEL(k_add64_vv, "v_lshl_add_u64 v[40:41], v[40:41], 0, v[42:43]\n v_lshl_add_u64 v[42:43], v[42:43], 0, v[44:45]\n"
                   "v_lshl_add_u64 v[44:45], v[44:45], 0, v[46:47]\n v_lshl_add_u64 v[46:47], v[46:47], 0, v[48:49]\n"
                   "v_lshl_add_u64 v[48:49], v[48:49], 0, v[50:51]\n v_lshl_add_u64 v[50:51], v[50:51], 0, v[52:53]\n"
                   "v_lshl_add_u64 v[52:53], v[52:53], 0, v[54:55]\n v_lshl_add_u64 v[54:55], v[54:55], 0, v[40:41]\n")
KERNEL(k_add64_vs, "v_lshl_add_u64 v[40:41], v[40:41], 0, s[40:41]\n v_lshl_add_u64 v[42:43], v[42:43], 0, s[40:41]\n"
                   "v_lshl_add_u64 v[44:45], v[44:45], 0, s[40:41]\n v_lshl_add_u64 v[46:47], v[46:47], 0, s[40:41]\n"
                   "v_lshl_add_u64 v[48:49], v[48:49], 0, s[40:41]\n v_lshl_add_u64 v[50:51], v[50:51], 0, s[40:41]\n"
                   "v_lshl_add_u64 v[52:53], v[52:53], 0, s[40:41]\n v_lshl_add_u64 v[54:55], v[54:55], 0, s[40:41]\n")
// select: bfi (3 VGPR sources) against v_cndmask_b32 (VOP2, mask in vcc / in an SGPR pair)
KERNEL(k_bfi, "v_bfi_b32 v40, v60, v41, v42\n v_bfi_b32 v42, v61, v43, v44\n v_bfi_b32 v44, v62, v45, v46\n v_bfi_b32 v46, v63, v47, v48\n"
              "v_bfi_b32 v48, v60, v49, v50\n v_bfi_b32 v50, v61, v51, v52\n v_bfi_b32 v52, v62, v53, v54\n v_bfi_b32 v54, v63, v55, v40\n")
KERNEL(k_cndmask_vcc, "v_cndmask_b32 v40, v41, v42, vcc\n v_cndmask_b32 v42, v43, v44, vcc\n v_cndmask_b32 v44, v45, v46, vcc\n v_cndmask_b32 v46, v47, v48, vcc\n"
                      "v_cndmask_b32 v48, v49, v50, vcc\n v_cndmask_b32 v50, v51, v52, vcc\n v_cndmask_b32 v52, v53, v54, vcc\n v_cndmask_b32 v54, v55, v40, vcc\n")
KERNEL(k_cndmask_sgpr, "v_cndmask_b32 v40, v41, v42, s[42:43]\n v_cndmask_b32 v42, v43, v44, s[42:43]\n v_cndmask_b32 v44, v45, v46, s[42:43]\n v_cndmask_b32 v46, v47, v48, s[42:43]\n"
                       "v_cndmask_b32 v48, v49, v50, s[42:43]\n v_cndmask_b32 v50, v51, v52, s[42:43]\n v_cndmask_b32 v52, v53, v54, s[42:43]\n v_cndmask_b32 v54, v55, v40, s[42:43]\n")
KERNEL(k_cmp_i32, "v_cmp_gt_i32 vcc, 0, v40\n v_cmp_gt_i32 vcc, 0, v42\n v_cmp_gt_i32 vcc, 0, v44\n v_cmp_gt_i32 vcc, 0, v46\n"
                  "v_cmp_gt_i32 vcc, 0, v48\n v_cmp_gt_i32 vcc, 0, v50\n v_cmp_gt_i32 vcc, 0, v52\n v_cmp_gt_i32 vcc, 0, v54\n")
KERNEL(k_cmp_u64, "v_cmp_lt_u64 vcc, v[40:41], v[42:43]\n v_cmp_lt_u64 vcc, v[42:43], v[44:45]\n v_cmp_lt_u64 vcc, v[44:45], v[46:47]\n v_cmp_lt_u64 vcc, v[46:47], v[48:49]\n"
                  "v_cmp_lt_u64 vcc, v[48:49], v[50:51]\n v_cmp_lt_u64 vcc, v[50:51], v[52:53]\n v_cmp_lt_u64 vcc, v[52:53], v[54:55]\n v_cmp_lt_u64 vcc, v[54:55], v[40:41]\n")
KERNEL(k_and, "v_and_b32 v40, v60, v41\n v_and_b32 v42, v61, v43\n v_and_b32 v44, v62, v45\n v_and_b32 v46, v63, v47\n"
              "v_and_b32 v48, v60, v49\n v_and_b32 v50, v61, v51\n v_and_b32 v52, v62, v53\n v_and_b32 v54, v63, v55\n")
KERNEL(k_mov, "v_mov_b32 v40, v41\n v_mov_b32 v42, v43\n v_mov_b32 v44, v45\n v_mov_b32 v46, v47\n v_mov_b32 v48, v49\n v_mov_b32 v50, v51\n v_mov_b32 v52, v53\n v_mov_b32 v54, v55\n")
KERNEL(k_mul_lo_vv, "v_mul_lo_u32 v40, v62, v63\n v_mul_lo_u32 v42, v60, v61\n v_mul_lo_u32 v44, v62, v63\n v_mul_lo_u32 v46, v60, v61\n"
                    "v_mul_lo_u32 v48, v62, v63\n v_mul_lo_u32 v50, v60, v61\n v_mul_lo_u32 v52, v62, v63\n v_mul_lo_u32 v54, v60, v61\n")
KERNEL(k_add3, "v_add3_u32 v40, v41, v42, v43\n v_add3_u32 v42, v43, v44, v45\n v_add3_u32 v44, v45, v46, v47\n v_add3_u32 v46, v47, v48, v49\n"
               "v_add3_u32 v48, v49, v50, v51\n v_add3_u32 v50, v51, v52, v53\n v_add3_u32 v52, v53, v54, v55\n v_add3_u32 v54, v55, v40, v41\n")
// mixed streams: does a 2-cycle instruction pair with a 4-cycle one?  4 mads + 4 v_not
KERNEL(k_mix_mad_not, "v_mad_u64_u32 v[40:41], vcc, v62, v63, v[40:41]\n v_not_b32 v64, v65\n v_mad_u64_u32 v[42:43], vcc, v60, v61, v[42:43]\n v_not_b32 v66, v67\n"
                      "v_mad_u64_u32 v[44:45], vcc, v62, v63, v[44:45]\n v_not_b32 v68, v69\n v_mad_u64_u32 v[46:47], vcc, v60, v61, v[46:47]\n v_not_b32 v64, v67\n")

template <class K>
static void run(const char *name, K kern, u64 *d, int blocks_per_cu)
{
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double winstr = 5.0 * blocks_per_cu * (double)ITER * 16 * 8;
    printf("%-34s %8.3f ms   %5.2f cycles per wave64 instruction per SIMD (at 2.4 GHz nominal)\n", name, ms / 5, ms * 1e-3 * 2.4e9 / winstr);
}

int main(int argc, char **argv)
{
    const int wps = argc > 1 ? atoi(argv[1]) : 4;
    printf("%d waves per SIMD\n", wps);
    u64 *d;
    hipMalloc((void **)&d, 256 * 8 * 256 * 8);
    run("mad src0 == src1", k_same, d, wps);
    run("mad 4 banks", k_free, d, wps);
    run("mad sources in acc banks", k_conf2, d, wps);
    run("mad all in one bank", k_conf_all, d, wps);
    run("mad sources share a bank", k_src_same_bank, d, wps);
    run("mad sgpr src1", k_sgpr, d, wps);
    run("mad sgpr src1, src0 in acc bank", k_sgpr_conf, d, wps);
    run("mad +0, 2 banks", k_zero_free, d, wps);
    run("mad +0, sources share a bank", k_zero_conf, d, wps);
    run("mad +0, sgpr src1", k_zero_sgpr, d, wps);
    run("mad dst != src2", k_dst_other, d, wps);
    run("lshl_add_u64 v + v", k_add64_vv, d, wps);
    run("lshl_add_u64 v + s", k_add64_vs, d, wps);
    run("v_bfi_b32", k_bfi, d, wps);
    run("v_cndmask_b32 vcc", k_cndmask_vcc, d, wps);
    run("v_cndmask_b32 sgpr pair", k_cndmask_sgpr, d, wps);
    run("v_cmp_gt_i32", k_cmp_i32, d, wps);
    run("v_cmp_lt_u64", k_cmp_u64, d, wps);
    run("v_and_b32", k_and, d, wps);
    run("v_mov_b32", k_mov, d, wps);
    run("v_mul_lo_u32 v, v", k_mul_lo_vv, d, wps);
    run("v_add3_u32", k_add3, d, wps);
    run("mad + v_not alternating", k_mix_mad_not, d, wps);
    return 0;
}
