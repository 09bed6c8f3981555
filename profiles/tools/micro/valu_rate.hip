// What does one wave64 instruction of a given opcode cost a gfx950 SIMD when the SIMD is full of waves?
// Every workgroup is 512 threads = 8 waves, 2 per SIMD; the grid puts 4 of them on every CU (8 waves per SIMD) or 1
// (2 waves per SIMD).  Each wave runs a loop of 32 x 8 INDEPENDENT instructions of one kind (eight accumulators); lane 0 of
// every wave stamps s_memtime around the loop.  Reported: ns per instruction and SIMD from the kernel's wall time (HIP
// events) -- the figure that prices a kernel's opcode histogram -- and s_memtime ticks per instruction and SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate [filter]
// Round 4: every opcode the encode kernels use more than a handful of times (profiles/r04/opcodes_*.txt), the packed 16-bit
// (VOP3P) and three-operand forms that could replace them, the scalar unit, LDS, and VALU + SALU side by side.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

// eight copies of one instruction on the eight accumulators %0..%7; %8, %9 = two 64-bit accumulators; %10, %11 = loop-invariant vector operands
#define A2(op)   op " %0, %0, %10\n" op " %1, %1, %10\n" op " %2, %2, %10\n" op " %3, %3, %10\n" op " %4, %4, %10\n" op " %5, %5, %10\n" op " %6, %6, %10\n" op " %7, %7, %10\n"
#define A2R(op)  op " %0, %10, %0\n" op " %1, %10, %1\n" op " %2, %10, %2\n" op " %3, %10, %3\n" op " %4, %10, %4\n" op " %5, %10, %5\n" op " %6, %10, %6\n" op " %7, %10, %7\n"
#define A1(op)   op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define A3(op)   op " %0, %0, %10, %11\n" op " %1, %1, %10, %11\n" op " %2, %2, %10, %11\n" op " %3, %3, %10, %11\n" op " %4, %4, %10, %11\n" op " %5, %5, %10, %11\n" op " %6, %6, %10, %11\n" op " %7, %7, %10, %11\n"
#define A3X(op, x) op " %0, %0, %10, %11 " x "\n" op " %1, %1, %10, %11 " x "\n" op " %2, %2, %10, %11 " x "\n" op " %3, %3, %10, %11 " x "\n" op " %4, %4, %10, %11 " x "\n" op " %5, %5, %10, %11 " x "\n" op " %6, %6, %10, %11 " x "\n" op " %7, %7, %10, %11 " x "\n"
#define A2X(op, x) op " %0, %0, %10 " x "\n" op " %1, %1, %10 " x "\n" op " %2, %2, %10 " x "\n" op " %3, %3, %10 " x "\n" op " %4, %4, %10 " x "\n" op " %5, %5, %10 " x "\n" op " %6, %6, %10 " x "\n" op " %7, %7, %10 " x "\n"
#define ACC "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define RUNV(tmpl) REP32(asm volatile(tmpl : ACC, "+v"(q0), "+v"(q1) : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc");)

// clang-format off
#define KINDS(X) \
  X( 0, "v_add_u32",            A2("v_add_u32")) \
  X( 1, "v_sub_u32",            A2("v_sub_u32")) \
  X( 2, "v_and_b32",            A2("v_and_b32")) \
  X( 3, "v_or_b32",             A2("v_or_b32")) \
  X( 4, "v_xor_b32",            A2("v_xor_b32")) \
  X( 5, "v_mov_b32",            "v_mov_b32 %0, %10\nv_mov_b32 %1, %10\nv_mov_b32 %2, %10\nv_mov_b32 %3, %10\nv_mov_b32 %4, %10\nv_mov_b32 %5, %10\nv_mov_b32 %6, %10\nv_mov_b32 %7, %10\n") \
  X( 6, "v_not_b32",            A1("v_not_b32")) \
  X( 7, "v_lshlrev_b32",        A2R("v_lshlrev_b32")) \
  X( 8, "v_lshrrev_b32",        A2R("v_lshrrev_b32")) \
  X( 9, "v_ashrrev_i32",        A2R("v_ashrrev_i32")) \
  X(10, "v_min_u32",            A2("v_min_u32")) \
  X(11, "v_max_i32",            A2("v_max_i32")) \
  X(12, "v_cndmask_b32 (vcc)",  "v_cndmask_b32 %0, %0, %10, vcc\nv_cndmask_b32 %1, %1, %10, vcc\nv_cndmask_b32 %2, %2, %10, vcc\nv_cndmask_b32 %3, %3, %10, vcc\nv_cndmask_b32 %4, %4, %10, vcc\nv_cndmask_b32 %5, %5, %10, vcc\nv_cndmask_b32 %6, %6, %10, vcc\nv_cndmask_b32 %7, %7, %10, vcc\n") \
  X(13, "v_cndmask_b32_e64 (sgpr pair)", "v_cndmask_b32 %0, %0, %10, s[20:21]\nv_cndmask_b32 %1, %1, %10, s[20:21]\nv_cndmask_b32 %2, %2, %10, s[20:21]\nv_cndmask_b32 %3, %3, %10, s[20:21]\nv_cndmask_b32 %4, %4, %10, s[20:21]\nv_cndmask_b32 %5, %5, %10, s[20:21]\nv_cndmask_b32 %6, %6, %10, s[20:21]\nv_cndmask_b32 %7, %7, %10, s[20:21]\n") \
  X(14, "v_cmp_lt_u32 -> vcc",  "v_cmp_lt_u32 vcc, %0, %10\nv_cmp_lt_u32 vcc, %1, %10\nv_cmp_lt_u32 vcc, %2, %10\nv_cmp_lt_u32 vcc, %3, %10\nv_cmp_lt_u32 vcc, %4, %10\nv_cmp_lt_u32 vcc, %5, %10\nv_cmp_lt_u32 vcc, %6, %10\nv_cmp_lt_u32 vcc, %7, %10\n") \
  X(15, "v_cmp_lt_u32_e64 -> sgpr pair", "v_cmp_lt_u32 s[20:21], %0, %10\nv_cmp_lt_u32 s[22:23], %1, %10\nv_cmp_lt_u32 s[24:25], %2, %10\nv_cmp_lt_u32 s[26:27], %3, %10\nv_cmp_lt_u32 s[20:21], %4, %10\nv_cmp_lt_u32 s[22:23], %5, %10\nv_cmp_lt_u32 s[24:25], %6, %10\nv_cmp_lt_u32 s[26:27], %7, %10\n") \
  X(16, "v_bfe_u32",            "v_bfe_u32 %0, %0, 3, 9\nv_bfe_u32 %1, %1, 3, 9\nv_bfe_u32 %2, %2, 3, 9\nv_bfe_u32 %3, %3, 3, 9\nv_bfe_u32 %4, %4, 3, 9\nv_bfe_u32 %5, %5, 3, 9\nv_bfe_u32 %6, %6, 3, 9\nv_bfe_u32 %7, %7, 3, 9\n") \
  X(17, "v_bfi_b32",            A3("v_bfi_b32")) \
  X(18, "v_add3_u32",           A3("v_add3_u32")) \
  X(19, "v_lshl_add_u32",       "v_lshl_add_u32 %0, %0, 3, %10\nv_lshl_add_u32 %1, %1, 3, %10\nv_lshl_add_u32 %2, %2, 3, %10\nv_lshl_add_u32 %3, %3, 3, %10\nv_lshl_add_u32 %4, %4, 3, %10\nv_lshl_add_u32 %5, %5, 3, %10\nv_lshl_add_u32 %6, %6, 3, %10\nv_lshl_add_u32 %7, %7, 3, %10\n") \
  X(20, "v_lshl_or_b32",        "v_lshl_or_b32 %0, %0, 3, %10\nv_lshl_or_b32 %1, %1, 3, %10\nv_lshl_or_b32 %2, %2, 3, %10\nv_lshl_or_b32 %3, %3, 3, %10\nv_lshl_or_b32 %4, %4, 3, %10\nv_lshl_or_b32 %5, %5, 3, %10\nv_lshl_or_b32 %6, %6, 3, %10\nv_lshl_or_b32 %7, %7, 3, %10\n") \
  X(21, "v_and_or_b32",         A3("v_and_or_b32")) \
  X(22, "v_or3_b32",            A3("v_or3_b32")) \
  X(23, "v_xad_u32",            A3("v_xad_u32")) \
  X(24, "v_perm_b32",           A3("v_perm_b32")) \
  X(25, "v_alignbit_b32",       A3("v_alignbit_b32")) \
  X(26, "v_alignbyte_b32",      A3("v_alignbyte_b32")) \
  X(27, "v_bitop3_b32",         A3X("v_bitop3_b32", "bitop3:0xf4")) \
  X(28, "v_ffbh_u32",           A1("v_ffbh_u32")) \
  X(29, "v_bcnt_u32_b32",       A2("v_bcnt_u32_b32")) \
  X(30, "v_mbcnt_lo_u32_b32",   A2("v_mbcnt_lo_u32_b32")) \
  X(31, "v_pk_add_u16",         A2("v_pk_add_u16")) \
  X(32, "v_pk_sub_i16",         A2("v_pk_sub_i16")) \
  X(33, "v_pk_min_u16",         A2("v_pk_min_u16")) \
  X(34, "v_pk_max_i16",         A2("v_pk_max_i16")) \
  X(35, "v_pk_lshlrev_b16",     A2R("v_pk_lshlrev_b16")) \
  X(36, "v_pk_lshrrev_b16",     A2R("v_pk_lshrrev_b16")) \
  X(37, "v_pk_ashrrev_i16",     A2R("v_pk_ashrrev_i16")) \
  X(38, "v_pk_mul_lo_u16",      A2("v_pk_mul_lo_u16")) \
  X(39, "v_pk_mad_u16",         A3("v_pk_mad_u16")) \
  X(40, "v_sad_u8",             A3("v_sad_u8")) \
  X(41, "v_sad_u16",            A3("v_sad_u16")) \
  X(42, "v_mul_lo_u32",         A2("v_mul_lo_u32")) \
  X(43, "v_mul_u32_u24",        A2("v_mul_u32_u24")) \
  X(44, "v_mad_u32_u24",        A3("v_mad_u32_u24")) \
  X(45, "v_lshlrev_b64",        "v_lshlrev_b64 %8, 1, %8\nv_lshlrev_b64 %9, 1, %9\nv_lshlrev_b64 %8, 1, %8\nv_lshlrev_b64 %9, 1, %9\nv_lshlrev_b64 %8, 1, %8\nv_lshlrev_b64 %9, 1, %9\nv_lshlrev_b64 %8, 1, %8\nv_lshlrev_b64 %9, 1, %9\n") \
  X(46, "v_lshl_add_u64",       "v_lshl_add_u64 %8, %8, 1, %9\nv_lshl_add_u64 %9, %9, 1, %8\nv_lshl_add_u64 %8, %8, 1, %9\nv_lshl_add_u64 %9, %9, 1, %8\nv_lshl_add_u64 %8, %8, 1, %9\nv_lshl_add_u64 %9, %9, 1, %8\nv_lshl_add_u64 %8, %8, 1, %9\nv_lshl_add_u64 %9, %9, 1, %8\n") \
  X(47, "v_add_u32_sdwa",       A2X("v_add_u32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1")) \
  X(48, "v_min_u32_sdwa",       A2X("v_min_u32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1")) \
  X(49, "v_add_u32_dpp row_shr:1", "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n") \
  X(50, "v_mov_b32_dpp wave_shr:1", "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n") \
  X(51, "v_readlane_b32",       "v_readlane_b32 s20, %0, 3\nv_readlane_b32 s21, %1, 3\nv_readlane_b32 s22, %2, 3\nv_readlane_b32 s23, %3, 3\nv_readlane_b32 s24, %4, 3\nv_readlane_b32 s25, %5, 3\nv_readlane_b32 s26, %6, 3\nv_readlane_b32 s27, %7, 3\n") \
  X(52, "v_readfirstlane_b32",  "v_readfirstlane_b32 s20, %0\nv_readfirstlane_b32 s21, %1\nv_readfirstlane_b32 s22, %2\nv_readfirstlane_b32 s23, %3\nv_readfirstlane_b32 s24, %4\nv_readfirstlane_b32 s25, %5\nv_readfirstlane_b32 s26, %6\nv_readfirstlane_b32 s27, %7\n") \
  X(53, "s_add_u32",            "s_add_u32 s20, s20, s28\ns_add_u32 s21, s21, s28\ns_add_u32 s22, s22, s28\ns_add_u32 s23, s23, s28\ns_add_u32 s24, s24, s28\ns_add_u32 s25, s25, s28\ns_add_u32 s26, s26, s28\ns_add_u32 s27, s27, s28\n") \
  X(54, "s_and_b64",            "s_and_b64 s[20:21], s[20:21], s[28:29]\ns_and_b64 s[22:23], s[22:23], s[28:29]\ns_and_b64 s[24:25], s[24:25], s[28:29]\ns_and_b64 s[26:27], s[26:27], s[28:29]\ns_and_b64 s[20:21], s[20:21], s[28:29]\ns_and_b64 s[22:23], s[22:23], s[28:29]\ns_and_b64 s[24:25], s[24:25], s[28:29]\ns_and_b64 s[26:27], s[26:27], s[28:29]\n") \
  X(55, "v_min_u32 + s_add_u32 interleaved (per pair)", "v_min_u32 %0, %0, %10\ns_add_u32 s20, s20, s28\nv_min_u32 %1, %1, %10\ns_add_u32 s21, s21, s28\nv_min_u32 %2, %2, %10\ns_add_u32 s22, s22, s28\nv_min_u32 %3, %3, %10\ns_add_u32 s23, s23, s28\nv_min_u32 %4, %4, %10\ns_add_u32 s24, s24, s28\nv_min_u32 %5, %5, %10\ns_add_u32 s25, s25, s28\nv_min_u32 %6, %6, %10\ns_add_u32 s26, s26, s28\nv_min_u32 %7, %7, %10\ns_add_u32 s27, s27, s28\n") \
  X(56, "v_add_u32 + s_add_u32 interleaved (per pair)", "v_add_u32 %0, %0, %10\ns_add_u32 s20, s20, s28\nv_add_u32 %1, %1, %10\ns_add_u32 s21, s21, s28\nv_add_u32 %2, %2, %10\ns_add_u32 s22, s22, s28\nv_add_u32 %3, %3, %10\ns_add_u32 s23, s23, s28\nv_add_u32 %4, %4, %10\ns_add_u32 s24, s24, s28\nv_add_u32 %5, %5, %10\ns_add_u32 s25, s25, s28\nv_add_u32 %6, %6, %10\ns_add_u32 s26, s26, s28\nv_add_u32 %7, %7, %10\ns_add_u32 s27, s27, s28\n") \
  X(57, "ds_read_b32 (8 in flight)", "ds_read_b32 %0, %11\nds_read_b32 %1, %11 offset:256\nds_read_b32 %2, %11 offset:512\nds_read_b32 %3, %11 offset:768\nds_read_b32 %4, %11 offset:1024\nds_read_b32 %5, %11 offset:1280\nds_read_b32 %6, %11 offset:1536\nds_read_b32 %7, %11 offset:1792\ns_waitcnt lgkmcnt(0)\n") \
  X(58, "ds_write_b32 (8 in flight)", "ds_write_b32 %11, %0\nds_write_b32 %11, %1 offset:256\nds_write_b32 %11, %2 offset:512\nds_write_b32 %11, %3 offset:768\nds_write_b32 %11, %4 offset:1024\nds_write_b32 %11, %5 offset:1280\nds_write_b32 %11, %6 offset:1536\nds_write_b32 %11, %7 offset:1792\ns_waitcnt lgkmcnt(0)\n") \
  X(59, "ds_write_b8 (8 in flight)", "ds_write_b8 %11, %0\nds_write_b8 %11, %1 offset:256\nds_write_b8 %11, %2 offset:512\nds_write_b8 %11, %3 offset:768\nds_write_b8 %11, %4 offset:1024\nds_write_b8 %11, %5 offset:1280\nds_write_b8 %11, %6 offset:1536\nds_write_b8 %11, %7 offset:1792\ns_waitcnt lgkmcnt(0)\n") \
  X(60, "ds_read_u8 (8 in flight)", "ds_read_u8 %0, %11\nds_read_u8 %1, %11 offset:256\nds_read_u8 %2, %11 offset:512\nds_read_u8 %3, %11 offset:768\nds_read_u8 %4, %11 offset:1024\nds_read_u8 %5, %11 offset:1280\nds_read_u8 %6, %11 offset:1536\nds_read_u8 %7, %11 offset:1792\ns_waitcnt lgkmcnt(0)\n") \
  X(61, "v_cmp + s_and_saveexec + s_or exec (per triple)", "v_cmp_lt_u32 vcc, %0, %10\ns_and_saveexec_b64 s[20:21], vcc\ns_or_b64 exec, exec, s[20:21]\nv_cmp_lt_u32 vcc, %1, %10\ns_and_saveexec_b64 s[20:21], vcc\ns_or_b64 exec, exec, s[20:21]\nv_cmp_lt_u32 vcc, %2, %10\ns_and_saveexec_b64 s[20:21], vcc\ns_or_b64 exec, exec, s[20:21]\nv_cmp_lt_u32 vcc, %3, %10\ns_and_saveexec_b64 s[20:21], vcc\ns_or_b64 exec, exec, s[20:21]\nv_cmp_lt_u32 vcc, %4, %10\ns_and_saveexec_b64 s[20:21], vcc\ns_or_b64 exec, exec, s[20:21]\nv_cmp_lt_u32 vcc, %5, %10\ns_and_saveexec_b64 s[20:21], vcc\ns_or_b64 exec, exec, s[20:21]\nv_cmp_lt_u32 vcc, %6, %10\ns_and_saveexec_b64 s[20:21], vcc\ns_or_b64 exec, exec, s[20:21]\nv_cmp_lt_u32 vcc, %7, %10\ns_and_saveexec_b64 s[20:21], vcc\ns_or_b64 exec, exec, s[20:21]\n") \
  X(62, "v_lshrrev_b64",        "v_lshrrev_b64 %8, 1, %8\nv_lshrrev_b64 %9, 1, %9\nv_lshrrev_b64 %8, 1, %8\nv_lshrrev_b64 %9, 1, %9\nv_lshrrev_b64 %8, 1, %8\nv_lshrrev_b64 %9, 1, %9\nv_lshrrev_b64 %8, 1, %8\nv_lshrrev_b64 %9, 1, %9\n") \
  X(63, "v_mad_u64_u32",        "v_mad_u64_u32 %8, vcc, %2, %3, %8\nv_mad_u64_u32 %9, vcc, %2, %3, %9\nv_mad_u64_u32 %8, vcc, %2, %3, %8\nv_mad_u64_u32 %9, vcc, %2, %3, %9\nv_mad_u64_u32 %8, vcc, %2, %3, %8\nv_mad_u64_u32 %9, vcc, %2, %3, %9\nv_mad_u64_u32 %8, vcc, %2, %3, %8\nv_mad_u64_u32 %9, vcc, %2, %3, %9\n") \
  X(64, "v_max3_u32",           A3("v_max3_u32")) \
  X(65, "v_min3_u32",           A3("v_min3_u32")) \
  X(66, "v_med3_i32",           A3("v_med3_i32")) \
  X(67, "v_sub_u32_sdwa",       A2X("v_sub_u32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1")) \
  X(68, "v_pk_sub_u16 clamp",   A2X("v_pk_sub_u16", "clamp")) \
  X(69, "v_lshlrev_b16",        A2R("v_lshlrev_b16")) \
  X(70, "v_cmp_lt_u32 vcc + v_cndmask_b32_e32 vcc (per pair)", "v_cmp_lt_u32 vcc, %0, %10\nv_cndmask_b32 %0, %0, %10, vcc\nv_cmp_lt_u32 vcc, %1, %10\nv_cndmask_b32 %1, %1, %10, vcc\nv_cmp_lt_u32 vcc, %2, %10\nv_cndmask_b32 %2, %2, %10, vcc\nv_cmp_lt_u32 vcc, %3, %10\nv_cndmask_b32 %3, %3, %10, vcc\nv_cmp_lt_u32 vcc, %4, %10\nv_cndmask_b32 %4, %4, %10, vcc\nv_cmp_lt_u32 vcc, %5, %10\nv_cndmask_b32 %5, %5, %10, vcc\nv_cmp_lt_u32 vcc, %6, %10\nv_cndmask_b32 %6, %6, %10, vcc\nv_cmp_lt_u32 vcc, %7, %10\nv_cndmask_b32 %7, %7, %10, vcc\n") \
  X(71, "v_cndmask_b32_e64 vcc", "v_cndmask_b32_e64 %0, %0, %10, vcc\nv_cndmask_b32_e64 %1, %1, %10, vcc\nv_cndmask_b32_e64 %2, %2, %10, vcc\nv_cndmask_b32_e64 %3, %3, %10, vcc\nv_cndmask_b32_e64 %4, %4, %10, vcc\nv_cndmask_b32_e64 %5, %5, %10, vcc\nv_cndmask_b32_e64 %6, %6, %10, vcc\nv_cndmask_b32_e64 %7, %7, %10, vcc\n") \
  X(72, "s_mov vcc + 8 v_cndmask_b32_e32 vcc (per cndmask)", "s_mov_b64 vcc, s[20:21]\nv_cndmask_b32 %0, %0, %10, vcc\nv_cndmask_b32 %1, %1, %10, vcc\nv_cndmask_b32 %2, %2, %10, vcc\nv_cndmask_b32 %3, %3, %10, vcc\nv_cndmask_b32 %4, %4, %10, vcc\nv_cndmask_b32 %5, %5, %10, vcc\nv_cndmask_b32 %6, %6, %10, vcc\nv_cndmask_b32 %7, %7, %10, vcc\n") \
  X(73, "v_lshlrev_b32 by constant", "v_lshlrev_b32 %0, 3, %0\nv_lshlrev_b32 %1, 3, %1\nv_lshlrev_b32 %2, 3, %2\nv_lshlrev_b32 %3, 3, %3\nv_lshlrev_b32 %4, 3, %4\nv_lshlrev_b32 %5, 3, %5\nv_lshlrev_b32 %6, 3, %6\nv_lshlrev_b32 %7, 3, %7\n") \
  X(74, "v_and_b32 with literal", "v_and_b32 %0, 0x00ff00ff, %0\nv_and_b32 %1, 0x00ff00ff, %1\nv_and_b32 %2, 0x00ff00ff, %2\nv_and_b32 %3, 0x00ff00ff, %3\nv_and_b32 %4, 0x00ff00ff, %4\nv_and_b32 %5, 0x00ff00ff, %5\nv_and_b32 %6, 0x00ff00ff, %6\nv_and_b32 %7, 0x00ff00ff, %7\n") \
  X(75, "v_addc_co_u32", "v_addc_co_u32 %0, vcc, %0, %10, vcc\nv_addc_co_u32 %1, vcc, %1, %10, vcc\nv_addc_co_u32 %2, vcc, %2, %10, vcc\nv_addc_co_u32 %3, vcc, %3, %10, vcc\nv_addc_co_u32 %4, vcc, %4, %10, vcc\nv_addc_co_u32 %5, vcc, %5, %10, vcc\nv_addc_co_u32 %6, vcc, %6, %10, vcc\nv_addc_co_u32 %7, vcc, %7, %10, vcc\n") \
  X(76, "ds_or_b32 (8 in flight)", "ds_or_b32 %11, %0 offset:0\nds_or_b32 %11, %1 offset:256\nds_or_b32 %11, %2 offset:512\nds_or_b32 %11, %3 offset:768\nds_or_b32 %11, %4 offset:1024\nds_or_b32 %11, %5 offset:1280\nds_or_b32 %11, %6 offset:1536\nds_or_b32 %11, %7 offset:1792\ns_waitcnt lgkmcnt(0)\n") \
  X(77, "ds_write_b64 (8 in flight)", "ds_write_b64 %11, %8 offset:0\nds_write_b64 %11, %9 offset:512\nds_write_b64 %11, %8 offset:1024\nds_write_b64 %11, %9 offset:1536\nds_write_b64 %11, %8 offset:2048\nds_write_b64 %11, %9 offset:2560\nds_write_b64 %11, %8 offset:3072\nds_write_b64 %11, %9 offset:3584\ns_waitcnt lgkmcnt(0)\n") \
  X(78, "ds_read_b64 (8 in flight)", "ds_read_b64 %8, %11 offset:0\nds_read_b64 %9, %11 offset:512\nds_read_b64 %8, %11 offset:1024\nds_read_b64 %9, %11 offset:1536\nds_read_b64 %8, %11 offset:2048\nds_read_b64 %9, %11 offset:2560\nds_read_b64 %8, %11 offset:3072\nds_read_b64 %9, %11 offset:3584\ns_waitcnt lgkmcnt(0)\n") \
  X(79, "ds_add_rtn_u32 (8 in flight)", "ds_add_rtn_u32 %0, %11, %10 offset:0\nds_add_rtn_u32 %1, %11, %10 offset:256\nds_add_rtn_u32 %2, %11, %10 offset:512\nds_add_rtn_u32 %3, %11, %10 offset:768\nds_add_rtn_u32 %4, %11, %10 offset:1024\nds_add_rtn_u32 %5, %11, %10 offset:1280\nds_add_rtn_u32 %6, %11, %10 offset:1536\nds_add_rtn_u32 %7, %11, %10 offset:1792\ns_waitcnt lgkmcnt(0)\n") \
  X(80, "ds_bpermute_b32 (8 in flight)", "ds_bpermute_b32 %0, %11, %0\nds_bpermute_b32 %1, %11, %1\nds_bpermute_b32 %2, %11, %2\nds_bpermute_b32 %3, %11, %3\nds_bpermute_b32 %4, %11, %4\nds_bpermute_b32 %5, %11, %5\nds_bpermute_b32 %6, %11, %6\nds_bpermute_b32 %7, %11, %7\ns_waitcnt lgkmcnt(0)\n") \
  X(81, "v_min_u16", "v_min_u16 %0, %0, %10\nv_min_u16 %1, %1, %10\nv_min_u16 %2, %2, %10\nv_min_u16 %3, %3, %10\nv_min_u16 %4, %4, %10\nv_min_u16 %5, %5, %10\nv_min_u16 %6, %6, %10\nv_min_u16 %7, %7, %10\n") \
  X(82, "v_max_u16", "v_max_u16 %0, %0, %10\nv_max_u16 %1, %1, %10\nv_max_u16 %2, %2, %10\nv_max_u16 %3, %3, %10\nv_max_u16 %4, %4, %10\nv_max_u16 %5, %5, %10\nv_max_u16 %6, %6, %10\nv_max_u16 %7, %7, %10\n") \
  X(83, "v_add_u16", "v_add_u16 %0, %0, %10\nv_add_u16 %1, %1, %10\nv_add_u16 %2, %2, %10\nv_add_u16 %3, %3, %10\nv_add_u16 %4, %4, %10\nv_add_u16 %5, %5, %10\nv_add_u16 %6, %6, %10\nv_add_u16 %7, %7, %10\n") \
  X(84, "v_sub_u16", "v_sub_u16 %0, %0, %10\nv_sub_u16 %1, %1, %10\nv_sub_u16 %2, %2, %10\nv_sub_u16 %3, %3, %10\nv_sub_u16 %4, %4, %10\nv_sub_u16 %5, %5, %10\nv_sub_u16 %6, %6, %10\nv_sub_u16 %7, %7, %10\n") \
  X(85, "v_lshrrev_b16", "v_lshrrev_b16 %0, %10, %0\nv_lshrrev_b16 %1, %10, %1\nv_lshrrev_b16 %2, %10, %2\nv_lshrrev_b16 %3, %10, %3\nv_lshrrev_b16 %4, %10, %4\nv_lshrrev_b16 %5, %10, %5\nv_lshrrev_b16 %6, %10, %6\nv_lshrrev_b16 %7, %10, %7\n") \
  X(86, "v_ashrrev_i16", "v_ashrrev_i16 %0, %10, %0\nv_ashrrev_i16 %1, %10, %1\nv_ashrrev_i16 %2, %10, %2\nv_ashrrev_i16 %3, %10, %3\nv_ashrrev_i16 %4, %10, %4\nv_ashrrev_i16 %5, %10, %5\nv_ashrrev_i16 %6, %10, %6\nv_ashrrev_i16 %7, %10, %7\n") \
  X(87, "v_mul_lo_u16", "v_mul_lo_u16 %0, %0, %10\nv_mul_lo_u16 %1, %1, %10\nv_mul_lo_u16 %2, %2, %10\nv_mul_lo_u16 %3, %3, %10\nv_mul_lo_u16 %4, %4, %10\nv_mul_lo_u16 %5, %5, %10\nv_mul_lo_u16 %6, %6, %10\nv_mul_lo_u16 %7, %7, %10\n") \
  X(88, "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1 %0, %0\nv_cvt_f32_ubyte1 %1, %1\nv_cvt_f32_ubyte1 %2, %2\nv_cvt_f32_ubyte1 %3, %3\nv_cvt_f32_ubyte1 %4, %4\nv_cvt_f32_ubyte1 %5, %5\nv_cvt_f32_ubyte1 %6, %6\nv_cvt_f32_ubyte1 %7, %7\n") \
  X(89, "v_frexp_exp_i32_f32", "v_frexp_exp_i32_f32 %0, %0\nv_frexp_exp_i32_f32 %1, %1\nv_frexp_exp_i32_f32 %2, %2\nv_frexp_exp_i32_f32 %3, %3\nv_frexp_exp_i32_f32 %4, %4\nv_frexp_exp_i32_f32 %5, %5\nv_frexp_exp_i32_f32 %6, %6\nv_frexp_exp_i32_f32 %7, %7\n") \
  X(90, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0\nv_cvt_f32_u32 %1, %1\nv_cvt_f32_u32 %2, %2\nv_cvt_f32_u32 %3, %3\nv_cvt_f32_u32 %4, %4\nv_cvt_f32_u32 %5, %5\nv_cvt_f32_u32 %6, %6\nv_cvt_f32_u32 %7, %7\n") \
  X(91, "v_lshrrev_b32_e64 (sgpr value)", "v_lshrrev_b32_e64 %0, %0, s20\nv_lshrrev_b32_e64 %1, %1, s20\nv_lshrrev_b32_e64 %2, %2, s20\nv_lshrrev_b32_e64 %3, %3, s20\nv_lshrrev_b32_e64 %4, %4, s20\nv_lshrrev_b32_e64 %5, %5, s20\nv_lshrrev_b32_e64 %6, %6, s20\nv_lshrrev_b32_e64 %7, %7, s20\n") \
  X(92, "v_add_u32_e64 (two sgprs... one sgpr)", "v_add_u32_e64 %0, %0, s20\nv_add_u32_e64 %1, %1, s20\nv_add_u32_e64 %2, %2, s20\nv_add_u32_e64 %3, %3, s20\nv_add_u32_e64 %4, %4, s20\nv_add_u32_e64 %5, %5, s20\nv_add_u32_e64 %6, %6, s20\nv_add_u32_e64 %7, %7, s20\n") \
  X(93, "v_cmp_lt_u16 -> vcc", "v_cmp_lt_u16 vcc, %0, %10\nv_cmp_lt_u16 vcc, %1, %10\nv_cmp_lt_u16 vcc, %2, %10\nv_cmp_lt_u16 vcc, %3, %10\nv_cmp_lt_u16 vcc, %4, %10\nv_cmp_lt_u16 vcc, %5, %10\nv_cmp_lt_u16 vcc, %6, %10\nv_cmp_lt_u16 vcc, %7, %10\n") \
  X(94, "v_max_u32", "v_max_u32 %0, %0, %10\nv_max_u32 %1, %1, %10\nv_max_u32 %2, %2, %10\nv_max_u32 %3, %3, %10\nv_max_u32 %4, %4, %10\nv_max_u32 %5, %5, %10\nv_max_u32 %6, %6, %10\nv_max_u32 %7, %7, %10\n") \
  X(95, "v_min_i16", "v_min_i16 %0, %0, %10\nv_min_i16 %1, %1, %10\nv_min_i16 %2, %2, %10\nv_min_i16 %3, %3, %10\nv_min_i16 %4, %4, %10\nv_min_i16 %5, %5, %10\nv_min_i16 %6, %6, %10\nv_min_i16 %7, %7, %10\n") \
  X(96, "v_subrev_u32", "v_subrev_u32 %0, %0, %10\nv_subrev_u32 %1, %1, %10\nv_subrev_u32 %2, %2, %10\nv_subrev_u32 %3, %3, %10\nv_subrev_u32 %4, %4, %10\nv_subrev_u32 %5, %5, %10\nv_subrev_u32 %6, %6, %10\nv_subrev_u32 %7, %7, %10\n") \
  X(97, "v_xnor_b32", "v_xnor_b32 %0, %0, %10\nv_xnor_b32 %1, %1, %10\nv_xnor_b32 %2, %2, %10\nv_xnor_b32 %3, %3, %10\nv_xnor_b32 %4, %4, %10\nv_xnor_b32 %5, %5, %10\nv_xnor_b32 %6, %6, %10\nv_xnor_b32 %7, %7, %10\n") \
  X(98, "v_add_co_u32", "v_add_co_u32 %0, vcc, %0, %10\nv_add_co_u32 %1, vcc, %1, %10\nv_add_co_u32 %2, vcc, %2, %10\nv_add_co_u32 %3, vcc, %3, %10\nv_add_co_u32 %4, vcc, %4, %10\nv_add_co_u32 %5, vcc, %5, %10\nv_add_co_u32 %6, vcc, %6, %10\nv_add_co_u32 %7, vcc, %7, %10\n") \
  X(99, "v_lshlrev_b32 + v_add_u32 interleaved (per pair)", "v_lshlrev_b32 %0, %10, %0\nv_add_u32 %0, %0, %10\nv_lshlrev_b32 %1, %10, %1\nv_add_u32 %1, %1, %10\nv_lshlrev_b32 %2, %10, %2\nv_add_u32 %2, %2, %10\nv_lshlrev_b32 %3, %10, %3\nv_add_u32 %3, %3, %10\nv_lshlrev_b32 %4, %10, %4\nv_add_u32 %4, %4, %10\nv_lshlrev_b32 %5, %10, %5\nv_add_u32 %5, %5, %10\nv_lshlrev_b32 %6, %10, %6\nv_add_u32 %6, %6, %10\nv_lshlrev_b32 %7, %10, %7\nv_add_u32 %7, %7, %10\n") \
  X(100, "mix c e c e c e c e (add/min, independent; per 8)", "v_add_u32 %0, %0, %10\nv_min_u32 %1, %1, %10\nv_add_u32 %2, %2, %10\nv_min_u32 %3, %3, %10\nv_add_u32 %4, %4, %10\nv_min_u32 %5, %5, %10\nv_add_u32 %6, %6, %10\nv_min_u32 %7, %7, %10\n") \
  X(101, "mix c c e e c c e e (per 8)", "v_add_u32 %0, %0, %10\nv_add_u32 %1, %1, %10\nv_min_u32 %2, %2, %10\nv_min_u32 %3, %3, %10\nv_add_u32 %4, %4, %10\nv_add_u32 %5, %5, %10\nv_min_u32 %6, %6, %10\nv_min_u32 %7, %7, %10\n") \
  X(102, "mix c c c c e e e e (per 8)", "v_add_u32 %0, %0, %10\nv_add_u32 %1, %1, %10\nv_add_u32 %2, %2, %10\nv_add_u32 %3, %3, %10\nv_min_u32 %4, %4, %10\nv_min_u32 %5, %5, %10\nv_min_u32 %6, %6, %10\nv_min_u32 %7, %7, %10\n") \
  X(103, "mix c c c e c c c e (per 8)", "v_add_u32 %0, %0, %10\nv_add_u32 %1, %1, %10\nv_add_u32 %2, %2, %10\nv_min_u32 %3, %3, %10\nv_add_u32 %4, %4, %10\nv_add_u32 %5, %5, %10\nv_add_u32 %6, %6, %10\nv_min_u32 %7, %7, %10\n") \
  X(104, "mix c c c c c c c e (per 8)", "v_add_u32 %0, %0, %10\nv_add_u32 %1, %1, %10\nv_add_u32 %2, %2, %10\nv_add_u32 %3, %3, %10\nv_add_u32 %4, %4, %10\nv_add_u32 %5, %5, %10\nv_add_u32 %6, %6, %10\nv_min_u32 %7, %7, %10\n") \
  X(105, "mix add/lshl c L c L c L c L (per 8)", "v_add_u32 %0, %0, %10\nv_lshlrev_b32 %1, %10, %1\nv_add_u32 %2, %2, %10\nv_lshlrev_b32 %3, %10, %3\nv_add_u32 %4, %4, %10\nv_lshlrev_b32 %5, %10, %5\nv_add_u32 %6, %6, %10\nv_lshlrev_b32 %7, %10, %7\n") \
  X(106, "mix add/bitop3 c b c b c b c b (per 8)", "v_add_u32 %0, %0, %10\nv_bitop3_b32 %1, %1, %10, %11 bitop3:0xe4\nv_add_u32 %2, %2, %10\nv_bitop3_b32 %3, %3, %10, %11 bitop3:0xe4\nv_add_u32 %4, %4, %10\nv_bitop3_b32 %5, %5, %10, %11 bitop3:0xe4\nv_add_u32 %6, %6, %10\nv_bitop3_b32 %7, %7, %10, %11 bitop3:0xe4\n") \
  X(107, "mix add/min_u16 alternating (per 8)", "v_add_u32 %0, %0, %10\nv_min_u16 %1, %1, %10\nv_add_u32 %2, %2, %10\nv_min_u16 %3, %3, %10\nv_add_u32 %4, %4, %10\nv_min_u16 %5, %5, %10\nv_add_u32 %6, %6, %10\nv_min_u16 %7, %7, %10\n") \
  X(108, "mix e c c c c c c c x8 dependent chain a0 only: add add add min (per 8)", "v_add_u32 %0, %0, %10\nv_add_u32 %0, %0, %10\nv_add_u32 %0, %0, %10\nv_min_u32 %0, %0, %10\nv_add_u32 %0, %0, %10\nv_add_u32 %0, %0, %10\nv_add_u32 %0, %0, %10\nv_min_u32 %0, %0, %10\n")
// clang-format on

constexpr int NKINDS = 109;

template <int KIND>
__global__ __launch_bounds__(512) void k(unsigned long long *out, unsigned *sink, int iters) {
    __shared__ unsigned lds[4096];
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long q0 = a0, q1 = a1;
    const unsigned b = (blockIdx.x & 7u) | 1u;
    const unsigned c = (threadIdx.x & 63u) * 4u;  // LDS address of the ds_* kinds (conflict-free), third operand of the others
    lds[threadIdx.x] = a0;
    asm volatile("s_mov_b64 s[28:29], -1\ns_mov_b64 s[20:21], 0x55\ns_mov_b64 s[22:23], 0x33\ns_mov_b64 s[24:25], 0xf\ns_mov_b64 s[26:27], 1\ns_mov_b64 vcc, 0x5555" ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "vcc");
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#define X(N, NAME, TMPL)                                                                                                     \
    if constexpr (KIND == N) {                                                                                               \
        RUNV(TMPL)                                                                                                           \
    }
        KINDS(X)
#undef X
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)q0 + (unsigned)q1 + lds[threadIdx.x] == 0x12345u) sink[0] = 1;
}

template <int KIND>
void run(const char *name, int wgs_per_cu) {
    const int cus = 256, wgs = cus * wgs_per_cu, iters = 500, per_iter = 256;
    unsigned long long *d;
    unsigned *sink;
    (void)hipMalloc(&d, wgs * 8 * 8);
    (void)hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(512), 0, 0, d, sink, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(512), 0, 0, d, sink, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(wgs * 8);
    (void)hipMemcpy(h.data(), d, wgs * 64, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const int waves_per_simd = wgs_per_cu * 2;
    const double n = (double)iters * per_iter * waves_per_simd;
    printf("%-52s %d w/SIMD  %6.3f ns/instr/SIMD  %5.2f ticks/instr/SIMD  kernel %7.3f ms  %4.2f Gticks/s  -> chip %5.3f T wave-instr/s\n", name,
           waves_per_simd, ms * 1e6 / n, med / n, ms, med / ms / 1e6, 1024.0 / (ms * 1e6 / n) / 1e3);
    fflush(stdout);
    (void)hipFree(d);
    (void)hipFree(sink);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

template <int N>
void run_all(const char *filter, int w) {
    if constexpr (N < NKINDS) {
        const char *name = nullptr;
#define X(I, NAME, TMPL) if (N == I) name = NAME;
        KINDS(X)
#undef X
        if (name && (!filter || strstr(name, filter))) run<N>(name, w);
        run_all<N + 1>(filter, w);
    }
}

int main(int argc, char **argv) {
    const char *filter = argc > 1 ? argv[1] : nullptr;
    // warm the clocks: a second or so of the first kind
    for (int i = 0; i < 3; i++) run<0>("(warm-up) v_add_u32", 4);
    for (int w : {4, 1}) run_all<0>(filter, w);
    return 0;
}
