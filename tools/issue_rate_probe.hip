// Issue rates of the instructions the tensor-game kernels lean on (gfx950), whole chip, W wavefronts per SIMD:
// every wavefront runs N instructions of one kind over 8 independent registers (no dependent-issue stalls);
// events give wave-instructions per second per SIMD, s_memtime (shader clock) gives cycles per instruction of wave 0.
// Build: hipcc -O3 --offload-arch=gfx950 tools/issue_rate_probe.hip -o issue_rate_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define BODY8(ASM)                                                                               \
  asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                            \
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
               : "v"(y), "v"(z))

#define K_VALU(NAME, ASM)                                                                        \
  __global__ void NAME(int n, unsigned* sink, unsigned long long* ticks) {                       \
    unsigned r[8], y = blockIdx.x | 1, z = threadIdx.x;                                          \
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x + i;                                          \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                  \
    _Pragma("unroll 1") for (int i = 0; i < n; i += 32) { BODY8(ASM); BODY8(ASM); BODY8(ASM); BODY8(ASM); } \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                  \
    unsigned x = 0;                                                                              \
    for (int i = 0; i < 8; ++i) x ^= r[i];                                                       \
    if (x == 0xdeadbeef) sink[0] = x;                                                            \
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;                                 \
  }

#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n\t"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n\t"
#define A_SDWA(i) "v_mul_i32_i24_sdwa %" #i ", sext(%8), sext(%9) dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_1\n\t"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n\t"
#define A_MAX3(i) "v_max3_i32 %" #i ", %" #i ", %8, %9\n\t"
#define A_ALIGN(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, 1\n\t"
#define A_MAD24(i) "v_mad_i32_i24 %" #i ", %8, %9, %" #i "\n\t"
#define A_PKMAD(i) "v_pk_mad_i16 %" #i ", %8, %9, %" #i "\n\t"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n\t"
#define A_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n\t"
#define A_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n\t"

#define A_DPP(i) "v_add_u32_dpp %" #i ", %8, %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define A_BFE(i) "v_bfe_i32 %" #i ", %" #i ", 8, 8\n\t"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n\t"
#define A_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n\t"
K_VALU(k_add, A_ADD)
K_VALU(k_mov, A_MOV)
K_VALU(k_sdwa, A_SDWA)
K_VALU(k_perm, A_PERM)
K_VALU(k_max3, A_MAX3)
K_VALU(k_align, A_ALIGN)
K_VALU(k_mad24, A_MAD24)
K_VALU(k_pkmad, A_PKMAD)
K_VALU(k_mullo, A_MULLO)
K_VALU(k_mulhi, A_MULHI)
K_VALU(k_mul24, A_MUL24)
K_VALU(k_dpp, A_DPP)
K_VALU(k_bfe, A_BFE)
K_VALU(k_xor, A_XOR)
K_VALU(k_or3, A_OR3)

__global__ void k_mad64(int n, unsigned* sink, unsigned long long* ticks) {
  unsigned long long r[8];
  unsigned y = blockIdx.x | 1, z = threadIdx.x;
  for (int i = 0; i < 8; ++i) r[i] = threadIdx.x + i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < n; i += 32) {
#pragma unroll
    for (int j = 0; j < 32; ++j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[j & 7]) : "v"(y), "v"(z) : "vcc");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long x = 0;
  for (int i = 0; i < 8; ++i) x ^= r[i];
  if (x == 0xdeadbeefull) sink[0] = (unsigned)x;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

__global__ void k_mfma(int n, unsigned* sink, unsigned long long* ticks) {
  v16i acc[2];
  for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0;
  v4i a = {1, 2, 3, (int)threadIdx.x}, b = {4, 5, 6, (int)blockIdx.x};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < n; i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j & 1], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  int x = 0;
  for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) x ^= acc[j][i];
  if (x == 0x7eadbeef) sink[0] = x;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE>  // 1 ds_read_b32, 2 ds_read_u8, 3 ds_write_b32, 4 ds_write_b128
__global__ void k_lds(int n, unsigned* sink, unsigned long long* ticks) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned x = 0;
  unsigned base = wave * 8192;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < n; i += 8) {
    asm volatile("" : "+v"(base));  // (the addresses are not loop invariant for the compiler)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 1) x ^= *reinterpret_cast<const unsigned*>(lds + base + 4 * lane + 256 * j);
      if (MODE == 2) x ^= lds[base + lane + 80 * j];
      if (MODE == 3) *reinterpret_cast<unsigned*>(lds + base + 4 * lane + 256 * j) = x + j;
      if (MODE == 4) *reinterpret_cast<uint4*>(lds + base + 16 * lane + 1024 * (j & 3)) = uint4{x, x, x, x + j};
    }
    asm volatile("" : "+v"(x));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (x == 0xdeadbeef) sink[0] = x + lds[lane];
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
  unsigned long long* d; unsigned* sink;
  CK(hipMalloc(&d, 64)); CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const int n = 1 << 18;
  struct K { const char* name; void (*fn)(int, unsigned*, unsigned long long*); };
  K ks[] = {{"v_add_u32", k_add}, {"v_mov_b32", k_mov}, {"v_xor_b32", k_xor}, {"v_or3_b32", k_or3}, {"v_bfe_i32", k_bfe}, {"v_mul_i32_i24_sdwa (byte sel, preserve)", k_sdwa}, {"v_perm_b32", k_perm},
            {"v_max3_i32", k_max3}, {"v_alignbyte_b32", k_align}, {"v_mad_i32_i24", k_mad24}, {"v_pk_mad_i16", k_pkmad},
            {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_mul_u32_u24", k_mul24}, {"v_mad_u64_u32", k_mad64}, {"v_add_u32_dpp row_shr", k_dpp}, {"v_mfma_i32_32x32x32_i8", k_mfma},
            {"ds_read_b32", k_lds<1>},
            {"ds_read_u8", k_lds<2>}, {"ds_write_b32", k_lds<3>}, {"ds_write_b128", k_lds<4>}};
  printf("%-42s %28s %28s\n", "instruction", "1 wave/SIMD: cyc/instr", "4 waves/SIMD: cyc/instr per SIMD");
  for (auto& k : ks) {
    double res[2];
    for (int wi = 0; wi < 2; ++wi) {
      const int wps = wi ? 4 : 1;
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k.fn, dim3(cus * wps), dim3(256), 0, 0, n, sink, d); CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize()); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      unsigned long long h; CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
      res[wi] = wi ? (best * 1e-3 * 2.4e9) / ((double)n * wps) : (double)h / n;  // wi=1: wall at 2.4 GHz / instructions per SIMD
    }
    printf("%-42s %28.2f %28.2f\n", k.name, res[0], res[1]);
  }
  return 0;
}
