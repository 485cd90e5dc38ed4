// Micro-benchmark (development aid, not part of the product): what ONE SIMD of gfx950 sustains for the instruction
// classes of the render kernel, by waves per SIMD, by independent instructions per wave (ILP) and by EXEC mask.
// Every measured instruction is an `asm volatile` statement with opaque register operands, so the compiler can
// neither fold, fuse (v_pk_*) nor reorder them; tools/ubench/run_issue_rates3.sh disassembles the code object and
// checks that each kernel's loop holds exactly the instructions it claims (round 1's issue_rates2.hip let the
// compiler fold three rows and pack a fourth: VERDICT r1 item 8).
//
// Output: cycles per wave-instruction per wave (s_memtime ticks = shader cycles, MI355X_MICROARCH.md) and the SIMD's
// view (that divided by the waves it holds).
//
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/issue_rates3.hip -o tools/ubench/_build/issue_rates3
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

constexpr int kUnroll = 16;      // asm statements per register per loop iteration
constexpr int kIters = 256;

// One kernel per instruction form.  ILP independent chains x kUnroll statements per iteration.
#define S2R_UB_KERNEL(NAME, ILP, DECL, STMT)                                                            \
    __global__ void __launch_bounds__(1024) ub_##NAME##_ilp##ILP(unsigned long long *ticks, float *sink, \
                                                                 float a, float b, unsigned m, int half) { \
        DECL                                                                                            \
        if (half && (threadIdx.x & 32u)) return;      /* EXEC = lanes 0..31 of every wave */            \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                     \
        for (int i = 0; i < kIters; i++) {                                                              \
            _Pragma("unroll") for (int u = 0; u < kUnroll; u++) {                                       \
                _Pragma("unroll") for (int j = 0; j < ILP; j++) { STMT }                                \
            }                                                                                           \
        }                                                                                               \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                     \
        float s = 0.0f;                                                                                 \
        for (int j = 0; j < ILP; j++) s += x[j] + (float)y[j].x + z[j].w;                                        \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                \
        if ((threadIdx.x & 63u) == 0) ticks[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
    }

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
#define DECL_REGS(ILP)                                                                                  \
    float x[ILP]; f2 y[ILP]; unsigned w[ILP];                                                           \
    for (int j = 0; j < ILP; j++) { x[j] = threadIdx.x * 1e-3f + j; y[j] = (f2){x[j], x[j] + 1.0f}; w[j] = threadIdx.x * 77u + j; } \
    f2 a2 = {a, a}, b2 = {b, b}; (void)a2; (void)b2; (void)w;                                           \
    unsigned sreg[ILP]; for (int j = 0; j < ILP; j++) sreg[j] = m + j; (void)sreg;                       \
    __shared__ __attribute__((aligned(16))) float lds_buf[4096 + 1024]; const unsigned lds_addr = (unsigned)(size_t)(lds_buf) + threadIdx.x * 4u; (void)lds_addr; \
    const unsigned lds_addr8 = (unsigned)(size_t)(lds_buf) + threadIdx.x * 8u, lds_addr16 = (unsigned)(size_t)(lds_buf) + threadIdx.x * 16u; (void)lds_addr8; (void)lds_addr16; \
    f4v z[ILP]; for (int j = 0; j < ILP; j++) z[j] = (f4v){x[j], x[j], x[j], x[j]}; (void)z;            \
    asm volatile("s_mov_b32 m0, %0" : : "s"(__builtin_amdgcn_readfirstlane((int)((unsigned)(size_t)(lds_buf) + (threadIdx.x >> 6) * 256u))) : "m0");   \
    asm volatile("s_mov_b64 s[24:25], exec\n s_mov_b64 s[22:23], 0" : : : "s22", "s23", "s24", "s25");

#define FORMS(ILP)                                                                                                    \
    S2R_UB_KERNEL(fma, ILP, DECL_REGS(ILP), asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b));)   \
    S2R_UB_KERNEL(add, ILP, DECL_REGS(ILP), asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[j]) : "v"(a));)               \
    S2R_UB_KERNEL(mul, ILP, DECL_REGS(ILP), asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(a));)               \
    S2R_UB_KERNEL(pkfma, ILP, DECL_REGS(ILP), asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[j]) : "v"(a2), "v"(b2));) \
    S2R_UB_KERNEL(pkmul, ILP, DECL_REGS(ILP), asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[j]) : "v"(a2));)         \
    S2R_UB_KERNEL(pkadd, ILP, DECL_REGS(ILP), asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[j]) : "v"(a2));)         \
    S2R_UB_KERNEL(fract, ILP, DECL_REGS(ILP), asm volatile("v_fract_f32 %0, %0" : "+v"(x[j]));)                        \
    S2R_UB_KERNEL(cvtu, ILP, DECL_REGS(ILP), asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[j]));)                       \
    S2R_UB_KERNEL(xor, ILP, DECL_REGS(ILP), asm volatile("v_xor_b32 %0, %0, %1" : "+v"(w[j]) : "v"(m));)               \
    S2R_UB_KERNEL(addu, ILP, DECL_REGS(ILP), asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[j]) : "v"(m));)              \
    S2R_UB_KERNEL(ashr, ILP, DECL_REGS(ILP), asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(w[j]));)                    \
    S2R_UB_KERNEL(mullo, ILP, DECL_REGS(ILP), asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(w[j]) : "v"(m));)          \
    S2R_UB_KERNEL(pkmullo16, ILP, DECL_REGS(ILP), asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(w[j]) : "v"(m));)   \
    S2R_UB_KERNEL(pkaddu16, ILP, DECL_REGS(ILP), asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(w[j]) : "v"(m));)       \
    S2R_UB_KERNEL(cndmask, ILP, DECL_REGS(ILP), asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[j]) : "v"(a));)  \
    S2R_UB_KERNEL(cmp, ILP, DECL_REGS(ILP), asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[j]), "v"(a) : "vcc");)   \
    S2R_UB_KERNEL(rcp, ILP, DECL_REGS(ILP), asm volatile("v_rcp_f32 %0, %0" : "+v"(x[j]));)                            \
    S2R_UB_KERNEL(fma64, ILP, DECL_REGS(ILP), asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(y[j]) : "v"(a2), "v"(b2));) \
    S2R_UB_KERNEL(readlane, ILP, DECL_REGS(ILP), { unsigned s_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s_) : "v"(w[j])); }) \
    S2R_UB_KERNEL(max3, ILP, DECL_REGS(ILP), asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b));)  \
    /* a scalar instruction alone, and between two vector ones: does it take one of the wave's issue turns? */          \
    S2R_UB_KERNEL(salu, ILP, DECL_REGS(ILP), asm volatile("s_add_u32 %0, %0, 7" : "+s"(sreg[j]) : : "scc");)            \
    S2R_UB_KERNEL(fma_salu, ILP, DECL_REGS(ILP), asm volatile("v_fma_f32 %0, %0, %2, %3\n s_add_u32 %1, %1, 7" : "+v"(x[j]), "+s"(sreg[j]) : "v"(a), "v"(b) : "scc");) \
    /* a compare into an SGPR pair + the scalar OR that collects it; a select on an SGPR-pair mask */                  \
    S2R_UB_KERNEL(cmp_sor, ILP, DECL_REGS(ILP), asm volatile("v_cmp_eq_f32 s[20:21], %0, %1\n s_or_b64 s[22:23], s[22:23], s[20:21]" : : "v"(x[j]), "v"(a) : "s20", "s21", "s22", "s23", "scc");) \
    S2R_UB_KERNEL(cnd_sgpr, ILP, DECL_REGS(ILP), asm volatile("v_cndmask_b32 %0, %0, %1, s[24:25]" : "+v"(x[j]) : "v"(a) : "s24", "s25");) \
    /* LDS stores and loads alone and between vector instructions */                                                    \
    S2R_UB_KERNEL(ldsw, ILP, DECL_REGS(ILP), asm volatile("ds_write_b32 %0, %1" : : "v"(lds_addr), "v"(x[j]) : "memory");)  \
    S2R_UB_KERNEL(fma_ldsw, ILP, DECL_REGS(ILP), asm volatile("v_fma_f32 %0, %0, %2, %3\n ds_write_b32 %1, %0" : "+v"(x[j]) : "v"(lds_addr), "v"(a), "v"(b) : "memory");) \
    S2R_UB_KERNEL(fma3_ldsw, ILP, DECL_REGS(ILP), asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n ds_write_b32 %1, %0" : "+v"(x[j]) : "v"(lds_addr), "v"(a), "v"(b) : "memory");) \
    /* the other LDS forms: wide stores, the address-free store (address = M0 + offset + 4 * lane), loads */             \
    S2R_UB_KERNEL(ldsw2, ILP, DECL_REGS(ILP), asm volatile("ds_write2_b32 %0, %1, %1 offset1:65" : : "v"(lds_addr), "v"(x[j]) : "memory");) \
    S2R_UB_KERNEL(ldsw64, ILP, DECL_REGS(ILP), asm volatile("ds_write_b64 %0, %1" : : "v"(lds_addr8), "v"(y[j]) : "memory");) \
    S2R_UB_KERNEL(ldsw128, ILP, DECL_REGS(ILP), asm volatile("ds_write_b128 %0, %1" : : "v"(lds_addr16), "v"(z[j]) : "memory");) \
    S2R_UB_KERNEL(ldswtid, ILP, DECL_REGS(ILP), asm volatile("ds_write_addtid_b32 %0 offset:260" : : "v"(x[j]) : "memory");) \
    S2R_UB_KERNEL(fma3_ldswtid, ILP, DECL_REGS(ILP), asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n ds_write_addtid_b32 %0 offset:260" : "+v"(x[j]) : "v"(a), "v"(b) : "memory");) \
    S2R_UB_KERNEL(ldsr, ILP, DECL_REGS(ILP), { float t_; asm volatile("ds_read_b32 %0, %1" : "=v"(t_) : "v"(lds_addr) : "memory"); }) \
    S2R_UB_KERNEL(ldsr2, ILP, DECL_REGS(ILP), { f2 t_; asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(t_) : "v"(lds_addr) : "memory"); }) \
    S2R_UB_KERNEL(ldsr64, ILP, DECL_REGS(ILP), { f2 t_; asm volatile("ds_read_b64 %0, %1" : "=v"(t_) : "v"(lds_addr8) : "memory"); }) \
    S2R_UB_KERNEL(ldsr128, ILP, DECL_REGS(ILP), { f4v t_; asm volatile("ds_read_b128 %0, %1" : "=v"(t_) : "v"(lds_addr16) : "memory"); }) \
    S2R_UB_KERNEL(fma3_ldsr128, ILP, DECL_REGS(ILP), { f4v t_; asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n ds_read_b128 %1, %4" : "+v"(x[j]), "=v"(t_) : "v"(a), "v"(b), "v"(lds_addr16) : "memory"); }) \
    S2R_UB_KERNEL(fma_pkfma, ILP, DECL_REGS(ILP), asm volatile("v_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %4, %5" : "+v"(x[j]), "+v"(y[j]) : "v"(a), "v"(b), "v"(a2), "v"(b2));)

FORMS(1)
FORMS(4)
FORMS(8)

typedef void (*kern_t)(unsigned long long *, float *, float, float, unsigned, int);
struct Row { const char *name; kern_t k1, k4, k8; };
#define ROW(NAME) {#NAME, ub_##NAME##_ilp1, ub_##NAME##_ilp4, ub_##NAME##_ilp8}

int main(int argc, char **argv) {
    const Row rows[] = {ROW(fma), ROW(add), ROW(mul), ROW(pkfma), ROW(pkmul), ROW(pkadd), ROW(fract), ROW(cvtu), ROW(xor),
                        ROW(addu), ROW(ashr), ROW(mullo), ROW(pkmullo16), ROW(pkaddu16), ROW(cndmask), ROW(cmp), ROW(rcp),
                        ROW(fma64), ROW(readlane), ROW(max3), ROW(salu), ROW(fma_salu), ROW(cmp_sor), ROW(cnd_sgpr), ROW(ldsw),
                        ROW(fma_ldsw), ROW(fma3_ldsw), ROW(ldsw2), ROW(ldsw64), ROW(ldsw128), ROW(ldswtid), ROW(fma3_ldswtid), ROW(ldsr),
                        ROW(ldsr2), ROW(ldsr64), ROW(ldsr128), ROW(fma3_ldsr128), ROW(fma_pkfma)};
    unsigned long long *ticks; float *sink;
    hipMalloc(&ticks, 256 * 16 * sizeof(unsigned long long));
    hipMalloc(&sink, 256 * 1024 * sizeof(float));
    std::vector<unsigned long long> h(256 * 16);
    printf("# cycles per wave-instruction as one wave sees them (median over waves, s_memtime) | the SIMD's cycles per wave-instruction\n");
    printf("# 256 workgroups (one per CU); %d statements per chain per iteration, %d iterations\n", kUnroll, kIters);
    printf("%-10s %4s %5s %5s | %8s %8s\n", "instr", "ilp", "w/SIMD", "exec", "per wave", "per SIMD");
    setvbuf(stdout, nullptr, _IOLBF, 0);
    for (const Row &r : rows) {
        if (argc > 1 && !strstr(argv[1], r.name)) continue;      // run only the rows named in argv[1] ("fma,ldsw,...")
        for (int ilp : {1, 8}) {
            kern_t k = ilp == 1 ? r.k1 : ilp == 4 ? r.k4 : r.k8;
            for (int waves : {1, 2, 4}) {
                for (int half : {0, 1}) {
                    if (half && (ilp != 8 || waves != 4)) continue;      // (round 2, first run: a half-empty EXEC costs the same everywhere)
                    const int threads = 256 * waves;
                    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, ticks, sink, 1.0001f, 0.5f, 0x1234567u, half);
                    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, ticks, sink, 1.0001f, 0.5f, 0x1234567u, half);
                    hipDeviceSynchronize();
                    const int n_w = 256 * threads / 64;
                    hipMemcpy(h.data(), ticks, n_w * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                    std::sort(h.begin(), h.begin() + n_w);
                    const double cyc = (double)h[n_w / 2] / ((double)kIters * kUnroll * ilp);
                    // (rows that name several instructions: cycles per STATEMENT, i.e. per group of them)
                    printf("%-10s %4d %5d %5s | %8.2f %8.2f\n", r.name, ilp, waves, half ? "0-31" : "all", cyc, cyc / waves);
                }
            }
        }
    }
    return 0;
}
