"""TEMPORARY instrumentation of the staging GEMM (gemm_split_kernel.h): s_memtime stamps per wavefront and stage, kept in LDS and copied to a
buffer set by ptamd_debug_trace().  Applies in place to the sources - use through profiles/tools/r03_trace_build.sh, which
builds libptamd_trace.so and restores the tree.  Readers: r03_gemm_trace.py / r03_hp_trace.py (PTAMD_LIB_TAG=trace)."""
import os
base = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'protein_transformer_amd', 'csrc') + os.sep
p=base+'gemm_common.h'
s=open(p).read()
s=s.replace("  int reserved_cus;  // CUs the persistent grid leaves free (room for a concurrent collective kernel); 0 = none\n};","  int reserved_cus;  // CUs the persistent grid leaves free (room for a concurrent collective kernel); 0 = none\n  unsigned int *trace;\n};")
open(p,'w').write(s)
p=base+'gemm.hip'
s=open(p).read()
s=s.replace('extern "C" {\n','unsigned int *g_trace_ptr = nullptr;\nextern "C" {\nvoid ptamd_debug_trace(void *ptr) { g_trace_ptr = (unsigned int *)ptr; }\n',1)
s=s.replace("  p.slab = 0;\n","  p.slab = 0;\n  p.trace = g_trace_ptr;\n",1)
open(p,'w').write(s)
p=base+'gemm_split_kernel.h'
s=open(p).read()
def rep(o,n):
    global s
    assert o in s, o[:60]
    s=s.replace(o,n,1)
rep("  constexpr bool F16 = NPROD == 3;  // two scaled f16 terms and three products instead of three bf16 terms and six\n",
"""  constexpr bool F16 = NPROD == 3;  // two scaled f16 terms and three products instead of three bf16 terms and six
  unsigned int *const tr_lds = reinterpret_cast<unsigned int *>(cs_area);
  const bool traced = p.trace != nullptr && (blockIdx.x == 0 || blockIdx.x == 100);
  auto stamp = [&](int stage, int slot) __attribute__((always_inline)) {
    if (traced && stage < 64 && (threadIdx.x & 63) == 0)
      tr_lds[((threadIdx.x >> 6) * 64 + stage) * 4 + slot] = (unsigned int)__builtin_readcyclecounter();
  };
""")
rep("      for (int u = 1; u <= NSETS; ++u) {\n        produce(ra[u % NSETS], rb[u % NSETS], rskip[u % NSETS], rsa[u % NSETS], rsb[u % NSETS], u & 1);  // stage g + u -> buffer (g + u) & 1\n        __syncthreads();",
"      for (int u = 1; u <= NSETS; ++u) {\n        tr_stage = g + u;\n        stamp(tr_stage, 0);\n        produce(ra[u % NSETS], rb[u % NSETS], rskip[u % NSETS], rsa[u % NSETS], rsb[u % NSETS], u & 1);  // stage g + u -> buffer (g + u) & 1\n        stamp(tr_stage, 3);\n        __syncthreads();")
rep("    auto produce = [&](float4 (&a)[NVA], float4 (&b)[NVB], int &kskip, float (&sa_)[MAX_NV], float (&sb_)[MAX_NV], int buf) __attribute__((always_inline)) {\n      unsigned short *sa = smem + buf * STAGE, *sb = sa + G::NPLANES * PLANE_A;\n",
"    int tr_stage = 0;\n    auto produce = [&](float4 (&a)[NVA], float4 (&b)[NVB], int &kskip, float (&sa_)[MAX_NV], float (&sb_)[MAX_NV], int buf) __attribute__((always_inline)) {\n      unsigned short *sa = smem + buf * STAGE, *sb = sa + G::NPLANES * PLANE_A;\n      stamp(tr_stage, 1);\n")
rep("      __builtin_amdgcn_sched_barrier(0);\n      if (F16) fetch_tail(sa_, sb_);","      __builtin_amdgcn_sched_barrier(0);\n      stamp(tr_stage, 2);\n      if (F16) fetch_tail(sa_, sb_);")
rep("      const unsigned short *sa = smem + (g & 1) * STAGE, *sb = sa + G::NPLANES * PLANE_A;\n      bf16x8 fa[TI][3], fb[2][3];","      const unsigned short *sa = smem + (g & 1) * STAGE, *sb = sa + G::NPLANES * PLANE_A;\n      stamp(g, 0);\n      bf16x8 fa[TI][3], fb[2][3];")
rep("      if (NPROD != 3) { mul(1, 1); mul(1, 0); mul(0, 1); mul(0, 0); }\n      __syncthreads();  // buffer g & 1 is released, buffer (g + 1) & 1 holds stage g + 1",
"      if (NPROD != 3) { mul(1, 1); mul(1, 0); mul(0, 1); mul(0, 0); }\n      __builtin_amdgcn_sched_barrier(0);\n      stamp(g, 1);\n      __syncthreads();  // buffer g & 1 is released, buffer (g + 1) & 1 holds stage g + 1\n      stamp(g, 2);")
rep("      advance(cc);\n    }\n  }\n}\n\nconstexpr int TI2_COST_NUM","      advance(cc);\n    }\n  }\n  __syncthreads();\n  if (traced) {\n    unsigned int *dst = p.trace + (blockIdx.x == 0 ? 0 : 8 * 64 * 4);\n    for (int i = threadIdx.x; i < 8 * 64 * 4; i += NTHREADS) dst[i] = tr_lds[i];\n  }\n}\n\nconstexpr int TI2_COST_NUM")
open(p,'w').write(s)
