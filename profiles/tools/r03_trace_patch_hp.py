"""TEMPORARY instrumentation of the LDS-DMA GEMM (gemm_hp.hip): s_memtime stamps per wavefront and stage, kept in LDS and copied to a
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
s=s.replace("  p.slab = 0;\n","  p.slab = 0;\n  p.trace = nullptr;\n",1)
open(p,'w').write(s)
p=base+'gemm_hp.hip'
s=open(p).read()
def rep(o,n):
    global s
    assert o in s, o[:70]
    s=s.replace(o,n,1)
rep('  int buf = 0;\n  for (;;) {\n    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront\'s pieces of the stage have landed ...\n    __syncthreads();                                   // ... and everybody\'s; everybody is done with the other buffer\n',
'''  int buf = 0;
  int tr_stage = 0;
  const bool traced = p.g.trace != nullptr && (blockIdx.x == 0 || blockIdx.x == 100);
  auto stamp = [&](int slot) __attribute__((always_inline)) {
    if (traced && tr_stage < 64 && lane == 0)
      reinterpret_cast<unsigned int *>(scratch)[(wave * 64 + tr_stage) * 4 + slot] = (unsigned int)__builtin_readcyclecounter();
  };
  for (;;) {
    stamp(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's pieces of the stage have landed ...
    stamp(1);
    __syncthreads();                                   // ... and everybody's; everybody is done with the other buffer
    stamp(2);
''')
rep("    if (more_loads) more_loads = advance(ld);\n    if (cc.k0 + BK >= cc.it.kend) {  // that was the item's last stage (uniform)",
    "    __builtin_amdgcn_sched_barrier(0);\n    stamp(3);\n    ++tr_stage;\n    if (more_loads) more_loads = advance(ld);\n    if (traced && tr_stage == 64) {\n      __syncthreads();\n      for (int i = tid; i < 8 * 64 * 4; i += G::THREADS) p.g.trace[(blockIdx.x == 0 ? 0 : 8 * 64 * 4) + i] = reinterpret_cast<unsigned int *>(scratch)[i];\n      __syncthreads();\n    }\n    if (cc.k0 + BK >= cc.it.kend) {  // that was the item's last stage (uniform)")
open(p,'w').write(s)
s=open(p).read()
s=s.replace("  g.slab = 0;\n  float *user_c = a->C;","  g.slab = 0;\n  g.trace = g_trace_ptr;\n  float *user_c = a->C;",1)
s=s.replace("namespace pthp {","extern unsigned int *g_trace_ptr;\nnamespace pthp {",1)
open(p,'w').write(s)
