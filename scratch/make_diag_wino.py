"""Builds scratch/libsrgan_diag.so: wino_kernel with s_memtime stamps around each phase of the main loop
(the epilogue is replaced by a dump of the per-wave cycle counts).  Read with scratch/diag_wino.py.
Development tool only; the product library is untouched."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "style-restricted_gan_amd", "csrc")
s = open(os.path.join(csrc, "conv_wino.hip")).read()

def sub(old, new):
    global s
    assert s.count(old) >= 1, old[:60]
    s = s.replace(old, new, 1)

sub('''  const int nk = p.nchunk;
  load_x();
  load_u(0, I0{});''', '''  const int nk = p.nchunk;
  long long t_store = 0, t_bar = 0, t_m0 = 0, t_m1 = 0, t_tail = 0;
  const long long t_begin = __builtin_amdgcn_s_memtime();
  load_x();
  load_u(0, I0{});''')
sub('''    read_frags(P, 1);
    mfma_steps(par, 0, 0, 2);
    if constexpr (ST) {''', '''    const long long q0 = __builtin_amdgcn_s_memtime();
    read_frags(P, 1);
    mfma_steps(par, 0, 0, 2);
    __builtin_amdgcn_sched_barrier(0);
    const long long q1 = __builtin_amdgcn_s_memtime();
    t_m0 += q1 - q0;
    if constexpr (ST) {''')
sub('''      if constexpr (LD) load_x();
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_steps(par, 0, 2, 4);
    mfma_steps(par, 1, 0, 2);
    __builtin_amdgcn_sched_barrier(0);     // keep those MFMAs in front of the barrier
    __syncthreads();                       // chunk kc+1 visible; every wave holds its last fragments of chunk kc
    if constexpr (ST) read_frags(1 - P, 0);
    mfma_steps(par, 1, 2, 4);
  };''', '''      if constexpr (LD) load_x();
      __builtin_amdgcn_sched_barrier(0);
    }
    const long long q2 = __builtin_amdgcn_s_memtime();
    t_store += q2 - q1;
    mfma_steps(par, 0, 2, 4);
    mfma_steps(par, 1, 0, 2);
    __builtin_amdgcn_sched_barrier(0);
    const long long q3 = __builtin_amdgcn_s_memtime();
    t_m1 += q3 - q2;
    __syncthreads();
    const long long q4 = __builtin_amdgcn_s_memtime();
    t_bar += q4 - q3;
    if constexpr (ST) read_frags(1 - P, 0);
    mfma_steps(par, 1, 2, 4);
    __builtin_amdgcn_sched_barrier(0);
    t_tail += __builtin_amdgcn_s_memtime() - q4;
  };''')
sub('''    read_frags(P, 1);
    mfma_steps(par, 0, 0, 4);
    mfma_steps(par, 1, 0, 2);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if constexpr (R1) read_frags(1 - P, 0);
    if constexpr (S2) {
      __builtin_amdgcn_sched_barrier(0);
      store_v(P);''', '''    const long long q0 = __builtin_amdgcn_s_memtime();
    read_frags(P, 1);
    mfma_steps(par, 0, 0, 4);
    __builtin_amdgcn_sched_barrier(0);
    const long long q1 = __builtin_amdgcn_s_memtime();
    t_m0 += q1 - q0;
    mfma_steps(par, 1, 0, 2);
    __builtin_amdgcn_sched_barrier(0);
    const long long q2 = __builtin_amdgcn_s_memtime();
    t_m1 += q2 - q1;
    __syncthreads();
    const long long q3 = __builtin_amdgcn_s_memtime();
    t_bar += q3 - q2;
    if constexpr (R1) read_frags(1 - P, 0);
    if constexpr (S2) {
      __builtin_amdgcn_sched_barrier(0);
      store_v(P);''')
sub('''      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_steps(par, 1, 2, 4);
    if constexpr (S2) load_u(kc + 2, par);''', '''      __builtin_amdgcn_sched_barrier(0);
    }
    const long long q4 = __builtin_amdgcn_s_memtime();
    t_store += q4 - q3;
    mfma_steps(par, 1, 2, 4);
    if constexpr (S2) load_u(kc + 2, par);
    __builtin_amdgcn_sched_barrier(0);
    t_tail += __builtin_amdgcn_s_memtime() - q4;''')
sub('''  const int cl = tid & 31, tg = tid >> 5;''', '''  {
    const long long t_loop = __builtin_amdgcn_s_memtime() - t_begin;
    if (blockIdx.x == 17 && lane == 0) {
      float* o = p.dst + wave * 8;
      o[0] = (float)t_loop; o[1] = (float)t_store; o[2] = (float)t_m0; o[3] = (float)t_m1; o[4] = (float)t_bar; o[5] = (float)t_tail;
      o[6] = (float)nk; o[7] = (float)cls;
    }
    float keep = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) keep += acc[a][i][j][e];
    p.dst[4096 + (size_t)blockIdx.x * 512 + tid] = keep;
    return;
  }
  const int cl = tid & 31, tg = tid >> 5;''')
tmp = os.path.join(csrc, "conv_wino_diag.hip")
open(tmp, "w").write(s)
try:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                           "-I" + os.path.join(root, "include"), "-I" + csrc, "-c", tmp, "-o", "/tmp/wino_diag.o"])
finally:
    os.remove(tmp)
objs = [os.path.join(csrc, "build", n + ".o") for n in
        ("conv_igemm", "conv_narrow", "norm", "pointwise", "losses", "preprocess", "api")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "/tmp/wino_diag.o",
                       "-o", os.path.join(root, "scratch", "libsrgan_diag.so")])
print("built scratch/libsrgan_diag.so")
