// Self-test of the cross-lane primitives of dynamicslamtool_amd/csrc/kernels_common.h (DPP, v_permlane16/32_swap, v_readlane) against their ds_bpermute forms,
// on random data, one wave per check.  Built and run by tests/test_dpp_primitives.py on the GPU box:  hipcc --offload-arch=gfx950 -I csrc dpp_selftest.hip && ./a.out
#include "mor_device.h"
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels_common.h"

struct Pair { int c, b; };
__global__ void k_check(const int *in, const float *fin, const long long *lin, int *bad) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, i = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = in[i]; const float f = fin[i]; const long long q = lin[i];
  int err = 0;
  // all-reduce: sum (int, long long), min / max (float, int), lexicographic best of a pair
  int s = v; float mn = f, mx = f; long long qs = q; int imn = v; Pair best = {v & 7, v};
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64); mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); imn = min(imn, __shfl_xor(imn, o, 64));
    qs += ((long long)__shfl_xor((int)(qs >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)(unsigned)qs, o, 64);
    const int c2 = __shfl_xor(best.c, o, 64), b2 = __shfl_xor(best.b, o, 64); if (c2 > best.c || (c2 == best.c && b2 < best.b)) { best.c = c2; best.b = b2; }
  }
  err |= (wave_sum(v) != s) << 0; err |= (wave_fmin(f) != mn) << 1; err |= (wave_fmax(f) != mx) << 2; err |= (wave_sum(q) != qs) << 3; err |= (wave_imin(v) != imn) << 4;
  const int2 b3 = wave_allreduce(make_int2(v & 7, v), [](int2 a, int2 b) { return (b.x > a.x || (b.x == a.x && b.y < a.y)) ? b : a; });
  err |= (b3.x != best.c || b3.y != best.b) << 5;
  // group sums of floats: bit-identical to the xor butterfly that starts with the nearest partner (offsets 1, 2, 4, …)
  { float a4 = f, a16 = f; for (int o = 1; o < 4; o <<= 1) a4 += __shfl_xor(a4, o, 64); for (int o = 1; o < 16; o <<= 1) a16 += __shfl_xor(a16, o, 64);
    auto add = [](float a, float b) { return a + b; };
    err |= (__float_as_int(wave_group_allreduce<4>(f, add)) != __float_as_int(a4)) << 6; err |= (__float_as_int(wave_group_allreduce<16>(f, add)) != __float_as_int(a16)) << 7;
    float a64 = f; for (int o = 1; o < 64; o <<= 1) a64 += __shfl_xor(a64, o, 64);
    err |= (__float_as_int(wave_group_allreduce<64>(f, add)) != __float_as_int(a64)) << 8; }
  // inclusive scan, shifts by one, broadcast of a uniform lane
  { int sc = v & 1023; const int x = sc; for (int o = 1; o < 64; o <<= 1) { const int n = __shfl_up(sc, o, 64); if (l >= o) sc += n; }
    err |= (wave_incl_scan(x) != sc) << 9; }
  { int up = __shfl_up(v, 1, 64); if (l == 0) up = -5; int dn = __shfl_down(v, 1, 64); if (l == 63) dn = -6;
    err |= (wave_shift_up1(v, -5) != up) << 10; err |= (wave_shift_down1(v, -6) != dn) << 11;
    long long qu = ((long long)__shfl_up((int)(q >> 32), 1, 64) << 32) | (unsigned)__shfl_up((int)(unsigned)q, 1, 64); if (l == 0) qu = 77;
    err |= (wave_shift_up1(q, 77ll) != qu) << 12; }
  { const int pick = (in[blockIdx.x * blockDim.x + w * 64] >> 3) & 63;   // uniform in the wave
    err |= (wave_bcast(v, pick) != __shfl(v, pick, 64)) << 13; err |= (wave_bcast(q, 63) != (((long long)__shfl((int)(q >> 32), 63, 64) << 32) | (unsigned)__shfl((int)(unsigned)q, 63, 64))) << 14; }
  // segmented inclusive sums through the DPP steps of acc_segmented_scan's shape (heads from the data): a long long payload
  { const bool head = l == 0 || (v & 3) == 0; const unsigned long long heads = __ballot(head);
    const int hl = 63 - __clzll((long long)(heads & (lanemask_lt() | (1ull << l))));
    long long ref = q; for (int o = 1; o < 64; o <<= 1) { const long long t = ((long long)__shfl_up((int)(ref >> 32), o, 64) << 32) | (unsigned)__shfl_up((int)(unsigned)ref, o, 64); if (l - o >= hl) ref += t; }
    long long S = q; const int li = l & 15;
#define STEP(CTRL, ROWS, COND) { const WaveWords<long long> a = to_words(S); WaveWords<long long> b; for (int k = 0; k < 2; ++k) b.w[k] = dpp_mov<CTRL, ROWS>(a.w[k], a.w[k]); const long long t = from_words<long long>(b); if (COND) S += t; }
    STEP(0x111, 0xF, li >= 1 && l - 1 >= hl) STEP(0x112, 0xF, li >= 2 && l - 2 >= hl) STEP(0x114, 0xF, li >= 4 && l - 4 >= hl) STEP(0x118, 0xF, li >= 8 && l - 8 >= hl)
    STEP(0x142, 0xA, (l & 16) && hl < (l & ~15)) STEP(0x143, 0xC, l >= 32 && hl < 32)
#undef STEP
    err |= (S != ref) << 15; }
  if (err) atomicOr(bad, err);
}

int main() {
  const int n = 256 * 64;
  std::vector<int> a(n); std::vector<float> f(n); std::vector<long long> q(n);
  srand(12345);
  for (int i = 0; i < n; ++i) { a[i] = rand() - RAND_MAX / 2; f[i] = (float)(rand() % 200001 - 100000) * 0.001f; q[i] = ((long long)rand() << 24) ^ rand(); }
  int *da, *dbad; float *df; long long *dq; int bad = 0;
  if (hipMalloc(&da, n * 4) != hipSuccess || hipMalloc(&df, n * 4) != hipSuccess || hipMalloc(&dq, n * 8) != hipSuccess || hipMalloc(&dbad, 4) != hipSuccess) { printf("no device memory\n"); return 2; }
  hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(df, f.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dq, q.data(), n * 8, hipMemcpyHostToDevice); hipMemset(dbad, 0, 4);
  hipLaunchKernelGGL(k_check, dim3(64), dim3(256), 0, 0, da, df, dq, dbad);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
  hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
  printf("dpp_selftest: mismatch mask 0x%x over %d lanes\n", bad, n);
  return bad ? 1 : 0;
}
