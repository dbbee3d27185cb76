// micro-benchmark of the priority-mode heap primitives (devprio.hip): H heaps of n random keys, one workgroup each,
// `pops` pop-mins per heap, then `pops` pushes.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../space_filling_forest_star_amd/csrc
//   -I../../include heap_microbench.hip -o heap_microbench
#include "devprio.hip"
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
using namespace sffk;
__global__ __launch_bounds__(64) void k_micro(int32_t* v, double* key, int32_t* pos, int32_t* size, int cap, int pops, long long* out, int mode) {
  const int h = blockIdx.x;
  HeapRef hr;
  hr.v = v + (size_t)h * cap; hr.key = key + (size_t)h * cap; hr.pos = pos + (size_t)h * cap; hr.size_p = size + h;
  hr.n = uni_i32(size[h]);
#ifdef SFFK_PRIO_DEBUG
  for (int i = 0; i < 6; ++i) hr.dbg[i] = 0;
#endif
  const long long t0 = wall_clock64();
  int acc = 0;
  if (mode == 0) for (int i = 0; i < pops; ++i) acc += heap_pop(hr);
  else if (mode == 1) for (int i = 0; i < pops; ++i) heap_push(hr, cap - 1 - i, hk_bits(1.0 + 1e-3 * ((i * 7919) % 1000)));
  else for (int i = 0; i < pops; ++i) { const int nd = (int)(((unsigned)i * 2654435761u) % (unsigned)(cap - 4096)); heap_remove(hr, nd, hr.pos[nd]); }
  heap_drain();
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out[2 * h] = t1 - t0; out[2 * h + 1] = acc; size[h] = hr.n; }
}
int main(int argc, char** argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 90, n = argc > 2 ? atoi(argv[2]) : 30000, pops = argc > 3 ? atoi(argv[3]) : 90;
  const int cap = n + 4096;
  std::mt19937_64 g(1);
  std::vector<int32_t> v((size_t)H * cap), pos((size_t)H * cap, -1), size(H, n);
  std::vector<double> key((size_t)H * cap);
  for (int h = 0; h < H; ++h) {
    std::vector<std::pair<double, int>> e(n);
    for (int i = 0; i < n; ++i) e[i] = {std::uniform_real_distribution<double>(0, 100)(g), i};
    std::make_heap(e.begin(), e.end(), std::greater<>());
    for (int i = 0; i < n; ++i) { v[(size_t)h * cap + i] = e[i].second; key[(size_t)h * cap + i] = e[i].first; pos[(size_t)h * cap + e[i].second] = i; }
  }
  int32_t *dv, *dp, *ds; double* dk; long long* dout;
  hipMalloc(&dv, v.size() * 4); hipMalloc(&dp, pos.size() * 4); hipMalloc(&ds, H * 4); hipMalloc(&dk, key.size() * 8); hipMalloc(&dout, H * 16);
  char* junk; hipMalloc(&junk, 512 << 20);
  for (int rep = 0; rep < 1; ++rep)
    for (int mode : {0, 1, 2}) {
      if (mode != 1) {
        hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dp, pos.data(), pos.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dk, key.data(), key.size() * 8, hipMemcpyHostToDevice); hipMemcpy(ds, size.data(), H * 4, hipMemcpyHostToDevice);
      }
      hipMemset(junk, rep, 512 << 20);   // (evict the caches)
      hipDeviceSynchronize();
      hipLaunchKernelGGL(k_micro, dim3(H), dim3(64), 0, 0, dv, dk, dp, ds, cap, pops, dout, mode);
      hipDeviceSynchronize();
      std::vector<long long> out(2 * H);
      hipMemcpy(out.data(), dout, H * 16, hipMemcpyDeviceToHost);
      long long mx = 0, sum = 0;
      for (int h = 0; h < H; ++h) { mx = std::max(mx, out[2 * h]); sum += out[2 * h]; }
      printf("%s mode %d rep %d: H %d n %d ops %d: avg %.2f us per op (slowest heap %.2f), check %lld\n", mode == 1 ? "push" : mode == 2 ? "rmv " : "pop ", mode, rep, H, n, pops,
             sum * 0.01 / H / pops, mx * 0.01 / pops, out[1]);
    }
  return 0;
}
