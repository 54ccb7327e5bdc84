// Read-pattern microbenchmark: the S3 RHS reads doubles [112,400) of every 400-double element record.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// each 32-lane group reads one element's 144 double2 (q + 32k), `ahead` elements in flight per group
template <int AHEAD>
__global__ __launch_bounds__(256) void rd(const double *gp, int64_t nel, double *sink, int skip) {
  const int tid = threadIdx.x, el = tid >> 5, q = tid & 31;
  double2 acc = make_double2(0, 0);
  const int64_t stride = (int64_t)gridDim.x * 8;
  for (int64_t e0 = (int64_t)blockIdx.x * 8 + el; e0 < nel; e0 += stride * AHEAD) {
    double2 v[AHEAD][5];
#pragma unroll
    for (int a = 0; a < AHEAD; a++) {
      const int64_t e = e0 + a * stride;
      if (e < nel) {
        const double2 *src = reinterpret_cast<const double2 *>(gp + e * 400 + skip);
#pragma unroll
        for (int k = 0; k < 5; k++) v[a][k] = (q + 32 * k < (400 - skip) / 2) ? src[q + 32 * k] : make_double2(0, 0);
      } else {
#pragma unroll
        for (int k = 0; k < 5; k++) v[a][k] = make_double2(0, 0);
      }
    }
#pragma unroll
    for (int a = 0; a < AHEAD; a++)
#pragma unroll
      for (int k = 0; k < 5; k++) { acc.x += v[a][k].x; acc.y += v[a][k].y; }
  }
  if (acc.x == 123.456) sink[0] = acc.y;
}
// the RHS kernel's traversal: every wave owns groups of 16 consecutive elements and walks them two elements at a
// time (32 lanes each), two pairs ahead; MODE 0: consecutive waves own consecutive groups (a wave's reads are 6.4 KB
// chunks 51 KB apart from its neighbours'); MODE 1: the 4 waves of a workgroup share 64 consecutive elements and read
// 8 consecutive elements per step (25.6 KB chunks)
template <int MODE>
__global__ __launch_bounds__(256) void coop(const double *gp, int64_t nel, double *sink, double *out = nullptr, int wmode = 0) {
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, el = lane >> 5, q = lane & 31;
  const int qk4 = (q + 128 < 144) ? q + 128 : 143;
  double2 acc = make_double2(0, 0);
  double2 pins[2][5];
  typedef double d2v __attribute__((ext_vector_type(2)));
  auto req = [&](int64_t e, double2 *pin) {
    if (e >= nel) e = nel - 1;
    const double2 *src = reinterpret_cast<const double2 *>(gp + e * 400 + 112);
    if (wmode >= 10) {  // non-temporal loads
      const d2v *sv = reinterpret_cast<const d2v *>(src);
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const d2v v = __builtin_nontemporal_load(&sv[k < 4 ? q + 32 * k : qk4]);
        pin[k] = make_double2(v.x, v.y);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) pin[k] = src[q + 32 * k];
      pin[4] = src[qk4];
    }
  };
  const int64_t ngroups = nel / 16;
  if (MODE == 0) {
    const int64_t gstride = (int64_t)gridDim.x * 4;
    int64_t g = (int64_t)blockIdx.x * 4 + w;
    req(16 * g + el, pins[0]);
    req(16 * g + 2 + el, pins[1]);
    for (; g < ngroups; g += gstride)
#pragma unroll 1
      for (int it2 = 0; it2 < 4; it2++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int it = 2 * it2 + h;
#pragma unroll
          for (int k = 0; k < 5; k++) { acc.x += pins[h][k].x; acc.y += pins[h][k].y; }
          const int64_t en = (it < 6) ? 16 * g + 2 * (it + 2) : 16 * (g + gstride) + 2 * (it - 6);
          req(en + el, pins[h]);
          if (out && it == 7) {  // the element vectors of the group: 3 KB
            if (wmode == 0 || wmode == 10) {      // one contiguous block per group
#pragma unroll
              for (int k = 0; k < 3; k++) *reinterpret_cast<double2 *>(out + g * 384 + (lane + 64 * k) * 2) = acc;
            } else if (wmode == 2) {  // contiguous block, non-temporal
#pragma unroll
              for (int k = 0; k < 3; k++) {
                double *o = out + g * 384 + (lane + 64 * k) * 2;
                __builtin_nontemporal_store(acc.x, o);
                __builtin_nontemporal_store(acc.y, o + 1);
              }
            } else if (wmode >= 20 && wmode <= 23) {  // contiguous block, NT loads, store cache-policy bits
#pragma unroll
              for (int k = 0; k < 3; k++) {
                double *o = out + g * 384 + (lane + 64 * k) * 2;
                d2v av; av.x = acc.x; av.y = acc.y;
                if (wmode == 20) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(o), "v"(av) : "memory");
                if (wmode == 21) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(o), "v"(av) : "memory");
                if (wmode == 22) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(o), "v"(av) : "memory");
                if (wmode == 23) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(o), "v"(av) : "memory");
              }
            } else if (wmode == 3) {  // contiguous block, only every 4th group writes 4x the data (same bytes, burstier)
              if ((g / gstride) % 4 == 3)
#pragma unroll
                for (int k = 0; k < 12; k++) *reinterpret_cast<double2 *>(out + (g / gstride / 4 * gstride + g % gstride) * 1536 + (lane + 64 * k) * 2) = acc;
            } else if (wmode == 1) {  // 24 rows 16 MiB apart (component-major)
#pragma unroll
              for (int k = 0; k < 3; k++) {
                const int idx = lane + 64 * k, t = idx >> 3, part = idx & 7;
                *reinterpret_cast<double2 *>(out + t * (int64_t)2097152 + 16 * g + 2 * part) = acc;
              }
            }
          }
        }
  } else {
    const int64_t bstride = (int64_t)gridDim.x * 64;
    int64_t b = (int64_t)blockIdx.x * 64;
    req(b + 2 * w + el, pins[0]);
    req(b + 8 + 2 * w + el, pins[1]);
    for (; b < nel; b += bstride)
#pragma unroll 1
      for (int it2 = 0; it2 < 4; it2++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int it = 2 * it2 + h;
#pragma unroll
          for (int k = 0; k < 5; k++) { acc.x += pins[h][k].x; acc.y += pins[h][k].y; }
          const int64_t en = (it < 6) ? b + 8 * (it + 2) : b + bstride + 8 * (it - 6);
          req(en + 2 * w + el, pins[h]);
        }
  }
  if (acc.x == 123.456) sink[0] = acc.y;
}
__global__ __launch_bounds__(256) void wr_only(double *out, int64_t ngroups) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double2 acc = make_double2(1.0, 2.0);
  for (int64_t g = (int64_t)blockIdx.x * 4 + w; g < ngroups; g += (int64_t)gridDim.x * 4)
#pragma unroll
    for (int k = 0; k < 3; k++) *reinterpret_cast<double2 *>(out + g * 384 + (lane + 64 * k) * 2) = acc;
}
// the RHS runs kernel's input side alone: one wave per workgroup, WPC per CU, 8 consecutive elements per step as 18
// non-temporal 16-byte loads per lane, one step ahead; LDS = 1: commit to LDS (padded layout) and read one value back
template <int LDS>
__global__ __launch_bounds__(64) void runs_like(const double *gp, int64_t nel, double *sink) {
  __shared__ double2 sh[8 * 148];
  typedef double d2v __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x;
  double2 pin[18];
  double acc = 0.0;
  auto req = [&](int64_t e0) {
    if (e0 > nel - 8) e0 = nel - 8;
    const d2v *blk = reinterpret_cast<const d2v *>(gp + e0 * 400 + 112);
#pragma unroll
    for (int k = 0; k < 18; k++) {
      const int c = 64 * k + lane, xe = c / 144, rr = c - 144 * xe;
      const d2v v = __builtin_nontemporal_load(blk + (xe * 200 + rr));
      pin[k] = make_double2(v.x, v.y);
      asm volatile("" ::: "memory");
    }
  };
  const int64_t nstep = nel / 8;
  int64_t st = blockIdx.x;
  req(8 * st);
  for (; st < nstep; st += gridDim.x) {
    if (LDS) {
#pragma unroll
      for (int k = 0; k < 18; k++) {
        const int c = 64 * k + lane, xe = c / 144, rr = c - 144 * xe;
        sh[xe * 148 + rr] = pin[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 18; k++) acc += pin[k].x + pin[k].y;
    }
    req(8 * (st + gridDim.x));
    if (LDS) {
      __builtin_amdgcn_wave_barrier();
      acc += sh[(lane >> 3) * 148 + (lane & 7)].x;
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (acc == 123.456) sink[0] = acc;
}
// the same with the runs kernel's traversal: task = (chunk of ZC element layers, row, run of 16 elements), a wave
// walks its run through the chunk's layers (two 8-element steps per layer); COORD = 1 adds the per-step coordinate load
template <int ZC, int COORD>
__global__ __launch_bounds__(64) void runs_trav(const double *gp, const double *coords, int mx, int my, int mz, double *sink) {
  __shared__ double2 sh[8 * 148];
  typedef double d2v __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x;
  double2 pin[18];
  double pxc = 0.0, acc = 0.0;
  const uint32_t nr = (mx + 15) / 16, nzc = (mz + ZC - 1) / ZC;
  const int64_t ntasks = (int64_t)nzc * my * nr;
  auto req = [&](int64_t t, int l, int h) {
    if (t >= ntasks) { t = ntasks - 1; l = 0; h = 0; }
    const uint32_t t32 = (uint32_t)t, tt = t32 / nr, r = t32 - tt * nr, ch = tt / (uint32_t)my, ey = tt - ch * (uint32_t)my;
    const int64_t e0 = 16 * (int64_t)r + 8 * h + (int64_t)mx * (ey + (int64_t)my * (ch * ZC + l));
    const d2v *blk = reinterpret_cast<const d2v *>(gp + e0 * 400 + 112);
#pragma unroll
    for (int k = 0; k < 18; k++) {
      const int c = 64 * k + lane, xe = c / 144, rr = c - 144 * xe;
      const d2v v = __builtin_nontemporal_load(blk + (xe * 200 + rr));
      pin[k] = make_double2(v.x, v.y);
      asm volatile("" ::: "memory");
    }
    if (COORD) pxc = coords[3 * ((16 * r + 8 * h + (lane >> 3)) + (int64_t)(mx + 1) * (ey + (int64_t)(my + 1) * (ch * ZC + l))) + (lane & 1)];
  };
  int64_t t = blockIdx.x;
  req(t, 0, 0);
  for (; t < ntasks; t += gridDim.x) {
    const uint32_t ch = (uint32_t)t / nr / (uint32_t)my;
    const int nlz = (mz - (int)ch * ZC < ZC) ? mz - (int)ch * ZC : ZC;
    for (int l = 0; l < nlz; l++)
      for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int k = 0; k < 18; k++) {
          const int c = 64 * k + lane, xe = c / 144, rr = c - 144 * xe;
          sh[xe * 148 + rr] = pin[k];
        }
        acc += pxc;
        int64_t tn = t; int ln = l, hn = h + 1;
        if (hn == 2) { hn = 0; ln = l + 1; }
        if (ln == nlz) { ln = 0; tn = t + gridDim.x; }
        req(tn, ln, hn);
        __builtin_amdgcn_wave_barrier();
        acc += sh[(lane >> 3) * 148 + (lane & 7)].x;
        __builtin_amdgcn_wave_barrier();
      }
  }
  if (acc == 123.456) sink[0] = acc;
}
// plain linear read
__global__ void lin(const double2 *p, int64_t n2, double *sink) {
  double2 acc = make_double2(0, 0);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    const double2 v = p[i]; acc.x += v.x; acc.y += v.y;
  }
  if (acc.x == 123.456) sink[0] = acc.y;
}
template <typename F>
void timeit(const char *name, double bytes, F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-50s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
}
int main() {
  const int64_t nel = 2097152;
  double *gp, *sink; hipMalloc(&gp, nel * 400 * 8); hipMalloc(&sink, 64);
  double *outb; hipMalloc(&outb, (size_t)24 * 2097152 * 8 + 4096);
  hipMemset(gp, 0x3f, nel * 400 * 8);  // non-zero bytes
  const double b288 = nel * 288.0 * 8, b400 = nel * 400.0 * 8;
  for (int wpc : {7, 10, 14}) {
    printf("--- runs-like, %d one-wave workgroups per CU ---\n", wpc);
    timeit("18 NT loads/lane one step ahead, registers only", b288, [&] { runs_like<0><<<256 * wpc, 64>>>(gp, nel, sink); });
    timeit("the same + LDS commit", b288, [&] { runs_like<1><<<256 * wpc, 64>>>(gp, nel, sink); });
  }
  {
    double *co; hipMalloc(&co, (size_t)129 * 129 * 129 * 3 * 8); hipMemset(co, 0, (size_t)129 * 129 * 129 * 3 * 8);
    printf("--- runs traversal (128^3 elements), 7 one-wave workgroups per CU ---\n");
    timeit("chunks of 32 layers", b288, [&] { runs_trav<32, 0><<<256 * 7, 64>>>(gp, co, 128, 128, 128, sink); });
    timeit("chunks of 32 layers + coordinate load", b288, [&] { runs_trav<32, 1><<<256 * 7, 64>>>(gp, co, 128, 128, 128, sink); });
    timeit("chunks of 16 layers", b288, [&] { runs_trav<16, 0><<<256 * 7, 64>>>(gp, co, 128, 128, 128, sink); });
    timeit("chunks of 8 layers", b288, [&] { runs_trav<8, 0><<<256 * 7, 64>>>(gp, co, 128, 128, 128, sink); });
  }
  for (int wg : {3, 8}) {
    printf("--- %d WG/CU ---\n", wg);
    timeit("linear 16B/lane, whole array", b400, [&] { lin<<<256 * wg, 256>>>((const double2 *)gp, nel * 200, sink); });
    timeit("records [112,400), 1 element ahead", b288, [&] { rd<1><<<256 * wg, 256>>>(gp, nel, sink, 112); });
    timeit("records [112,400), 2 ahead", b288, [&] { rd<2><<<256 * wg, 256>>>(gp, nel, sink, 112); });
    timeit("records [112,400), 4 ahead", b288, [&] { rd<4><<<256 * wg, 256>>>(gp, nel, sink, 112); });
    timeit("coop traversal, wave-owned groups of 16", b288, [&] { coop<0><<<256 * wg, 256>>>(gp, nel, sink); });
    timeit("coop traversal + 3 KB block stores per group", b288, [&] { coop<0><<<256 * wg, 256>>>(gp, nel, sink, outb, 0); });
    timeit("coop traversal + block stores, non-temporal", b288, [&] { coop<0><<<256 * wg, 256>>>(gp, nel, sink, outb, 2); });
    timeit("coop traversal + 12 KB stores every 4th group", b288, [&] { coop<0><<<256 * wg, 256>>>(gp, nel, sink, outb, 3); });
    timeit("coop traversal, NT loads, no stores", b288, [&] { coop<0><<<256 * wg, 256>>>(gp, nel, sink, nullptr, 10); });
    timeit("coop traversal, NT loads + block stores", b288, [&] { coop<0><<<256 * wg, 256>>>(gp, nel, sink, outb, 10); });
    timeit("block stores only (bytes = 192/el)", nel * 192.0, [&] { wr_only<<<256 * wg, 256>>>(outb, nel / 16); });
    timeit("NT loads + stores sc0 sc1", b288, [&] { coop<0><<<256 * wg, 256>>>(gp, nel, sink, outb, 20); });
    timeit("NT loads + stores sc1", b288, [&] { coop<0><<<256 * wg, 256>>>(gp, nel, sink, outb, 21); });
    timeit("NT loads + stores sc0 sc1 nt", b288, [&] { coop<0><<<256 * wg, 256>>>(gp, nel, sink, outb, 22); });
    timeit("NT loads + stores nt", b288, [&] { coop<0><<<256 * wg, 256>>>(gp, nel, sink, outb, 23); });
    timeit("coop traversal + component-major stores", b288, [&] { coop<0><<<256 * wg, 256>>>(gp, nel, sink, outb, 1); });
    timeit("coop traversal, workgroup-owned 64, 8 consecutive/step", b288, [&] { coop<1><<<256 * wg, 256>>>(gp, nel, sink); });
    timeit("records [0,400) whole, 2 ahead (bytes=400)", b400, [&] { rd<2><<<256 * wg, 256>>>(gp, nel, sink, 0); });
  }
  return 0;
}
