// Diagnostic: who waits for whom in the persistent producer / consumer convolution (ds_conv3p.hip)?  Wave 0 (a consumer) and wave 4
// (a producer) of every workgroup stamp s_memtime when they ARRIVE at each of the first 63 barriers; a barrier opens when the later
// of the two arrives (the other six waves are not stamped), so per step: busy time of each role = arrival - previous opening.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fno-slp-vectorize tools/conv3p_stamp.hip -o tools/bin/conv3p_stamp
//   tools/bin/conv3p_stamp [B] [C] [S] [residual 0/1] [reps]
#define DS_STAMP 1
#include "../diffsci_amd/csrc/ds_api.hip"
#include "../diffsci_amd/csrc/ds_conv3h.hip"
#include "../diffsci_amd/csrc/ds_conv3p.hip"
#include <algorithm>
#include <vector>
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, C = argc > 2 ? atoi(argv[2]) : 64, S = argc > 3 ? atoi(argv[3]) : 128;
  const bool res = argc > 4 && atoi(argv[4]) != 0;
  const int reps = argc > 5 ? atoi(argv[5]) : 100;
  const size_t n = (size_t)B * C * S * S;
  float *in, *out, *w, *r1 = nullptr, *tab, *stats, *shift; void* wp; unsigned* oa;
  hipMalloc(&in, n * 4); hipMalloc(&out, n * 4); hipMalloc(&w, (size_t)C * C * 9 * 4); hipMalloc(&shift, (size_t)B * C * 4);
  {
    std::vector<float> hx(n), hw((size_t)C * C * 9);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f; };
    for (auto& v : hx) v = rnd() * 2.f;
    for (auto& v : hw) v = rnd() * 0.04f;
    hipMemcpy(in, hx.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    if (res) { hipMalloc(&r1, n * 4); hipMemcpy(r1, hx.data(), n * 4, hipMemcpyHostToDevice); }
    hipMemset(shift, 0, (size_t)B * C * 4);
  }
  hipMalloc(&wp, ds_conv2d_h3_packed_bytes(C, C));
  ds_conv2d_h3_pack_weights(wp, w, C, C, 0, nullptr);
  {
    const size_t nt = (size_t)B * ((C + 15) / 16 * 16) * 4;
    std::vector<float> ht(nt, 0.f);
    for (size_t i = 0; i < nt; i += 4) { ht[i + 1] = 1.f; ht[i + 3] = 0.125f; }
    hipMalloc(&tab, nt * 4); hipMemcpy(tab, ht.data(), nt * 4, hipMemcpyHostToDevice);
    hipMalloc(&stats, (size_t)B * C * (S / 8) * (S / 32) * 16);
    hipMalloc(&oa, B * 4); hipMemset(oa, 0, B * 4);
  }
  const int wgs = 256, SL = 64;
  hipMalloc(&g_stamps, (size_t)wgs * 2 * SL * 8);
  hipMemset(g_stamps, 0, (size_t)wgs * 2 * SL * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 20; ++it) ds_conv2d_h3(out, in, wp, 0, nullptr, shift, C, r1, nullptr, B, C, C, S, S, 0, tab, stats, nullptr, res ? oa : nullptr, nullptr);
  hipEventRecord(e0);
  for (int it = 0; it < reps; ++it) ds_conv2d_h3(out, in, wp, 0, nullptr, shift, C, r1, nullptr, B, C, C, S, S, 0, tab, stats, nullptr, res ? oa : nullptr, nullptr);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  printf("B=%d C=%d S=%d residual=%d: %.1f us per launch (stamped build)\n", B, C, S, (int)res, ms / reps * 1e3);
  std::vector<unsigned long long> h((size_t)wgs * 2 * SL);
  hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost);
  const int n_steps = 3 * (C / 16);
  printf("step: consumer busy / producer busy / step length (cycles, mean over workgroups); item boundary every %d steps\n", n_steps);
  double tot_c = 0, tot_p = 0, tot_l = 0; int cnt = 0;
  for (int k = 1; k < 48; ++k) {
    double sc = 0, sp = 0, sl = 0; int m = 0;
    for (int g = 0; g < wgs; ++g) {
      const unsigned long long* c = &h[((size_t)g * 2 + 0) * SL];
      const unsigned long long* p = &h[((size_t)g * 2 + 1) * SL];
      if (!c[k] || !p[k] || !c[k - 1] || !p[k - 1]) continue;
      const unsigned long long open = std::max(c[k - 1], p[k - 1]), next = std::max(c[k], p[k]);
      sc += (double)(c[k] - open); sp += (double)(p[k] - open); sl += (double)(next - open); ++m;
    }
    if (!m) continue;
    printf("  %2d (%s%d) %7.0f %7.0f %7.0f\n", k, ((k - 1) % 6) < 3 ? "E," : "O,", (k - 1) % 3, sc / m, sp / m, sl / m);
    if (k > n_steps) { tot_c += sc / m; tot_p += sp / m; tot_l += sl / m; ++cnt; }
  }
  {
    // fine stamps of producer wave 4 inside steps (O,0) and (O,1) of the second item's last chunk pair (slots 48 ..)
    const char* names[8] = {"(O,0) start", "(O,0) DMA issued", "(O,0) 24 loads issued", "", "(O,1) start", "(O,1) DMA issued", "(O,1) activated + split", ""};
    for (int k = 1; k < 8; ++k) {
      if (k == 3 || k == 4 || k == 7) continue;
      double sum = 0; int m = 0;
      for (int g = 0; g < wgs; ++g) {
        const unsigned long long* p = &h[((size_t)g * 2 + 1) * SL];
        if (p[48 + k] && p[48 + k - 1]) { sum += (double)(p[48 + k] - p[48 + k - 1]); ++m; }
      }
      if (m) printf("  fine: %-26s +%6.0f cycles\n", names[k], sum / m);
    }
  }
  if (cnt) printf("steady state (steps past the first item): consumer %.0f  producer %.0f  step %.0f cycles; MFMA cycles per step (4 x 4 x 3 x 1.5 pairs x 16) = 1152\n", tot_c / cnt, tot_p / cnt, tot_l / cnt);
  return 0;
}
