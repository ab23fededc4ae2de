// Diagnostic: where does a fp16x3 conv workgroup spend its life?  (s_memrealtime stamps, 100 MHz)
#define DS_STAMP 1
#include "../diffsci_amd/csrc/ds_api.hip"
#include "../diffsci_amd/csrc/ds_conv3h.hip"
#include <map>
#include <vector>
#include <algorithm>
int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 64, Cin = argc > 2 ? atoi(argv[2]) : 64, Cout = Cin, S = argc > 3 ? atoi(argv[3]) : 128;
  const bool pre = argc > 4 && atoi(argv[4]) != 0;     // fused norm+SiLU loader and tile statistics
  size_t nin = (size_t)B * Cin * S * S, nout = (size_t)B * Cout * S * S;
  float *in, *out, *w; void* wp;
  hipMalloc(&in, nin * 4); hipMalloc(&out, nout * 4); hipMalloc(&w, (size_t)Cout * Cin * 9 * 4);
  {   // random operands: zero-filled data runs at a higher clock (MI355X_MICROARCH.md, DVFS give-back)
    std::vector<float> hx(nin), hw((size_t)Cout * Cin * 9);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.04f;
    hipMemcpy(in, hx.data(), nin * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  }
  hipMalloc(&wp, ds_conv2d_h3_packed_bytes(Cout, Cin));
  ds_conv2d_h3_pack_weights(wp, w, Cout, Cin, 0, nullptr);
  int blocks = B * (S / 8) * (S / 32) * ((Cout + 63) / 64);
  float *tab = nullptr, *stats = nullptr;
  if (pre) {
    const size_t nt = (size_t)B * ((Cin + 15) / 16 * 16) * 4;
    std::vector<float> ht(nt, 0.f);
    for (size_t i = 1; i < nt; i += 4) ht[i] = 1.f;                       // (M, A, C) = (0, 1, 0)
    hipMalloc(&tab, nt * 4); hipMemcpy(tab, ht.data(), nt * 4, hipMemcpyHostToDevice);
    hipMalloc(&stats, (size_t)B * Cout * ((S + 7) / 8 * ((S + 31) / 32)) * 16);
  }
  hipMalloc(&g_stamps, (size_t)blocks * 16 * 8);
  const int reps = argc > 5 ? atoi(argv[5]) : 200;    // sustained launches: the clock settles under load
  for (int it = 0; it < reps; ++it) ds_conv2d_h3(out, in, wp, 0, nullptr, nullptr, 0, nullptr, nullptr, B, Cin, Cout, S, S, 0, tab, stats, nullptr);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)blocks * 16);
  hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int b = 0; b < blocks; ++b) { t0 = std::min(t0, h[b * 16]); t1 = std::max(t1, h[b * 16 + 5]); }
  double seg[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < blocks; ++b) for (int k = 0; k < 5; ++k) seg[k] += (double)(h[b * 16 + k + 1] - h[b * 16 + k]);
  printf("B=%d C=%d S=%d pre=%d blocks=%d: kernel span %.1f us\n", B, Cin, S, (int)pre, blocks, (t1 - t0) / 100.0);
  const char* names[5] = {"plan+issue loads", "x_store+barrier (load latency)", "main loop", "epilogue issue", "store drain"};
  for (int k = 0; k < 5; ++k) printf("  %-32s avg %.2f us\n", names[k], seg[k] / blocks / 100.0);
  {
    std::vector<double> clk;
    for (int b = 0; b < blocks; ++b) {
      const double dr = (double)(h[b * 16 + 3] - h[b * 16 + 2]), dc = (double)(h[b * 16 + 7] - h[b * 16 + 6]);
      if (dr > 0) clk.push_back(dc / dr * 100.0);
    }
    std::sort(clk.begin(), clk.end());
    printf("  in-kernel clock over the main loop: median %.0f MHz (p10 %.0f, p90 %.0f)\n", clk[clk.size() / 2], clk[clk.size() / 10], clk[clk.size() * 9 / 10]);
  }
  {
    // co-residency: group workgroups by physical CU (XCC id, SE / SH / CU fields of HW_ID) and measure how much of the
    // main-loop time of a CU is spent with two workgroups in their main loops at once (lockstep) vs one alone
    struct Ev { unsigned long long t; int d; };
    std::map<unsigned long long, std::vector<Ev>> cus;
    for (int b = 0; b < blocks; ++b) {
      const unsigned long long hw = h[b * 16 + 8], xcc = h[b * 16 + 9] & 0xf;
      const unsigned long long key = (xcc << 32) | (hw & 0xff00ull) | ((hw >> 13) & 0x7) << 16;   // CU_ID[11:8], SH_ID[12], SE_ID[15:13]
      cus[key].push_back({h[b * 16 + 2], +1});
      cus[key].push_back({h[b * 16 + 3], -1});
    }
    double one = 0, two = 0, span = 0; size_t maxw = 0;
    for (auto& kv : cus) {
      auto& ev = kv.second;
      std::sort(ev.begin(), ev.end(), [](const Ev& x, const Ev& y) { return x.t < y.t || (x.t == y.t && x.d < y.d); });
      int n = 0; unsigned long long last = ev[0].t;
      for (auto& e : ev) { if (n == 1) one += e.t - last; if (n >= 2) two += e.t - last; n += e.d; last = e.t; }
      span += ev.back().t - ev.front().t; maxw = std::max(maxw, ev.size() / 2);
    }
    printf("  %zu CUs seen (%zu workgroups on the busiest); per CU: main loop alone %.1f us, two at once %.1f us, neither %.1f us\n",
           cus.size(), maxw, one / cus.size() / 100.0, two / cus.size() / 100.0, (span - one - two) / cus.size() / 100.0);
  }
  // start-time distribution
  std::vector<double> st; for (int b = 0; b < blocks; ++b) st.push_back((h[b * 16] - t0) / 100.0);
  std::sort(st.begin(), st.end());
  printf("  WG start times: p0 %.1f p25 %.1f p50 %.1f p75 %.1f p100 %.1f us\n", st[0], st[blocks / 4], st[blocks / 2], st[3 * blocks / 4], st[blocks - 1]);
  return 0;
}
