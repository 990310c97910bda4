// concurrent_fits_native: k NATIVE host threads (std::thread, no interpreter lock between them) fit side by side on one context
// through the C ABI -- what a Rust / C++ host does.  Every fit (3 runs x 150 evaluations, fixed work) is followed by a predict at 8
// points and compared bit for bit with the same fit alone.
//   g++ -O2 -std=c++17 -Iinclude tools/concurrent_fits_native.cpp -o build/concurrent_fits_native -Lhbetune_rs_amd -lhbegp -lpthread -Wl,-rpath,$PWD/hbetune_rs_amd
//   python3 tools/dump_workload.py 128 /tmp/wl.bin && ./build/concurrent_fits_native /tmp/wl.bin 1 4 16
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "hbegp.h"

struct Result {
  double lml;
  std::vector<double> theta, pred;
  bool operator==(const Result& o) const {
    return memcmp(&lml, &o.lml, 8) == 0 && theta.size() == o.theta.size() && memcmp(theta.data(), o.theta.data(), 8 * theta.size()) == 0 &&
           memcmp(pred.data(), o.pred.data(), 8 * pred.size()) == 0;
  }
};

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s workload.bin k [k ...]\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  int hdr[4];
  if (fread(hdr, 4, 4, f) != 4) return 2;
  const int n = hdr[0], d = hdr[1], nr = hdr[2], p = d + 2;
  std::vector<double> X((size_t)n * d), y(n), theta0(p), lo(p), hi(p), starts((size_t)nr * p);
  auto rd = [&](std::vector<double>& v) { return fread(v.data(), 8, v.size(), f) == v.size(); };
  if (!(rd(X) && rd(y) && rd(theta0) && rd(lo) && rd(hi) && rd(starts))) { fprintf(stderr, "short file\n"); return 2; }
  fclose(f);
  hbegp_ctx* ctx = nullptr;
  const int dev0 = 0;
  if (hbegp_ctx_create(1, &dev0, &ctx) != HBEGP_OK) { fprintf(stderr, "ctx: %s\n", hbegp_last_error()); return 1; }
  auto fit = [&](Result& r) {
    hbegp_fit_options opt = HBEGP_FIT_OPTIONS_INIT;
    opt.maxeval = 150;
    opt.fixed_work = 1;
    hbegp_model* m = nullptr;
    r.theta.assign(p, 0.0);
    if (hbegp_fit_f64(ctx, X.data(), y.data(), n, d, 2.5, theta0.data(), lo.data(), hi.data(), starts.data(), nr, &opt, r.theta.data(), &r.lml, &m) != HBEGP_OK) {
      fprintf(stderr, "fit: %s\n", hbegp_last_error());
      exit(1);
    }
    r.pred.assign(16, 0.0);
    int nw = 0;
    if (hbegp_predict_f64(m, X.data(), 8, r.pred.data(), r.pred.data() + 8, &nw) != HBEGP_OK) { fprintf(stderr, "predict: %s\n", hbegp_last_error()); exit(1); }
    hbegp_model_release(m);
  };
  Result solo;
  fit(solo);
  for (int a = 2; a < argc; ++a) {
    const int k = atoi(argv[a]);
    const int reps = n <= 256 ? std::max(8, 24 / k) : std::max(2, 8 / k);
    std::vector<int> bad(k, 0);
    auto work = [&](int i, int r) {
      for (int q = 0; q < r; ++q) {
        Result res;
        fit(res);
        if (!(res == solo)) ++bad[i];
      }
    };
    {  // one round to warm every thread's pools
      std::vector<std::thread> ts;
      for (int i = 0; i < k; ++i) ts.emplace_back(work, i, 1);
      for (auto& t : ts) t.join();
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> ts;
    for (int i = 0; i < k; ++i) ts.emplace_back(work, i, reps);
    for (auto& t : ts) t.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int nbad = 0;
    for (int b : bad) nbad += b;
    printf("native threads, n=%d k=%d: %.2f fits/s aggregate (%.1f ms per round of %d); fits that differ from the solo fit: %d\n", n, k, k * reps / dt,
           dt / reps * 1e3, k, nbad);
    fflush(stdout);
  }
  hbegp_ctx_destroy(ctx);
  return 0;
}
