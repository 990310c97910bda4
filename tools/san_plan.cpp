// san_plan.cpp -- host only: plan builder (both plans, f64/f32 stage depths, 1..40 blocks, four workgroup counts), validator and the
// bounded L-BFGS under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only; tests/test_dag_plan_cpu.py runs it).
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -I csrc -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/san_plan.cpp
#include <cstdio>
#include "dag_plan.hpp"
#include "lbfgsb.hpp"
using namespace hbegp;
int main() {
  int bad = 0;
  for (int nb = 1; nb <= 40; ++nb)
    for (int rl = 0; rl < 2; ++rl)
      for (int bk : {16, 32})
        for (int nwg : {0, 3, 96, 256}) {
          DagBuilder b(bk, rl ? 4 : 8, nwg, true, 1);
          b.set_rl(32, 1, nb % 2 == 0);
          b.set_rl_progressive(rl != 0 && nb % 3 != 0, -1, -1, nb % 2 == 1);  // two thirds of the right-looking plans: row-progressive inverse
          b.set_big128(nb % 4 < 2);
          DagPlan p = b.build(0, nb, true, rl != 0);
          if (p.tasks.empty()) { if (nb >= 2) ++bad; continue; }
          if (nb <= 24 && nwg == 96) { const std::string why = dag_plan_validate(p, nb); if (!why.empty()) { printf("nb=%d rl=%d: %s\n", nb, rl, why.c_str()); ++bad; } }
        }
  // optimiser: slanted plane (gradmin.rs:75-101)
  double x[2] = {0.5, -0.3}, lo[2] = {-2, -2}, hi[2] = {2, 2};
  Objective f = [](const double* v, double* g) { g[0] = 1; g[1] = 1; return v[0] + v[1]; };
  LbfgsOptions o; o.maxeval = 50;
  lbfgsb_minimize(f, x, lo, hi, 2, o);
  if (x[0] != -2 || x[1] != -2) { printf("lbfgs: %g %g\n", x[0], x[1]); ++bad; }
  printf("sanitised plan / optimiser run: %d problems\n", bad);
  return bad != 0;
}
