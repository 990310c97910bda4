// plan_hist.cpp -- host only: task mix of the right-looking task-queue plan (kind x contraction depth), where the per-task
// fixed cost lands.   g++ -O2 -std=c++17 -I csrc -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/plan_hist.cpp -o build/tools/plan_hist
#include <cstdio>
#include <map>
#include "dag_plan.hpp"
using namespace hbegp;
int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 32, nwg = argc > 2 ? atoi(argv[2]) : 96, bk = argc > 3 ? atoi(argv[3]) : 16;
  DagBuilder b(bk, 4, nwg, true, 1);
  b.set_rl(32, 1, false);
  DagPlan p = b.build(0, nb, true, true);
  std::map<std::pair<int, int>, std::pair<int, double>> h;
  double tot = 0;
  for (const DagTask& t : p.tasks) {
    auto& e = h[{t.kind, t.kind == DAG_LEAF ? 0 : t.kend - t.kbeg}];
    e.first++;
    e.second += t.cost * 0.1;
    tot += t.cost * 0.1;
  }
  printf("nb=%d tasks=%zu counters=%zu gflop=%.2f crit=%.0f us sim(%d wg)=%.0f us total cost %.0f us\n", nb, p.tasks.size(), p.totals.size(), p.gflop, p.crit_us, nwg, p.sim_us, tot);
  for (auto& kv : h) printf("  kind %d depth %5d: %5d tasks, %8.0f us (%.1f %%), fixed 4.4 us share of these tasks %.0f %%\n", kv.first.first, kv.first.second, kv.second.first,
                            kv.second.second, 100 * kv.second.second / tot, 100 * 4.4 * kv.second.first / kv.second.second);
}
