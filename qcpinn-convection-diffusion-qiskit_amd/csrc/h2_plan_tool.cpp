// Build-time tool (host only, plain g++): runs the HBM-family planner (qc_hbm2_plan.h) on one gate program and prints
// the plan as a constexpr record for the compile-time stage kernels (qc_circuit_h2s_kernels.h).  Driven by
// gen_static.py:  h2_plan_tool <struct name> <n_qubits> <absorb> <rb>  < rows, one "op ba bb slot" per line (device
// encoding: bit positions).  Also prints the flat description (h2_describe) the library compares with the plan it
// builds at run time before it trusts the generated kernels.
#include "qc_hbm2_plan.h"

#include <cstdio>
#include <cstdlib>
#include <string>

int main(int argc, char** argv) {
  if (argc != 5) return 2;
  const std::string name = argv[1];
  const int n = atoi(argv[2]), absorb = atoi(argv[3]), rb = atoi(argv[4]);
  std::vector<QcGate> g;
  QcGate x;
  while (scanf("%d %d %d %d", &x.op, &x.ba, &x.bb, &x.slot) == 4) g.push_back(x);
  const H2Plan P = h2_make_plan(g.data(), (int)g.size(), n, absorb, rb);
  printf("struct %s {\n", name.c_str());
  printf("  static constexpr int N = %d, RB = %d, ABSORB = %d, NSTAGES = %d, NROUNDS = %d, NGATES = %d, NTABLES = %d;\n", n,
         P.rbits, absorb, (int)P.stages.size(), (int)P.rounds.size(), (int)P.gates.size(), (int)P.tables.size());
  printf("  static constexpr H2Stage stages[%d] = {\n", (int)P.stages.size());
  for (const H2Stage& s : P.stages) {
    printf("    {%d, %d, {", s.nloc, s.ngb);
    for (int j = 0; j < H2_T; ++j) printf("%d%s", j < s.nloc ? s.lb[j] : 0, j + 1 < H2_T ? ", " : "");
    printf("}, {");
    for (int j = 0; j < 24; ++j) printf("%d%s", j < s.ngb ? s.gb[j] : 0, j + 1 < 24 ? ", " : "");
    printf("}, %d, %d, %d, %d, {%d, %d}, {%d, %d}, {", s.r0, s.nr, s.np, s.ntab, s.ntab > 0 ? s.tab[0] : 0, s.ntab > 1 ? s.tab[1] : 0,
           s.ntab > 0 ? s.tab_round[0] : 0, s.ntab > 1 ? s.tab_round[1] : 0);
    for (int j = 0; j < 24; ++j) printf("%d%s", s.where[j], j + 1 < 24 ? ", " : "");
    printf("}},\n");
  }
  printf("  };\n");
  const int nr = (int)P.rounds.size() > 0 ? (int)P.rounds.size() : 1;
  printf("  static constexpr H2Round rounds[%d] = {\n", nr);
  for (const H2Round& r : P.rounds)
    printf("    {%d, %d, {%d, %d, %d, %d}, %d, %d, %d, %d, %d, %d, %d, %d},\n", r.kind, r.nrb, r.rb[0], r.rb[1], r.rb[2], r.rb[3], r.g0, r.ng,
           r.table, r.tslot, r.tab_pre, r.ts_pre, r.tab_post, r.ts_post);
  if (P.rounds.empty()) printf("    {},\n");
  printf("  };\n");
  const int ng = (int)P.gates.size() > 0 ? (int)P.gates.size() : 1;
  printf("  static constexpr H2Gate gates[%d] = {\n", ng);
  for (const H2Gate& h : P.gates)
    printf("    {%d, %d, %d, %d, %d, %d, %d, %d, %d},\n", h.op, h.kind, h.tq, h.cq, h.tbit, h.cbit, h.gi, h.slot, h.pidx);
  if (P.gates.empty()) printf("    {},\n");
  printf("  };\n};\n");
  const std::vector<int> d = h2_describe(P);
  printf("static const int %s_describe[%d] = {", name.c_str(), (int)d.size());
  for (size_t i = 0; i < d.size(); ++i) printf("%d%s", d[i], i + 1 < d.size() ? ", " : "");
  printf("};\n");
  return 0;
}
