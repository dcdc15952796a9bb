// Execution plan of the HBM-resident circuit family (9 <= n <= 20 qubits), second generation.
//
// A statevector of 2^n complex64 amplitudes (512 KiB at n = 16) lives in HBM.  The gate program is cut ONCE, on
// the host, into STAGES; one stage = one pass of every statevector through LDS:
//   * a stage owns a LOCAL set of <= 12 index bits (always containing the low 4 bits, so a tile moves as >= 128-byte
//     runs); a block loads one TILE (2^nloc amplitudes: local bits vary, the others are fixed), runs the stage on
//     it, stores it back;
//   * inside a stage the gates are grouped into ROUNDS.  A round names <= 4 REGISTER bits among the local bits:
//     every thread pulls the 2^4 amplitudes spanned by them from LDS into VGPRs, applies all gates of the round
//     there (targets on register bits; controls on any local or non-local bit), and writes them back - one LDS
//     round trip per round, not per gate;
//   * gates are scheduled by DEPENDENCY, not program order: gates on disjoint bits commute, diagonal gates
//     (RZ, CRZ) commute with each other, so e.g. a layer RX(w) on all 16 wires splits into "the 12 local ones now,
//     the other 4 in the next stage" and cross_mesh (n = 16) runs in TWO stages;
//   * a long run of diagonal gates is ONE table of 2^n unit phases (rebuilt when the parameters change), applied
//     as an element-wise multiply inside a stage - diagonal gates need no locality at all; short runs ride along
//     in a round as phase multiplies.  Gradients of a table's gates come from the Walsh-Hadamard coefficients of
//     t[k] = sum_channels Im(conj(lam_k) chi_k) (weights 0, 1, 2 only), formed per tile in LDS.
// The plan is plain data (no HIP types) so that it can be inspected from tests (qc_hbm_plan_describe).
#pragma once
#include "qc_types.h"

#include <algorithm>
#include <vector>

constexpr int H2_T = 12;        // local bits per tile (32 KiB of complex64)
constexpr int H2_LOW = 4;       // low bits that are always local
constexpr int H2_MAXP = 64;     // parametric in-round gates per stage (LDS accumulator rows)
constexpr int H2_MAXTAB = 2;    // diagonal tables per stage (register accumulators of t)
constexpr int H2_TABLE_MIN = 8; // shortest diagonal run that becomes a table (shorter runs ride in rounds as phase gates)
constexpr int H2_MAXRG = 24;    // gates per round (parametric ones need a partial-sum register each: <= 8 of them)
constexpr int H2_MAXRP = 8;

enum { H2_ROUND_GATES = 0, H2_ROUND_TABLE = 1 };
// how a gate of a round is executed
enum { H2_K_REG1 = 0,    // one-bit gate, target = register bit tq
       H2_K_REG2 = 1,    // controlled gate, control and target both register bits (cq, tq)
       H2_K_PRED = 2,    // controlled gate, target = register bit tq, control = lane or non-local bit cbit
       H2_K_PHASE = 3,   // diagonal gate (RZ / CRZ) whose target is not a register bit: phase multiply by index bits
       H2_K_U4 = 4 };    // fixed two-wire unitary on register bits (tq = high, cq = low)

struct H2Gate {
  int op;          // QcOp
  int kind;        // H2_K_*
  int tq, cq;      // register indices (or -1)
  int tbit, cbit;  // global bit positions of target / control (U4: high / low), -1 if none
  int gi;          // index into the trig table (= position in the program)
  int slot;        // parameter slot, U4 slot, or -1
  int pidx;        // index among the stage's parametric in-round gates (program order), or -1
};

struct H2Round {
  int kind;        // H2_ROUND_*
  int nrb;         // register bits of this round (gate rounds)
  int rb[4];       // their LOCAL positions, ascending
  int g0, ng;      // gate range in the plan's gate array
  int table;       // stand-alone table rounds: global table index
  int tslot;       // stand-alone table rounds: which of the stage's t accumulators it owns
  // gate rounds: a diagonal table applied to the amplitudes as they are loaded (pre) / before they are stored (post),
  // in the round's own register mapping - no LDS round trip of its own; -1 = none
  int tab_pre, ts_pre, tab_post, ts_post;
};

struct H2Stage {
  int nloc, ngb;
  int lb[H2_T];    // global bit of local position j, ascending
  int gb[24];      // non-local bits, ascending
  int r0, nr;      // rounds
  int np;          // parametric in-round gates
  int ntab;        // table rounds
  int tab[H2_MAXTAB];
  int tab_round[H2_MAXTAB];   // round (index within the stage) whose amplitude mapping the table's t accumulator uses
  int where[24];   // index bit b: its local position, or -(1 + j) when it is the non-local bit gb[j]
};

struct H2DiagGate { int op, bt, bc, gi, slot; };   // diagonal gate of a table, global bit numbering
struct H2Table { int g0, ng; };

struct H2Plan {
  int n = 0, absorb = 0, rbits = 0;
  std::vector<H2Stage> stages;
  std::vector<H2Round> rounds;
  std::vector<H2Gate> gates;
  std::vector<H2Table> tables;
  std::vector<H2DiagGate> dgates;
  std::vector<int> sparse_idx;   // local indices (nloc bits of the FIRST stage) of weight <= 3, ascending by (weight, value)
};

inline bool h2_is_diag(int op) { return op == QC_RZ || op == QC_CRZ; }
inline bool h2_is_param(int op) { return op == QC_RX || op == QC_RY || op == QC_RZ || op == QC_CRX || op == QC_CRZ; }

// Builds the plan for gates [g_first, n_gates) of the program (g_first = n when the leading RX layer is folded
// into the embedding angles).
inline H2Plan h2_make_plan(const QcGate* gates, int n_gates, int n, int absorb, int rb12 = 4) {
  H2Plan P;
  P.n = n;
  P.absorb = absorb;
  // tile bits T and register bits per round RB with T % RB == 0 (the register groups partition the local positions):
  // n >= 12: T = 12, RB = 4 (256 threads) or 3 (512 threads); n = 10, 11: T = 10, RB = 2 (256 threads); n = 9: T = 9,
  // RB = 3 (64 threads)
  const int T = n >= H2_T ? H2_T : (n >= 10 ? 10 : n);
  const int LOW = T < H2_LOW ? T : H2_LOW;
  // (T = 12 also runs as RB = 3 with 512 threads: half the registers per thread, four waves per SIMD; rb12 picks)
  P.rbits = T == 12 ? (rb12 == 3 ? 3 : 4) : (T == 10 ? 2 : 3);
  const int g_first = absorb ? n : 0;
  const int G = n_gates - g_first;
  auto gate = [&](int j) -> const QcGate& { return gates[g_first + j]; };
  auto bits_of = [&](int j, int (&b)[2]) {
    const QcGate& g = gate(j);
    b[0] = g.ba;
    b[1] = g.bb;
    return g.bb >= 0 ? 2 : 1;
  };
  // dependencies: an earlier gate sharing a bit, unless both are diagonal
  std::vector<std::vector<int>> pred(G);
  for (int j = 0; j < G; ++j) {
    int bj[2];
    const int nj = bits_of(j, bj);
    for (int i = 0; i < j; ++i) {
      if (h2_is_diag(gate(i).op) && h2_is_diag(gate(j).op)) continue;
      int bi[2];
      const int ni = bits_of(i, bi);
      bool share = false;
      for (int a = 0; a < ni; ++a)
        for (int b = 0; b < nj; ++b) share = share || bi[a] == bj[b];
      if (share) pred[j].push_back(i);
    }
  }
  // candidate order of the non-diagonal gates: by the highest bit that must be tile-local, then by program position
  std::vector<int> low_first(G);
  {
    auto key = [&](int j) {
      const QcGate& g = gate(j);
      if (g.op == QC_U4) return g.ba > g.bb ? g.ba : g.bb;
      if (g.op == QC_CNOT || g.op == QC_CRX) return g.bb;
      return g.ba;
    };
    for (int j = 0; j < G; ++j) low_first[j] = j;
    std::stable_sort(low_first.begin(), low_first.end(), [&](int a, int b) { return key(a) < key(b); });
  }
  std::vector<int> state(G, 0);   // 0 = waiting, 1 = in the current stage, 2 = done
  int n_done = 0;
  while (n_done < G) {
    // ---- choose the stage's items: alternately every ready diagonal gate (no locality needed), then every ready
    // non-diagonal gate whose bits still fit the tile
    std::vector<int> need;   // bits >= LOW that must be local
    struct Item { int kind; std::vector<int> g; };   // kind 0: one gate, 1: diagonal run
    std::vector<Item> items;
    int np = 0, ntab = 0;
    auto ready = [&](int j) {
      if (state[j] != 0) return false;
      for (int i : pred[j])
        if (state[i] == 0) return false;
      return true;
    };
    bool progress = true;
    while (progress) {
      progress = false;
      // (1) diagonal gates
      std::vector<int> run;
      for (int j = 0; j < G; ++j)
        if (h2_is_diag(gate(j).op) && ready(j)) run.push_back(j);
      if (!run.empty()) {
        // a table pays for runs with controlled phases (one multiply instead of one per gate); a run of plain RZ gates
        // rides in the rounds that hold their bits in registers
        bool has_ctl = false;
        for (int j : run) has_ctl = has_ctl || gate(j).op == QC_CRZ;
        const bool as_table = (int)run.size() >= H2_TABLE_MIN && has_ctl;
        bool ok = true;
        if (as_table) {
          ok = ntab < H2_MAXTAB;
        } else {
          ok = np + (int)run.size() <= H2_MAXP;
        }
        if (ok) {
          if (as_table) {
            items.push_back({1, run});
            ++ntab;
          } else {
            for (int j : run) items.push_back({0, {j}});
            np += (int)run.size();
          }
          for (int j : run) state[j] = 1;
          progress = true;
        }
      }
      // (2) non-diagonal gates, LOW target bits first (then program order), repeated until nothing more fits: the
      // stage that runs most of a layer then owns the low index bits (a tile = one contiguous run of HBM), and the
      // bits left to the next stage are high ones - its register bits - so that its lanes, too, walk contiguous memory
      bool any = true;
      while (any) {
        any = false;
        for (int jj = 0; jj < G; ++jj) {
          const int j = low_first[jj];
          if (h2_is_diag(gate(j).op) || !ready(j)) continue;
          // bits that must be local: the target (a control may sit anywhere), both bits of a two-wire unitary
          int b[2];
          int nb = 1;
          const QcGate& gj = gate(j);
          if (gj.op == QC_U4) {
            b[0] = gj.ba;
            b[1] = gj.bb;
            nb = 2;
          } else if (gj.op == QC_CNOT || gj.op == QC_CRX) {
            b[0] = gj.bb;
          } else {
            b[0] = gj.ba;
          }
          std::vector<int> add;
          for (int a = 0; a < nb; ++a)
            if (b[a] >= LOW && std::find(need.begin(), need.end(), b[a]) == need.end() &&
                std::find(add.begin(), add.end(), b[a]) == add.end())
              add.push_back(b[a]);
          const bool par = h2_is_param(gate(j).op);
          if ((int)(need.size() + add.size()) > T - LOW || (par && np >= H2_MAXP)) continue;
          for (int v : add) need.push_back(v);
          items.push_back({0, {j}});
          if (par) ++np;
          state[j] = 1;
          any = true;
          progress = true;
        }
      }
    }
    // ---- local set: the low bits, the needed bits, then the lowest unused bits
    std::vector<int> loc;
    for (int b = 0; b < LOW; ++b) loc.push_back(b);
    for (int b : need) loc.push_back(b);
    for (int b = LOW; b < n && (int)loc.size() < T; ++b)
      if (std::find(loc.begin(), loc.end(), b) == loc.end()) loc.push_back(b);
    std::sort(loc.begin(), loc.end());
    H2Stage sd = {};
    sd.nloc = (int)loc.size();
    for (int j = 0; j < sd.nloc; ++j) sd.lb[j] = loc[j];
    for (int b = 0; b < n; ++b)
      if (std::find(loc.begin(), loc.end(), b) == loc.end()) sd.gb[sd.ngb++] = b;
    auto pos = [&](int b) {
      auto it = std::find(loc.begin(), loc.end(), b);
      return it == loc.end() ? -1 : (int)(it - loc.begin());
    };
    // ---- rounds: list scheduling.  Standard register groups: positions [k RB, k RB + RB); the last one overlaps if
    // nloc % RB != 0.  A gate joins the EARLIEST round that (a) comes no earlier than its predecessors and (b) holds its
    // target (a two-wire unitary: both bits) among the register bits - gates that do not depend on each other share a
    // round whatever their distance in the program (e.g. RX(b), RZ(b), then the fixed unitary on (b, b')); only when no
    // such round exists a new one is opened at the end.  Positions in time: 3 r + {0: table applied as round r loads,
    // 1: gates of round r, 2: table applied before round r stores}.
    const int RB = P.rbits;
    auto group_of = [&](int p0, int p1, int (&rb)[4]) {   // a register set containing positions p0 (and p1 >= 0)
      const int ngrp = (sd.nloc + RB - 1) / RB;
      for (int k = 0; k < ngrp; ++k) {
        int lo = k * RB;
        if (lo + RB > sd.nloc) lo = sd.nloc - RB;
        if (p0 >= lo && p0 < lo + RB && (p1 < 0 || (p1 >= lo && p1 < lo + RB))) {
          for (int q = 0; q < RB; ++q) rb[q] = lo + q;
          return;
        }
      }
      // no standard group holds both: p0, p1 and other positions, those above the always-local low bits first (a round
      // whose register bits all lie above them can read / write HBM in its own mapping)
      std::vector<int> s = {p0, p1};
      for (int p = LOW; p < sd.nloc && (int)s.size() < RB; ++p)
        if (p != p0 && p != p1) s.push_back(p);
      for (int p = 0; p < LOW && p < sd.nloc && (int)s.size() < RB; ++p)
        if (p != p0 && p != p1) s.push_back(p);
      std::sort(s.begin(), s.end());
      for (int q = 0; q < RB; ++q) rb[q] = s[q];
    };
    struct RoundB {
      H2Round r;
      std::vector<H2Gate> g;
      std::vector<int> gid;          // program index (relative to g_first) of every gate of the round
      std::vector<int> tab_gid_pre, tab_gid_post;
      int np;
    };
    std::vector<RoundB> RS;
    std::vector<int> when(G, -1);    // time of every gate scheduled in this stage
    sd.np = 0;
    sd.ntab = 0;
    auto in_rb = [&](const H2Round& r, int p) {
      for (int q = 0; q < r.nrb; ++q)
        if (r.rb[q] == p) return q;
      return -1;
    };
    auto tmin_of = [&](const std::vector<int>& js) {
      int t = -1;
      for (int j : js)
        for (int i : pred[j])
          if (state[i] == 1 && when[i] > t) t = when[i];
      return t;
    };
    auto new_round = [&](const int (&want)[4]) {
      RoundB nb;
      nb.r = {};
      nb.r.kind = H2_ROUND_GATES;
      nb.r.nrb = RB;
      for (int q = 0; q < RB; ++q) nb.r.rb[q] = want[q];
      nb.r.table = nb.r.tslot = -1;
      nb.r.tab_pre = nb.r.ts_pre = nb.r.tab_post = nb.r.ts_post = -1;
      nb.np = 0;
      RS.push_back(nb);
      return (int)RS.size() - 1;
    };
    int pend_tab = -1, pend_ts = -1;   // a table without predecessors, waiting for the stage's first round (its tab_pre)
    std::vector<int> pend_gid;
    for (const Item& it : items) {
      if (it.kind == 1) {
        const int table = (int)P.tables.size(), tslot = sd.ntab;
        sd.tab[sd.ntab++] = table;
        H2Table tb = {(int)P.dgates.size(), (int)it.g.size()};
        for (int j : it.g) {
          const QcGate& g = gate(j);
          const bool ctl = g.op == QC_CRZ;
          P.dgates.push_back({g.op, ctl ? g.bb : g.ba, ctl ? g.ba : -1, g_first + j, g.slot});
        }
        P.tables.push_back(tb);
        const int tmin = tmin_of(it.g);
        int tnow;
        if (RS.empty() && pend_tab < 0) {
          pend_tab = table;
          pend_ts = tslot;
          pend_gid = it.g;
          tnow = 0;
        } else if (!RS.empty() && RS.back().r.kind == H2_ROUND_GATES && RS.back().r.tab_post < 0 &&
                   tmin <= 3 * ((int)RS.size() - 1) + 1) {
          RS.back().r.tab_post = table;
          RS.back().r.ts_post = tslot;
          RS.back().tab_gid_post = it.g;
          tnow = 3 * ((int)RS.size() - 1) + 2;
        } else {   // a round of its own (element-wise pass over the tile in LDS)
          RoundB nb;
          nb.r = {};
          nb.r.kind = H2_ROUND_TABLE;
          nb.r.table = table;
          nb.r.tslot = tslot;
          nb.r.tab_pre = nb.r.ts_pre = nb.r.tab_post = nb.r.ts_post = -1;
          nb.np = 0;
          nb.gid = it.g;
          RS.push_back(nb);
          tnow = 3 * ((int)RS.size() - 1) + 1;
        }
        for (int j : it.g) when[j] = tnow;
        continue;
      }
      const int j = it.g[0];
      const QcGate& g = gate(j);
      H2Gate hg = {};
      hg.op = g.op;
      hg.gi = g_first + j;
      hg.slot = g.slot;
      hg.tq = hg.cq = -1;
      hg.tbit = hg.cbit = -1;
      hg.pidx = -1;
      const bool ctl = (g.op == QC_CNOT || g.op == QC_CRX || g.op == QC_CRZ);
      if (g.op == QC_U4) {
        hg.tbit = g.ba;   // high bit of the 4x4 index
        hg.cbit = g.bb;
      } else if (ctl) {
        hg.cbit = g.ba;
        hg.tbit = g.bb;
      } else {
        hg.tbit = g.ba;
      }
      const int tp = pos(hg.tbit), cp = hg.cbit >= 0 ? pos(hg.cbit) : -1;
      const bool par = h2_is_param(g.op);
      const bool diag = h2_is_diag(g.op);
      const int tmin = tmin_of(it.g);
      const int rmin = tmin <= 1 ? 0 : (tmin + 1) / 3;   // smallest r with 3 r + 1 >= tmin
      auto room = [&](const RoundB& rbk) {
        return rbk.r.kind == H2_ROUND_GATES && (int)rbk.g.size() < H2_MAXRG && !(par && rbk.np >= H2_MAXRP);
      };
      int at = -1;
      for (int r = rmin; r < (int)RS.size() && at < 0; ++r) {
        if (!room(RS[r])) continue;
        const bool t_in = tp >= 0 && in_rb(RS[r].r, tp) >= 0, c_in = cp >= 0 && in_rb(RS[r].r, cp) >= 0;
        if (g.op == QC_U4 ? (t_in && c_in) : (diag && ctl ? (t_in && c_in) : t_in)) at = r;
      }
      if (at < 0 && diag)   // a diagonal gate rides in any round as a phase multiply by index bits
        for (int r = rmin; r < (int)RS.size() && at < 0; ++r)
          if (room(RS[r])) at = r;
      if (at < 0) {
        int want[4] = {-1, -1, -1, -1};
        if (g.op == QC_U4) group_of(tp, cp, want);
        else group_of(tp >= 0 ? tp : 0, -1, want);
        at = new_round(want);
      }
      RoundB& rbk = RS[at];
      const int tq = tp >= 0 ? in_rb(rbk.r, tp) : -1, cq = cp >= 0 ? in_rb(rbk.r, cp) : -1;
      if (g.op == QC_U4) {
        hg.kind = H2_K_U4;
        hg.tq = tq;
        hg.cq = cq;
      } else if (diag && (tq < 0 || (ctl && cq < 0))) {
        hg.kind = H2_K_PHASE;
      } else if (ctl) {
        hg.tq = tq;
        if (cq >= 0) {
          hg.kind = H2_K_REG2;
          hg.cq = cq;
        } else {
          hg.kind = H2_K_PRED;
        }
      } else {
        hg.kind = H2_K_REG1;
        hg.tq = tq;
      }
      if (par) {
        hg.pidx = 0;   // numbered when the rounds are laid out
        ++rbk.np;
      }
      rbk.g.push_back(hg);
      rbk.gid.push_back(j);
      when[j] = 3 * at + 1;
    }
    // the pending first table: tab_pre of the first round if that is a gate round, else a round of its own in front
    if (pend_tab >= 0) {
      if (!RS.empty() && RS[0].r.kind == H2_ROUND_GATES && RS[0].r.tab_pre < 0) {
        RS[0].r.tab_pre = pend_tab;
        RS[0].r.ts_pre = pend_ts;
        RS[0].tab_gid_pre = pend_gid;
      } else {
        RoundB nb;
        nb.r = {};
        nb.r.kind = H2_ROUND_TABLE;
        nb.r.table = pend_tab;
        nb.r.tslot = pend_ts;
        nb.r.tab_pre = nb.r.ts_pre = nb.r.tab_post = nb.r.ts_post = -1;
        nb.np = 0;
        nb.gid = pend_gid;
        RS.insert(RS.begin(), nb);
      }
    }
    // ---- round order: the stage's last round stores straight to HBM (and the adjoint sweep starts straight from HBM)
    // when its register bits lie above the always-local low bits.  If the last round cannot, move one that can behind
    // it, provided nothing scheduled after it depends on it.
    auto direct_ok = [&](const RoundB& rbk) { return rbk.r.kind == H2_ROUND_GATES && rbk.r.rb[0] >= LOW; };
    if ((int)RS.size() >= 2 && !direct_ok(RS.back())) {
      for (int k = (int)RS.size() - 2; k >= 0; --k) {
        if (!direct_ok(RS[k]) || RS[k].r.tab_post >= 0) continue;
        if (RS[k].r.tab_pre >= 0 && !(k == 0 && RS[1].r.kind == H2_ROUND_GATES && RS[1].r.tab_pre < 0)) continue;
        bool legal = true;
        auto depends = [&](const std::vector<int>& js) {
          for (int j : js)
            for (int i : pred[j])
              if (std::find(RS[k].gid.begin(), RS[k].gid.end(), i) != RS[k].gid.end()) return true;
          return false;
        };
        for (int r = k + 1; r < (int)RS.size() && legal; ++r)
          legal = !depends(RS[r].gid) && !depends(RS[r].tab_gid_pre) && !depends(RS[r].tab_gid_post);
        if (!legal) continue;
        RoundB mv = RS[k];
        if (mv.r.tab_pre >= 0) {   // the stage's first table stays in front
          RS[1].r.tab_pre = mv.r.tab_pre;
          RS[1].r.ts_pre = mv.r.ts_pre;
          RS[1].tab_gid_pre = mv.tab_gid_pre;
          mv.r.tab_pre = mv.r.ts_pre = -1;
          mv.tab_gid_pre.clear();
        }
        RS.erase(RS.begin() + k);
        RS.push_back(mv);
        break;
      }
    }
    // ---- lay the rounds out
    sd.r0 = (int)P.rounds.size();
    for (int r = 0; r < (int)RS.size(); ++r) {
      RoundB& rbk = RS[r];
      if (rbk.r.kind == H2_ROUND_TABLE) {
        sd.tab_round[rbk.r.tslot] = r;
      } else {
        if (rbk.r.tab_pre >= 0) sd.tab_round[rbk.r.ts_pre] = r;
        if (rbk.r.tab_post >= 0) sd.tab_round[rbk.r.ts_post] = r;
        rbk.r.g0 = (int)P.gates.size();
        rbk.r.ng = (int)rbk.g.size();
        for (H2Gate& hg : rbk.g) {
          if (h2_is_param(hg.op)) hg.pidx = sd.np++;
          P.gates.push_back(hg);
        }
      }
      P.rounds.push_back(rbk.r);
    }
    sd.nr = (int)P.rounds.size() - sd.r0;
    for (int b = 0; b < 24; ++b) sd.where[b] = 0;
    for (int j = 0; j < sd.nloc; ++j) sd.where[sd.lb[j]] = j;
    for (int j = 0; j < sd.ngb; ++j) sd.where[sd.gb[j]] = -(1 + j);
    P.stages.push_back(sd);
    for (int j = 0; j < G; ++j)
      if (state[j] == 1) {
        state[j] = 2;
        ++n_done;
      }
  }
  if (P.stages.empty()) {   // a program without gates: one empty stage (the embedding alone)
    H2Stage sd = {};
    sd.nloc = T;
    for (int j = 0; j < T; ++j) sd.lb[j] = j;
    sd.ngb = 0;
    for (int b = T; b < n; ++b) sd.gb[sd.ngb++] = b;
    for (int j = 0; j < sd.nloc; ++j) sd.where[sd.lb[j]] = j;
    for (int j = 0; j < sd.ngb; ++j) sd.where[sd.gb[j]] = -(1 + j);
    P.stages.push_back(sd);
  }
  // local indices of weight <= 3 in the first stage's tile (sparse read-out of the un-embedded cotangents)
  {
    const int nl = P.stages[0].nloc;
    for (int w = 0; w <= 3; ++w)
      for (int l = 0; l < (1 << nl); ++l)
        if (__builtin_popcount((unsigned)l) == w) P.sparse_idx.push_back(l);
  }
  return P;
}

// Flat int32 description (tests re-execute it on the CPU): see tests/test_hbm_plan.py for the reader.
inline std::vector<int> h2_describe(const H2Plan& P) {
  std::vector<int> o = {0x48324832, P.n, P.absorb, (int)P.stages.size(), (int)P.tables.size(), P.rbits};
  for (const H2Stage& s : P.stages) {
    o.push_back(s.nloc);
    o.push_back(s.ngb);
    for (int j = 0; j < s.nloc; ++j) o.push_back(s.lb[j]);
    for (int j = 0; j < s.ngb; ++j) o.push_back(s.gb[j]);
    o.push_back(s.nr);
    o.push_back(s.np);
    o.push_back(s.ntab);
    for (int r = s.r0; r < s.r0 + s.nr; ++r) {
      const H2Round& rd = P.rounds[r];
      o.push_back(rd.kind);
      o.push_back(rd.nrb);
      for (int q = 0; q < rd.nrb; ++q) o.push_back(rd.rb[q]);
      o.push_back(rd.table);
      o.push_back(rd.tslot);
      o.push_back(rd.tab_pre);
      o.push_back(rd.ts_pre);
      o.push_back(rd.tab_post);
      o.push_back(rd.ts_post);
      o.push_back(rd.kind == H2_ROUND_GATES ? rd.ng : 0);
      if (rd.kind == H2_ROUND_GATES)
        for (int g = rd.g0; g < rd.g0 + rd.ng; ++g) {
          const H2Gate& hg = P.gates[g];
          const int v[10] = {hg.op, hg.kind, hg.tq, hg.cq, hg.tbit, hg.cbit, hg.gi, hg.slot, hg.pidx, 0};
          o.insert(o.end(), v, v + 10);
        }
    }
  }
  for (const H2Table& t : P.tables) {
    o.push_back(t.ng);
    for (int g = t.g0; g < t.g0 + t.ng; ++g) {
      const H2DiagGate& d = P.dgates[g];
      const int v[5] = {d.op, d.bt, d.bc, d.gi, d.slot};
      o.insert(o.end(), v, v + 5);
    }
  }
  return o;
}
