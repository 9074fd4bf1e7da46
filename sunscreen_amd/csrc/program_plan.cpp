// sunscreen_amd/csrc/program_plan.cpp -- the scheduled executor for compiled FHE program graphs.
//
// The reference runs a compiled FheProgram with `traverse` (sunscreen_runtime/src/run.rs:372-472): every node whose
// operands are complete is spawned on the rayon pool, so all ready nodes run concurrently, each as one SEAL call on one
// ciphertext.  On a GPU "concurrently" means "in one launch": this file turns the graph -- once per Program, cached --
// into a PLAN of rounds, and a run walks the plan:
//
//   * Add / Sub / Negate trees whose inner nodes have a single user collapse into n-ary signed sums (exact canonical
//     arithmetic: any association order gives the bits of the node-by-node evaluation, run.rs:217-236,283-311); all sums
//     that are ready in a round are ONE table-driven launch (nary_sum_kernel).  Operands of different ciphertext sizes
//     are accepted as run.rs accepts them.  A tree node whose terms cancel identically (x - x) is never folded into its
//     user: it stays a result of its own, so the transparent-ciphertext failure of the reference's SEAL build
//     (seal_fhe/build.rs:46-66, sunscreen/tests/features.rs:8-34) still happens where SEAL raises it.
//   * a sum all of whose terms are products by plaintexts (examples/pir/src/main.rs:26-36: col[i] = sum_j db[i][j] *
//     col_query[j]) stays in the TRANSFORM DOMAIN: every ciphertext is transformed once however many products consume
//     it, the products are accumulated there (sum INTT(x_i) = INTT(sum x_i) exactly) and one inverse transform per sum
//     follows; sums over the same ciphertext list (the rows of a matrix-vector product) are one launch.  Plaintext
//     arguments may arrive already lifted and transformed (ProgramInput kind 2: a server's static database).
//   * ready Multiply->Relinearize pairs, stand-alone relinearisations and rotations by one Galois element become ONE
//     batched launch sequence over (members x batch) ciphertexts when the batch is small (a single input set -- the
//     reference's own call shape -- is latency-bound: five kernels per product); operands that are not already adjacent
//     in memory are staged by a pointer-table gather.  With a large batch every member already fills the device and
//     runs on its own, and an Add whose other operand is complete is folded into the key switch's last kernel.
//   * results that only feed an OutputCiphertext node are written straight into the caller's output buffer.
//
// The node-by-node executor of rounds 1-2 (program.cpp run_serial, HIPBFV_PROGRAM_SERIAL=1) is the cross-check arm:
// tests/test_gpu_program.py runs every graph through both and against the oracle interpreter.
#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <functional>
#include <iterator>
#include <map>
#include <set>
#include <unordered_map>

#include "program.hpp"

namespace hipbfv {

namespace {

enum StepKind : int {
  kStepNary = 0,    // all ready Add/Sub/Negate trees of a round: one table-driven launch
  kStepLinComb,     // sums of ciphertext x plaintext products over ONE ciphertext list, in the transform domain
  kStepMulRelin,    // fused Multiply -> Relinearize members (all squares or none)
  kStepMultiply,    // Multiply without a fused Relinearize (any operand sizes): one member
  kStepRelin,       // Relinearize of a materialised ciphertext (size 3 -> 2; size 2 is a copy)
  kStepGalois,      // rotations by one signed step count (one Galois element; NAF chain when its key is absent)
  kStepPlainOp,     // AddPlaintext / SubPlaintext / MultiplyPlaintext outside a transform-domain sum: one member
  kStepOutput,      // OutputCiphertext: copy unless the producer wrote the caller's buffer directly
};

struct Term {
  int slot;
  int sign;
};

}  // namespace

struct Program::Plan {
  struct Fold {  // an Add that can ride in this key-switching member's last kernel when the member runs unmerged
    int nary_step = -1, nary_member = -1, other_slot = -1;
  };
  // r06: a whole Add / Sub chain around a fused product -- the sum U = mult * product + sign * other, the product used by nothing else --
  // that the product's last kernel can write in the product's place (kernels.hpp MemberTail): examples/chi_sq's 2 x^2, 2 y^2 and
  // 4 n0 n2 - n1^2.  Decided at run time like Fold (the evaluator must run the all-FP64 fused path); the sum's own step skips it then.
  struct LinFold {
    int mult = 0, sign = 0, other_slot = -1;
    int nary_step = -1, nary_member = -1;
  };
  struct Step {
    int kind = 0;
    std::vector<int> node;   // member -> graph node (diagnostics, literal lookup)
    std::vector<int> out;    // member -> output slot
    std::vector<int> a, b;   // member -> operand slots (b: second factor; -1 when unused)
    std::vector<Fold> fold;  // key-switching kinds: per member
    std::vector<LinFold> lin;  // kStepMulRelin: per member (mult == 0: none)
    // kStepNary
    std::vector<u32> first;  // member -> offset into terms (size = members + 1)
    std::vector<Term> terms;
    // kStepLinComb: cts = the shared ciphertext list (cols); plain[member * cols + j] = plaintext NODE of term j
    std::vector<int> cts;
    std::vector<int> plain;
    bool square = false;
    int rot_steps = 0;       // kStepGalois: signed step count; swap = true for SwapRows
    bool swap = false;
    int plain_op = 0;        // kStepPlainOp: OpKind
    u32 out_size = 2;
    int out_index = -1;      // kStepOutput
    std::vector<int> release;  // slots whose last reader is this step
  };
  int rc = kOk;
  std::string err;
  std::vector<Step> steps;
  int nslots = 0;
  std::vector<u32> slot_size;
  std::vector<int> slot_input;   // slot -> program argument index (input ciphertexts), else -1
  std::vector<int> slot_direct;  // slot -> output index when its only reader is that OutputCiphertext node, else -1
  std::vector<int> plain_nodes;  // every InputPlaintext / LiteralPlaintext node some step reads
  std::vector<int> literal_nodes;
};

void Program::drop_plan() {
  std::lock_guard<std::mutex> g(plan_mu_);
  plan_.reset();
}

std::shared_ptr<const Program::Plan> Program::plan() const {
  std::lock_guard<std::mutex> g(plan_mu_);
  if (plan_) return plan_;
  auto P = std::make_shared<Plan>();
  auto fail = [&](int rc, const char* m) {
    P->rc = rc;
    P->err = m;
    plan_ = P;
    return plan_;
  };
  std::string verr;
  if (int rc = validate(&verr)) return fail(rc, verr.c_str());
  std::vector<int> order;
  if (!topo_order(&order)) return fail(kInvalidArg, "program graph has a cycle");
  const int nn = (int)nodes_.size();

  // ---- types, sizes, users ----
  enum { kTyNone = 0, kTyCt, kTyPlain, kTyU64 };
  std::vector<int> ty(nn, kTyNone);
  std::vector<u32> size(nn, 0);
  std::vector<int> uses(nn, 0);
  std::vector<int> user0(nn, -1);  // one user (the only one when uses == 1)
  for (int i = 0; i < nn; i++)
    for (int src : {nodes_[i].left, nodes_[i].right})
      if (src >= 0) {
        uses[src]++;
        user0[src] = i;
      }
  std::vector<std::vector<int>> users(nn);  // every user, once per operand slot
  for (int i = 0; i < nn; i++)
    for (int src : {nodes_[i].left, nodes_[i].right})
      if (src >= 0) users[src].push_back(i);
  auto is_linear = [&](int i) { return nodes_[i].op == kOpAdd || nodes_[i].op == kOpSub || nodes_[i].op == kOpNegate; };
  for (int id : order) {
    const Node& nd = nodes_[id];
    const int L = nd.left, R = nd.right;
    switch (nd.op) {
      case kOpInputCiphertext:
        ty[id] = kTyCt, size[id] = 2;
        break;
      case kOpInputPlaintext:
      case kOpLiteralPlaintext:
        ty[id] = kTyPlain;
        break;
      case kOpLiteralU64:
        ty[id] = kTyU64;
        break;
      case kOpMultiply:
        if (ty[L] != kTyCt) return fail(kInvalidArg, "left operand is not a ciphertext");
        if (ty[R] != kTyCt) return fail(kInvalidArg, "right operand is not a ciphertext");
        ty[id] = kTyCt, size[id] = size[L] + size[R] - 1;
        break;
      case kOpAdd:
      case kOpSub:
        if (ty[L] != kTyCt) return fail(kInvalidArg, "left operand is not a ciphertext");
        if (ty[R] != kTyCt) return fail(kInvalidArg, "right operand is not a ciphertext");
        ty[id] = kTyCt, size[id] = std::max(size[L], size[R]);
        break;
      case kOpNegate:
        if (ty[L] != kTyCt) return fail(kInvalidArg, "left operand is not a ciphertext");
        ty[id] = kTyCt, size[id] = size[L];
        break;
      case kOpRelinearize:
        if (ty[L] != kTyCt) return fail(kInvalidArg, "left operand is not a ciphertext");
        if (size[L] != 2 && size[L] != 3) return fail(kInvalidArg, "operation failed");  // SEAL relinearises size 3 -> 2 only
        ty[id] = kTyCt, size[id] = 2;
        break;
      case kOpAddPlaintext:
      case kOpSubPlaintext:
      case kOpMultiplyPlaintext:
        if (ty[L] != kTyCt) return fail(kInvalidArg, "left operand is not a ciphertext");
        if (ty[R] != kTyPlain) return fail(kInvalidArg, "right operand is not a plaintext");
        ty[id] = kTyCt, size[id] = size[L];
        break;
      case kOpShiftLeft:
      case kOpShiftRight:
        if (ty[L] != kTyCt) return fail(kInvalidArg, "left operand is not a ciphertext");
        if (nodes_[R].op != kOpLiteralU64) return fail(kInvalidArg, "shift amount must be a Literal::U64 (run.rs:177-183)");
        if (size[L] != 2) return fail(kInvalidArg, "rotation needs a size-2 ciphertext");
        ty[id] = kTyCt, size[id] = 2;
        break;
      case kOpSwapRows:
        if (ty[L] != kTyCt) return fail(kInvalidArg, "left operand is not a ciphertext");
        if (size[L] != 2) return fail(kInvalidArg, "rotation needs a size-2 ciphertext");
        ty[id] = kTyCt, size[id] = 2;
        break;
      case kOpOutputCiphertext:
        if (ty[L] != kTyCt || size[L] != 2) return fail(kInvalidArg, "program output must be a size-2 ciphertext");
        break;
      default:
        return fail(kInvalidArg, "unsupported operation");
    }
    if (ty[id] == kTyCt && size[id] > 16) return fail(kInvalidArg, "operation failed");
  }

  // ---- Multiply -> Relinearize fusion: the product of two size-2 ciphertexts with the Relinearize as its only user ----
  std::vector<char> virt(nn, 0);  // nodes that never materialise (fused products, absorbed sums, products inside transform-domain sums)
  std::vector<int> fused_mul(nn, -1);
  for (int i = 0; i < nn; i++)
    if (nodes_[i].op == kOpRelinearize) {
      const int m = nodes_[i].left;
      if (nodes_[m].op == kOpMultiply && uses[m] == 1 && size[nodes_[m].left] == 2 && size[nodes_[m].right] == 2) fused_mul[i] = m, virt[m] = 1;
    }

  // ---- Add / Sub / Negate trees -> n-ary sums over LEAF nodes ----
  // terms[v] exists for linear nodes that are roots (or are still waiting to be absorbed by their user)
  struct Lin {
    std::vector<std::pair<int, int>> t;  // (leaf node, sign)
    long sum = 0;                        // sum of the signs
  };
  std::vector<Lin> lin(nn);
  // Would SEAL have refused this node's result?  Its build throws on a TRANSPARENT ciphertext (every polynomial but the first
  // identically zero).  For a node that materialises, the device check sees that; a node folded into its user never
  // materialises, so the fold is only allowed when the node's "tail" -- polynomials 1.. as a formal combination of opaque
  // values -- does not vanish identically.  Opaque values are compared by VALUE NUMBER (same operation on the same operands
  // is the same value: x*y - x*y cancels although the two products are different nodes); AddPlaintext / SubPlaintext leave the
  // tail of their operand unchanged (x - (x + p) is transparent without being zero).  Coincidental cancellation of random
  // residues is not a concern (probability 2^-(bits of q) per word).
  std::vector<int> vn(nn, -1);
  {
    std::map<std::array<long long, 4>, int> table;
    for (int id : order) {
      const Node& nd = nodes_[id];
      long long l = nd.left >= 0 ? vn[nd.left] : -1, r = nd.right >= 0 ? vn[nd.right] : -1;
      if ((nd.op == kOpAdd || nd.op == kOpMultiply) && l > r) std::swap(l, r);
      const bool has_arg = nd.op == kOpInputCiphertext || nd.op == kOpInputPlaintext || nd.op == kOpLiteralU64 || nd.op == kOpLiteralPlaintext;
      if (nd.op == kOpOutputCiphertext) {
        vn[id] = id;
        continue;
      }
      auto it = table.emplace(std::array<long long, 4>{(long long)nd.op, l, r, has_arg ? (long long)nd.arg : 0}, id);
      vn[id] = it.first->second;
    }
  }
  typedef std::unordered_map<int, long> TailMap;
  std::unordered_map<int, TailMap> tail_memo;
  long tail_budget = 4000000;  // map entries this analysis may create; beyond it the answer is the conservative "may cancel"
  std::function<const TailMap*(int)> tail_of = [&](int v) -> const TailMap* {
    v = vn[v];
    auto it = tail_memo.find(v);
    if (it != tail_memo.end()) return &it->second;
    if (tail_budget <= 0) return nullptr;
    const Node& nd = nodes_[v];
    TailMap m;
    if (nd.op == kOpAddPlaintext || nd.op == kOpSubPlaintext) {
      const TailMap* a = tail_of(nd.left);
      if (!a) return nullptr;
      m = *a;
    } else if (is_linear(v)) {
      const TailMap* a = tail_of(nd.left);
      if (!a) return nullptr;
      const long sa = nd.op == kOpNegate ? -1 : 1;
      for (auto& kv : *a) m[kv.first] += sa * kv.second;
      if (nd.op != kOpNegate) {
        const TailMap* b = tail_of(nd.right);
        if (!b) return nullptr;
        const long sb = nd.op == kOpSub ? -1 : 1;
        for (auto& kv : *b) m[kv.first] += sb * kv.second;
      }
      for (auto i2 = m.begin(); i2 != m.end();) i2 = i2->second ? std::next(i2) : m.erase(i2);
    } else {
      m[v] = 1;
    }
    tail_budget -= (long)m.size() + 1;
    return &tail_memo.emplace(v, std::move(m)).first->second;
  };
  auto opaque = [&](int v) { return !is_linear(v) && nodes_[v].op != kOpAddPlaintext && nodes_[v].op != kOpSubPlaintext; };
  auto cancels = [&](int node, const Lin& l) {
    // the cheap certificate first: distinct opaque values with a non-zero sign sum cannot cancel (a sum of products, a chain of Adds)
    bool simple = l.sum != 0;
    for (size_t i = 0; i < l.t.size() && simple; i++) simple = opaque(l.t[i].first);
    if (simple) {
      bool all_plus = true;
      for (auto& p : l.t) all_plus = all_plus && p.second > 0;
      if (all_plus) return false;
    }
    const TailMap* m = tail_of(node);
    return !m || m->empty();
  };
  std::vector<int> absorbed_into(nn, -1);  // linear node -> the linear node that took its terms
  for (int id : order) {
    if (!is_linear(id)) continue;
    const Node& nd = nodes_[id];
    Lin& me = lin[id];
    // r06: x + x for a linear x that nothing else uses (chi_sq: t = p + p, then t + t): both operand slots of this Add are the node's
    // only two uses -- its terms are taken twice (the executor lists repeated terms one by one; modular addition: same bits)
    if (nd.op == kOpAdd && nd.left == nd.right && is_linear(nd.left) && uses[nd.left] == 2 && !virt[nd.left] && !cancels(nd.left, lin[nd.left])) {
      Lin& o = lin[nd.left];
      for (int rep = 0; rep < 2; rep++)
        for (auto& p : o.t) me.t.push_back(p);
      me.sum += 2 * o.sum;
      std::vector<std::pair<int, int>>().swap(o.t);
      virt[nd.left] = 1;
      absorbed_into[nd.left] = id;
      continue;
    }
    auto take = [&](int src, int sign) {
      // absorb a linear operand whose only user is this node and whose tail does not vanish identically
      if (is_linear(src) && uses[src] == 1 && !virt[src] && !cancels(src, lin[src])) {
        Lin& o = lin[src];
        for (auto& p : o.t) me.t.push_back({p.first, p.second * sign});
        me.sum += o.sum * sign;
        std::vector<std::pair<int, int>>().swap(o.t);
        virt[src] = 1;
        absorbed_into[src] = id;
      } else {
        me.t.push_back({src, sign});
        me.sum += sign;
      }
    };
    take(nd.left, nd.op == kOpNegate ? -1 : 1);
    if (nd.op != kOpNegate) take(nd.right, nd.op == kOpSub ? -1 : 1);
  }
  // roots all of whose terms are +1 products by plaintexts of size-2 ciphertexts, each product used here only: transform-domain sums
  std::vector<char> lincomb(nn, 0);
  for (int id = 0; id < nn; id++) {
    if (!is_linear(id) || virt[id]) continue;
    const Lin& l = lin[id];
    if (l.t.size() < 2) continue;
    bool ok = true;
    for (auto& p : l.t) {
      const int leaf = p.first;
      if (p.second != 1 || nodes_[leaf].op != kOpMultiplyPlaintext || uses[leaf] != 1 || size[leaf] != 2 || virt[leaf]) {
        ok = false;
        break;
      }
    }
    // a product may appear once only (a repeated leaf would have uses >= 2 anyway)
    if (ok) {
      lincomb[id] = 1;
      for (auto& p : l.t) virt[p.first] = 1;
    }
  }

  // ---- value slots: one per materialised ciphertext node ----
  std::vector<int> slot(nn, -1);
  for (int id = 0; id < nn; id++)
    if (ty[id] == kTyCt && !virt[id]) {
      slot[id] = P->nslots++;
      P->slot_size.push_back(size[id]);
      P->slot_input.push_back(nodes_[id].op == kOpInputCiphertext ? (int)nodes_[id].arg : -1);
    }
  P->slot_direct.assign(P->nslots, -1);
  {
    int out_idx = 0;
    for (int i = 0; i < nn; i++)
      if (nodes_[i].op == kOpOutputCiphertext) {
        const int src = nodes_[i].left;
        if (uses[src] == 1 && nodes_[src].op != kOpInputCiphertext) P->slot_direct[slot[src]] = out_idx;
        out_idx++;
      }
  }

  // ---- macro nodes and their operand slots ----
  // work[id] = 1 for nodes that execute as a member of some step
  std::vector<char> work(nn, 0);
  std::vector<std::vector<int>> deps(nn);  // operand NODES (materialised ciphertexts) a macro node waits for
  for (int id = 0; id < nn; id++) {
    const Node& nd = nodes_[id];
    if (virt[id]) continue;
    if (nd.op == kOpOutputCiphertext) {
      work[id] = 1;
      deps[id] = {nd.left};
      continue;
    }
    if (ty[id] != kTyCt || nd.op == kOpInputCiphertext) continue;
    work[id] = 1;
    if (is_linear(id)) {
      for (auto& p : lin[id].t) deps[id].push_back(lincomb[id] ? nodes_[p.first].left : p.first);
    } else if (nd.op == kOpRelinearize && fused_mul[id] >= 0) {
      deps[id] = {nodes_[fused_mul[id]].left, nodes_[fused_mul[id]].right};
    } else if (nd.op == kOpMultiply || nd.op == kOpAdd || nd.op == kOpSub) {
      deps[id] = {nd.left, nd.right};
    } else {
      deps[id] = {nd.left};
    }
  }
  // ---- rounds ----
  std::vector<char> done(nn, 0);
  std::vector<int> produced_at(P->nslots, -1);  // step that wrote the slot (-1: program input)
  for (int id = 0; id < nn; id++)
    if (!work[id]) done[id] = 1;  // inputs, literals, virtual nodes: nothing to wait for (virtual nodes are never operands)
  std::vector<int> pending;
  for (int id : order)
    if (work[id]) pending.push_back(id);
  auto ready = [&](int id) {
    for (int d : deps[id])
      if (!done[d]) return false;
    return true;
  };
  auto is_cheap = [&](int id) {
    const OpKind op = nodes_[id].op;
    return (is_linear(id) && !lincomb[id]) || op == kOpAddPlaintext || op == kOpSubPlaintext || op == kOpOutputCiphertext;
  };
  std::vector<int> nary_step_of(nn, -1), nary_member_of(nn, -1);
  while (!pending.empty()) {
    std::vector<int> rd, rest;
    for (int id : pending) (ready(id) ? rd : rest).push_back(id);
    if (rd.empty()) return fail(kInvalidArg, "program graph has a cycle");
    std::vector<int> cheap, heavy;
    for (int id : rd) (is_cheap(id) ? cheap : heavy).push_back(id);
    std::vector<int> run_now = cheap.empty() ? heavy : cheap;
    if (!cheap.empty()) rest.insert(rest.end(), heavy.begin(), heavy.end());  // the expensive kinds wait until every cheap node that can run has run: more of them meet in one launch
    // keep `pending` in topological order for determinism
    {
      std::vector<char> in_rest(nn, 0);
      for (int id : rest) in_rest[id] = 1;
      std::vector<int> nxt;
      for (int id : pending)
        if (in_rest[id]) nxt.push_back(id);
      pending.swap(nxt);
    }
    if (!cheap.empty()) {
      Plan::Step nary;
      nary.kind = kStepNary;
      for (int id : run_now) {
        const Node& nd = nodes_[id];
        if (is_linear(id)) {
          nary.node.push_back(id);
          nary.out.push_back(slot[id]);
          nary.first.push_back((u32)nary.terms.size());
          for (auto& p : lin[id].t) nary.terms.push_back(Term{slot[p.first], p.second});
        } else if (nd.op == kOpOutputCiphertext) {
          continue;  // below, after the sums of this round
        } else {
          Plan::Step st;
          st.kind = kStepPlainOp;
          st.plain_op = nd.op;
          st.node = {id};
          st.out = {slot[id]};
          st.a = {slot[nd.left]};
          st.plain = {nd.right};
          st.out_size = size[id];
          produced_at[slot[id]] = (int)P->steps.size();
          P->steps.push_back(std::move(st));
        }
      }
      if (!nary.node.empty()) {
        nary.first.push_back((u32)nary.terms.size());
        const int sidx = (int)P->steps.size();
        for (size_t m = 0; m < nary.node.size(); m++) {
          produced_at[nary.out[m]] = sidx;
          nary_step_of[nary.node[m]] = sidx;
          nary_member_of[nary.node[m]] = (int)m;
        }
        P->steps.push_back(std::move(nary));
      }
      {
        int out_idx = 0;
        std::vector<int> out_index_of(nn, -1);
        for (int i = 0; i < nn; i++)
          if (nodes_[i].op == kOpOutputCiphertext) out_index_of[i] = out_idx++;
        for (int id : run_now)
          if (nodes_[id].op == kOpOutputCiphertext) {
            Plan::Step st;
            st.kind = kStepOutput;
            st.node = {id};
            st.a = {slot[nodes_[id].left]};
            st.out_index = out_index_of[id];
            P->steps.push_back(std::move(st));
          }
      }
      for (int id : run_now) done[id] = 1;
      continue;
    }
    // ---- the expensive kinds: group the ready members ----
    std::vector<Plan::Step> groups;
    auto group_for = [&](auto&& match, auto&& init) -> Plan::Step& {
      for (auto& g : groups)
        if (match(g)) return g;
      groups.emplace_back();
      init(groups.back());
      return groups.back();
    };
    for (int id : run_now) {
      const Node& nd = nodes_[id];
      if (lincomb[id]) {
        std::vector<int> cts;
        for (auto& p : lin[id].t) cts.push_back(slot[nodes_[p.first].left]);
        Plan::Step& g = group_for([&](const Plan::Step& s) { return s.kind == kStepLinComb && s.cts == cts; },
                                  [&](Plan::Step& s) {
                                    s.kind = kStepLinComb;
                                    s.cts = cts;
                                  });
        g.node.push_back(id);
        g.out.push_back(slot[id]);
        for (auto& p : lin[id].t) g.plain.push_back(nodes_[p.first].right);
      } else if (nd.op == kOpRelinearize && fused_mul[id] >= 0) {
        const Node& mn = nodes_[fused_mul[id]];
        const bool sq = mn.left == mn.right;
        Plan::Step& g = group_for([&](const Plan::Step& s) { return s.kind == kStepMulRelin && s.square == sq; },
                                  [&](Plan::Step& s) {
                                    s.kind = kStepMulRelin;
                                    s.square = sq;
                                  });
        g.node.push_back(id);
        g.out.push_back(slot[id]);
        g.a.push_back(slot[mn.left]);
        g.b.push_back(slot[mn.right]);
      } else if (nd.op == kOpRelinearize) {
        const u32 in_size = size[nd.left];
        Plan::Step& g = group_for([&](const Plan::Step& s) { return s.kind == kStepRelin && s.out_size == in_size; },
                                  [&](Plan::Step& s) {
                                    s.kind = kStepRelin;
                                    s.out_size = in_size;  // the INPUT size of a relinearisation group (its result is always 2)
                                  });
        g.node.push_back(id);
        g.out.push_back(slot[id]);
        g.a.push_back(slot[nd.left]);
        g.b.push_back(-1);
      } else if (nd.op == kOpShiftLeft || nd.op == kOpShiftRight || nd.op == kOpSwapRows) {
        const bool swap = nd.op == kOpSwapRows;
        const int k = swap ? 0 : (int)nodes_[nd.right].arg;
        const int steps = nd.op == kOpShiftLeft ? k : -k;
        Plan::Step& g = group_for([&](const Plan::Step& s) { return s.kind == kStepGalois && s.swap == swap && s.rot_steps == steps; },
                                  [&](Plan::Step& s) {
                                    s.kind = kStepGalois;
                                    s.swap = swap;
                                    s.rot_steps = steps;
                                  });
        g.node.push_back(id);
        g.out.push_back(slot[id]);
        g.a.push_back(slot[nd.left]);
        g.b.push_back(-1);
      } else if (nd.op == kOpMultiply) {
        groups.emplace_back();
        Plan::Step& g = groups.back();
        g.kind = kStepMultiply;
        g.node = {id};
        g.out = {slot[id]};
        g.a = {slot[nd.left]};
        g.b = {slot[nd.right]};
        g.out_size = size[id];
      } else if (nd.op == kOpMultiplyPlaintext) {
        groups.emplace_back();
        Plan::Step& g = groups.back();
        g.kind = kStepPlainOp;
        g.plain_op = nd.op;
        g.node = {id};
        g.out = {slot[id]};
        g.a = {slot[nd.left]};
        g.plain = {nd.right};
        g.out_size = size[id];
      } else {
        return fail(kInvalidArg, "unsupported operation");
      }
    }
    // ---- r06: linear folds of fused products (Plan::LinFold) ----
    // U = the one sum root every use of the product ends in; U's terms = mult copies of the product (all +1, mult <= 4) and at most
    // one other ciphertext (sign +-1, size 2) that exists when the product's launch runs: made in an earlier round, or by another
    // group of THIS round -- the groups are then ordered producer first.  Two products never share a U.
    {
      const size_t ng = groups.size();
      std::vector<int> group_of(nn, -1);
      for (size_t gi = 0; gi < ng; gi++)
        for (int id : groups[gi].node) group_of[id] = (int)gi;
      std::vector<std::vector<int>> before(ng);  // before[g]: groups whose results g's folds read
      std::set<int> claimed;
      for (size_t gi = 0; gi < ng; gi++) {
        Plan::Step& g = groups[gi];
        if (g.kind != kStepMulRelin) continue;
        g.lin.assign(g.node.size(), Plan::LinFold());
        for (size_t m = 0; m < g.node.size(); m++) {
          const int id = g.node[m];
          int U = -1;
          bool ok = !users[id].empty();
          for (int u : users[id]) {
            if (!is_linear(u)) {
              ok = false;
              break;
            }
            int r = u;
            while (absorbed_into[r] >= 0) r = absorbed_into[r];
            if (U >= 0 && r != U) ok = false;
            U = r;
          }
          if (!ok || U < 0 || virt[U] || lincomb[U] || size[U] != 2 || claimed.count(U)) continue;
          int mult = 0, other = -1, osign = 0;
          for (auto& p : lin[U].t) {
            if (p.first == id) {
              if (p.second != 1) ok = false;
              mult++;
            } else if (other < 0) {
              other = p.first, osign = p.second;
            } else {
              ok = false;
            }
          }
          if (!ok || mult < 1 || mult > 4 || (mult == 1 && other < 0)) continue;
          if (other >= 0) {
            if (size[other] != 2 || ty[other] != kTyCt) continue;
            if (!done[other]) {
              const int og = group_of[other];
              if (og < 0 || og == (int)gi) continue;  // not made yet, or made by this very launch
              before[gi].push_back(og);
            }
          }
          claimed.insert(U);
          g.lin[m].mult = mult;
          g.lin[m].sign = other >= 0 ? osign : 0;
          g.lin[m].other_slot = other >= 0 ? slot[other] : -1;
          g.lin[m].nary_member = U;  // node id for now; resolved to (step, member) after the sum is scheduled
        }
      }
      // producer-first order (stable); a cycle drops the folds that caused it
      std::vector<size_t> ord;
      std::vector<char> placed(ng, 0);
      while (ord.size() < ng) {
        bool any = false;
        for (size_t gi = 0; gi < ng; gi++) {
          if (placed[gi]) continue;
          bool free = true;
          for (int b : before[gi]) free = free && placed[b];
          if (!free) continue;
          placed[gi] = 1, ord.push_back(gi), any = true;
        }
        if (!any) {
          for (size_t gi = 0; gi < ng; gi++)
            if (!placed[gi]) {
              for (auto& lf : groups[gi].lin)
                if (lf.other_slot >= 0) lf = Plan::LinFold();
              before[gi].clear();
            }
        }
      }
      std::vector<Plan::Step> sorted;
      for (size_t gi : ord) sorted.push_back(std::move(groups[gi]));
      groups.swap(sorted);
    }
    for (auto& g : groups) {
      const int sidx = (int)P->steps.size();
      if (g.kind == kStepMulRelin || g.kind == kStepRelin || g.kind == kStepGalois) {
        g.fold.resize(g.node.size());
        // an Add (a two-term sum, both +1, size 2) of this member's result -- its only user -- and a ciphertext that is
        // already complete: the member's last kernel can add it when the member runs on its own
        for (size_t m = 0; m < g.node.size(); m++) {
          const int id = g.node[m];
          if (g.kind == kStepRelin && g.out_size != 3) continue;
          if (uses[id] != 1) continue;
          const int u = user0[id];
          if (!is_linear(u) || virt[u] || lincomb[u]) continue;
          const Lin& l = lin[u];
          if (l.t.size() != 2 || l.t[0].second != 1 || l.t[1].second != 1 || size[u] != 2) continue;
          const int other = l.t[0].first == id ? l.t[1].first : l.t[0].first;
          if (other == id || size[other] != 2 || !done[other]) continue;
          g.fold[m].other_slot = slot[other];
          g.fold[m].nary_member = u;  // node id for now; resolved to (step, member) after the sum is scheduled
        }
      }
      for (size_t m = 0; m < g.out.size(); m++) produced_at[g.out[m]] = sidx;
      P->steps.push_back(std::move(g));
    }
    for (int id : run_now) done[id] = 1;
  }
  // resolve the folds: the sum node -> its (step, member)
  for (auto& st : P->steps)
    for (auto& f : st.fold)
      if (f.other_slot >= 0) {
        const int u = f.nary_member;
        f.nary_step = nary_step_of[u];
        f.nary_member = nary_member_of[u];
        if (f.nary_step < 0) f = Plan::Fold();
      }
  for (auto& st : P->steps)
    for (auto& f : st.lin)
      if (f.mult) {
        const int u = f.nary_member;
        f.nary_step = nary_step_of[u];
        f.nary_member = nary_member_of[u];
        if (f.nary_step < 0) f = Plan::LinFold();
      }

  // ---- release lists: the last step that reads each slot ----
  std::vector<int> last_read(P->nslots, -1);
  for (int si = 0; si < (int)P->steps.size(); si++) {
    const Plan::Step& st = P->steps[si];
    auto rd = [&](int sl) {
      if (sl >= 0) last_read[sl] = si;
    };
    for (int sl : st.a) rd(sl);
    for (int sl : st.b) rd(sl);
    for (int sl : st.cts) rd(sl);
    for (auto& t : st.terms) rd(t.slot);
  }
  for (int sl = 0; sl < P->nslots; sl++) {
    if (P->slot_input[sl] >= 0) continue;
    const int at = last_read[sl] >= 0 ? last_read[sl] : produced_at[sl];  // a value nobody reads dies where it was made
    if (at >= 0) P->steps[at].release.push_back(sl);
  }
  for (int id = 0; id < nn; id++)
    if (nodes_[id].op == kOpLiteralPlaintext) P->literal_nodes.push_back(id);
  plan_ = P;
  return plan_;
}

int Program::describe(std::string* out) const {
  std::shared_ptr<const Plan> Pp = plan();
  const Plan& P = *Pp;
  if (P.rc) {
    *out = "error: " + P.err;
    return P.rc;
  }
  static const char* names[] = {"sum", "plain_matrix", "mul_relin", "multiply", "relinearize", "rotate", "plain_op", "output"};
  std::string t;
  for (const Plan::Step& st : P.steps) {
    t += names[st.kind];
    t += " members=" + std::to_string(st.node.size());
    if (st.kind == kStepNary) t += " terms=" + std::to_string(st.terms.size());
    if (st.kind == kStepLinComb) t += " columns=" + std::to_string(st.cts.size());
    if (st.kind == kStepMulRelin && st.square) t += " square";
    if (st.kind == kStepGalois) t += st.swap ? " swap_rows" : " steps=" + std::to_string(st.rot_steps);
    size_t folds = 0;
    for (auto& f : st.fold) folds += f.other_slot >= 0;
    if (folds) t += " add_foldable=" + std::to_string(folds);
    size_t lins = 0;
    for (auto& f : st.lin) lins += f.mult != 0;
    if (lins) t += " lin_foldable=" + std::to_string(lins);
    size_t direct = 0;
    for (int sl : st.out) direct += sl >= 0 && P.slot_direct[sl] >= 0;
    if (direct && st.kind != kStepOutput) t += " direct_outputs=" + std::to_string(direct);
    t += "\n";
  }
  *out = t;
  return kOk;
}

// =====================================================================================
// one run of a plan
// =====================================================================================
namespace {

// pinned, device-addressable staging for the descriptor tables of one run (grown on demand).  A run ends by RECORDING an event
// behind its last table copy instead of draining the stream (r05; ADVICE r03's last item): the arena is not touched again until
// that event has passed -- `wait_idle` at the start of the run that reuses it.
struct TableArena {
  unsigned char* host = nullptr;
  size_t cap = 0, used = 0;
  hipEvent_t done = nullptr;  // behind the last copy out of this arena
  bool pending = false;
  // the device has finished reading this arena (usually long ago: the event is from two runs back)
  bool wait_idle() {
    if (pending && hipEventSynchronize(done) != hipSuccess) return false;
    pending = false;
    return true;
  }
  bool reserve(size_t bytes) {
    if (bytes <= cap) return true;
    if (host) (void)hipHostFree(host);
    host = nullptr;
    cap = 0;
    const size_t want = std::max<size_t>(bytes, 1 << 20);
    if (hipHostMalloc((void**)&host, want, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
      host = nullptr;
      (void)hipGetLastError();
      return false;
    }
    cap = want;
    return true;
  }
  // the copies out of the arena are enqueued on `s`: mark their end.  The event belongs to the device it was created on: a host
  // thread that drives evaluators on two devices gets a fresh one when the device changes.  If the event cannot be
  // created or recorded the stream is drained instead -- the arena is then idle by construction, never "pending" with copies in flight.
  int done_dev = -1;
  bool mark(hipStream_t s) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done && dev != done_dev) {
      (void)hipEventDestroy(done);
      done = nullptr;
    }
    if (!done && hipEventCreateWithFlags(&done, hipEventDisableTiming) == hipSuccess) done_dev = dev;
    if (done && hipEventRecord(done, s) == hipSuccess) {
      pending = true;
      return true;
    }
    (void)hipGetLastError();
    pending = false;
    return hipStreamSynchronize(s) == hipSuccess;
  }
  ~TableArena() {
    (void)wait_idle();
    if (done) (void)hipEventDestroy(done);
    if (host) (void)hipHostFree(host);
  }
};
// Two arenas per host thread, used alternately (thread_local in run_plan; a thread that ends gives its pinned memory back, ADVICE
// r03): while the device still copies run i's tables out of one, the host builds run i + 1's in the other, and a scheduled run
// returns without a stream synchronisation of its own.
struct TableArenas {
  TableArena a[2];
  unsigned turn = 0;
  TableArena& next() { return a[turn++ & 1u]; }
};

constexpr size_t merge_max_batch() { return 32; }  // (measured in r03: DESIGN.md, the scheduled executor)

}  // namespace

int Program::run_plan(Evaluator& ev, size_t batch, const ProgramInput* inputs, size_t num_inputs, const ProgramKeys& keys,
                      u64* const* outputs, size_t num_outputs_given, hipStream_t s,
                      std::string* err) const {
  const KeySel relin_key = keys.relin_sel();
  auto fail = [&](int code, const char* m) {
    if (err) *err = m;
    return code;
  };
  std::shared_ptr<const Plan> Pp = plan();
  const Plan& P = *Pp;
  if (P.rc) return fail(P.rc, P.err.c_str());
  if (num_outputs_given != num_outputs()) return fail(kInvalidArg, "wrong number of output buffers");
  if (!batch) return fail(kInvalidArg, "empty batch");
  // the table-driven kernels put (output, item) on grid z: hipbfv_Program_Run hands over at most this many input sets per call
  // (larger batches are a sequence of runs over offset pointers, capi.cpp)
  if (batch > 32768) return fail(kInvalidArg, "more than 32768 input sets in one scheduled run");
  Context* ctx = ev.ctx();
  const DevCtx& h = ctx->host();
  const size_t n = ctx->n(), K = ctx->K();
  const size_t poly = K * n;
  ScratchPool& pool = ev.scratch();

  // ---- slot storage: blocks with reference counts (members of a merged launch share one block; a folded Add shares its producer's) ----
  struct Block {
    void* ptr;
    int refs;
    bool owned;
  };
  std::vector<Block> blocks;
  std::vector<const u64*> sp(P.nslots, nullptr);
  std::vector<int> block_of(P.nslots, -1);
  std::vector<char> direct_written(num_outputs_given, 0);
  std::vector<void*> temps;  // staging buffers released when the run ends (stream-ordered)
  struct SideProduct {
    const u64* base = nullptr;      // the product's output block (rows x batch ciphertexts)
    size_t rows = 0;
    std::vector<size_t> row_end;    // chunk c covers rows [row_end[c - 1], row_end[c])
    std::vector<hipEvent_t> done;   // recorded on the side stream behind chunk c
    bool noted = false;
  } side;
  auto cleanup = [&](int code, const char* m) {
    // (nothing of the side stream may still run when the run's buffers go back to the pool)
    for (hipEvent_t e : side.done) {
      (void)hipEventSynchronize(e);
      (void)hipEventDestroy(e);
    }
    side.done.clear();
    side.base = nullptr;
    for (Block& b : blocks)
      if (b.owned && b.refs > 0) pool.release(b.ptr, s);
    for (void* t : temps) pool.release(t, s);
    (void)hipStreamSynchronize(s);  // descriptor tables of this run live in this thread's arena
    return fail(code, m);
  };
  auto new_block = [&](size_t words) -> int {
    void* p = pool.acquire(words * sizeof(u64), s);
    if (!p) return -1;
    blocks.push_back(Block{p, 0, true});
    return (int)blocks.size() - 1;
  };
  auto bind = [&](int slot, int blk, const u64* ptr) {
    sp[slot] = ptr;
    block_of[slot] = blk;
    if (blk >= 0) blocks[blk].refs++;
  };
  auto unbind = [&](int slot) {
    const int blk = block_of[slot];
    if (blk < 0) return;
    block_of[slot] = -1;
    if (--blocks[blk].refs == 0 && blocks[blk].owned) pool.release(blocks[blk].ptr, s);
  };
  auto slot_words = [&](int slot) { return batch * (size_t)P.slot_size[slot] * poly; };
  // a member's own output buffer: the caller's output buffer when the value only feeds an OutputCiphertext node
  auto alloc_member = [&](int slot) -> u64* {
    const int d = P.slot_direct[slot];
    if (d >= 0) {
      direct_written[d] = 1;
      blocks.push_back(Block{outputs[d], 0, false});
      bind(slot, (int)blocks.size() - 1, outputs[d]);
      return outputs[d];
    }
    const int blk = new_block(slot_words(slot));
    if (blk < 0) return nullptr;
    bind(slot, blk, (const u64*)blocks[blk].ptr);
    return (u64*)blocks[blk].ptr;
  };

  // ---- program arguments ----
  for (int sl = 0; sl < P.nslots; sl++) {
    const int arg = P.slot_input[sl];
    if (arg < 0) continue;
    if ((size_t)arg >= num_inputs) return fail(kInvalidArg, "input index out of range");
    if (inputs[arg].kind != 0 || !inputs[arg].ptr) return fail(kInvalidArg, "argument is not a ciphertext");
    sp[sl] = inputs[arg].ptr;
  }
  // plaintext operands: program arguments as they are, literals uploaded once per run
  struct PlainVal {
    const u64* ptr = nullptr;
    size_t stride = 0;
    int kind = 1;
  };
  std::unordered_map<int, PlainVal> literal_val;
  if (!P.literal_nodes.empty()) {
    const std::vector<u64>& kp = ctx->key_primes();
    u64* dev = (u64*)pool.acquire(P.literal_nodes.size() * n * sizeof(u64), s);
    if (!dev) return fail(kOutOfMemory, "scratch allocation failed");
    temps.push_back(dev);
    std::vector<u64> host(P.literal_nodes.size() * n, 0);
    for (size_t i = 0; i < P.literal_nodes.size(); i++) {
      const PlainLiteral& lit = literals_[nodes_[P.literal_nodes[i]].arg];
      if (lit.n != n || lit.t != h.t || lit.primes.size() != kp.size() || !std::equal(kp.begin(), kp.end(), lit.primes.begin()))
        return cleanup(kInvalidArg, "plaintext literal was built for different encryption parameters");
      std::copy(lit.coeffs.begin(), lit.coeffs.end(), host.begin() + i * n);
      literal_val[P.literal_nodes[i]] = PlainVal{dev + i * n, 0, 1};
    }
    // ordered after earlier users of the recycled buffer on this stream; drained so `host` may go out of scope
    if (hipMemcpyAsync(dev, host.data(), host.size() * sizeof(u64), hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      return cleanup(kHipError, "copy failed");
  }
  auto plain_of = [&](int node, PlainVal* out) -> const char* {
    const Node& nd = nodes_[node];
    if (nd.op == kOpLiteralPlaintext) {
      *out = literal_val[node];
      return nullptr;
    }
    if (nd.arg >= num_inputs) return "input index out of range";
    const ProgramInput& in = inputs[nd.arg];
    if ((in.kind != 1 && in.kind != 2) || !in.ptr) return "argument is not a plaintext";
    *out = PlainVal{in.ptr, in.stride, in.kind};
    return nullptr;
  };

  // ---- descriptor tables: pinned host arena -> one device buffer per table ----
  thread_local TableArenas arenas;
  TableArena& arena = arenas.next();
  if (!arena.wait_idle()) return cleanup(kHipError, "event synchronisation failed");
  arena.used = 0;
  bool tables_used = false;
  // reserve the worst case up front so that the host pointers handed out stay valid for the whole run
  {
    size_t need = 0;
    for (const Plan::Step& st : P.steps) {
      need += (st.node.size() + 1 + st.terms.size() / 8) * sizeof(NaryOut) + 2 * st.terms.size() * sizeof(NaryTerm) + st.plain.size() * sizeof(PlainNttRef) +
              (st.a.size() + st.b.size() + st.cts.size() + 4) * sizeof(u64*) + 512;
    }
    if (!arena.reserve(need + 4096)) return cleanup(kOutOfMemory, "descriptor table allocation failed");
  }
  // a table is either copied into the arena (stage_table) or built in place there (stage_begin / stage_commit: the 1 MB
  // plaintext table of a 256 x 256 lookup is written once instead of built, then copied)
  auto stage_begin = [&](size_t bytes) -> unsigned char* {
    const size_t off = (arena.used + 63) & ~(size_t)63;
    arena.used = off + ((bytes + 7) / 8) * 8;
    return arena.host + off;
  };
  auto stage_commit = [&](const unsigned char* hp, size_t bytes) -> const void* {  // returns the DEVICE copy
    const size_t words = (bytes + 7) / 8;
    void* dev = pool.acquire(words * 8, s);
    if (!dev) return nullptr;
    temps.push_back(dev);
    // a kernel, not hipMemcpyAsync: a copy enqueued on the caller's stream waits for the stream to drain on the HOST when that
    // stream is the null stream (PyTorch's default), and every drained launch queue costs the device idle time
    if (launch_copy_words((const u64*)hp, (u64*)dev, words, s) != hipSuccess) return nullptr;
    tables_used = true;
    return dev;
  };
  auto stage_table = [&](const void* src, size_t bytes) -> const void* {
    unsigned char* hp = stage_begin(bytes);
    std::memcpy(hp, src, bytes);
    return stage_commit(hp, bytes);
  };

  auto galois_key = [&](u32 elt) -> KeySel { return keys.galois_sel(elt); };
  // rotate `in` by `steps` into `out` following SEAL's rotate_internal (direct key or NAF chain)
  std::function<int(const u64*, int, u64*)> rotate = [&](const u64* in, int steps, u64* out) -> int {
    if (steps == 0) {
      if (in != out && hipMemcpyAsync(out, in, batch * 2 * poly * sizeof(u64), hipMemcpyDeviceToDevice, s) != hipSuccess) return kHipError;
      return kOk;
    }
    const u32 elt = ev.galois_elt_from_step(steps);
    if (!elt) return kInvalidArg;
    if (const KeySel key = galois_key(elt); key.present()) return ev.apply_galois(in, elt, key, out, batch, s);
    std::vector<int> naf;
    const bool neg = steps < 0;
    int v = neg ? -steps : steps;
    for (int i = 0; v; i++) {
      const int zi = (v & 1) ? 2 - (v & 3) : 0;
      v = (v - zi) >> 1;
      if (zi) naf.push_back((neg ? -zi : zi) * (1 << i));
    }
    if (naf.size() == 1) return kNoKey;
    const u64* cur = in;
    for (int part : naf) {
      if ((size_t)(part < 0 ? -part : part) == (n >> 1)) continue;
      int rc = rotate(cur, part, out);
      if (rc) return rc;
      cur = out;
    }
    return kOk;
  };

  // operands of a merged launch as ONE array: used in place when the members' buffers are adjacent and in order, staged otherwise
  auto as_array = [&](const std::vector<int>& slots, size_t words_each, const u64** out) -> int {
    bool adjacent = true;
    for (size_t m = 1; m < slots.size() && adjacent; m++) adjacent = sp[slots[m]] == sp[slots[0]] + m * words_each;
    if (adjacent) {
      *out = sp[slots[0]];
      return kOk;
    }
    std::vector<const u64*> tab(slots.size());
    for (size_t m = 0; m < slots.size(); m++) tab[m] = sp[slots[m]];
    const u64* const* dtab = (const u64* const*)stage_table(tab.data(), tab.size() * sizeof(u64*));
    u64* stage = (u64*)pool.acquire(slots.size() * words_each * sizeof(u64), s);
    if (!dtab || !stage) return kOutOfMemory;
    temps.push_back(stage);
    for (size_t off = 0; off < slots.size(); off += 65535) {
      const size_t c = std::min<size_t>(65535, slots.size() - off);
      if (launch_gather_items(dtab + off, stage + off * words_each, words_each, c, s) != hipSuccess) return kHipError;
    }
    *out = stage;
    return kOk;
  };

  // which sums were folded into their producer at run time: (step, member) -> the slot that carries the result
  std::vector<std::vector<int>> folded_into(P.steps.size());
  auto try_fold = [&](const Plan::Step& st, size_t m) -> const u64* {  // the addend to hand to the key-switching kernel, or nullptr
    if (st.fold.empty() || st.fold[m].other_slot < 0) return nullptr;
    return sp[st.fold[m].other_slot];
  };
  auto mark_folded = [&](const Plan::Step& st, size_t m) {
    const Plan::Fold& f = st.fold[m];
    auto& v = folded_into[f.nary_step];
    if (v.empty()) v.assign(P.steps[f.nary_step].node.size(), -1);
    v[f.nary_member] = st.out[m];
  };
  // the buffer a key-switching member writes when its Add is folded: the sum's own destination
  auto alloc_for_fold = [&](const Plan::Step& st, size_t m) -> u64* {
    const Plan::Fold& f = st.fold[m];
    const int sum_slot = P.steps[f.nary_step].out[f.nary_member];
    u64* p = alloc_member(sum_slot);  // binds the SUM's slot; the member's own slot aliases it
    if (!p) return nullptr;
    bind(st.out[m], block_of[sum_slot], p);
    return p;
  };

  const bool small = batch <= merge_max_batch();
  // HIPBFV_PROGRAM_TRACE=1: drain the stream after every step and print where the time goes (diagnostics only)
  const char* trace_env = std::getenv("HIPBFV_PROGRAM_TRACE");
  const bool trace = trace_env && trace_env[0] == '1';
  auto now_us = [] {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
  };
  static const char* step_names[] = {"sum", "plain_matrix", "mul_relin", "multiply", "relinearize", "rotate", "plain_op", "output"};
  // ---- r06: the plaintext-matrix product on a side stream (examples/pir) ----
  // The product of a LinComb step streams the database at 6+ TB/s and hardly uses the vector units; the merged multiply +
  // relinearize that consumes its rows is bound by them.  When that is the very next step, the product runs on a SIDE stream in
  // row chunks and the main stream takes each chunk's rows through the multiply as soon as its event has passed: chunk c + 1 of
  // the product streams while chunk c is multiplied (measured with the batch primitives in r05: +2...4 %, same bits --
  // tools/pir_overlap_probe.py).  Anything else that follows waits for every chunk first (`side_flush`).
  // OPT-IN (HIPBFV_PIR_OVERLAP=1, read per run): measured inside this executor it LOSES -- examples/pir, 512 x 256 entries at
  // n = 16384, interleaved on one box (profiles/r06_s11_ab_pir_overlap*.txt): serial 3.93 M entries/s, 2 chunks 3.84 M, 4 chunks
  // 3.75 M, 8 chunks 3.48 M.  The probe's +2...4 % was against a CHUNKED serial schedule; the executor's serial schedule is one
  // 512-row product and one 512-member multiply, and next to a concurrent multiply the product stream slows down by more (22.2 ->
  // 24.6 ms) than the multiply hides.  Kept exercised by tests/test_gpu_program.py.
  const bool side_enabled = [] {
    const char* e = std::getenv("HIPBFV_PIR_OVERLAP");
    return e && e[0] == '1';
  }();
  const bool member_tails = [] {  // HIPBFV_NO_MEMBER_TAILS=1: the sums around fused products as launches of their own (the cross-check arm)
    const char* e = std::getenv("HIPBFV_NO_MEMBER_TAILS");
    return !(e && e[0] == '1');
  }();
  const bool merge_products = [] {  // HIPBFV_NO_MERGED_PRODUCTS=1: merged product launches below merge_max_batch() only (r03 ... r06 s26)
    const char* e = std::getenv("HIPBFV_NO_MERGED_PRODUCTS");
    return !(e && e[0] == '1');
  }() || small;
  auto side_stream = [&]() -> hipStream_t {
    thread_local hipStream_t streams[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipStream_t& ss = streams[dev & 15];
    if (!ss && hipStreamCreateWithFlags(&ss, hipStreamNonBlocking) != hipSuccess) ss = nullptr;
    return ss;
  };
  auto side_drop = [&]() {
    for (hipEvent_t e : side.done) (void)hipEventDestroy(e);
    side = SideProduct{};
  };
  // the main stream waits for every outstanding chunk (a consumer other than the pipelined multiply, or the end of the run)
  auto side_flush = [&]() -> int {
    if (!side.base) return kOk;
    int rc2 = kOk;
    for (hipEvent_t e : side.done)
      if (hipStreamWaitEvent(s, e, 0) != hipSuccess) rc2 = kHipError;
    if (!rc2 && !side.noted) rc2 = ev.note_result(side.base, 2, (u32)K, side.rows * batch, s);
    side_drop();
    return rc2;
  };
  for (size_t si = 0; si < P.steps.size(); si++) {
    const Plan::Step& st = P.steps[si];
    const size_t members = st.node.size();
    int rc = kOk;
    // the one consumer that takes the side product chunk by chunk: a merged multiply + relinearize right behind it (below)
    bool side_consumer = side.base && st.kind == kStepMulRelin && small && members > 1 && members == side.rows && !st.square;
    if (side_consumer) {  // member m's left (or right) operand is row m of the product, in place
      bool as_a = true, as_b = true;
      for (size_t m = 0; m < members; m++) {
        as_a = as_a && sp[st.a[m]] == side.base + m * batch * 2 * poly;
        as_b = as_b && sp[st.b[m]] == side.base + m * batch * 2 * poly;
      }
      side_consumer = as_a || as_b;
    }
    if (side.base && !side_consumer)
      if ((rc = side_flush())) return cleanup(rc, "operation failed");
    const double t_begin = trace ? now_us() : 0.0;
    struct TraceEnd {
      bool on;
      double t0;
      hipStream_t s;
      const char* name;
      size_t members;
      decltype(now_us)& now;
      ~TraceEnd() {
        if (!on) return;
        const double t1 = now();
        (void)hipStreamSynchronize(s);
        const double t2 = now();
        fprintf(stderr, "[program] %-12s members=%zu host %.0f us, drained after %.0f us\n", name, members, t1 - t0, t2 - t0);
      }
    } trace_end{trace, t_begin, s, step_names[st.kind], members, now_us};
    switch (st.kind) {
      case kStepNary: {
        // A long sum (examples/pir: 256 products into one ciphertext) is one output per thread column: its terms are added
        // one after the other.  Sums of more than kLongSum terms go in two levels -- partial sums of kSumChunk terms, which
        // are many independent outputs, then the sum of the partials (modular addition is associative: same bits).
        constexpr u32 kLongSum = 32, kSumChunk = 16;
        std::vector<NaryOut> pre_outs, outs;
        std::vector<NaryTerm> pre_terms, terms;
        u32 max_size = 0, pre_max = 0;
        const std::vector<int>& fv = folded_into[si];
        for (size_t m = 0; m < members; m++) {
          if (!fv.empty() && fv[m] >= 0) continue;  // its producer already added the other operand and wrote this slot
          const int osl = st.out[m];
          u64* o = alloc_member(osl);
          if (!o) return cleanup(kOutOfMemory, "out of device memory");
          const u32 count = st.first[m + 1] - st.first[m], osz = P.slot_size[osl];
          NaryOut d;
          d.out = o;
          d.first = (u32)terms.size();
          d.size = osz;
          d.pad = 0;
          max_size = std::max(max_size, osz);
          if (count > kLongSum) {
            const u32 chunks = (count + kSumChunk - 1) / kSumChunk;
            u64* part = (u64*)pool.acquire((size_t)chunks * batch * osz * poly * sizeof(u64), s);
            if (!part) return cleanup(kOutOfMemory, "out of device memory");
            temps.push_back(part);
            pre_max = std::max(pre_max, osz);
            for (u32 c = 0; c < chunks; c++) {
              NaryOut pd;
              pd.out = part + (size_t)c * batch * osz * poly;
              pd.first = (u32)pre_terms.size();
              pd.count = std::min(kSumChunk, count - c * kSumChunk);
              pd.size = osz;
              pd.pad = 0;
              for (u32 t = 0; t < pd.count; t++) {
                const Term& tm = st.terms[st.first[m] + c * kSumChunk + t];
                pre_terms.push_back(NaryTerm{sp[tm.slot], P.slot_size[tm.slot], tm.sign});
              }
              pre_outs.push_back(pd);
              terms.push_back(NaryTerm{pd.out, osz, 1});
            }
            d.count = chunks;
          } else {
            d.count = count;
            for (u32 t = st.first[m]; t < st.first[m + 1]; t++) terms.push_back(NaryTerm{sp[st.terms[t].slot], P.slot_size[st.terms[t].slot], st.terms[t].sign});
          }
          outs.push_back(d);
        }
        if (outs.empty()) break;
        if (!pre_outs.empty()) {
          const NaryOut* dpo = (const NaryOut*)stage_table(pre_outs.data(), pre_outs.size() * sizeof(NaryOut));
          const NaryTerm* dpt = (const NaryTerm*)stage_table(pre_terms.data(), pre_terms.size() * sizeof(NaryTerm));
          if (!dpo || !dpt) return cleanup(kOutOfMemory, "descriptor table allocation failed");
          ev.profiler().begin(kKernEltwise, pre_outs.size() * batch * pre_max * K, s);
          const hipError_t e = launch_nary_sum(ctx->dev(), (u32)n, (u32)K, dpo, dpt, (u32)pre_outs.size(), pre_max, (u32)batch, s);
          ev.profiler().end(s);
          if (e != hipSuccess) return cleanup(kHipError, "operation failed");
        }
        const NaryOut* douts = (const NaryOut*)stage_table(outs.data(), outs.size() * sizeof(NaryOut));
        const NaryTerm* dterms = (const NaryTerm*)stage_table(terms.data(), terms.size() * sizeof(NaryTerm));
        if (!douts || !dterms) return cleanup(kOutOfMemory, "descriptor table allocation failed");
        ev.profiler().begin(kKernEltwise, outs.size() * batch * max_size * K, s);
        const hipError_t e = launch_nary_sum(ctx->dev(), (u32)n, (u32)K, douts, dterms, (u32)outs.size(), max_size, (u32)batch, s);
        ev.profiler().end(s);
        if (e != hipSuccess) return cleanup(kHipError, "operation failed");
        rc = ev.note_nary(douts, (u32)outs.size(), (u32)batch, s);
        break;
      }
      case kStepMulRelin:
      case kStepRelin:
      case kStepGalois: {
        const bool is_mul = st.kind == kStepMulRelin, is_rot = st.kind == kStepGalois;
        if (!is_rot && !relin_key.present() && !(st.kind == kStepRelin && st.out_size == 2)) return cleanup(kNoKey, "operation failed");
        u32 elt = 0;
        KeySel gkey;
        if (is_rot) {
          if (!ctx->batching()) return cleanup(kUnsupported, "encryption parameters do not support batching");
          elt = st.swap ? 2 * (u32)n - 1 : (st.rot_steps ? ev.galois_elt_from_step(st.rot_steps) : 0);
          if (st.swap || st.rot_steps) {
            if (!elt) return cleanup(kInvalidArg, "operation failed");
            gkey = galois_key(elt);
            if (st.swap && !gkey.present()) return cleanup(kNoKey, "Galois key for the column rotation is missing");
          }
        }
        const u32 in_size = st.kind == kStepRelin ? st.out_size : 2;
        const size_t in_words = batch * in_size * poly, out_words = batch * 2 * poly;
        const bool direct_ks = is_mul || (st.kind == kStepRelin && in_size == 3) || (is_rot && gkey.present());
        // r06: products with a linear fold (Plan::LinFold) or a caller's buffer to write take each member's destination, multiplier and
        // signed addend from a table in their last kernel (kernels.hpp MemberTail): the sums around the products and the copies into the
        // outputs cost no pass of their own (examples/chi_sq: 2 x^2, 2 y^2, 4 n0 n2 - n1^2, x y).  Members [m0, m1) in one launch.
        auto has_lin = [&](size_t m) { return !st.lin.empty() && st.lin[m].mult != 0; };
        auto tail_launch = [&](size_t m0, size_t m1) -> int {
          const size_t cnt = m1 - m0;
          std::vector<MemberHead> heads(cnt);  // the operands where they are (no gather into one array)
          for (size_t m = m0; m < m1; m++) heads[m - m0] = MemberHead{sp[st.a[m]], sp[st.square ? st.a[m] : st.b[m]]};
          std::vector<MemberTail> tab(cnt);
          for (size_t m = m0; m < m1; m++) {
            const Plan::LinFold* lf = has_lin(m) ? &st.lin[m] : nullptr;
            const int target = lf ? P.steps[lf->nary_step].out[lf->nary_member] : st.out[m];
            u64* o = alloc_member(target);  // the caller's buffer when the value is a program output
            if (!o) return (int)kOutOfMemory;
            if (lf) {
              bind(st.out[m], block_of[target], o);  // (nothing reads the bare product: every use of it is inside the sum)
              auto& v = folded_into[lf->nary_step];
              if (v.empty()) v.assign(P.steps[lf->nary_step].node.size(), -1);
              v[lf->nary_member] = st.out[m];
            }
            const bool has_other = lf && lf->other_slot >= 0;
            tab[m - m0] = MemberTail{o, has_other ? sp[lf->other_slot] : nullptr, lf ? (u32)lf->mult : 1u, has_other ? lf->sign : 0};
          }
          const MemberTail* dtab = (const MemberTail*)stage_table(tab.data(), tab.size() * sizeof(MemberTail));
          const MemberHead* dheads = (const MemberHead*)stage_table(heads.data(), heads.size() * sizeof(MemberHead));
          if (!dtab || !dheads) return (int)kOutOfMemory;
          KeySel sub = relin_key;  // (member-major item numbering: member m0's items start at m0 * batch)
          sub.first = relin_key.first + m0 * batch;
          int r = ev.multiply_relin(nullptr, nullptr, sub, nullptr, cnt * batch, s, nullptr, dtab, (u32)batch, dheads, st.square);
          for (size_t m = 0; m < cnt && !r; m++) r = ev.note_result(tab[m].out, 2, (u32)K, batch, s);
          return r;
        };
        const bool tails_on = is_mul && !side_consumer && member_tails;
        // With both tables a merged launch costs nothing to set up (operands and results stay where they are), so the ready products
        // of a round run as ONE launch sequence at every batch size, not only below merge_max_batch(): examples/chi_sq at one GPU's
        // share of 128 sets runs 384 + 256 + 128 items instead of six launches of 128.
        if (tails_on && members > 1 && members <= 64 && merge_products && ev.member_tail_ok(members * batch)) {
          rc = tail_launch(0, members);
          break;
        }
        if (small && members > 1 && direct_ks) {
          // ONE launch sequence over members x batch ciphertexts
          const u64 *A = nullptr, *B2 = nullptr;
          if ((rc = as_array(st.a, in_words, &A))) return cleanup(rc, "operation failed");
          if (is_mul && !st.square && (rc = as_array(st.b, in_words, &B2))) return cleanup(rc, "operation failed");
          const int blk = new_block(members * out_words);
          if (blk < 0) return cleanup(kOutOfMemory, "out of device memory");
          u64* out = (u64*)blocks[blk].ptr;
          for (size_t m = 0; m < members; m++) bind(st.out[m], blk, out + m * out_words);
          const size_t count = members * batch;
          if (side_consumer && is_mul) {
            // rows of the side-stream product, chunk by chunk: the main stream waits for chunk c's event only, so the product's
            // later chunks stream while this one is multiplied
            size_t r0 = 0;
            for (size_t c = 0; c < side.done.size() && !rc; c++) {
              const size_t r1 = side.row_end[c];
              if (hipStreamWaitEvent(s, side.done[c], 0) != hipSuccess) rc = kHipError;
              KeySel sub = relin_key;
              sub.first = relin_key.first + r0 * batch;
              if (!rc) rc = ev.multiply_relin(A + r0 * batch * 2 * poly, B2 + r0 * batch * 2 * poly, sub, out + r0 * batch * 2 * poly, (r1 - r0) * batch, s);
              r0 = r1;
            }
            if (!rc) rc = ev.note_result(side.base, 2, (u32)K, side.rows * batch, s);
            side.noted = true;
            side_drop();
            break;
          }
          rc = is_mul ? ev.multiply_relin(A, st.square ? A : B2, relin_key, out, count, s)
             : is_rot ? ev.apply_galois(A, elt, gkey, out, count, s)
                      : ev.relinearize(A, relin_key, out, count, s);
          break;
        }
        for (size_t m = 0; m < members && !rc; m++) {
          if (tails_on && has_lin(m) && !(small && members > 1) && ev.member_tail_ok(batch)) {  // (a merged launch has no tables: whole or not at all)
            rc = tail_launch(m, m + 1);
            continue;
          }
          const u64* addend = direct_ks ? try_fold(st, m) : nullptr;
          u64* out = addend ? alloc_for_fold(st, m) : alloc_member(st.out[m]);
          if (!out) return cleanup(kOutOfMemory, "out of device memory");
          const u64* a = sp[st.a[m]];
          if (is_mul)
            rc = ev.multiply_relin(a, sp[st.b[m]], relin_key, out, batch, s, addend);
          else if (st.kind == kStepRelin && in_size == 2)
            rc = hipMemcpyAsync(out, a, out_words * sizeof(u64), hipMemcpyDeviceToDevice, s) == hipSuccess ? (int)kOk : (int)kHipError;
          else if (st.kind == kStepRelin)
            rc = ev.relinearize(a, relin_key, out, batch, s, addend);
          else if (gkey.present())
            rc = ev.apply_galois(a, elt, gkey, out, batch, s, addend);
          else
            rc = rotate(a, st.rot_steps, out);
          if (addend && !rc) mark_folded(st, m);
        }
        break;
      }
      case kStepMultiply: {
        u64* out = alloc_member(st.out[0]);
        if (!out) return cleanup(kOutOfMemory, "out of device memory");
        rc = ev.multiply(sp[st.a[0]], P.slot_size[st.a[0]], sp[st.b[0]], P.slot_size[st.b[0]], out, batch, s);
        break;
      }
      case kStepPlainOp: {
        PlainVal pv;
        if (const char* m = plain_of(st.plain[0], &pv)) return cleanup(kInvalidArg, m);
        const u32 sz = P.slot_size[st.a[0]];
        if (pv.kind == 2 && st.plain_op != kOpMultiplyPlaintext)
          return cleanup(kInvalidArg, "a transform-domain plaintext argument can only be an operand of MultiplyPlaintext");
        u64* out = alloc_member(st.out[0]);
        if (!out) return cleanup(kOutOfMemory, "out of device memory");
        if (st.plain_op == kOpAddPlaintext)
          rc = ev.add_plain(sp[st.a[0]], sz, pv.ptr, pv.stride, out, batch, s);
        else if (st.plain_op == kOpSubPlaintext)
          rc = ev.sub_plain(sp[st.a[0]], sz, pv.ptr, pv.stride, out, batch, s);
        else if (pv.kind == 1)
          rc = ev.multiply_plain(sp[st.a[0]], sz, pv.ptr, pv.stride, out, batch, s);
        else
          rc = ev.multiply_plain_ntt(sp[st.a[0]], sz, pv.ptr, pv.stride, out, batch, s);
        break;
      }
      case kStepLinComb: {
        // out[row] = sum_j cts[j] (.) plain[row][j]: transform the ciphertexts once, accumulate in the transform domain, one inverse per row
        const size_t cols = st.cts.size(), rows = members;
        const size_t ct_words = batch * 2 * poly;
        // The ciphertexts first: their copy and transforms need no table, so the device works on them while the host builds the
        // plaintext table below (65 536 descriptors for a 256 x 256 database: ~0.15 ms during which the device used to idle).
        const u64* staged = nullptr;
        {
          // always a copy: the transform runs in place.  Adjacent operands are copied with one memcpy.
          bool adjacent = true;
          for (size_t j = 1; j < cols && adjacent; j++) adjacent = sp[st.cts[j]] == sp[st.cts[0]] + j * ct_words;
          u64* ctn = (u64*)pool.acquire(cols * ct_words * sizeof(u64), s);
          if (!ctn) return cleanup(kOutOfMemory, "out of device memory");
          temps.push_back(ctn);
          if (adjacent) {
            if (hipMemcpyAsync(ctn, sp[st.cts[0]], cols * ct_words * sizeof(u64), hipMemcpyDeviceToDevice, s) != hipSuccess) return cleanup(kHipError, "copy failed");
          } else {
            std::vector<const u64*> ctab(cols);
            for (size_t j = 0; j < cols; j++) ctab[j] = sp[st.cts[j]];
            const u64* const* dctab = (const u64* const*)stage_table(ctab.data(), cols * sizeof(u64*));
            if (!dctab) return cleanup(kOutOfMemory, "descriptor table allocation failed");
            for (size_t off = 0; off < cols; off += 65535) {
              const size_t c = std::min<size_t>(65535, cols - off);
              if (launch_gather_items(dctab + off, ctn + off * ct_words, ct_words, c, s) != hipSuccess) return cleanup(kHipError, "operation failed");
            }
          }
          if ((rc = ev.ct_to_ntt(ctn, 2, ctn, cols * batch, s))) return cleanup(rc, "operation failed");
          staged = ctn;
        }
        // Then the plaintexts (host work, then one table upload).  Transform-domain arguments are used as they are; coefficient-form ones
        // are lifted and transformed now (runs of adjacent plaintexts in one call), and an all-zero one raises the
        // transparent-result failure SEAL's multiply_plain raises.
        const size_t entries = rows * cols;
        PlainNttRef* const tab = reinterpret_cast<PlainNttRef*>(stage_begin(entries * sizeof(PlainNttRef)));  // built in place in the pinned arena
        size_t need_ntt = 0;
        for (size_t e = 0; e < entries; e++) {
          const Node& nd = nodes_[st.plain[e]];
          if (nd.op == kOpInputPlaintext && nd.arg < num_inputs && inputs[nd.arg].kind == 2 && inputs[nd.arg].ptr) {  // the common case, kept tight
            tab[e] = PlainNttRef{inputs[nd.arg].ptr, (u64)inputs[nd.arg].stride};
            continue;
          }
          PlainVal v;
          if (const char* m = plain_of(st.plain[e], &v)) return cleanup(kInvalidArg, m);
          tab[e] = PlainNttRef{nullptr, 0};  // filled below
          need_ntt += v.stride ? batch : 1;
        }
        if (need_ntt) {
          u64* pscratch = (u64*)pool.acquire(need_ntt * poly * sizeof(u64), s);
          if (!pscratch) return cleanup(kOutOfMemory, "out of device memory");
          temps.push_back(pscratch);
          size_t pos = 0;
          for (size_t e = 0; e < entries;) {
            if (tab[e].ptr) {
              e++;
              continue;
            }
            PlainVal v;
            (void)plain_of(st.plain[e], &v);
            if (v.stride) {  // per-item plaintexts u64[batch][N]
              if ((rc = ev.plain_to_ntt(v.ptr, v.stride, pscratch + pos * poly, batch, s, 1))) return cleanup(rc, "operation failed");
              tab[e] = PlainNttRef{pscratch + pos * poly, (u64)poly};
              pos += batch;
              e++;
              continue;
            }
            size_t run = 1;  // shared plaintexts that sit next to each other in memory: one lift + transform launch for the run
            while (e + run < entries && !tab[e + run].ptr) {
              PlainVal w;
              (void)plain_of(st.plain[e + run], &w);
              if (w.stride || w.ptr != v.ptr + run * n) break;
              run++;
            }
            if ((rc = ev.plain_to_ntt(v.ptr, n, pscratch + pos * poly, run, s, 2))) return cleanup(rc, "operation failed");
            for (size_t r = 0; r < run; r++) tab[e + r] = PlainNttRef{pscratch + (pos + r) * poly, 0};
            pos += run;
            e += run;
          }
        }
        const PlainNttRef* dtab = (const PlainNttRef*)stage_commit(reinterpret_cast<const unsigned char*>(tab), entries * sizeof(PlainNttRef));
        if (!dtab) return cleanup(kOutOfMemory, "descriptor table allocation failed");
        const int blk = new_block(rows * ct_words);
        if (blk < 0) return cleanup(kOutOfMemory, "out of device memory");
        u64* out = (u64*)blocks[blk].ptr;
        for (size_t m = 0; m < rows; m++) bind(st.out[m], blk, out + m * ct_words);
        {
          // on the side stream, in row chunks, when the next step is the merged multiply + relinearize of these rows
          bool pipelined = false;
          if (side_enabled && small && rows >= 64 && si + 1 < P.steps.size()) {
            const Plan::Step& nx = P.steps[si + 1];
            bool consumes = nx.kind == kStepMulRelin && nx.node.size() == rows && !nx.square;
            if (consumes) {  // member m of the multiply takes row m of this product, as its left or its right operand, rows in order
              bool as_a = true, as_b = true;
              for (size_t m = 0; m < rows; m++) as_a = as_a && nx.a[m] == st.out[m], as_b = as_b && nx.b[m] == st.out[m];
              consumes = as_a || as_b;
            }
            hipStream_t ss = consumes ? side_stream() : nullptr;
            if (ss) {
              const size_t kChunks = [] { const char* e = std::getenv("HIPBFV_PIR_CHUNKS"); const long v = e ? atol(e) : 2; return (size_t)(v >= 1 && v <= 64 ? v : 2); }();
              hipEvent_t ready = nullptr;
              bool ok = hipEventCreateWithFlags(&ready, hipEventDisableTiming) == hipSuccess && hipEventRecord(ready, s) == hipSuccess &&
                        hipStreamWaitEvent(ss, ready, 0) == hipSuccess;  // the staged ciphertexts and the table were enqueued on `s`
              if (ready) (void)hipEventDestroy(ready);
              if (ok) {
                side.base = out;
                side.rows = rows;
                for (size_t c = 0; c < kChunks && ok; c++) {
                  const size_t r0 = rows * c / kChunks, r1 = rows * (c + 1) / kChunks;
                  if ((rc = ev.dot_plain_tab(staged, (u32)cols, dtab + r0 * cols, (u32)(r1 - r0), (u32)batch, out + r0 * ct_words, ss))) break;
                  hipEvent_t e = nullptr;
                  ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess && hipEventRecord(e, ss) == hipSuccess;
                  if (e) side.done.push_back(e);
                  side.row_end.push_back(r1);
                }
                if (rc || !ok) {  // launched chunks finish on the side stream; wait for them here and fail the run
                  (void)hipStreamSynchronize(ss);
                  return cleanup(rc ? rc : (int)kHipError, "operation failed");
                }
                pipelined = true;
              }
            }
          }
          if (pipelined) break;
        }
        if ((rc = ev.dot_plain_tab(staged, (u32)cols, dtab, (u32)rows, (u32)batch, out, s))) return cleanup(rc, "operation failed");
        rc = ev.note_result(out, 2, (u32)K, rows * batch, s);
        break;
      }
      case kStepOutput: {
        if (direct_written[st.out_index]) break;
        if (hipMemcpyAsync(outputs[st.out_index], sp[st.a[0]], batch * 2 * poly * sizeof(u64), hipMemcpyDeviceToDevice, s) != hipSuccess)
          return cleanup(kHipError, "copy failed");
        break;
      }
      default:
        return cleanup(kInvalidArg, "unsupported operation");
    }
    if (rc) return cleanup(rc, st.kind == kStepMulRelin ? "multiply+relinearize failed" : "operation failed");
    for (int sl : st.release) unbind(sl);
  }
  if (int rc2 = side_flush()) return cleanup(rc2, "operation failed");
  for (Block& b : blocks)
    if (b.owned && b.refs > 0) pool.release(b.ptr, s);
  for (void* t : temps) pool.release(t, s);
  // the descriptor tables were copied from this thread's pinned arena: an event behind the copies guards its reuse (two runs from
  // now: TableArenas) -- the run itself does not wait for the device
  if (tables_used && !arena.mark(s)) return fail(kHipError, "stream synchronisation failed");
  return kOk;
}

}  // namespace hipbfv
