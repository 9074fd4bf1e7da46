// sunscreen_amd/csrc/program.hpp -- batched execution of FHE program graphs (see program.cpp).
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "evaluator.hpp"

namespace hipbfv {

// node kinds of sunscreen_fhe_program::Operation (operation.rs:12-94)
enum OpKind : int {
  kOpShiftLeft = 0,
  kOpShiftRight,
  kOpSwapRows,
  kOpRelinearize,
  kOpMultiply,
  kOpMultiplyPlaintext,
  kOpAdd,
  kOpAddPlaintext,
  kOpNegate,
  kOpSub,
  kOpSubPlaintext,
  kOpInputCiphertext,   // arg = argument index
  kOpInputPlaintext,    // arg = argument index
  kOpLiteralU64,        // arg = value
  kOpOutputCiphertext,
  kOpLiteralPlaintext,  // arg = index into the program's literal table (Literal::Plaintext, literal.rs:8-18)
  kOpCount
};

enum EdgeKind : int { kEdgeLeft = 0, kEdgeRight = 1, kEdgeUnary = 2 };

// One program argument for a batch: kind 0 = ciphertexts u64[batch][2][K][N], kind 1 = plaintexts
// u64[batch][N] (stride N) or one shared plaintext (stride 0), kind 2 = plaintexts already in transform form
// u64[batch][K][N] (stride K*N) or shared (stride 0) -- the output of hipbfv_batch_plain_to_ntt: a server's static data
// (examples/pir's database) is lifted and transformed once, not once per query; only MultiplyPlaintext consumes it.
// Ciphertext and plaintext arguments share one index space, as in run.rs:160-172.
struct ProgramInput {
  int kind;
  const u64* ptr;
  size_t stride;
};

// The keys of one run.  The reference's call takes ONE set (run.rs:100-105: `relin_keys`, `galois_keys`); a server that batches the
// runs of many clients holds one set per client, and input set i of the batch uses set index[i].  index == nullptr: set 0.
struct ProgramKeys {
  std::vector<const u64*> relin;                   // per key set: the relinearisation key, nullptr if the set has none
  std::vector<std::map<u32, const u64*>> galois;   // per key set: (galois_elt - 1) / 2 -> key
  const u32* index = nullptr;                      // HOST, `period` entries
  size_t period = 0;  // = the run's batch (merged launches number their items member * batch + set)
  KeySel relin_sel() const;
  // the key of `elt` in every set; !present() when any set lacks it (the rotation then takes SEAL's NAF chain, as with one set)
  KeySel galois_sel(u32 elt) const;

 private:
  mutable std::map<u32, std::vector<const u64*>> tables_;  // host pointer tables handed out by galois_sel (alive for the run)
};

class Program {
 public:
  int add_node(OpKind op, u64 arg);
  // Literal::Plaintext(bytes): bytes = bincode(InnerPlaintext::Seal([WithContext{Params, SEAL-serialised Plaintext}]))
  // exactly as the compiler stores it (sunscreen/src/fhe/mod.rs:370-376, decoded at sunscreen_runtime/src/run.rs:312-328)
  int add_plaintext_literal(const uint8_t* bytes, size_t len, std::string* err);
  int add_edge(int src, int dst, EdgeKind kind);
  int load_json(const char* text, size_t len, std::string* err);
  int validate(std::string* err) const;
  size_t num_nodes() const { return nodes_.size(); }
  size_t num_outputs() const;
  // outputs: one device buffer u64[batch][2][K][N] per OutputCiphertext node, in node-index order (run.rs:343-356)
  // The scheduled executor (program_plan.cpp) unless HIPBFV_PROGRAM_SERIAL=1 selects the node-by-node one (program.cpp).
  int run(Evaluator& ev, size_t batch, const ProgramInput* inputs, size_t num_inputs, const ProgramKeys& keys,
          u64* const* outputs, size_t num_outputs, hipStream_t s,
          std::string* err) const;
  // one node at a time in topological order on one stream (round 1 / 2's executor; kept as the A/B and cross-check arm)
  int run_serial(Evaluator& ev, size_t batch, const ProgramInput* inputs, size_t num_inputs, const ProgramKeys& keys,
                 u64* const* outputs, size_t num_outputs, hipStream_t s,
                 std::string* err) const;
  // Level-scheduled execution: every node that is ready runs in the same round (run.rs:372-472 runs them concurrently on
  // rayon); ready nodes of one kind become ONE batched launch, Add / Sub / Negate trees become n-ary sums, sums of
  // ciphertext-plaintext products stay in the transform domain.
  int run_plan(Evaluator& ev, size_t batch, const ProgramInput* inputs, size_t num_inputs, const ProgramKeys& keys,
               u64* const* outputs, size_t num_outputs, hipStream_t s,
               std::string* err) const;
  struct Plan;
  // one line per step of the schedule: "<kind> members=<m> ..." (diagnostics; tests/test_program_plan_cpu.py reads it)
  int describe(std::string* out) const;

 private:
  struct Node {
    OpKind op;
    u64 arg;
    int left, right;  // operand node ids (unary operand in `left`)
  };
  struct PlainLiteral {
    u64 n, t;
    std::vector<u64> primes;  // Params::coeff_modulus (key-level primes)
    std::vector<u64> coeffs;
  };
  bool topo_order(std::vector<int>* order) const;
  std::shared_ptr<const Plan> plan() const;  // built on first use, dropped whenever the graph changes
  void drop_plan();
  std::vector<Node> nodes_;
  std::vector<PlainLiteral> literals_;
  mutable std::mutex plan_mu_;
  mutable std::shared_ptr<const Plan> plan_;
};

}  // namespace hipbfv
