// sunscreen_amd/csrc/program.hpp -- batched execution of FHE program graphs (see program.cpp).
#pragma once
#include <cstdint>
#include <functional>
#include <map>
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
// u64[batch][N] (stride N) or one shared plaintext (stride 0).  Ciphertext and plaintext arguments share
// one index space, as in run.rs:160-172.
struct ProgramInput {
  int kind;
  const u64* ptr;
  size_t stride;
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
  int run(Evaluator& ev, size_t batch, const ProgramInput* inputs, size_t num_inputs, const u64* relin_key,
          const std::map<u32, const u64*>& galois_keys, u64* const* outputs, size_t num_outputs, hipStream_t s,
          std::string* err) const;

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
  std::vector<Node> nodes_;
  std::vector<PlainLiteral> literals_;
};

}  // namespace hipbfv
