// sunscreen_amd/csrc/program.cpp -- the GPU batch executor for compiled FHE program graphs.
//
// Replaces the reference's graph interpreter (sunscreen_runtime/src/run.rs:100-357,
// `run_program_unchecked` + the rayon `traverse` at run.rs:372-472): instead of one evaluator FFI
// call per node per ciphertext, one graph is executed over `batch` independent input sets, one
// sequence of kernel launches per node, with every intermediate ciphertext device-resident.
// The graph uses the reference's node kinds (sunscreen_fhe_program/src/operation.rs:12-94) and edge
// kinds Left/Right/Unary (sunscreen_compiler_common/src/context.rs:60-85); `load_json` accepts the serde
// JSON form of `FheProgram` (petgraph StableGraph: {"nodes":[{"operation":..}],"edges":[[src,dst,"Left"]]}).
#include "program.hpp"

#include "wire.hpp"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>

namespace hipbfv {

namespace {

// ---------------------------------------------------------------- minimal JSON reader
struct JValue {
  enum Kind { kNull, kBool, kNum, kStr, kArr, kObj } kind = kNull;
  bool b = false;
  double num = 0;
  unsigned long long unum = 0;
  std::string str;
  std::vector<JValue> arr;
  std::vector<std::pair<std::string, JValue>> obj;
  const JValue* get(const char* key) const {
    for (auto& kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
};

struct JParser {
  const char* p;
  const char* end;
  bool ok = true;
  int depth = 0;
  static constexpr int kMaxDepth = 64;  // serde's FheProgram JSON nests 5 deep; the parser recurses, so untrusted input is bounded
  void ws() {
    while (p < end && std::isspace((unsigned char)*p)) p++;
  }
  bool lit(const char* s) {
    size_t n = std::strlen(s);
    if ((size_t)(end - p) >= n && std::strncmp(p, s, n) == 0) {
      p += n;
      return true;
    }
    return false;
  }
  JValue parse() {
    JValue v;
    if (++depth > kMaxDepth) {
      ok = false;
      --depth;
      return v;
    }
    v = parse_value();
    --depth;
    return v;
  }
  JValue parse_value() {
    JValue v;
    ws();
    if (p >= end) {
      ok = false;
      return v;
    }
    if (*p == '{') {
      v.kind = JValue::kObj;
      p++;
      ws();
      if (p < end && *p == '}') {
        p++;
        return v;
      }
      while (ok) {
        ws();
        JValue k = parse();
        if (k.kind != JValue::kStr) {
          ok = false;
          break;
        }
        ws();
        if (p >= end || *p != ':') {
          ok = false;
          break;
        }
        p++;
        v.obj.emplace_back(k.str, parse());
        ws();
        if (p < end && *p == ',') {
          p++;
          continue;
        }
        if (p < end && *p == '}') {
          p++;
          break;
        }
        ok = false;
      }
    } else if (*p == '[') {
      v.kind = JValue::kArr;
      p++;
      ws();
      if (p < end && *p == ']') {
        p++;
        return v;
      }
      while (ok) {
        v.arr.push_back(parse());
        ws();
        if (p < end && *p == ',') {
          p++;
          continue;
        }
        if (p < end && *p == ']') {
          p++;
          break;
        }
        ok = false;
      }
    } else if (*p == '"') {
      v.kind = JValue::kStr;
      p++;
      while (p < end && *p != '"') {
        if (*p == '\\' && p + 1 < end) p++;
        v.str.push_back(*p++);
      }
      if (p >= end)
        ok = false;
      else
        p++;
    } else if (lit("true")) {
      v.kind = JValue::kBool;
      v.b = true;
    } else if (lit("false")) {
      v.kind = JValue::kBool;
    } else if (lit("null")) {
      v.kind = JValue::kNull;
    } else {
      // the buffer is (pointer, length), not NUL-terminated: copy the number token before handing it to strtod/strtoull
      char tok[64];
      size_t len = 0;
      while (p + len < end && len + 1 < sizeof(tok) && (std::isdigit((unsigned char)p[len]) || std::strchr("+-.eE", p[len]))) len++;
      std::memcpy(tok, p, len);
      tok[len] = 0;
      char* e = nullptr;
      v.kind = JValue::kNum;
      v.num = std::strtod(tok, &e);
      if (e == tok) {
        ok = false;
        return v;
      }
      if (tok[0] != '-') v.unum = std::strtoull(tok, nullptr, 10);
      p += e - tok;
    }
    return v;
  }
};

const struct {
  const char* name;
  OpKind kind;
} kOpNames[] = {
    {"ShiftLeft", kOpShiftLeft},       {"ShiftRight", kOpShiftRight}, {"SwapRows", kOpSwapRows},
    {"Relinearize", kOpRelinearize},   {"Multiply", kOpMultiply},     {"MultiplyPlaintext", kOpMultiplyPlaintext},
    {"Add", kOpAdd},                   {"AddPlaintext", kOpAddPlaintext}, {"Negate", kOpNegate},
    {"Sub", kOpSub},                   {"SubPlaintext", kOpSubPlaintext}, {"InputCiphertext", kOpInputCiphertext},
    {"InputPlaintext", kOpInputPlaintext}, {"Literal", kOpLiteralU64},    {"OutputCiphertext", kOpOutputCiphertext},
};

bool op_from_name(const std::string& s, OpKind* k) {
  for (auto& e : kOpNames)
    if (s == e.name) {
      *k = e.kind;
      return true;
    }
  return false;
}

}  // namespace

int Program::add_node(OpKind op, u64 arg) {
  drop_plan();
  nodes_.push_back(Node{op, arg, -1, -1});
  return (int)nodes_.size() - 1;
}

// bincode 1.x default configuration: little-endian fixed-width integers, u64 sequence lengths, u32 enum variant index
int Program::add_plaintext_literal(const uint8_t* bytes, size_t len, std::string* err) {
  auto fail = [&](const char* m) {
    if (err) *err = m;
    return -1;
  };
  size_t pos = 0;
  auto rd = [&](size_t width, u64* out) {
    if (len - pos < width) return false;
    u64 v = 0;
    for (size_t i = 0; i < width; i++) v |= (u64)bytes[pos + i] << (8 * i);
    pos += width;
    *out = v;
    return true;
  };
  u64 variant, count, n, nprimes, t, scheme, sec, blob;
  if (!rd(4, &variant) || variant != 0) return fail("plaintext literal: not InnerPlaintext::Seal");
  if (!rd(8, &count) || count != 1) return fail("plaintext literal: must hold exactly one plaintext (run.rs:319-322)");
  PlainLiteral lit;
  if (!rd(8, &n) || !rd(8, &nprimes) || nprimes > 64) return fail("plaintext literal: malformed Params");
  lit.n = n;
  for (u64 i = 0; i < nprimes; i++) {
    u64 p;
    if (!rd(8, &p)) return fail("plaintext literal: malformed Params");
    lit.primes.push_back(p);
  }
  if (!rd(8, &t) || !rd(4, &scheme) || !rd(4, &sec) || scheme != 0) return fail("plaintext literal: malformed Params");
  lit.t = t;
  if (!rd(8, &blob) || blob != len - pos) return fail("plaintext literal: malformed data field");
  WirePlaintext wp;
  size_t used = 0;
  if (wire_unpack_plaintext(bytes + pos, (size_t)blob, &wp, &used) != kWireOk || used != blob)
    return fail("plaintext literal: not a SEAL plaintext");
  if (wp.coeffs.size() > n) return fail("plaintext literal: more coefficients than the polynomial degree");
  for (unsigned long long c : wp.coeffs) {
    if (c >= t) return fail("plaintext literal: coefficient not below the plain modulus");
    lit.coeffs.push_back((u64)c);
  }
  literals_.push_back(std::move(lit));
  return add_node(kOpLiteralPlaintext, literals_.size() - 1);
}

int Program::add_edge(int src, int dst, EdgeKind kind) {
  if (src < 0 || dst < 0 || src >= (int)nodes_.size() || dst >= (int)nodes_.size() || src == dst) return kInvalidArg;
  drop_plan();
  Node& d = nodes_[dst];
  if (kind == kEdgeRight) {
    if (d.right >= 0) return kInvalidArg;
    d.right = src;
  } else {
    if (d.left >= 0) return kInvalidArg;
    d.left = src;
  }
  return kOk;
}

int Program::load_json(const char* text, size_t len, std::string* err) {
  JParser jp{text, text + len};
  JValue root = jp.parse();
  auto fail = [&](const char* m) {
    if (err) *err = m;
    return (int)kInvalidArg;
  };
  if (!jp.ok || root.kind != JValue::kObj) return fail("malformed JSON");
  const JValue* g = root.get("graph") ? root.get("graph") : &root;
  const JValue* nodes = g->get("nodes");
  const JValue* edges = g->get("edges");
  if (!nodes || !edges || nodes->kind != JValue::kArr || edges->kind != JValue::kArr) return fail("graph needs nodes and edges");
  const JValue* holes = g->get("node_holes");
  if (holes && holes->kind == JValue::kArr && !holes->arr.empty()) return fail("graphs with node holes are not supported");
  drop_plan();
  nodes_.clear();
  literals_.clear();
  for (const JValue& nv : nodes->arr) {
    const JValue* op = nv.kind == JValue::kObj ? nv.get("operation") : nullptr;
    if (!op) return fail("node without operation");
    OpKind kind;
    u64 arg = 0;
    if (op->kind == JValue::kStr) {  // unit variant
      if (!op_from_name(op->str, &kind)) return fail("unknown operation");
      if (kind == kOpInputCiphertext || kind == kOpInputPlaintext || kind == kOpLiteralU64) return fail("operation needs a payload");
    } else if (op->kind == JValue::kObj && op->obj.size() == 1) {  // {"InputCiphertext": 0} / {"Literal": {"U64": 3}}
      if (!op_from_name(op->obj[0].first, &kind)) return fail("unknown operation");
      const JValue& payload = op->obj[0].second;
      if (kind == kOpLiteralU64) {
        const JValue* u = payload.kind == JValue::kObj ? payload.get("U64") : nullptr;
        const JValue* pl = payload.kind == JValue::kObj ? payload.get("Plaintext") : nullptr;
        if (pl) {  // serde_json writes Vec<u8> as an array of numbers
          if (pl->kind != JValue::kArr) return fail("Literal::Plaintext must be a byte array");
          std::vector<uint8_t> bytes;
          for (const JValue& b : pl->arr) {
            if (b.kind != JValue::kNum || b.unum > 255) return fail("Literal::Plaintext must be a byte array");
            bytes.push_back((uint8_t)b.unum);
          }
          std::string lerr;
          if (add_plaintext_literal(bytes.data(), bytes.size(), &lerr) < 0) {
            if (err) *err = lerr;
            return (int)kInvalidArg;
          }
          continue;
        }
        if (!u) return fail("Literal must be U64 or Plaintext");
        arg = u->unum;
      } else if (kind == kOpInputCiphertext || kind == kOpInputPlaintext) {
        if (payload.kind != JValue::kNum) return fail("input index must be a number");
        arg = payload.unum;
      } else {
        return fail("unexpected payload");
      }
    } else {
      return fail("malformed operation");
    }
    add_node(kind, arg);
  }
  for (const JValue& ev : edges->arr) {
    if (ev.kind != JValue::kArr || ev.arr.size() != 3 || ev.arr[2].kind != JValue::kStr) return fail("malformed edge");
    EdgeKind ek;
    if (ev.arr[2].str == "Left")
      ek = kEdgeLeft;
    else if (ev.arr[2].str == "Right")
      ek = kEdgeRight;
    else if (ev.arr[2].str == "Unary")
      ek = kEdgeUnary;
    else
      return fail("unsupported edge kind");
    if (add_edge((int)ev.arr[0].unum, (int)ev.arr[1].unum, ek) != kOk) return fail("invalid edge");
  }
  return kOk;
}

int Program::validate(std::string* err) const {
  auto fail = [&](const char* m) {
    if (err) *err = m;
    return (int)kInvalidArg;
  };
  for (const Node& nd : nodes_) {
    switch (nd.op) {
      case kOpInputCiphertext:
      case kOpInputPlaintext:
      case kOpLiteralU64:
      case kOpLiteralPlaintext:
        if (nd.left >= 0 || nd.right >= 0) return fail("input/literal nodes take no operands");
        break;
      case kOpNegate:
      case kOpSwapRows:
      case kOpRelinearize:
      case kOpOutputCiphertext:
        if (nd.left < 0 || nd.right >= 0) return fail("unary node needs exactly one operand");
        break;
      default:
        if (nd.left < 0 || nd.right < 0) return fail("binary node needs a left and a right operand");
    }
  }
  return kOk;
}

size_t Program::num_outputs() const {
  size_t c = 0;
  for (const Node& nd : nodes_) c += nd.op == kOpOutputCiphertext;
  return c;
}

// Kahn topological order (the reference walks the same dependency structure with rayon: run.rs:372-472)
bool Program::topo_order(std::vector<int>* order) const {
  const int n = (int)nodes_.size();
  std::vector<int> indeg(n, 0);
  std::vector<std::vector<int>> users(n);
  for (int i = 0; i < n; i++) {
    for (int src : {nodes_[i].left, nodes_[i].right})
      if (src >= 0) {
        indeg[i]++;
        users[src].push_back(i);
      }
  }
  std::vector<int> ready;
  for (int i = n - 1; i >= 0; i--)
    if (!indeg[i]) ready.push_back(i);
  order->clear();
  while (!ready.empty()) {
    int v = ready.back();
    ready.pop_back();
    order->push_back(v);
    for (int u : users[v])
      if (--indeg[u] == 0) ready.push_back(u);
  }
  return (int)order->size() == n;
}

KeySel ProgramKeys::relin_sel() const {
  if (!index) return KeySel(relin.empty() ? nullptr : relin[0]);
  for (const u64* k : relin)
    if (!k) return KeySel();
  KeySel sel;
  sel.keys = relin.data();
  sel.nkeys = (u32)relin.size();
  sel.index = index;
  sel.period = period;
  return sel;
}

KeySel ProgramKeys::galois_sel(u32 elt) const {
  const u32 id = (elt - 1) >> 1;
  if (!index) {
    if (galois.empty()) return KeySel();
    auto it = galois[0].find(id);
    return KeySel(it == galois[0].end() ? nullptr : it->second);
  }
  auto cached = tables_.find(id);
  if (cached == tables_.end()) {
    std::vector<const u64*> tab;
    for (auto& g : galois) {
      auto it = g.find(id);
      if (it == g.end() || !it->second) {
        tab.clear();
        break;
      }
      tab.push_back(it->second);
    }
    if (tab.size() != galois.size()) tab.clear();
    cached = tables_.emplace(id, std::move(tab)).first;
  }
  if (cached->second.empty()) return KeySel();
  KeySel sel;
  sel.keys = cached->second.data();
  sel.nkeys = (u32)cached->second.size();
  sel.index = index;
  sel.period = period;
  return sel;
}

int Program::run(Evaluator& ev, size_t batch, const ProgramInput* inputs, size_t num_inputs, const ProgramKeys& keys,
                 u64* const* outputs, size_t num_outputs_given, hipStream_t s,
                 std::string* err) const {
  // read per run (not cached): the tests switch executors inside one process to compare their outputs word for word
  const char* env = std::getenv("HIPBFV_PROGRAM_SERIAL");
  const bool serial = env && env[0] == '1';
  return serial ? run_serial(ev, batch, inputs, num_inputs, keys, outputs, num_outputs_given, s, err)
                : run_plan(ev, batch, inputs, num_inputs, keys, outputs, num_outputs_given, s, err);
}

int Program::run_serial(Evaluator& ev, size_t batch, const ProgramInput* inputs, size_t num_inputs, const ProgramKeys& keys,
                        u64* const* outputs, size_t num_outputs_given, hipStream_t s,
                        std::string* err) const {
  const KeySel relin_key = keys.relin_sel();
  auto fail = [&](int code, const char* m) {
    if (err) *err = m;
    return code;
  };
  if (int rc = validate(err)) return rc;
  if (num_outputs_given != num_outputs()) return fail(kInvalidArg, "wrong number of output buffers");
  std::vector<int> order;
  if (!topo_order(&order)) return fail(kInvalidArg, "program graph has a cycle");
  Context* ctx = ev.ctx();
  const size_t n = ctx->n();
  const int nn = (int)nodes_.size();

  struct Value {
    const u64* ct = nullptr;  // device ciphertext batch
    u32 size = 0;
    bool owned = false;       // allocated from the scratch pool by this run
    const u64* plain = nullptr;
    size_t pstride = 0;
    int uses = 0;
  };
  std::vector<Value> val(nn);
  for (int i = 0; i < nn; i++)
    for (int src : {nodes_[i].left, nodes_[i].right})
      if (src >= 0) val[src].uses++;

  ScratchPool& pool = ev.scratch();
  std::vector<void*> live;
  auto alloc_ct = [&](u32 size) -> u64* {
    void* p = pool.acquire(batch * ctx->ct_words(size) * sizeof(u64), s);
    if (p) live.push_back(p);
    return (u64*)p;
  };
  auto release = [&](int node) {
    Value& v = val[node];
    if (--v.uses == 0 && v.owned) {
      pool.release((void*)v.ct, s);
      live.erase(std::remove(live.begin(), live.end(), (void*)v.ct), live.end());
      v.owned = false;
    }
  };
  auto cleanup = [&](int code, const char* m) {
    for (void* p : live) pool.release(p, s);
    return fail(code, m);
  };
  auto galois_key = [&](u32 elt) -> KeySel { return keys.galois_sel(elt); };
  // rotate `in` by `steps` into `out` following SEAL's rotate_internal (direct key or NAF chain)
  std::function<int(const u64*, int, u64*)> rotate = [&](const u64* in, int steps, u64* out) -> int {
    if (steps == 0) {
      if (in != out && hipMemcpyAsync(out, in, batch * ctx->ct_words(2) * sizeof(u64), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return kHipError;
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

  size_t out_idx = 0;
  std::vector<int> out_slot(nn, -1);
  for (int i = 0; i < nn; i++)
    if (nodes_[i].op == kOpOutputCiphertext) out_slot[i] = (int)out_idx++;

  // a Multiply consumed only by one Relinearize is executed as the fused multiply_relin
  std::vector<int> fused_into(nn, -1);
  for (int i = 0; i < nn; i++) {
    if (nodes_[i].op == kOpRelinearize) {
      const int m = nodes_[i].left;
      if (nodes_[m].op == kOpMultiply && val[m].uses == 1) fused_into[m] = i;
    }
  }

  // An Add one of whose operands comes from a key-switching node (Relinearize / rotation) that has no other user: the
  // producer adds the other operand inside its last kernel -- when that operand already exists at that point of the
  // topological order -- and the Add node then just adopts the result (no separate element-wise pass).
  std::vector<int> add_user(nn, -1), folded(nn, -1);
  for (int i = 0; i < nn; i++) {
    if (nodes_[i].op != kOpAdd || nodes_[i].left == nodes_[i].right) continue;
    for (int src : {nodes_[i].left, nodes_[i].right}) {
      const OpKind k = nodes_[src].op;
      if ((k == kOpRelinearize || k == kOpShiftLeft || k == kOpShiftRight || k == kOpSwapRows) && val[src].uses == 1) add_user[src] = i;
    }
  }
  auto addend_for = [&](int producer) -> const u64* {
    const int u = add_user[producer];
    if (u < 0 || folded[u] >= 0) return nullptr;
    const int other = nodes_[u].left == producer ? nodes_[u].right : nodes_[u].left;
    const Value& o = val[other];
    return (o.ct && o.size == 2) ? o.ct : nullptr;
  };

  for (int id : order) {
    const Node& nd = nodes_[id];
    Value& v = val[id];
    const Value* L = nd.left >= 0 ? &val[nd.left] : nullptr;
    const Value* R = nd.right >= 0 ? &val[nd.right] : nullptr;
    int rc = kOk;
    switch (nd.op) {
      case kOpInputCiphertext:
      case kOpInputPlaintext: {
        if (nd.arg >= num_inputs) return cleanup(kInvalidArg, "input index out of range");
        const ProgramInput& in = inputs[nd.arg];
        if (nd.op == kOpInputCiphertext) {
          if (in.kind != 0 || !in.ptr) return cleanup(kInvalidArg, "argument is not a ciphertext");
          v.ct = in.ptr;
          v.size = 2;
        } else {
          if (in.kind == 2) return cleanup(kUnsupported, "transform-domain plaintext arguments need the scheduled executor (unset HIPBFV_PROGRAM_SERIAL)");
          if (in.kind != 1 || !in.ptr) return cleanup(kInvalidArg, "argument is not a plaintext");
          v.plain = in.ptr;
          v.pstride = in.stride;
        }
        continue;
      }
      case kOpLiteralU64:
        continue;
      case kOpLiteralPlaintext: {
        const PlainLiteral& lit = literals_[nd.arg];
        const std::vector<u64>& kp = ctx->key_primes();
        if (lit.n != n || lit.t != ctx->host().t || lit.primes.size() != kp.size() || !std::equal(kp.begin(), kp.end(), lit.primes.begin()))
          return cleanup(kInvalidArg, "plaintext literal was built for different encryption parameters");
        u64* dev = (u64*)pool.acquire(n * sizeof(u64), s);
        if (!dev) return cleanup(kOutOfMemory, "scratch allocation failed");
        live.push_back(dev);
        std::vector<u64> host(n, 0);
        std::copy(lit.coeffs.begin(), lit.coeffs.end(), host.begin());
        // ordered after earlier users of the recycled buffer on this stream; drained so `host` may go out of scope
        if (hipMemcpyAsync(dev, host.data(), n * sizeof(u64), hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
          return cleanup(kHipError, "copy failed");
        v.plain = dev;
        v.pstride = 0;  // one plaintext shared by the whole batch
        continue;
      }
      case kOpOutputCiphertext: {
        if (!L->ct || L->size != 2) return cleanup(kInvalidArg, "program output must be a size-2 ciphertext");
        if (hipMemcpyAsync(outputs[out_slot[id]], L->ct, batch * ctx->ct_words(2) * sizeof(u64), hipMemcpyDeviceToDevice, s) != hipSuccess)
          return cleanup(kHipError, "copy failed");
        release(nd.left);
        continue;
      }
      default:
        break;
    }
    const bool fused_relin = nd.op == kOpRelinearize && fused_into[nd.left] == id;
    if (!L || (!L->ct && !fused_relin)) return cleanup(kInvalidArg, "left operand is not a ciphertext");
    switch (nd.op) {
      case kOpMultiply: {
        if (!R->ct) return cleanup(kInvalidArg, "right operand is not a ciphertext");
        if (fused_into[id] >= 0) {  // executed when the Relinearize node is reached
          continue;
        }
        v.size = L->size + R->size - 1;
        u64* out = alloc_ct(v.size);
        if (!out) return cleanup(kOutOfMemory, "out of device memory");
        rc = ev.multiply(L->ct, L->size, R->ct, R->size, out, batch, s);
        v.ct = out;
        v.owned = true;
        break;
      }
      case kOpRelinearize: {
        const Node& mnode = nodes_[nd.left];
        u64* out = alloc_ct(2);
        if (!out) return cleanup(kOutOfMemory, "out of device memory");
        v.size = 2;
        if (fused_into[nd.left] == id) {
          const Value& A = val[mnode.left];
          const Value& B = val[mnode.right];
          if (A.size != 2 || B.size != 2) {
            // general sizes: unfused fallback
            u64* tmp = alloc_ct(A.size + B.size - 1);
            if (!tmp) return cleanup(kOutOfMemory, "out of device memory");
            rc = ev.multiply(A.ct, A.size, B.ct, B.size, tmp, batch, s);
            if (!rc) rc = A.size + B.size - 1 == 3 ? ev.relinearize(tmp, relin_key, out, batch, s) : (int)kInvalidArg;
            pool.release(tmp, s);
            live.erase(std::remove(live.begin(), live.end(), (void*)tmp), live.end());
          } else {
            const u64* addend = addend_for(id);
            rc = relin_key.present() ? ev.multiply_relin(A.ct, B.ct, relin_key, out, batch, s, addend) : (int)kNoKey;
            if (addend && !rc) folded[add_user[id]] = id;
          }
          v.ct = out;
          v.owned = true;
          if (rc) return cleanup(rc, "multiply+relinearize failed");
          release(mnode.left);
          release(mnode.right);
          val[nd.left].uses = 0;
          continue;
        }
        if (L->size == 2) {
          if (hipMemcpyAsync(out, L->ct, batch * ctx->ct_words(2) * sizeof(u64), hipMemcpyDeviceToDevice, s) != hipSuccess) rc = kHipError;
        } else if (L->size == 3) {
          const u64* addend = addend_for(id);
          rc = relin_key.present() ? ev.relinearize(L->ct, relin_key, out, batch, s, addend) : (int)kNoKey;
          if (addend && !rc) folded[add_user[id]] = id;
        } else {
          rc = kInvalidArg;
        }
        v.ct = out;
        v.owned = true;
        break;
      }
      case kOpAdd:
      case kOpSub: {
        if (nd.op == kOpAdd && folded[id] >= 0) {  // the producer already added the other operand: adopt its buffer
          Value& p = val[folded[id]];
          v.ct = p.ct;
          v.size = 2;
          v.owned = p.owned;
          p.owned = false;
          break;
        }
        if (!R->ct) return cleanup(kInvalidArg, "right operand is not a ciphertext");
        if (L->size != R->size) return cleanup(kInvalidArg, "add/sub of different ciphertext sizes is not supported in batched programs");
        v.size = L->size;
        u64* out = alloc_ct(v.size);
        if (!out) return cleanup(kOutOfMemory, "out of device memory");
        rc = nd.op == kOpAdd ? ev.add(L->ct, R->ct, out, v.size, batch, s) : ev.sub(L->ct, R->ct, out, v.size, batch, s);
        v.ct = out;
        v.owned = true;
        break;
      }
      case kOpNegate: {
        v.size = L->size;
        u64* out = alloc_ct(v.size);
        if (!out) return cleanup(kOutOfMemory, "out of device memory");
        rc = ev.negate(L->ct, out, v.size, batch, s);
        v.ct = out;
        v.owned = true;
        break;
      }
      case kOpAddPlaintext:
      case kOpSubPlaintext:
      case kOpMultiplyPlaintext: {
        if (!R->plain) return cleanup(kInvalidArg, "right operand is not a plaintext");
        v.size = L->size;
        u64* out = alloc_ct(v.size);
        if (!out) return cleanup(kOutOfMemory, "out of device memory");
        if (nd.op == kOpAddPlaintext)
          rc = ev.add_plain(L->ct, L->size, R->plain, R->pstride, out, batch, s);
        else if (nd.op == kOpSubPlaintext)
          rc = ev.sub_plain(L->ct, L->size, R->plain, R->pstride, out, batch, s);
        else
          rc = ev.multiply_plain(L->ct, L->size, R->plain, R->pstride, out, batch, s);
        v.ct = out;
        v.owned = true;
        break;
      }
      case kOpShiftLeft:
      case kOpShiftRight: {
        if (nodes_[nd.right].op != kOpLiteralU64) return cleanup(kInvalidArg, "shift amount must be a Literal::U64 (run.rs:177-183)");
        if (L->size != 2) return cleanup(kInvalidArg, "rotation needs a size-2 ciphertext");
        if (!ctx->batching()) return cleanup(kUnsupported, "encryption parameters do not support batching");
        const int k = (int)nodes_[nd.right].arg;
        v.size = 2;
        u64* out = alloc_ct(2);
        if (!out) return cleanup(kOutOfMemory, "out of device memory");
        {
          const int steps = nd.op == kOpShiftLeft ? k : -k;
          const u32 elt = steps ? ev.galois_elt_from_step(steps) : 0;
          const KeySel key = elt ? galois_key(elt) : KeySel();
          const u64* addend = key.present() ? addend_for(id) : nullptr;  // single key switch: the Add can ride along
          if (addend) {
            rc = ev.apply_galois(L->ct, elt, key, out, batch, s, addend);
            if (!rc) folded[add_user[id]] = id;
          } else {
            rc = rotate(L->ct, steps, out);
          }
        }
        v.ct = out;
        v.owned = true;
        break;
      }
      case kOpSwapRows: {
        if (L->size != 2) return cleanup(kInvalidArg, "rotation needs a size-2 ciphertext");
        if (!ctx->batching()) return cleanup(kUnsupported, "encryption parameters do not support batching");
        const u32 elt = 2 * (u32)n - 1;
        const KeySel key = galois_key(elt);
        if (!key.present()) return cleanup(kNoKey, "Galois key for the column rotation is missing");
        v.size = 2;
        u64* out = alloc_ct(2);
        if (!out) return cleanup(kOutOfMemory, "out of device memory");
        {
          const u64* addend = addend_for(id);
          rc = ev.apply_galois(L->ct, elt, key, out, batch, s, addend);
          if (addend && !rc) folded[add_user[id]] = id;
        }
        v.ct = out;
        v.owned = true;
        break;
      }
      default:
        return cleanup(kInvalidArg, "unsupported operation");
    }
    if (rc) return cleanup(rc, "operation failed");
    if (nd.left >= 0) release(nd.left);
    if (nd.right >= 0 && nodes_[nd.right].op != kOpLiteralU64) release(nd.right);
    if (v.uses == 0 && v.owned) {  // dead value (pruned graphs should not have any)
      pool.release((void*)v.ct, s);
      live.erase(std::remove(live.begin(), live.end(), (void*)v.ct), live.end());
      v.owned = false;
    }
  }
  for (void* p : live) pool.release(p, s);
  return kOk;
}

}  // namespace hipbfv
