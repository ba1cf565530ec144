// Host-side layout assembler + level scheduler: the MI355X-native stand-in for
// Assembler.assemble (models_clevr/nmn3_assembler.py:153-222) and for TensorFlow-Fold's
// Loom (td.Compiler.build_feed_dict + depth-wise dynamic batching,
// models_clevr/nmn3_model.py:55-159).
//
// Fold batches "all instances of one module type at one depth" into one TF op.  Here the unit of
// batching is a *stage* of a level instead: every operator that can run at a level is put in one
// of three launches (A: light attention/answer operators and the Find-type epilogues,
// B: softmax-attention pooling, C: answer heads), so a whole minibatch of heterogeneous trees
// needs ~3 launches per level instead of one launch per (module type, depth).  The text-independent
// conv_image GEMM of Find / Filter / FindSameProperty is hoisted out of the tree and computed once
// per image and weight set.
#include "program.h"

#include <algorithm>
#include <cstring>

namespace n2nmn {

void Program::clear() {
  N = T = num_rows = num_levels = 0;
  nodes.clear(); dev_nodes.clear(); st_kind.clear(); st_op.clear(); st_remains.clear();
  tab.clear(); launches.clear();
  num_text = num_pool = num_find_img = num_fsp_img = num_inputs0 = num_inputs1 = 0;
  error.clear();
}

// models_clevr/nmn3_assembler.py:9-24
int op_arity(int op) {
  switch (op) {
    case N2NMN_OP_SCENE: case N2NMN_OP_FIND: return 0;
    case N2NMN_OP_FILTER: case N2NMN_OP_FIND_SAME_PROPERTY: case N2NMN_OP_TRANSFORM:
    case N2NMN_OP_EXIST: case N2NMN_OP_COUNT: case N2NMN_OP_DESCRIBE: return 1;
    case N2NMN_OP_AND: case N2NMN_OP_OR: case N2NMN_OP_EQUAL_NUM: case N2NMN_OP_MORE_NUM:
    case N2NMN_OP_LESS_NUM: case N2NMN_OP_SAME_PROPERTY: return 2;
    case OP_INPUT: return 0;
    default: return -1;
  }
}

// models_clevr/nmn3_assembler.py:26-41
bool op_is_answer(int op) {
  switch (op) {
    case N2NMN_OP_EXIST: case N2NMN_OP_COUNT: case N2NMN_OP_EQUAL_NUM: case N2NMN_OP_MORE_NUM:
    case N2NMN_OP_LESS_NUM: case N2NMN_OP_SAME_PROPERTY: case N2NMN_OP_DESCRIBE: return true;
    default: return false;
  }
}

// which operators read a text parameter (word_vecs row): models_clevr/nmn3_modules.py
// (_slice_word_vecs call sites :78,117,139,189,407,459)
bool op_has_text(int op) {
  switch (op) {
    case N2NMN_OP_FIND: case N2NMN_OP_FILTER: case N2NMN_OP_FIND_SAME_PROPERTY:
    case N2NMN_OP_TRANSFORM: case N2NMN_OP_SAME_PROPERTY: case N2NMN_OP_DESCRIBE: return true;
    default: return false;
  }
}

static bool op_is_pool(int op) {
  return op == N2NMN_OP_FIND_SAME_PROPERTY || op == N2NMN_OP_SAME_PROPERTY ||
         op == N2NMN_OP_DESCRIBE;
}

// text-map weight set: Filter reuses FindModule's fc_text (nmn3_modules.py:129)
static int text_weight_set(int op) {
  switch (op) {
    case N2NMN_OP_FIND: case N2NMN_OP_FILTER: return 0;
    case N2NMN_OP_FIND_SAME_PROPERTY: return 1;
    case N2NMN_OP_TRANSFORM: return 2;
    case N2NMN_OP_SAME_PROPERTY: return 3;
    case N2NMN_OP_DESCRIBE: return 4;
    default: return -1;
  }
}

int assemble_tokens(Program& p, const int32_t* tokens, int T, int N, const int32_t* token_op,
                    int V, uint8_t* validity) {
  p.clear();
  if (!tokens || !token_op || T <= 0 || N <= 0 || V <= 0) {
    p.error = "assemble: bad arguments";
    return N2NMN_EINVAL;
  }
  p.N = N; p.T = T; p.num_rows = N;
  p.st_kind.assign(N, N2NMN_ASM_OK);
  p.st_op.assign(N, -1);
  p.st_remains.assign(N, 0);
  std::vector<int32_t> stack;
  stack.reserve(T);
  for (int n = 0; n < N; ++n) {
    const size_t first_node = p.nodes.size();
    int kind = N2NMN_ASM_OK, bad_op = -1, remains = 0;
    // a layout must contain <eos> (nmn3_assembler.py:172-173)
    bool has_eos = false;
    for (int t = 0; t < T; ++t) {
      const int tok = tokens[(size_t)t * N + n];
      if (tok < 0 || tok >= V) {
        p.error = "assemble: token out of range";
        return N2NMN_EINVAL;
      }
      if (token_op[tok] < 0) has_eos = true;
    }
    stack.clear();
    if (!has_eos) {
      kind = N2NMN_ASM_NO_EOS;
    } else {
      for (int t = 0; t < T && kind == N2NMN_ASM_OK; ++t) {
        const int op = token_op[tokens[(size_t)t * N + n]];
        if (op < 0) break;                                   // <eos>
        const int k = op_arity(op);
        if (k < 0 || op == OP_INPUT) {
          p.error = "assemble: unknown op code in token_op";
          return N2NMN_EKEY;
        }
        if ((int)stack.size() < k) {                         // :189-191
          kind = N2NMN_ASM_NOT_ENOUGH; bad_op = op;
          break;
        }
        n2nmn_node nd;
        nd.op = op; nd.time_idx = t; nd.batch_idx = n; nd.in0 = nd.in1 = -1;
        nd.level = 0; nd.out_row = -1; nd.reserved = 0;
        for (int j = k - 1; j >= 0; --j) {                   // :194-199 input_{k-1} = stack top
          const int32_t top = stack.back();
          stack.pop_back();
          if (op_is_answer(p.nodes[top].op)) {
            kind = N2NMN_ASM_INCOMPATIBLE; bad_op = op;
            break;
          }
          (j == 0 ? nd.in0 : nd.in1) = top;
        }
        if (kind != N2NMN_ASM_OK) break;
        stack.push_back((int32_t)p.nodes.size());
        p.nodes.push_back(nd);
      }
      if (kind == N2NMN_ASM_OK) {
        if (stack.size() != 1) {                             // :205-206
          kind = N2NMN_ASM_STACK_SIZE; remains = (int)stack.size();
        } else if (!op_is_answer(p.nodes[stack[0]].op)) {    // :209-211
          kind = N2NMN_ASM_NOT_ANS;
        }
      }
    }
    if (kind != N2NMN_ASM_OK) {
      p.nodes.resize(first_node);                            // INVALID_EXPR: zero logits row
    } else {
      p.nodes[stack[0]].out_row = n;
    }
    p.st_kind[n] = kind; p.st_op[n] = bad_op; p.st_remains[n] = remains;
    if (validity) validity[n] = (kind == N2NMN_ASM_OK) ? 1 : 0;
  }
  return schedule(p);
}

int from_nodes(Program& p, const n2nmn_node* nodes, int num_nodes, int num_rows) {
  p.clear();
  if (num_nodes < 0 || num_rows < 0 || (num_nodes > 0 && !nodes)) {
    p.error = "program_from_nodes: bad arguments";
    return N2NMN_EINVAL;
  }
  p.num_rows = num_rows;
  p.nodes.assign(nodes, nodes + num_nodes);
  int maxn = -1;
  for (int i = 0; i < num_nodes; ++i) {
    const n2nmn_node& nd = p.nodes[i];
    const int k = op_arity(nd.op);
    if (k < 0) { p.error = "program_from_nodes: unknown op"; return N2NMN_EKEY; }
    const int ins[2] = {nd.in0, nd.in1};
    for (int j = 0; j < 2; ++j) {
      if (j < k) {
        if (ins[j] < 0 || ins[j] >= i) {
          p.error = "program_from_nodes: inputs must precede their consumer";
          return N2NMN_EINVAL;
        }
        if (op_is_answer(p.nodes[ins[j]].op)) {
          p.error = "program_from_nodes: an answer node cannot be an input";
          return N2NMN_EINVAL;
        }
      } else if (ins[j] != -1) {
        p.error = "program_from_nodes: unused input must be -1";
        return N2NMN_EINVAL;
      }
    }
    if (op_is_answer(nd.op)) {
      if (nd.out_row < 0 || nd.out_row >= num_rows) {
        p.error = "program_from_nodes: answer node needs a valid out_row";
        return N2NMN_EINVAL;
      }
    }
    if (nd.batch_idx < 0 || nd.time_idx < 0) {
      p.error = "program_from_nodes: negative index";
      return N2NMN_EINVAL;
    }
    maxn = std::max(maxn, nd.batch_idx);
  }
  p.N = maxn + 1;
  p.st_kind.assign(num_rows, N2NMN_ASM_OK);
  p.st_op.assign(num_rows, -1);
  p.st_remains.assign(num_rows, 0);
  return schedule(p);
}

int schedule(Program& p) {
  const int nn = (int)p.nodes.size();
  p.dev_nodes.assign(nn, DevNode{});
  p.tab.clear(); p.launches.clear();
  p.num_text = p.num_pool = p.num_find_img = p.num_fsp_img = 0;
  p.num_inputs0 = p.num_inputs1 = 0;

  // ---- levels ---------------------------------------------------------------------------
  // exec level: launch group the node runs in; out level: level after which its attention map
  // is complete.  Stage order inside a level is A (light ops) -> B (pooling) -> C (heads), so a
  // pooling operator can consume maps produced by stage A of the same level.
  std::vector<int> outl(nn, 0);
  int max_level = -1;
  for (int i = 0; i < nn; ++i) {
    n2nmn_node& nd = p.nodes[i];
    int in_max = -1;
    if (nd.in0 >= 0) in_max = std::max(in_max, outl[nd.in0]);
    if (nd.in1 >= 0) in_max = std::max(in_max, outl[nd.in1]);
    int lvl;
    if (nd.op == OP_INPUT) {
      lvl = -1; outl[i] = -1;
    } else if (op_arity(nd.op) == 0) {
      lvl = 0; outl[i] = 0;
    } else if (op_is_pool(nd.op)) {
      lvl = std::max(in_max, 0);                       // pooled in stage B of this level
      // FindSameProperty's epilogue runs in stage A of the next level
      outl[i] = (nd.op == N2NMN_OP_FIND_SAME_PROPERTY) ? lvl + 1 : lvl;
    } else {
      lvl = in_max + 1; outl[i] = lvl;
    }
    nd.level = lvl;
    max_level = std::max(max_level, outl[i]);
  }
  p.num_levels = max_level + 1;

  // ---- slots ----------------------------------------------------------------------------
  std::vector<int> find_slot_of_img, fsp_slot_of_img;
  auto img_slot = [](std::vector<int>& map, int img, int& counter) {
    if ((int)map.size() <= img) map.resize(img + 1, -1);
    if (map[img] < 0) map[img] = counter++;
    return map[img];
  };
  for (int i = 0; i < nn; ++i) {
    const n2nmn_node& nd = p.nodes[i];
    DevNode& d = p.dev_nodes[i];
    d.op = nd.op; d.t = nd.time_idx; d.n = nd.batch_idx; d.in0 = nd.in0; d.in1 = nd.in1;
    d.out_row = nd.out_row; d.level = nd.level;
    d.tslot = op_has_text(nd.op) ? p.num_text++ : -1;
    d.pslot = op_is_pool(nd.op) ? p.num_pool++ : -1;
    d.mslot = -1;
    if (nd.op == N2NMN_OP_FIND || nd.op == N2NMN_OP_FILTER)
      d.mslot = img_slot(find_slot_of_img, nd.batch_idx, p.num_find_img);
    else if (nd.op == N2NMN_OP_FIND_SAME_PROPERTY)
      d.mslot = img_slot(fsp_slot_of_img, nd.batch_idx, p.num_fsp_img);
    if (nd.op == OP_INPUT) {
      // time_idx selects which external array (0/1), batch_idx its row
      if (nd.time_idx == 0) p.num_inputs0++; else p.num_inputs1++;
    }
  }

  auto begin_launch = [&](int kind, int level) {
    Launch l; l.kind = kind; l.level = level; l.offset = (int)p.tab.size(); l.count = 0;
    p.launches.push_back(l);
  };
  auto end_launch = [&]() {
    if (p.launches.back().count == 0) p.launches.pop_back();
  };

  // ---- text maps: groups of <= TM_GROUP nodes sharing a weight set ------------------------
  begin_launch(LK_TEXTMAP, -1);
  for (int ws = 0; ws < 5; ++ws) {
    std::vector<int> ids;
    for (int i = 0; i < nn; ++i)
      if (text_weight_set(p.nodes[i].op) == ws) ids.push_back(i);
    for (size_t g = 0; g < ids.size(); g += TM_GROUP) {
      const int cnt = (int)std::min<size_t>(TM_GROUP, ids.size() - g);
      p.tab.push_back(ws);
      p.tab.push_back(cnt);
      for (int j = 0; j < TM_GROUP; ++j) p.tab.push_back(j < cnt ? ids[g + j] : -1);
      p.launches.back().count++;
    }
  }
  end_launch();

  // ---- hoisted conv_image GEMMs: image lists ---------------------------------------------
  auto conv_launch = [&](int kind, const std::vector<int>& slot_of_img, int count) {
    begin_launch(kind, -1);
    std::vector<int> imgs(count, 0);
    for (size_t img = 0; img < slot_of_img.size(); ++img)
      if (slot_of_img[img] >= 0) imgs[slot_of_img[img]] = (int)img;
    for (int v : imgs) p.tab.push_back(v);
    p.launches.back().count = count;
    end_launch();
  };
  conv_launch(LK_CONV_FIND, find_slot_of_img, p.num_find_img);
  conv_launch(LK_CONV_FSP, fsp_slot_of_img, p.num_fsp_img);

  // ---- per level stages -------------------------------------------------------------------
  for (int lvl = 0; lvl <= max_level; ++lvl) {
    // stage A
    begin_launch(LK_ATT, lvl);
    for (int i = 0; i < nn; ++i) {
      const n2nmn_node& nd = p.nodes[i];
      int parts = 0;
      if (nd.op == OP_INPUT) continue;
      if (nd.op == N2NMN_OP_FIND_SAME_PROPERTY) {
        if (nd.level + 1 == lvl) parts = FIND_PARTS;           // epilogue one level later
      } else if (op_is_pool(nd.op)) {
        parts = 0;
      } else if (nd.level == lvl) {
        if (nd.op == N2NMN_OP_FIND || nd.op == N2NMN_OP_FILTER) parts = FIND_PARTS;
        else if (nd.op == N2NMN_OP_TRANSFORM) parts = TRANSFORM_PARTS;
        else parts = 1;
      }
      for (int pt = 0; pt < parts; ++pt) {
        p.tab.push_back(i); p.tab.push_back(pt); p.tab.push_back(parts); p.tab.push_back(0);
        p.launches.back().count++;
      }
    }
    end_launch();
    // stage B
    begin_launch(LK_POOL, lvl);
    for (int i = 0; i < nn; ++i) {
      const n2nmn_node& nd = p.nodes[i];
      if (!op_is_pool(nd.op) || nd.level != lvl) continue;
      for (int pt = 0; pt < POOL_PARTS; ++pt) {
        p.tab.push_back(i); p.tab.push_back(pt);
        p.launches.back().count++;
      }
    }
    end_launch();
    // stage C
    begin_launch(LK_HEAD, lvl);
    for (int i = 0; i < nn; ++i) {
      const n2nmn_node& nd = p.nodes[i];
      if ((nd.op == N2NMN_OP_DESCRIBE || nd.op == N2NMN_OP_SAME_PROPERTY) && nd.level == lvl) {
        p.tab.push_back(i);
        p.launches.back().count++;
      }
    }
    end_launch();
  }
  return N2NMN_OK;
}

}  // namespace n2nmn

// ------------------------------------------------------------------------------------------
// C ABI (host part)
// ------------------------------------------------------------------------------------------
namespace n2nmn { void set_last_error(const std::string& s); }

extern "C" {

int n2nmn_program_create(n2nmn_program** out) {
  if (!out) { n2nmn::set_last_error("program_create: null out"); return N2NMN_EINVAL; }
  *out = new (std::nothrow) n2nmn_program();
  if (!*out) { n2nmn::set_last_error("program_create: out of memory"); return N2NMN_EINVAL; }
  return N2NMN_OK;
}

int n2nmn_program_destroy(n2nmn_program* p) {
  delete p;
  return N2NMN_OK;
}

int n2nmn_assemble(n2nmn_program* p, const int32_t* tokens_host, int T, int N,
                   const int32_t* token_op_host, int V, uint8_t* validity_host) {
  if (!p) { n2nmn::set_last_error("assemble: null program"); return N2NMN_EINVAL; }
  const int rc = n2nmn::assemble_tokens(p->prog, tokens_host, T, N, token_op_host, V,
                                        validity_host);
  if (rc != N2NMN_OK) n2nmn::set_last_error(p->prog.error);
  return rc;
}

int n2nmn_program_from_nodes(n2nmn_program* p, const n2nmn_node* nodes_host, int num_nodes,
                             int num_rows) {
  if (!p) { n2nmn::set_last_error("program_from_nodes: null program"); return N2NMN_EINVAL; }
  const int rc = n2nmn::from_nodes(p->prog, nodes_host, num_nodes, num_rows);
  if (rc != N2NMN_OK) n2nmn::set_last_error(p->prog.error);
  return rc;
}

int n2nmn_program_num_nodes(const n2nmn_program* p) { return p ? (int)p->prog.nodes.size() : N2NMN_EINVAL; }
int n2nmn_program_num_rows(const n2nmn_program* p) { return p ? p->prog.num_rows : N2NMN_EINVAL; }
int n2nmn_program_num_levels(const n2nmn_program* p) { return p ? p->prog.num_levels : N2NMN_EINVAL; }
int n2nmn_program_num_launches(const n2nmn_program* p) { return p ? (int)p->prog.launches.size() : N2NMN_EINVAL; }

int n2nmn_program_get_nodes(const n2nmn_program* p, n2nmn_node* out_host, int capacity) {
  if (!p || (!out_host && capacity > 0)) {
    n2nmn::set_last_error("program_get_nodes: null argument");
    return N2NMN_EINVAL;
  }
  const int n = (int)p->prog.nodes.size();
  if (capacity < n) { n2nmn::set_last_error("program_get_nodes: capacity too small"); return N2NMN_ECAPACITY; }
  if (n) std::memcpy(out_host, p->prog.nodes.data(), sizeof(n2nmn_node) * n);
  return n;
}

int n2nmn_program_status(const n2nmn_program* p, int example, int32_t* kind, int32_t* op,
                         int32_t* remains) {
  if (!p || example < 0 || example >= (int)p->prog.st_kind.size()) {
    n2nmn::set_last_error("program_status: bad example index");
    return N2NMN_EINVAL;
  }
  if (kind) *kind = p->prog.st_kind[example];
  if (op) *op = p->prog.st_op[example];
  if (remains) *remains = p->prog.st_remains[example];
  return N2NMN_OK;
}


// CRC-32C of the tensor-bundle checkpoints (include/n2nmn.h section 8).  The x86 crc32 instruction computes
// exactly this polynomial; without SSE4.2 a byte table.
namespace {
struct Crc32cTable {
  uint32_t t[256];
  Crc32cTable() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82f63b78u : c >> 1;
      t[i] = c;
    }
  }
};
#if defined(__x86_64__)
__attribute__((target("sse4.2"))) uint32_t crc32c_hw(uint32_t c, const unsigned char* p, size_t n) {
  uint64_t c64 = c;
  while (n && (reinterpret_cast<uintptr_t>(p) & 7u)) { c64 = __builtin_ia32_crc32qi((uint32_t)c64, *p++); --n; }
  for (; n >= 8; n -= 8, p += 8) {
    uint64_t v;
    std::memcpy(&v, p, 8);
    c64 = __builtin_ia32_crc32di(c64, v);
  }
  while (n--) c64 = __builtin_ia32_crc32qi((uint32_t)c64, *p++);
  return (uint32_t)c64;
}
#endif
}  // namespace

uint32_t n2nmn_crc32c(uint32_t crc, const void* data, size_t n) {
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t c = crc ^ 0xffffffffu;
  if (!p || !n) return crc;
#if defined(__x86_64__)
  static const bool hw = __builtin_cpu_supports("sse4.2");
  if (hw) return crc32c_hw(c, p, n) ^ 0xffffffffu;
#endif
  static const Crc32cTable tab;
  while (n--) c = tab.t[(c ^ *p++) & 0xffu] ^ (c >> 8);
  return c ^ 0xffffffffu;
}

}  // extern "C"
