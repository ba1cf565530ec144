// Internal representation of a packed, level-scheduled batch of layout trees.
// Host-only (no HIP types) so the scheduler can be unit-tested without a GPU.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/n2nmn.h"

namespace n2nmn {

constexpr int OP_INPUT = 100;   // pseudo-op: attention map supplied by the caller (module_forward)

// Per-node record as the kernels see it (16 x int32, uploaded verbatim).
struct DevNode {
  int32_t op, t, n, in0, in1, out_row;
  int32_t tslot;    // row of the text-map buffer (fc_text / text_fc output), -1 if none
  int32_t pslot;    // pooling-job slot (Describe / SameProperty / FindSameProperty), -1 if none
  int32_t mslot;    // row of the conv_image buffer for this node's image (Find/Filter: Find
                    // weights; FindSameProperty: its own weights), -1 if none
  int32_t level;    // execution level
  int32_t pad[6];
};
static_assert(sizeof(DevNode) == 64, "DevNode must be 16 int32");

enum LaunchKind : int32_t {
  LK_TEXTMAP = 0,   // text maps for every node with a text parameter (groups of <= TM_GROUP)
  LK_CONV_FIND,     // conv_image GEMM with FindModule weights over the listed images
  LK_CONV_FSP,      // conv_image GEMM with FindSamePropertyModule weights
  LK_ATT,           // stage A of a level: light attention / answer operators, Find epilogues
  LK_POOL,          // stage B: softmax-attention pooling + partial fc_att
  LK_HEAD,          // stage C: Describe / SameProperty answer heads
};

struct Launch {
  int32_t kind;
  int32_t level;
  int32_t offset;   // offset (in int32) of this launch's work table inside Program::tab
  int32_t count;    // number of work items (workgroups, or images for the conv launches)
};

constexpr int TM_GROUP = 8;        // nodes per text-map workgroup (share one weight stream)
constexpr int FIND_PARTS = 4;      // row split of a Find-type epilogue
constexpr int TRANSFORM_PARTS = 3; // pixel split of a Transform node (<= 64 pixels per pass)
constexpr int POOL_PARTS = 4;      // channel split of an attention-pooling job

struct Program {
  int N = 0, T = 0;
  int num_rows = 0;
  int num_levels = 0;
  std::vector<n2nmn_node> nodes;             // public view
  std::vector<DevNode> dev_nodes;            // kernel view
  std::vector<int32_t> st_kind, st_op, st_remains;   // per example assembly status
  std::vector<int32_t> tab;                  // all launch work tables, packed
  std::vector<Launch> launches;
  int num_text = 0, num_pool = 0, num_find_img = 0, num_fsp_img = 0;
  int num_inputs0 = 0, num_inputs1 = 0;      // OP_INPUT nodes (module_forward only)
  std::string error;

  void clear();
};

int op_arity(int op);          // -1 for unknown
bool op_is_answer(int op);
bool op_has_text(int op);

// tokens [T,N] -> nodes (+ per example status); then schedule().
int assemble_tokens(Program& p, const int32_t* tokens, int T, int N, const int32_t* token_op,
                    int V, uint8_t* validity);
int from_nodes(Program& p, const n2nmn_node* nodes, int num_nodes, int num_rows);
// compute levels, slots and launch tables from p.nodes
int schedule(Program& p);

}  // namespace n2nmn

struct n2nmn_program {
  n2nmn::Program prog;
};
