// C ABI of the N2NMN hot path (include/n2nmn.h): context / weight store, phase-1 and phase-2
// launch orchestration.  Host code only; the kernels live in the .hip files.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <array>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.h"
#include "program.h"

#include "ctx.h"

namespace n2nmn {
static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
}  // namespace n2nmn

using namespace n2nmn;

namespace n2nmn {

const n2nmn_ctx* root(const n2nmn_ctx* c) { return c->parent ? c->parent : c; }
bool is_committed(const n2nmn_ctx* c) { return root(c)->committed; }
bool has_tables(const n2nmn_ctx* c) { return root(c)->have_tables; }

static void add_var(n2nmn_ctx* c, const std::string& name, std::vector<int64_t> shape,
                    bool present = true) {
  Var v;
  v.name = name;
  v.shape = std::move(shape);
  v.present = present;
  v.numel = present ? 1 : 0;
  if (present) {
    for (auto s : v.shape) v.numel *= (size_t)s;
    c->index[name] = (int)c->vars.size();
    c->pub.push_back((int)c->vars.size());
  } else {
    v.set = true;                       // nothing to load
  }
  c->vars.push_back(std::move(v));
}

// Variable names follow TF 1.0.0 scoping of the reference graph (SURVEY.md Appendix A.6;
// models_clevr/nmn3_model.py:22-49, nmn3_netgen_att.py:82-86,102-103,139-156,
// nmn3_modules.py scope= arguments, util/cnn.py:19-25,104-109).  Order must match enum VarId.
static void build_vars(n2nmn_ctx* c) {
  const n2nmn_dims& d = c->d;
  const int64_t L = d.lstm_dim, E = d.embed_dim_txt, En = d.embed_dim_nmn, M = d.map_dim,
                C = d.num_choices, D = d.D, HW = (int64_t)d.H * d.W, ks = d.kernel_size;
  const std::string enc = "neural_module_network/layout_generation/encoder_decoder/encoder/";
  const std::string dec = "neural_module_network/layout_generation/encoder_decoder/decoder/";
  const std::string mod = "neural_module_network/layout_execution/module_variables/";
  auto lstm = [&](const std::string& base, int layer, const char* kind) {
    return base + "lstm/multi_rnn_cell/cell_" + std::to_string(layer) + "/basic_lstm_cell/" + kind;
  };
  add_var(c, enc + "embedding_mat", {d.num_vocab_txt, E});
  add_var(c, lstm(enc, 0, "weights"), {E + L, 4 * L});
  add_var(c, lstm(enc, 0, "biases"), {4 * L});
  add_var(c, lstm(enc, 1, "weights"), {2 * L, 4 * L});
  add_var(c, lstm(enc, 1, "biases"), {4 * L});
  add_var(c, enc + "encoder_h_transform/weights", {L, L});
  add_var(c, enc + "encoder_h_transform/biases", {L});
  add_var(c, dec + "embedding_mat", {d.num_vocab_nmn, En});
  add_var(c, dec + "go_embedding", {1, En});
  add_var(c, dec + "att_prediction/v", {L});
  add_var(c, dec + "att_prediction/weights", {L, L});
  add_var(c, dec + "att_prediction/biases", {L});
  add_var(c, dec + "token_prediction/weights", {2 * L, d.num_vocab_nmn});
  add_var(c, dec + "token_prediction/biases", {d.num_vocab_nmn});
  add_var(c, lstm(dec, 0, "weights"), {En + L, 4 * L});
  add_var(c, lstm(dec, 0, "biases"), {4 * L});
  add_var(c, lstm(dec, 1, "weights"), {2 * L, 4 * L});
  add_var(c, lstm(dec, 1, "biases"), {4 * L});
  const bool vqa = d.variant == N2NMN_VARIANT_VQA;
  auto layer = [&](const char* scope, const char* name, std::vector<int64_t> shape,
                   bool present = true) {
    const int64_t out = shape.back();
    add_var(c, mod + scope + "/" + name + "/weights", shape, present);
    add_var(c, mod + scope + "/" + name + "/biases", {out}, present);
  };
  layer("FindModule", "conv_image", {D, M});
  layer("FindModule", "fc_text", {E, M});
  layer("FindModule", "conv_eltwise", {M, 1});
  // models_vqa: the three-way product module is called TransformModule
  // (models_vqa/nmn3_modules.py:123-171); it occupies the FindSameProperty slots
  const char* fsp = vqa ? "TransformModule" : "FindSamePropertyModule";
  layer(fsp, "conv_image", {D, M});
  layer(fsp, "fc_text", {E, M});
  layer(fsp, "fc_att", {D, M});
  layer(fsp, "conv_eltwise", {M, 1});
  layer("TransformModule", "conv_maps", {ks, ks, 1, M}, !vqa);
  layer("TransformModule", "text_fc", {E, M}, !vqa);
  layer("TransformModule", "conv_eltwise", {M, 1}, !vqa);
  layer("ExistModule", "fc_scores", {3, C}, !vqa);
  layer("CountModule", "fc_scores", {HW + 2, C}, !vqa);
  layer("EqualNumModule", "fc_scores", {2 * HW + 4, C}, !vqa);
  layer("MoreNumModule", "fc_scores", {2 * HW + 4, C}, !vqa);
  layer("LessNumModule", "fc_scores", {2 * HW + 4, C}, !vqa);
  layer("SamePropertyModule", "fc_text", {E, M}, !vqa);
  layer("SamePropertyModule", "fc_att_0", {D, M}, !vqa);
  layer("SamePropertyModule", "fc_att_1", {D, M}, !vqa);
  layer("SamePropertyModule", "fc_eltwise", {M, C}, !vqa);
  layer("DescribeModule", "fc_text", {E, M});
  layer("DescribeModule", "fc_att", {D, M});
  layer("DescribeModule", "fc_eltwise", {M, C});
  // models_vqa/question_prior_net.py:10-28 (scope neural_module_network/question_prior_net)
  const bool qpn = vqa && d.qpn_hidden > 0;
  const int64_t Hq = d.qpn_hidden > 0 ? d.qpn_hidden : 1;
  const std::string q = "neural_module_network/question_prior_net/";
  add_var(c, q + "fc1/weights", {2 * L, Hq}, qpn);
  add_var(c, q + "fc1/biases", {Hq}, qpn);
  add_var(c, q + "fc2/weights", {Hq, C}, qpn);
  add_var(c, q + "fc2/biases", {C}, qpn);
}

static size_t carve_weights(n2nmn_ctx* c, char* base) {
  const n2nmn_dims& d = c->d;
  const size_t L = d.lstm_dim, E = d.embed_dim_txt, N = d.N, T = d.T_encoder, Td = d.T_decoder,
               V = d.num_vocab_nmn, Vt = d.num_vocab_txt, D = d.D, HW = (size_t)d.H * d.W;
  const size_t Mp = c->Mp, HWp = c->HWp;
  Carver k(base);
  for (auto& v : c->vars) v.mirror = k.take<float>(v.numel);
  c->enc_W0x_p = k.take<float>((size_t)c->KpE * 4 * L);
  c->dec_W0x_p = k.take<float>((size_t)c->KpE * 4 * L);
  c->enc_xtab = c->big_vocab ? nullptr : k.take<float>(Vt * 4 * L);
  c->dec_xtab = k.take<float>((V + 1) * 4 * L);
  c->enc_b0_t = k.take<float>(4 * L); c->dec_b0_t = k.take<float>(4 * L);
  c->enc_b1_t = k.take<float>(4 * L); c->dec_b1_t = k.take<float>(4 * L);
  c->enc_W0h_t = k.take<float>(L * 4 * L);
  c->enc_W1_t = k.take<float>(2 * L * 4 * L);
  c->dec_W0h_t = k.take<float>(L * 4 * L);
  c->dec_W1_t = k.take<float>(2 * L * 4 * L);
  if (L % 128 == 0) {
    c->enc_W0h_64 = k.take<float>(L * 4 * L); c->enc_W1_64 = k.take<float>(2 * L * 4 * L);
    c->dec_W0h_64 = k.take<float>(L * 4 * L); c->dec_W1_64 = k.take<float>(2 * L * 4 * L);
    c->enc_W0h_b3 = k.take<uint16_t>(3 * L * 4 * L); c->enc_W1_b3 = k.take<uint16_t>(3 * 2 * L * 4 * L);
    c->dec_W0h_b3 = k.take<uint16_t>(3 * L * 4 * L); c->dec_W1_b3 = k.take<uint16_t>(3 * 2 * L * 4 * L);
#ifdef N2NMN_DIAG      // (operand planes of gemm_dma3_kernel: diagnostic library only, kernels_gemm.hip)
    c->eht_W_b3 = k.take<uint16_t>(3 * (size_t)c->KpL * L); c->att_W_b3 = k.take<uint16_t>(3 * (size_t)c->KpL * L);
    if (Mp % 128 == 0) {
      c->find_img_b3 = k.take<uint16_t>(3 * (size_t)c->KpD * Mp);
      c->fsp_img_b3 = k.take<uint16_t>(3 * (size_t)c->KpD * Mp);
    }
#endif
  }
  c->eht_W_p = k.take<float>((size_t)c->KpL * L);
  c->att_W_t = k.take<float>(L * L);
  c->att_W_p = k.take<float>((size_t)c->KpL * L);
  c->find_img_p = k.take<float>((size_t)c->KpD * Mp);
  c->fsp_img_p = k.take<float>((size_t)c->KpD * Mp);
  c->dec_emb_cat = k.take<float>((V + 1) * (size_t)d.embed_dim_nmn);
  if (c->big_heads) {
    const size_t n = (size_t)round_up(d.map_dim, 32) * round_up(d.num_choices, 64);
    c->wans_de_p = k.take<float>(n);
    if (d.variant == N2NMN_VARIANT_CLEVR) c->wans_sp_p = k.take<float>(n);
  }
  if (d.variant == N2NMN_VARIANT_VQA && d.qpn_hidden > 0) {
    c->qpn_W1_p = k.take<float>((size_t)round_up(2 * d.lstm_dim, 32) * round_up(d.qpn_hidden, 64));
    c->qpn_W2_p = k.take<float>((size_t)round_up(d.qpn_hidden, 32) * round_up(d.num_choices, 64));
  }
  for (int i = 0; i < 5; ++i) c->wtxt_pad[i] = k.take<float>(E * Mp);
  for (int i = 0; i < 5; ++i) c->btxt_pad[i] = k.take<float>(Mp);
  c->tr_At = k.take<float>((size_t)((d.kernel_size * d.kernel_size + 1 + 3) & ~3) * round_up(d.map_dim, 16));
  if (d.variant == N2NMN_VARIANT_CLEVR && d.num_vocab_txt <= 4096)
    for (int i = 0; i < 5; ++i) {
      c->wtxt_pk[i] = k.take<float>((size_t)c->KpE * Mp);
      c->ew[i] = k.take<float>((size_t)d.num_vocab_txt * Mp);
    }
  for (int i = 0; i < 4; ++i) c->watt_pad[i] = k.take<float>(D * Mp);
  for (int i = 0; i < 3; ++i) c->we_pad[i] = k.take<float>(Mp);
  for (int i = 0; i < 4; ++i) c->batt_pad[i] = k.take<float>(Mp);
  c->packs.dev = k.take<PackJob>(kMaxPackJobs);
  c->packs_rest.dev = k.take<PackJob>(kMaxPackJobs);
  c->packs_infer.dev = k.take<PackJob>(8);
  c->P = k.take<int32_t>(V * 3);
  c->Wv = k.take<int32_t>(3 * V * 4);
  c->bv = k.take<int32_t>(V * 4);
  c->token_op = k.take<int32_t>(V);
  (void)N; (void)T; (void)Td; (void)HW; (void)HWp;
  return align_up(k.off, 256);
}

// lays out the per-context workspace (activations, module arena, program tables)
static size_t carve_workspace(n2nmn_ctx* c, char* base) {
  const n2nmn_dims& d = c->d;
  const size_t L = d.lstm_dim, E = d.embed_dim_txt, N = d.N, T = d.T_encoder, Td = d.T_decoder,
               HW = (size_t)d.H * d.W;
  const size_t Mp = c->Mp, HWp = c->HWp;
  Carver k(base);
  // recurrent state: one contiguous block so the encoder can clear it with one memset
  // (block A and its bf16 planes are one allocation: enc_prepare clears both with one range)
  float* st = k.take<float>(25 * N * L);
  c->st_A = st; c->hb_A = reinterpret_cast<uint16_t*>(st + 10 * N * L);
  c->eh0[0] = st; c->eh0[1] = st + N * L; c->eh1[0] = st + 2 * N * L; c->eh1[1] = st + 3 * N * L;
  c->ec0 = st + 4 * N * L; c->ec1 = st + 5 * N * L;
  // final encoder state in ORIGINAL row order (written at each row's last valid step)
  c->fc0 = st + 6 * N * L; c->fh0 = st + 7 * N * L; c->fc1 = st + 8 * N * L; c->fh1 = st + 9 * N * L;
  // dropped copies of the layer-0 outputs (encoder_dropout / decoder_dropout: DropoutWrapper on
  // every layer but the last, nmn3_netgen_att.py:17-44 of both model families)
  float* stb = k.take<float>(25 * N * L);         // block B (10 N L floats) + its planes
  c->st_B = stb; c->hb_B = reinterpret_cast<uint16_t*>(stb + 10 * N * L);
  for (int i = 0; i < 2; ++i) { c->ehd[i] = stb + (size_t)i * N * L; c->dhd[i] = stb + (size_t)(2 + i) * N * L; }
  if (c->big_vocab) {
    c->xproj = k.take<float>(T * N * 4 * L);
    c->iota = k.take<int32_t>(T * N);
  }
  c->perm = k.take<int32_t>(N);
  c->nact = k.take<int32_t>(T + 1);
  c->enc_rows = k.take<int32_t>(T * N);
  c->enc_rows_n = k.take<int32_t>(4);
  c->dlen = k.take<int32_t>(N); c->dperm = k.take<int32_t>(N); c->dnact = k.take<int32_t>(Td + 1);
  c->drows = k.take<int32_t>(Td * N); c->drows_n = k.take<int32_t>(4);
  float* ds = c->st_B + 4 * N * L;                // decoder states: inside block B
  c->dh0[0] = ds; c->dh0[1] = ds + N * L; c->dh1[0] = ds + 2 * N * L; c->dh1[1] = ds + 3 * N * L;
  c->dc0 = ds + 4 * N * L; c->dc1 = ds + 5 * N * L;
  c->enc_out = k.take<float>(T * N * L);
  c->eht = k.take<float>(T * N * L);
  c->qbuf = k.take<float>(Td * N * L);
  c->dec_h1_all = k.take<float>(Td * N * L);
  c->ent_t = k.take<float>(Td * N);
  c->dh1_rm = k.take<float>(N * L);
  if (d.variant == N2NMN_VARIANT_VQA && d.qpn_hidden > 0) {
    c->qpn_h = k.take<float>(N * 2 * L);
    c->qpn_hid = k.take<float>(N * (size_t)d.qpn_hidden);
  }
  c->state = k.take<int32_t>(N * 3);
  c->next_idx = k.take<int32_t>(N);
  c->tokens = k.take<int32_t>(Td * N);
  c->tprobs = k.take<float>(Td * N);
  c->negent = k.take<float>(N);
  c->atts = k.take<float>(Td * T * N);
  c->word_vecs = k.take<float>(Td * N * E);
  // module workspace
  c->arena = k.take<float>((size_t)c->max_nodes * HWp);
  c->tmap = k.take<float>((size_t)c->max_text * Mp);
  c->pfc = k.take<float>((size_t)c->max_pool * 2 * POOL_PARTS * Mp);
  c->mfind = k.take<float>(N * HW * Mp);
  c->mfsp = k.take<float>(N * HW * Mp);
  if (c->big_heads) {
    c->ev_out = k.take<float>((size_t)c->max_pool * Mp);
    c->ev_rows = k.take<int32_t>(2 * (size_t)c->max_pool);
  }
  c->dev_nodes = k.take<DevNode>(c->max_nodes);
  c->dev_tab = k.take<int32_t>(c->max_tab);
  c->dsched = k.take<int32_t>(2 * (1 + 3 * SCHED_MAX_T) + 8);
  c->walk_stats = k.take<unsigned long long>(WALK_STATS);
  c->wtmap = k.take<float>(Td * N * Mp);
  c->watt = k.take<float>(Td * N * HWp);
  c->wpjob = k.take<int32_t>(N);
  c->wpw = k.take<float>(N * 2 * HWp);
  c->wptm = k.take<float>(N * Mp);
  c->wpooled = k.take<float>(N * 2 * (size_t)d.D);
  c->wpfc = k.take<float>(N * 2 * WALK_POOL_PARTS * Mp);
  c->wprog = k.take<WalkProg>(N);
  c->wfpart = k.take<float>(N * Td * WALK_POOL_PARTS * Mp);
  // job lists of the staged walker: a region per nesting level (kernels.h, WalkArgs::hoff), sized for the
  // worst case -- at most T / (lv + 2) nodes of level lv per question, per operator
  {
    size_t off = 0;
    for (int lv = 0; lv < WALK_HLEVELS; ++lv) {
      c->whoff[lv] = (int)off;
      off += (size_t)2 * WALK_MAX_BATCHES * N * (WALK_MAX_T / (lv + 2));
    }
    c->whoff[WALK_HLEVELS] = (int)off;
    c->whjobs = k.take<int32_t>(off);
  }
  c->wfblist = k.take<int32_t>((size_t)WALK_MAX_BATCHES * N);
  c->wcnt = k.take<int32_t>(2 * WALK_CNT);
  c->wplist = k.take<int32_t>((size_t)2 * WALK_MAX_BATCHES * N);
  return align_up(k.off, 256);
}

const char* kFamilyNames[F_COUNT] = {
  "lstm_step(enc L0+L1)", "lstm_step(dec L0 | pipelined L0+L1)", "lstm_step(dec L1)",
  "lstm_step(linear q)", "dec_attn", "gemm_pk(encoder_h_transform)", "word_vecs", "textmap", "gemm_pk(conv_image)",
  "att_ops", "pool", "heads",
  "lstm_bwd_step", "gemm_tn(weight grads)", "backward misc (modules/attention/gemm_nt)",
  "optimiser", "walk(layout walker)", "gemm_pkn(encoder_h_transform + q [+ conv_image when it rides in phase 1])",
  "walk_find(Find / Filter epilogues over the conv_image maps)", "walk_tmap(text maps from the attention tables)",
  "sched(layout assembler + level scheduler on the device)"};


hipStream_t S(n2nmn_stream s) { return reinterpret_cast<hipStream_t>(s); }

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_error(std::string(what) + ": " + hipGetErrorString(e));
    return N2NMN_EHIP;
  }
  return N2NMN_OK;
}

ModuleWeights module_weights(const n2nmn_ctx* c) {
  ModuleWeights w{};
  auto m = [&](int id) { return (const float*)c->vars[id].mirror; };
  const int txtw[5] = {V_FIND_TXT_W, V_FSP_TXT_W, V_TR_TXT_W, V_SP_TXT_W, V_DE_TXT_W};
  (void)txtw;
  for (int i = 0; i < 5; ++i) { w.Wtxt[i] = c->wtxt_pad[i]; w.btxt[i] = c->btxt_pad[i]; }
  for (int i = 0; i < 3; ++i) w.we[i] = c->we_pad[i];
  w.be[0] = m(V_FIND_E_B); w.be[1] = m(V_FSP_E_B); w.be[2] = m(V_TR_E_B);
  w.Kt = m(V_TR_MAPS_W); w.bt = m(V_TR_MAPS_B); w.trA = root(c)->tr_At;
  const int attw[4] = {V_FSP_ATT_W, V_SP_ATT0_W, V_SP_ATT1_W, V_DE_ATT_W};
  (void)attw;
  for (int i = 0; i < 4; ++i) { w.Watt[i] = c->watt_pad[i]; w.batt[i] = c->batt_pad[i]; }
  const int answ[7] = {V_EXIST_W, V_COUNT_W, V_EQ_W, V_MORE_W, V_LESS_W, V_SP_E_W, V_DE_E_W};
  for (int i = 0; i < 7; ++i) { w.Wans[i] = m(answ[i]); w.bans[i] = m(answ[i] + 1); }
  return w;
}

// tile choice of the pipelined two-layer step launches (launch_lstm_step's `wide`)
static int lstm_wide(const n2nmn_ctx* c) {
  return (c->mode == N2NMN_MODE_THROUGHPUT || c->mode == N2NMN_MODE_THROUGHPUT_BF16X3) ? 2
         : c->mode == N2NMN_MODE_THROUGHPUT_KSPLIT ? 1 : 0;
}

// a switch of n2nmn_debug_set: this context's entry, else its parent's, else the default
const char* knob_str(const n2nmn_ctx* c, const char* key) {
  for (const n2nmn_ctx* p = c; p; p = p->parent) {
    auto it = p->knobs.find(key);
    if (it != p->knobs.end()) return it->second.c_str();
  }
  return nullptr;
}
int knob_int(const n2nmn_ctx* c, const char* key, int dflt) {
  const char* v = knob_str(c, key);
  return v ? atoi(v) : dflt;
}

// encoder steps with at most this many active rows use the K-split tiles (needs the host lengths)
static int tile_min_rows(const n2nmn_ctx* c) { return knob_int(c, "tile_min_rows", 192); }

// state buffers (eh0/eh1/dh0/dh1) are k-interleaved [L/4][R][4] with R = capacity N
// split-operand bf16 mode of a pass of N rows (a training forward keeps the exact kernels)
static bool lstm_b3(const n2nmn_ctx* c, int N) {
  return c->mode == N2NMN_MODE_THROUGHPUT_BF16X3 && N >= 128 && !c->rec && root(c)->b3_on && c->enc_W0h_b3;
}
static uint16_t* planes_of(const n2nmn_ctx* c, const float* p) {
  const size_t n = (size_t)10 * c->d.N * c->d.lstm_dim;
  if (p >= c->st_A && p < c->st_A + n) return c->hb_A + 3 * (size_t)(p - c->st_A);
  if (p >= c->st_B && p < c->st_B + n) return c->hb_B + 3 * (size_t)(p - c->st_B);
  return nullptr;
}
// plane companions of a job's state buffers and weights (LstmJob::A0b ...)
static void attach_planes(const n2nmn_ctx* c, LstmJob& j) {
  j.A0b = planes_of(c, j.A0); j.A1b = j.A1 ? planes_of(c, j.A1) : nullptr;
  j.h_new_b = planes_of(c, j.h_new); j.fin_h_b = j.fin_h ? planes_of(c, j.fin_h) : nullptr;
  j.h_drop_b = j.h_drop ? planes_of(c, j.h_drop) : nullptr;
  j.Wb3 = j.Wp64 == c->enc_W0h_64 ? c->enc_W0h_b3 : j.Wp64 == c->enc_W1_64 ? c->enc_W1_b3 :
          j.Wp64 == c->dec_W0h_64 ? c->dec_W0h_b3 : j.Wp64 == c->dec_W1_64 ? c->dec_W1_b3 : nullptr;
}

void packed_state(const n2nmn_ctx* c, LstmJob& j) {
  j.a_rs = 4; j.a_ks = 4 * c->d.N; j.hp_R = c->d.N;
}
void rowmajor_a(const n2nmn_ctx* c, LstmJob& j) {
  j.a_rs = c->d.lstm_dim; j.a_ks = 4; j.hp_R = 0;
}

// The two hoisted conv_image problems (FindModule / FindSamePropertyModule weight sets) over the N
// images of a batch; tokens != nullptr gates the second per row tile by the layouts.
static void conv_image_problems(n2nmn_ctx* c, const float* image_feat, int N, const int32_t* tokens,
                                int T_dec, GemmArgs ga[2]) {
  const n2nmn_dims& d = c->d;
  const int HW = d.H * d.W;
  const n2nmn_ctx* r = root(c);
  for (int fsp = 0; fsp < 2; ++fsp) {
    GemmArgs& g = ga[fsp];
    g = GemmArgs{};
    g.A = image_feat; g.lda = d.D; g.M = N * HW; g.K = d.D; g.group_size = HW;
    g.Bp = fsp ? c->fsp_img_p : c->find_img_p; g.Np = c->Mp; g.Kp = c->KpD;
    if (lstm_b3(c, N)) g.Bp3 = fsp ? c->fsp_img_b3 : c->find_img_b3;       // (opt-in mode: gemm_dma3_kernel)
    g.bias = c->vars[fsp ? V_FSP_IMG_B : V_FIND_IMG_B].mirror; g.N = d.map_dim;
    g.C = fsp ? c->mfsp : c->mfind; g.ldc = c->Mp; g.n_store = c->Mp;
    if (fsp && tokens) {
      g.gate_tokens = tokens; g.gate_token_op = r->token_op; g.gate_T = T_dec; g.gate_N = N;
      g.gate_V = d.num_vocab_nmn; g.gate_op = N2NMN_OP_FIND_SAME_PROPERTY; g.gate_rows = HW;
    }
  }
}

// defer_eht != nullptr: the encoder_h_transform GEMM is described there instead of launched -- the
// decoder puts it into its own GEMM launch (launch_gemm_pkn), nothing in between reads `eht`
int encoder_impl(n2nmn_ctx* c, const n2nmn_seq2seq_io* io, hipStream_t s, GemmArgs* defer_eht) {
  const n2nmn_dims& d = c->d;
  N2_REQUIRE(is_committed(c), N2NMN_ENOWEIGHT, "encoder_forward: weights not committed");
  N2_REQUIRE(io && io->input_seq && io->seq_length, N2NMN_EINVAL, "encoder_forward: null input");
  const int T = io->T_enc, N = io->N, L = d.lstm_dim;
  // (a context that trains refreshes the 64-column tiles of lstm_tile_kernel on its side stream: only
  // a pass that can reach that kernel waits for them -- a training forward, 64 rows, must not)
  if (lstm_wide(c) >= 2 && N >= 128) train_infer_wait(root(c), s);
  N2_REQUIRE(T >= 1 && T <= d.T_encoder && N >= 1 && N <= d.N, N2NMN_ECAPACITY,
             "encoder_forward: T_enc / N exceed the context capacity");
  // encoder_h_transform over the rows inside their question's length only (44 % of T*N are past it at
  // the eval mix; every reader of `eht` takes the bias vector for those: DecStepArgs::eht_bias) -- when
  // the GEMM rides in the decoder's launch of a large pass and nobody was promised the full matrix
  const bool eht_rows = knob_int(c, "eht_rows", 1) != 0 && defer_eht && !c->rec && (size_t)T * N >= 8192;
  const bool b3 = lstm_b3(c, N);      // (with the planes of block A behind it, see carve_workspace)
  launch_enc_prepare(io->seq_length, N, T, c->perm, c->nact, c->eh0[0], (b3 ? 25 : 10) * (size_t)d.N * L, s,
                     eht_rows ? c->enc_rows_n : nullptr);
  if (eht_rows) launch_enc_rows(io->seq_length, T, N, c->enc_rows, c->enc_rows_n, s);
  const float* W0x_bias_table = c->enc_xtab;
  if (c->big_vocab) {
    // x . W_x + b of the batch's own words: one GEMM whose A rows are gathered by word index
    GemmArgs g{};
    g.A = c->vars[V_ENC_EMB].mirror; g.lda = d.embed_dim_txt; g.M = T * N; g.K = d.embed_dim_txt;
    g.group_idx = io->input_seq; g.group_size = 1; g.src_rows = d.num_vocab_txt;
    g.Bp = c->enc_W0x_p; g.Np = 4 * L; g.Kp = c->KpE; g.bias = c->enc_b0_t; g.N = 4 * L;
    g.C = c->xproj; g.ldc = 4 * L; g.n_store = 4 * L;
    ProfScope ps(c, F_GEMM_EHT, 2.0 * T * N * d.embed_dim_txt * 4.0 * L,
                 4.0 * ((double)T * N * (d.embed_dim_txt + 4.0 * L) + 4.0 * L * d.embed_dim_txt), s);
    launch_gemm_pk(g, s);
    W0x_bias_table = c->xproj;
  }
  // rows still active at step t, when the caller handed over a host copy of the lengths: the tail of
  // the length-sorted encoder has too few row blocks for the 64 x 64 tiles (a launch of <= 256
  // workgroups leaves the K = 2L tiles alone on their CUs), the K-split tiles scale with the rows
  std::vector<int> act_host;
  // (the split-operand mode too: the K-split kernels write the bf16 planes of the states they produce
  // when a job carries plane pointers, so a row's final state reaches the decoder with its planes)
  auto rows_active = [&](const int32_t* lens, std::vector<int>& act) {
    std::vector<int> cnt(T + 2, 0);
    for (int n = 0; n < N; ++n) cnt[std::min(std::max(lens[n], 0), T)] += 1;
    act.assign(T, 0);
    for (int t = T - 1, run = 0; t >= 0; --t) {    // rows with length > t
      run += cnt[t + 1];
      act[t] = run;
    }
  };
  if (io->seq_length_host && (c->mode == N2NMN_MODE_THROUGHPUT || c->mode == N2NMN_MODE_THROUGHPUT_BF16X3))
    rows_active(io->seq_length_host, act_host);
  // The profile counters report EXECUTED work: the step kernels skip the 16-row MFMA tiles past the
  // active rows of the length-sorted state, the encoder_h_transform GEMM runs over the listed rows.  A
  // profiled pass may synchronise, so without a host copy of the lengths it fetches them (counters only:
  // the launches are the unprofiled pass's).
  std::vector<int> act_prof;
  std::vector<int32_t> len_prof;
  const int32_t* len_cnt = io->seq_length_host;
  if (!len_cnt && c->prof_on) {
    len_prof.resize(N);
    N2_HIP(hipStreamSynchronize(s));
    N2_HIP(hipMemcpy(len_prof.data(), io->seq_length, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
    len_cnt = len_prof.data();
  }
  if (len_cnt && c->prof_on && !c->rec) rows_active(len_cnt, act_prof);
  auto exec_rows = [&](int t) -> double {          // rows of the 16-row tiles that hold an active row at step t
    if (act_prof.empty() || t < 0 || t >= T) return N;
    return std::min(N, round_up(act_prof[t], 16));
  };
  // software-pipelined over time: launch k runs layer-0 step k and layer-1 step k-1
  for (int k = 0; k <= T; ++k) {
    LstmJob jobs[2];
    LstmJob& j0 = jobs[0];
    j0 = LstmJob{};
    j0.active = k < T;
    packed_state(c, j0);
    j0.A0 = c->eh0[(k + 1) & 1]; j0.A1 = nullptr; j0.K = L; j0.Wp = c->enc_W0h_t; j0.Wp64 = c->enc_W0h_64;
    j0.xtab = W0x_bias_table; j0.xidx_const = 0;
    j0.xidx = (c->big_vocab ? c->iota : io->input_seq) + (size_t)k * N;
    j0.bias = nullptr; j0.c_in = c->ec0; j0.c_out = c->ec0; j0.ntiles = L / 4;
    j0.h_old = c->eh0[(k + 1) & 1]; j0.h_new = c->eh0[k & 1];
    j0.out_seq = nullptr; j0.seq_len = io->seq_length; j0.t = k;
    j0.perm = c->perm; j0.n_active = c->nact + (k < T ? k : T - 1); j0.fin_c = c->fc0; j0.fin_h = c->fh0;
    LstmJob& j1 = jobs[1];
    j1 = LstmJob{};
    const int st = k - 1;
    j1.active = st >= 0;
    packed_state(c, j1);
    j1.A0 = c->eh0[st & 1]; j1.A1 = c->eh1[(st + 1) & 1]; j1.K = 2 * L; j1.Wp = c->enc_W1_t; j1.Wp64 = c->enc_W1_64;
    if (io->drop_enc0) {                 // DropoutWrapper on layer 0's output (models_vqa)
      const size_t nl = (size_t)N * L;
      if (j0.active) {
        j0.drop = io->drop_enc0 + (size_t)k * nl; j0.h_drop = c->ehd[k & 1];
        j0.save_hd = c->rec ? c->rec->eh0d + (size_t)k * nl : nullptr;
      }
      j1.A0 = c->ehd[st & 1];
    }
    j1.xtab = nullptr; j1.xidx = nullptr; j1.bias = c->enc_b1_t;
    j1.c_in = c->ec1; j1.c_out = c->ec1; j1.ntiles = L / 4;
    j1.h_old = c->eh1[(st + 1) & 1]; j1.h_new = c->eh1[st & 1];
    j1.out_seq = st >= 0 ? c->enc_out + (size_t)st * N * L : nullptr;
    j1.seq_len = io->seq_length; j1.t = st;
    j1.perm = c->perm; j1.n_active = c->nact + (st >= 0 ? st : 0); j1.fin_c = c->fc1; j1.fin_h = c->fh1;
    if (c->rec) {                      // training: keep gates / cell / hidden sequences
      const size_t nl = (size_t)N * L;
      if (j0.active) {
        j0.save_gates = c->rec->eg0 + (size_t)k * nl; j0.save_c = c->rec->ec0s + (size_t)(k + 1) * nl;
        j0.save_h = c->rec->eh0s + (size_t)(k + 1) * nl;
      }
      if (j1.active) {
        j1.save_gates = c->rec->eg1 + (size_t)st * nl; j1.save_c = c->rec->ec1s + (size_t)(st + 1) * nl;
        j1.save_h = c->rec->eh1s + (size_t)(st + 1) * nl;
      }
    }
    {
      const double r0 = exec_rows(k), r1 = exec_rows(st);
      const double fl = 2.0 * 4 * L * ((j0.active ? r0 * L : 0) + (j1.active ? r1 * 2 * L : 0));
      const double by = 4.0 * ((j0.active ? (double)L * 4 * L + 3.0 * r0 * L : 0) +
                               (j1.active ? 2.0 * L * 4 * L + 5.0 * r1 * L : 0));
      ProfScope ps(c, F_LSTM_ENC, fl, by, s);
      int wide = lstm_wide(c);
      if (!act_host.empty() && act_host[std::min(k, T - 1)] <= tile_min_rows(c)) wide = 1;
      if (b3) {
        attach_planes(c, j0); attach_planes(c, j1);
        if (wide == 1) {              // tail of the length-sorted encoder: exact-fp32 K-split tiles (+ planes)
          launch_lstm_step(jobs, 2, N, L, 64, s, 1);
        } else {
          N2_REQUIRE(lstm_tile3_supported(jobs, 2, L), N2NMN_EINVAL, "encoder_forward: bf16x3 mode: unsupported job");
          launch_lstm_tile3(jobs, 2, N, L, s);
        }
      } else {
        launch_lstm_step(jobs, 2, N, L, 64, s, wide);
      }
    }
  }
  // encoder_h_transformed = fc(encoder_outputs)          (nmn3_netgen_att.py:102-106)
  GemmArgs g{};
  g.A = c->enc_out; g.lda = L; g.M = T * N; g.K = L; g.group_idx = nullptr; g.group_size = 1;
  g.Bp = c->eht_W_p; g.Np = L; g.Kp = c->KpL; g.bias = c->vars[V_EHT_B].mirror; g.N = L;
  if (b3) g.Bp3 = c->eht_W_b3;
  g.C = c->eht; g.ldc = L; g.n_store = L;
  if (eht_rows) {
    g.group_idx = c->enc_rows; g.group_size = 1; g.src_rows = T * N;
    g.c_row_idx = c->enc_rows; g.m_dev = c->enc_rows_n;
  }
  c->eht_partial = eht_rows;
  c->eht_listed_rows = -1;
  if (eht_rows && len_cnt) {                    // what the launch really computes, for the profile counters
    long rows = 0;
    for (int n = 0; n < N; ++n) rows += std::min(std::max(len_cnt[n], 0), T);
    c->eht_listed_rows = rows;
  }
  if (defer_eht) {
    *defer_eht = g;
  } else {
    const double rows = c->eht_listed_rows >= 0 ? (double)c->eht_listed_rows : (double)T * N;
    ProfScope ps(c, F_GEMM_EHT, 2.0 * rows * L * L, 4.0 * (2.0 * rows * L + (double)L * L), s);
    launch_gemm_pk(g, s);
  }
  c->enc_T = T; c->enc_N = N; c->enc_seq = io->input_seq; c->enc_len = io->seq_length;
  const size_t nl = sizeof(float) * (size_t)N * L;
  if (io->encoder_outputs)
    N2_HIP(hipMemcpyAsync(io->encoder_outputs, c->enc_out, nl * T, hipMemcpyDeviceToDevice, s));
  if (io->encoder_h_transformed) {
    N2_REQUIRE(!c->eht_partial, N2NMN_EINVAL, "encoder_forward: encoder_h_transformed copy of a partial matrix");
    N2_HIP(hipMemcpyAsync(io->encoder_h_transformed, c->eht, nl * T, hipMemcpyDeviceToDevice, s));
  }
  if (io->encoder_states) {
    launch_unpack_h(c->fc0, io->encoder_states, N, L, d.N, s);       // c shares h's packed layout
    launch_unpack_h(c->fh0, io->encoder_states + (size_t)N * L, N, L, d.N, s);
    launch_unpack_h(c->fc1, io->encoder_states + (size_t)2 * N * L, N, L, d.N, s);
    launch_unpack_h(c->fh1, io->encoder_states + (size_t)3 * N * L, N, L, d.N, s);
  }
  return check_launch("encoder_forward");
}

// pre / npre: GEMM problems that must be complete before the attention runs (the deferred
// encoder_h_transform); they and -- with io->image_feat -- the hoisted conv_image problems share the
// decoder's own GEMM launch.
int decoder_impl(n2nmn_ctx* c, const n2nmn_seq2seq_io* io, hipStream_t s, const GemmArgs* pre, int npre) {
  const n2nmn_dims& d = c->d;
  N2_REQUIRE(is_committed(c), N2NMN_ENOWEIGHT, "decoder_forward: weights not committed");
  N2_REQUIRE(has_tables(c), N2NMN_ENOWEIGHT,
             "decoder_forward: validity tables (assembler P/W/b) not set");
  N2_REQUIRE(io, N2NMN_EINVAL, "decoder_forward: null io");
  train_infer_wait(root(c), s);        // (a context that trains packs the decoder's operands on its side stream)
  N2_REQUIRE(c->enc_T > 0 && io->N == c->enc_N && io->T_enc == c->enc_T, N2NMN_EINVAL,
             "decoder_forward: no matching encoder results in the context");
  const int T = c->enc_T, N = c->enc_N, L = d.lstm_dim, Td = io->T_dec, V = d.num_vocab_nmn;
  N2_REQUIRE(Td >= 1 && Td <= d.T_decoder, N2NMN_ECAPACITY, "decoder_forward: T_dec too large");
  N2_REQUIRE(!io->use_gt_layout || io->gt_layout, N2NMN_EINVAL,
             "decoder_forward: use_gt_layout without gt_layout");
  int32_t* tokens = io->predicted_tokens ? io->predicted_tokens : c->tokens;
  float* tprobs = io->token_probs ? io->token_probs : c->tprobs;
  float* negent = io->neg_entropy ? io->neg_entropy : c->negent;
  float* atts = io->atts ? io->atts : c->atts;
  float* wv = io->word_vecs ? io->word_vecs : c->word_vecs;
  const double fl0 = 2.0 * N * L * 4 * L, fl1 = 2.0 * N * 2 * L * 4 * L;
  DecStepArgs a{};
  a.eht = c->eht; a.eout = c->enc_out; a.seq_len = c->enc_len; a.v = c->vars[V_ATT_V].mirror;
  a.order = c->perm;            // enc_prepare's length ranking of this pass's rows
  a.eht_bias = c->vars[V_EHT_B].mirror;      // rows past a question's length: the bias, never `eht`
  N2_REQUIRE(!c->eht_partial || (a.eht_bias && !c->rec), N2NMN_EINVAL,
             "decoder_forward: partial encoder_h_transform without the bias substitution");
  a.Wy = c->vars[V_TOK_W].mirror; a.by = c->vars[V_TOK_B].mirror; a.P = c->P; a.Wv = c->Wv;
  a.bv = c->bv; a.use_gt = io->use_gt_layout; a.T = T; a.N = N; a.L = L; a.V = V;
  a.state = c->state;
  const double att_fl = (double)N * (4.0 * T * L + 2.0 * 2 * L * V);
  const double att_by = 4.0 * N * (2.0 * T * L + 2.0 * L + T + V) + 4.0 * 2 * L * V;
  // Teacher forcing knows every token up front: the LSTM no longer waits for the attention /
  // token step, so the two layers pipeline over time like the encoder and the attention of ALL
  // T_dec steps runs as one launch.
  const bool batched = io->use_gt_layout && !io->forced_tokens && !io->sample_uniforms;
  // ---- eos_retire (N2NMN_S2S_EOS_RETIRE, include/n2nmn.h): rows leave the decoder at their layout's first
  // <eos>.  Preconditions: an inference pass of >= 128 rows in a throughput mode (the tile kernels take the
  // rows in ranked order and a device-side live count), nothing fetched that needs the dead steps.
  a.q = c->qbuf; a.out = c->dec_h1_all; a.gt = io->gt_layout; a.Td = Td;     // (dec_question_supported reads these)
  const bool retire = batched && (io->flags & N2NMN_S2S_EOS_RETIRE) &&
                      !c->rec && !io->drop_dec0 && !io->token_scores &&
                      lstm_wide(c) >= 2 && N >= 128 && Td <= 63 && root(c)->have_token_ops &&
                      dec_len_supported(a, Td);
  int kmax = Td;                      // last launch of the pipelined loop
  std::vector<int> dact_host;         // rows with layout length > t (host copy of the lengths given)
  if (retire) {
    launch_dec_len(io->gt_layout, root(c)->token_op, V, Td, N, c->dlen, s);
    launch_enc_prepare(c->dlen, N, Td, c->dperm, c->dnact, nullptr, 0, s, c->drows_n);
    launch_enc_rows(c->dlen, Td, N, c->drows, c->drows_n, s);
    // the encoder's final states (ORIGINAL row order) in ranked order; block B's dropout buffers are free
    // in an inference pass without dropout and carry plane companions like every state buffer
    // (GatherArgs pairs the planes with src[0] / src[2]: h0, c0, h1, c1)
    const float* src[4] = {c->fh0, c->fc0, c->fh1, c->fc1};
    float* dst[4] = {c->ehd[0], c->ehd[1], c->dhd[0], c->dhd[1]};
    if (lstm_b3(c, N)) {
      const uint16_t* sb[2] = {planes_of(c, c->fh0), planes_of(c, c->fh1)};
      uint16_t* db[2] = {planes_of(c, c->ehd[0]), planes_of(c, c->dhd[0])};
      launch_gather_state(src, dst, sb, db, c->dperm, N, L, d.N, s);
    } else {
      launch_gather_state(src, dst, nullptr, nullptr, c->dperm, N, L, d.N, s);
    }
    if (io->gt_length_host) {
      std::vector<int> cnt(Td + 2, 0);
      int longest = 0;
      for (int n = 0; n < N; ++n) {
        const int l = std::min(std::max(io->gt_length_host[n], 0), Td);
        cnt[l] += 1; longest = std::max(longest, l);
      }
      dact_host.assign(Td + 1, 0);
      for (int t = Td - 1, run = 0; t >= 0; --t) { run += cnt[t + 1]; dact_host[t] = run; }
      kmax = longest;
    }
  }
  // (profile counters of a retired pass without host lengths: a profiled pass may synchronise)
  std::vector<int> dact_prof = dact_host;
  if (retire && dact_prof.empty() && c->prof_on) {
    std::vector<int32_t> lh(N);
    N2_HIP(hipStreamSynchronize(s));
    N2_HIP(hipMemcpy(lh.data(), c->dlen, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
    std::vector<int> cnt(Td + 2, 0);
    for (int n = 0; n < N; ++n) cnt[std::min(std::max(lh[n], 0), Td)] += 1;
    dact_prof.assign(Td + 1, 0);
    for (int t = Td - 1, run = 0; t >= 0; --t) { run += cnt[t + 1]; dact_prof[t] = run; }
  }
  c->dec_retired = retire;
  if (batched) {
    for (int k = 0; k <= kmax; ++k) {
      LstmJob jobs[2];
      LstmJob& j0 = jobs[0];
      j0 = LstmJob{};
      j0.active = k < Td;
      packed_state(c, j0);
      j0.A0 = k == 0 ? c->fh0 : c->dh0[(k + 1) & 1]; j0.K = L; j0.Wp = c->dec_W0h_t; j0.Wp64 = c->dec_W0h_64;
      j0.ntiles = L / 4; j0.xtab = c->dec_xtab;
      j0.xidx = k == 0 ? nullptr : io->gt_layout + (size_t)(k - 1) * N; j0.xidx_const = V;
      j0.c_in = k == 0 ? c->fc0 : c->dc0; j0.c_out = c->dc0;
      j0.h_old = j0.A0; j0.h_new = c->dh0[k & 1];
      LstmJob& j1 = jobs[1];
      j1 = LstmJob{};
      const int st = k - 1;
      j1.active = st >= 0;
      packed_state(c, j1);
      j1.A0 = c->dh0[st & 1]; j1.A1 = st == 0 ? c->fh1 : c->dh1[(st + 1) & 1];
      j1.K = 2 * L; j1.Wp = c->dec_W1_t; j1.Wp64 = c->dec_W1_64; j1.ntiles = L / 4; j1.bias = c->dec_b1_t;
      if (io->drop_dec0) {
        const size_t nl = (size_t)N * L;
        if (j0.active) {
          j0.drop = io->drop_dec0 + (size_t)k * nl; j0.h_drop = c->dhd[k & 1];
          j0.save_hd = c->rec ? c->rec->dh0d + (size_t)k * nl : nullptr;
        }
        j1.A0 = c->dhd[st & 1];
      }
      j1.c_in = st == 0 ? c->fc1 : c->dc1; j1.c_out = c->dc1;
      j1.h_old = j1.A1; j1.h_new = c->dh1[st & 1];
      j1.out_seq = st >= 0 ? c->dec_h1_all + (size_t)st * N * L : nullptr;
      if (c->rec) {
        const size_t nl = (size_t)N * L;
        if (j0.active) {
          j0.save_gates = c->rec->dg0 + (size_t)k * nl; j0.save_c = c->rec->dc0s + (size_t)(k + 1) * nl;
          j0.save_h = c->rec->dh0s + (size_t)(k + 1) * nl;
        }
        if (j1.active) {
          j1.save_gates = c->rec->dg1 + (size_t)st * nl; j1.save_c = c->rec->dc1s + (size_t)(st + 1) * nl;
          j1.save_h = c->rec->dh1s + (size_t)(st + 1) * nl;
        }
      }
      int wide = lstm_wide(c);
      double live0 = N, live1 = N;      // rows the launch computes (16-row MFMA tiles), for the counters
      if (retire) {
        // state rows in ranked order (dperm), the rows alive at a step are a prefix whose length the
        // kernels read from the device; a row past its layout keeps its state (as dynamic_rnn does past
        // a question's length: the same kernel path)
        if (k == 0) { j0.A0 = c->ehd[0]; j0.c_in = c->ehd[1]; j0.h_old = j0.A0; }
        if (st == 0) { j1.A1 = c->dhd[0]; j1.c_in = c->dhd[1]; j1.h_old = j1.A1; }
        j0.perm = j1.perm = c->dperm; j0.seq_len = j1.seq_len = c->dlen; j0.t = k; j1.t = st;
        j0.n_active = c->dnact + std::min(k, Td - 1); j1.n_active = c->dnact + std::max(st, 0);
        if (!dact_prof.empty()) {
          live0 = std::min(N, round_up(j0.active ? dact_prof[k] : 0, 16));
          live1 = std::min(N, round_up(j1.active ? dact_prof[st] : 0, 16));
        }
        if (!dact_host.empty()) {
          const int a0 = j0.active ? dact_host[k] : 0, a1 = j1.active ? dact_host[st] : 0;
          if (std::max(a0, a1) <= tile_min_rows(c)) wide = 1;      // few rows left: the K-split tiles scale with rows
        }
      }
      ProfScope ps(c, F_LSTM_DEC0, (j0.active ? fl0 * live0 / N : 0) + (j1.active ? fl1 * live1 / N : 0),
                   (j0.active ? 4.0 * ((double)L * 4 * L + 3.0 * live0 * L) : 0) +
                   (j1.active ? 4.0 * (2.0 * L * 4 * L + 5.0 * live1 * L) : 0), s);
      if (lstm_b3(c, N)) {
        attach_planes(c, j0); attach_planes(c, j1);
        if (wide == 1) {              // (exact-fp32 K-split tiles; they write the planes of what they produce)
          launch_lstm_step(jobs, 2, N, L, 64, s, 1);
        } else {
          N2_REQUIRE(lstm_tile3_supported(jobs, 2, L), N2NMN_EINVAL, "decoder_forward: bf16x3 mode: unsupported job");
          launch_lstm_tile3(jobs, 2, N, L, s);
        }
      } else {
        launch_lstm_step(jobs, 2, N, L, 64, s, wide);
      }
    }
    // q = out . W_a + b_a for all steps (nmn3_netgen_att.py:185), in ONE launch with whatever else
    // is due before the attention / the layout walk: encoder_h_transform, conv_image
    {
      GemmArgs list[4];
      int nl = 0;
      double fl = 0, by = 0;
      for (int i = 0; i < npre && nl < 2; ++i) {
        list[nl++] = pre[i];
        // (a listed-row problem computes *m_dev rows, not M: counted when the host knows the lengths)
        const double rows = pre[i].m_dev && c->eht_listed_rows >= 0 ? (double)c->eht_listed_rows : pre[i].M;
        fl += 2.0 * rows * pre[i].K * pre[i].N;
        by += 4.0 * (rows * (pre[i].K + pre[i].N) + (double)pre[i].K * pre[i].N);
      }
      GemmArgs& gq = list[nl++];
      gq = GemmArgs{};
      gq.A = c->dec_h1_all; gq.lda = L; gq.M = Td * N; gq.K = L; gq.group_size = 1;
      gq.Bp = c->att_W_p; gq.Np = L; gq.Kp = c->KpL; gq.bias = c->vars[V_ATT_B].mirror; gq.N = L;
      if (lstm_b3(c, N)) gq.Bp3 = c->att_W_b3;
      gq.C = c->qbuf; gq.ldc = L; gq.n_store = L;
      double qrows = (double)Td * N;
      if (retire) {                      // q of the live (step, row) pairs only (the eht GEMM's row-list form)
        gq.group_idx = c->drows; gq.group_size = 1; gq.src_rows = Td * N;
        gq.c_row_idx = c->drows; gq.m_dev = c->drows_n;
        if (!dact_host.empty()) {
          long tot = 0;
          for (int t = 0; t < Td; ++t) tot += dact_host[t];
          gq.M = (int)std::max<long>(tot, 1);
        }
        if (!dact_prof.empty()) {
          long tot = 0;
          for (int t = 0; t < Td; ++t) tot += dact_prof[t];
          qrows = (double)tot;
        }
      }
      fl += 2.0 * qrows * L * L; by += 4.0 * ((double)L * L + 2.0 * qrows * L);
      // the hoisted conv_image problems ride in this launch when the list has room (it holds four);
      // otherwise they get a launch of their own below -- never dropped: the walker reads their maps
      const bool conv = io->image_feat != nullptr;
      const bool conv_here = conv && nl <= 2;
      GemmArgs cvl[2];
      double cfl = 0, cby = 0;
      if (conv) {
        const bool gate = root(c)->have_token_ops;
        conv_image_problems(c, io->image_feat, N, gate ? io->gt_layout : nullptr, Td, cvl);
        const double HW = d.H * d.W, frac = gate ? 1.1 : 2.0;     // gated share: see n2nmn_conv_image
        cfl = frac * 2.0 * N * HW * d.D * d.map_dim;
        cby = frac * 4.0 * N * HW * (d.D + c->Mp) + 4.0 * d.D * d.map_dim;
      }
      if (conv_here) {
        list[nl++] = cvl[0]; list[nl++] = cvl[1];
        fl += cfl; by += cby;
      }
      {
        ProfScope ps(c, nl > 1 ? F_GEMM_MULTI : F_LINEAR_Q, fl, by, s);
        launch_gemm_pkn(list, nl, s);
      }
      if (conv && !conv_here) {
        ProfScope ps(c, F_CONV_IMAGE, cfl, cby, s);
        launch_gemm_pkn(cvl, 2, s);
      }
    }
    a.q = c->qbuf; a.out = c->dec_h1_all; a.gt = io->gt_layout; a.uni = nullptr; a.forced = nullptr;
    a.tokens = tokens; a.tprobs = tprobs; a.ent_t = c->ent_t; a.atts = atts;
    a.scores = io->token_scores; a.next_idx = nullptr;
    a.Td = Td;
    a.dec_len = retire ? c->dlen : nullptr;
    double att_live = 1.0;             // share of the (step, row) pairs the attention evaluates
    if (retire && !dact_prof.empty()) {
      long tot = 0;
      for (int t = 0; t < Td; ++t) tot += dact_prof[t];
      att_live = (double)tot / ((double)Td * N);
    }
    if (c->rec) {
      a.ctx_out = c->rec->ctx;
      if (!a.scores) a.scores = c->rec->tscores;
      a.valid_bits = c->rec->valid_bits;
    }
    {
      ProfScope ps(c, F_DEC_STEP, Td * att_fl * att_live, Td * att_by * att_live, s);
      launch_dec_attn(a, Td, s);
    }
  } else {
    GemmArgs cv[2];
    if (io->image_feat) conv_image_problems(c, io->image_feat, N, nullptr, Td, cv);
    if (npre > 0 || io->image_feat) {
      GemmArgs list[4];
      int nl = 0;
      double fl = 0, by = 0;
      for (int i = 0; i < npre && nl < 3; ++i) {
        list[nl++] = pre[i];
        const double rows = pre[i].m_dev && c->eht_listed_rows >= 0 ? (double)c->eht_listed_rows : pre[i].M;
        fl += 2.0 * rows * pre[i].K * pre[i].N;
        by += 4.0 * (rows * (pre[i].K + pre[i].N) + (double)pre[i].K * pre[i].N);
      }
      if (io->image_feat) {
        list[nl++] = cv[0];
        fl += 2.0 * cv[0].M * cv[0].K * cv[0].N;
        by += 4.0 * ((double)cv[0].M * (cv[0].K + c->Mp) + (double)cv[0].K * cv[0].N);
      }
      ProfScope ps(c, nl > 1 ? F_GEMM_MULTI : (io->image_feat ? F_CONV_IMAGE : F_GEMM_EHT), fl, by, s);
      launch_gemm_pkn(list, nl, s);
    }
    launch_dec_init(c->state, N, Td, s);
    // ---- eos_retire for sequential decoding: a row that has emitted <eos> leaves the recurrence.  After
    // every step dec_compact_kernel re-partitions the state rows (live ones in front, gathered into the
    // alternate state buffers) and the launches of the next step run over that dense prefix; the host
    // issues every launch (it never learns the counts) and the workgroups past the prefix return at once.
    // Inference only, >= 128 rows in a throughput mode, dec_attn_seq_kernel's dimensions.
    a.q = c->qbuf; a.out = c->dh1_rm;
    const bool retire_seq = (io->flags & N2NMN_S2S_EOS_RETIRE) && !c->rec && !io->drop_dec0 &&
                            !io->token_scores && !io->forced_tokens && !io->use_gt_layout &&
                            lstm_wide(c) >= 2 && N >= 128 &&
                            root(c)->have_token_ops && root(c)->eos_token >= 0 && root(c)->retire_ok &&
                            dec_seq_retire_supported(a);
    c->dec_retired = retire_seq;
    // state buffers of the loop: h ping-pongs as always; c is updated in place in cb0 / cb1; a compaction
    // moves all four into their alternates (block B's dropout buffers are free without dropout)
    float* hb0[2] = {c->dh0[0], c->dh0[1]};
    float* hb1[2] = {c->dh1[0], c->dh1[1]};
    float* cb0[2] = {c->dc0, c->dhd[0]};
    float* cb1[2] = {c->dc1, c->dhd[1]};
    int32_t* pbuf[2] = {c->dperm, c->drows};
    int h0w = 0, h1w = 0, c0i = 0, c1i = 0, pi = 0;     // buffer step t WRITES h into; buffer holding c; perm in use
    for (int t = 0; t < Td; ++t) {
      const int32_t* live_perm = retire_seq && t > 0 ? pbuf[pi] : nullptr;
      const int32_t* live_n = retire_seq && t > 0 ? c->dnact + t : nullptr;
      double live = N;                     // rows the recurrent launches compute, for the profile counters
      if (live_n && c->prof_on) {          // (a profiled pass may synchronise)
        int32_t nl = N;
        N2_HIP(hipStreamSynchronize(s));
        N2_HIP(hipMemcpy(&nl, live_n, sizeof(int32_t), hipMemcpyDeviceToHost));
        live = std::min(N, round_up(std::max(nl, 0), 16));
      }
      LstmJob j0{};
      j0.active = 1;
      packed_state(c, j0);
      j0.A0 = t == 0 ? c->fh0 : hb0[h0w ^ 1]; j0.K = L; j0.Wp = c->dec_W0h_t; j0.Wp64 = c->dec_W0h_64;
      j0.ntiles = L / 4;
      j0.xtab = c->dec_xtab; j0.xidx = t == 0 ? nullptr : c->next_idx; j0.xidx_const = V;  // <go>
      j0.c_in = t == 0 ? c->fc0 : cb0[c0i]; j0.c_out = cb0[c0i];
      j0.h_old = j0.A0; j0.h_new = hb0[h0w];
      j0.perm = live_perm; j0.n_active = live_n;
      if (io->drop_dec0) {             // sampling from the network with dropout (policy gradient)
        j0.drop = io->drop_dec0 + (size_t)t * N * L; j0.h_drop = c->dhd[t & 1];
      }
      {
        ProfScope ps(c, F_LSTM_DEC0, fl0 * live / N, 4.0 * ((double)L * 4 * L + 3.0 * live * L), s);
        // (throughput mode, >= 128 rows: the 64 x 64 LDS-DMA tiles, one round of 512 workgroups per
        // layer -- the greedy decoder's two layers cannot share a launch, token t feeds layer 0 of t+1)
        if (lstm_b3(c, N)) {
          attach_planes(c, j0);
          N2_REQUIRE(lstm_tile3_supported(&j0, 1, L), N2NMN_EINVAL, "decoder_forward: bf16x3 mode: unsupported job");
          launch_lstm_tile3(&j0, 1, N, L, s);
        } else {
          launch_lstm_step(&j0, 1, N, L, 32, s, lstm_wide(c));
        }
      }
      LstmJob j1{};
      j1.active = 1;
      packed_state(c, j1);
      j1.out_seq = c->dh1_rm;          // row-major copy of the top-layer h for dec_attn (by ORIGINAL row)
      j1.A0 = io->drop_dec0 ? c->dhd[t & 1] : hb0[h0w];
      j1.A1 = t == 0 ? c->fh1 : hb1[h1w ^ 1]; j1.K = 2 * L;
      j1.Wp = c->dec_W1_t; j1.Wp64 = c->dec_W1_64; j1.ntiles = L / 4; j1.bias = c->dec_b1_t;
      j1.c_in = t == 0 ? c->fc1 : cb1[c1i]; j1.c_out = cb1[c1i];
      j1.h_old = j1.A1; j1.h_new = hb1[h1w];
      j1.perm = live_perm; j1.n_active = live_n;
      {
        ProfScope ps(c, F_LSTM_DEC1, fl1 * live / N, 4.0 * (2.0 * L * 4 * L + 5.0 * live * L), s);
        if (lstm_b3(c, N)) {
          attach_planes(c, j1);
          N2_REQUIRE(lstm_tile3_supported(&j1, 1, L), N2NMN_EINVAL, "decoder_forward: bf16x3 mode: unsupported job");
          launch_lstm_tile3(&j1, 1, N, L, s);
        } else {
          launch_lstm_step(&j1, 1, N, L, 32, s, lstm_wide(c));
        }
      }
      LstmJob jq{};                    // q = out . W_a + b_a            (nmn3_netgen_att.py:185)
      packed_state(c, jq);
      jq.hp_R = 0;
      jq.active = 1; jq.mode = 1; jq.A0 = hb1[h1w]; jq.K = L; jq.Wp = c->att_W_t;     // (q in STATE-row order)
      jq.ntiles = L / 16; jq.bias = c->vars[V_ATT_B].mirror; jq.h_new = c->qbuf; jq.ldo = L;
      {
        ProfScope ps(c, F_LINEAR_Q, 2.0 * N * L * L, 4.0 * ((double)L * L + 2.0 * N * L), s);
        launch_lstm_step(&jq, 1, N, L, 32, s);
      }
      a.q = c->qbuf; a.out = c->dh1_rm;
      a.gt = io->gt_layout ? io->gt_layout + (size_t)t * N : nullptr;
      a.uni = io->sample_uniforms ? io->sample_uniforms + (size_t)t * N : nullptr;
      a.forced = io->forced_tokens ? io->forced_tokens + (size_t)t * N : nullptr;
      a.tokens = tokens + (size_t)t * N; a.tprobs = tprobs + (size_t)t * N;
      a.ent_t = c->ent_t + (size_t)t * N; a.atts = atts + (size_t)t * T * N;
      a.scores = io->token_scores ? io->token_scores + (size_t)t * N * V : nullptr;
      a.next_idx = c->next_idx;
      a.live_perm = live_perm; a.live_n = live_n; a.eos_token = root(c)->eos_token;
      {
        ProfScope ps(c, F_DEC_STEP, att_fl * live / N, att_by * live / N, s);
        launch_dec_attn(a, 1, s);
      }
      h0w ^= 1; h1w ^= 1;
      if (retire_seq && t + 1 < Td) {
        // rows whose token of this step is <eos> retire: live rows to the front of the alternate buffers
        const float* src[4] = {hb0[h0w ^ 1], cb0[c0i], hb1[h1w ^ 1], cb1[c1i]};
        float* dst[4] = {hb0[h0w], cb0[c0i ^ 1], hb1[h1w], cb1[c1i ^ 1]};
        if (lstm_b3(c, N)) {               // the split planes of the two hidden states move with them
          const uint16_t* sb[2] = {planes_of(c, src[0]), planes_of(c, src[2])};
          uint16_t* db[2] = {planes_of(c, dst[0]), planes_of(c, dst[2])};
          launch_dec_compact(a.tokens, root(c)->token_op, V, live_perm, live_n, pbuf[pi ^ 1], c->dnact + t + 1,
                             src, dst, sb, db, N, L, d.N, s);
        } else {
          launch_dec_compact(a.tokens, root(c)->token_op, V, live_perm, live_n, pbuf[pi ^ 1], c->dnact + t + 1,
                             src, dst, nullptr, nullptr, N, L, d.N, s);
        }
        h0w ^= 1; h1w ^= 1; c0i ^= 1; c1i ^= 1; pi ^= 1;
      }
    }
    if (io->image_feat) {              // FindSameProperty maps of the layouts the decoder chose
      conv_image_problems(c, io->image_feat, N, root(c)->have_token_ops ? tokens : nullptr, Td, cv);
      const double frac = root(c)->have_token_ops ? 0.1 : 1.0;
      ProfScope ps(c, F_CONV_IMAGE, frac * 2.0 * cv[1].M * cv[1].K * cv[1].N,
                   frac * 4.0 * cv[1].M * (cv[1].K + c->Mp), s);
      launch_gemm_pkn(cv + 1, 1, s);
    }
  }
  if (!(io->flags & N2NMN_S2S_NO_WORD_VECS)) {
    const double E = d.embed_dim_txt;
    ProfScope ps(c, F_WORD_VECS, 2.0 * Td * T * N * E, 4.0 * N * (T * E + Td * T + Td * E), s);
    launch_word_vecs(atts, c->enc_seq, c->vars[V_ENC_EMB].mirror, Td, T, N, d.embed_dim_txt, wv,
                     tprobs, c->ent_t, negent,
                     io->log_seq_prob ? io->log_seq_prob : (c->rec ? c->rec->lsp : nullptr), s);
  }
  return check_launch("decoder_forward");
}

// scores += fc2(drop(relu(fc1(drop(h_concat)))))   (models_vqa/question_prior_net.py:10-28).
// drop_h / drop_fc1: dropout multipliers (training) or nullptr.  What the backward pass needs stays
// in the workspace: qpn_h (the dropped input) and qpn_hid (the dropped ReLU output).
int qpn_forward(n2nmn_ctx* c, int N, float* scores, const float* drop_h, const float* drop_fc1,
                hipStream_t s) {
  const n2nmn_dims& d = c->d;
  const int L = d.lstm_dim, Hq = d.qpn_hidden, C = d.num_choices;
  const n2nmn_ctx* r = root(c);
  train_infer_wait(r, s);
  // h_concat = [h of layer 0, h of layer 1]  (question_prior_net.py:14-20), row-major [N][2L]
  launch_unpack_h2(c->fh0, c->fh1, c->qpn_h, N, L, d.N, s);
  if (drop_h) launch_ew_mul(c->qpn_h, drop_h, (size_t)N * 2 * L, s);
  GemmArgs g{};
  g.A = c->qpn_h; g.lda = 2 * L; g.M = N; g.K = 2 * L; g.group_size = 1;
  g.Bp = r->qpn_W1_p; g.Np = round_up(Hq, 64); g.Kp = round_up(2 * L, 32);
  g.bias = r->vars[V_QPN_B1].mirror; g.N = Hq; g.C = c->qpn_hid; g.ldc = Hq; g.n_store = Hq;
  g.relu = 1;                                      // fc_relu (util/cnn.py:121-126)
  launch_gemm_pk(g, s);
  if (drop_fc1) launch_ew_mul(c->qpn_hid, drop_fc1, (size_t)N * Hq, s);
  GemmArgs g2{};
  g2.A = c->qpn_hid; g2.lda = Hq; g2.M = N; g2.K = Hq; g2.group_size = 1;
  g2.Bp = r->qpn_W2_p; g2.Np = round_up(C, 64); g2.Kp = round_up(Hq, 32);
  g2.bias = r->vars[V_QPN_B2].mirror; g2.N = C; g2.C = scores; g2.ldc = C; g2.n_store = C;
  g2.accumulate = 1;                               // scores = scores_nmn + scores_qpn
  launch_gemm_pk(g2, s);
  return check_launch("question_prior_net");
}

int run_program(n2nmn_ctx* c, Program& p, const float* feat, const float* word_vecs,
                       int N_full, float* scores, const float* ext0, const float* ext1,
                       float* att_out, int att_out_first, int att_out_count, hipStream_t s,
                       int stages) {
  // stages (bit mask, RP_ALL by default): RP_PREP = checks, INVALID_EXPR zero rows, program upload;
  // RP_CONV = the hoisted conv_image GEMMs, which need only the image features (the training
  // forward runs them on its side stream beside the encoder); RP_REST = everything else
  const n2nmn_dims& d = c->d;
  const bool do_prep = stages & RP_PREP, do_conv = stages & RP_CONV, do_rest = stages & RP_REST;
  if (do_conv || do_rest) train_infer_wait(root(c), s);
  N2_REQUIRE(is_committed(c), N2NMN_ENOWEIGHT, "execute_program: weights not committed");
  N2_REQUIRE(N_full >= 1 && N_full <= d.N, N2NMN_ECAPACITY, "execute_program: N_full > capacity");
  const int nn = (int)p.dev_nodes.size();
  N2_REQUIRE(nn <= c->max_nodes && p.num_text <= c->max_text && p.num_pool <= c->max_pool &&
                 (int)p.tab.size() <= c->max_tab,
             N2NMN_ECAPACITY, "execute_program: program larger than the context workspace");
  if (do_prep) for (const DevNode& nd : p.dev_nodes) {
    N2_REQUIRE(nd.op == OP_INPUT || (nd.n < N_full && nd.t < d.T_decoder), N2NMN_EINVAL,
               "execute_program: batch_idx / time_idx out of range");
    if (d.variant == N2NMN_VARIANT_VQA)
      N2_REQUIRE(nd.op == OP_INPUT || nd.op == N2NMN_OP_FIND || nd.op == N2NMN_OP_AND ||
                     nd.op == N2NMN_OP_FIND_SAME_PROPERTY || nd.op == N2NMN_OP_DESCRIBE,
                 N2NMN_EKEY, "execute_program: operator does not exist in models_vqa");
  }
  const int HW = d.H * d.W, C = d.num_choices;
  if (do_prep && scores && p.num_rows > 0)
    N2_HIP(hipMemsetAsync(scores, 0, sizeof(float) * (size_t)p.num_rows * C, s));  // INVALID_EXPR
  if (nn == 0) return N2NMN_OK;
  if (do_prep) {
    // nodes + tables travel through a pinned staging slot so the upload is truly asynchronous;
    // a slot is reused only after the copy that last read it has completed
    const size_t nb = sizeof(DevNode) * nn, tb = sizeof(int32_t) * p.tab.size();
    const int slot = c->stage_next;
    c->stage_next = (slot + 1) % n2nmn_ctx::kStage;
    N2_HIP(hipEventSynchronize(c->stage_ev[slot]));
    std::memcpy(c->stage[slot], p.dev_nodes.data(), nb);
    if (tb) std::memcpy(c->stage[slot] + align_up(nb, 256), p.tab.data(), tb);
    N2_HIP(hipMemcpyAsync(c->dev_nodes, c->stage[slot], nb, hipMemcpyHostToDevice, s));
    if (tb)
      N2_HIP(hipMemcpyAsync(c->dev_tab, c->stage[slot] + align_up(nb, 256), tb,
                            hipMemcpyHostToDevice, s));
    N2_HIP(hipEventRecord(c->stage_ev[slot], s));
  }
  // externally supplied attention maps (module_forward): node i <- ext[time_idx][batch_idx]
  for (int i = 0; do_rest && i < nn; ++i) {
    const DevNode& nd = p.dev_nodes[i];
    if (nd.op != OP_INPUT) continue;
    const float* src = (nd.t == 0 ? ext0 : ext1);
    N2_REQUIRE(src, N2NMN_EINVAL, "module_forward: missing attention input");
    N2_HIP(hipMemcpyAsync(c->arena + (size_t)i * c->HWp, src + (size_t)nd.n * HW,
                          sizeof(float) * HW, hipMemcpyDeviceToDevice, s));
  }
  ModuleWeights w = module_weights(c);
  ModuleBuffers b{};
  b.nodes = c->dev_nodes; b.tab = c->dev_tab; b.arena = c->arena; b.tmap = c->tmap;
  b.pfc = c->pfc; b.mfind = c->mfind; b.mfsp = c->mfsp; b.feat = feat; b.word_vecs = word_vecs;
  b.scores = scores; b.N_full = N_full; b.H = d.H; b.W = d.W; b.D = d.D; b.M = d.map_dim;
  b.pooled = c->rec ? c->rec->pooled : nullptr;
  b.vqa = d.variant == N2NMN_VARIANT_VQA;
  b.ev_out = c->big_heads ? c->ev_out : nullptr; b.ev_rows = c->ev_rows; b.ev_stride = c->max_pool;
  b.Mp = c->Mp; b.wl_cap = d.map_dim * C <= 10240 ? d.map_dim * C : 0; b.E = d.embed_dim_txt; b.C = C; b.HWp = c->HWp; b.ksize = d.kernel_size;
  const double dE = d.embed_dim_txt, dM = d.map_dim, dD = d.D, dHW = HW, dC = C, dMp = c->Mp;
  for (const Launch& l : p.launches) {
    const bool is_conv = l.kind == LK_CONV_FIND || l.kind == LK_CONV_FSP;
    if (is_conv ? !do_conv : !do_rest) continue;
    switch (l.kind) {
      case LK_TEXTMAP: {
        // groups of <= TM_GROUP nodes; one [E,M] weight stream per group
        ProfScope ps(c, F_TEXTMAP, 2.0 * p.num_text * dE * dM,
                     4.0 * (l.count * dE * dM + p.num_text * (dE + dMp)), s);
        launch_textmap(w, b, l.offset, l.count, s);
        break;
      }
      case LK_CONV_FIND:
      case LK_CONV_FSP: {
        const bool fsp = l.kind == LK_CONV_FSP;
        GemmArgs g{};
        g.A = feat; g.lda = d.D; g.M = l.count * HW; g.K = d.D;
        g.group_idx = c->dev_tab + l.offset; g.group_size = HW; g.src_rows = N_full * HW;
        g.Bp = fsp ? c->fsp_img_p : c->find_img_p; g.Np = c->Mp; g.Kp = c->KpD;
        if (lstm_b3(c, N_full)) g.Bp3 = fsp ? c->fsp_img_b3 : c->find_img_b3;   // (opt-in mode: gemm_dma3_kernel)
        g.bias = c->vars[fsp ? V_FSP_IMG_B : V_FIND_IMG_B].mirror; g.N = d.map_dim;
        g.C = fsp ? c->mfsp : c->mfind; g.ldc = c->Mp; g.n_store = c->Mp;
        ProfScope ps(c, F_CONV_IMAGE, 2.0 * l.count * dHW * dD * dM,
                     4.0 * (l.count * dHW * (dD + dMp) + dD * dM), s);
        launch_gemm_pk(g, s);
        break;
      }
      case LK_ATT: {
        // algorithmic bytes: conv_image map read by the Find-type epilogues + attention maps
        double by = 0, fl = 0;
        for (int i = 0; i < l.count; ++i) {
          const int* e = p.tab.data() + l.offset + 4 * i;
          const DevNode& nd = p.dev_nodes[e[0]];
          const double part = 1.0 / e[2];
          if (nd.op == N2NMN_OP_FIND || nd.op == N2NMN_OP_FILTER ||
              nd.op == N2NMN_OP_FIND_SAME_PROPERTY) {
            by += part * 4.0 * (dHW * dMp + 2 * dMp + dHW); fl += part * 5.0 * dHW * dM;
          } else if (nd.op == N2NMN_OP_TRANSFORM) {
            by += part * 4.0 * (2 * dHW) + 4.0 * (d.kernel_size * d.kernel_size + 3) * dM;
            fl += part * dHW * dM * (2.0 * d.kernel_size * d.kernel_size + 5);
          } else {
            by += 4.0 * 3 * dHW; fl += 2.0 * (2 * dHW + 4) * dC;
          }
        }
        ProfScope ps(c, F_ATT_OPS, fl, by, s);
        launch_att_ops(w, b, l.offset, l.count, s);
        break;
      }
      case LK_POOL: {
        // per job: the [HW, D] feature map once (shared by both inputs of SameProperty),
        // the attention logits, the partial fc_att rows; fc_att weights once per launch
        double by = 4.0 * dD * dM, fl = 0;
        for (int i = 0; i < l.count; i += POOL_PARTS) {
          const DevNode& nd = p.dev_nodes[p.tab[l.offset + 2 * i]];
          const double nin = nd.op == N2NMN_OP_SAME_PROPERTY ? 2 : 1;
          by += 4.0 * (dHW * dD + nin * dHW + nin * POOL_PARTS * dMp);
          fl += nin * (2.0 * dHW * dD + 2.0 * dD * dM + 3.0 * dHW);
        }
        ProfScope ps(c, F_POOL, fl, by, s);
        launch_pool(w, b, l.offset, l.count, s);
        break;
      }
      case LK_HEAD: {
        ProfScope ps(c, F_HEADS, l.count * (2.0 * dM * dC + 8.0 * dM),
                     4.0 * (l.count * (2.0 * POOL_PARTS * dMp + dMp + dC) + dM * dC), s);
        launch_heads(w, b, l.offset, l.count, s);
        if (c->big_heads) {       // fc_eltwise of the whole launch as a GEMM, rows scattered to scores
          bool any_sp = false, any_de = false;
          for (int i = 0; i < l.count; ++i) {
            const int op = p.dev_nodes[p.tab[l.offset + i]].op;
            any_sp |= op == N2NMN_OP_SAME_PROPERTY; any_de |= op == N2NMN_OP_DESCRIBE;
          }
          const n2nmn_ctx* r = root(c);
          for (int which = 0; which < 2; ++which) {
            if (!(which == 0 ? any_de : any_sp)) continue;
            GemmArgs g{};
            g.A = c->ev_out; g.lda = c->Mp; g.M = l.count; g.K = d.map_dim; g.group_size = 1;
            g.Bp = which == 0 ? r->wans_de_p : r->wans_sp_p; g.Np = round_up(C, 64);
            g.Kp = round_up(d.map_dim, 32);
            g.bias = c->vars[which == 0 ? V_DE_E_B : V_SP_E_B].mirror; g.N = C; g.C = scores;
            g.ldc = C; g.n_store = C; g.c_row_idx = c->ev_rows + which * c->max_pool;
            launch_gemm_pk(g, s);
          }
        }
        break;
      }
      default: break;
    }
  }
  if (do_rest && att_out && att_out_count > 0) {
    N2_HIP(hipMemcpy2DAsync(att_out, sizeof(float) * HW,
                            c->arena + (size_t)att_out_first * c->HWp, sizeof(float) * c->HWp,
                            sizeof(float) * HW, att_out_count, hipMemcpyDeviceToDevice, s));
  }
  return check_launch("execute_program");
}


// The level path with the program assembled and scheduled ON THE DEVICE (sched_kernel): what
// n2nmn_execute_program does for a host-assembled program, for layouts that exist only as device
// tokens -- models_vqa (no layout walker for its dimensions) and, as a cross-check of the scheduler,
// any other variant.  Nothing is read back: every level of the capacity (T_dec) gets its three
// persistent-grid launches, which find their work tables and lengths in HBM; the Describe /
// SameProperty fc_eltwise of a large answer vocabulary runs once, as a GEMM over the questions, after
// the last level.  (models_vqa/nmn3_model.py:55-121, exp_vqa/eval_vqa2.py:103-137.)
int run_tokens_levels(n2nmn_ctx* c, const int32_t* tokens, int T_dec, int N, const float* feat,
                      const float* word_vecs, float* scores, int32_t* validity, hipStream_t s) {
  const n2nmn_dims& d = c->d;
  N2_REQUIRE(is_committed(c), N2NMN_ENOWEIGHT, "execute_tokens: weights not committed");
  N2_REQUIRE(root(c)->have_token_ops, N2NMN_EINVAL, "execute_tokens: call n2nmn_set_token_ops first");
  N2_REQUIRE(N >= 1 && N <= d.N, N2NMN_ECAPACITY, "execute_tokens: N > capacity");
  N2_REQUIRE(T_dec >= 1 && T_dec <= d.T_decoder && T_dec <= SCHED_MAX_T, N2NMN_ECAPACITY,
             "execute_tokens: T_dec > capacity");
  train_infer_wait(root(c), s);
  const int HW = d.H * d.W, C = d.num_choices, levels = T_dec;
  N2_HIP(hipMemsetAsync(scores, 0, sizeof(float) * (size_t)N * C, s));   // INVALID_EXPR rows stay zero
  SchedArgs sa{};
  sa.tokens = tokens; sa.token_op = root(c)->token_op; sa.T = T_dec; sa.N = N; sa.V = d.num_vocab_nmn;
  sa.levels = levels; sa.nodes = c->dev_nodes; sa.tab = c->dev_tab; sa.tab_cap = c->max_tab;
  sa.dsched = c->dsched; sa.validity = validity;
  sa.ev_rows = c->big_heads ? c->ev_rows : nullptr; sa.ev_stride = c->max_pool;
  sa.overflow = c->dsched + 2 * (1 + 3 * SCHED_MAX_T);
  {
    ProfScope ps(c, F_SCHED, 0.0, 0.0, s);
    launch_sched(sa, s);
  }
  ModuleWeights w = module_weights(c);
  ModuleBuffers b{};
  b.nodes = c->dev_nodes; b.tab = c->dev_tab; b.arena = c->arena; b.tmap = c->tmap;
  b.pfc = c->pfc; b.mfind = c->mfind; b.mfsp = c->mfsp; b.feat = feat; b.word_vecs = word_vecs;
  b.scores = scores; b.N_full = N; b.H = d.H; b.W = d.W; b.D = d.D; b.M = d.map_dim;
  b.pooled = nullptr;
  b.vqa = d.variant == N2NMN_VARIANT_VQA;
  b.ev_out = c->big_heads ? c->ev_out : nullptr; b.ev_rows = c->ev_rows; b.ev_stride = c->max_pool;
  b.Mp = c->Mp; b.wl_cap = d.map_dim * C <= 10240 ? d.map_dim * C : 0; b.E = d.embed_dim_txt; b.C = C;
  b.HWp = c->HWp; b.ksize = d.kernel_size;
  b.dsched = c->dsched; b.ev_by_q = 1;
  const double dE = d.embed_dim_txt, dM = d.map_dim, dD = d.D, dHW = HW, dC = C, dMp = c->Mp;
  // persistent grids: level 0 carries every Find epilogue and most pooling jobs; deeper levels hold a
  // fraction of the questions (or nothing: an empty launch costs its workgroups' start-up only)
  const int g_text = std::min((N * T_dec + TM_GROUP - 1) / TM_GROUP + 5, 1024);
  auto g_att = [&](int l) { return l == 0 ? std::min(std::max(2 * N * FIND_PARTS, 256), 4096) : std::min(std::max(2 * N, 128), 1024); };
  auto g_pool = [&](int l) { return l <= 1 ? std::min(std::max(N * POOL_PARTS, 256), 2048) : std::min(std::max(N, 128), 512); };
  const int g_head = std::min(std::max(N, 64), 512);
  // (profile lines of this path count what a question of the reference's layout mix does: two text
  // maps, 1.6 Find-type epilogues, 1.1 pooling jobs -- the work is not known on the host)
  {
    ProfScope ps(c, F_TEXTMAP, 2.0 * 2.6 * N * dE * dM, 4.0 * (0.33 * N * dE * dM + 2.6 * N * (dE + dMp)), s);
    launch_textmap(w, b, 0, g_text, s, 0);
  }
  if (c->big_heads) N2_HIP(hipMemsetAsync(c->ev_out, 0, sizeof(float) * (size_t)N * c->Mp, s));
  for (int l = 0; l < levels; ++l) {
    {
      ProfScope ps(c, F_ATT_OPS, l == 0 ? 1.6 * N * 5.0 * dHW * dM : 0.0,
                   l == 0 ? 1.6 * N * 4.0 * (dHW * dMp + 2 * dMp + dHW) : 0.0, s);
      launch_att_ops(w, b, 0, g_att(l), s, 1 + 3 * l);
    }
    {
      ProfScope ps(c, F_POOL, l == 0 ? 1.1 * N * (2.0 * dHW * dD + 2.0 * dD * dM) : 0.0,
                   l == 0 ? 4.0 * dD * dM + 1.1 * N * 4.0 * (dHW * dD + dHW + POOL_PARTS * dMp) : 0.0, s);
      launch_pool(w, b, 0, g_pool(l), s, 2 + 3 * l);
    }
    {
      ProfScope ps(c, F_HEADS, l == 0 ? N * (2.0 * dM * dC + 8.0 * dM) : 0.0,
                   l == 0 ? 4.0 * (N * (2.0 * POOL_PARTS * dMp + dMp + dC) + dM * dC) : 0.0, s);
      launch_heads(w, b, 0, g_head, s, 3 + 3 * l);
    }
  }
  if (c->big_heads) {     // fc_eltwise over the questions: Describe rows, then SameProperty rows
    const n2nmn_ctx* r = root(c);
    for (int which = 0; which < 2; ++which) {
      if (which == 1 && !r->wans_sp_p) continue;
      GemmArgs g{};
      g.A = c->ev_out; g.lda = c->Mp; g.M = N; g.K = d.map_dim; g.group_size = 1;
      g.Bp = which == 0 ? r->wans_de_p : r->wans_sp_p; g.Np = round_up(C, 64);
      g.Kp = round_up(d.map_dim, 32);
      g.bias = c->vars[which == 0 ? V_DE_E_B : V_SP_E_B].mirror; g.N = C; g.C = scores;
      g.ldc = C; g.n_store = C; g.c_row_idx = c->ev_rows + which * c->max_pool;
      ProfScope ps(c, F_HEADS, 2.0 * N * dM * dC, 4.0 * (N * (dMp + dC) + dM * dC), s);
      launch_gemm_pk(g, s);
    }
  }
  return check_launch("execute_tokens (device-scheduled levels)");
}

static int finish_create(n2nmn_ctx* c, n2nmn_ctx* parent) {
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) { set_last_error(std::string("hipSetDevice: ") + hipGetErrorString(e)); return N2NMN_EHIP; }
  if (parent) {
    c->parent = parent;
    carve_weights(c, parent->base);           // same layout -> same pointers as the parent
    for (size_t i = 0; i < c->vars.size(); ++i) c->vars[i].set = true;
  } else {
    c->bytes = carve_weights(c, nullptr);
    e = hipMalloc(reinterpret_cast<void**>(&c->base), c->bytes);
    if (e != hipSuccess) {
      set_last_error(std::string("ctx_create: hipMalloc of ") + std::to_string(c->bytes) +
                     " bytes failed: " + hipGetErrorString(e));
      return N2NMN_EHIP;
    }
    carve_weights(c, c->base);
  }
  c->ws_bytes = carve_workspace(c, nullptr);
  e = hipMalloc(reinterpret_cast<void**>(&c->ws_base), c->ws_bytes);
  if (e != hipSuccess) {
    set_last_error(std::string("ctx_create: hipMalloc of ") + std::to_string(c->ws_bytes) +
                   " bytes failed: " + hipGetErrorString(e));
    if (!parent && c->base) (void)hipFree(c->base);
    return N2NMN_EHIP;
  }
  carve_workspace(c, c->ws_base);
  N2_HIP(hipMemset(c->wcnt, 0, sizeof(int32_t) * 2 * WALK_CNT));     // the walker's two counter sets
  // (the staged walker reports its deepest nesting through one host-mapped word: no copy, no sync)
  if (hipHostMalloc(reinterpret_cast<void**>(&c->walk_hint_host), 64, hipHostMallocMapped) == hipSuccess) {
    c->walk_hint_host[0] = 0;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&c->walk_hint_dev), c->walk_hint_host, 0) != hipSuccess)
      c->walk_hint_dev = nullptr;
  } else {
    (void)hipGetLastError();
    c->walk_hint_host = nullptr;
  }
  if (c->iota) {
    std::vector<int32_t> io((size_t)c->d.T_encoder * c->d.N);
    for (size_t i = 0; i < io.size(); ++i) io[i] = (int32_t)i;
    N2_HIP(hipMemcpy(c->iota, io.data(), sizeof(int32_t) * io.size(), hipMemcpyHostToDevice));
  }
  c->stage_bytes = align_up(sizeof(DevNode) * (size_t)c->max_nodes, 256) +
                   sizeof(int32_t) * (size_t)c->max_tab;
  for (int i = 0; i < n2nmn_ctx::kStage; ++i) {
    N2_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->stage[i]), c->stage_bytes, 0));
    N2_HIP(hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming));
  }
  if (n2nmn_program_create(&c->scratch_prog) != N2NMN_OK) return N2NMN_EINVAL;
  return N2NMN_OK;
}

}  // namespace n2nmn

namespace {
// a device allocation that is freed on EVERY exit of the debug / plumbing entries below (N2_HIP returns early)
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
  template <class T> T* as() const { return static_cast<T*>(p); }
};
}  // namespace

// =============================================================================================
extern "C" {

const char* n2nmn_last_error(void) { return g_last_error.c_str(); }
const char* n2nmn_version(void) { return "n2nmn-mi355x 0.1 (gfx950)"; }

int n2nmn_ctx_create(const n2nmn_dims* dims, int device, n2nmn_ctx** out) {
  N2_REQUIRE(dims && out, N2NMN_EINVAL, "ctx_create: null argument");
  const n2nmn_dims& d = *dims;
  N2_REQUIRE(d.num_layers == 2, N2NMN_EINVAL, "ctx_create: only num_layers == 2 is supported");
  N2_REQUIRE(d.lstm_dim > 0 && d.lstm_dim % 128 == 0, N2NMN_EINVAL,
             "ctx_create: lstm_dim must be a multiple of 128");
  N2_REQUIRE(d.embed_dim_txt % 4 == 0 && d.embed_dim_nmn == d.embed_dim_txt, N2NMN_EINVAL,
             "ctx_create: embed dims must be equal multiples of 4");
  N2_REQUIRE(d.D % (4 * POOL_PARTS) == 0 && d.D <= 4096, N2NMN_EINVAL,
             "ctx_create: D must be a multiple of 16, at most 4096");
  N2_REQUIRE(d.variant == N2NMN_VARIANT_CLEVR || d.variant == N2NMN_VARIANT_VQA, N2NMN_EINVAL,
             "ctx_create: unknown model variant");
  N2_REQUIRE(d.qpn_hidden >= 0 && (d.qpn_hidden == 0 || d.variant == N2NMN_VARIANT_VQA),
             N2NMN_EINVAL, "ctx_create: qpn_hidden is a models_vqa option");
  N2_REQUIRE(d.map_dim <= 1024, N2NMN_EINVAL, "ctx_create: map_dim must be <= 1024");
  N2_REQUIRE(d.kernel_size == 3 || d.kernel_size == 5, N2NMN_EINVAL,
             "ctx_create: kernel_size must be 3 or 5");
  N2_REQUIRE(d.num_vocab_nmn >= 2 && d.num_vocab_nmn <= 16, N2NMN_EINVAL,
             "ctx_create: num_vocab_nmn must be in [2, 16]");
  N2_REQUIRE(d.H > 0 && d.W > 0 && d.map_dim > 0 && d.num_choices > 0 && d.N > 0 &&
                 d.T_encoder > 0 && d.T_decoder > 0 && d.num_vocab_txt > 0,
             N2NMN_EINVAL, "ctx_create: non-positive dimension");
  N2_REQUIRE((size_t)d.T_encoder * d.embed_dim_txt + (size_t)d.T_decoder * d.T_encoder <= 16000,
             N2NMN_EINVAL, "ctx_create: T_encoder*embed_dim_txt too large for the LDS staging");
  n2nmn_ctx* c = new (std::nothrow) n2nmn_ctx();
  N2_REQUIRE(c, N2NMN_EINVAL, "ctx_create: out of host memory");
  c->d = d; c->device = device;
  build_vars(c);
  c->Mp = round_up(d.map_dim, 64);
  c->HWp = round_up(d.H * d.W, 32);
  c->KpE = round_up(d.embed_dim_txt, 32);
  c->KpL = round_up(d.lstm_dim, 32);
  c->KpD = round_up(d.D, 32);
  c->max_nodes = d.N * std::max(d.T_decoder, 4);
  c->max_text = c->max_nodes;
  c->max_pool = c->max_nodes;
  // work tables per node at most: stage A 4 ints x FIND_PARTS (Find epilogues are split in row parts) = 16,
  // stage B 2 ints x POOL_PARTS / ... = 8, stage C 1, text-map groups (2 + TM_GROUP) / TM_GROUP <= 2: 28
  // with these constants (sched_kernel flags an overflow as validity 0, never as silent zero logits)
  static_assert(4 * FIND_PARTS <= 16 && 2 * POOL_PARTS <= 16 && (2 + TM_GROUP + TM_GROUP - 1) / TM_GROUP <= 3,
                "max_tab: 28 ints per node no longer bound the work tables");
  c->max_tab = c->max_nodes * 28 + 4096;
  c->big_heads = (size_t)d.map_dim * d.num_choices > 65536;
  c->big_vocab = d.num_vocab_txt > 4096;
  int rc = finish_create(c, nullptr);
  if (rc != N2NMN_OK) { delete c; return rc; }
  *out = c;
  return N2NMN_OK;
}

/* A second context on the same device that SHARES the weight store of `parent` (mirrors, packed
 * operands, validity tables: committed through the parent) and owns only its workspace.  Lets
 * several batches be in flight on different streams / host threads without duplicating the
 * weights in HBM and L2.  The parent must outlive its forks. */
int n2nmn_ctx_fork(n2nmn_ctx* parent, n2nmn_ctx** out) {
  N2_REQUIRE(parent && out, N2NMN_EINVAL, "ctx_fork: null argument");
  N2_REQUIRE(!parent->parent, N2NMN_EINVAL, "ctx_fork: fork the root context");
  n2nmn_ctx* c = new (std::nothrow) n2nmn_ctx();
  N2_REQUIRE(c, N2NMN_EINVAL, "ctx_fork: out of host memory");
  c->d = parent->d; c->device = parent->device;
  build_vars(c);
  c->Mp = parent->Mp; c->HWp = parent->HWp; c->KpE = parent->KpE; c->KpL = parent->KpL;
  c->KpD = parent->KpD; c->max_nodes = parent->max_nodes; c->max_text = parent->max_text;
  c->max_pool = parent->max_pool; c->max_tab = parent->max_tab; c->big_heads = parent->big_heads;
  c->big_vocab = parent->big_vocab;
  int rc = finish_create(c, parent);
  if (rc != N2NMN_OK) { delete c; return rc; }
  *out = c;
  return N2NMN_OK;
}

int n2nmn_ctx_destroy(n2nmn_ctx* ctx) {
  if (!ctx) return N2NMN_OK;
  if (ctx->base && !ctx->parent) (void)hipFree(ctx->base);
  if (ctx->ws_base) (void)hipFree(ctx->ws_base);
  for (int i = 0; i < n2nmn_ctx::kStage; ++i) {
    if (ctx->stage[i]) (void)hipHostFree(ctx->stage[i]);
    if (ctx->stage_ev[i]) (void)hipEventDestroy(ctx->stage_ev[i]);
  }
  for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
  if (ctx->walk_hint_host) (void)hipHostFree(ctx->walk_hint_host);
  if (ctx->train) train_state_destroy(ctx->train);
  n2nmn_program_destroy(ctx->scratch_prog);
  delete ctx;
  return N2NMN_OK;
}

// the four recurrent weight matrices as three bf16 planes per 16-unit tile (lstm_tile3_kernel)
static void pack_b3(n2nmn_ctx* r, hipStream_t s) {
  const int L = r->d.lstm_dim, E0 = r->d.embed_dim_txt, E1 = r->d.embed_dim_nmn;
  launch_pack_tiles64_b3(r->vars[V_ENC_W0].mirror, 4 * L, E0, L, L, r->enc_W0h_b3, s);
  launch_pack_tiles64_b3(r->vars[V_ENC_W1].mirror, 4 * L, 0, 2 * L, L, r->enc_W1_b3, s);
  launch_pack_tiles64_b3(r->vars[V_DEC_W0].mirror, 4 * L, E1, L, L, r->dec_W0h_b3, s);
  launch_pack_tiles64_b3(r->vars[V_DEC_W1].mirror, 4 * L, 0, 2 * L, L, r->dec_W1_b3, s);
  // the PK-packed GEMM weights of a pass (they are packed earlier in the same commit)
  if (r->eht_W_b3) {
    launch_pack_pk_b3(r->eht_W_p, r->KpL, L, r->eht_W_b3, s);
    launch_pack_pk_b3(r->att_W_p, r->KpL, L, r->att_W_b3, s);
  }
  if (r->find_img_b3) {
    launch_pack_pk_b3(r->find_img_p, r->KpD, r->Mp, r->find_img_b3, s);
    launch_pack_pk_b3(r->fsp_img_p, r->KpD, r->Mp, r->fsp_img_b3, s);
  }
}

int n2nmn_ctx_set_mode(n2nmn_ctx* ctx, int mode) {
  N2_REQUIRE(ctx, N2NMN_EINVAL, "ctx_set_mode: null context");
  N2_REQUIRE(mode == N2NMN_MODE_LATENCY || mode == N2NMN_MODE_THROUGHPUT ||
                 mode == N2NMN_MODE_THROUGHPUT_KSPLIT || mode == N2NMN_MODE_THROUGHPUT_BF16X3,
             N2NMN_EINVAL, "ctx_set_mode: unknown mode");
  if (mode == N2NMN_MODE_THROUGHPUT_BF16X3) {
    // (lstm_tile3_kernel stages 32 k per step over at least 8 steps: the layer-0 contraction K = lstm_dim
    // must reach 256)
    N2_REQUIRE(ctx->enc_W0h_b3 && ctx->d.lstm_dim >= 256, N2NMN_EINVAL,
               "ctx_set_mode: bf16x3 needs lstm_dim % 128 == 0 and lstm_dim >= 256");
    n2nmn_ctx* r = ctx->parent ? ctx->parent : ctx;
    if (!r->b3_on) {
      // from now on every commit packs the split weights too; if weights are committed already, now
      // (a configuration call: it may wait for the device)
      r->b3_on = true;
      if (r->committed) {
        N2_HIP(hipSetDevice(r->device));
        N2_HIP(hipDeviceSynchronize());
        pack_b3(r, nullptr);
        N2_HIP(hipDeviceSynchronize());
      }
    }
  }
  ctx->mode = mode;
  return N2NMN_OK;
}

int n2nmn_ctx_dims(const n2nmn_ctx* ctx, n2nmn_dims* out) {
  N2_REQUIRE(ctx && out, N2NMN_EINVAL, "ctx_dims: null argument");
  *out = ctx->d;
  return N2NMN_OK;
}

int n2nmn_num_variables(const n2nmn_ctx* ctx) { return ctx ? (int)ctx->pub.size() : N2NMN_EINVAL; }

int n2nmn_variable_info(const n2nmn_ctx* ctx, int i, const char** name, int64_t shape[4],
                        int* ndim) {
  N2_REQUIRE(ctx && i >= 0 && i < (int)ctx->pub.size(), N2NMN_EINVAL, "variable_info: bad index");
  const Var& v = ctx->vars[ctx->pub[i]];
  if (name) *name = v.name.c_str();
  if (ndim) *ndim = (int)v.shape.size();
  if (shape)
    for (size_t k = 0; k < 4; ++k) shape[k] = k < v.shape.size() ? v.shape[k] : 1;
  return N2NMN_OK;
}

int n2nmn_set_weight(n2nmn_ctx* ctx, const char* name, const float* data, const int64_t* shape,
                     int ndim) {
  N2_REQUIRE(ctx && name && data && shape, N2NMN_EINVAL, "set_weight: null argument");
  N2_REQUIRE(!ctx->parent, N2NMN_EINVAL, "set_weight: forked contexts share the parent's weights");
  auto it = ctx->index.find(name);
  if (it == ctx->index.end()) {
    set_last_error(std::string("set_weight: unknown variable '") + name + "'");
    return N2NMN_EKEY;
  }
  Var& v = ctx->vars[it->second];
  bool ok = ndim == (int)v.shape.size();
  for (int k = 0; ok && k < ndim; ++k) ok = shape[k] == v.shape[k];
  if (!ok) {
    set_last_error(std::string("set_weight: shape mismatch for '") + name + "'");
    return N2NMN_EINVAL;
  }
  // the copy into the context-owned mirror is issued here on the NULL stream so the caller's
  // buffer is only referenced during this call
  train_infer_host_wait(ctx);          // (an optimiser step's second half may still be updating it)
  N2_HIP(hipMemcpy(v.mirror, data, sizeof(float) * v.numel, hipMemcpyDeviceToDevice));
  v.set = true;
  ctx->committed = false;
  return N2NMN_OK;
}

/* validity automaton of the layout vocabulary: Assembler.P [V,3], .W [3,V,4], .b [V,4]
 * (models_clevr/nmn3_assembler.py:50-119 -> nmn3_netgen_att.py:59-62), host int32 pointers */
// Does the installed automaton (nmn3_netgen_att.py:8-15: token s is valid in state x iff all_c (x . W[:, s, c] - b[s, c]
// >= 0); x += P[token]) force <eos> for ever behind <eos> and behind every answer operator?  Breadth-first over
// every state reachable from (0, 0, T) for T = 1 .. T_decoder (47 k states for the reference's automaton,
// nmn3_assembler.py:50-135; all-zero tables -- models_shapes -- fail at the first state).  The retirement of
// finished rows by dec_compact_kernel is exact only if it holds.
static bool automaton_forces_eos(const int32_t* P, const int32_t* W, const int32_t* b, const int32_t* token_op, int V,
                                 int T_decoder) {
  int eos = -1;
  for (int i = 0; i < V; ++i) if (token_op[i] < 0) { eos = i; break; }
  if (eos < 0 || V <= 0 || T_decoder <= 0) return false;
  auto valid = [&](const int x[3], int s) {
    for (int c = 0; c < 4; ++c) {
      long v = -(long)b[s * 4 + c];
      for (int k = 0; k < 3; ++k) v += (long)x[k] * W[(k * V + s) * 4 + c];
      if (v < 0) return false;
    }
    return true;
  };
  auto finishing = [&](int s) {
    const int op = token_op[s];
    return op < 0 || (op >= N2NMN_OP_EXIST && op <= N2NMN_OP_DESCRIBE);
  };
  const int lim = 4 * T_decoder + 8;                   // states outside [-lim, lim]^3 are not reachable in T steps
  auto key = [&](const int x[3]) {
    return ((long)(x[0] + lim) * (2 * lim + 1) + (x[1] + lim)) * (2 * lim + 1) + (x[2] + lim);
  };
  // (state, steps left, reached behind a finishing token): a pass decodes T_decoder tokens, so T steps from (0, 0, T)
  std::unordered_map<long, bool> seen;
  std::vector<std::array<int, 5>> todo;                // x0, x1, x2, steps left, finished
  for (int T = 1; T <= T_decoder; ++T) todo.push_back({0, 0, T, T, 0});
  size_t steps = 0;
  while (!todo.empty()) {
    const auto cur = todo.back();
    todo.pop_back();
    if (cur[3] == 0) continue;
    const int x[3] = {cur[0], cur[1], cur[2]};
    for (int k = 0; k < 3; ++k) if (x[k] < -lim || x[k] > lim) return false;
    const long kx = (key(x) * (T_decoder + 1) + cur[3]) * 2 + cur[4];
    if (seen.count(kx)) continue;
    seen[kx] = true;
    if (++steps > 4000000) return false;               // (not an automaton of this family: do not retire)
    for (int s = 0; s < V; ++s) {
      if (!valid(x, s)) continue;
      if (cur[4] && s != eos) return false;            // a finished row could emit something else
      todo.push_back({x[0] + P[s * 3 + 0], x[1] + P[s * 3 + 1], x[2] + P[s * 3 + 2], cur[3] - 1,
                      (cur[4] || finishing(s)) ? 1 : 0});
    }
  }
  return true;
}

static void refresh_retire_ok(n2nmn_ctx* r) {
  const int V = r->d.num_vocab_nmn;
  r->retire_ok = r->have_tables && r->have_token_ops && (int)r->P_host.size() == V * 3 &&
                 (int)r->token_op_host.size() == V &&
                 automaton_forces_eos(r->P_host.data(), r->W_host.data(), r->b_host.data(), r->token_op_host.data(), V,
                                      r->d.T_decoder);
}

int n2nmn_automaton_forces_eos(const int32_t* P_host, const int32_t* W_host, const int32_t* b_host,
                               const int32_t* token_op_host, int V, int T_decoder) {
  N2_REQUIRE(P_host && W_host && b_host && token_op_host, N2NMN_EINVAL, "automaton_forces_eos: null argument");
  return automaton_forces_eos(P_host, W_host, b_host, token_op_host, V, T_decoder) ? 1 : 0;
}

int n2nmn_set_validity_tables(n2nmn_ctx* ctx, const int32_t* P_host, const int32_t* W_host,
                              const int32_t* b_host) {
  N2_REQUIRE(ctx && P_host && W_host && b_host, N2NMN_EINVAL, "set_validity_tables: null argument");
  N2_REQUIRE(!ctx->parent, N2NMN_EINVAL, "set_validity_tables: set them on the root context");
  const int V = ctx->d.num_vocab_nmn;
  N2_HIP(hipMemcpy(ctx->P, P_host, sizeof(int32_t) * V * 3, hipMemcpyHostToDevice));
  N2_HIP(hipMemcpy(ctx->Wv, W_host, sizeof(int32_t) * 3 * V * 4, hipMemcpyHostToDevice));
  N2_HIP(hipMemcpy(ctx->bv, b_host, sizeof(int32_t) * V * 4, hipMemcpyHostToDevice));
  ctx->have_tables = true;
  ctx->P_host.assign(P_host, P_host + V * 3);
  ctx->W_host.assign(W_host, W_host + 3 * V * 4);
  ctx->b_host.assign(b_host, b_host + V * 4);
  refresh_retire_ok(ctx);
  return N2NMN_OK;
}

int n2nmn_commit_weights(n2nmn_ctx* c, n2nmn_stream stream) {
  return commit_weights_on(c, S(stream), S(stream));
}

// The commit in two halves.  `s`: every operand the ENCODER reads (its packed weights, biases, the
// input-projection table, encoder_h_transform) -- all functions of the encoder's variables only.
// `rest`: the decoder's and the module network's operands, then what only inference reads.  A training
// step hands its side stream as `rest` (n2nmn_adam_step): the next forward pass starts its encoder
// while those are still being packed; decoder / module network / walker entry points wait for the
// marker behind them (train_infer_wait).  rest == s: one stream, the order below.
extern "C++" int n2nmn::commit_weights_on(n2nmn_ctx* c, hipStream_t s, hipStream_t rest) {
  N2_REQUIRE(c, N2NMN_EINVAL, "commit_weights: null context");
  N2_REQUIRE(!c->parent, N2NMN_EINVAL, "commit_weights: commit through the root context");
  for (const Var& v : c->vars)
    if (!v.set) {
      set_last_error("commit_weights: variable never set: " + v.name);
      return N2NMN_ENOWEIGHT;
    }
  const n2nmn_dims& d = c->d;
  const int L = d.lstm_dim, E = d.embed_dim_txt, M = d.map_dim, V = d.num_vocab_nmn, Mp = c->Mp;
  auto m = [&](int id) { return c->vars[id].mirror; };
  // every operand re-pack of the commit is one job of ONE launch (the job table is built and
  // uploaded on the first commit; pointers and shapes never change afterwards)
  if (!c->packs.uploaded) {
    PackBatch& pb = c->packs;            // functions of the encoder's variables only
    PackBatch& pr = c->packs_rest;       // everything else
    auto has = [&](int id) { return c->vars[id].present; };
    // layer-0 input projections (rows [0,E) of the LSTM weights) -> PK, then the tables
    // input halves of the layer-0 weights with the gate columns in tile order: the x-tables come out
    // of their GEMMs in the order the step kernels read them (LstmJob::xtab)
    pb.pk_gates(m(V_ENC_W0), 4 * L, E, L, c->enc_W0x_p, c->KpE);
    pr.pk_gates(m(V_DEC_W0), 4 * L, E, L, c->dec_W0x_p, c->KpE);
    pb.vec_gates(m(V_ENC_B0), L, c->enc_b0_t);
    pr.vec_gates(m(V_DEC_B0), L, c->dec_b0_t);
    pb.vec_gates(m(V_ENC_B1), L, c->enc_b1_t);
    pr.vec_gates(m(V_DEC_B1), L, c->dec_b1_t);
    // recurrent parts -> gate-interleaved column tiles
    pb.tiles(m(V_ENC_W0), 4 * L, E, L, L / 4, L, c->enc_W0h_t);
    pb.tiles(m(V_ENC_W1), 4 * L, 0, 2 * L, L / 4, L, c->enc_W1_t);
    pr.tiles(m(V_DEC_W0), 4 * L, E, L, L / 4, L, c->dec_W0h_t);
    pr.tiles(m(V_DEC_W1), 4 * L, 0, 2 * L, L / 4, L, c->dec_W1_t);
    if (c->enc_W0h_64) {
      // operands of lstm_tile_kernel (passes of >= 128 rows): a batch of their own -- a training step
      // (64 rows) commits every iteration and never reads them, see below
      PackBatch& pi = c->packs_infer;
      pi.tiles64(m(V_ENC_W0), 4 * L, E, L, L, c->enc_W0h_64);
      pi.tiles64(m(V_ENC_W1), 4 * L, 0, 2 * L, L, c->enc_W1_64);
      pi.tiles64(m(V_DEC_W0), 4 * L, E, L, L, c->dec_W0h_64);
      pi.tiles64(m(V_DEC_W1), 4 * L, 0, 2 * L, L, c->dec_W1_64);
      N2_HIP(hipMemcpy(pi.dev, pi.jobs.data(), sizeof(PackJob) * pi.jobs.size(), hipMemcpyHostToDevice));
    }
    pb.pk(m(V_EHT_W), L, L, L, c->eht_W_p, c->KpL, L);
    pr.tiles(m(V_ATT_W), L, 0, L, L / 16, 0, c->att_W_t);
    pr.pk(m(V_ATT_W), L, L, L, c->att_W_p, c->KpL, L);
    pr.pk(m(V_FIND_IMG_W), M, d.D, M, c->find_img_p, c->KpD, Mp);
    pr.pk(m(V_FSP_IMG_W), M, d.D, M, c->fsp_img_p, c->KpD, Mp);
    pr.pad(m(V_DEC_EMB), V, E, c->dec_emb_cat, E);
    pr.pad(m(V_DEC_GO), 1, E, c->dec_emb_cat + (size_t)V * E, E);
    // zero-padded copies of the [M] vectors / [.., M] matrices read with float4 lanes
    const int wes[3] = {V_FIND_E_W, V_FSP_E_W, V_TR_E_W};
    for (int i = 0; i < 3; ++i)
      if (has(wes[i])) pr.pad(m(wes[i]), 1, M, c->we_pad[i], Mp);
    if (has(V_TR_MAPS_W)) {          // k-major [KK + 1 (+pad)][Mq] operand of the walker's MFMA Transform
      const int KK = d.kernel_size * d.kernel_size, Mq = round_up(M, 16);
      pr.pad(m(V_TR_MAPS_W), KK, M, c->tr_At, Mq);
      pr.pad(m(V_TR_MAPS_B), 1, M, c->tr_At + (size_t)KK * Mq, Mq);
    }
    const int txs[5] = {V_FIND_TXT_W, V_FSP_TXT_W, V_TR_TXT_W, V_SP_TXT_W, V_DE_TXT_W};
    for (int i = 0; i < 5; ++i) {
      if (!has(txs[i])) continue;
      pr.pad(m(txs[i]), E, M, c->wtxt_pad[i], Mp);
      pr.pad(m(txs[i] + 1), 1, M, c->btxt_pad[i], Mp);
      if (c->wtxt_pk[i]) pr.pk(m(txs[i]), M, E, M, c->wtxt_pk[i], c->KpE, Mp);
    }
    const int ats[4] = {V_FSP_ATT_W, V_SP_ATT0_W, V_SP_ATT1_W, V_DE_ATT_W};
    for (int i = 0; i < 4; ++i)
      if (has(ats[i])) pr.pad(m(ats[i]), d.D, M, c->watt_pad[i], Mp);
    const int bas[4] = {V_FSP_ATT_B, V_SP_ATT0_B, V_SP_ATT1_B, V_DE_ATT_B};
    for (int i = 0; i < 4; ++i)
      if (has(bas[i])) pr.pad(m(bas[i]), 1, M, c->batt_pad[i], Mp);
    if (c->big_heads) {
      const int Kp = round_up(M, 32), Np = round_up(d.num_choices, 64);
      pr.pk(m(V_DE_E_W), d.num_choices, M, d.num_choices, c->wans_de_p, Kp, Np);
      if (c->wans_sp_p) pr.pk(m(V_SP_E_W), d.num_choices, M, d.num_choices, c->wans_sp_p, Kp, Np);
    }
    if (c->qpn_W1_p) {
      pr.pk(m(V_QPN_W1), d.qpn_hidden, 2 * L, d.qpn_hidden, c->qpn_W1_p, round_up(2 * L, 32),
            round_up(d.qpn_hidden, 64));
      pr.pk(m(V_QPN_W2), d.num_choices, d.qpn_hidden, d.num_choices, c->qpn_W2_p,
            round_up(d.qpn_hidden, 32), round_up(d.num_choices, 64));
    }
    N2_REQUIRE((int)pb.jobs.size() <= kMaxPackJobs && (int)pr.jobs.size() <= kMaxPackJobs,
               N2NMN_ECAPACITY, "commit_weights: pack job table");
    N2_HIP(hipMemcpy(pb.dev, pb.jobs.data(), sizeof(PackJob) * pb.jobs.size(), hipMemcpyHostToDevice));
    N2_HIP(hipMemcpy(pr.dev, pr.jobs.data(), sizeof(PackJob) * pr.jobs.size(), hipMemcpyHostToDevice));
    pb.uploaded = true;
  }
  // ---- what the encoder reads: on `s` ------------------------------------------------------------
  launch_pack_jobs(c->packs.dev, (int)c->packs.jobs.size(), c->packs.blocks, s);
  // xtab[v] = emb[v] . W_x + b : the whole input half of the layer-0 gate pre-activations
  GemmArgs g{};
  g.A = m(V_ENC_EMB); g.lda = E; g.M = d.num_vocab_txt; g.K = E; g.group_size = 1;
  g.Bp = c->enc_W0x_p; g.Np = 4 * L; g.Kp = c->KpE; g.bias = c->enc_b0_t; g.N = 4 * L;
  g.C = c->enc_xtab; g.ldc = 4 * L; g.n_store = 4 * L;
  if (!c->big_vocab) launch_gemm_pk(g, s);
  // ---- the decoder's and the module network's operands: on `rest` ---------------------------------
  {
    const int KK = d.kernel_size * d.kernel_size, KD = (KK + 1 + 3) & ~3, Mq = round_up(d.map_dim, 16);
    if (KD > KK + 1)
      N2_HIP(hipMemsetAsync(c->tr_At + (size_t)(KK + 1) * Mq, 0, sizeof(float) * (size_t)(KD - KK - 1) * Mq, rest));
  }
  launch_pack_jobs(c->packs_rest.dev, (int)c->packs_rest.jobs.size(), c->packs_rest.blocks, rest);
  g.A = c->dec_emb_cat; g.M = V + 1; g.Bp = c->dec_W0x_p; g.bias = c->dec_b0_t; g.C = c->dec_xtab;
  launch_gemm_pk(g, rest);
  // ---- what only inference reads -- the 64-column recurrent tiles of passes >= 128 rows and the
  // walker's text-map tables (embedding_mat . W_txt: they read the ENCODER's embedding, which `s`
  // has just updated, hence the fork from `s` here) -- behind the rest on the training step's side
  // stream when the context trains; encoder (tile kernel only) / decoder / module network / walker
  // entry points wait for the marker behind it (train_infer_wait).
  hipStream_t si = train_infer_fork(c, s);
  if (rest != s && si != rest) si = rest;      // (training without a side stream: everything on `s`)
  if (!c->packs_infer.jobs.empty())
    launch_pack_jobs(c->packs_infer.dev, (int)c->packs_infer.jobs.size(), c->packs_infer.blocks, si);
  if (c->b3_on && c->enc_W0h_b3) pack_b3(c, si);
  for (int i = 0; i < 5; ++i) {        // ew[ws] = embedding_mat . W_txt[ws]  (walker text maps)
    if (!c->ew[i]) continue;
    GemmArgs t{};
    t.A = m(V_ENC_EMB); t.lda = E; t.M = d.num_vocab_txt; t.K = E; t.group_size = 1;
    t.Bp = c->wtxt_pk[i]; t.Np = c->Mp; t.Kp = c->KpE; t.bias = nullptr; t.N = d.map_dim;
    t.C = c->ew[i]; t.ldc = c->Mp; t.n_store = c->Mp;
    launch_gemm_pk(t, si);
  }
  train_infer_done(c, si);
  c->committed = true;
  c->commit_epoch++;
  c->enc_T = 0;
  return check_launch("commit_weights");
}

int n2nmn_encoder_forward(n2nmn_ctx* ctx, const n2nmn_seq2seq_io* io, n2nmn_stream stream) {
  N2_REQUIRE(ctx, N2NMN_EINVAL, "encoder_forward: null context");
  return encoder_impl(ctx, io, S(stream), nullptr);
}

int n2nmn_decoder_forward(n2nmn_ctx* ctx, const n2nmn_seq2seq_io* io, n2nmn_stream stream) {
  N2_REQUIRE(ctx, N2NMN_EINVAL, "decoder_forward: null context");
  return decoder_impl(ctx, io, S(stream), nullptr, 0);
}

int n2nmn_seq2seq_forward(n2nmn_ctx* ctx, const n2nmn_seq2seq_io* io, n2nmn_stream stream) {
  N2_REQUIRE(ctx, N2NMN_EINVAL, "seq2seq_forward: null context");
  // the encoder_h_transform GEMM rides in the decoder's GEMM launch unless the caller wants a copy
  // of it (the copy is made by the encoder half)
  GemmArgs eht{};
  const bool defer = io && !io->encoder_h_transformed;
  const int rc = encoder_impl(ctx, io, S(stream), defer ? &eht : nullptr);
  if (rc != N2NMN_OK) return rc;
  return decoder_impl(ctx, io, S(stream), defer ? &eht : nullptr, defer ? 1 : 0);
}

int n2nmn_execute_program(n2nmn_ctx* ctx, n2nmn_program* p, const float* image_feat,
                          const float* word_vecs, int N_full, float* scores,
                          n2nmn_stream stream) {
  N2_REQUIRE(ctx && p && image_feat && word_vecs && scores, N2NMN_EINVAL,
             "execute_program: null argument");
  return run_program(ctx, p->prog, image_feat, word_vecs, N_full, scores, nullptr, nullptr,
                     nullptr, 0, 0, S(stream));
}

int n2nmn_module_forward(n2nmn_ctx* ctx, int op, int Nb, const float* input_0,
                         const float* input_1, const int32_t* time_idx_host,
                         const int32_t* batch_idx_host, const float* image_feat,
                         const float* word_vecs, int N_full, float* out, n2nmn_stream stream) {
  N2_REQUIRE(ctx, N2NMN_EINVAL, "module_forward: null context");
  const int k = op_arity(op);
  N2_REQUIRE(k >= 0 && op != OP_INPUT, N2NMN_EKEY, "module_forward: unknown module operator");
  N2_REQUIRE(Nb >= 0, N2NMN_EINVAL, "module_forward: negative batch");
  if (Nb == 0) return N2NMN_OK;          // Fold's zero-size batches need no work at all
  N2_REQUIRE(out && time_idx_host && batch_idx_host, N2NMN_EINVAL,
             "module_forward: null argument");
  N2_REQUIRE((k < 1 || input_0) && (k < 2 || input_1), N2NMN_EINVAL,
             "module_forward: missing attention input for this operator");
  N2_REQUIRE(image_feat && word_vecs, N2NMN_EINVAL, "module_forward: null feature / word_vecs");
  std::vector<n2nmn_node> nodes;
  nodes.reserve((size_t)Nb * (k + 1));
  for (int j = 0; j < k; ++j)
    for (int i = 0; i < Nb; ++i) {
      n2nmn_node nd{};
      nd.op = OP_INPUT; nd.time_idx = j; nd.batch_idx = i; nd.in0 = nd.in1 = -1; nd.out_row = -1;
      nodes.push_back(nd);
    }
  const bool ans = op_is_answer(op);
  for (int i = 0; i < Nb; ++i) {
    n2nmn_node nd{};
    nd.op = op; nd.time_idx = time_idx_host[i]; nd.batch_idx = batch_idx_host[i];
    nd.in0 = k >= 1 ? i : -1; nd.in1 = k >= 2 ? Nb + i : -1;
    nd.out_row = ans ? i : -1;
    nodes.push_back(nd);
  }
  Program& p = ctx->scratch_prog->prog;
  int rc = from_nodes(p, nodes.data(), (int)nodes.size(), ans ? Nb : 0);
  if (rc != N2NMN_OK) { set_last_error(p.error); return rc; }
  return run_program(ctx, p, image_feat, word_vecs, N_full, ans ? out : nullptr, input_0, input_1,
                     ans ? nullptr : out, k * Nb, ans ? 0 : Nb, S(stream));
}


int n2nmn_set_token_ops(n2nmn_ctx* ctx, const int32_t* token_op_host, int V) {
  N2_REQUIRE(ctx && token_op_host, N2NMN_EINVAL, "set_token_ops: null argument");
  N2_REQUIRE(!ctx->parent, N2NMN_EINVAL, "set_token_ops: set them on the root context");
  N2_REQUIRE(V == ctx->d.num_vocab_nmn, N2NMN_EINVAL, "set_token_ops: V != num_vocab_nmn");
  for (int i = 0; i < V; ++i)
    N2_REQUIRE(token_op_host[i] < 0 || (op_arity(token_op_host[i]) >= 0 && token_op_host[i] != OP_INPUT),
               N2NMN_EKEY, "set_token_ops: unknown op code");
  N2_HIP(hipMemcpy(ctx->token_op, token_op_host, sizeof(int32_t) * V, hipMemcpyHostToDevice));
  ctx->have_token_ops = true;
  ctx->eos_token = -1;
  for (int i = 0; i < V; ++i) if (token_op_host[i] < 0) { ctx->eos_token = i; break; }
  ctx->token_op_host.assign(token_op_host, token_op_host + V);
  refresh_retire_ok(ctx);
  return N2NMN_OK;
}

int n2nmn_walk_set_defer_pool(n2nmn_ctx* c, int mode) {
  N2_REQUIRE(c && mode >= -1 && mode <= 1, N2NMN_EINVAL, "walk_set_defer_pool: mode is -1, 0 or 1");
  c->walk_defer_pool = mode;
  return N2NMN_OK;
}

int n2nmn_walk_set_front_end(n2nmn_ctx* c, int mode) {
  N2_REQUIRE(c && mode >= -1 && mode <= 1, N2NMN_EINVAL, "walk_set_front_end: mode is -1, 0 or 1");
  c->walk_pre_find = mode;
  return N2NMN_OK;
}

int n2nmn_walk_set_staged(n2nmn_ctx* c, int mode) {
  N2_REQUIRE(c && mode >= -1 && mode <= 1, N2NMN_EINVAL, "walk_set_staged: mode is -1, 0 or 1");
  c->walk_staged = mode;
  return N2NMN_OK;
}

int n2nmn_walk_set_levels(n2nmn_ctx* c, int levels) {
  N2_REQUIRE(c && levels >= -1 && levels <= WALK_HLEVELS, N2NMN_EINVAL,
             "walk_set_levels: 0 (every reachable level), -1 (adaptive) or 1 .. 24");
  c->walk_levels = levels;
  return N2NMN_OK;
}

int n2nmn_walk_set_nesting_bound(n2nmn_ctx* c, int bound) {
  N2_REQUIRE(c && bound >= -1 && bound <= WALK_MAX_T, N2NMN_EINVAL, "walk_set_nesting_bound: -1 (none) or 0 .. 32");
  c->walk_nesting_bound = bound;
  return N2NMN_OK;
}

int n2nmn_walk_supported(const n2nmn_ctx* c) {
  if (!c) return 0;
  const n2nmn_dims& d = c->d;
  if (d.variant != N2NMN_VARIANT_CLEVR || c->big_heads) return 0;
  return walk_supported(d.H, d.W, d.D, d.map_dim, c->Mp, c->HWp, d.embed_dim_txt, d.num_choices,
                        d.T_decoder, d.kernel_size, d.T_encoder);
}

int n2nmn_conv_image(n2nmn_ctx* c, const float* image_feat, int N, int which,
                     const int32_t* tokens, int T_dec, n2nmn_stream stream) {
  N2_REQUIRE(c && image_feat, N2NMN_EINVAL, "conv_image: null argument");
  N2_REQUIRE(is_committed(c), N2NMN_ENOWEIGHT, "conv_image: weights not committed");
  const n2nmn_dims& d = c->d;
  N2_REQUIRE(N >= 1 && N <= d.N, N2NMN_ECAPACITY, "conv_image: N > capacity");
  N2_REQUIRE(!tokens || (root(c)->have_token_ops && T_dec >= 1), N2NMN_EINVAL,
             "conv_image: gating by tokens needs n2nmn_set_token_ops and T_dec");
  const int HW = d.H * d.W;
  hipStream_t s = S(stream);
  train_infer_wait(root(c), s);
  const double dHW = HW, dD = d.D, dM = d.map_dim, dMp = c->Mp;
  GemmArgs ga[2];
  conv_image_problems(c, image_feat, N, tokens, T_dec, ga);
  // algorithmic work of the gated problem is not known on the host: the profile line counts the
  // ungated FindModule GEMM in full and the gated one as the reference mix's 1 image in 10
  const double one_fl = 2.0 * N * dHW * dD * dM, one_by = 4.0 * (N * dHW * (dD + dMp)) + 4.0 * dD * dM;
  const bool both = (which & N2NMN_CONV_FIND) && (which & N2NMN_CONV_FSP);
  if (both) {
    const double frac = tokens ? 1.1 : 2.0;
    ProfScope ps(c, F_CONV_IMAGE, frac * one_fl, frac * one_by, s);
    launch_gemm_pkn(ga, 2, s);
  } else {
    for (int fsp = 0; fsp < 2; ++fsp) {
      if (!(which & (fsp ? N2NMN_CONV_FSP : N2NMN_CONV_FIND))) continue;
      const double frac = (fsp && tokens) ? 0.1 : 1.0;
      ProfScope ps(c, F_CONV_IMAGE, frac * one_fl, frac * one_by, s);
      launch_gemm_pk(ga[fsp], s);
    }
  }
  return check_launch("conv_image");
}

// The staged walker's launches between walk_find and the fall-back walker: per nesting level walk_heavy_kernel
// (stage A of the FindSameProperty nodes -- pooling + fc_att shares -- and the Transform halves), then
// walk_fspepi_kernel (stage B: the FindSameProperty map epilogues); the light rest of every question at the
// end.  (Built and measured slower: one launch with device-side dependencies between work items,
// tools/rejected/walk_stage_single_launch.hip.txt; level 0 at the tail of the Find launch,
// tools/rejected/walk_front_kernel.diff.txt.)
static void walk_staged_launches(const ModuleWeights& w, WalkArgs& a, hipStream_t s) {
  for (int lv = 0; lv < a.hlevels; ++lv) { a.hlevel = lv; launch_walk_heavy(w, a, s); launch_walk_fspepi(w, a, s); }
  a.hlevel = 0;
  launch_walk_light(w, a, s);
}

// n2nmn_walk_set_conv_inline: the conv_image maps of the walker's batches, computed inside the walker call
// right before walk_find reads them -- FindSameProperty's (gated by the layouts) first, Find's 154 KB per image
// LAST, so that what walk_find streams is the most recently written data of the pass and comes back from the
// Infinity Cache instead of HBM (measured at 1024 questions: walk_find 36.5 -> 29.5 us, the attention-module
// path 103 -> 96 us, pass time unchanged; profiles/r05_notes.md section 1).
static void walk_conv_inline(n2nmn_ctx* c, const n2nmn_walk_batch* batches, int K, int N, int T_dec,
                             hipStream_t s) {
  const n2nmn_dims& d = c->d;
  for (int k = 0; k < K; ++k) {
    n2nmn_ctx* owner = const_cast<n2nmn_ctx*>(batches[k].ctx ? batches[k].ctx : c);
    GemmArgs ga[2];
    conv_image_problems(owner, batches[k].image_feat, N, batches[k].tokens, T_dec, ga);
    std::swap(ga[0], ga[1]);
    const double HW = d.H * d.W, frac = 1.1;       // gated share: see n2nmn_conv_image
    ProfScope ps(c, F_CONV_IMAGE, frac * 2.0 * N * HW * d.D * d.map_dim,
                 frac * 4.0 * N * HW * (d.D + c->Mp) + 4.0 * d.D * d.map_dim, s);
    launch_gemm_pkn(ga, 2, s);
  }
}

int n2nmn_walk_set_conv_inline(n2nmn_ctx* c, int on) {
  N2_REQUIRE(c, N2NMN_EINVAL, "walk_set_conv_inline: null context");
  c->walk_conv_inline = on != 0;
  return N2NMN_OK;
}

int n2nmn_walk_layouts(n2nmn_ctx* c, const n2nmn_walk_batch* batches, int K, int T_dec, int T_enc,
                       int N, n2nmn_stream stream) {
  N2_REQUIRE(c && batches, N2NMN_EINVAL, "walk_layouts: null argument");
  // the two one-call promises (n2nmn_walk_set_nesting_bound / _conv_inline) are consumed HERE, whichever way
  // this call ends: a rejected argument further down must not leave them standing for some later pass
  const int nesting_bound = c->walk_nesting_bound;
  const bool conv_inline = c->walk_conv_inline;
  c->walk_nesting_bound = -1;
  c->walk_conv_inline = false;
  N2_REQUIRE(is_committed(c), N2NMN_ENOWEIGHT, "walk_layouts: weights not committed");
  N2_REQUIRE(root(c)->have_token_ops, N2NMN_EINVAL, "walk_layouts: call n2nmn_set_token_ops first");
  N2_REQUIRE(n2nmn_walk_supported(c), N2NMN_EINVAL,
             "walk_layouts: dimensions outside the walker's tiling (use n2nmn_execute_program)");
  const n2nmn_dims& d = c->d;
  N2_REQUIRE(K >= 1 && K <= WALK_MAX_BATCHES, N2NMN_ECAPACITY, "walk_layouts: 1 <= K <= 16");
  N2_REQUIRE(N >= 1 && N <= d.N, N2NMN_ECAPACITY, "walk_layouts: N > capacity");
  N2_REQUIRE(T_dec >= 1 && T_dec <= d.T_decoder && T_dec <= WALK_MAX_T, N2NMN_ECAPACITY,
             "walk_layouts: T_dec > capacity");
  WalkArgs a{};
  train_infer_wait(root(c), S(stream));
  const bool use_table = batches[0].atts != nullptr;
  N2_REQUIRE(!use_table || root(c)->ew[0], N2NMN_EINVAL,
             "walk_layouts: attention-table text maps need num_vocab_txt <= 4096");
  N2_REQUIRE(!use_table || (T_enc >= 1 && T_enc <= d.T_encoder), N2NMN_ECAPACITY,
             "walk_layouts: T_enc out of range");
  for (int k = 0; k < K; ++k) {
    const n2nmn_walk_batch& b = batches[k];
    N2_REQUIRE(b.tokens && b.image_feat && (b.word_vecs || b.atts) && b.scores, N2NMN_EINVAL,
               "walk_layouts: null buffer in a batch");
    const n2nmn_ctx* owner = b.ctx ? b.ctx : c;
    N2_REQUIRE(root(owner) == root(c), N2NMN_EINVAL,
               "walk_layouts: a batch's context does not share this context's weights");
    N2_REQUIRE(!b.atts == !use_table, N2NMN_EINVAL,
               "walk_layouts: either every batch gives atts / input_seq / seq_length or none does");
    N2_REQUIRE(!use_table || (b.input_seq && b.seq_length), N2NMN_EINVAL,
               "walk_layouts: atts needs input_seq and seq_length");
    a.b[k].atts = b.atts; a.b[k].seq = b.input_seq; a.b[k].seq_len = b.seq_length;
    a.b[k].tokens = b.tokens; a.b[k].feat = b.image_feat; a.b[k].word_vecs = b.word_vecs;
    a.b[k].scores = b.scores; a.b[k].validity = b.validity;
    a.b[k].mfind = owner->mfind; a.b[k].mfsp = owner->mfsp; a.b[k].tmap = owner->wtmap;
    a.b[k].watt = owner->watt;
    a.b[k].pjob = owner->wpjob; a.b[k].pw = owner->wpw; a.b[k].ptm = owner->wptm;
    a.b[k].pooled = owner->wpooled; a.b[k].pfc = owner->wpfc;
    a.b[k].prog = owner->wprog; a.b[k].fpart = owner->wfpart;
  }
  a.K = K; a.N = N; a.T = T_dec; a.V = d.num_vocab_nmn; a.token_op = root(c)->token_op;
  a.H = d.H; a.W = d.W; a.D = d.D; a.M = d.map_dim; a.Mp = c->Mp; a.HWp = c->HWp;
  a.E = d.embed_dim_txt; a.C = d.num_choices; a.ksize = d.kernel_size;
  // (profile mode: the walker kernels count their nodes into device words -- the source of the families' algorithmic
  // bytes, and a few hundred contended atomics that triple the tree-dependent launches' time; "profile_walk_stats" = 0
  // keeps the event pairs and drops the counting, for the passes whose DURATIONS are read)
  a.stats = c->prof_on && knob_int(c, "profile_walk_stats", 1) != 0 ? c->walk_stats : nullptr;
  a.timeline = c->walk_timeline;
  a.T_enc = use_table ? T_enc : 0;
  for (int i = 0; i < 5; ++i) a.ew[i] = root(c)->ew[i];
  a.V_txt = d.num_vocab_txt;
  ModuleWeights w = module_weights(c);
  hipStream_t s = S(stream);
  if (!use_table) {
    // text maps of all K batches: <= ceil(N/8) * 5 * T_dec * K workgroups, most exit after the scan
    const double dE = d.embed_dim_txt, dM = d.map_dim;
    ProfScope ps(c, F_TEXTMAP, 0.0, 4.0 * 5 * dE * dM, s);
    launch_walk_textmap(w, a, s);
  }
  // throughput mode (>= 128 questions in the launch, or forced by the mode switch): the pooling
  // answer operators leave the walker and run as chip-wide launches of their own
  const int dp_env = c->walk_defer_pool;
  a.defer_pool = (dp_env < 0 ? K * N >= 128 : dp_env > 0) && walk_pool_supported(d.H, d.W, d.D);
  // passes of many questions: the text maps and the Find / Filter epilogues run chip-wide before the
  // walker, which then finds both in HBM (600 bytes per node) and keeps only the tree-dependent work
  const int pf_env = c->walk_pre_find;
  const bool pre = use_table && (pf_env < 0 ? K * N >= 128 : pf_env > 0);
  // staged walker: with both of the above the tree-dependent work leaves the one-workgroup-per-question
  // chain too (kernels_walk.hip: walk_heavy_kernel / walk_fspepi_kernel / walk_light_kernel); n2nmn_walk_set_staged(ctx, 0)
  // keeps the round-3 walker
  const bool staged = pre && a.defer_pool && c->walk_staged != 0 &&
                      K * N < (1 << 22) && T_dec <= 255;
  if (a.defer_pool) {
    // the per-pass counters (two sets, used alternately: walk_fcatt_kernel clears the other one) and the
    // pooled-root job lists
    a.cnt = c->wcnt + WALK_CNT * c->walk_parity; a.cnt_next = c->wcnt + WALK_CNT * (c->walk_parity ^ 1);
    c->walk_parity ^= 1;
    a.plist = c->wplist; a.pcap = WALK_MAX_BATCHES * d.N;
  }
  if (staged) {
    a.staged = 1;
    a.hjobs = c->whjobs; a.fblist = c->wfblist;
    for (int i = 0; i <= WALK_HLEVELS; ++i) a.hoff[i] = c->whoff[i];
    // How many nesting levels of Transform / FindSameProperty get a launch of their own.  The route of a
    // question (a level launch or the one-workgroup fall-back) decides the summation order of its answer head,
    // i.e. the last bits of its logits, so by DEFAULT it must depend on nothing but the pass itself
    // (models_clevr/nmn3_model.py:134-159: Fold's result does not depend on batching): every
    // level a layout of T_dec tokens can reach gets its launch (T_dec - 1; the level kernels are persistent
    // grids that leave at once when their list is empty) and the fall-back is not needed.
    // n2nmn_walk_set_levels(ctx, -1) opts into the adaptive form instead: as deep as the last two passes went
    // (a host-mapped word the kernels write and the host reads without waiting) -- fewer empty launches,
    // history-dependent last bits for nested layouts.
    int seen = 0;
    if (c->walk_hint_host) {
      seen = *reinterpret_cast<volatile int32_t*>(c->walk_hint_host);
      a.hint = c->walk_hint_dev;
    }
    const int full = std::max(T_dec - 1, 1);
    const int want = c->walk_levels > 0 ? c->walk_levels
                     : c->walk_levels < 0 ? std::max(seen, c->walk_hint_prev) : full;
    c->walk_hint_prev = seen;
    a.hlevels = std::min(std::max(want, 1), WALK_HLEVELS);
    if (c->walk_levels == 0 && full <= WALK_HLEVELS) a.no_fallback = 1;     // (no layout can nest deeper)
    // the caller knows this pass's layouts (host copies of ground-truth layouts): exactly as many levels
    // as they nest, and no launch of the fall-back walker (an empty one costs 4 - 5 us of the pass)
    if (nesting_bound >= 0 && nesting_bound <= WALK_HLEVELS) {
      a.hlevels = std::max(nesting_bound, 1);
      a.no_fallback = 1;
    }
  }
  if (conv_inline && !pre) walk_conv_inline(c, batches, K, N, T_dec, s);
  if (pre) {
    a.pre_find = 1;
    {
      ProfScope ps(c, F_WALK_TMAP, 0.0, 0.0, s);
      launch_walk_tmap(w, a, s);
    }
    if (conv_inline) walk_conv_inline(c, batches, K, N, T_dec, s);   // behind the text maps, in front of their reader
    {
      ProfScope ps(c, F_WALK_FIND, 0.0, 0.0, s);   // bytes from the device counters (walk stats [8])
      launch_walk_find(w, a, s);
    }
    c->last_walk_T_enc = a.T_enc;
    a.T_enc = 0;                 // the walker reads tmap instead of building the maps itself
  }
  {
    ProfScope ps(c, F_WALK, 0.0, 0.0, s);     // work filled in from the device counters
    if (a.staged) walk_staged_launches(w, a, s);
    if (!a.no_fallback) launch_walk(w, a, s); // staged: only the questions listed as nested too deep
  }
  c->last_walk = a;
  c->have_last_walk = true;
  if (a.defer_pool) {
    {
      ProfScope ps(c, F_POOL, 0.0, 0.0, s);   // jobs are counted on the device (walk stats [6], [7])
      launch_walk_pool(w, a, s);
    }
    {
      ProfScope ps(c, F_HEADS, 0.0, 0.0, s);
      launch_walk_heads(w, a, s);
    }
  }
  return check_launch("walk_layouts");
}

__global__ void n2nmn_empty_kernel() {}

int n2nmn_debug_event_overhead(n2nmn_ctx* c, int iters, double* us_pair, n2nmn_stream stream) {
  N2_REQUIRE(c && us_pair && iters >= 1 && iters <= 4096, N2NMN_EINVAL, "debug_event_overhead: bad argument");
  hipStream_t s = S(stream);
  std::vector<hipEvent_t> ev(2 * (size_t)iters);
  for (auto& e : ev) N2_HIP(hipEventCreate(&e));
  for (int i = 0; i < iters; ++i) {
    N2_HIP(hipEventRecord(ev[2 * i], s));
    hipLaunchKernelGGL(n2nmn_empty_kernel, dim3(1), dim3(64), 0, s);
    N2_HIP(hipEventRecord(ev[2 * i + 1], s));
  }
  N2_HIP(hipStreamSynchronize(s));
  double tot = 0;
  for (int i = 0; i < iters; ++i) {
    float ms = 0.f;
    N2_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
    tot += ms;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  *us_pair = 1e3 * tot / iters;
  return N2NMN_OK;
}

int n2nmn_debug_walk_replay(n2nmn_ctx* c, int which, int iters, double* us_avg, n2nmn_stream stream) {
  N2_REQUIRE(c && us_avg && iters >= 1 && iters <= 10000, N2NMN_EINVAL, "debug_walk_replay: bad argument");
  N2_REQUIRE(c->have_last_walk, N2NMN_EINVAL, "debug_walk_replay: no walker launch to replay");
  N2_REQUIRE(((which & 0xf) != 1 && (which & 0xf) != 2) || c->last_walk.defer_pool, N2NMN_EINVAL,
             "debug_walk_replay: the last launch did not defer its pooling jobs");
  N2_REQUIRE((which & 0xf) < 3 || c->last_walk.pre_find, N2NMN_EINVAL,
             "debug_walk_replay: the last launch ran its front end inside the walker");
  hipStream_t s = S(stream);
  ModuleWeights w = module_weights(c);
  WalkArgs a = c->last_walk;
  a.stats = nullptr; a.timeline = nullptr;
  const bool pairs = (which & 0x10) != 0;       // one event pair PER launch (calibrates the pair cost)
  which &= 0xf;
  auto one = [&]() {
    if (which == 0) {
      WalkArgs t = a;
      t.plist = nullptr;                 // (a replay must not append the pooled roots to the lists again)
      if (t.staged) {
        walk_staged_launches(w, t, s);
      }
      if (!t.no_fallback) launch_walk(w, t, s);
    }
    else if (which == 1) launch_walk_pool(w, a, s);
    else if (which == 2) launch_walk_heads(w, a, s);
    else if (which == 3) launch_walk_find(w, a, s);
    else if (which >= 5 && which <= 8 && a.staged) {      // the staged walker's launches one by one (level 0)
      WalkArgs t = a;
      t.plist = nullptr; t.hlevel = 0;
      if (which == 5) launch_walk_heavy(w, t, s);
      else if (which == 6) launch_walk_fspepi(w, t, s);
      else if (which == 7) launch_walk_light(w, t, s);
      else if (!t.no_fallback) launch_walk(w, t, s);
    }
    else { WalkArgs t = a; t.T_enc = c->last_walk_T_enc; t.staged = 0; launch_walk_tmap(w, t, s); }   // (no list appends)
  };
  for (int i = 0; i < 3; ++i) one();
  std::vector<hipEvent_t> ev(pairs ? 2 * (size_t)iters : 2);
  for (auto& e : ev) N2_HIP(hipEventCreate(&e));
  if (!pairs) N2_HIP(hipEventRecord(ev[0], s));
  for (int i = 0; i < iters; ++i) {
    if (pairs) N2_HIP(hipEventRecord(ev[2 * i], s));
    one();
    if (pairs) N2_HIP(hipEventRecord(ev[2 * i + 1], s));
  }
  if (!pairs) N2_HIP(hipEventRecord(ev[1], s));
  N2_HIP(hipStreamSynchronize(s));
  double tot = 0;
  for (size_t i = 0; i + 1 < ev.size(); i += 2) {
    float ms = 0.f;
    N2_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
    tot += ms;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  *us_avg = 1e3 * tot / iters;
  return check_launch("debug_walk_replay");
}

int n2nmn_debug_walk_timeline(n2nmn_ctx* c, long long* timeline_dev) {
  N2_REQUIRE(c, N2NMN_EINVAL, "debug_walk_timeline: null context");
  c->walk_timeline = timeline_dev;
  return N2NMN_OK;
}

int n2nmn_execute_tokens(n2nmn_ctx* c, const int32_t* tokens, int T_dec, int N,
                         const float* image_feat, const float* word_vecs, float* scores,
                         int32_t* validity, n2nmn_stream stream) {
  N2_REQUIRE(c && tokens && image_feat && word_vecs && scores, N2NMN_EINVAL,
             "execute_tokens: null argument");
  // dimensions outside the walker's tiling (models_vqa): the level path, scheduled on the device
  if (!n2nmn_walk_supported(c) || c->tokens_via_levels) {
    c->walk_nesting_bound = -1;   // (the level path has no use for the walker's one-call promises: drop them)
    c->walk_conv_inline = false;
    int rc = n2nmn_conv_image(c, image_feat, N, N2NMN_CONV_FIND | N2NMN_CONV_FSP, tokens, T_dec, stream);
    if (rc != N2NMN_OK) return rc;
    return run_tokens_levels(c, tokens, T_dec, N, image_feat, word_vecs, scores, validity, S(stream));
  }
  c->walk_conv_inline = true;     // the walker call computes the maps itself, right before it reads them
  n2nmn_walk_batch b{};
  b.ctx = c; b.tokens = tokens; b.image_feat = image_feat; b.word_vecs = word_vecs;
  b.scores = scores; b.validity = validity;
  return n2nmn_walk_layouts(c, &b, 1, T_dec, 0, N, stream);
}

int n2nmn_set_tokens_via_levels(n2nmn_ctx* c, int on) {
  N2_REQUIRE(c, N2NMN_EINVAL, "set_tokens_via_levels: null context");
  c->tokens_via_levels = on != 0;
  return N2NMN_OK;
}

int n2nmn_add_coords(n2nmn_ctx* c, const float* feat, int N, int D0, float* out,
                     n2nmn_stream stream) {
  N2_REQUIRE(c && feat && out, N2NMN_EINVAL, "add_coords: null argument");
  N2_REQUIRE(N >= 1 && D0 >= 1 && D0 + 2 <= c->d.D, N2NMN_EINVAL,
             "add_coords: the context's feature depth must be >= D0 + 2");
  launch_add_coords(feat, N, c->d.H, c->d.W, D0, c->d.D, out, S(stream));
  return check_launch("add_coords");
}

int n2nmn_question_prior_add(n2nmn_ctx* c, int N, float* scores, n2nmn_stream stream) {
  N2_REQUIRE(c && scores, N2NMN_EINVAL, "question_prior_add: null argument");
  N2_REQUIRE(c->qpn_h, N2NMN_EINVAL, "question_prior_add: the context has no question prior net "
                                     "(variant VQA with qpn_hidden > 0)");
  N2_REQUIRE(is_committed(c), N2NMN_ENOWEIGHT, "question_prior_add: weights not committed");
  N2_REQUIRE(c->enc_T > 0 && N == c->enc_N, N2NMN_EINVAL,
             "question_prior_add: no matching encoder results in the context");
  return qpn_forward(c, N, scores, nullptr, nullptr, S(stream));
}

int n2nmn_profile_begin(n2nmn_ctx* ctx) {
  N2_REQUIRE(ctx, N2NMN_EINVAL, "profile_begin: null context");
  ctx->prof_recs.clear();
  for (int i = 0; i < 24; ++i) {
    ctx->prof_ms[i] = ctx->prof_flops[i] = ctx->prof_bytes[i] = 0;
    ctx->prof_launches[i] = 0;
  }
  N2_HIP(hipMemset(ctx->walk_stats, 0, sizeof(unsigned long long) * WALK_STATS));
  ctx->prof_on = true;
  return N2NMN_OK;
}

int n2nmn_profile_end(n2nmn_ctx* ctx, n2nmn_stream stream) {
  N2_REQUIRE(ctx, N2NMN_EINVAL, "profile_end: null context");
  ctx->prof_on = false;
  N2_HIP(hipStreamSynchronize(S(stream)));
  for (size_t i = 0; i < ctx->prof_recs.size(); ++i) {
    float ms = 0.f;
    N2_HIP(hipEventElapsedTime(&ms, ctx->prof_events[2 * i], ctx->prof_events[2 * i + 1]));
    const auto& r = ctx->prof_recs[i];
    ctx->prof_ms[r.fam] += ms; ctx->prof_flops[r.fam] += r.flops;
    ctx->prof_bytes[r.fam] += r.bytes; ctx->prof_launches[r.fam] += 1;
  }
  if (ctx->prof_launches[F_WALK] > 0) {
    // the walker decodes its layouts on the device, so the host learns what it executed from the
    // node counters the profiled launches accumulated (algorithmic work)
    unsigned long long st[WALK_STATS];
    N2_HIP(hipMemcpy(st, ctx->walk_stats, sizeof(st), hipMemcpyDeviceToHost));
    const n2nmn_dims& d = ctx->d;
    const double HW = (double)d.H * d.W, D = d.D, M = d.map_dim, Mp = ctx->Mp, E = d.embed_dim_txt,
                 C = d.num_choices, KK = (double)d.kernel_size * d.kernel_size;
    const double n_find = (double)st[0], n_pool_in = (double)st[1], n_pool = (double)st[2],
                 n_text = (double)st[3], n_tr = (double)st[4], n_q = (double)st[5];
    const double n_pre = (double)st[8];          // map passes of walk_find_kernel
    if (n_pre > 0) {
      ctx->prof_bytes[F_WALK_FIND] += 4.0 * n_pre * (HW * Mp + HW);
      ctx->prof_flops[F_WALK_FIND] += n_pre * 5.0 * HW * M;
      ctx->prof_bytes[F_WALK_TMAP] += 4.0 * n_text * Mp;
    }
    ctx->prof_bytes[F_WALK] += 4.0 * (n_find * (HW * Mp + HW) + n_pool * HW * D + n_pool_in * HW +
                                      n_text * E + n_q * C) +
                               ctx->prof_launches[F_WALK] * 4.0 * (5 * E * M + 4 * D * M);
    ctx->prof_flops[F_WALK] += n_find * 5.0 * HW * M + n_pool_in * (2.0 * HW * D + 2.0 * D * M) +
                               n_text * 2.0 * E * M + n_tr * HW * M * (2.0 * KK + 5.0);
    // pooling jobs the walker handed to walk_pool_kernel: their feature read belongs to that kernel
    const double n_def = (double)st[6], n_def_in = (double)st[7];
    if (n_def > 0) {
      const double fb = 4.0 * (n_def * HW * D + n_def_in * (HW + D));
      ctx->prof_bytes[F_WALK] -= 4.0 * (n_def * HW * D + n_def_in * HW);
      ctx->prof_bytes[F_POOL] += fb;
      ctx->prof_flops[F_POOL] += n_def_in * 2.0 * HW * D;
      ctx->prof_flops[F_WALK] -= n_def_in * (2.0 * HW * D + 2.0 * D * M);
      ctx->prof_flops[F_HEADS] += n_def_in * 2.0 * D * M + n_def * 2.0 * M * C;
      ctx->prof_bytes[F_HEADS] += 4.0 * (n_def_in * (D + D * M) + n_def * (M * C + C));
    }
    ctx->walk_jobs_deferred = n_def;
  }
  return (int)ctx->prof_recs.size();
}

int n2nmn_debug_walk_stats(n2nmn_ctx* ctx, uint64_t* out10) {
  N2_REQUIRE(ctx && out10, N2NMN_EINVAL, "debug_walk_stats: null argument");
  static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "counter width");
  N2_HIP(hipSetDevice(ctx->device));
  N2_HIP(hipMemcpy(out10, ctx->walk_stats, sizeof(uint64_t) * WALK_STATS, hipMemcpyDeviceToHost));
  return N2NMN_OK;
}

int n2nmn_profile_num_families(void) { return F_COUNT; }

int n2nmn_profile_get(const n2nmn_ctx* ctx, int family, const char** name, int64_t* launches,
                      double* total_ms, double* flops, double* bytes) {
  N2_REQUIRE(ctx && family >= 0 && family < F_COUNT, N2NMN_EINVAL, "profile_get: bad family");
  if (name) *name = kFamilyNames[family];
  if (launches) *launches = ctx->prof_launches[family];
  if (total_ms) *total_ms = ctx->prof_ms[family];
  if (flops) *flops = ctx->prof_flops[family];
  if (bytes) *bytes = ctx->prof_bytes[family];
  return N2NMN_OK;
}

/* Times `iters` back-to-back encoder-style LSTM launches (layer-0 step + layer-1 step per launch,
 * N rows) of a kernel variant with one HIP event pair; returns the average microseconds per
 * launch in *us.  variant: 0 shipped kernel, 1 loads pinned before MFMAs, 2 loads only,
 * 3 MFMA only, 4 neither (LDS reduce + epilogue only), 5 empty kernel.  jobs: 2 = L0+L1, 1 = L1. */
int n2nmn_debug_lstm_bench(n2nmn_ctx* c, int variant, int rows_per_wg, int njobs, int N, int iters,
                           double* us, n2nmn_stream stream) {
  // 1000 + v: lstm_tile3_kernel (split-operand bf16), v = 0 shipped, else <row groups><variant><stages>
  // (kernels_lstm_tile3.hip launch_lstm_tile3: 403 / 404 64-row workgroups, 803 / 804 128-row, x1x no DMA,
  // x2x no MFMA, x3x DMA + barriers only); needs n2nmn_ctx_set_mode(..., N2NMN_MODE_THROUGHPUT_BF16X3) first
  const int tile3 = variant >= 1000 ? variant - 1000 : -1;
  if (tile3 >= 0) variant = 0;
  N2_REQUIRE(tile3 < 0 || (c && root(c)->b3_on && c->enc_W0h_b3), N2NMN_EINVAL,
             "debug_lstm_bench: set the bf16x3 mode first");
  const int tile = variant >= 20 ? variant - 20 : 0;   // 23 / 24 / 26: lstm_tile_kernel, 3 / 4 / 6 stages
  if (tile) variant = 0;
  const int layout = variant < 10;       // variant >= 10: row-major h (A/B against the packed state)
  variant %= 10;
  N2_REQUIRE(c && us && is_committed(c), N2NMN_EINVAL, "debug_lstm_bench: bad argument");
  N2_REQUIRE(N >= 1 && N <= c->d.N && iters >= 1, N2NMN_EINVAL, "debug_lstm_bench: bad size");
  hipStream_t s = S(stream);
  const int L = c->d.lstm_dim;
  hipEvent_t e0, e1;
  N2_HIP(hipEventCreate(&e0));
  N2_HIP(hipEventCreate(&e1));
  N2_HIP(hipMemsetAsync(c->eh0[0], 0, sizeof(float) * (tile3 >= 0 ? 25 : 6) * (size_t)c->d.N * L, s));
  for (int rep = 0; rep < 2; ++rep) {
    if (rep == 1) N2_HIP(hipEventRecord(e0, s));
    for (int k = 0; k < (rep == 0 ? 5 : iters); ++k) {
      LstmJob jobs[2];
      LstmJob& j0 = jobs[njobs == 2 ? 0 : 1];
      j0 = LstmJob{};
      if (layout) packed_state(c, j0); else rowmajor_a(c, j0);
      j0.active = 1; j0.A0 = c->eh0[(k + 1) & 1]; j0.K = L; j0.Wp = c->enc_W0h_t; j0.Wp64 = c->enc_W0h_64;
      j0.ntiles = L / 4; j0.xtab = c->big_vocab ? c->xproj : c->enc_xtab; j0.xidx = nullptr;
      j0.xidx_const = 1;
      j0.c_in = c->ec0; j0.c_out = c->ec0; j0.h_old = j0.A0; j0.h_new = c->eh0[k & 1];
      LstmJob& j1 = jobs[njobs == 2 ? 1 : 0];
      j1 = LstmJob{};
      if (layout) packed_state(c, j1); else rowmajor_a(c, j1);
      j1.active = 1; j1.A0 = c->eh0[(k + 1) & 1]; j1.A1 = c->eh1[(k + 1) & 1]; j1.K = 2 * L;
      j1.Wp = c->enc_W1_t; j1.Wp64 = c->enc_W1_64; j1.ntiles = L / 4; j1.bias = c->enc_b1_t;
      j1.c_in = c->ec1; j1.c_out = c->ec1; j1.h_old = j1.A1; j1.h_new = c->eh1[k & 1];
      j1.out_seq = c->enc_out;
      if (tile3 >= 0) {
        for (int i = 0; i < 2; ++i) { attach_planes(c, jobs[i]); jobs[i].save_gates = nullptr; }
        if (!lstm_tile3_supported(jobs, njobs, L)) { set_last_error("debug_lstm_bench: tile3 unsupported"); return N2NMN_EINVAL; }
        launch_lstm_tile3(jobs, njobs, N, L, s, tile3);
      } else if (tile) launch_lstm_step(jobs, njobs, N, L, rows_per_wg, s, tile);
      else launch_lstm_step_dbg(jobs, njobs, N, L, rows_per_wg, variant, s);
    }
  }
  N2_HIP(hipEventRecord(e1, s));
  N2_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  N2_HIP(hipEventElapsedTime(&ms, e0, e1));
  *us = 1e3 * ms / iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return check_launch("debug_lstm_bench");
}

int n2nmn_debug_set(n2nmn_ctx* ctx, const char* key, const char* value) {
  N2_REQUIRE(ctx && key, N2NMN_EINVAL, "debug_set: null argument");
  static const char* const known[] = {"tile_min_rows", "eht_rows", "debug_gemm_b3", "train_overlap", "train_bg_wgs",
                                      "train_schedule", "train_chunks", "profile_walk_stats"};
  bool ok = false;
  for (const char* k : known) ok = ok || strcmp(k, key) == 0;
  if (!ok) {
    set_last_error(std::string("debug_set: unknown key '") + key + "'");
    return N2NMN_EKEY;
  }
  if (value) ctx->knobs[key] = value; else ctx->knobs.erase(key);
  return N2NMN_OK;
}

int n2nmn_debug_gemm(n2nmn_ctx* ctx, const float* A, const float* B, const float* bias, float* C,
                     int M, int N, int K, n2nmn_stream stream) {
  N2_REQUIRE(ctx && A && B && C, N2NMN_EINVAL, "debug_gemm: null argument");
  N2_REQUIRE(M > 0 && N > 0 && K > 0 && K % 4 == 0, N2NMN_EINVAL,
             "debug_gemm: K must be a positive multiple of 4");
  // n2nmn_debug_set(ctx, "debug_gemm_b3", "n") (read per call): n < 0: -n launches (timing loops); n > 0: n launches
  // in the split-operand bf16 form -- gemm_dma3_kernel, which only the diagnostic library has (kernels_gemm.hip)
  const char* e3 = knob_str(ctx, "debug_gemm_b3");
#ifdef N2NMN_DIAG
  const bool b3 = e3 && atoi(e3) > 0;
#else
  N2_REQUIRE(!(e3 && atoi(e3) > 0), N2NMN_EINVAL,
             "debug_gemm: the split-operand GEMM is not part of this library (tools/diag/build_diag.py)");
  const bool b3 = false;
#endif
  const int Kp = round_up(K, 32), Np = round_up(N, b3 ? 128 : 64);
  DevBuf pack, pack3;                     // (freed on every exit)
  uint16_t* Bp3 = nullptr;
  N2_HIP(pack.alloc(sizeof(float) * (size_t)Kp * Np));
  float* Bp = pack.as<float>();
  hipStream_t s = S(stream);
  launch_pack_pk(B, N, K, N, Bp, Kp, Np, s);
  GemmArgs g{};
  g.A = A; g.lda = K; g.M = M; g.K = K; g.group_size = 1; g.Bp = Bp; g.Np = Np; g.Kp = Kp;
  g.bias = bias; g.N = N; g.C = C; g.ldc = N; g.n_store = N;
  if (b3) {
    N2_HIP(pack3.alloc(sizeof(uint16_t) * 3 * (size_t)Kp * Np));
    Bp3 = pack3.as<uint16_t>();
    launch_pack_pk_b3(Bp, Kp, Np, Bp3, s);
    g.Bp3 = Bp3;
  }
  const int reps = e3 && atoi(e3) < 0 ? -atoi(e3) : 1;       // (negative: fp32 form, that many launches)
  // (the split-operand kernel is off in the product path, kernels_gemm.hip use_gemm_dma3; this debug entry
  // keeps exercising it directly -- single-stream accuracy and timing)
  for (int i = 0; i < (b3 ? atoi(e3) : reps); ++i) {
    const int tiles3 = ((g.M + 63) / 64) * ((g.n_store + 127) / 128);      // (the product path's old threshold)
    if (b3 && gemm_dma3_supported(g) && tiles3 >= 256) launch_gemm_dma3(&g, 1, s);
    else launch_gemm_pk(g, s);
  }
  N2_HIP(hipStreamSynchronize(s));       // debug entry only: the packs are freed right away
  return check_launch("debug_gemm");
}

// out = [relu](A . W + bias): util/cnn.py:87-126 (fc_layer / fc_relu_layer) and, on im2col rows, the VALID
// strided convolutions of models_shapes/shapes_convnet.py:8-17 (conv_relu_layer).  W is packed per call (a
// small operator of the SHAPES plumbing, BASELINE configs[0]; the hot-path GEMMs use commit-time packs).
int n2nmn_fc_forward(n2nmn_ctx* ctx, const float* A, const float* W, const float* bias, float* out,
                     int M, int N, int K, int relu, n2nmn_stream stream) {
  N2_REQUIRE(ctx && A && W && out, N2NMN_EINVAL, "fc_forward: null argument");
  N2_REQUIRE(M > 0 && N > 0 && K > 0 && K % 4 == 0, N2NMN_EINVAL,
             "fc_forward: K must be a positive multiple of 4");
  const int Kp = round_up(K, 32), Np = round_up(N, 64);
  DevBuf pack;                            // (freed on every exit, after the stream has drained)
  N2_HIP(pack.alloc(sizeof(float) * (size_t)Kp * Np));
  float* Bp = pack.as<float>();
  hipStream_t s = S(stream);
  launch_pack_pk(W, N, K, N, Bp, Kp, Np, s);
  GemmArgs g{};
  g.A = A; g.lda = K; g.M = M; g.K = K; g.group_size = 1; g.Bp = Bp; g.Np = Np; g.Kp = Kp;
  g.bias = bias; g.N = N; g.C = out; g.ldc = N; g.n_store = N; g.relu = relu ? 1 : 0;
  launch_gemm_pk(g, s);
  const hipError_t done = hipStreamSynchronize(s);       // the pack may go only behind the GEMM that reads it
  N2_HIP(done);
  return check_launch("fc_forward");
}

}  // extern "C"
