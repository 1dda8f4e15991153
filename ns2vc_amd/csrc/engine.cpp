// Host side of libns2vc_hip.so: weight packing, workspace, launch plan,
// hipGraph-captured sampling loop and the C ABI declared in include/ns2vc_hip.h.
//
// The plan restates the op sequence of the reference forward
// (unet1d/unet_1d_condition.py:743-1037; blocks unet1d/unet_1d_blocks.py:949-1016,
// 1071-1097, 602-623, 2070-2131, 2182-2207; resnet.py:591-641; transformer_1d.py:256-295;
// attention.py:130-203) as a flat list of kernel launches on channels-last tensors.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include <mutex>

using namespace ns2vc;

namespace ns2vc {
hipError_t pack_ffn_stream(const float* w1p, const float* w2f, const float* w0, int dim, int prec, std::vector<unsigned short>& out);   // ffn.hip
hipError_t pack_rowchain_stream(const float* w1, const float* w2, int dim, int n2, int prec, std::vector<unsigned short>& out, int slices = 1);
int rowchain_slice_blocks(int n2, int slices);   // rowchain.hip
void set_ffn_trace(unsigned long long* p);
typedef ::ns2vc_geglu_args GegluArgs;
bool geglu_eligible(int dim, int T, int prec);                                                                                                       // geglu.hip
hipError_t pack_geglu_stream(const float* w1p, const float* bias1p, int dim, int prec, std::vector<unsigned short>& stream, std::vector<float>& consts);
hipError_t launch_geglu(const GegluArgs& a, int prec, hipStream_t s);
hipError_t init_geglu_attributes();
void set_rc_trace(unsigned long long* p);
void set_ts_trace(unsigned long long* p);
void set_gg_trace(unsigned long long* p);
void set_attn_optimistic(int on);
}

namespace {

int g_geglu_min_rows = 4608;   // rows from which the token-stationary GEGLU kernel replaces the GEMM (tests: ns2vc_debug_set_geglu_min_rows)

thread_local std::string g_err;

int fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return 1;
}
#define HIPCHK(expr)                                                                                       \
  do {                                                                                                     \
    hipError_t _e = (expr);                                                                                \
    if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
  size_t numel() const { size_t n = 1; for (auto s : shape) n *= (size_t)s; return n; }
};

struct PackedW {     // device-resident packed GEMM weight
  void* w = nullptr;
  void* wt = nullptr;      // k = 3 convs: the same weights tile-major for conv3ts_kernel (pack_conv3_tiled)
  float* bias = nullptr;
  float* wsum = nullptr;   // [N] row sums of the packed (rounded) weights: LayerNorm-by-linearity consumers only
  int N = 0, K = 0;
};

struct ResnetW {
  std::string prefix;
  int cin = 0, cout = 0, temb_off = 0;
  bool shortcut = false;
  float *n1g = nullptr, *n1b = nullptr, *n2g = nullptr, *n2b = nullptr;
  PackedW conv1, conv2, sc;
};
struct AttnW {
  std::string prefix;
  int dim = 0, kv_off = 0;
  float *ng = nullptr, *nb = nullptr;
  PackedW proj_in, qkv, o1, q2, o2, ff1, ff2, proj_out;
  PackedW ffpo;    // ff.net.2 folded into proj_out: [W_po W_2 | W_po], K = 4*dim + dim (see pack_all)
  void* ffn_pre_stream = nullptr; // the same stream with attn2.to_out in front (pre-stage of the fused kernel)
  void* ffn_stream = nullptr;     // fused feed-forward + proj_out (ffn.hip): weight tile stream ...
  float* ffn_consts = nullptr;    // ... and (rowsum, bias) per packed ff.net.0 row; 16-bit precisions, dim <= 256 only
  void* geglu_stream = nullptr;   // token-stationary GEGLU projection (geglu.hip): weight tile stream and constants; 16-bit precisions, dim 384
  float* geglu_consts = nullptr;
  // token-local chains (rowchain.hip; 16-bit precisions, dim <= 256): proj_in -> norm1 -> q|k|v and attn1.to_out -> norm2 -> attn2.to_q
  void *chain_in = nullptr, *chain_mid = nullptr;          // weight tile streams
  void* chain_in_s2 = nullptr;                              // ... of the first chain packed for two N-slices (dim 384, r4)
  float *chain_in_consts = nullptr, *chain_mid_consts = nullptr;   // (rowsum, bias) per LayerNorm-folded stage-2 row
};
struct BlockW {
  std::string kind;   // down | mid | up
  int index = 0, level = 0, channels = 0;
  std::vector<ResnetW> res;
  std::vector<AttnW> attn;
  int sampler = 0;    // 0 none, 1 down, 2 up
  PackedW samp;
};

struct Op {
  std::string name;
  std::function<hipError_t(hipStream_t)> fn;
  int kind = 0;          // 0 other, 1 implicit GEMM, 2 attention, 3 norm statistics, 4 copy
  double flops = 0.0;    // algorithmic FLOPs (2*MAC) of this launch
  double bytes = 0.0;    // algorithmic (compulsory) HBM bytes: operands read once + result written once
  // what the forward's one clear launch does for THIS op (zero the arrival words of a cooperative GroupNorm prologue): only
  // ns2vc_unet_profile_forward needs it, because it repeats an op without the rest of the forward in between
  std::function<hipError_t(hipStream_t)> rearm;
};
struct Tap {
  std::string name;
  float* copy = nullptr;
  int rows = 0, cols = 0;
};

}  // namespace

struct ns2vc_unet {
  ns2vc_unet_cfg cfg{};
  std::map<std::string, HostTensor> raw;
  std::vector<std::pair<std::string, std::vector<int64_t>>> expected;
  int prec = -1;
  bool finalized = false;
  std::vector<void*> weight_allocs;

  // packed weights
  std::vector<BlockW> blocks;
  PackedW conv_in_x, conv_in_c, conv_out, temb_all, kv_all, pool_qkv;
  PackedW conv_in_xp, conv_in_cp, conv_outp;                     // r6, 16-bit engines: the first three against hi + lo operand pairs (option split_io; K = 3 x)
  PackedW conv_in_x32, conv_in_c32, conv_out32, temb_all32;      // r6, 16-bit engines: the same four in fp32 (option exact_io)
  float *t_w1t = nullptr, *t_b1 = nullptr, *t_w2t = nullptr, *t_b2 = nullptr;
  float *p_n1g = nullptr, *p_n1b = nullptr, *p_pos = nullptr, *p_projT = nullptr, *p_projb = nullptr, *p_n2g = nullptr, *p_n2b = nullptr;
  float *out_ng = nullptr, *out_nb = nullptr;
  int n_temb = 0, n_kv = 0;
  int CP = 128;       // padded latent channels in the engine's channels-last x buffers

  // workspace / plan
  int B = 0, T = 0, Lp = 0;
  void* arena = nullptr;
  size_t arena_bytes = 0, arena_used = 0;
  std::vector<Op> cond_ops, fwd_ops;
  size_t cond_split = 0;      // cond_ops[0 .. cond_split) depend on the content only, the rest on the prompt (+ mask) only
  bool debug = false;
  int device = 0;             // HIP device the engine (weights, arena, graph) lives on; every entry point binds to it
  // LayerNorm by linearity (csrc/gemm.hip): no normalisation pass; NS2VC_LN_LINEAR=0 (or ns2vc_unet_set_option) restores
  // the ln_apply kernels.  The consumers record max |mean| * rstd over all LayerNorm rows in `ln_health`: the 16-bit
  // modes round the RAW x before centring, so their error on a row grows with that ratio (ns2vc_unet_ln_ratio).
  bool ln_linear = true;
  // ff.net.2 folded into proj_out at pack time (one GEMM with a second K segment instead of two launches; the
  // post-feed-forward stream tensor is never materialised).  NS2VC_FOLD_FF=0 restores the two launches.
  bool fold_ff = true;
  // GEGLU feed-forward + ff.net.2 + proj_out in ONE launch per transformer block (csrc/ffn.hip) where eligible
  // (16-bit precisions, dim 128 / 256, LayerNorm by linearity and the fold on).  NS2VC_FUSE_FFN=0 restores the two GEMMs.
  bool fuse_ffn = true;
  bool fuse_rows = true;     // proj_in+q|k|v and attn1.to_out+attn2.to_q as one launch each (rowchain.hip)
  bool attn_fp8 = false;     // PV product of every attention on the fp8 MFMA (16-bit precisions; BASELINE config 5's fp8 path; costs parity)
  bool fuse_ffn_pre = true;  // attn2.to_out + residual computed inside the fused feed-forward kernel
  bool fuse_geglu = true;    // r5: token-stationary GEGLU projection (csrc/geglu.hip) where the fused feed-forward does not apply (dim 384)
  bool fuse_rows_gn = true;  // ... and the transformer's GroupNorm computed in the prologue of the first of them
  // GroupNorm-apply as the prologue of the GEMM that consumes it (gemm.hip gn_prologue, ns2vc_gemm_args.gnp_*) wherever the norm has
  // one source, one consumer and epilogue statistics: resnet norm2 -> conv2, norm1 -> conv1 of the resnets without a
  // shortcut, the transformer norm in front of a plain proj_in, conv_norm_out -> conv_out.  Bit-identical to the gn_apply
  // launches it removes (210 -> 174 launches at the bench shape).  NS2VC_FUSE_GN_GEMM=0 restores them.  (r3 had this off: not
  // run-to-run deterministic; root cause and fix in r4, profiles/r04_gn_prologue_rootcause.txt.)
  bool fuse_gn_gemm = true;
  bool fuse_gn_cat = true;                // ... also where the norm's input is a concat of two tensors and / or a raw operand copy is wanted (first resnet of a level, up blocks)
  bool gn_coop = true;                    // the column tiles of one row block split the GroupNorm prologue's rows between them (ns2vc_gemm_args.gnp_sync)
  bool conv_ts = true;         // k = 3 convolutions on the tap-sharing kernel (convts.hip, r5)
  bool conv_wtiled = true;     // ... reading tile-major weights (PackedW.wt)
  bool split_io = true;       // r6 (16-bit engines): conv_in and conv_out -- 23 % of the forward error's energy in two launches (profiles/r06_error_budget.txt) -- on hi + lo operand
                              //   pairs: x * w ~= hi(x) hi(w) + lo(x) hi(w) + hi(x) lo(w), three times the K of two small convolutions instead of their fp32 MFMA rate (exact_io)
                              //   ON: forward error 8.08e-4 -> 7.10e-4 at the bench shape for +0.3 % of the step (profiles/r06_ab_split_io.txt)
  bool exact_io = false;       // r6 (16-bit engines): conv_in, conv_out and the time_emb_proj GEMM -- three single launches that carry 23 + 6 % of the forward error's energy
                               // (profiles/r06_error_budget.txt) -- with fp32 operands on the fp32 MFMA: a precision-for-time knob
  float* content_f32 = nullptr;      // ... their fp32 inputs: the content rows, SiLU(emb)
  float* emb_act_f32 = nullptr;
  int conv_out_prec = -1;
  bool fork_temb = false;      // (measured: +3 % -- a graph with a parallel branch replays SLOWER than the linear chain, 3.696 vs 3.587 ms/step, profiles/r06_ab_fork_temb.txt; off)
                               // r6: inside the captured step graph the timestep-embedding branch (time_embed + time_emb_proj.all: two small launches that depend on the step
                               // counter only) runs BESIDE conv_in and the first resnet's conv1 on a forked stream and joins in front of the first consumer of the scale / shift rows
  int temb_begin = -1, temb_end = -1, temb_join = -1;        // ... their places in fwd_ops (build_plan)
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool fuse_xattn = false;     // the prompt cross-attention of attn2 runs inside the fused feed-forward kernel (ffn.hip ATT, r6): no attn2.sdpa launch at dim 128 / 256.
                               // Correct (kernel + engine tests), 10 launches fewer, measured SLOWER (3.53 vs 3.39 ms/step; profiles/r06_ab_fuse_xattn.txt): with 8 waves per workgroup
                               // (2 per SIMD, 212-252 VGPRs) the attention phase runs 2-3x off its VALU bound -- a tested option, off
  bool fuse_solver = false;    // the sampling loop's solver update runs in conv_out's epilogue (GemmArgs.sol_*, r6) instead of as its own launch: bit-identical,
                               // one launch and 88 MB of HBM traffic less per step, but 0.1-0.4 % SLOWER in three same-box A/Bs (profiles/r06_ab_fuse_solver.txt) -- a tested option, off
  GemmArgs conv_out_g;         // ... conv_out's launch arguments and its place in fwd_ops, kept by build_plan for that
  int conv_out_idx = -1;
  bool warned_wtiled = false;
  int bn128_min = 160;         // workgroups a 128-column tap-sharing tiling must still give on THIS engine's device (5/8 of its CUs); per engine, not process-global
  bool gn_inloop = false;      // ... normalising inside its K loop (gnpro.h GnInloop, r6) instead of materialising the rows in a prologue (GemmArgs.algo 0 vs 2).
                               // Bit-identical results, measured SLOWER (profiles/r06_ab_gn_inloop.txt: 3.85 vs 3.61 ms/step; SiLU of a 128 x 64 chunk is 1.7 k VALU cycles per SIMD,
                               // more than the consumers need for the chunk, and every column tile repeats it): a tested option, off
  int gn_coop_min = 2;         // fewest column tiles of a row block for which the cooperative prologue is used (tuning: NS2VC_GN_COOP_MIN under NS2VC_DEBUG_ENV)
  int cus = 256;               // compute units of this device (hipDeviceProp_t.multiProcessorCount): the "one round of workgroups" heuristics scale with it
  int xcd_probe = -1;          // misc.hip's placement probe of this device: 1 = workgroup ids 8 apart share an XCD
  bool slice_rows = true;      // first row chain of a dim-384 block as two N-slices per token block (r4; see Planner::transformer)
  unsigned* ln_health = nullptr;
  bool attn_optimistic = true;   // attention without the per-tile maximum + exact fallback (attn.hip OPT); 0 = exact pass only, on every device
  unsigned* attn_fallbacks = nullptr;     // device counter: workgroups that needed the fallback (ns2vc_unet_attn_fallbacks)
  unsigned long long coef_hash = 0;       // FNV-1a of the loaded solver table (handoff compares)
  std::vector<Tap> taps;
  bool has_mask = false;

  // named persistent buffers
  float *xe = nullptr, *xbar = nullptr, *d1 = nullptr, *mprev = nullptr, *x0 = nullptr;
  float *content_conv = nullptr, *prompt = nullptr, *maskbias = nullptr;
  float *aug = nullptr, *emb = nullptr, *temb = nullptr;
  float *seq = nullptr, *pool_qkv_buf = nullptr, *pooled = nullptr;
  void *xe_op = nullptr, *content_op = nullptr, *prompt_op = nullptr, *emb_act_op = nullptr, *kv = nullptr, *seq_op = nullptr;   // operand-typed
  float *t_dev = nullptr;
  uint8_t* mask_dev = nullptr;
  int* step_dev = nullptr;
  size_t stats_bytes = 8;
  float* coef_dev = nullptr;
  float* temb_table = nullptr;   // [kMaxSteps][time_embed_dim]: the timestep MLP of every row of the solver table (sampling loop only)
  bool temb_table_valid = false;
  int steps = 0;
  int next_step = -1;            // sampling loop position (host mirror of step_dev + 1); -1 = no loop begun
  bool use_step_table = false;
  // LayerNorm-health read-out (ns2vc_unet_ln_ratio*): snapshot slot in the arena, pinned host mailbox, completion event
  unsigned* ln_mail = nullptr;
  hipEvent_t ln_event = nullptr;
  bool ln_posted = false;

  hipStream_t cap_stream = nullptr;
  hipGraphExec_t step_graph = nullptr;

  ~ns2vc_unet() {
    if (step_graph) (void)hipGraphExecDestroy(step_graph);
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
    if (side_stream) (void)hipStreamDestroy(side_stream);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (ln_event) (void)hipEventDestroy(ln_event);
    if (ln_mail) (void)hipHostFree(ln_mail);
    if (arena) (void)hipFree(arena);
    if (coef_dev) (void)hipFree(coef_dev);
    if (temb_table) (void)hipFree(temb_table);
    for (void* p : weight_allocs) (void)hipFree(p);
  }
};

namespace {

// ------------------------------------------------------------------------------------
// topology (same rules as ns2vc_amd/spec.py::topology)
// ------------------------------------------------------------------------------------
std::vector<BlockW> make_topology(const ns2vc_unet_cfg& c) {
  std::vector<BlockW> out;
  const int n = c.n_levels;
  int out_c = c.block_out_channels[0];
  for (int i = 0; i < n; ++i) {
    const int in_c = out_c;
    out_c = c.block_out_channels[i];
    BlockW b;
    b.kind = "down"; b.index = i; b.level = i; b.channels = out_c;
    const bool cross = (i != n - 1);      // ("CrossAttnDownBlock2D",)*3 + ("DownBlock2D",)
    for (int j = 0; j < c.layers_per_block; ++j) {
      ResnetW r;
      r.prefix = "down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j);
      r.cin = j == 0 ? in_c : out_c; r.cout = out_c; r.shortcut = r.cin != r.cout;
      b.res.push_back(r);
      if (cross) {
        AttnW a;
        a.prefix = "down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j);
        a.dim = out_c;
        b.attn.push_back(a);
      }
    }
    b.sampler = (i != n - 1) ? 1 : 0;
    out.push_back(b);
  }
  {
    BlockW m;
    const int mc = c.block_out_channels[n - 1];
    m.kind = "mid"; m.index = 0; m.level = n - 1; m.channels = mc;
    for (int j = 0; j < 2; ++j) {
      ResnetW r;
      r.prefix = "mid_block.resnets." + std::to_string(j);
      r.cin = r.cout = mc; r.shortcut = false;
      m.res.push_back(r);
    }
    AttnW a;
    a.prefix = "mid_block.attentions.0"; a.dim = mc;
    m.attn.push_back(a);
    out.push_back(m);
  }
  out_c = c.block_out_channels[n - 1];
  for (int i = 0; i < n; ++i) {
    const int prev_c = out_c;
    out_c = c.block_out_channels[n - 1 - i];
    const int in_c = c.block_out_channels[std::max(n - 2 - i, 0)];
    BlockW b;
    b.kind = "up"; b.index = i; b.level = n - 1 - i; b.channels = out_c;
    const bool cross = (i != 0);          // ("UpBlock2D",) + ("CrossAttnUpBlock2D",)*3
    const int nl = c.layers_per_block + 1;
    for (int j = 0; j < nl; ++j) {
      ResnetW r;
      r.prefix = "up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j);
      const int skip_c = (j == nl - 1) ? in_c : out_c;
      r.cin = (j == 0 ? prev_c : out_c) + skip_c; r.cout = out_c; r.shortcut = true;
      b.res.push_back(r);
      if (cross) {
        AttnW a;
        a.prefix = "up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j);
        a.dim = out_c;
        b.attn.push_back(a);
      }
    }
    b.sampler = (i != n - 1) ? 2 : 0;
    out.push_back(b);
  }
  return out;
}

void expect(ns2vc_unet* h, const std::string& k, std::vector<int64_t> shape) { h->expected.emplace_back(k, std::move(shape)); }

void build_expected(ns2vc_unet* h) {
  const auto& c = h->cfg;
  const int64_t c0 = c.block_out_channels[0], temb = 4 * c0, cross = c.cross_attention_dim;
  const int64_t cin = c.latent_channels + c.content_channels;
  expect(h, "conv_in.weight", {c0, cin, 3}); expect(h, "conv_in.bias", {c0});
  expect(h, "time_embedding.linear_1.weight", {temb, c0}); expect(h, "time_embedding.linear_1.bias", {temb});
  expect(h, "time_embedding.linear_2.weight", {temb, temb}); expect(h, "time_embedding.linear_2.bias", {temb});
  expect(h, "add_embedding.norm1.weight", {cross}); expect(h, "add_embedding.norm1.bias", {cross});
  expect(h, "add_embedding.pool.positional_embedding", {1, cross});
  for (const char* p : {"k_proj", "q_proj", "v_proj"}) {
    expect(h, std::string("add_embedding.pool.") + p + ".weight", {cross, cross});
    expect(h, std::string("add_embedding.pool.") + p + ".bias", {cross});
  }
  expect(h, "add_embedding.proj.weight", {temb, cross}); expect(h, "add_embedding.proj.bias", {temb});
  expect(h, "add_embedding.norm2.weight", {temb}); expect(h, "add_embedding.norm2.bias", {temb});
  for (const auto& b : h->blocks) {
    for (const auto& a : b.attn) {
      const int64_t d = a.dim;
      const std::string t = a.prefix + ".transformer_blocks.0";
      expect(h, a.prefix + ".norm.weight", {d}); expect(h, a.prefix + ".norm.bias", {d});
      expect(h, a.prefix + ".proj_in.weight", {d, d, 1}); expect(h, a.prefix + ".proj_in.bias", {d});
      for (const char* nn : {"norm1", "norm2", "norm3"}) { expect(h, t + "." + nn + ".weight", {d}); expect(h, t + "." + nn + ".bias", {d}); }
      expect(h, t + ".attn1.to_q.weight", {d, d}); expect(h, t + ".attn1.to_k.weight", {d, d}); expect(h, t + ".attn1.to_v.weight", {d, d});
      expect(h, t + ".attn1.to_out.0.weight", {d, d}); expect(h, t + ".attn1.to_out.0.bias", {d});
      expect(h, t + ".attn2.to_q.weight", {d, d}); expect(h, t + ".attn2.to_k.weight", {d, cross}); expect(h, t + ".attn2.to_v.weight", {d, cross});
      expect(h, t + ".attn2.to_out.0.weight", {d, d}); expect(h, t + ".attn2.to_out.0.bias", {d});
      expect(h, t + ".ff.net.0.proj.weight", {8 * d, d}); expect(h, t + ".ff.net.0.proj.bias", {8 * d});
      expect(h, t + ".ff.net.2.weight", {d, 4 * d}); expect(h, t + ".ff.net.2.bias", {d});
      expect(h, a.prefix + ".proj_out.weight", {d, d, 1}); expect(h, a.prefix + ".proj_out.bias", {d});
    }
    for (const auto& r : b.res) {
      expect(h, r.prefix + ".norm1.weight", {r.cin}); expect(h, r.prefix + ".norm1.bias", {r.cin});
      expect(h, r.prefix + ".conv1.weight", {r.cout, r.cin, 3}); expect(h, r.prefix + ".conv1.bias", {r.cout});
      expect(h, r.prefix + ".time_emb_proj.weight", {2 * r.cout, temb}); expect(h, r.prefix + ".time_emb_proj.bias", {2 * r.cout});
      expect(h, r.prefix + ".norm2.weight", {r.cout}); expect(h, r.prefix + ".norm2.bias", {r.cout});
      expect(h, r.prefix + ".conv2.weight", {r.cout, r.cout, 3}); expect(h, r.prefix + ".conv2.bias", {r.cout});
      if (r.shortcut) { expect(h, r.prefix + ".conv_shortcut.weight", {r.cout, r.cin, 1}); expect(h, r.prefix + ".conv_shortcut.bias", {r.cout}); }
    }
    if (b.sampler) {
      const std::string p = b.kind == "down" ? "down_blocks." + std::to_string(b.index) + ".downsamplers.0.conv"
                                             : "up_blocks." + std::to_string(b.index) + ".upsamplers.0.conv";
      expect(h, p + ".weight", {b.channels, b.channels, 3}); expect(h, p + ".bias", {b.channels});
    }
  }
  expect(h, "conv_norm_out.weight", {c0}); expect(h, "conv_norm_out.bias", {c0});
  expect(h, "conv_out.weight", {c.latent_channels, c0, 3}); expect(h, "conv_out.bias", {c.latent_channels});
}

// ------------------------------------------------------------------------------------
// weight packing (host, fp32/double) -> device
// ------------------------------------------------------------------------------------
// sum_k of the operand-rounded weight row (what the MFMA will actually multiply), accumulated in double
static std::vector<float> rounded_rowsum(const float* rows, int N, int K, int Np, int prec) {
  std::vector<float> ws(Np, 0.f);
  for (int n = 0; n < N; ++n) {
    double acc = 0.0;
    for (int k = 0; k < K; ++k) {
      float v = rows[(size_t)n * K + k];
      if (prec != PREC_F32) v = op16_bits_to_f32(f32_to_op16_bits(v, prec), prec);
      acc += (double)v;
    }
    ws[n] = (float)acc;
  }
  return ws;
}

struct Packer {
  ns2vc_unet* h;
  int err = 0;

  const HostTensor& T(const std::string& k) {
    auto it = h->raw.find(k);
    if (it == h->raw.end()) { err = fail("weight %s missing", k.c_str()); static HostTensor empty; return empty; }
    return it->second;
  }
  float* upload_f32(const std::vector<float>& v) {
    void* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(float)) != hipSuccess) { err = fail("hipMalloc failed (weights)"); return nullptr; }
    h->weight_allocs.push_back(d);
    if (hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) err = fail("hipMemcpy failed (weights)");
    return (float*)d;
  }
  float* vec(const std::string& k) { return upload_f32(T(k).data); }
  // rows: [N][K] fp32, bias: [N] or empty.  Pads N to a multiple of 128 with zero rows.
  // tile3_ctot > 0: a k = 3 conv weight (K = 3 * tile3_ctot + tile3_c2): also packed tile-major for the tap-sharing kernel
  int prec_override = -1;      // pack the next weights in this precision instead of the engine's (exact_io: fp32 copies of four weights of a 16-bit engine)
  PackedW pack(const std::vector<float>& rows, int N, int K, const std::vector<float>& bias, bool want_wsum = false, int tile3_ctot = 0, int tile3_c2 = 0) {
    const int wprec = prec_override >= 0 ? prec_override : h->prec;
    PackedW p;
    const int Np = round_up(N, 128);
    p.N = Np; p.K = K;
    // (the second, tile-major copy only when the tap-sharing kernel will read it; when the layout cannot be built the kernel falls back to the
    //  [N][K] rows and the engine says so once)
    if (tile3_ctot > 0 && K == 3 * tile3_ctot + tile3_c2 && h->conv_ts && h->conv_wtiled) {
      std::vector<unsigned char> img;
      if (pack_conv3_tiled(rows.data(), N, tile3_ctot, tile3_c2, wprec, img) != hipSuccess) {
        if (!h->warned_wtiled) { fprintf(stderr, "ns2vc: tile-major conv weights unavailable for a %d x %d weight (channels not a multiple of the chunk): row-major fallback\n", N, K); h->warned_wtiled = true; }
      } else {
        void* dt = nullptr;
        if (hipMalloc(&dt, img.size()) != hipSuccess) { err = fail("hipMalloc failed (tile-major weights)"); return p; }
        h->weight_allocs.push_back(dt);
        if (hipMemcpy(dt, img.data(), img.size(), hipMemcpyHostToDevice) != hipSuccess) err = fail("hipMemcpy failed");
        p.wt = dt;
      }
    }
    if (want_wsum) p.wsum = upload_f32(rounded_rowsum(rows.data(), N, K, Np, wprec));
    void* d = nullptr;
    if (wprec != PREC_F32) {
      std::vector<uint16_t> q((size_t)Np * K, 0);
      for (size_t i = 0; i < (size_t)N * K; ++i) q[i] = f32_to_op16_bits(rows[i], wprec);
      if (hipMalloc(&d, q.size() * 2) != hipSuccess) { err = fail("hipMalloc failed (weights)"); return p; }
      h->weight_allocs.push_back(d);
      if (hipMemcpy(d, q.data(), q.size() * 2, hipMemcpyHostToDevice) != hipSuccess) err = fail("hipMemcpy failed");
    } else {
      std::vector<float> q((size_t)Np * K, 0.f);
      memcpy(q.data(), rows.data(), (size_t)N * K * sizeof(float));
      if (hipMalloc(&d, q.size() * 4) != hipSuccess) { err = fail("hipMalloc failed (weights)"); return p; }
      h->weight_allocs.push_back(d);
      if (hipMemcpy(d, q.data(), q.size() * 4, hipMemcpyHostToDevice) != hipSuccess) err = fail("hipMemcpy failed");
    }
    p.w = d;
    if (!bias.empty()) {
      std::vector<float> bb(Np, 0.f);
      memcpy(bb.data(), bias.data(), (size_t)N * sizeof(float));
      p.bias = upload_f32(bb);
    }
    return p;
  }

  // conv weight (Cout, Cin, taps) -> rows [Cout][tap*CinP + c'], channels [c_lo, c_hi) of the input, padded to CinP
  std::vector<float> conv_rows(const HostTensor& w, int c_lo, int c_hi, int CinP) {
    const int Cout = (int)w.shape[0], Cin = (int)w.shape[1], taps = (int)w.shape[2];
    std::vector<float> rows((size_t)Cout * taps * CinP, 0.f);
    for (int n = 0; n < Cout; ++n)
      for (int c = c_lo; c < c_hi; ++c)
        for (int t = 0; t < taps; ++t) rows[((size_t)n * taps + t) * CinP + (c - c_lo)] = w.data[((size_t)n * Cin + c) * taps + t];
    return rows;
  }
  // rows [N][taps][C] -> [N][taps][3 C] for activations laid out [hi(x) | lo(x) | hi(x)]: (w, w, w - round(w)); pack() then rounds each -> (hi(w), hi(w), lo(w))
  std::vector<float> pair_rows(const std::vector<float>& rows, int N, int taps, int C) {
    std::vector<float> out((size_t)N * taps * 3 * C);
    for (size_t nt = 0; nt < (size_t)N * taps; ++nt)
      for (int c = 0; c < C; ++c) {
        const float w = rows[nt * C + c];
        out[nt * 3 * C + c] = out[nt * 3 * C + C + c] = w;
        out[nt * 3 * C + 2 * C + c] = w - op16_bits_to_f32(f32_to_op16_bits(w, h->prec), h->prec);
      }
    return out;
  }
  PackedW conv(const std::string& prefix, bool tile3 = false) {
    const HostTensor& w = T(prefix + ".weight");
    if (err) return {};
    const int Cout = (int)w.shape[0], Cin = (int)w.shape[1], taps = (int)w.shape[2];
    return pack(conv_rows(w, 0, Cin, Cin), Cout, taps * Cin, T(prefix + ".bias").data, false, (tile3 && taps == 3) ? Cin : 0, 0);
  }
  // Linear whose input is LayerNorm(x): fold gamma into W and beta into the bias.
  void ln_fold(const HostTensor& w, const HostTensor* b, const HostTensor& g, const HostTensor& be, std::vector<float>& rows,
               std::vector<float>& bias) {
    const int N = (int)w.shape[0], K = (int)w.shape[1];
    const size_t r0 = rows.size(), b0 = bias.size();
    rows.resize(r0 + (size_t)N * K);
    bias.resize(b0 + N);
    for (int n = 0; n < N; ++n) {
      double acc = b ? (double)b->data[n] : 0.0;
      for (int k = 0; k < K; ++k) {
        const float wv = w.data[(size_t)n * K + k];
        rows[r0 + (size_t)n * K + k] = wv * g.data[k];
        acc += (double)wv * (double)be.data[k];
      }
      bias[b0 + n] = (float)acc;
    }
  }
};

// weight stream + stage-2 constants of one token-local chain (rowchain.hip): w1 [d][d] plain, w2 [n2][d] LayerNorm-folded
// with its folded bias; only for the shapes / precisions the kernel serves, otherwise both outputs stay null
static int chain_stream(Packer& P, const std::vector<float>& w1, const std::vector<float>& w2, const std::vector<float>& b2, int d, int n2,
                        void*& stream_dev, float*& consts_dev, void** sliced_dev = nullptr) {
  ns2vc_unet* h = P.h;
  stream_dev = nullptr; consts_dev = nullptr;
  if (P.err) return 1;
  if (!rowchain_eligible(d, n2, 64, h->prec)) return 0;
  std::vector<unsigned short> st;
  if (pack_rowchain_stream(w1.data(), w2.data(), d, n2, h->prec, st) != hipSuccess) return fail("row-chain stream packing failed");
  void* dev = nullptr;
  if (hipMalloc(&dev, st.size() * 2) != hipSuccess) return fail("hipMalloc failed (weights)");
  h->weight_allocs.push_back(dev);
  if (hipMemcpy(dev, st.data(), st.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy failed (weights)");
  if (sliced_dev && d == 384) {     // the same chain packed for two N-slices (planner: where 64-token blocks fill less than half of the chip)
    *sliced_dev = nullptr;
    if (pack_rowchain_stream(w1.data(), w2.data(), d, n2, h->prec, st, 2) != hipSuccess) return fail("row-chain stream packing failed (sliced)");
    void* dev2 = nullptr;
    if (hipMalloc(&dev2, st.size() * 2) != hipSuccess) return fail("hipMalloc failed (weights)");
    h->weight_allocs.push_back(dev2);
    if (hipMemcpy(dev2, st.data(), st.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy failed (weights)");
    *sliced_dev = dev2;
  }
  const std::vector<float> ws = rounded_rowsum(w2.data(), n2, d, n2, h->prec);
  std::vector<float> cs((size_t)n2 * 2);
  for (int r = 0; r < n2; ++r) { cs[2 * r] = ws[r]; cs[2 * r + 1] = b2[r]; }
  stream_dev = dev;
  consts_dev = P.upload_f32(cs);
  return P.err;
}

int pack_all(ns2vc_unet* h) {
  Packer P{h};
  const auto& c = h->cfg;
  const int c0 = c.block_out_channels[0], temb = 4 * c0, cross = c.cross_attention_dim;
  const int lat = c.latent_channels, CP = h->CP;
  // conv_in split: x part (latent channels, padded to CP) per step, content part hoisted
  {
    const HostTensor& w = P.T("conv_in.weight");
    if (P.err) return 1;
    h->conv_in_x = P.pack(P.conv_rows(w, 0, lat, CP), c0, 3 * CP, {}, false, CP, 0);
    h->conv_in_c = P.pack(P.conv_rows(w, lat, lat + c.content_channels, c.content_channels), c0, 3 * c.content_channels, P.T("conv_in.bias").data);
    if (h->prec != PREC_F32) {
      const int cc = c.content_channels;
      h->conv_in_xp = P.pack(P.pair_rows(P.conv_rows(w, 0, lat, CP), c0, 3, CP), c0, 9 * CP, {}, false, 3 * CP, 0);
      h->conv_in_cp = P.pack(P.pair_rows(P.conv_rows(w, lat, lat + cc, cc), c0, 3, cc), c0, 9 * cc, P.T("conv_in.bias").data, false, 3 * cc, 0);
    }
    if (h->prec != PREC_F32) {
      P.prec_override = PREC_F32;
      h->conv_in_x32 = P.pack(P.conv_rows(w, 0, lat, CP), c0, 3 * CP, {}, false, CP, 0);
      h->conv_in_c32 = P.pack(P.conv_rows(w, lat, lat + c.content_channels, c.content_channels), c0, 3 * c.content_channels, P.T("conv_in.bias").data);
      P.prec_override = -1;
    }
  }
  // time MLP, transposed to [in][out] for coalesced GEMV reads
  auto transpose = [&](const HostTensor& w) {
    const int N = (int)w.shape[0], K = (int)w.shape[1];
    std::vector<float> t((size_t)N * K);
    for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) t[(size_t)k * N + n] = w.data[(size_t)n * K + k];
    return t;
  };
  h->t_w1t = P.upload_f32(transpose(P.T("time_embedding.linear_1.weight"))); h->t_b1 = P.vec("time_embedding.linear_1.bias");
  h->t_w2t = P.upload_f32(transpose(P.T("time_embedding.linear_2.weight"))); h->t_b2 = P.vec("time_embedding.linear_2.bias");
  // add_embedding
  h->p_n1g = P.vec("add_embedding.norm1.weight"); h->p_n1b = P.vec("add_embedding.norm1.bias");
  h->p_pos = P.vec("add_embedding.pool.positional_embedding");
  {
    std::vector<float> rows, bias;
    for (const char* p : {"q_proj", "k_proj", "v_proj"}) {
      const HostTensor& w = P.T(std::string("add_embedding.pool.") + p + ".weight");
      const HostTensor& b = P.T(std::string("add_embedding.pool.") + p + ".bias");
      if (P.err) return 1;
      rows.insert(rows.end(), w.data.begin(), w.data.end());
      bias.insert(bias.end(), b.data.begin(), b.data.end());
    }
    h->pool_qkv = P.pack(rows, 3 * cross, cross, bias);
  }
  h->p_projT = P.upload_f32(transpose(P.T("add_embedding.proj.weight"))); h->p_projb = P.vec("add_embedding.proj.bias");
  h->p_n2g = P.vec("add_embedding.norm2.weight"); h->p_n2b = P.vec("add_embedding.norm2.bias");
  if (P.err) return 1;

  std::vector<float> temb_rows, temb_bias, kv_rows;
  int temb_off = 0, kv_off = 0;
  for (auto& b : h->blocks) {
    for (auto& r : b.res) {
      r.n1g = P.vec(r.prefix + ".norm1.weight"); r.n1b = P.vec(r.prefix + ".norm1.bias");
      r.n2g = P.vec(r.prefix + ".norm2.weight"); r.n2b = P.vec(r.prefix + ".norm2.bias");
      r.conv1 = P.conv(r.prefix + ".conv1", true);
      if (r.shortcut) {   // conv2 and the 1x1 shortcut share one GEMM: K = 3*cout + cin, biases summed
        const HostTensor& w2 = P.T(r.prefix + ".conv2.weight");
        const HostTensor& ws = P.T(r.prefix + ".conv_shortcut.weight");
        const HostTensor& b2 = P.T(r.prefix + ".conv2.bias");
        const HostTensor& bs = P.T(r.prefix + ".conv_shortcut.bias");
        if (P.err) return 1;
        const std::vector<float> r2 = P.conv_rows(w2, 0, r.cout, r.cout);
        const int K1 = 3 * r.cout, K2 = r.cin;
        std::vector<float> rows((size_t)r.cout * (K1 + K2)), bias(r.cout);
        for (int n = 0; n < r.cout; ++n) {
          memcpy(&rows[(size_t)n * (K1 + K2)], &r2[(size_t)n * K1], K1 * sizeof(float));
          memcpy(&rows[(size_t)n * (K1 + K2) + K1], &ws.data[(size_t)n * K2], K2 * sizeof(float));
          bias[n] = b2.data[n] + bs.data[n];
        }
        r.conv2 = P.pack(rows, r.cout, K1 + K2, bias, false, r.cout, K2);
      } else {
        r.conv2 = P.conv(r.prefix + ".conv2", true);
      }
      const HostTensor& tw = P.T(r.prefix + ".time_emb_proj.weight");
      const HostTensor& tb = P.T(r.prefix + ".time_emb_proj.bias");
      if (P.err) return 1;
      r.temb_off = temb_off;
      temb_rows.insert(temb_rows.end(), tw.data.begin(), tw.data.end());
      temb_bias.insert(temb_bias.end(), tb.data.begin(), tb.data.end());
      temb_off += 2 * r.cout;
    }
    for (auto& a : b.attn) {
      const std::string t = a.prefix + ".transformer_blocks.0";
      const int d = a.dim;
      a.ng = P.vec(a.prefix + ".norm.weight"); a.nb = P.vec(a.prefix + ".norm.bias");
      a.proj_in = P.conv(a.prefix + ".proj_in");
      a.proj_out = P.conv(a.prefix + ".proj_out");
      {  // fused q|k|v of the self-attention, LayerNorm(norm1) folded in
        std::vector<float> rows, bias;
        for (const char* nm : {"to_q", "to_k", "to_v"}) P.ln_fold(P.T(t + ".attn1." + nm + ".weight"), nullptr, P.T(t + ".norm1.weight"), P.T(t + ".norm1.bias"), rows, bias);
        if (P.err) return 1;
        a.qkv = P.pack(rows, 3 * d, d, bias, true);
        if (chain_stream(P, P.T(a.prefix + ".proj_in.weight").data, rows, bias, d, 3 * d, a.chain_in, a.chain_in_consts, &a.chain_in_s2)) return 1;
      }
      a.o1 = P.pack(P.T(t + ".attn1.to_out.0.weight").data, d, d, P.T(t + ".attn1.to_out.0.bias").data);
      {
        std::vector<float> rows, bias;
        P.ln_fold(P.T(t + ".attn2.to_q.weight"), nullptr, P.T(t + ".norm2.weight"), P.T(t + ".norm2.bias"), rows, bias);
        if (P.err) return 1;
        a.q2 = P.pack(rows, d, d, bias, true);
        if (chain_stream(P, P.T(t + ".attn1.to_out.0.weight").data, rows, bias, d, d, a.chain_mid, a.chain_mid_consts)) return 1;
      }
      a.o2 = P.pack(P.T(t + ".attn2.to_out.0.weight").data, d, d, P.T(t + ".attn2.to_out.0.bias").data);
      std::vector<float> ff1_rows, ff1_bias;     // packed ff.net.0 (kept for the fused feed-forward stream below)
      {  // GEGLU projection: LayerNorm(norm3) folded, rows interleaved in (32 value | 32 gate) groups
        std::vector<float> rows, bias;
        P.ln_fold(P.T(t + ".ff.net.0.proj.weight"), &P.T(t + ".ff.net.0.proj.bias"), P.T(t + ".norm3.weight"), P.T(t + ".norm3.bias"), rows, bias);
        if (P.err) return 1;
        const int inner = 4 * d;
        std::vector<float> rows2((size_t)8 * d * d), bias2(8 * d);
        for (int gidx = 0; gidx < inner / 32; ++gidx)
          for (int i = 0; i < 32; ++i) {
            const int v_src = 32 * gidx + i, g_src = inner + 32 * gidx + i;
            const int v_dst = 64 * gidx + i, g_dst = 64 * gidx + 32 + i;
            memcpy(&rows2[(size_t)v_dst * d], &rows[(size_t)v_src * d], d * sizeof(float));
            memcpy(&rows2[(size_t)g_dst * d], &rows[(size_t)g_src * d], d * sizeof(float));
            bias2[v_dst] = bias[v_src];
            bias2[g_dst] = bias[g_src];
          }
        a.ff1 = P.pack(rows2, 8 * d, d, bias2, true);
        ff1_rows.swap(rows2); ff1_bias.swap(bias2);
      }
      a.ff2 = P.pack(P.T(t + ".ff.net.2.weight").data, d, 4 * d, P.T(t + ".ff.net.2.bias").data);
      {  // ff.net.2 folded into proj_out (attention.py:178-203 + transformer_1d.py:287-295):
         //   proj_out(y + W2 g + b2) + x = [Wpo W2 | Wpo] [g | y] + (Wpo b2 + bpo) + x
         // one GEMM, K = 4d (GEGLU output) + d (second K segment: the operand copy of y).  Products in double.
        const HostTensor& w2 = P.T(t + ".ff.net.2.weight");      // [d][4d]
        const HostTensor& b2 = P.T(t + ".ff.net.2.bias");
        const HostTensor& wp = P.T(a.prefix + ".proj_out.weight"); // [d][d][1]
        const HostTensor& bp = P.T(a.prefix + ".proj_out.bias");
        if (P.err) return 1;
        const int K1 = 4 * d, K2 = d;
        std::vector<float> rows((size_t)d * (K1 + K2)), bias(d);
        std::vector<double> acc(K1);
        for (int n = 0; n < d; ++n) {
          std::fill(acc.begin(), acc.end(), 0.0);
          double bacc = (double)bp.data[n];
          for (int j = 0; j < d; ++j) {
            const double wpj = (double)wp.data[(size_t)n * d + j];
            const float* w2r = &w2.data[(size_t)j * K1];
            for (int k = 0; k < K1; ++k) acc[k] += wpj * (double)w2r[k];
            bacc += wpj * (double)b2.data[j];
          }
          float* r = &rows[(size_t)n * (K1 + K2)];
          for (int k = 0; k < K1; ++k) r[k] = (float)acc[k];
          memcpy(r + K1, &wp.data[(size_t)n * d], (size_t)K2 * sizeof(float));
          bias[n] = (float)bacc;
        }
        a.ffpo = P.pack(rows, d, K1 + K2, bias);
        if (h->prec != PREC_F32 && (d == 128 || d == 256)) {   // fused feed-forward + proj_out (csrc/ffn.hip)
          std::vector<unsigned short> st;
          for (int pre = 0; pre < 2; ++pre) {
            const float* w0 = pre ? P.T(t + ".attn2.to_out.0.weight").data.data() : nullptr;
            if (P.err) return 1;
            if (pack_ffn_stream(ff1_rows.data(), rows.data(), w0, d, h->prec, st) != hipSuccess) return fail("ffn stream packing failed");
            void* dev = nullptr;
            if (hipMalloc(&dev, st.size() * 2) != hipSuccess) return fail("hipMalloc failed (weights)");
            h->weight_allocs.push_back(dev);
            if (hipMemcpy(dev, st.data(), st.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy failed (weights)");
            (pre ? a.ffn_pre_stream : a.ffn_stream) = dev;
          }
          const std::vector<float> ws = rounded_rowsum(ff1_rows.data(), 8 * d, d, 8 * d, h->prec);
          std::vector<float> cs((size_t)8 * d * 2);
          for (int r = 0; r < 8 * d; ++r) { cs[2 * r] = ws[r]; cs[2 * r + 1] = ff1_bias[r]; }
          a.ffn_consts = P.upload_f32(cs);
        } else {
          a.ffn_stream = nullptr; a.ffn_pre_stream = nullptr; a.ffn_consts = nullptr;
        }
        if (geglu_eligible(d, 1, h->prec)) {                   // token-stationary GEGLU projection (csrc/geglu.hip)
          std::vector<unsigned short> st;
          std::vector<float> cs;
          if (pack_geglu_stream(ff1_rows.data(), ff1_bias.data(), d, h->prec, st, cs) != hipSuccess) return fail("geglu stream packing failed");
          void* dev = nullptr;
          if (hipMalloc(&dev, st.size() * 2) != hipSuccess) return fail("hipMalloc failed (weights)");
          h->weight_allocs.push_back(dev);
          if (hipMemcpy(dev, st.data(), st.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy failed (weights)");
          a.geglu_stream = dev;
          a.geglu_consts = P.upload_f32(cs);
        }
      }
      {  // cross-attention k|v of this block into the hoisted all-blocks projection
        const HostTensor& wk = P.T(t + ".attn2.to_k.weight");
        const HostTensor& wv = P.T(t + ".attn2.to_v.weight");
        if (P.err) return 1;
        a.kv_off = kv_off;
        kv_rows.insert(kv_rows.end(), wk.data.begin(), wk.data.end());
        kv_rows.insert(kv_rows.end(), wv.data.begin(), wv.data.end());
        kv_off += 2 * d;
      }
    }
    if (b.sampler) {
      const std::string p = b.kind == "down" ? "down_blocks." + std::to_string(b.index) + ".downsamplers.0.conv"
                                             : "up_blocks." + std::to_string(b.index) + ".upsamplers.0.conv";
      b.samp = P.conv(p);
    }
    if (P.err) return 1;
  }
  h->n_temb = temb_off;
  h->temb_all = P.pack(temb_rows, temb_off, temb, temb_bias);
  if (h->prec != PREC_F32) { P.prec_override = PREC_F32; h->temb_all32 = P.pack(temb_rows, temb_off, temb, temb_bias); P.prec_override = -1; }
  h->n_kv = kv_off;
  h->kv_all = P.pack(kv_rows, kv_off, cross, {});
  h->out_ng = P.vec("conv_norm_out.weight"); h->out_nb = P.vec("conv_norm_out.bias");
  {
    const HostTensor& w = P.T("conv_out.weight");
    if (P.err) return 1;
    h->conv_out = P.pack(P.conv_rows(w, 0, c0, c0), lat, 3 * c0, P.T("conv_out.bias").data, false, c0, 0);
    if (h->prec != PREC_F32) h->conv_outp = P.pack(P.pair_rows(P.conv_rows(w, 0, c0, c0), lat, 3, c0), lat, 9 * c0, P.T("conv_out.bias").data, false, 3 * c0, 0);
    if (h->prec != PREC_F32) { P.prec_override = PREC_F32; h->conv_out32 = P.pack(P.conv_rows(w, 0, c0, c0), lat, 3 * c0, P.T("conv_out.bias").data, false, c0, 0); P.prec_override = -1; }
  }
  return P.err;
}

// ------------------------------------------------------------------------------------
// plan building.  Two kinds of activation tensors:
//   fp32  "stream" tensors : residual stream, skips, GroupNorm inputs (statistics stay fp32)
//   "op"  operand tensors  : what GEMMs / attention read — bf16 (perf) or fp32 (parity)
// ------------------------------------------------------------------------------------
struct Planner {
  ns2vc_unet* h;
  std::vector<Op>* ops;
  bool sizing = false;       // first pass: only measure the arena
  size_t off = 0;
  int B, T, Lp, G, prec;
  size_t opsz = 2;
  // scratch shared by all layers (stream-ordered)
  double* gn_partial = nullptr;
  void *xn = nullptr, *xr = nullptr;     // GroupNorm-applied / raw operand copies of a resnet input
  float *rs1 = nullptr, *rs2 = nullptr, *rs3 = nullptr;   // LayerNorm-by-linearity row statistics [M][C/64][2] (norm1/2/3)
  int gn_rows = 64;
  // GroupNorm statistics accumulated by the producing GEMM's epilogue (int64 fixed point, [B][C/16][2]);
  // one zeroed slab per produced tensor, all carved from stats_pool (cleared by one memset per forward)
  long long* stats_pool = nullptr;
  size_t stats_cap = 0, stats_used = 0;
  std::map<const void*, long long*> stats_of;
  long long* new_stats(const float* tensor, int Tl, int C) {
    if (Tl < 64 || (C & 15)) { stats_of.erase(tensor); return nullptr; }
    const size_t n = (size_t)B * (C / 16) * 2;
    if (stats_used + n > stats_cap) { stats_of.erase(tensor); return nullptr; }
    long long* p = stats_pool ? stats_pool + stats_used : reinterpret_cast<long long*>(sizeof(long long) * (stats_used + 1));  // sizing pass: non-null token
    stats_used += n;
    stats_of[tensor] = p;
    return p;
  }
  unsigned* new_sync(size_t nblocks) {       // 64-bit arrival words of a cooperative GroupNorm prologue, one per row block
    nblocks = (nblocks + 1) & ~(size_t)1;     // (whole 16-byte units, 16-byte aligned: Op::rearm zeroes them with 16-byte stores)
    stats_used = (stats_used + 1) & ~(size_t)1;
    if (stats_used + nblocks > stats_cap) return nullptr;
    long long* p = stats_pool ? stats_pool + stats_used : reinterpret_cast<long long*>(sizeof(long long) * (stats_used + 1));
    stats_used += nblocks;
    return reinterpret_cast<unsigned*>(p);
  }
  long long* find_stats(const float* tensor) const {
    auto it = stats_of.find(tensor);
    return it == stats_of.end() ? nullptr : it->second;
  }

  char* alloc_bytes(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    char* p = sizing ? nullptr : reinterpret_cast<char*>(h->arena) + off;
    off += bytes;
    return p;
  }
  template <typename Tp> Tp* alloc(size_t count) { return reinterpret_cast<Tp*>(alloc_bytes(count * sizeof(Tp))); }
  void* alloc_op(size_t count) { return alloc_bytes(count * opsz); }
  void* op_off(void* p, size_t elems) const { return p ? static_cast<void*>(static_cast<char*>(p) + elems * opsz) : nullptr; }

  void add(const std::string& name, std::function<hipError_t(hipStream_t)> fn, int kind = 0, double flops = 0.0, double bytes = 0.0) {
    if (sizing) return;
    Op op;
    op.name = name; op.fn = std::move(fn); op.kind = kind; op.flops = flops; op.bytes = bytes;
    ops->push_back(std::move(op));
  }
  void tap(const std::string& name, const float* src, int rows, int cols) {
    if (!h->debug) return;
    float* cp = alloc<float>((size_t)rows * cols);
    if (sizing) return;
    h->taps.push_back({name, cp, rows, cols});
    const size_t bytes = (size_t)rows * cols * sizeof(float);
    add("tap:" + name, [=](hipStream_t s) { return hipMemcpyAsync(cp, src, bytes, hipMemcpyDeviceToDevice, s); }, 4, 0.0, 2.0 * bytes);
  }

  void gemm(const std::string& name, GemmArgs g, int pr_override = -1) {
    const int pr = pr_override >= 0 ? pr_override : prec;
    const double osz = (double)opsz;
    const double nout = g.geglu ? g.N / 2 : g.N;
    // the ALGORITHMIC figures of the op (what the roofline fractions are priced on): a launch on hi + lo operand pairs (split_io: [hi | lo] + hi again against
    // (hi(w) | hi(w) | lo(w)), three times the K) counts as the plain convolution it computes, not as the MFMA work and bytes it spends on it
    const bool pair = g.c1 && g.a1 == g.a0 && g.c0 == 2 * g.c1 && g.c2 == 0;
    const double Kalg = pair ? g.K / 3.0 : (double)g.K, cin = pair ? g.c1 : g.c0 + g.c1 + g.c2, cgn = pair ? g.c1 : g.c0;
    const double flops = 2.0 * g.M * (double)g.N * Kalg;
    const double in_rows = (double)g.B * g.Tin;
    const double bytes = in_rows * cin * osz + (double)g.N * Kalg * osz + (g.out_f32 ? g.M * nout * 4.0 : 0.0) +
                         (g.out_op ? g.M * nout * osz : 0.0) + (g.res ? g.M * nout * 4.0 : 0.0);
    // (a GroupNorm prologue reads the fp32 rows and writes + re-reads the operand rows it builds)
    const double pro = g.gnp_x ? in_rows * cgn * (4.0 + osz * (g.gnp_raw ? 2.0 : 1.0)) : 0.0;
    if (g.taps == 3 && g.tmode == TMODE_SAME && !g.conv_bn) g.conv_bn = convts_bn_for(g, h->bn128_min);     // the column tile is a PLAN decision (this engine's device)
    if (!sizing && g.gnp_temb && ops == &h->fwd_ops && h->temb_join < 0) h->temb_join = (int)ops->size();    // first launch that reads the time scale / shift rows
    add(g.gnp_x ? name + "[+norm]" : name, [=](hipStream_t s) { return launch_gemm(g, pr, s); }, 1, flops, bytes + pro);
    if (!sizing && g.gnp_x && g.gnp_sync) {
      unsigned* words = g.gnp_sync;
      const size_t nbytes = (((size_t)g.B * g.Tin + 63) / 64) * 8;
      ops->back().rearm = [=](hipStream_t s) { return launch_zero(words, (nbytes + 15) & ~(size_t)15, s); };
    }
  }
  // A = operand tensor [B*Tin][c0]; results to out_f32 and/or out_op (row stride = logical width)
  GemmArgs base(const void* a0, int lda0, int c0, int Tin, int Tout, const PackedW& w, float* out_f32, void* out_op, int ldo) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a0 = a0; g.lda0 = lda0; g.c0 = c0;
    g.B = B; g.Tin = Tin; g.Tout = Tout; g.M = B * Tout;
    g.taps = 1; g.tmode = TMODE_SAME;
    g.w = w.w; g.K = w.K; g.N = w.N; g.bias = w.bias;
    g.w_tiled = h->conv_wtiled ? w.wt : nullptr;     // (only the k = 3 / stride-1 launches of the tap-sharing kernel look at it)
    g.out_f32 = out_f32; g.ldo_f32 = ldo;
    g.out_op = out_op; g.ldo_op = ldo;
    g.algo = h->conv_ts ? (h->gn_inloop ? 0 : 2) : 1;
    return g;
  }
  // GroupNorm of a (possibly concatenated) fp32 input: statistics -> per-(b,c) affine -> operand tensor `dst`
  // (= act(GN(x)) with the concat materialised), optionally also the raw concat `raw` for a 1x1 shortcut.
  // `consumer_n` > 0: `dst` has exactly one reader, a GEMM with that many output columns that is planned next -- where the
  // norm qualifies (see fuse_gn_gemm) no launch is added and the returned GnPro is handed to that GEMM with gn_fuse().
  struct GnPro { const float* x = nullptr; int ldx = 0; const long long* st = nullptr; const float* gamma = nullptr; const float* beta = nullptr;
                 const float* temb = nullptr; int ldtemb = 0; float eps = 0.f; int G = 0, silu = 0; unsigned* sync = nullptr; unsigned* alone = nullptr;
                 const float* x1 = nullptr; int ldx1 = 0, c1 = 0; const long long* st1 = nullptr; void* raw = nullptr; };
  static void gn_fuse(GemmArgs& g, const GnPro& p) {
    if (!p.x) return;
    g.gnp_x = p.x; g.gnp_ldx = p.ldx; g.gnp_stats = p.st; g.gnp_gamma = p.gamma; g.gnp_beta = p.beta;
    g.gnp_temb = p.temb; g.gnp_ldtemb = p.ldtemb; g.gnp_eps = p.eps; g.gnp_G = p.G; g.gnp_silu = p.silu;
    g.gnp_sync = p.sync; g.gnp_alone = p.alone;
    g.gnp_x1 = p.x1; g.gnp_ldx1 = p.ldx1; g.gnp_c1 = p.c1; g.gnp_stats1 = p.st1; g.gnp_raw = p.raw;
  }
  GnPro groupnorm(const std::string& name, const float* a0, int lda0, int c0, const float* a1, int lda1, int c1, int Tl, float eps,
                  const float* gamma, const float* beta, const float* temb, int temb_off, int cout, int silu, void* dst, void* raw,
                  int consumer_n = 0, int consumer_taps = 1, int pair = 0) {
    (void)cout;
    const int nchunk = (Tl + gn_rows - 1) / gn_rows, rows = gn_rows, Bq = B, Gq = G, ldt = h->temb_all.N, pr = prec;
    double* part = gn_partial;
    const double n = (double)Bq * Tl * (c0 + c1);
    const long long* st0 = find_stats(a0);
    const long long* st1 = a1 ? find_stats(a1) : nullptr;
    const bool epi = st0 && (!a1 || st1) && (((c0 + c1) / Gq) % 16 == 0) && (c0 % 16 == 0);
    // (pair: the prologue that writes hi + lo pairs exists in the tap-sharing conv kernel only)
    if (epi && h->fuse_gn_gemm && consumer_n > 0 && (consumer_n % 128) == 0 && (h->fuse_gn_cat || (!a1 && !raw)) && Tl >= 66 && c0 + c1 <= 1024 &&
        (!pair || (h->conv_ts && consumer_taps == 3 && ((c0 + c1) % 64) == 0)) &&
        ((c0 + c1) % Gq) == 0 && Gq <= 8 && (lda0 & 3) == 0 && (!a1 || ((lda1 & 3) == 0 && (c1 & 15) == 0))) {
      GnPro p;
      p.x = a0; p.ldx = lda0; p.st = st0;
      if (a1) { p.x1 = a1; p.ldx1 = lda1; p.c1 = c1; p.st1 = st1; }      // a concat of two sources (up blocks), normalised as one tensor
      p.raw = raw;                                                         // ... and the un-normalised operand copy for the 1x1 shortcut
      p.gamma = gamma; p.beta = beta; p.temb = temb ? temb + temb_off : nullptr; p.ldtemb = ldt;
      p.eps = eps; p.G = Gq; p.silu = silu;
      // wider than one column tile: the column tiles of a row block share the prologue's rows (one 64-bit count per 64-row block, zeroed with the arena)
      // (r5: the counts live in the statistics pool, so the forward's one clear launch also zeroes them: a launch that was cut short cannot
      //  leave a remainder behind for the next forward)
      int nshare = consumer_n / 128;           // column tiles of a row block: N / 128 in gemm4_kernel, N / BN in the tap-sharing conv kernel
      if (consumer_taps == 3 && h->conv_ts && Tl >= 66) {
        GemmArgs t;
        memset(&t, 0, sizeof(t));
        t.B = Bq; t.Tin = t.Tout = Tl; t.N = consumer_n;
        nshare = consumer_n / convts_bn_for(t, h->bn128_min);
      }
      if (h->gn_coop && nshare >= std::max(2, h->gn_coop_min)) { p.sync = new_sync(((size_t)Bq * Tl + 63) / 64); p.alone = (p.sync && h->ln_health) ? h->ln_health + 48 : nullptr; }
      return p;
    }
    if (!epi) {
      st0 = st1 = nullptr;
      add(name + ".gn_stats", [=](hipStream_t s) { return launch_gn_partial(a0, lda0, c0, a1, lda1, c1, Bq, Tl, Gq, part, nchunk, rows, s); },
          3, 3.0 * n, 4.0 * n);
    }
    add(name + ".gn_apply", [=](hipStream_t s) {
      return launch_gn_apply(a0, lda0, c0, a1, lda1, c1, Bq, Tl, Gq, eps, part, nchunk, st0, st1, gamma, beta, temb, ldt, temb_off, silu, dst,
                             raw, pr, s, pair);                  // (pair: the rows as a hi + lo operand pair, split_io's conv_out)
    }, 3, 4.0 * n, n * (4.0 + opsz * (raw ? 2.0 : 1.0) + opsz * (pair ? 1.0 : 0.0)));
    return GnPro();
  }

  // ResnetBlock2D (resnet.py:591-641).  out (fp32) [+ out_op operand copy when a conv consumes it next]
  void resnet(const ResnetW& r, const float* a0, int lda0, int c0, const float* a1, int lda1, int c1, int Tl, float* h1, void* hn,
              float* out, void* out_op) {
    const int cin = c0 + c1;
    // ---- conv1(act(norm1(x)))
    const GnPro p1 = groupnorm(r.prefix + ".norm1", a0, lda0, c0, a1, lda1, c1, Tl, 1e-5f, r.n1g, r.n1b, nullptr, 0, 0, 1, xn,
                               r.shortcut ? xr : nullptr, r.conv1.N, 3);
    GemmArgs g = base(xn, cin, cin, Tl, Tl, r.conv1, h1, nullptr, r.cout);
    g.taps = 3;
    gn_fuse(g, p1);
    g.stats = new_stats(h1, Tl, r.cout);
    gemm(r.prefix + ".conv1", g);
    // ---- conv2(act(norm2(h) * (1 + scale) + shift)) + shortcut
    const GnPro p2 = groupnorm(r.prefix + ".norm2", h1, r.cout, r.cout, nullptr, 0, 0, Tl, 1e-5f, r.n2g, r.n2b, h->temb, r.temb_off, r.cout, 1, hn,
                               nullptr, r.conv2.N, 3);
    GemmArgs g2 = base(hn, r.cout, r.cout, Tl, Tl, r.conv2, out, out_op, r.cout);
    g2.taps = 3;
    gn_fuse(g2, p2);
    if (r.shortcut) {      // out = conv2(hn) + conv_shortcut(x): the 1x1 conv rides along as a second K segment
      g2.a2 = xr; g2.lda2 = cin; g2.c2 = cin;
    } else {
      g2.res = a0; g2.ldres = lda0;
    }
    g2.stats = new_stats(out, Tl, r.cout);
    gemm(r.prefix + ".conv2", g2);
  }

  void attention(const std::string& name, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int Lq, int Lk,
                 const float* bias, int hd, void* out, int ldo) {
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
    a.B = B; a.H = h->cfg.heads; a.Lq = Lq; a.Lk = Lk; a.bias = bias;
    a.scale = 1.0f / std::sqrt((float)hd);
    a.out = out; a.ldo = ldo;
    a.pv_fp8 = (h->attn_fp8 && prec != PREC_F32) ? 1 : 0;
    a.exact_only = h->attn_optimistic ? 0 : 1;
    a.fallbacks = h->attn_fallbacks;
    const int pr = prec;
    add(name, [=](hipStream_t s) { return launch_attention(a, hd, pr, s); }, 2, 4.0 * B * a.H * (double)Lq * Lk * hd,
        (double)opsz * B * a.H * hd * (2.0 * Lq + 2.0 * Lk));
  }

  // r6: does this block run its prompt cross-attention inside the fused feed-forward kernel?  (the plan of the pre-stage form, 8 heads of 16 / 32 channels)
  std::map<std::string, void*> xattn_vt;      // per transformer block: the k | v fragment image of its hoisted rows (built by the condition plan)
  bool xattn_fused(const AttnW& a, int Tl) const {
    const int d = a.dim;
    const bool lin = h->ln_linear && (d % 128 == 0) && d <= 512;
    return h->fuse_xattn && lin && h->fold_ff && h->fuse_ffn && a.ffn_stream && ffn_eligible(d, Tl, prec) && h->fuse_ffn_pre && a.ffn_pre_stream &&
           h->cfg.heads == 8 && (d == 128 || d == 256);
  }
  // Transformer2DModel + BasicTransformerBlock (transformer_1d.py:256-295, attention.py:130-203)
  void transformer(const AttnW& a, const float* x, int Tl, float* y, void* yn, void* qkv, void* ao, void* qb, void* ffh, float* out,
                   void* out_op) {
    const int d = a.dim, M = B * Tl, hd = d / h->cfg.heads, pr = prec;
    const std::string t = a.prefix + ".transformer_blocks.0";
    GemmArgs g;
    auto layernorm = [&](const std::string& nm) {
      add(nm, [=](hipStream_t s) { return launch_ln_apply_op(y, d, M, d, 1e-5f, yn, pr, s); }, 3, 8.0 * M * d, (4.0 + opsz) * M * d);
    };
    // LayerNorm by linearity (h->ln_linear): the producer of every LayerNorm input also writes the raw operand copy
    // `yn` and per-row statistics; the consumer GEMM reads yn and normalises in its epilogue -- no ln_apply pass
    const bool lin = h->ln_linear && (d % 128 == 0) && d <= 512;
    auto consume = [&](GemmArgs& gg, float* rs, const PackedW& w) {
      if (rs) { gg.ln_stats = rs; gg.ln_wsum = w.wsum; gg.ln_eps = 1e-5f; gg.ln_dim = d; gg.ln_health = h->ln_health; }
    };
    float* r1 = lin ? rs1 : nullptr;
    // token-local chains in one launch each (rowchain.hip): same arithmetic and rounding points as the two GEMMs they replace
    auto rowchain = [&](const std::string& nm, const void* a_op, const long long* gn_st, void* stream, const float* bias1, const float* consts2,
                        const float* res, void* z_op, int n2) {
      ns2vc_rowchain_args c;
      memset(&c, 0, sizeof(c));
      c.a_op = a_op; c.lda = d; c.wstream = stream; c.bias1 = bias1; c.consts2 = consts2;
      c.res = res; c.ldres = d; c.out1_f32 = y; c.ldo1 = d; c.out2_op = z_op; c.ldo2 = n2;
      c.ln_eps = 1e-5f; c.M = M; c.dim = d; c.n2 = n2; c.ln_health = h->ln_health;
      if (gn_st) {       // A = GroupNorm(x) built in the kernel's prologue from the producer's epilogue statistics
        c.a_op = nullptr; c.gn_x = x; c.ldx = d; c.gn_stats = gn_st; c.gn_gamma = a.ng; c.gn_beta = a.nb; c.gn_eps = 1e-6f; c.T = Tl; c.G = G;
      }
      // r4: two N-slices per token block where that still is one round of workgroups (dim 384 at the bench batch: 118 blocks on 256 CUs);
      // only for the chain without a residual (the second chain reads and rewrites y in place: two slices would race on it)
      if (!res && stream == a.chain_in && a.chain_in_s2 && h->slice_rows && 2 * ((M + 63) / 64) <= h->cus + 8) { c.wstream = a.chain_in_s2; c.slices = 2; }   // (one round of workgroups on this device's CUs)
      add(c.slices == 2 ? nm + "[2 slices]" : nm, [=](hipStream_t s) { return launch_rowchain(c, pr, s); }, 1, 2.0 * M * (double)d * (d + n2),
          (double)M * (d * ((gn_st ? 4.0 : opsz) + 4.0 + (res ? 4.0 : 0.0)) + n2 * opsz) + (double)(d + n2) * d * opsz);
    };
    const bool rows_ok = lin && h->fuse_rows && a.chain_in && a.chain_mid && rowchain_eligible(d, d, Tl, pr);
    const long long* xst = (rows_ok && h->fuse_rows_gn && Tl >= 64 && (d % G) == 0 && ((d / G) % 16) == 0) ? find_stats(x) : nullptr;
    GnPro pn;
    if (!xst) pn = groupnorm(a.prefix + ".norm", x, d, d, nullptr, 0, 0, Tl, 1e-6f, a.ng, a.nb, nullptr, 0, 0, 0, xn, nullptr, rows_ok ? 0 : a.proj_in.N);
    if (rows_ok) {
      rowchain(a.prefix + (xst ? ".rows[norm+proj_in+qkv]" : ".rows[proj_in+qkv]"), xn, xst, a.chain_in, a.proj_in.bias, a.chain_in_consts, nullptr, qkv,
               3 * d);
    } else {
      g = base(xn, d, d, Tl, Tl, a.proj_in, y, r1 ? yn : nullptr, d);
      gn_fuse(g, pn);
      g.rowstats = r1;
      gemm(a.prefix + ".proj_in", g);
      // self attention
      if (!r1) layernorm(t + ".norm1");
      g = base(yn, d, d, Tl, Tl, a.qkv, nullptr, qkv, 3 * d);
      consume(g, r1, a.qkv);
      gemm(t + ".attn1.qkv", g);
    }
    attention(t + ".attn1.sdpa", qkv, 3 * d, op_off(qkv, d), 3 * d, op_off(qkv, 2 * d), 3 * d, Tl, Tl, nullptr, hd, ao, d);
    float* r2 = lin ? rs2 : nullptr;
    if (rows_ok) {
      rowchain(t + ".rows[attn1.to_out+attn2.to_q]", ao, nullptr, a.chain_mid, a.o1.bias, a.chain_mid_consts, y, qb, d);
    } else {
      g = base(ao, d, d, Tl, Tl, a.o1, y, r2 ? yn : nullptr, d);
      g.res = y; g.ldres = d;
      g.rowstats = r2;
      gemm(t + ".attn1.to_out", g);
      if (!r2) layernorm(t + ".norm2");
      g = base(yn, d, d, Tl, Tl, a.q2, nullptr, qb, d);
      consume(g, r2, a.q2);
      gemm(t + ".attn2.to_q", g);
    }
    // cross attention (k|v hoisted into h->kv by set_condition)
    const int nkv = h->kv_all.N;
    const bool xatt = xattn_fused(a, Tl) && xattn_vt.count(a.prefix);
    if (!xatt)
      attention(t + ".attn2.sdpa", qb, d, op_off(h->kv, a.kv_off), nkv, op_off(h->kv, a.kv_off + d), nkv, Tl, Lp,
                h->has_mask ? h->maskbias : nullptr, hd, ao, d);
    float* r3 = lin ? rs3 : nullptr;
    // With the feed-forward output folded into proj_out, proj_out reads the RAW operand copy of y next to the GEGLU
    // output.  LayerNorm by linearity writes that copy anyway (yn); the explicit-LayerNorm plan overwrites yn with the
    // normalised rows, so there the raw copy goes to qb (the cross-attention query buffer, free by now).
    const bool fold = h->fold_ff;
    void* yraw = r3 ? yn : (fold ? qb : nullptr);
    const bool ffn_ok = fold && r3 && h->fuse_ffn && a.ffn_stream && ffn_eligible(d, Tl, pr);
    // attn2.to_out + residual as the pre-stage of the fused feed-forward kernel: y after the cross-attention is never stored
    const bool ffn_pre = ffn_ok && h->fuse_ffn_pre && a.ffn_pre_stream;
    if (!ffn_pre) {
      g = base(ao, d, d, Tl, Tl, a.o2, y, yraw, d);
      g.res = y; g.ldres = d;
      g.rowstats = r3;
      gemm(t + ".attn2.to_out", g);
    }
    // feed-forward (GEGLU)
    if (ffn_ok) {
      // LayerNorm(norm3) -> GEGLU -> ff.net.2 -> + y -> proj_out -> + x in ONE launch: the hidden tensor never exists
      ns2vc_ffn_args f;
      memset(&f, 0, sizeof(f));
      f.yn = yn; f.ldy = d; f.ln_stats = r3; f.ln_eps = 1e-5f;
      f.wstream = a.ffn_stream; f.consts = a.ffn_consts; f.bias2 = a.ffpo.bias;
      if (ffn_pre) {
        f.yn = nullptr; f.ln_stats = nullptr; f.wstream = a.ffn_pre_stream;
        f.pre_a = ao; f.pre_lda = d; f.pre_bias = a.o2.bias; f.pre_res = y; f.pre_ldres = d;
      }
      if (xatt) {         // (implies ffn_pre) the cross-attention's output never exists: the kernel builds its token panel from q, the hoisted k rows and V^T
        f.pre_a = nullptr;
        f.att_q = qb; f.att_ldq = d; f.att_kv = xattn_vt[a.prefix];
        f.att_bias = h->has_mask ? h->maskbias : nullptr; f.att_scale = 1.0f / std::sqrt((float)hd); f.att_Lk = Lp;
      }
      f.res = x; f.ldres = d;
      f.out_f32 = out; f.ldo_f32 = d; f.out_op = out_op; f.ldo_op = d;
      f.stats = new_stats(out, Tl, d);
      f.B = B; f.T = Tl; f.M = M; f.dim = d; f.ln_health = h->ln_health;
      const double fl = 2.0 * M * (double)d * ((ffn_pre ? 14.0 : 13.0) * d) + (xatt ? 4.0 * B * h->cfg.heads * (double)Tl * Lp * hd : 0.0);
      add(a.prefix + (xatt ? ".ffn[attn2.sdpa+to_out+geglu+ff.out+proj_out]" : ffn_pre ? ".ffn[attn2.to_out+geglu+ff.out+proj_out]" : ".ffn[geglu+ff.out+proj_out]"),
          [=](hipStream_t s) { return launch_ffn(f, pr, s); }, 1, fl,
          (double)M * d * (opsz + 8.0 + (ffn_pre ? 4.0 : 0.0) + (out_op ? opsz : 0.0)) + (ffn_pre ? 14.0 : 13.0) * d * d * opsz +
              (xatt ? (double)opsz * B * d * 2.0 * Lp : 0.0));
      return;
    }
    if (!r3) layernorm(t + ".norm3");
    // (a workgroup of that kernel sweeps a quarter of the hidden units for its 128 tokens -- 36 dependent tile steps: worth it once the token blocks
    //  fill the chip; below ~144 workgroups the GEMM's 24 column tiles per row block finish sooner.  r5 batch sweep: batch 1-4 +0.1 ms/step without this; crossover between 3760 and 5640 rows)
    if (r3 && h->fuse_geglu && a.geglu_stream && geglu_eligible(d, Tl, pr) && M >= g_geglu_min_rows) {
      // the token rows stay in LDS, only weights stream (csrc/geglu.hip): half the L2 -> LDS bytes of the GEMM below
      ns2vc_geglu_args f;
      memset(&f, 0, sizeof(f));
      f.yn = yn; f.ldy = d; f.ln_stats = r3; f.ln_eps = 1e-5f;
      f.wstream = a.geglu_stream; f.consts = a.geglu_consts;
      f.out_op = ffh; f.ldo = 4 * d; f.M = M; f.dim = d; f.ln_health = h->ln_health;
      add(t + ".ff.geglu[token-stationary]", [=](hipStream_t s) { return launch_geglu(f, pr, s); }, 1, 2.0 * M * (double)d * 8.0 * d,
          (double)M * d * opsz * 5.0 + (double)M * (d / 64) * 8.0 + 8.0 * d * d * opsz);
    } else {
      g = base(yn, d, d, Tl, Tl, a.ff1, nullptr, ffh, 4 * d);
      g.geglu = 1;
      consume(g, r3, a.ff1);
      gemm(t + ".ff.geglu", g);
    }
    if (fold) {
      // out = [Wpo W2 | Wpo] [ffh | yn] + (Wpo b2 + bpo) + x : ff.net.2 and proj_out in one launch
      g = base(ffh, 4 * d, 4 * d, Tl, Tl, a.ffpo, out, out_op, d);
      g.a2 = yraw; g.lda2 = d; g.c2 = d;
      g.res = x; g.ldres = d;
      g.stats = new_stats(out, Tl, d);
      gemm(a.prefix + ".ff.out+proj_out", g);
    } else {
      g = base(ffh, 4 * d, 4 * d, Tl, Tl, a.ff2, nullptr, yn, d);     // y_final = y + ff(...) is only consumed by proj_out: operand copy only
      g.res = y; g.ldres = d;
      gemm(t + ".ff.out", g);
      g = base(yn, d, d, Tl, Tl, a.proj_out, out, out_op, d);
      g.res = x; g.ldres = d;
      g.stats = new_stats(out, Tl, d);
      gemm(a.prefix + ".proj_out", g);
    }
  }
};

int build_plan(ns2vc_unet* h, bool sizing) {
  const auto& c = h->cfg;
  const int B = h->B, T = h->T, Lp = h->Lp, nl = c.n_levels;
  const int c0 = c.block_out_channels[0], E = 4 * c0, cross = c.cross_attention_dim, CP = h->CP;
  std::vector<int> Ts(nl);
  Ts[0] = T;
  for (int l = 1; l < nl; ++l) Ts[l] = (Ts[l - 1] + 1) / 2;

  Planner P;
  P.h = h; P.sizing = sizing; P.B = B; P.T = T; P.Lp = Lp; P.G = c.norm_num_groups; P.prec = h->prec;
  P.opsz = operand_bytes(h->prec);
  const int prec = h->prec;
  if (!sizing) { h->cond_ops.clear(); h->fwd_ops.clear(); h->taps.clear(); }

  size_t maxMC = 0, maxIn = 0;   // max over levels of B*Tl*C (outputs) / over resnets of B*Tl*Cin (concat inputs)
  int maxC = 0;
  for (int l = 0; l < nl; ++l) {
    // an upsampler writes the COARSER level's channel count at this level's length
    const int cmax = std::max(c.block_out_channels[l], c.block_out_channels[std::min(l + 1, nl - 1)]);
    maxMC = std::max(maxMC, (size_t)B * Ts[l] * cmax);
    maxC = std::max(maxC, c.block_out_channels[l]);
  }
  for (const auto& b : h->blocks)
    for (const auto& r : b.res) maxIn = std::max(maxIn, (size_t)B * Ts[b.level] * r.cin);
  maxIn = std::max(maxIn, maxMC);
  // ---- persistent state
  h->xe = P.alloc<float>((size_t)B * T * CP); h->xbar = P.alloc<float>((size_t)B * T * CP);
  h->d1 = P.alloc<float>((size_t)B * T * CP); h->mprev = P.alloc<float>((size_t)B * T * CP);
  h->x0 = P.alloc<float>((size_t)B * T * CP);
  // 16-bit engines keep the two inputs of conv_in as hi + lo operand pairs, rows [hi(C) | lo(C)] (common.h op_rest): the lo planes are read with split_io only
  const int pw = prec != PREC_F32 ? 2 : 1;
  h->xe_op = P.alloc_op((size_t)B * T * CP * pw);
  h->content_op = P.alloc_op((size_t)B * T * c.content_channels * pw);
  const bool pio = h->split_io && prec != PREC_F32 && !h->exact_io;
  const bool xio = h->exact_io && prec != PREC_F32;            // conv_in / conv_out / time_emb_proj with fp32 operands inside a 16-bit engine
  h->content_f32 = xio ? P.alloc<float>((size_t)B * T * c.content_channels) : nullptr;
  h->emb_act_f32 = xio ? P.alloc<float>((size_t)B * E) : nullptr;
  h->content_conv = P.alloc<float>((size_t)B * T * c0);
  h->prompt = P.alloc<float>((size_t)B * Lp * cross);
  h->prompt_op = P.alloc_op((size_t)B * Lp * cross);
  h->maskbias = P.alloc<float>((size_t)B * Lp);
  h->mask_dev = P.alloc<uint8_t>((size_t)B * Lp);
  h->aug = P.alloc<float>((size_t)B * E); h->emb = P.alloc<float>((size_t)B * E);
  h->emb_act_op = P.alloc_op((size_t)B * E);
  h->temb = P.alloc<float>((size_t)B * h->temb_all.N);
  h->kv = P.alloc_op((size_t)B * Lp * h->kv_all.N);
  h->seq = P.alloc<float>((size_t)B * (Lp + 1) * cross);
  h->seq_op = P.alloc_op((size_t)B * (Lp + 1) * cross);
  h->pool_qkv_buf = P.alloc<float>((size_t)B * (Lp + 1) * h->pool_qkv.N);
  h->pooled = P.alloc<float>((size_t)B * cross);
  h->t_dev = P.alloc<float>((size_t)B);
  h->step_dev = P.alloc<int>(64);
  h->ln_health = P.alloc<unsigned>(64);
  h->attn_fallbacks = h->ln_health + 32;          // (same zero-initialised block; the LayerNorm read-out uses words 0 and 16, the cooperative GroupNorm prologue's counter word 48)
  // ---- shared scratch
  P.gn_rows = 32;
  P.gn_partial = P.alloc<double>((size_t)B * ((T + P.gn_rows - 1) / P.gn_rows) * c.norm_num_groups * 2);
  P.xn = P.alloc_op(maxIn); P.xr = P.alloc_op(maxIn);
  P.rs1 = P.alloc<float>(maxMC / 32); P.rs2 = P.alloc<float>(maxMC / 32); P.rs3 = P.alloc<float>(maxMC / 32);
  float* h1 = P.alloc<float>(maxMC);
  void* hn = P.alloc_op(maxMC);
  float* y = P.alloc<float>(maxMC);
  void* yn = P.alloc_op(maxMC);
  void* qkv = P.alloc_op(3 * maxMC);
  void* ao = P.alloc_op(maxMC);
  void* qb = P.alloc_op(maxMC);
  void* ffh = P.alloc_op(4 * maxMC);
  void* samp_in = P.alloc_op(maxMC);        // operand copy of a block output that a down/up-sampling conv reads
  float* ua = P.alloc<float>(maxMC);
  float* ub = P.alloc<float>(maxMC);
  float* uc = P.alloc<float>(maxMC);

  // ================= condition plan (once per utterance batch) =================
  P.ops = &h->cond_ops;
  {
    float *prompt = h->prompt, *seq = h->seq, *pq = h->pool_qkv_buf, *pooled = h->pooled, *aug = h->aug;
    void *prompt_op = h->prompt_op, *seq_op = h->seq_op;
    // content half of conv_in (+ conv_in bias)
    const int cc = c.content_channels;
    GemmArgs g = xio ? P.base(h->content_f32, cc, cc, T, T, h->conv_in_c32, h->content_conv, nullptr, c0)
               : pio ? P.base(h->content_op, 2 * cc, 2 * cc, T, T, h->conv_in_cp, h->content_conv, nullptr, c0)
                     : P.base(h->content_op, pw * cc, cc, T, T, h->conv_in_c, h->content_conv, nullptr, c0);
    if (pio) { g.a1 = h->content_op; g.lda1 = 2 * cc; g.c1 = cc; }       // [hi | lo] then hi once more, against (hi(w) | hi(w) | lo(w))
    g.taps = 3;
    P.gemm("cond.conv_in.content", g, xio ? PREC_F32 : -1);
    if (!sizing) h->cond_split = h->cond_ops.size();
    // all cross-attention k|v projections in one GEMM: prompt [B*Lp][cross] x [n_kv][cross]^T -> operand tensor
    const size_t np = (size_t)B * Lp * cross;
    P.add("cond.prompt.cast", [=](hipStream_t s) { return launch_cast_op(prompt, np, prompt_op, prec, s); });
    g = P.base(prompt_op, cross, cross, Lp, Lp, h->kv_all, nullptr, h->kv, h->kv_all.N);
    P.gemm("cond.cross_kv", g);
    // r6: the V^T images of the blocks whose cross-attention runs inside the fused feed-forward kernel (ffn.hip ATT): one small launch each, once per utterance
    for (const auto& b : h->blocks)
      for (const auto& at : b.attn)
        if (P.xattn_fused(at, Ts[b.level])) {
          const int ldv = h->kv_all.N, hdv = at.dim / 8, Bq = B, Lq = Lp;
          void* vt = P.alloc_op(xattn_pack_bytes(B, Lp, hdv) / 2);
          P.xattn_vt[at.prefix] = vt;
          const void* ksrc = P.op_off(h->kv, (size_t)at.kv_off);
          const void* vsrc = P.op_off(h->kv, (size_t)(at.kv_off + at.dim));
          P.add("cond.cross_kv_image." + at.prefix, [=](hipStream_t s) { return launch_xattn_pack(ksrc, ldv, vsrc, ldv, Bq, Lq, hdv, vt, prec, s); }, 4);
        }
    // add_embedding = TextTimeEmbedding(prompt)
    const float *n1g = h->p_n1g, *n1b = h->p_n1b, *pos = h->p_pos, *projT = h->p_projT, *projb = h->p_projb, *n2g = h->p_n2g, *n2b = h->p_n2b;
    const int ph_ = c.pool_heads;
    const size_t ns = (size_t)B * (Lp + 1) * cross;
    P.add("cond.pool.ln1", [=](hipStream_t s) { return launch_ln_apply(prompt, B * Lp, cross, 1e-5f, n1g, n1b, seq, Lp, 0, s); });
    P.add("cond.pool.cls", [=](hipStream_t s) { return launch_pool_cls(seq, B, Lp, cross, pos, s); });
    P.add("cond.pool.cast", [=](hipStream_t s) { return launch_cast_op(seq, ns, seq_op, prec, s); });
    g = P.base(seq_op, cross, cross, Lp + 1, Lp + 1, h->pool_qkv, pq, nullptr, h->pool_qkv.N);
    P.gemm("cond.pool.qkv", g);
    const int ldq = h->pool_qkv.N;
    if (ldq != 3 * cross) return fail("pool qkv width %d must equal 3*cross=%d (cross must be a multiple of 128)", ldq, 3 * cross);
    if ((ns & 3) || (np & 3)) return fail("internal: cast sizes must be multiples of 4");
    P.add("cond.pool.attn", [=](hipStream_t s) { return launch_pool_attn(pq, B, Lp + 1, cross, ph_, pooled, s); });
    P.add("cond.pool.proj", [=](hipStream_t s) { return launch_pool_proj(pooled, B, cross, projT, projb, E, n2g, n2b, 1e-5f, aug, s); });
    P.tap("aug", aug, B, E);
  }

  // ================= per-step forward plan =================
  P.ops = &h->fwd_ops;
  {
    const size_t cap = (size_t)1 << 20;                 // 8 MB of int64 statistics slots
    P.stats_pool = P.alloc<long long>(cap);
    P.stats_cap = cap; P.stats_used = 0;
    long long* pool = P.stats_pool;
    ns2vc_unet* hq = h;
    // first launch of every forward: clears the statistics pool and, in the sampling loop, advances the step counter
    // (ns2vc_sampler_run starts it at -1; a plain forward does not read it)
    P.add("gn_stats.clear", [=](hipStream_t s) { return launch_zero(pool, hq->stats_bytes, s, hq->step_dev); }, 4);
  }
  {
    ns2vc_unet* hh = h;
    const float *w1t = h->t_w1t, *b1 = h->t_b1, *w2t = h->t_w2t, *b2 = h->t_b2, *aug = h->aug;
    float *emb = h->emb, *tdev = h->t_dev;
    void* emb_act = xio ? (void*)h->emb_act_f32 : h->emb_act_op;
    const int eprec = xio ? PREC_F32 : prec;                    // type SiLU(emb) is written in
    const int tdim = c0;
    if (!sizing) { h->temb_begin = (int)h->fwd_ops.size(); h->temb_join = -1; }
    P.add("time_embed", [=](hipStream_t s) {
      // sampling loop: the MLP of every step's timestep was evaluated once for the table (ns2vc_sampler_run), a step adds aug
      if (hh->use_step_table) return launch_emb_from_table(hh->temb_table, hh->step_dev, aug, emb, emb_act, eprec, B, E, s);
      return launch_time_embed(tdev, 1, nullptr, 0, w1t, b1, w2t, b2, aug, emb, emb_act, eprec, B, tdim, E, s);
    });
    P.tap("emb", emb, B, E);
    // every resnet's time_emb_proj(SiLU(emb)) in one GEMM (M = B)
    GemmArgs g = P.base(emb_act, E, E, 1, 1, xio ? h->temb_all32 : h->temb_all, h->temb, nullptr, h->temb_all.N);
    P.gemm("time_emb_proj.all", g, xio ? PREC_F32 : -1);
    if (!sizing) h->temb_end = (int)h->fwd_ops.size();
  }
  // skip stack
  struct Skip { float* p; int C; int l; };
  std::vector<Skip> skips;
  auto new_skip = [&](int l) { float* p = P.alloc<float>((size_t)B * Ts[l] * c.block_out_channels[l]); skips.push_back({p, c.block_out_channels[l], l}); return p; };
  {
    float* s0 = new_skip(0);
    GemmArgs g = xio ? P.base(h->xe, CP, CP, T, T, h->conv_in_x32, s0, nullptr, c0)          // (the fp32 solver state IS the fp32 operand: no copy involved)
               : pio ? P.base(h->xe_op, 2 * CP, 2 * CP, T, T, h->conv_in_xp, s0, nullptr, c0)
                     : P.base(h->xe_op, pw * CP, CP, T, T, h->conv_in_x, s0, nullptr, c0);
    if (pio) { g.a1 = h->xe_op; g.lda1 = 2 * CP; g.c1 = CP; }
    g.taps = 3; g.res = h->content_conv; g.ldres = c0;
    g.stats = P.new_stats(s0, T, c0);
    P.gemm("conv_in", g, xio ? PREC_F32 : -1);
    P.tap("conv_in", s0, B * T, c0);
  }
  const float* cur = skips.back().p;
  int curC = c0;
  for (const auto& b : h->blocks) {
    const int l = b.level, Tl = Ts[l];
    const std::string tag = b.kind == "mid" ? "mid" : b.kind + std::to_string(b.index);
    if (b.kind == "down") {
      for (size_t j = 0; j < b.res.size(); ++j) {
        const bool has_attn = !b.attn.empty();
        const bool last = (j + 1 == b.res.size());
        float* rout = has_attn ? ua : new_skip(l);
        P.resnet(b.res[j], cur, curC, curC, nullptr, 0, 0, Tl, h1, hn, rout, (!has_attn && last && b.sampler) ? samp_in : nullptr);
        P.tap(tag + ".res" + std::to_string(j), rout, B * Tl, b.channels);
        cur = rout; curC = b.channels;
        if (has_attn) {
          float* aout = new_skip(l);
          P.transformer(b.attn[j], cur, Tl, y, yn, qkv, ao, qb, ffh, aout, (last && b.sampler) ? samp_in : nullptr);
          P.tap(tag + ".attn" + std::to_string(j), aout, B * Tl, b.channels);
          cur = aout;
        }
      }
      if (b.sampler == 1) {
        float* ds = new_skip(l + 1);
        skips.back().C = b.channels;     // this block's channels at the next level's length
        GemmArgs g = P.base(samp_in, curC, curC, Tl, Ts[l + 1], b.samp, ds, nullptr, b.channels);
        g.taps = 3; g.tmode = TMODE_DOWN2;
        g.stats = P.new_stats(ds, Ts[l + 1], b.channels);
        P.gemm(tag + ".downsample", g);
        P.tap(tag + ".ds", ds, B * Ts[l + 1], b.channels);
        cur = ds;
      }
    } else if (b.kind == "mid") {
      P.resnet(b.res[0], cur, curC, curC, nullptr, 0, 0, Tl, h1, hn, ua, nullptr);
      P.tap("mid.res0", ua, B * Tl, b.channels);
      P.transformer(b.attn[0], ua, Tl, y, yn, qkv, ao, qb, ffh, ub, nullptr);
      P.tap("mid.attn0", ub, B * Tl, b.channels);
      P.resnet(b.res[1], ub, b.channels, b.channels, nullptr, 0, 0, Tl, h1, hn, uc, nullptr);
      P.tap("mid.res1", uc, B * Tl, b.channels);
      cur = uc; curC = b.channels;
    } else {
      for (size_t j = 0; j < b.res.size(); ++j) {
        const Skip sk = skips.back();
        skips.pop_back();
        const bool last = (j + 1 == b.res.size());
        const bool has_attn = !b.attn.empty();
        if (sk.l != l) return fail("internal: skip level mismatch at %s", b.res[j].prefix.c_str());
        if (curC + sk.C != b.res[j].cin) return fail("internal: concat width %d+%d != %d at %s", curC, sk.C, b.res[j].cin, b.res[j].prefix.c_str());
        float* rout = (cur == ua) ? ub : ua;
        if (rout == cur) rout = uc;
        P.resnet(b.res[j], cur, curC, curC, sk.p, sk.C, sk.C, Tl, h1, hn, rout, (!has_attn && last && b.sampler) ? samp_in : nullptr);
        P.tap(tag + ".res" + std::to_string(j), rout, B * Tl, b.channels);
        cur = rout; curC = b.channels;
        if (has_attn) {
          float* aout = (cur == ua) ? ub : ua;
          P.transformer(b.attn[j], cur, Tl, y, yn, qkv, ao, qb, ffh, aout, (last && b.sampler) ? samp_in : nullptr);
          P.tap(tag + ".attn" + std::to_string(j), aout, B * Tl, b.channels);
          cur = aout;
        }
      }
      if (b.sampler == 2) {
        float* us = (cur == uc) ? ua : uc;
        GemmArgs g = P.base(samp_in, curC, curC, Tl, Ts[l - 1], b.samp, us, nullptr, b.channels);
        g.taps = 3; g.tmode = TMODE_UP2;
        g.stats = P.new_stats(us, Ts[l - 1], b.channels);
        P.gemm(tag + ".upsample", g);
        P.tap(tag + ".us", us, B * Ts[l - 1], b.channels);
        cur = us;
      }
    }
  }
  if (!skips.empty()) return fail("internal: %zu skips left over", skips.size());
  {
    // split_io: conv_out reads a hi + lo operand pair -- written by its fused GroupNorm prologue (gnp_pair) or by the gn_apply launch, the same bytes either way
    const bool po = pio && 2 * curC <= 3 * c0 && h->conv_outp.w;
    const auto pno = P.groupnorm("conv_norm_out", cur, curC, curC, nullptr, 0, 0, T, 1e-5f, h->out_ng, h->out_nb, nullptr, 0, 0, 1, P.xn, nullptr, h->conv_out.N, 3, po ? 1 : 0);
    // exact_io: only where the norm is the conv's prologue (it then writes fp32 operand rows: xn holds 2-byte elements of up to 3 x 128 channels per row, i.e. room for 128 fp32)
    const bool xo = xio && pno.x != nullptr && (size_t)curC * 4 <= (size_t)3 * c0 * P.opsz;
    GemmArgs g = po ? P.base(P.xn, 2 * curC, 2 * curC, T, T, h->conv_outp, h->x0, nullptr, CP)
                    : P.base(P.xn, curC, curC, T, T, xo ? h->conv_out32 : h->conv_out, h->x0, nullptr, CP);
    if (po) { g.a1 = P.xn; g.lda1 = 2 * curC; g.c1 = curC; g.gnp_pair = pno.x ? 1 : 0; }       // [hi | lo] then hi once more (3 x 128 columns: what a row of xn holds)
    g.taps = 3;
    P.gn_fuse(g, pno);
    if (h->conv_out.N != CP) return fail("internal: conv_out padded width %d != %d", h->conv_out.N, CP);
    P.gemm("conv_out", g, xo ? PREC_F32 : -1);
    if (!sizing) { h->conv_out_g = g; h->conv_out_idx = (int)h->fwd_ops.size() - 1; h->conv_out_prec = xo ? PREC_F32 : prec; }
    P.tap("out", h->x0, B * T, CP);
  }
  if (sizing) h->arena_bytes = P.off;
  else if (P.off > h->arena_bytes) {    // a rebuild must never carve past the allocation the sizing pass measured
    h->cond_ops.clear(); h->fwd_ops.clear(); h->taps.clear();
    return fail("internal: plan needs %zu bytes but the arena holds %zu (re-run ns2vc_unet_prepare)", P.off, h->arena_bytes);
  }
  h->arena_used = P.off;
  h->stats_bytes = std::max<size_t>(P.stats_used, 1) * sizeof(long long);
  return 0;
}

int run_ops(const std::vector<Op>& ops, hipStream_t s, size_t first = 0, size_t last = (size_t)-1) {
  for (size_t i = first; i < std::min(last, ops.size()); ++i) {
    hipError_t e = ops[i].fn(s);
    if (e != hipSuccess) {
      const int line = last_gemm_refusal_line();
      if (line) return fail("launch of '%s' failed: %s (refused by the argument check at gemm.hip:%d)", ops[i].name.c_str(), hipGetErrorString(e), line);
      return fail("launch of '%s' failed: %s", ops[i].name.c_str(), hipGetErrorString(e));
    }
  }
  return 0;
}

// Every entry point that touches the device binds the calling thread to the engine's device first: weights, arena and
// the captured graph live there, and launches go to the CURRENT device (a model on cuda:1 called while cuda:0 is
// current would otherwise launch on the wrong GPU with cross-device pointers).
int bind_device(ns2vc_unet* h) {
  int cur = -1;
  HIPCHK(hipGetDevice(&cur));
  if (cur != h->device) HIPCHK(hipSetDevice(h->device));
  return 0;
}
int check_ready(ns2vc_unet* h, bool need_plan) {
  if (!h) return fail("null engine handle");
  if (bind_device(h)) return 1;
  if (!h->finalized) return fail("weights not finalized (call ns2vc_unet_finalize_weights)");
  if (need_plan && !h->arena) return fail("engine not prepared (call ns2vc_unet_prepare)");
  return 0;
}
void drop_plan(ns2vc_unet* h) {
  if (h->step_graph) { (void)hipGraphExecDestroy(h->step_graph); h->step_graph = nullptr; }
  if (h->arena) { (void)hipDeviceSynchronize(); (void)hipFree(h->arena); h->arena = nullptr; }
  h->cond_ops.clear(); h->fwd_ops.clear(); h->taps.clear();
  h->arena_bytes = h->arena_used = 0;
  h->next_step = -1;           // the solver state lived in the arena
  h->ln_posted = false;
  h->attn_fallbacks = nullptr; // (the counters lived in the arena too)
  h->ln_health = nullptr;
}

}  // namespace

// ====================================================================================
// C ABI
// ====================================================================================
static bool* option_ptr(ns2vc_unet* h, const char* name) {
  if (!strcmp(name, "ln_linear")) return &h->ln_linear;
  if (!strcmp(name, "fold_ff")) return &h->fold_ff;
  if (!strcmp(name, "fuse_ffn")) return &h->fuse_ffn;
  if (!strcmp(name, "fuse_rows")) return &h->fuse_rows;
  if (!strcmp(name, "fuse_rows_gn")) return &h->fuse_rows_gn;
  if (!strcmp(name, "fuse_gn_gemm")) return &h->fuse_gn_gemm;
  if (!strcmp(name, "gn_coop")) return &h->gn_coop;
  if (!strcmp(name, "fuse_gn_cat")) return &h->fuse_gn_cat;
  if (!strcmp(name, "fuse_ffn_pre")) return &h->fuse_ffn_pre;
  if (!strcmp(name, "fuse_geglu")) return &h->fuse_geglu;
  if (!strcmp(name, "attn_fp8")) return &h->attn_fp8;
  if (!strcmp(name, "attn_optimistic")) return &h->attn_optimistic;
  if (!strcmp(name, "slice_rows")) return &h->slice_rows;
  if (!strcmp(name, "conv_ts")) return &h->conv_ts;
  if (!strcmp(name, "conv_wtiled")) return &h->conv_wtiled;
  if (!strcmp(name, "gn_inloop")) return &h->gn_inloop;
  if (!strcmp(name, "fuse_solver")) return &h->fuse_solver;
  if (!strcmp(name, "fuse_xattn")) return &h->fuse_xattn;
  if (!strcmp(name, "fork_temb")) return &h->fork_temb;
  if (!strcmp(name, "exact_io")) return &h->exact_io;
  if (!strcmp(name, "split_io")) return &h->split_io;
  return nullptr;
}

extern "C" {

int ns2vc_abi_version(void) { return NS2VC_ABI_VERSION; }
const char* ns2vc_last_error(void) { return g_err.c_str(); }

int ns2vc_device_count(int* out_count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *out_count = 0; return fail("hipGetDeviceCount: %s", hipGetErrorString(e)); }
  *out_count = n;
  return 0;
}
int ns2vc_set_device(int device) { HIPCHK(hipSetDevice(device)); return 0; }
int ns2vc_device_name(char* buf, int buflen) {
  hipDeviceProp_t p;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  HIPCHK(hipGetDeviceProperties(&p, dev));
  snprintf(buf, buflen, "%s|%s|CUs=%d|clock_khz=%d|mem_gb=%.1f", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate,
           (double)p.totalGlobalMem / 1e9);
  return 0;
}

// per device (the process may serve several): the result of misc.hip's placement probe, taken at the first engine of a device
static int xcd_round_robin_of_current_device() {
  static std::mutex mu;
  static std::map<int, int> seen;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  std::lock_guard<std::mutex> lock(mu);
  auto it = seen.find(dev);
  if (it != seen.end()) return it->second;
  const int r = probe_xcd_round_robin();
  seen[dev] = r;
  return r;
}
int ns2vc_device_xcd_round_robin(int* out) {
  if (!out) return fail("null argument");
  *out = xcd_round_robin_of_current_device();
  return 0;
}

int ns2vc_unet_create(const ns2vc_unet_cfg* cfg, ns2vc_unet** out) {
  if (!cfg || !out) return fail("null argument");
  if (cfg->n_levels < 2 || cfg->n_levels > NS2VC_MAX_LEVELS) return fail("n_levels out of range");
  if (cfg->latent_channels <= 0 || cfg->latent_channels > 128) return fail("latent_channels must be in 1..128");
  if (cfg->content_channels % 64) return fail("content_channels must be a multiple of 64");
  if (cfg->cross_attention_dim % 128) return fail("cross_attention_dim must be a multiple of 128");
  if (cfg->block_out_channels[0] != 128) return fail("block_out_channels[0] must be 128 (padded latent width)");
  for (int l = 0; l < cfg->n_levels; ++l) {
    const int c = cfg->block_out_channels[l];
    if (c % 64 || c > 512) return fail("block_out_channels[%d]=%d must be a multiple of 64 and <= 512", l, c);
    if (c % cfg->heads) return fail("channels %d not divisible by heads %d", c, cfg->heads);
    const int hd = c / cfg->heads;
    if (hd != 16 && hd != 32 && hd != 48 && hd != 64) return fail("head_dim %d unsupported (16/32/48/64)", hd);
    if (c % cfg->norm_num_groups || (c / cfg->norm_num_groups) % 4) return fail("channels %d incompatible with %d groups", c, cfg->norm_num_groups);
  }
  if (cfg->norm_num_groups < 1 || cfg->norm_num_groups > 8) return fail("norm_num_groups=%d unsupported (1..8)", cfg->norm_num_groups);
  if (cfg->cross_attention_dim % cfg->pool_heads || cfg->cross_attention_dim / cfg->pool_heads > 8) return fail("pool heads unsupported");
  hipError_t e = init_gemm_attributes();
  if (e == hipSuccess) e = init_convts_attributes();
  if (e == hipSuccess) e = init_attn_attributes();
  if (e == hipSuccess) e = init_ffn_attributes();
  if (e == hipSuccess) e = init_geglu_attributes();
  if (e == hipSuccess) e = init_rowchain_attributes();
  if (e != hipSuccess) return fail("kernel attribute setup failed: %s (is a gfx950 GPU visible?)", hipGetErrorString(e));
  auto* h = new ns2vc_unet();
  h->cfg = *cfg;
  if (hipGetDevice(&h->device) != hipSuccess) { delete h; return fail("hipGetDevice failed"); }
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) h->cus = prop.multiProcessorCount;
    h->bn128_min = h->cus * 5 / 8;             // 128-column tap-sharing tiles while they still give 5/8 of the CUs a workgroup (160 of 256)
  }
  // rows shared between workgroups through an XCD's L2 only where the placement probe has SEEN ids 8 apart on one XCD (and even then
  // every workgroup checks its own placement, gnpro.h)
  h->xcd_probe = xcd_round_robin_of_current_device();
  h->gn_coop = h->xcd_probe == 1;
  // plan switches from the environment: tuning / A-B runs only (tools/ab_libs.sh), honoured when NS2VC_DEBUG_ENV=1 -- a served engine's
  // launch plan is set through ns2vc_unet_set_option and never changes behind the caller's back
  if (const char* dbg = getenv("NS2VC_DEBUG_ENV"); dbg && atoi(dbg) != 0) {
    static const struct { const char* env; const char* opt; } sw[] = {
      {"NS2VC_LN_LINEAR", "ln_linear"}, {"NS2VC_FOLD_FF", "fold_ff"}, {"NS2VC_FUSE_FFN", "fuse_ffn"}, {"NS2VC_FUSE_ROWS", "fuse_rows"},
      {"NS2VC_FUSE_ROWS_GN", "fuse_rows_gn"}, {"NS2VC_FUSE_GN_GEMM", "fuse_gn_gemm"}, {"NS2VC_GN_COOP", "gn_coop"}, {"NS2VC_FUSE_GN_CAT", "fuse_gn_cat"},
      {"NS2VC_SLICE_ROWS", "slice_rows"}, {"NS2VC_FUSE_FFN_PRE", "fuse_ffn_pre"}, {"NS2VC_FUSE_GEGLU", "fuse_geglu"}, {"NS2VC_ATTN_FP8", "attn_fp8"}, {"NS2VC_ATTN_OPTIMISTIC", "attn_optimistic"},
      {"NS2VC_CONV_TS", "conv_ts"}, {"NS2VC_CONV_WTILED", "conv_wtiled"}, {"NS2VC_GN_INLOOP", "gn_inloop"}, {"NS2VC_FUSE_SOLVER", "fuse_solver"}, {"NS2VC_FUSE_XATTN", "fuse_xattn"}, {"NS2VC_FORK_TEMB", "fork_temb"}, {"NS2VC_EXACT_IO", "exact_io"}, {"NS2VC_SPLIT_IO", "split_io"}};
    for (const auto& s : sw)
      if (const char* v = getenv(s.env)) {
        if (bool* o = option_ptr(h, s.opt)) *o = atoi(v) != 0;
      }
    if (h->xcd_probe != 1) h->gn_coop = false;
    {   // loader waves (4 | 8) and consumer layout (NS2VC_TS_KS = 1: K-split, 0: plain) of the tap-sharing conv kernel, process-wide
      const char* nl = getenv("NS2VC_TS_NL");
      const char* ks = getenv("NS2VC_TS_KS");
      if (nl || ks) set_forced_gemm_tile(-4, ks ? (atoi(ks) ? 1 : 2) : 0, nl ? atoi(nl) : 0);
    }
    if (const char* v = getenv("NS2VC_GN_COOP_MIN")) h->gn_coop_min = atoi(v);
    if (const char* v = getenv("NS2VC_TS_BN128_MIN")) h->bn128_min = atoi(v) > 0 ? atoi(v) : h->bn128_min;        // workgroups a 128-column tiling must still give to be chosen
  }
  h->blocks = make_topology(*cfg);
  build_expected(h);
  *out = h;
  return 0;
}

int ns2vc_unet_destroy(ns2vc_unet* h) {
  if (h) (void)bind_device(h);
  delete h;
  return 0;
}

int ns2vc_unet_load_weight(ns2vc_unet* h, const char* key, const void* data, const int64_t* shape, int ndim) {
  if (!h || !key || !data) return fail("null argument");
  if (bind_device(h)) return 1;
  std::string k(key);
  const std::vector<int64_t>* want = nullptr;
  for (const auto& e : h->expected) if (e.first == k) { want = &e.second; break; }
  if (!want) return fail("unexpected weight key '%s'", key);
  if ((int)want->size() != ndim) return fail("%s: ndim %d != expected %zu", key, ndim, want->size());
  for (int i = 0; i < ndim; ++i) if (shape[i] != (*want)[i]) return fail("%s: shape[%d]=%lld != expected %lld", key, i, (long long)shape[i], (long long)(*want)[i]);
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.data.resize(t.numel());
  HIPCHK(hipMemcpy(t.data.data(), data, t.numel() * sizeof(float), hipMemcpyDefault));
  h->raw[k] = std::move(t);
  h->finalized = false;
  return 0;
}

int ns2vc_unet_num_missing_weights(ns2vc_unet* h, char* first_missing, int buflen) {
  int n = 0;
  for (const auto& e : h->expected)
    if (!h->raw.count(e.first)) {
      if (n == 0 && first_missing && buflen > 0) snprintf(first_missing, buflen, "%s", e.first.c_str());
      ++n;
    }
  return n;
}

int ns2vc_unet_finalize_weights(ns2vc_unet* h, int precision) {
  if (!h) return fail("null engine handle");
  if (bind_device(h)) return 1;
  if (precision != NS2VC_PREC_F32 && precision != NS2VC_PREC_BF16 && precision != NS2VC_PREC_F16) return fail("unknown precision %d", precision);
  char first[256] = {0};
  const int miss = ns2vc_unet_num_missing_weights(h, first, sizeof(first));
  if (miss) return fail("%d weights missing, first: %s", miss, first);
  for (void* p : h->weight_allocs) (void)hipFree(p);
  h->weight_allocs.clear();
  h->prec = precision;
  if (pack_all(h)) return 1;
  h->finalized = true;
  h->temb_table_valid = false;
  drop_plan(h);       // a plan built for other weights holds stale pointers
  return 0;
}

int ns2vc_unet_set_debug(ns2vc_unet* h, int enable) {
  if (!h) return fail("null engine handle");
  if (bind_device(h)) return 1;
  if (h->debug != (enable != 0)) {
    h->debug = enable != 0;
    drop_plan(h);     // the tap copies change the arena size: a later rebuild must not carve a stale allocation
  }
  return 0;
}

int ns2vc_unet_set_option(ns2vc_unet* h, const char* name, int value) {
  if (!h || !name) return fail("null argument");
  if (bind_device(h)) return 1;
  bool* opt = option_ptr(h, name);
  if (!opt) return fail("unknown option '%s' (ln_linear, fold_ff, fuse_ffn, fuse_ffn_pre, fuse_geglu, fuse_rows, fuse_rows_gn, fuse_gn_gemm, fuse_gn_cat, gn_coop, slice_rows, attn_fp8, attn_optimistic, conv_ts, conv_wtiled, gn_inloop, fuse_solver, fuse_xattn, fork_temb, exact_io, split_io)", name);
  // the cooperative GroupNorm prologue only where the placement probe of this device came back positive (r5)
  if (opt == &h->gn_coop && value != 0 && h->xcd_probe != 1) return fail("gn_coop needs workgroup ids 8 apart on one XCD; the placement probe of this device returned %d", h->xcd_probe);
  if (*opt != (value != 0)) { *opt = value != 0; drop_plan(h); }
  return 0;
}

// The maximum lives in the arena and is raised by the consumers' atomicMax on whatever stream the forward runs on, so the
// read-and-reset is a kernel + an async copy ON THAT STREAM (a host-side memset on the legacy stream raced with the
// non-blocking streams the engine is driven on, and a device-wide synchronize stalled the overlapped pipeline).
int ns2vc_unet_ln_ratio_post(ns2vc_unet* h, void* stream) {
  if (check_ready(h, true)) return 1;
  hipStream_t s = (hipStream_t)stream;
  if (!h->ln_mail) {
    HIPCHK(hipHostMalloc((void**)&h->ln_mail, 64, hipHostMallocDefault));
    h->ln_mail[0] = 0;
  }
  if (!h->ln_event) HIPCHK(hipEventCreateWithFlags(&h->ln_event, hipEventDisableTiming));
  HIPCHK(launch_snapshot_u32(h->ln_health, h->ln_health + 16, s));
  HIPCHK(hipMemcpyAsync(h->ln_mail, h->ln_health + 16, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  HIPCHK(hipEventRecord(h->ln_event, s));
  h->ln_posted = true;
  return 0;
}

int ns2vc_unet_ln_ratio_poll(ns2vc_unet* h, float* out_ratio, int* out_ready) {
  if (!h || !out_ratio || !out_ready) return fail("null argument");
  *out_ready = 0;
  *out_ratio = 0.f;
  if (!h->ln_posted) return 0;
  if (bind_device(h)) return 1;
  const hipError_t q = hipEventQuery(h->ln_event);
  if (q == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
  if (q != hipSuccess) return fail("hipEventQuery failed: %s", hipGetErrorString(q));
  memcpy(out_ratio, h->ln_mail, sizeof(float));
  *out_ready = 1;
  h->ln_posted = false;
  return 0;
}

int ns2vc_unet_ln_ratio(ns2vc_unet* h, float* out_ratio, void* stream) {
  if (!out_ratio) return fail("null argument");
  if (ns2vc_unet_ln_ratio_post(h, stream)) return 1;
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  memcpy(out_ratio, h->ln_mail, sizeof(float));
  h->ln_posted = false;
  return 0;
}

int ns2vc_unet_prepare(ns2vc_unet* h, int B, int T, int Lp) {
  if (check_ready(h, false)) return 1;
  if (B <= 0 || T <= 0 || Lp <= 0) return fail("B, T, Lp must be positive");
  const int min_t = 1 << (h->cfg.n_levels - 1);
  if (T < min_t) return fail("T=%d too short for %d levels", T, h->cfg.n_levels);
  drop_plan(h);
  h->B = B; h->T = T; h->Lp = Lp;
  h->has_mask = false;
  if (build_plan(h, true)) return 1;
  HIPCHK(hipMalloc(&h->arena, h->arena_bytes));
  HIPCHK(hipMemset(h->arena, 0, h->arena_bytes));
  if (build_plan(h, false)) return 1;
  return 0;
}

int ns2vc_unet_workspace_bytes(ns2vc_unet* h, size_t* out) {
  if (!h || !out) return fail("null argument");
  *out = h->arena_bytes;
  return 0;
}

int ns2vc_unet_set_content(ns2vc_unet* h, const float* content_bct, void* stream) {
  if (check_ready(h, true)) return 1;
  if (!content_bct) return fail("null condition tensor");
  hipStream_t s = (hipStream_t)stream;
  const auto& c = h->cfg;
  { const int pw = h->prec != PREC_F32 ? 2 : 1;      // (16-bit: the hi + lo pair, see prepare)
    HIPCHK(launch_nct_to_btc(content_bct, c.content_channels, h->T, h->B, h->content_f32, h->content_op, h->prec, c.content_channels, c.content_channels, s,
                             pw * c.content_channels, pw == 2 ? c.content_channels : 0)); }
  return run_ops(h->cond_ops, s, 0, h->cond_split);
}

int ns2vc_unet_set_mask(ns2vc_unet* h, const uint8_t* mask_bl, void* stream) {
  if (check_ready(h, true)) return 1;
  hipStream_t s = (hipStream_t)stream;
  const bool want_mask = mask_bl != nullptr;
  if (want_mask != h->has_mask) {
    // the cross-attention launches bake in whether a bias is read: rebuild the (cheap) plan; workspace offsets do not
    // depend on it, so everything already hoisted stays valid
    h->has_mask = want_mask;
    if (h->step_graph) { (void)hipGraphExecDestroy(h->step_graph); h->step_graph = nullptr; }
    if (build_plan(h, false)) return 1;
  }
  if (want_mask) {
    HIPCHK(hipMemcpyAsync(h->mask_dev, mask_bl, (size_t)h->B * h->Lp, hipMemcpyDeviceToDevice, s));
    HIPCHK(launch_mask_bias(h->mask_dev, h->B * h->Lp, h->maskbias, s));
  }
  return 0;
}

int ns2vc_unet_set_prompt(ns2vc_unet* h, const float* prompt_blc, const uint8_t* mask_bl, void* stream) {
  if (!prompt_blc) return fail("null condition tensor");
  if (ns2vc_unet_set_mask(h, mask_bl, stream)) return 1;
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipMemcpyAsync(h->prompt, prompt_blc, (size_t)h->B * h->Lp * h->cfg.cross_attention_dim * sizeof(float), hipMemcpyDeviceToDevice, s));
  return run_ops(h->cond_ops, s, h->cond_split);
}

int ns2vc_unet_set_condition(ns2vc_unet* h, const float* content_bct, const float* prompt_blc, const uint8_t* mask_bl, void* stream) {
  if (!content_bct || !prompt_blc) return fail("null condition tensor");
  if (ns2vc_unet_set_prompt(h, prompt_blc, mask_bl, stream)) return 1;      // (first: it may rebuild the plan)
  return ns2vc_unet_set_content(h, content_bct, stream);
}

int ns2vc_unet_forward(ns2vc_unet* h, const float* x_bct, const float* t_b, float* out_bct, void* stream) {
  if (check_ready(h, true)) return 1;
  if (!x_bct || !t_b || !out_bct) return fail("null tensor");
  hipStream_t s = (hipStream_t)stream;
  const auto& c = h->cfg;
  h->use_step_table = false;
  HIPCHK(launch_nct_to_btc(x_bct, c.latent_channels, h->T, h->B, h->xe, h->xe_op, h->prec, h->CP, h->CP, s, h->prec != PREC_F32 ? 2 * h->CP : h->CP, h->prec != PREC_F32 ? h->CP : 0));
  HIPCHK(hipMemcpyAsync(h->t_dev, t_b, (size_t)h->B * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (run_ops(h->fwd_ops, s)) return 1;
  HIPCHK(launch_btc_to_nct(h->x0, h->CP, c.latent_channels, h->T, h->B, out_bct, s));
  return 0;
}

int ns2vc_sampler_load(ns2vc_unet* h, int steps, const float* coef_host) {
  const int kMaxSteps = 1024;   // fixed capacity: the table pointer is baked into the captured graph
  if (!h || !coef_host || steps <= 0) return fail("bad sampler table");
  if (steps > kMaxSteps) return fail("at most %d solver steps are supported", kMaxSteps);
  if (!h->coef_dev) HIPCHK(hipMalloc((void**)&h->coef_dev, (size_t)kMaxSteps * NS2VC_NCOEF * sizeof(float)));
  HIPCHK(hipDeviceSynchronize());   // a previous loop may still be reading the table
  HIPCHK(hipMemcpy(h->coef_dev, coef_host, (size_t)steps * NS2VC_NCOEF * sizeof(float), hipMemcpyHostToDevice));
  {
    unsigned long long hsh = 1469598103934665603ull;
    const unsigned char* pb = reinterpret_cast<const unsigned char*>(coef_host);
    for (size_t i = 0; i < (size_t)steps * NS2VC_NCOEF * sizeof(float); ++i) { hsh ^= pb[i]; hsh *= 1099511628211ull; }
    h->coef_hash = hsh;
  }
  h->steps = steps;
  h->temb_table_valid = false;
  h->next_step = -1;
  return 0;
}

// one evaluation + solver update.  r6: when conv_out is the plan's last launch and runs on the tap-sharing kernel, the update happens in its epilogue
// (same arithmetic, element for element: common.h solver_upd) -- x0 is never written, the state tensors are read and written once instead of twice
static bool solver_in_conv_out(ns2vc_unet* h, GemmArgs& g) {
  if (!h->fuse_solver || h->debug || h->conv_out_idx < 0 || h->conv_out_idx != (int)h->fwd_ops.size() - 1) return false;
  if (h->conv_out_prec != h->prec) return false;      // (exact_io: conv_out runs in fp32 there, the operand copy of the state is 16-bit)
  g = h->conv_out_g;
  g.out_f32 = nullptr;
  g.sol_coef = h->coef_dev; g.sol_step = h->step_dev; g.sol_ncoef = NS2VC_NCOEF;
  g.sol_xe = h->xe; g.sol_xe_op = h->xe_op; g.sol_xbar = h->xbar; g.sol_d1 = h->d1; g.sol_mprev = h->mprev; g.sol_ld = h->CP; g.sol_op_pair = h->prec != PREC_F32;
  return gemm_uses_convts(g, h->conv_out_prec);
}
static int run_step(ns2vc_unet* h, hipStream_t s, bool capturing = false) {
  GemmArgs g;
  const bool fold = solver_in_conv_out(h, g);
  const size_t last = fold ? (size_t)h->conv_out_idx : h->fwd_ops.size();
  size_t first = 0;
  // r6: under capture the timestep-embedding branch becomes a parallel branch of the graph (fork after the statistics clear, which advances the step counter
  // the branch reads; join in front of the first launch that reads the scale / shift rows).  Eager loops keep one stream: same launches, same results.
  if (capturing && h->fork_temb && !h->debug && h->temb_begin > 0 && h->temb_end > h->temb_begin && h->temb_join >= h->temb_end && (size_t)h->temb_join <= last) {
    if (!h->side_stream) HIPCHK(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
    if (!h->ev_fork) HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    if (!h->ev_join) HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    if (run_ops(h->fwd_ops, s, 0, (size_t)h->temb_begin)) return 1;
    HIPCHK(hipEventRecord(h->ev_fork, s));
    HIPCHK(hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
    if (run_ops(h->fwd_ops, h->side_stream, (size_t)h->temb_begin, (size_t)h->temb_end)) return 1;
    HIPCHK(hipEventRecord(h->ev_join, h->side_stream));
    if (run_ops(h->fwd_ops, s, (size_t)h->temb_end, (size_t)h->temb_join)) return 1;
    HIPCHK(hipStreamWaitEvent(s, h->ev_join, 0));
    first = (size_t)h->temb_join;
  }
  if (run_ops(h->fwd_ops, s, first, last)) return 1;
  if (fold) {
    HIPCHK(launch_gemm(g, h->conv_out_prec, s));
    return 0;
  }
  const size_t n = (size_t)h->B * h->T * h->CP;
  HIPCHK(launch_solver_update(h->coef_dev, h->step_dev, NS2VC_NCOEF, h->x0, h->xe, h->xe_op, h->prec, h->xbar, h->d1, h->mprev, n, s, h->prec != PREC_F32 ? h->CP : 0));
  return 0;
}

// The loop in three parts, so that a caller can hand the solver state to a second engine in mid-loop (mixed precision:
// ns2vc_sampler_handoff): begin = state from x_T, steps = the next n evaluations + updates, end = layout change back.
int ns2vc_sampler_begin(ns2vc_unet* h, const float* x_T_bct, void* stream) {
  if (check_ready(h, true)) return 1;
  if (!x_T_bct) return fail("null tensor");
  if (!h->coef_dev || h->steps <= 0) return fail("no solver table loaded (call ns2vc_sampler_load)");
  hipStream_t s = (hipStream_t)stream;
  const auto& c = h->cfg;
  const size_t n = (size_t)h->B * h->T * h->CP;
  HIPCHK(launch_nct_to_btc(x_T_bct, c.latent_channels, h->T, h->B, h->xe, h->xe_op, h->prec, h->CP, h->CP, s, h->prec != PREC_F32 ? 2 * h->CP : h->CP, h->prec != PREC_F32 ? h->CP : 0));
  HIPCHK(launch_copy16(h->xe, h->xbar, n * sizeof(float), s));
  HIPCHK(launch_zero(h->d1, n * sizeof(float), s));
  HIPCHK(launch_zero(h->mprev, n * sizeof(float), s));
  HIPCHK(launch_fill_i32(h->step_dev, -1, s));      // the first launch of every step advances it (gn_stats.clear)
  h->next_step = 0;
  return 0;
}

int ns2vc_sampler_steps(ns2vc_unet* h, int n_steps, int use_graph, void* stream) {
  if (check_ready(h, true)) return 1;
  if (!h->coef_dev || h->steps <= 0) return fail("no solver table loaded (call ns2vc_sampler_load)");
  if (h->next_step < 0) return fail("no sampling loop in progress (call ns2vc_sampler_begin or ns2vc_sampler_handoff)");
  if (n_steps < 0 || h->next_step + n_steps > h->steps) return fail("steps %d..%d outside the loaded table of %d", h->next_step, h->next_step + n_steps, h->steps);
  hipStream_t s = (hipStream_t)stream;
  const auto& c = h->cfg;
  h->use_step_table = true;
  if (!h->temb_table_valid) {      // new table or new weights: timestep MLP of every table row (column 0 = t), no prompt term
    const int E = c.block_out_channels[0] * 4;
    if (!h->temb_table) HIPCHK(hipMalloc((void**)&h->temb_table, (size_t)1024 * E * sizeof(float)));
    HIPCHK(launch_time_embed(h->coef_dev, NS2VC_NCOEF, nullptr, 0, h->t_w1t, h->t_b1, h->t_w2t, h->t_b2, nullptr, h->temb_table, nullptr,
                             h->prec, h->steps, c.block_out_channels[0], E, s));
    h->temb_table_valid = true;
  }
  if (use_graph && !h->step_graph && n_steps > 0) {
    if (!h->cap_stream) HIPCHK(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = run_step(h, h->cap_stream, true);
    hipError_t e = hipStreamEndCapture(h->cap_stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return 1; }
    if (e != hipSuccess) return fail("hipStreamEndCapture: %s", hipGetErrorString(e));
    e = hipGraphInstantiate(&h->step_graph, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { h->step_graph = nullptr; return fail("hipGraphInstantiate: %s", hipGetErrorString(e)); }
  }
  for (int i = 0; i < n_steps; ++i) {
    if (use_graph) HIPCHK(hipGraphLaunch(h->step_graph, s));
    else if (run_step(h, s)) return 1;
  }
  h->next_step += n_steps;
  return 0;
}

int ns2vc_sampler_end(ns2vc_unet* h, float* x_out_bct, void* stream) {
  if (check_ready(h, true)) return 1;
  if (!x_out_bct) return fail("null tensor");
  if (h->next_step < 0) return fail("no sampling loop in progress");
  const auto& c = h->cfg;
  HIPCHK(launch_btc_to_nct(h->xe, h->CP, c.latent_channels, h->T, h->B, x_out_bct, (hipStream_t)stream));
  h->next_step = -1;
  return 0;
}

// Solver state of `src` (x_e, x_bar, d1, m_prev, loop position) -> `dst`, which continues the SAME table from there: the
// engines may differ in precision (the state is fp32 in every mode; dst's operand copy of x_e is rebuilt in its own type).
// Both must be prepared for the same (B, T) and hold the same solver table and condition.
int ns2vc_sampler_handoff(ns2vc_unet* dst, ns2vc_unet* src, void* stream) {
  if (check_ready(dst, true) || check_ready(src, true)) return 1;
  if (dst == src) return fail("handoff to the same engine");
  if (dst->B != src->B || dst->T != src->T || dst->CP != src->CP) return fail("handoff between different shapes");
  if (dst->device != src->device) return fail("handoff between engines on different devices");
  if (src->next_step < 0) return fail("source engine has no sampling loop in progress");
  if (!dst->coef_dev || dst->steps != src->steps) return fail("destination engine must hold the same solver table (%d vs %d steps)", dst->steps, src->steps);
  if (dst->coef_hash != src->coef_hash) return fail("destination engine holds a DIFFERENT solver table with the same number of steps (solver / order / betas differ)");
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)src->B * src->T * src->CP;
  HIPCHK(launch_copy16(src->xe, dst->xe, n * sizeof(float), s));
  HIPCHK(launch_copy16(src->xbar, dst->xbar, n * sizeof(float), s));
  HIPCHK(launch_copy16(src->d1, dst->d1, n * sizeof(float), s));
  HIPCHK(launch_copy16(src->mprev, dst->mprev, n * sizeof(float), s));
  HIPCHK(launch_cast_op(dst->xe, n, dst->xe_op, dst->prec, s, dst->prec != PREC_F32 ? dst->CP : 0));
  HIPCHK(launch_fill_i32(dst->step_dev, src->next_step - 1, s));
  dst->next_step = src->next_step;
  src->next_step = -1;
  return 0;
}

int ns2vc_sampler_peek(ns2vc_unet* h, float* x_out_bct, void* stream) {
  if (check_ready(h, true)) return 1;
  if (!x_out_bct) return fail("null tensor");
  if (h->next_step < 0) return fail("no sampling loop in progress");
  HIPCHK(launch_btc_to_nct(h->xe, h->CP, h->cfg.latent_channels, h->T, h->B, x_out_bct, (hipStream_t)stream));
  return 0;
}

int ns2vc_unet_attn_fallbacks(ns2vc_unet* h, unsigned long long* count, int reset, void* stream) {
  if (!h || !count) return fail("null argument");
  *count = 0;
  if (!h->attn_fallbacks) return 0;             // no plan yet
  unsigned v = 0;
  HIPCHK(hipMemcpyAsync(&v, h->attn_fallbacks, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  *count = v;
  if (reset) HIPCHK(launch_zero(h->attn_fallbacks, 16, (hipStream_t)stream));
  return 0;
}

int ns2vc_unet_gn_coop_alone(ns2vc_unet* h, unsigned long long* count, int reset, void* stream) {
  if (!h || !count) return fail("null argument");
  *count = 0;
  if (!h->ln_health) return 0;                  // no plan yet
  unsigned v = 0;
  HIPCHK(hipMemcpyAsync(&v, h->ln_health + 48, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  *count = v;
  if (reset) HIPCHK(launch_zero(h->ln_health + 48, 16, (hipStream_t)stream));
  return 0;
}

int ns2vc_sampler_run(ns2vc_unet* h, float* x_inout_bct, int use_graph, void* stream) {
  if (ns2vc_sampler_begin(h, x_inout_bct, stream)) return 1;
  if (ns2vc_sampler_steps(h, h->steps, use_graph, stream)) return 1;
  return ns2vc_sampler_end(h, x_inout_bct, stream);
}

int ns2vc_unet_num_taps(ns2vc_unet* h) { return h ? (int)h->taps.size() : 0; }
int ns2vc_unet_tap_info(ns2vc_unet* h, int idx, char* name, int buflen, int* rows, int* cols) {
  if (!h || idx < 0 || idx >= (int)h->taps.size()) return fail("tap index out of range");
  snprintf(name, buflen, "%s", h->taps[idx].name.c_str());
  *rows = h->taps[idx].rows; *cols = h->taps[idx].cols;
  return 0;
}
int ns2vc_unet_tap_read(ns2vc_unet* h, int idx, float* host_dst) {
  if (!h || idx < 0 || idx >= (int)h->taps.size()) return fail("tap index out of range");
  HIPCHK(hipDeviceSynchronize());
  const Tap& t = h->taps[idx];
  HIPCHK(hipMemcpy(host_dst, t.copy, (size_t)t.rows * t.cols * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}
int ns2vc_unet_num_launches(ns2vc_unet* h, int* per_forward, int* per_condition) {
  if (!h) return fail("null engine handle");
  if (per_forward) *per_forward = (int)h->fwd_ops.size();
  if (per_condition) *per_condition = (int)h->cond_ops.size();
  return 0;
}

int ns2vc_unet_op_info(ns2vc_unet* h, int which, int idx, char* name, int buflen, int* kind, double* flops, double* bytes) {
  if (!h) return fail("null engine handle");
  const std::vector<Op>& ops = which ? h->cond_ops : h->fwd_ops;
  if (idx < 0 || idx >= (int)ops.size()) return fail("op index out of range");
  snprintf(name, buflen, "%s", ops[idx].name.c_str());
  *kind = ops[idx].kind; *flops = ops[idx].flops; *bytes = ops[idx].bytes;
  return 0;
}

// Times every launch of the per-step forward plan on `stream`: each op is launched `reps` times back to back
// between one hipEvent pair (so the event/launch overhead is amortised and the figure approaches the kernel's
// own duration, comparable with rocprofv3's kernel trace).  ms[i] = average milliseconds of launch i.
// The tensors hold garbage afterwards (in-place ops were repeated).  Synchronous.
int ns2vc_unet_profile_forward(ns2vc_unet* h, float* ms, int n_ms, int reps, void* stream) {
  if (check_ready(h, true)) return 1;
  const size_t n = h->fwd_ops.size();
  if ((size_t)n_ms < n) return fail("ms buffer too small: need %zu", n);
  if (reps < 1) reps = 1;
  hipStream_t s = (hipStream_t)stream;
  h->use_step_table = false;      // time with the (B,) timestep buffer of the plain forward
  std::vector<hipEvent_t> ev(2 * n);
  for (auto& e : ev) HIPCHK(hipEventCreate(&e));
  int rc = 0;
  // An op with a cooperative GroupNorm prologue starts from zeroed arrival words (the forward's clear launch): repeated here, it gets
  // its own small clear in front of every repetition, and the time of `reps` such clears alone (measured once, below) is taken off.
  hipEvent_t rz0 = nullptr, rz1 = nullptr;
  const Op* rz_op = nullptr;
  for (size_t i = 0; i < n && !rc; ++i) {
    const Op& op = h->fwd_ops[i];
    if (op.rearm && !rz_op) rz_op = &op;
    if (hipEventRecord(ev[2 * i], s) != hipSuccess) rc = fail("hipEventRecord failed");
    for (int r = 0; r < reps && !rc; ++r) {
      hipError_t e = op.rearm ? op.rearm(s) : hipSuccess;
      if (e == hipSuccess) e = op.fn(s);
      if (e != hipSuccess) rc = fail("launch of '%s' failed: %s", op.name.c_str(), hipGetErrorString(e));
    }
    if (!rc && hipEventRecord(ev[2 * i + 1], s) != hipSuccess) rc = fail("hipEventRecord failed");
  }
  if (!rc && rz_op) {
    if (hipEventCreate(&rz0) != hipSuccess || hipEventCreate(&rz1) != hipSuccess) rc = fail("hipEventCreate failed");
    if (!rc) (void)hipEventRecord(rz0, s);
    for (int r = 0; r < reps && !rc; ++r)
      if (rz_op->rearm(s) != hipSuccess) rc = fail("clear launch failed");
    if (!rc) (void)hipEventRecord(rz1, s);
  }
  if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = fail("stream sync failed: %s", hipGetErrorString(hipGetLastError()));
  float rz_ms = 0.f;
  if (!rc && rz_op && hipEventElapsedTime(&rz_ms, rz0, rz1) != hipSuccess) rc = fail("hipEventElapsedTime failed");
  for (size_t i = 0; i < n && !rc; ++i) {
    if (hipEventElapsedTime(&ms[i], ev[2 * i], ev[2 * i + 1]) != hipSuccess) rc = fail("hipEventElapsedTime failed");
    if (h->fwd_ops[i].rearm) ms[i] = std::max(ms[i] - rz_ms, 0.f);
    ms[i] /= (float)reps;
  }
  if (rz0) (void)hipEventDestroy(rz0);
  if (rz1) (void)hipEventDestroy(rz1);
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

// ---- raw helpers ----------------------------------------------------------------------
int ns2vc_dev_malloc(void** out, size_t bytes) { HIPCHK(hipMalloc(out, bytes ? bytes : 1)); return 0; }
int ns2vc_dev_free(void* p) { HIPCHK(hipFree(p)); return 0; }
int ns2vc_memcpy_h2d(void* dst, const void* src, size_t bytes) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return 0; }
int ns2vc_memcpy_d2h(void* dst, const void* src, size_t bytes) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return 0; }
int ns2vc_dev_sync(void) { HIPCHK(hipDeviceSynchronize()); return 0; }
int ns2vc_stream_create(void** out) { hipStream_t s; HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); *out = s; return 0; }
// NOTE: hipExtStreamCreateWithCUMask makes a DEFAULT (blocking) stream: it synchronises implicitly with the legacy null stream, so work
// queued on the null stream serialises with it (keep PyTorch work on explicit streams when overlapping stages).  The mask is validated against the
// device: at least one bit set, and no bit at or beyond the CU count.
int ns2vc_stream_create_cu_mask(void** out, const uint32_t* mask_words, int n_words) {
  if (!out || !mask_words || n_words <= 0) return fail("null argument");
  int dev = 0, cus = 0;
  HIPCHK(hipGetDevice(&dev));
  HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  if (n_words > (cus + 31) / 32) return fail("CU mask of %d words for a device with %d CUs (at most %d words)", n_words, cus, (cus + 31) / 32);
  bool any = false;
  for (int w = 0; w < n_words; ++w)
    for (int b = 0; b < 32; ++b)
      if (mask_words[w] >> b & 1u) {
        if (w * 32 + b >= cus) return fail("CU mask bit %d set, the device has %d CUs", w * 32 + b, cus);
        any = true;
      }
  if (!any) return fail("empty CU mask");
  hipStream_t s;
  HIPCHK(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask_words));
  *out = s;
  return 0;
}
int ns2vc_stream_destroy(void* s) { HIPCHK(hipStreamDestroy((hipStream_t)s)); return 0; }
int ns2vc_stream_sync(void* s) { HIPCHK(hipStreamSynchronize((hipStream_t)s)); return 0; }
int ns2vc_event_create(void** out) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); *out = e; return 0; }
int ns2vc_event_destroy(void* e) { HIPCHK(hipEventDestroy((hipEvent_t)e)); return 0; }
int ns2vc_event_record(void* e, void* s) { HIPCHK(hipEventRecord((hipEvent_t)e, (hipStream_t)s)); return 0; }
int ns2vc_event_elapsed_ms(void* a, void* b, float* ms) {
  HIPCHK(hipEventSynchronize((hipEvent_t)b));
  HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
  return 0;
}

// ---- kernel-level entry points -----------------------------------------------------------
int ns2vc_pack_weight(const float* rows_host, int N, int K, int precision, void** out_dev) {
  if (!rows_host || !out_dev || N <= 0 || K <= 0) return fail("bad argument");
  static bool inited = false;
  if (!inited) {
    hipError_t e = init_gemm_attributes();
  if (e == hipSuccess) e = init_convts_attributes();
    if (e == hipSuccess) e = init_attn_attributes();
    if (e == hipSuccess) e = init_ffn_attributes();
    if (e == hipSuccess) e = init_geglu_attributes();
    if (e == hipSuccess) e = init_rowchain_attributes();
    if (e != hipSuccess) return fail("kernel attribute setup failed: %s", hipGetErrorString(e));
    inited = true;
  }
  void* d = nullptr;
  if (precision != NS2VC_PREC_F32) {
    std::vector<uint16_t> q((size_t)N * K);
    for (size_t i = 0; i < q.size(); ++i) q[i] = f32_to_op16_bits(rows_host[i], precision);
    HIPCHK(hipMalloc(&d, q.size() * 2));
    HIPCHK(hipMemcpy(d, q.data(), q.size() * 2, hipMemcpyHostToDevice));
  } else {
    HIPCHK(hipMalloc(&d, (size_t)N * K * 4));
    HIPCHK(hipMemcpy(d, rows_host, (size_t)N * K * 4, hipMemcpyHostToDevice));
  }
  *out_dev = d;
  return 0;
}
int ns2vc_pack_conv3_tiled(const float* rows_host, int N, int ctot, int c2, int precision, void** out_dev) {
  if (!rows_host || !out_dev || N <= 0) return fail("bad argument");
  std::vector<unsigned char> img;
  if (pack_conv3_tiled(rows_host, N, ctot, c2, precision, img) != hipSuccess) return fail("pack_conv3_tiled: ctot / c2 must be multiples of the chunk (64 elements; 32 for fp32)");
  void* d = nullptr;
  HIPCHK(hipMalloc(&d, img.size()));
  HIPCHK(hipMemcpy(d, img.data(), img.size(), hipMemcpyHostToDevice));
  *out_dev = d;
  return 0;
}
int ns2vc_weight_rowsum(const float* rows_host, int N, int K, int precision, float** out_dev) {
  if (!rows_host || !out_dev || N <= 0 || K <= 0) return fail("bad argument");
  const std::vector<float> ws = rounded_rowsum(rows_host, N, K, N, precision);
  void* d = nullptr;
  HIPCHK(hipMalloc(&d, ws.size() * sizeof(float)));
  HIPCHK(hipMemcpy(d, ws.data(), ws.size() * sizeof(float), hipMemcpyHostToDevice));
  *out_dev = (float*)d;
  return 0;
}
int ns2vc_debug_set_attn_optimistic(int on) {
  set_attn_optimistic(on);
  return 0;
}
int ns2vc_debug_set_gemm_trace(void* dev_u64_blocks_x8) {
  set_gemm_trace((unsigned long long*)dev_u64_blocks_x8);
  set_ffn_trace((unsigned long long*)dev_u64_blocks_x8);
  set_rc_trace((unsigned long long*)dev_u64_blocks_x8);
  set_ts_trace((unsigned long long*)dev_u64_blocks_x8);     // (convts.hip and geglu.hip write 16 words per block)
  set_gg_trace((unsigned long long*)dev_u64_blocks_x8);
  return 0;
}
int ns2vc_debug_poison(unsigned pattern, int lds_bytes, void* stream) {
  static unsigned* sink = nullptr;
  if (lds_bytes < 4 || lds_bytes > 160 * 1024) return fail("poison: LDS bytes out of range");
  if (!sink) HIPCHK(hipMalloc((void**)&sink, 64));
  hipError_t e = launch_poison(pattern, lds_bytes & ~3, sink, (hipStream_t)stream);
  if (e != hipSuccess) return fail("launch_poison: %s", hipGetErrorString(e));
  return 0;
}
int ns2vc_debug_placement(void* stream, int n_blocks, int spin, uint32_t* out_host) {
  if (!out_host || n_blocks <= 0 || n_blocks > (1 << 20)) return fail("bad argument");
  unsigned* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, (size_t)n_blocks * 2 * sizeof(unsigned)));
  hipError_t e = launch_placement(d, n_blocks, spin, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e == hipSuccess) e = hipMemcpy(out_host, d, (size_t)n_blocks * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail("placement probe: %s", hipGetErrorString(e));
  return 0;
}
int ns2vc_debug_set_gemm_tile(int bm, int bn, int stages) { set_forced_gemm_tile(bm, bn, stages); return 0; }
int ns2vc_k_gemm(const ns2vc_gemm_args* a, int precision, void* stream) {
  if (!a) return fail("null args");
  hipError_t e = launch_gemm(*a, precision, (hipStream_t)stream);
  if (e != hipSuccess) {
    const int line = last_gemm_refusal_line();
    if (line) return fail("launch_gemm: %s (refused by the argument check at gemm.hip:%d)", hipGetErrorString(e), line);
    return fail("launch_gemm: %s", hipGetErrorString(e));
  }
  return 0;
}
int ns2vc_pack_ffn(const float* w1_packed_host, const float* w2f_host, int dim, int precision, void** out_stream_dev) {
  if (!w1_packed_host || !w2f_host || !out_stream_dev) return fail("null argument");
  static bool inited = false;
  if (!inited) {
    hipError_t e = init_ffn_attributes();
    if (e != hipSuccess) return fail("kernel attribute setup failed: %s", hipGetErrorString(e));
    inited = true;
  }
  std::vector<unsigned short> st;
  if (pack_ffn_stream(w1_packed_host, w2f_host, nullptr, dim, precision, st) != hipSuccess) return fail("ffn: dim must be 128 or 256 and the precision 16-bit");
  void* d = nullptr;
  HIPCHK(hipMalloc(&d, st.size() * 2));
  HIPCHK(hipMemcpy(d, st.data(), st.size() * 2, hipMemcpyHostToDevice));
  *out_stream_dev = d;
  return 0;
}
int ns2vc_pack_ffn_pre(const float* w1_packed_host, const float* w2f_host, const float* w0_host, int dim, int precision, void** out_stream_dev) {
  if (!w1_packed_host || !w2f_host || !w0_host || !out_stream_dev) return fail("null argument");
  hipError_t e = init_ffn_attributes();
  if (e != hipSuccess) return fail("kernel attribute setup failed: %s", hipGetErrorString(e));
  std::vector<unsigned short> st;
  if (pack_ffn_stream(w1_packed_host, w2f_host, w0_host, dim, precision, st) != hipSuccess) return fail("ffn: dim must be 128 or 256 and the precision 16-bit");
  void* d = nullptr;
  HIPCHK(hipMalloc(&d, st.size() * 2));
  HIPCHK(hipMemcpy(d, st.data(), st.size() * 2, hipMemcpyHostToDevice));
  *out_stream_dev = d;
  return 0;
}
int ns2vc_pack_rowchain(const float* w1_host, const float* w2_host, int dim, int n2, int precision, void** out_stream_dev) {
  if (!w1_host || !w2_host || !out_stream_dev) return fail("null argument");
  static bool inited = false;
  if (!inited) {
    hipError_t e = init_rowchain_attributes();
    if (e != hipSuccess) return fail("kernel attribute setup failed: %s", hipGetErrorString(e));
    inited = true;
  }
  std::vector<unsigned short> st;
  if (pack_rowchain_stream(w1_host, w2_host, dim, n2, precision, st) != hipSuccess)
    return fail("rowchain: dim must be 128, 256 or 384, n2 = dim or 3 dim, and the precision 16-bit");
  void* d = nullptr;
  HIPCHK(hipMalloc(&d, st.size() * 2));
  HIPCHK(hipMemcpy(d, st.data(), st.size() * 2, hipMemcpyHostToDevice));
  *out_stream_dev = d;
  return 0;
}
int ns2vc_pack_rowchain_sliced(const float* w1_host, const float* w2_host, int dim, int n2, int slices, int precision, void** out_stream_dev) {
  if (!w1_host || !w2_host || !out_stream_dev) return fail("null argument");
  static bool inited = false;
  if (!inited) {
    hipError_t e = init_rowchain_attributes();
    if (e != hipSuccess) return fail("kernel attribute setup failed: %s", hipGetErrorString(e));
    inited = true;
  }
  std::vector<unsigned short> st;
  if (pack_rowchain_stream(w1_host, w2_host, dim, n2, precision, st, slices) != hipSuccess)
    return fail("rowchain (sliced): dim 384, n2 = dim or 3 dim, 2 slices, 16-bit precision");
  void* d = nullptr;
  HIPCHK(hipMalloc(&d, st.size() * 2));
  HIPCHK(hipMemcpy(d, st.data(), st.size() * 2, hipMemcpyHostToDevice));
  *out_stream_dev = d;
  return 0;
}
int ns2vc_debug_set_attn_keys(int keys) {
  if (keys != 0 && keys != 64 && keys != 128) return fail("attention K/V tile: 0 (heuristic), 64 or 128 keys");
  set_forced_attn_keys(keys);
  return 0;
}
int ns2vc_debug_set_rowchain_tokens(int nt) {
  if (nt < 0 || nt > 2) return fail("rowchain tokens: 0 (heuristic), 1 (64-token blocks) or 2 (128-token blocks)");
  set_forced_rowchain_tokens(nt);
  return 0;
}
int ns2vc_k_rowchain(const ns2vc_rowchain_args* a, int precision, void* stream) {
  if (!a) return fail("null args");
  hipError_t e = launch_rowchain(*a, precision, (hipStream_t)stream);
  if (e != hipSuccess) return fail("launch_rowchain: %s", hipGetErrorString(e));
  return 0;
}
size_t ns2vc_xattn_pack_bytes(int B, int Lk, int hd) { return xattn_pack_bytes(B, Lk, hd); }
int ns2vc_k_xattn_pack(const void* k, int ldk, const void* v, int ldv, int B, int Lk, int hd, void* out, int precision, void* stream) {
  hipError_t e = launch_xattn_pack(k, ldk, v, ldv, B, Lk, hd, out, precision, (hipStream_t)stream);
  if (e != hipSuccess) return fail("xattn_pack: %s (16-bit precisions, head dim 16 | 32, 8 heads, 16-byte aligned k rows)", hipGetErrorString(e));
  return 0;
}
int ns2vc_k_ffn(const ns2vc_ffn_args* a, int precision, void* stream) {
  if (!a) return fail("null args");
  hipError_t e = launch_ffn(*a, precision, (hipStream_t)stream);
  if (e != hipSuccess) return fail("launch_ffn: %s", hipGetErrorString(e));
  return 0;
}
int ns2vc_pack_geglu(const float* w1_packed_host, const float* bias1_packed_host, int dim, int precision, void** out_stream_dev, float** out_consts_dev) {
  if (!w1_packed_host || !out_stream_dev || !out_consts_dev) return fail("null argument");
  hipError_t e = init_geglu_attributes();
  if (e != hipSuccess) return fail("kernel attribute setup failed: %s", hipGetErrorString(e));
  std::vector<unsigned short> st;
  std::vector<float> cs;
  if (pack_geglu_stream(w1_packed_host, bias1_packed_host, dim, precision, st, cs) != hipSuccess) return fail("geglu: dim must be 384 and the precision 16-bit");
  void* d = nullptr;
  float* c = nullptr;
  HIPCHK(hipMalloc(&d, st.size() * 2));
  HIPCHK(hipMemcpy(d, st.data(), st.size() * 2, hipMemcpyHostToDevice));
  HIPCHK(hipMalloc((void**)&c, cs.size() * 4));
  HIPCHK(hipMemcpy(c, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
  *out_stream_dev = d;
  *out_consts_dev = c;
  return 0;
}
int ns2vc_debug_set_geglu_min_rows(int rows) { g_geglu_min_rows = rows < 0 ? 4608 : rows; return 0; }
int ns2vc_pack_geglu_host(const float* w1_packed_host, const float* bias1_packed_host, int dim, int precision, uint16_t* stream_out, float* consts_out) {
  if (!w1_packed_host || !stream_out || !consts_out) return fail("null argument");
  std::vector<unsigned short> st;
  std::vector<float> cs;
  if (pack_geglu_stream(w1_packed_host, bias1_packed_host, dim, precision, st, cs) != hipSuccess) return fail("geglu: dim must be 384 and the precision 16-bit");
  memcpy(stream_out, st.data(), st.size() * 2);
  memcpy(consts_out, cs.data(), cs.size() * 4);
  return 0;
}
int ns2vc_k_geglu(const ns2vc_geglu_args* a, int precision, void* stream) {
  if (!a) return fail("null args");
  hipError_t e = launch_geglu(*a, precision, (hipStream_t)stream);
  if (e != hipSuccess) return fail("launch_geglu: %s", hipGetErrorString(e));
  return 0;
}
int ns2vc_k_attention(const ns2vc_attn_args* a, int head_dim, int precision, void* stream) {
  if (!a) return fail("null args");
  hipError_t e = launch_attention(*a, head_dim, precision, (hipStream_t)stream);
  if (e != hipSuccess) return fail("launch_attention: %s", hipGetErrorString(e));
  return 0;
}
int ns2vc_k_groupnorm(const float* a0, int lda0, int c0, const float* a1, int lda1, int c1, int B, int T, int G, float eps,
                      const float* gamma, const float* beta, const float* temb, int ldtemb, int temb_off, int silu, void* out_op,
                      void* raw_op, int precision, void* stream) {
  const int rows = 32, nchunk = (T + rows - 1) / rows;
  double* part = nullptr;
  HIPCHK(hipMalloc((void**)&part, (size_t)B * nchunk * G * 2 * sizeof(double)));
  hipError_t e = launch_gn_partial(a0, lda0, c0, a1, lda1, c1, B, T, G, part, nchunk, rows, (hipStream_t)stream);
  if (e == hipSuccess)
    e = launch_gn_apply(a0, lda0, c0, a1, lda1, c1, B, T, G, eps, part, nchunk, nullptr, nullptr, gamma, beta, temb, ldtemb, temb_off, silu,
                        out_op, raw_op, precision, (hipStream_t)stream);
  hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(part);
  if (e != hipSuccess) return fail("groupnorm launch: %s", hipGetErrorString(e));
  if (e2 != hipSuccess) return fail("groupnorm sync: %s", hipGetErrorString(e2));
  return 0;
}
int ns2vc_k_groupnorm_stats(const float* a0, int lda0, int c0, const long long* stats0, int B, int T, int G, float eps, const float* gamma,
                            const float* beta, const float* temb, int ldtemb, int temb_off, int silu, void* out_op, int precision, void* stream) {
  if (!stats0 || (c0 % G) || ((c0 / G) & 15)) return fail("groupnorm_stats: needs the int64 epilogue statistics and groups of whole 16-channel blocks");
  hipError_t e = launch_gn_apply(a0, lda0, c0, nullptr, 0, 0, B, T, G, eps, nullptr, 0, stats0, nullptr, gamma, beta, temb, ldtemb, temb_off, silu,
                                 out_op, nullptr, precision, (hipStream_t)stream);
  if (e != hipSuccess) return fail("groupnorm_stats launch: %s", hipGetErrorString(e));
  return 0;
}
int ns2vc_k_groupnorm_stats2(const float* a0, int lda0, int c0, const long long* stats0, const float* a1, int lda1, int c1, const long long* stats1,
                             int B, int T, int G, float eps, const float* gamma, const float* beta, const float* temb, int ldtemb, int temb_off,
                             int silu, void* out_op, void* raw_op, int precision, void* stream) {
  if (!stats0 || !a1 || !stats1 || (c0 & 15) || (c1 & 15) || ((c0 + c1) % G) || (((c0 + c1) / G) & 15))
    return fail("groupnorm_stats2: needs both sources' int64 epilogue statistics, whole 16-channel blocks per source and per group");
  hipError_t e = launch_gn_apply(a0, lda0, c0, a1, lda1, c1, B, T, G, eps, nullptr, 0, stats0, stats1, gamma, beta, temb, ldtemb, temb_off, silu,
                                 out_op, raw_op, precision, (hipStream_t)stream);
  if (e != hipSuccess) return fail("groupnorm_stats2 launch: %s", hipGetErrorString(e));
  return 0;
}
int ns2vc_k_layernorm_apply(const float* x, int ldx, int M, int C, float eps, void* out_op, int precision, void* stream) {
  hipError_t e = launch_ln_apply_op(x, ldx, M, C, eps, out_op, precision, (hipStream_t)stream);
  if (e != hipSuccess) return fail("ln_apply launch: %s", hipGetErrorString(e));
  return 0;
}
int ns2vc_to_operand(const float* host, size_t n, int precision, void** out_dev) {
  if (!host || !out_dev) return fail("null argument");
  void* d = nullptr;
  if (precision != NS2VC_PREC_F32) {
    std::vector<uint16_t> q(n);
    for (size_t i = 0; i < n; ++i) q[i] = f32_to_op16_bits(host[i], precision);
    HIPCHK(hipMalloc(&d, std::max<size_t>(n, 1) * 2));
    HIPCHK(hipMemcpy(d, q.data(), n * 2, hipMemcpyHostToDevice));
  } else {
    HIPCHK(hipMalloc(&d, std::max<size_t>(n, 1) * 4));
    HIPCHK(hipMemcpy(d, host, n * 4, hipMemcpyHostToDevice));
  }
  *out_dev = d;
  return 0;
}
int ns2vc_round_to_operand(const float* host_in, size_t n, int precision, float* host_out) {
  if (!host_in || !host_out) return fail("null argument");
  if (precision != NS2VC_PREC_F32 && precision != NS2VC_PREC_BF16 && precision != NS2VC_PREC_F16) return fail("unknown precision %d", precision);
  for (size_t i = 0; i < n; ++i)
    host_out[i] = precision == NS2VC_PREC_F32 ? host_in[i] : op16_bits_to_f32(f32_to_op16_bits(host_in[i], precision), precision);
  return 0;
}
int ns2vc_from_operand(const void* dev, size_t n, int precision, float* host) {
  if (!dev || !host) return fail("null argument");
  HIPCHK(hipDeviceSynchronize());
  if (precision != NS2VC_PREC_F32) {
    std::vector<uint16_t> q(n);
    HIPCHK(hipMemcpy(q.data(), dev, n * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) host[i] = op16_bits_to_f32(q[i], precision);
  } else {
    HIPCHK(hipMemcpy(host, dev, n * 4, hipMemcpyDeviceToHost));
  }
  return 0;
}
int ns2vc_k_nct_to_btc(const float* src, int C, int T, int B, float* dst, int ldd, int cpad, void* stream) {
  hipError_t e = launch_nct_to_btc(src, C, T, B, dst, nullptr, PREC_F32, ldd, cpad, (hipStream_t)stream);
  if (e != hipSuccess) return fail("nct_to_btc launch: %s", hipGetErrorString(e));
  return 0;
}
int ns2vc_k_btc_to_nct(const float* src, int lds, int C, int T, int B, float* dst, void* stream) {
  hipError_t e = launch_btc_to_nct(src, lds, C, T, B, dst, (hipStream_t)stream);
  if (e != hipSuccess) return fail("btc_to_nct launch: %s", hipGetErrorString(e));
  return 0;
}

}  // extern "C"
