// Whole-model DiT forward / backward: the launch sequence over the kernels of this library.
//
// Replaces autograd over /root/reference/DiT/models.py:233-248 (DiT.forward) for the training step of
// DiT/forget.py:271-288,310-319.  One call enqueues the whole pass on a stream; nothing is allocated
// here (the caller owns the parameter arenas and one workspace), so the sequence is graph-capturable.
//
// Parameter arena (fp32 master + bf16 shadow + grads share offsets), canonical order:
//   patch-embed W,b | t_embedder W0,b0,W2,b2 | label table | adaLN weights of ALL blocks + final layer
//   contiguous [(6L+2)D][D] | their biases [(6L+2)D] | per block: qkv W,b, proj W,b, fc1 W,b, fc2 W,b |
//   final linear W,b | (frozen) pos_embed.
// The adaLN weights are contiguous so that the modulation of every block is ONE GEMM
// [B,D]x[D,(6L+2)D]: c = t_emb + y_emb is the same for all blocks (models.py:243-246).
#include <mutex>
#include <vector>
#include "common.h"
#include "../../include/sfron.h"

namespace {

struct Dims {
  int B, C, S, p, D, L, H, hd, F, ncls, fdim, Co;   // F = mlp hidden, Co = out channels
  int g, T, M, Kp, Po, NM;                          // g = S/p, T tokens, M = B*T, Kp = C*p*p, Po = p*p*Co, NM = (6L+2)*D
};

inline int make_dims(const sfron_dit_cfg* c, Dims& d) {
  if (!c) return SFRON_ERR_ARG;
  d.B = c->batch; d.C = c->in_channels; d.S = c->input_size; d.p = c->patch; d.D = c->hidden; d.L = c->depth;
  d.H = c->heads; d.F = c->mlp_hidden; d.ncls = c->num_classes; d.fdim = c->freq_dim; d.Co = c->out_channels;
  if (d.B <= 0 || d.C <= 0 || d.S <= 0 || d.p <= 0 || d.D <= 0 || d.L <= 0 || d.H <= 0 || d.F <= 0 || d.ncls <= 0) return SFRON_ERR_ARG;
  if (d.S % d.p || d.D % d.H || d.D % 8 || d.F % 8 || d.fdim % 8) return SFRON_ERR_ARG;
  d.hd = d.D / d.H; d.g = d.S / d.p; d.T = d.g * d.g; d.M = d.B * d.T; d.Kp = d.C * d.p * d.p; d.Po = d.p * d.p * d.Co;
  d.NM = (6 * d.L + 2) * d.D;
  // token counts: multiples of 64 (tiled attention) or fewer than 64 (the registry's patch-8 models at 256 px: 16 tokens -- plain-FMA
  // attention kernels, generic GEMM tiles); T must be even for the row-kernel chunking
  if (d.Kp % 8 || d.Po % 8 || (d.T % 64 && d.T > 64) || (d.T & 1)) return SFRON_ERR_UNSUPPORTED;
  // head widths the attention kernels take: multiples of 8 up to 80 on the tiled kernels (DiT 64 / 72), anything up to 128 on the
  // short-sequence kernels
  if (d.T >= 64 ? (d.hd % 8 || d.hd > 80) : d.hd > 128) return SFRON_ERR_UNSUPPORTED;
  return SFRON_OK;
}

// ---- parameter layout ---------------------------------------------------------------------------
struct ParamLayout {
  int64_t pe_w, pe_b, t0_w, t0_b, t2_w, t2_b, table, ada_w, ada_b, blocks, blk_stride, fin_w, fin_b, pos, total, trainable;
  int64_t o_qkv_w, o_qkv_b, o_proj_w, o_proj_b, o_fc1_w, o_fc1_b, o_fc2_w, o_fc2_b;   // offsets inside a block
};
inline ParamLayout make_layout(const Dims& d) {
  ParamLayout p{};
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += (n + 7) / 8 * 8; return r; };
  p.pe_w = take((int64_t)d.D * d.Kp); p.pe_b = take(d.D);
  p.t0_w = take((int64_t)d.D * d.fdim); p.t0_b = take(d.D);
  p.t2_w = take((int64_t)d.D * d.D); p.t2_b = take(d.D);
  p.table = take((int64_t)(d.ncls + 1) * d.D);
  p.ada_w = take((int64_t)d.NM * d.D); p.ada_b = take(d.NM);
  p.blocks = o;
  int64_t b0 = o;
  p.o_qkv_w = take((int64_t)3 * d.D * d.D) - b0; p.o_qkv_b = take(3 * d.D) - b0;
  p.o_proj_w = take((int64_t)d.D * d.D) - b0; p.o_proj_b = take(d.D) - b0;
  p.o_fc1_w = take((int64_t)d.F * d.D) - b0; p.o_fc1_b = take(d.F) - b0;
  p.o_fc2_w = take((int64_t)d.D * d.F) - b0; p.o_fc2_b = take(d.D) - b0;
  p.blk_stride = o - b0;
  o = b0 + p.blk_stride * d.L;
  p.fin_w = take((int64_t)d.Po * d.D); p.fin_b = take(d.Po);
  p.trainable = o;
  p.pos = take((int64_t)d.T * d.D);
  p.total = o;
  return p;
}

// ---- workspace layout ---------------------------------------------------------------------------
struct Workspace {
  // saved by forward, read by backward
  __bf16 *patches, *tfreq, *h1s, *sc;
  float *h1, *temb, *c, *mod;
  float* xs;                 // [2L+1][M][D] residual stream checkpoints
  __bf16 *xmod1, *qkv, *o, *a1, *xmod2, *hpre, *h, *a2;   // per block, stride below
  float *mean, *rstd;        // [2L+1][M]
  float* lse;                // [L][B*H*T]
  __bf16* xmodf;
  float* tok;
  // backward temporaries
  float *dx, *dmod, *d_sc, *d_c, *d_h1s, *delta, *part, *csum, *csum2, *slabs, *wslab, *dysum, *bpart, *bpart_qkv;
  float* kslab;              // split-K slabs of the few-tile products of a small batch * tokens (small_m_splits), or null
  __bf16 *d_tok, *d_xmod, *d_o, *dmod_bf, *d_c_bf, *d_h1_bf, *dx_bf;
  __bf16 *d_br[2], *d_br2[2], *d_hpre[2], *dqkv[2];   // read by the side stream: double-buffered by block parity so the
                                                       // dgrad chain may run one block ahead of the weight gradients
  size_t bytes;
};
constexpr int SPLIT_K_ADA = 256;   // adaLN dgrad reads 451 MB of weights with M = batch rows: 64 splits ran at 2.5 TB/s, 256 at ~4.5
constexpr int CSUM_PARTS = 64;     // row chunks of the bias column sums (stage 2 reads CSUM_PARTS x N floats with N/256 workgroups)

// Weight gradients dW[N][K] = dY[M][N]^T X[M][K] have few 192x192 output tiles and a long reduction (M = batch*tokens):
// a split of the reduction offers about 150-250 workgroups per GEMM.  Kept as an option (see wgrad_side): 1 = no split.
inline int wgrad_splits(int N, int K, int Mred) {
  if (N % 192 || K % 192 || Mred % 64) return 1;
  const int tiles = (N / 192) * (K / 192);
  int s = 224 / tiles;
  if (s > 4) s = 4;
  while (s > 1 && (Mred / 64) % s) --s;
  return s < 1 ? 1 : s;
}

// Few-tile products: a [M x N] output with M = batch * tokens small offers the pipelined 256 x 192 tile fewer than 128 workgroups (DiT-B/4 at
// batch 32: [2048 x 768] = 32 tiles on 256 CUs -- the dispatcher then takes 96 tiles of the register-staged 128 x 128 kernel at 37 us).  Instead the
// contraction is split over S workgroups per tile (each an even number >= 2 of 64-deep K-tiles: the interleaved schedule), S fp32 slabs are summed in
// index order by a finish kernel that also applies the epilogue (norm.hip k_split_sum_bf16 / k_split_gate_res).  1 = no split (every DiT-XL/2 shape).
#ifndef SFRON_SMALL_M_CAP
#define SFRON_SMALL_M_CAP 4      // same-box A-B at DiT-B/4 batch 32: cap 8 9.88 / 9.81 ms per step, 6 9.70 / 9.87, 4 9.54 / 9.64, 3 9.63 / 9.48, 2 10.30 / 10.21 (the slabs are traffic too)
#endif
#ifndef SFRON_SMALL_M_MIN_K
#define SFRON_SMALL_M_MIN_K 768   // a 12-K-tile product (proj forward / dgrad at D = 768) is faster as ONE launch: 15 us against 22-25 with its finish
#endif
inline int small_m_splits(int M, int N, int K) {
  if (M % 256 || N % 192 || K % 128) return 1;
  if (N % 144 == 0 && K % 192 == 0 && (M / 256) * (N / 144) >= 128) return 1;     // the 256 x 144 three-slot tile fills half the chip by itself
  const int tiles = (M / 256) * (N / 192);
  if (tiles >= 128) return 1;
  if (K <= SFRON_SMALL_M_MIN_K) return 1;
  int best = 1;
  const int kt = K / 64;
  for (int s = 2; s <= SFRON_SMALL_M_CAP; ++s)
    if (kt % s == 0 && (kt / s) % 2 == 0 && tiles * s <= 256) best = s;
  return best;
}

inline Workspace make_ws(const Dims& d, char* base) {
  Workspace w{};
  size_t o = 0;
  auto take = [&](size_t bytes) { char* r = base ? base + o : nullptr; o += (bytes + 255) / 256 * 256; return r; };
  const size_t M = d.M, D = d.D, L = d.L, B = d.B;
  w.patches = (__bf16*)take(M * d.Kp * 2);
  w.tfreq = (__bf16*)take(B * d.fdim * 2);
  w.h1 = (float*)take(B * D * 4); w.h1s = (__bf16*)take(B * D * 2);
  w.temb = (float*)take(B * D * 4); w.c = (float*)take(B * D * 4); w.sc = (__bf16*)take(B * D * 2);
  w.mod = (float*)take(B * (size_t)d.NM * 4);
  w.xs = (float*)take((2 * L + 1) * M * D * 4);
  w.xmod1 = (__bf16*)take(L * M * D * 2); w.qkv = (__bf16*)take(L * M * 3 * D * 2); w.o = (__bf16*)take(L * M * D * 2);
  w.a1 = (__bf16*)take(L * M * D * 2); w.xmod2 = (__bf16*)take(L * M * D * 2);
  w.hpre = (__bf16*)take(L * M * (size_t)d.F * 2); w.h = (__bf16*)take(L * M * (size_t)d.F * 2);
  w.a2 = (__bf16*)take(L * M * D * 2);
  w.mean = (float*)take((2 * L + 1) * M * 4); w.rstd = (float*)take((2 * L + 1) * M * 4);
  w.lse = (float*)take(L * B * (size_t)d.H * d.T * 4);
  w.xmodf = (__bf16*)take(M * D * 2);
  w.tok = (float*)take(M * (size_t)d.Po * 4);
  // backward
  w.dx = (float*)take(M * D * 4);
  w.dmod = (float*)take(B * (size_t)d.NM * 4); w.dmod_bf = (__bf16*)take(B * (size_t)d.NM * 2);
  w.d_sc = (float*)take(B * D * 4); w.d_c = (float*)take(B * D * 4); w.d_c_bf = (__bf16*)take(B * D * 2);
  w.d_h1s = (float*)take(B * D * 4); w.d_h1_bf = (__bf16*)take(B * D * 2);
  w.delta = (float*)take(B * (size_t)d.H * d.T * 4);
  w.part = (float*)take((L + 1) * 8 * (size_t)(M / sfron_rows_per_chunk(d.T)) * D * 4);   // [L+1][kind 4][buf 2][chunks][D] partial slots
  const size_t widest = (size_t)(d.F > 3 * d.D ? d.F : 3 * d.D) > (size_t)d.NM ? (size_t)(d.F > 3 * d.D ? d.F : 3 * d.D) : (size_t)d.NM;
  w.csum = (float*)take(CSUM_PARTS * widest * 4);
  w.csum2 = (float*)take(CSUM_PARTS * (size_t)(d.F > 3 * d.D ? d.F : 3 * d.D) * 4);   // side-stream colsum scratch
  w.slabs = (float*)take((size_t)SPLIT_K_ADA * B * D * 4);
  {                                                                       // split-K slabs of the block weight gradients
    const size_t shapes[4][2] = {{3 * D, D}, {D, D}, {(size_t)d.F, D}, {D, (size_t)d.F}};
    size_t mx = 0;
    for (auto& sh : shapes) { const size_t n = wgrad_splits((int)sh[0], (int)sh[1], (int)M) * sh[0] * sh[1]; mx = n > mx ? n : mx; }
    w.wslab = (float*)take(mx * 4);
  }
  {
    // the [M x D] outputs that may split: forward proj (K = D) / fc2 (K = F), dgrad qkv (K = 3D) / proj (K = D) / fc1 (K = F)
    int smax = 1;
    for (const int k : {d.D, d.F, 3 * d.D}) { const int sp = small_m_splits(d.M, d.D, k); smax = sp > smax ? sp : smax; }
    w.kslab = smax > 1 ? (float*)take((size_t)smax * M * D * 4) : nullptr;
  }
  w.dysum = (float*)take((L + 1) * 2 * B * D * 4);                       // [L][proj|fc2][B][D] token sums of dy (gated bias grads)
  // per-tile-row partials of the fc1 / qkv bias gradients, written by the kernels that PRODUCE d_hpre / dqkv (double-buffered
  // by block parity like those tensors: the side stream reduces them one block behind)
  w.bpart = (float*)take(2 * (size_t)((M + 255) / 256) * (size_t)d.F * 4);        // [parity][M / 256][F]
  w.bpart_qkv = (float*)take(2 * B * 3 * D * 4);                                  // [parity][B][3 D]
  w.d_tok = (__bf16*)take(M * (size_t)d.Po * 2);
  for (int i = 0; i < 2; ++i) {
    w.d_br[i] = (__bf16*)take(M * D * 2); w.d_br2[i] = (__bf16*)take(M * D * 2);
    w.d_hpre[i] = (__bf16*)take(M * (size_t)d.F * 2); w.dqkv[i] = (__bf16*)take(M * 3 * D * 2);
  }
  w.d_xmod = (__bf16*)take(M * D * 2); w.d_o = (__bf16*)take(M * D * 2); w.dx_bf = (__bf16*)take(M * D * 2);
  w.bytes = o;
  return w;
}

#define RUN(expr) do { int rc__ = (expr); if (rc__ != SFRON_OK) return rc__; } while (0)

inline sfron_gemm_desc gd(const void* A, int lda, const void* B, int ldb, int M, int N, int K) {
  sfron_gemm_desc g{};
  g.A = (const uint16_t*)A; g.B = (const uint16_t*)B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
  g.alpha = 1.0f; g.tokens = 1; g.split_k = 1;
  return g;
}
// Y[M,N] = X[M,K] W[N,K]^T
inline sfron_gemm_desc fwd_desc(const void* X, const void* W, int M, int N, int K) { return gd(X, K, W, K, M, N, K); }
// dX[M,K] = dY[M,N] W[N,K]
inline sfron_gemm_desc dgrad_desc(const void* dY, const void* W, int M, int N, int K) {
  sfron_gemm_desc g = gd(dY, N, W, K, M, K, N);
  g.b_transposed = 1;
  return g;
}
// dW[N,K] = dY[M,N]^T X[M,K] -> fp32 into the grad arena
inline sfron_gemm_desc wgrad_desc(const void* dY, const void* X, int M, int N, int K, float* dW) {
  sfron_gemm_desc g = gd(dY, N, X, K, N, K, M);
  g.a_transposed = 1; g.b_transposed = 1; g.epilogue = SFRON_EPI_F32; g.c_f32 = dW; g.ldc_f32 = K;
  return g;
}

}  // namespace

extern "C" {

int sfron_dit_param_layout(const sfron_dit_cfg* cfg, int64_t* out, int n_out) {
  Dims d;
  RUN(make_dims(cfg, d));
  ParamLayout p = make_layout(d);
  const int64_t v[SFRON_DIT_LAYOUT_LEN] = {p.total, p.trainable, p.pe_w, p.pe_b, p.t0_w, p.t0_b, p.t2_w, p.t2_b, p.table,
                                           p.ada_w, p.ada_b, p.blocks, p.blk_stride, p.o_qkv_w, p.o_qkv_b, p.o_proj_w,
                                           p.o_proj_b, p.o_fc1_w, p.o_fc1_b, p.o_fc2_w, p.o_fc2_b, p.fin_w, p.fin_b, p.pos};
  SFRON_CHECK_ARG(out && n_out >= SFRON_DIT_LAYOUT_LEN);
  for (int i = 0; i < SFRON_DIT_LAYOUT_LEN; ++i) out[i] = v[i];
  return SFRON_OK;
}

int64_t sfron_dit_workspace_bytes(const sfron_dit_cfg* cfg) {
  Dims d;
  if (make_dims(cfg, d) != SFRON_OK) return -1;
  return (int64_t)make_ws(d, nullptr).bytes;
}

// SFRON_ABLATE (debug / A-B measurement only): bit 0 = run LN backward and gate backward as separate kernels,
// bit 1 = split-K in the block weight gradients (off by default), bit 2 = side stream at the lowest priority, bit 3 = fc2 weight gradient after the fc2 dgrad, bit 4 = no split-K for the two small-output weight gradients, bit 5 = bias row sums inside the weight-gradient GEMM, bit 6 = fc1 bias gradient by a separate column-sum launch, bit 7 = qkv bias gradient likewise, bit 8 = proj weight gradient before (not beside) the attention backward, bit 9 = fc1 / fc2 weight gradients on 256 x 192 tiles (108 workgroups)
static int ablate_mask() {
#ifdef SFRON_DEBUG_KNOBS
  static const int m = [] { const char* e = getenv("SFRON_ABLATE"); return e ? atoi(e) : 0; }();
  return m;
#else
  return 0;          // the product build reads no environment variables
#endif
}

// ---- aux: a side stream + events so the weight-gradient GEMMs (which nothing downstream in the backward chain
// depends on) run concurrently with the dgrad / elementwise chain and fill the CUs its tile counts leave idle
struct Probe;
struct Aux { hipStream_t side, side2; hipEvent_t produced[4], consumed[8], done, join2, ada_ready, ada_factors; Probe* probe; const uint8_t* sq_mask; double* sq_partials; int dev; };

// The two weight-gradient streams of a handle come from a per-device FREE LIST and go back to it when the handle is destroyed: a process
// that builds one engine after another (bench.py's configuration legs, set_batch_size(), a test session) keeps running on the SAME
// two HIP streams.  Measured (profiles/r06_fp8_leg.txt): the runtime maps streams onto four hardware queues; with streams created and
// destroyed per engine, the third runner of a process got its weight-gradient stream onto a queue it shares with another stream of the
// step and ran 4 ms / step slower (DiT-XL/2 fp8: 63.8 ms against 59.3 as the first runner).  Two LIVE handles never share a pair.
namespace {
struct StreamPair { int dev; hipStream_t a, b; };
std::mutex g_pairs_mu;
std::vector<StreamPair> g_free_pairs;
}  // namespace

int sfron_aux_create(void** aux) {
  SFRON_CHECK_ARG(aux);
  Aux* a = new Aux{};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
  a->dev = dev;
  bool reused = false;
  if (!(ablate_mask() & 4)) {
    std::lock_guard<std::mutex> lk(g_pairs_mu);
    for (size_t i = 0; i < g_free_pairs.size(); ++i)
      if (g_free_pairs[i].dev == dev) {
        a->side = g_free_pairs[i].a; a->side2 = g_free_pairs[i].b;
        g_free_pairs.erase(g_free_pairs.begin() + i);
        reused = true;
        break;
      }
  }
  // equal priority with the caller's stream measured best (89.8 ms/step; lowest priority 92.8, highest 95.1)
  if (reused) {
  } else if (ablate_mask() & 4) {       // A-B knob: side stream at the lowest priority
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&a->side, hipStreamNonBlocking, lo) != hipSuccess) return (int)hipGetLastError();
  } else if (hipStreamCreateWithFlags(&a->side, hipStreamNonBlocking) != hipSuccess) return (int)hipGetLastError();
  for (int i = 0; i < 4; ++i)
    if (hipEventCreateWithFlags(&a->produced[i], hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
  for (int i = 0; i < 8; ++i)
    if (hipEventCreateWithFlags(&a->consumed[i], hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
  if (hipEventCreateWithFlags(&a->done, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
  // second weight-gradient stream: the 36-tile proj weight gradient runs BESIDE the 108-tile qkv one (see dit_backward_impl)
  if (reused) {
  } else if (ablate_mask() & 4) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&a->side2, hipStreamNonBlocking, lo) != hipSuccess) return (int)hipGetLastError();
  } else if (hipStreamCreateWithFlags(&a->side2, hipStreamNonBlocking) != hipSuccess) return (int)hipGetLastError();
  if (hipEventCreateWithFlags(&a->join2, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
  if (hipEventCreateWithFlags(&a->ada_ready, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
  if (hipEventCreateWithFlags(&a->ada_factors, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
  *aux = a;
  return SFRON_OK;
}
int sfron_aux_wait_ada(void* aux, void* stream) {
  SFRON_CHECK_ARG(aux);
  if (hipStreamWaitEvent((hipStream_t)stream, ((Aux*)aux)->ada_ready, 0) != hipSuccess) return (int)hipGetLastError();
  return SFRON_OK;
}
int sfron_aux_streams(void* aux, void** side, void** side2) {
  SFRON_CHECK_ARG(aux && side && side2);
  *side = (void*)((Aux*)aux)->side; *side2 = (void*)((Aux*)aux)->side2;
  return SFRON_OK;
}
int sfron_aux_wait_ada_factors(void* aux, void* stream) {
  SFRON_CHECK_ARG(aux);
  if (hipStreamWaitEvent((hipStream_t)stream, ((Aux*)aux)->ada_factors, 0) != hipSuccess) return (int)hipGetLastError();
  return SFRON_OK;
}
int sfron_aux_destroy(void* aux) {
  SFRON_CHECK_ARG(aux);
  Aux* a = (Aux*)aux;
  (void)sfron_take_stop_event();                 // never leave one of this handle's events armed for a later launch
  for (int i = 0; i < 4; ++i) (void)hipEventDestroy(a->produced[i]);
  for (int i = 0; i < 8; ++i) (void)hipEventDestroy(a->consumed[i]);
  (void)hipEventDestroy(a->done);
  (void)hipEventDestroy(a->join2);
  (void)hipEventDestroy(a->ada_ready);
  (void)hipEventDestroy(a->ada_factors);
  if (ablate_mask() & 4) {
    (void)hipStreamDestroy(a->side);
    (void)hipStreamDestroy(a->side2);
  } else {
    // back to the free list, behind everything the handle's owner queued on them (the next owner starts its passes with an event wait on
    // its own stream, not on whatever the previous owner left here)
    (void)hipStreamSynchronize(a->side);
    (void)hipStreamSynchronize(a->side2);
    std::lock_guard<std::mutex> lk(g_pairs_mu);
    g_free_pairs.push_back(StreamPair{a->dev, a->side, a->side2});
  }
  delete a;
  return SFRON_OK;
}

// ---- probe: HIP event pairs around chosen launches, recorded on the stream the launch goes to, for bench.py's live roofline
// measurement: the forward pass brackets the fc1 GEMM of block 0; a probe attached to the aux handle (sfron_aux_set_probe)
// brackets, on the weight-gradient stream, the four weight-gradient GEMMs of every 9th block
struct Probe { hipEvent_t* ev; int cap, used; };

int sfron_aux_set_probe(void* aux, void* probe) {
  SFRON_CHECK_ARG(aux);
  ((Aux*)aux)->probe = (Probe*)probe;
  return SFRON_OK;
}

// per block: number of sum-of-squares partials of its four weight gradients (qkv, proj, fc1, fc2), 0 if one of the shapes is unsupported
static int sumsq_counts(const Dims& d, int cnt[4]) {
  const int N[4] = {3 * d.D, d.D, d.F, d.D}, K[4] = {d.D, d.D, d.D, d.F};
  int tot = 0;
  for (int j = 0; j < 4; ++j) {
    cnt[j] = sfron_gemm_sumsq_partials(N[j], K[j], d.M);
    if (cnt[j] <= 0) return 0;
    tot += cnt[j];
  }
  return tot;
}
int sfron_dit_sumsq_partials_len(const sfron_dit_cfg* cfg) {
  Dims d;
  if (make_dims(cfg, d) != SFRON_OK) return 0;
  int cnt[4];
  return sumsq_counts(d, cnt) * d.L;
}
int sfron_aux_arm_sumsq(void* aux, const uint8_t* mask_arena, double* partials) {
  SFRON_CHECK_ARG(aux);
  /* partials == NULL disarms (a pass that raised between the arm and its backward must not leave a raw pointer behind for the next one) */
  ((Aux*)aux)->sq_mask = partials ? mask_arena : nullptr; ((Aux*)aux)->sq_partials = partials;
  return SFRON_OK;
}

int sfron_probe_create(int max_samples, void** probe) {
  SFRON_CHECK_ARG(probe && max_samples > 0);
  Probe* p = new Probe{new hipEvent_t[2 * max_samples], max_samples, 0};
  for (int i = 0; i < 2 * max_samples; ++i)
    if (hipEventCreate(&p->ev[i]) != hipSuccess) return (int)hipGetLastError();
  *probe = p;
  return SFRON_OK;
}
int sfron_probe_reset(void* probe) { SFRON_CHECK_ARG(probe); ((Probe*)probe)->used = 0; return SFRON_OK; }
int sfron_probe_read(void* probe, int* n_samples, double* total_ms) {
  SFRON_CHECK_ARG(probe && n_samples && total_ms);
  Probe* p = (Probe*)probe;
  double tot = 0;
  for (int i = 0; i < p->used; ++i) {
    float ms = 0;
    if (hipEventSynchronize(p->ev[2 * i + 1]) != hipSuccess) return (int)hipGetLastError();
    if (hipEventElapsedTime(&ms, p->ev[2 * i], p->ev[2 * i + 1]) != hipSuccess) return (int)hipGetLastError();
    tot += ms;
  }
  *n_samples = p->used; *total_ms = tot;
  return SFRON_OK;
}
int sfron_probe_destroy(void* probe) {
  SFRON_CHECK_ARG(probe);
  Probe* p = (Probe*)probe;
  for (int i = 0; i < 2 * p->cap; ++i) (void)hipEventDestroy(p->ev[i]);
  delete[] p->ev; delete p;
  return SFRON_OK;
}

// Round 6: where the fc1 / fc2 shapes run on the 256 x 192 pipelined tile (every registry model at 256 px and batch >= 8), the second output of
// fc1 + GELU -- which only the fc2 dgrad's GELU' reads -- is GELU'(pre-activation) as ONE byte per element (SFRON_EPI_GELU_Q / _DGELU_Q) instead
// of the bf16 pre-activation; the block's `hpre` slot then holds M * F bytes of codes.  The SAME rule decides in the forward passes (bf16 and
// fp8) and in the backward pass.  -DSFRON_TUNE_NO_GELU_Q (A-B builds, tools/build_variant.sh) keeps the bf16 pre-activation everywhere.
static bool use_gelu_q(const Dims& d) {
#ifdef SFRON_TUNE_NO_GELU_Q
  return false;
#else
  return sfron_gemm_gelu_q_supported(d.M, d.F, d.D) != 0;
#endif
}

// config 5: e4m3 weight shadow + scales + the e4m3 activation scratch of ONE block (only the forward GEMMs read them)
struct Fp8Ctx {
  const uint8_t* w8; const float* w_scales; float s_x, s_o, s_h;
  uint8_t *xmod8, *o8, *h8;
  uint32_t* act_amax;            // the caller's activation-range words (sfron_fp8_activation_amax), or null
};
static size_t fp8_ws(const Dims& d, char* base, Fp8Ctx* f) {
  size_t o = 0;
  auto take = [&](size_t bytes) { char* r = base ? base + o : nullptr; o += (bytes + 255) / 256 * 256; return r; };
  uint8_t* a = (uint8_t*)take((size_t)d.M * d.D); uint8_t* b = (uint8_t*)take((size_t)d.M * d.D); uint8_t* c = (uint8_t*)take((size_t)d.M * d.F);
  if (f) { f->xmod8 = a; f->o8 = b; f->h8 = c; }
  return o;
}

static int dit_forward_impl(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const float* x_t,
                            const int64_t* t, const int64_t* y, const uint8_t* drop, void* workspace, float* out,
                            void* probe, const Fp8Ctx* f8, void* stream, void* const* block_wait = nullptr, int phase = 0);

int sfron_dit_forward(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const float* x_t,
                      const int64_t* t, const int64_t* y, const uint8_t* drop, void* workspace, float* out, void* stream) {
  return dit_forward_impl(cfg, params, params_bf16, x_t, t, y, drop, workspace, out, nullptr, nullptr, stream);
}

int sfron_dit_forward_probed(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const float* x_t,
                             const int64_t* t, const int64_t* y, const uint8_t* drop, void* workspace, float* out,
                             void* probe, void* stream) {
  return dit_forward_impl(cfg, params, params_bf16, x_t, t, y, drop, workspace, out, probe, nullptr, stream);
}

int sfron_dit_forward_after(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const float* x_t,
                            const int64_t* t, const int64_t* y, const uint8_t* drop, void* workspace, float* out,
                            void* const* block_ready, void* probe, void* stream) {
  SFRON_CHECK_ARG(block_ready);
  return dit_forward_impl(cfg, params, params_bf16, x_t, t, y, drop, workspace, out, probe, nullptr, stream, block_ready);
}

int sfron_dit_forward_phase(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const float* x_t,
                            const int64_t* t, const int64_t* y, const uint8_t* drop, void* workspace, float* out,
                            void* const* block_ready, void* probe, int phase, void* stream) {
  SFRON_CHECK_ARG(phase >= 1 && phase <= 4);
  return dit_forward_impl(cfg, params, params_bf16, x_t, t, y, drop, workspace, out, probe, nullptr, stream, block_ready, phase);
}

int64_t sfron_dit_fp8_workspace_bytes(const sfron_dit_cfg* cfg) {
  Dims d;
  if (make_dims(cfg, d) != SFRON_OK) return -1;
  return (int64_t)fp8_ws(d, nullptr, nullptr);
}

static int dit_forward_fp8_impl(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const uint8_t* params_e4m3,
                                const float* w_scales, const float* act_scales, uint32_t* act_amax, const float* x_t, const int64_t* t, const int64_t* y,
                                const uint8_t* drop, void* workspace, void* workspace_e4m3, float* out, void* const* block_ready, int phase,
                                void* stream);
int sfron_dit_forward_fp8(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const uint8_t* params_e4m3,
                          const float* w_scales, const float* act_scales, uint32_t* act_amax, const float* x_t, const int64_t* t, const int64_t* y,
                          const uint8_t* drop, void* workspace, void* workspace_e4m3, float* out, void* const* block_ready, void* stream) {
  return dit_forward_fp8_impl(cfg, params, params_bf16, params_e4m3, w_scales, act_scales, act_amax, x_t, t, y, drop, workspace, workspace_e4m3, out,
                              block_ready, 0, stream);
}
int sfron_dit_forward_fp8_phase(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const uint8_t* params_e4m3,
                                const float* w_scales, const float* act_scales, uint32_t* act_amax, const float* x_t, const int64_t* t, const int64_t* y,
                                const uint8_t* drop, void* workspace, void* workspace_e4m3, float* out, void* const* block_ready, int phase,
                                void* stream) {
  SFRON_CHECK_ARG(phase >= 1 && phase <= 4);
  return dit_forward_fp8_impl(cfg, params, params_bf16, params_e4m3, w_scales, act_scales, act_amax, x_t, t, y, drop, workspace, workspace_e4m3, out,
                              block_ready, phase, stream);
}
static int dit_forward_fp8_impl(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const uint8_t* params_e4m3,
                                const float* w_scales, const float* act_scales, uint32_t* act_amax, const float* x_t, const int64_t* t, const int64_t* y,
                                const uint8_t* drop, void* workspace, void* workspace_e4m3, float* out, void* const* block_ready, int phase,
                                void* stream) {
  Dims d;
  RUN(make_dims(cfg, d));
  SFRON_CHECK_ARG(params_e4m3 && w_scales && act_scales && workspace_e4m3);
  SFRON_CHECK_ARG(act_scales[0] > 0.f && act_scales[1] > 0.f && act_scales[2] > 0.f);
  if (!sfron_fp8_gemm_supported(d.M, 3 * d.D, d.D) || !sfron_fp8_gemm_supported(d.M, d.D, d.D) || !sfron_fp8_gemm_supported(d.M, d.F, d.D) ||
      !sfron_fp8_gemm_supported(d.M, d.D, d.F))
    return SFRON_ERR_UNSUPPORTED;
  Fp8Ctx f{params_e4m3, w_scales, act_scales[0], act_scales[1], act_scales[2], nullptr, nullptr, nullptr, act_amax};
  (void)fp8_ws(d, (char*)workspace_e4m3, &f);
  return dit_forward_impl(cfg, params, params_bf16, x_t, t, y, drop, workspace, out, nullptr, &f, stream, block_ready, phase);
}

static int dit_forward_impl(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const float* x_t,
                            const int64_t* t, const int64_t* y, const uint8_t* drop, void* workspace, float* out,
                            void* probe, const Fp8Ctx* f8, void* stream, void* const* block_wait, int phase) {
  Dims d;
  RUN(make_dims(cfg, d));
  SFRON_CHECK_ARG(params && params_bf16 && x_t && t && y && workspace && out);
  const ParamLayout P = make_layout(d);
  Workspace w = make_ws(d, (char*)workspace);
  const uint16_t* wb = params_bf16;
  const int M = d.M, D = d.D, T = d.T, NM = d.NM;
  sfron_gemm_desc g;

  // phase 1 = only what stands in front of block 0 (the conditioning prologue: a chain of ten small dependent launches + the adaLN product),
  // phase 2 = only the blocks and the final layer on the workspace a phase-1 call filled, 0 = both.  A caller whose optimizer sweep runs
  // beside this pass on another stream starts that sweep BETWEEN the two calls: beside a bandwidth-heavy sweep every boundary between two of
  // the prologue's small launches costs 60-100 us instead of ~5 (profiles/r06_stage_boundary.txt), ~0.5 ms per pass.
  // phase 3 = phase 1 without its last launch, the adaLN product -- the only launch of the prologue that reads the adaLN matrix; phase 4 = that
  // launch alone.  A caller whose optimizer sweep of that matrix (a third of the parameters, ~1.1 ms) runs on another stream issues phase 3
  // beside it, then orders this stream behind the sweep, then phase 4 and phase 2 (engine.forward(ada_ready=...)).
  if (phase != 2 && phase != 4) {
  // x = x_embedder(x) + pos_embed                                        (models.py:240)
  RUN(sfron_patchify(x_t, d.B, d.C, d.S, d.S, d.p, 0, (uint16_t*)w.patches, d.Kp, stream));
  g = fwd_desc(w.patches, wb + P.pe_w, M, D, d.Kp);
  g.epilogue = SFRON_EPI_POS; g.bias = params + P.pe_b; g.c_f32 = w.xs; g.ldc_f32 = D; g.pos = params + P.pos; g.tokens = T;
  RUN(sfron_gemm_bf16(&g, stream));
  // t = t_embedder(t)                                                    (models.py:61-64,241)
  RUN(sfron_timestep_embed(t, d.B, d.fdim, (uint16_t*)w.tfreq, d.fdim, stream));
  g = fwd_desc(w.tfreq, wb + P.t0_w, d.B, D, d.fdim);
  g.epilogue = SFRON_EPI_F32; g.bias = params + P.t0_b; g.c_f32 = w.h1; g.ldc_f32 = D;
  RUN(sfron_gemm_bf16(&g, stream));
  RUN(sfron_silu_fwd(w.h1, (int64_t)d.B * D, (uint16_t*)w.h1s, stream));
  g = fwd_desc(w.h1s, wb + P.t2_w, d.B, D, D);
  g.epilogue = SFRON_EPI_F32; g.bias = params + P.t2_b; g.c_f32 = w.temb; g.ldc_f32 = D;
  RUN(sfron_gemm_bf16(&g, stream));
  // c = t + y_embedder(y); SiLU(c) feeds every adaLN_modulation           (models.py:242-243,114,132)
  RUN(sfron_cond_fwd(w.temb, params + P.table, y, drop, d.ncls, d.B, D, w.c, (uint16_t*)w.sc, stream));
  }
  if (phase != 2 && phase != 3) {
  g = fwd_desc(w.sc, wb + P.ada_w, d.B, NM, D);
  g.epilogue = SFRON_EPI_F32; g.bias = params + P.ada_b; g.c_f32 = w.mod; g.ldc_f32 = NM;
  RUN(sfron_gemm_bf16(&g, stream));
  }
  if (phase == 1 || phase == 3 || phase == 4) return SFRON_OK;

  // x_next = x + gate * (X W^T + b), branch output saved for the backward pass (models.py:120-121): one product with the gated-residual
  // epilogue, or -- few-tile shapes -- a split-K product + its finish kernel
  auto gate_res = [&](const void* X, const void* W, int K, const float* bias, float* x_next, const float* x_in, __bf16* branch,
                      const float* gate) -> int {
    sfron_gemm_desc q = fwd_desc(X, W, M, D, K);
    const int sp = w.kslab ? small_m_splits(M, D, K) : 1;
    if (sp > 1) {
      q.epilogue = SFRON_EPI_F32; q.c_f32 = w.kslab; q.ldc_f32 = D; q.split_k = sp; q.split_stride = (long)M * D;
      RUN(sfron_gemm_bf16(&q, stream));
      return sfron_split_gate_res(w.kslab, sp, (int64_t)M * D, bias, gate, NM, T, x_in, x_next, (uint16_t*)branch, M, D, stream);
    }
    q.epilogue = SFRON_EPI_GATE_RES; q.bias = bias; q.c_f32 = x_next; q.ldc_f32 = D; q.resid = x_in;
    q.aux = (uint16_t*)branch; q.ldaux = D; q.gate = gate; q.ldgate = NM; q.tokens = T;
    return sfron_gemm_bf16(&q, stream);
  };
  for (int l = 0; l < d.L; ++l) {
    const int64_t pb = P.blocks + (int64_t)l * P.blk_stride;
    // block l's weights may still be under an optimizer sweep that runs on another stream (sfron_dit_forward_after): wait for ITS event
    if (block_wait && block_wait[l] && hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)block_wait[l], 0) != hipSuccess)
      return (int)hipGetLastError();
    const float* mod = w.mod + (size_t)l * 6 * D;          // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
    float* x0 = w.xs + (size_t)(2 * l) * M * D;
    float* x1 = x0 + (size_t)M * D;
    float* x2 = x1 + (size_t)M * D;
    __bf16* xmod1 = w.xmod1 + (size_t)l * M * D; __bf16* qkv = w.qkv + (size_t)l * M * 3 * D;
    __bf16* o = w.o + (size_t)l * M * D; __bf16* a1 = w.a1 + (size_t)l * M * D;
    __bf16* xmod2 = w.xmod2 + (size_t)l * M * D; __bf16* hpre = w.hpre + (size_t)l * M * d.F;
    __bf16* h = w.h + (size_t)l * M * d.F; __bf16* a2 = w.a2 + (size_t)l * M * D;
    if (f8) {
      // config 5: the same block with its four products on the fp8 matrix core.  Every bf16 tensor the backward pass reads is
      // written exactly as in the bf16 path; the e4m3 copies (xmod8, o8, h8) live only until the next block overwrites them.
      const float* ws = f8->w_scales + (size_t)l * 4;
      auto g8 = [&](const uint8_t* A, int64_t w_off, int N, int K, const float* wsc, float asc, int epi) {
        sfron_fp8_gemm_desc q{};
        q.A = A; q.B = f8->w8 + w_off; q.M = M; q.N = N; q.K = K; q.w_scale = wsc; q.a_scale = asc; q.epilogue = epi; q.tokens = T;
        q.act_amax = f8->act_amax;
        return q;
      };
      RUN(sfron_ln_modulate_fwd_q(x0, mod, mod + D, NM, T, M, D, (uint16_t*)xmod1, f8->xmod8, f8->s_x, w.mean + (size_t)(2 * l) * M,
                                  w.rstd + (size_t)(2 * l) * M, f8->act_amax, stream));
      sfron_fp8_gemm_desc q = g8(f8->xmod8, pb + P.o_qkv_w, 3 * D, D, ws + 0, f8->s_x, SFRON_EPI_BF16);
      q.bias = params + pb + P.o_qkv_b; q.c_bf16 = (uint16_t*)qkv; q.ldc_bf16 = 3 * D;
      RUN(sfron_fp8_gemm(&q, stream));
      RUN(sfron_attn_fwd((const uint16_t*)qkv, (uint16_t*)o, w.lse + (size_t)l * d.B * d.H * T, d.B, T, d.H, d.hd, stream));
      RUN(sfron_cast_e4m3(o, 1, (int64_t)M * D, f8->s_o, f8->o8, f8->act_amax, stream));
      q = g8(f8->o8, pb + P.o_proj_w, D, D, ws + 1, f8->s_o, SFRON_EPI_GATE_RES);
      q.bias = params + pb + P.o_proj_b; q.c_f32 = x1; q.ldc_f32 = D; q.resid = x0; q.aux = (uint16_t*)a1; q.ldaux = D;
      q.gate = mod + 2 * D; q.ldgate = NM;
      RUN(sfron_fp8_gemm(&q, stream));
      RUN(sfron_ln_modulate_fwd_q(x1, mod + 3 * D, mod + 4 * D, NM, T, M, D, (uint16_t*)xmod2, f8->xmod8, f8->s_x,
                                  w.mean + (size_t)(2 * l + 1) * M, w.rstd + (size_t)(2 * l + 1) * M, f8->act_amax, stream));
      q = g8(f8->xmod8, pb + P.o_fc1_w, d.F, D, ws + 2, f8->s_x, SFRON_EPI_GELU);
      q.bias = params + pb + P.o_fc1_b; q.c_bf16 = (uint16_t*)h; q.ldc_bf16 = d.F; q.aux = (uint16_t*)hpre; q.ldaux = d.F;
      q.aux_q = use_gelu_q(d) ? 1 : 0;
      q.c_e4m3 = f8->h8; q.c_e4m3_scale = f8->s_h;
      RUN(sfron_fp8_gemm(&q, stream));
      q = g8(f8->h8, pb + P.o_fc2_w, D, d.F, ws + 3, f8->s_h, SFRON_EPI_GATE_RES);
      q.bias = params + pb + P.o_fc2_b; q.c_f32 = x2; q.ldc_f32 = D; q.resid = x1; q.aux = (uint16_t*)a2; q.ldaux = D;
      q.gate = mod + 5 * D; q.ldgate = NM;
      RUN(sfron_fp8_gemm(&q, stream));
      continue;
    }
    // x = x + gate_msa * attn(modulate(norm1(x), shift_msa, scale_msa))   (models.py:120)
    RUN(sfron_ln_modulate_fwd(x0, mod, mod + D, NM, T, M, D, (uint16_t*)xmod1, w.mean + (size_t)(2 * l) * M,
                              w.rstd + (size_t)(2 * l) * M, stream));
    g = fwd_desc(xmod1, wb + pb + P.o_qkv_w, M, 3 * D, D);
    g.bias = params + pb + P.o_qkv_b; g.c_bf16 = (uint16_t*)qkv; g.ldc_bf16 = 3 * D;
    RUN(sfron_gemm_bf16(&g, stream));
    RUN(sfron_attn_fwd((const uint16_t*)qkv, (uint16_t*)o, w.lse + (size_t)l * d.B * d.H * T, d.B, T, d.H, d.hd, stream));
    RUN(gate_res(o, wb + pb + P.o_proj_w, D, params + pb + P.o_proj_b, x1, x0, a1, mod + 2 * D));
    // x = x + gate_mlp * mlp(modulate(norm2(x), shift_mlp, scale_mlp))    (models.py:121)
    RUN(sfron_ln_modulate_fwd(x1, mod + 3 * D, mod + 4 * D, NM, T, M, D, (uint16_t*)xmod2, w.mean + (size_t)(2 * l + 1) * M,
                              w.rstd + (size_t)(2 * l + 1) * M, stream));
    g = fwd_desc(xmod2, wb + pb + P.o_fc1_w, M, d.F, D);
    g.epilogue = use_gelu_q(d) ? SFRON_EPI_GELU_Q : SFRON_EPI_GELU; g.bias = params + pb + P.o_fc1_b; g.c_bf16 = (uint16_t*)h; g.ldc_bf16 = d.F;
    g.aux = (uint16_t*)hpre; g.ldaux = d.F;
    Probe* pr = (Probe*)probe;
    const bool probing = pr && l == 0 && pr->used < pr->cap;
    if (probing) (void)hipEventRecord(pr->ev[2 * pr->used], (hipStream_t)stream);
    RUN(sfron_gemm_bf16(&g, stream));
    if (probing) { (void)hipEventRecord(pr->ev[2 * pr->used + 1], (hipStream_t)stream); pr->used++; }
    RUN(gate_res(h, wb + pb + P.o_fc2_w, d.F, params + pb + P.o_fc2_b, x2, x1, a2, mod + 5 * D));
  }
  // final layer + unpatchify                                              (models.py:138-142,218-231,247-248)
  const float* modf = w.mod + (size_t)6 * d.L * D;
  float* xL = w.xs + (size_t)(2 * d.L) * M * D;
  RUN(sfron_ln_modulate_fwd(xL, modf, modf + D, NM, T, M, D, (uint16_t*)w.xmodf, w.mean + (size_t)(2 * d.L) * M,
                            w.rstd + (size_t)(2 * d.L) * M, stream));
  g = fwd_desc(w.xmodf, wb + P.fin_w, M, d.Po, D);
  g.epilogue = SFRON_EPI_F32; g.bias = params + P.fin_b; g.c_f32 = w.tok; g.ldc_f32 = d.Po;
  RUN(sfron_gemm_bf16(&g, stream));
  RUN(sfron_unpatchify(w.tok, d.Po, d.B, d.Co, d.S, d.S, d.p, out, stream));
  return SFRON_OK;
}

// block_events (optional, L entries, may hold NULLs): hipEvent_t recorded on the weight-gradient stream when every gradient of
// block l that lives in its arena range is final EXCEPT proj.bias / fc2.bias; late_bias (optional, [L][2][D] fp32): where those
// two go instead of the arena.  Together they let a data-parallel host all-reduce a block's range while the backward pass is
// still running, and exchange the late biases (plus everything outside the blocks) once at the end.
static int dit_backward_impl(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const float* d_out,
                             const int64_t* y, const uint8_t* drop, void* workspace, float* grads, void* aux, void* const* block_events,
                             float* late_bias, uint16_t* ada_dmod_out, uint16_t* ada_sc_out, void* stream) {
  Dims d;
  RUN(make_dims(cfg, d));
  SFRON_CHECK_ARG(params && params_bf16 && d_out && y && workspace && grads);
  // An event armed for a producing launch (arm() below) must not outlive this call: a RUN(...) that returns early between arm(i) and its
  // launch would leave the thread's slot armed, and the next k_gemm_pipe / k_row_bwd / k_attn_bwd_fused launch of ANY caller on this thread
  // would hand it -- possibly a destroyed event of a closed engine -- to hipExtLaunchKernelGGL.  Cleared on entry and on every exit path.
  (void)sfron_take_stop_event();
  struct DisarmOnExit { ~DisarmOnExit() { (void)sfron_take_stop_event(); } } disarm_on_exit;
  const ParamLayout P = make_layout(d);
  Workspace w = make_ws(d, (char*)workspace);
  const uint16_t* wb = params_bf16;
  const int M = d.M, D = d.D, T = d.T, NM = d.NM, B = d.B;
  const int rpc = sfron_rows_per_chunk(T);
  SFRON_CHECK_ARG(rpc > 0);
  const int per = T / rpc, nch = M / rpc;
  const long slot_stride = (long)nch * D;
  // slot (layer, kind, buf): kind 0 = MLP gate, 1 = MLP LN, 2 = attention gate, 3 = attention LN; the final layer's LN is
  // (layer L, kind 3).  All token reductions are deferred to ONE k_reduce_slots launch after the last block.
  auto slot = [&](int layer, int kind, int buf) { return w.part + ((size_t)layer * 8 + kind * 2 + buf) * slot_stride; };
  hipStream_t hs = (hipStream_t)stream;
  sfron_gemm_desc g;
  const bool fuse = !(ablate_mask() & 1);
  // side stream for weight/bias gradients (falls back to the main stream without an aux handle)
  Aux* ax = (Aux*)aux;
  void* side = ax ? (void*)ax->side : stream;
  // one-shot (sfron_aux_arm_sumsq): this pass's block weight gradients also leave their masked sums of squares
  const uint8_t* const sq_mask = ax ? ax->sq_mask : nullptr;
  double* const sq_partials = ax ? ax->sq_partials : nullptr;
  if (ax) { ax->sq_mask = nullptr; ax->sq_partials = nullptr; }
  int sq_cnt[4] = {0, 0, 0, 0}, sq_per_block = 0;
  if (sq_partials) {
    sq_per_block = sumsq_counts(d, sq_cnt);
    if (sq_per_block == 0) return SFRON_ERR_UNSUPPORTED;
  }
  double* sq_next = nullptr;                        // set by the caller of wgrad_on for the block weight gradients
  // block weight gradient on the side stream, reduction split per wgrad_splits() into fp32 slabs + fixed-order sum
  // db (optional): the bias gradient sum_rows dY of the same Linear.  Where the three-slot weight-gradient kernel takes the
  // shape it comes out of that GEMM (row sums of dY^T against a ones fragment); otherwise a column-sum launch precedes it.
  bool probe_block = false;
  auto wgrad_on = [&](void* side, const void* dY, const void* X, int N, int K, float* dW, float* db) -> int {
    sfron_gemm_desc q = wgrad_desc(dY, X, M, N, K, dW);
    if (db) {
      if ((ablate_mask() & 32) && sfron_gemm_rowsum_supported(N, K, M)) {
        q.a_rowsum = db;
        q.rowsum_ws = ax ? w.csum2 : w.csum;            // [K / 192][N] partial rows (<= CSUM_PARTS rows of the widest output)
      } else RUN(sfron_colsum(dY, 1, M, N, N, ax ? w.csum2 : w.csum, CSUM_PARTS, db, side));
    }
    // roofline probe: an event pair on the weight-gradient stream around the GEMM itself (all four weight gradients of the probed blocks)
    Probe* pr = (ax && ax->probe && probe_block && ax->probe->used < ax->probe->cap) ? ax->probe : nullptr;
    if (pr) (void)hipEventRecord(pr->ev[2 * pr->used], (hipStream_t)side);
    struct Close { Probe* p; void* s; ~Close() { if (p) { (void)hipEventRecord(p->ev[2 * p->used + 1], (hipStream_t)s); p->used++; } } } close{pr, side};
    // measured: the splits shorten the side stream (proj 175 -> 60 us) but the backward pass is bound by total CU time, and
    // the slab traffic + reductions make the step 0.5-2 ms SLOWER -> off unless SFRON_ABLATE bit 1 asks for the A-B run
    const int sp = (ablate_mask() & 2) ? wgrad_splits(N, K, M) : 1;
    if (sp > 1) { q.c_f32 = w.wslab; q.split_k = sp; q.split_stride = (long)N * K; }
    if ((ablate_mask() & 512) && N % 256 == 0 && K % 192 == 0) q.tile_hint = 42;      // A-B knob: 256 x 192 tiles (fc1 / fc2: 108 workgroups instead of 144)
    if (sq_next && sp == 1 && q.tile_hint == 0 && !q.a_rowsum) {
      q.sumsq_partials = sq_next;
      q.sumsq_mask = sq_mask ? sq_mask + (dW - grads) : nullptr;
    } else if (sq_next) return SFRON_ERR_UNSUPPORTED;       // (debug-knob builds only: an ablation that changes the weight-gradient form)
    sq_next = nullptr;
    RUN(sfron_gemm_bf16(&q, side));
    if (sp > 1) RUN(sfron_reduce_chunks(w.wslab, 1, sp, N * K, dW, N * K, 0, side));
    return SFRON_OK;
  };
  auto wgrad_side = [&](const void* dY, const void* X, int N, int K, float* dW, float* db = nullptr) -> int {
    return wgrad_on(side, dY, X, N, K, dW, db);
  };
  // partials of matrix j (0 qkv, 1 proj, 2 fc1, 3 fc2) of block l, or null when the pass is not armed
  auto sq_slot = [&](int l, int j) -> double* {
    if (!sq_partials) return nullptr;
    int o = 0;
    for (int i = 0; i < j; ++i) o += sq_cnt[i];
    return sq_partials + (size_t)l * sq_per_block + o;
  };
  // dX [M x D] (bf16) = dY [M x N] W [N x D]: one product, or -- few-tile shapes -- a split-K product + its bf16-writing finish
  auto dgrad_bf16 = [&](const void* dY, const void* W, int N, __bf16* dX) -> int {
    sfron_gemm_desc q = dgrad_desc(dY, W, M, N, D);
    const int sp = w.kslab ? small_m_splits(M, D, N) : 1;
    if (sp > 1) {
      q.epilogue = SFRON_EPI_F32; q.c_f32 = w.kslab; q.ldc_f32 = D; q.split_k = sp; q.split_stride = (long)M * D;
      RUN(sfron_gemm_bf16(&q, stream));
      return sfron_split_sum_bf16(w.kslab, sp, (int64_t)M * D, (int64_t)M * D, (uint16_t*)dX, stream);
    }
    q.c_bf16 = (uint16_t*)dX; q.ldc_bf16 = D;
    return sfron_gemm_bf16(&q, stream);
  };
  // The attention backward needs a whole CU's LDS per workgroup (157 KB), and so does a weight-gradient workgroup (144 KB): beside the
  // 36-tile proj weight gradient it gets 220 CUs = 2.3 rounds of its 512 workgroups.  So proj waits for the attention backward and
  // then runs on a SECOND stream beside the 108-tile qkv weight gradient (36 + 108 = 144 tiles, the size of the fc1 / fc2 launches);
  // the attention backward meets an idle weight-gradient stream.  SFRON_ABLATE bit 8 = the old order.
  const bool pair_proj = ax && !(ablate_mask() & 256);
  // LN backward of one branch + gate backward of the branch before it, on the same dx rows
  auto ln_gate = [&](const float* x, const float* mean, const float* rstd, const float* scale, int acc, float* ps, float* pc,
                     const __bf16* branch, const float* gate, __bf16* d_branch, float* pg, float* pd) -> int {
    if (fuse)
      return sfron_ln_gate_bwd((const uint16_t*)w.d_xmod, x, mean, rstd, scale, NM, T, M, D, w.dx, acc, ps, pc,
                               (const uint16_t*)branch, gate, NM, (uint16_t*)d_branch, pg, pd, stream);
    RUN(sfron_ln_modulate_bwd((const uint16_t*)w.d_xmod, x, mean, rstd, scale, NM, T, M, D, w.dx, acc, ps, pc, stream));
    return sfron_gate_bwd(w.dx, (const uint16_t*)branch, gate, NM, T, M, D, (uint16_t*)d_branch, pg, pd, stream);
  };
  // buffer i (0 d_br mlp, 1 d_hpre, 2 d_br attn, 3 dqkv) of block l lives in copy l & 1: main records produced[i] after
  // writing it, the side stream waits for it before reading; the side stream records consumed[i][l & 1], and main waits
  // for that before block l - 2 overwrites the copy -- so main stalls only when the side stream is a whole block behind.
  // produced(i): buffer i is written -> the weight-gradient stream may read it.  arm(i) in front of the entry point whose ONE kernel writes it
  // lets that kernel's own dispatch carry the event (common.h SFRON_LAUNCH_EV: no marker packet in this stream, ~5 us each, four per
  // block); if that launch did not take it (another kernel path) the event is recorded the old way.
  bool armed = false;
  auto arm = [&](int i) { if (ax) { sfron_arm_stop_event(ax->produced[i]); armed = true; } };
  auto produced = [&](int i) {
    if (!ax) return;
    const bool left = sfron_take_stop_event() != nullptr;                     // armed, but the launch did not take it
    if (!armed || left) (void)hipEventRecord(ax->produced[i], hs);            // ... or never armed: record in the stream
    armed = false;
    (void)hipStreamWaitEvent(ax->side, ax->produced[i], 0);
  };
  auto consumed = [&](int i, int l) { if (ax) (void)hipEventRecord(ax->consumed[2 * i + (l & 1)], ax->side); };
  // The weight-gradient stream is in order, so its LAST event of block l + 2 (buffer 3) covers buffers 0..2 of that block too:
  // one wait per block on the main stream (buffer 0 of block l is written at the end of block l + 1 and waits there) instead
  // of four -- each wait is a barrier packet in the main queue (a few microseconds of bubble whether or not it blocks).
  auto before_overwrite = [&](int i, int l) {
    if (!ax || l + 2 > d.L - 1) return;
    if (i == 0) (void)hipStreamWaitEvent(hs, ax->consumed[2 * 3 + (l & 1)], 0);   // buffers 1..3 of block l are written later on this stream
  };
  if (ax) { (void)hipEventRecord(ax->done, hs); (void)hipStreamWaitEvent(ax->side, ax->done, 0); }   // side starts after everything before us
  const int fc1_rows = (ablate_mask() & 64) ? 0 : sfron_gemm_dgelu_colpart_rows(M, d.F, D);    // 0: shape not on a 256-row pipelined tile -> column-sum launch
  const bool qkv_fused = !(ablate_mask() & 128) && sfron_attn_bwd_bias_supported(T);           // one-kernel attention backward (T = 128 / 256)

  // ---- final layer
  RUN(sfron_patchify(d_out, B, d.Co, d.S, d.S, d.p, 1, (uint16_t*)w.d_tok, d.Po, stream));
  RUN(sfron_colsum(w.d_tok, 1, M, d.Po, d.Po, w.csum, CSUM_PARTS, grads + P.fin_b, stream));
  // small outputs with an M-long reduction (final layer [Po x D], patch embed [D x Kp]): one or nine output tiles would walk
  // all 8192 rows serially (~100 us each), so the reduction is split into slabs and summed in a fixed order
  auto small_wgrad = [&](const void* dY, const void* X, int N, int K, float* dW) -> int {
    sfron_gemm_desc q = wgrad_desc(dY, X, M, N, K, dW);
    int sp = M / 256;
    const long cap = (long)SPLIT_K_ADA * B * D / ((long)N * K);          // what fits in w.slabs
    if (sp > cap) sp = (int)cap;
    if (sp > 64) sp = 64;
    if (sp < 2 || (ablate_mask() & 16)) return sfron_gemm_bf16(&q, stream);
    q.c_f32 = w.slabs; q.split_k = sp; q.split_stride = (long)N * K;
    RUN(sfron_gemm_bf16(&q, stream));
    const int kchunk = cdiv(cdiv(M, sp), 64) * 64;
    return sfron_reduce_chunks(w.slabs, 1, cdiv(M, kchunk), N * K, dW, N * K, 0, stream);
  };
  RUN(small_wgrad(w.d_tok, w.xmodf, d.Po, D, grads + P.fin_w));
  g = dgrad_desc(w.d_tok, wb + P.fin_w, M, d.Po, D);
  g.c_bf16 = (uint16_t*)w.d_xmod; g.ldc_bf16 = D;
  RUN(sfron_gemm_bf16(&g, stream));
  const float* modf = w.mod + (size_t)6 * d.L * D;
  float* dmodf = w.dmod + (size_t)6 * d.L * D;
  // every LN backward is fused with the gate backward of the branch that precedes it in the forward order: the final
  // layer's LN with the last block's MLP gate, a block's MLP LN with its attention gate, its attention LN with the
  // previous block's MLP gate (only block 0's attention LN stands alone).
  {
    const int l = d.L - 1;
    if (fuse) arm(0);
    RUN(ln_gate(w.xs + (size_t)(2 * d.L) * M * D, w.mean + (size_t)(2 * d.L) * M, w.rstd + (size_t)(2 * d.L) * M, modf + D, 0,
                slot(d.L, 3, 0), slot(d.L, 3, 1), w.a2 + (size_t)l * M * D, w.mod + (size_t)l * 6 * D + 5 * D, w.d_br[l & 1],
                slot(l, 0, 0), slot(l, 0, 1)));
  }

  for (int l = d.L - 1; l >= 0; --l) {
    const int64_t pb = P.blocks + (int64_t)l * P.blk_stride;
    const float* mod = w.mod + (size_t)l * 6 * D;
    float* dmod = w.dmod + (size_t)l * 6 * D;
    const float* x0 = w.xs + (size_t)(2 * l) * M * D;
    const float* x1 = x0 + (size_t)M * D;
    const __bf16* xmod1 = w.xmod1 + (size_t)l * M * D; const __bf16* qkv = w.qkv + (size_t)l * M * 3 * D;
    const __bf16* o = w.o + (size_t)l * M * D; const __bf16* a1 = w.a1 + (size_t)l * M * D;
    const __bf16* xmod2 = w.xmod2 + (size_t)l * M * D; const __bf16* hpre = w.hpre + (size_t)l * M * d.F;
    const __bf16* h = w.h + (size_t)l * M * d.F; const __bf16* a2 = w.a2 + (size_t)l * M * D;
    const bool first = (l == d.L - 1);
    probe_block = (l % 9 == 0);
    // ---- MLP branch: x2 = x1 + gate_mlp * (fc2(gelu(fc1(xmod2))))
    const int pl = l & 1;
    const bool delay_fc2 = ablate_mask() & 8;      // A-B knob: start the fc2 weight gradient only after the fc2 dgrad (768 tiles)
    if (!delay_fc2) {
      produced(0);                                // d_br was written by the fused LN+gate kernel just before this block
      sq_next = sq_slot(l, 3);
      RUN(wgrad_side(w.d_br[pl], h, D, d.F, grads + pb + P.o_fc2_w));
      consumed(0, l);
    }
    before_overwrite(1, l);
    g = dgrad_desc(w.d_br[pl], wb + pb + P.o_fc2_w, M, D, d.F);
    g.epilogue = use_gelu_q(d) ? SFRON_EPI_DGELU_Q : SFRON_EPI_DGELU; g.c_bf16 = (uint16_t*)w.d_hpre[pl]; g.ldc_bf16 = d.F; g.aux = (uint16_t*)hpre; g.ldaux = d.F;
    // fc1.bias gradient = column sums of d_hpre: the epilogue that produces d_hpre leaves per-tile-row partials (fp32, before the
    // bf16 rounding); the side stream only adds the M / 256 partial rows (was: a second 75 MB pass over d_hpre per block)
    float* const bp_fc1 = w.bpart + (size_t)pl * fc1_rows * d.F;
    if (fc1_rows) g.col_partials = bp_fc1;
    arm(1);
    RUN(sfron_gemm_bf16(&g, stream));
    produced(1);
    if (delay_fc2) {
      sq_next = sq_slot(l, 3);
      RUN(wgrad_side(w.d_br[pl], h, D, d.F, grads + pb + P.o_fc2_w));
      consumed(0, l);
    }
    sq_next = sq_slot(l, 2);
    if (fc1_rows) {
      RUN(sfron_reduce_chunks(bp_fc1, 1, fc1_rows, d.F, grads + pb + P.o_fc1_b, d.F, 0, side));
      RUN(wgrad_side(w.d_hpre[pl], xmod2, d.F, D, grads + pb + P.o_fc1_w));
    } else RUN(wgrad_side(w.d_hpre[pl], xmod2, d.F, D, grads + pb + P.o_fc1_w, grads + pb + P.o_fc1_b));
    consumed(1, l);
    RUN(dgrad_bf16(w.d_hpre[pl], wb + pb + P.o_fc1_w, d.F, w.d_xmod));
    // ---- attention branch: x1 = x0 + gate_msa * proj(attn(qkv(xmod1)))
    before_overwrite(2, l);
    if (fuse) arm(2);
    RUN(ln_gate(x1, w.mean + (size_t)(2 * l + 1) * M, w.rstd + (size_t)(2 * l + 1) * M, mod + 4 * D, 1, slot(l, 1, 0),
                slot(l, 1, 1), a1, mod + 2 * D, w.d_br2[pl], slot(l, 2, 0), slot(l, 2, 1)));
    produced(2);
    if (!pair_proj) {
      sq_next = sq_slot(l, 1);
      RUN(wgrad_side(w.d_br2[pl], o, D, D, grads + pb + P.o_proj_w));
      consumed(2, l);
    }
    RUN(dgrad_bf16(w.d_br2[pl], wb + pb + P.o_proj_w, D, w.d_o));
    before_overwrite(3, l);
    int proj_rc = SFRON_OK;
    auto proj_beside = [&]() {
      if (!pair_proj) return;
      (void)hipStreamWaitEvent(ax->side2, ax->produced[3], 0);       // recorded after produced[2] on the same stream: covers d_br2
      sq_next = sq_slot(l, 1);
      proj_rc = wgrad_on((void*)ax->side2, w.d_br2[pl], o, D, D, grads + pb + P.o_proj_w, nullptr);
      (void)hipEventRecord(ax->join2, ax->side2);
    };
    if (qkv_fused) {
      // qkv.bias gradient = token sums of dqkv: the attention backward kernel leaves one partial row per sample (it holds dQ / dK / dV
      // of a whole (sample, head) in registers); the side stream adds the B rows (was: a second 57 MB pass over dqkv per block)
      float* const bp = w.bpart_qkv + (size_t)pl * B * 3 * D;
      arm(3);
      RUN(sfron_attn_bwd_bias((const uint16_t*)qkv, (const uint16_t*)o, (const uint16_t*)w.d_o, w.lse + (size_t)l * B * d.H * T,
                              (uint16_t*)w.dqkv[pl], bp, B, T, d.H, d.hd, stream));
      produced(3);
      proj_beside();
      RUN(sfron_reduce_chunks(bp, 1, B, 3 * D, grads + pb + P.o_qkv_b, 3 * D, 0, side));
      sq_next = sq_slot(l, 0);
      RUN(wgrad_side(w.dqkv[pl], xmod1, 3 * D, D, grads + pb + P.o_qkv_w));
    } else {
      RUN(sfron_attn_bwd((const uint16_t*)qkv, (const uint16_t*)o, (const uint16_t*)w.d_o, w.lse + (size_t)l * B * d.H * T,
                         w.delta, (uint16_t*)w.dqkv[pl], B, T, d.H, d.hd, stream));
      produced(3);
      proj_beside();
      sq_next = sq_slot(l, 0);
      RUN(wgrad_side(w.dqkv[pl], xmod1, 3 * D, D, grads + pb + P.o_qkv_w, grads + pb + P.o_qkv_b));
    }
    if (pair_proj) { (void)hipStreamWaitEvent(ax->side, ax->join2, 0); consumed(2, l); }   // the in-order stream's later events cover proj too
    consumed(3, l);
    if (proj_rc != SFRON_OK) return proj_rc;
    if (block_events && block_events[l]) {                 // block l: the four weight gradients, qkv.bias and fc1.bias are final
      if (hipEventRecord((hipEvent_t)block_events[l], (hipStream_t)side) != hipSuccess) return (int)hipGetLastError();
    }
    RUN(dgrad_bf16(w.dqkv[pl], wb + pb + P.o_qkv_w, 3 * D, w.d_xmod));
    if (l > 0) {
      before_overwrite(0, l - 1);
      if (fuse) arm(0);                     // consumed by produced(0) at the top of block l - 1
      RUN(ln_gate(x0, w.mean + (size_t)(2 * l) * M, w.rstd + (size_t)(2 * l) * M, mod + D, 1, slot(l, 3, 0), slot(l, 3, 1),
                  w.a2 + (size_t)(l - 1) * M * D, w.mod + (size_t)(l - 1) * 6 * D + 5 * D, w.d_br[(l - 1) & 1], slot(l - 1, 0, 0),
                  slot(l - 1, 0, 1)));
    } else {
      RUN(sfron_ln_modulate_bwd((const uint16_t*)w.d_xmod, x0, w.mean, w.rstd, mod + D, NM, T, M, D, w.dx, 1, slot(0, 3, 0),
                                slot(0, 3, 1), stream));
    }
  }
  {
    // deferred token reductions of the whole pass: d(shift, scale, gate) of every block + the dy token sums
    float* const dst_base[8] = {w.dmod + 5 * D, w.dysum + (size_t)B * D,          // MLP gate:  d gate_mlp | sum_t dy (fc2.bias)
                                w.dmod + 3 * D, w.dmod + 4 * D,                    // MLP LN:    d shift_mlp | d scale_mlp
                                w.dmod + 2 * D, w.dysum,                           // attn gate: d gate_msa | sum_t dy (proj.bias)
                                w.dmod, w.dmod + D};                               // attn LN:   d shift_msa | d scale_msa (final layer: layer L)
    const long dst_stride[8] = {6L * D, 2L * B * D, 6L * D, 6L * D, 6L * D, 2L * B * D, 6L * D, 6L * D};
    const int dst_ld[8] = {NM, D, NM, NM, NM, D, NM, NM};
    RUN(sfron_reduce_slots(w.part, slot_stride, d.L * 8, B, per, D, dst_base, dst_stride, dst_ld, stream));
    RUN(sfron_reduce2(slot(d.L, 3, 0), slot(d.L, 3, 1), B, per, D, dmodf, NM, dmodf + D, NM, stream));   // final layer's LN
  }
  if (ax) { (void)hipEventRecord(ax->done, ax->side); (void)hipStreamWaitEvent(hs, ax->done, 0); }   // join
  // proj.bias / fc2.bias gradients of every block: sum_b gate[b] * (sum_t dy[b,t])
  if (late_bias) RUN(sfron_gated_bias_grads(w.dysum, w.mod + 2 * D, NM, 6 * D, 3 * D, d.L, B, D, late_bias, 2L * D, 0, D, stream));
  else RUN(sfron_gated_bias_grads(w.dysum, w.mod + 2 * D, NM, 6 * D, 3 * D, d.L, B, D, grads + P.blocks, P.blk_stride, P.o_proj_b,
                                  P.o_fc2_b, stream));
  // ---- patch embed (pos_embed is frozen: no gradient)
  RUN(sfron_cast_bf16(w.dx, (uint16_t*)w.dx_bf, (int64_t)M * D, stream));
  RUN(sfron_colsum(w.dx, 0, M, D, D, w.csum, CSUM_PARTS, grads + P.pe_b, stream));
  RUN(small_wgrad(w.dx_bf, w.patches, D, d.Kp, grads + P.pe_w));
  // ---- adaLN modulation Linear of every block + final layer, as one problem
  RUN(sfron_cast_bf16(w.dmod, (uint16_t*)w.dmod_bf, (int64_t)B * NM, stream));
  RUN(sfron_colsum(w.dmod, 0, B, NM, NM, w.csum, CSUM_PARTS, grads + P.ada_b, stream));
  if (ada_dmod_out && ada_sc_out) {
    // data parallel: dW_ada = sum over the GLOBAL batch of dmod[b]^T (x) silu(c)[b] is a rank-(batch) product.  Handing out the
    // two factors (12.5 MB + 72 KB of bf16 per rank at DiT-XL/2) lets the host all-GATHER them and form the product once over
    // the global batch, instead of all-reducing the 892 MB result that only becomes final here, in the tail of the pass
    if (hipMemcpyAsync(ada_dmod_out, w.dmod_bf, (size_t)B * NM * 2, hipMemcpyDeviceToDevice, hs) != hipSuccess) return (int)hipGetLastError();
    if (hipMemcpyAsync(ada_sc_out, w.sc, (size_t)B * D * 2, hipMemcpyDeviceToDevice, hs) != hipSuccess) return (int)hipGetLastError();
    // the two factors are complete: the clip norm's share of the adaLN matrix (sfron_sumsq_lowrank reads nothing else) may start on another
    // stream now, beside the dgrad through the adaLN Linear and the embedders' backward (sfron_aux_wait_ada_factors)
    if (ax) (void)hipEventRecord(ax->ada_factors, hs);
  } else {
    g = wgrad_desc(w.dmod_bf, w.sc, B, NM, D, grads + P.ada_w);
    RUN(sfron_gemm_bf16(&g, stream));
  }
  g = dgrad_desc(w.dmod_bf, wb + P.ada_w, B, NM, D);
  g.epilogue = SFRON_EPI_F32; g.c_f32 = w.slabs; g.ldc_f32 = D; g.split_k = SPLIT_K_ADA; g.split_stride = (long)B * D;
  RUN(sfron_gemm_bf16(&g, stream));
  // From here on nothing in this pass reads the adaLN matrix (fp32 or bf16) or writes its gradient factors: a sweep of that matrix -- or the
  // clip norm's share of it -- may start on another stream while the embedders' backward (a chain of ~25 small launches) is still running
  // (sfron_aux_wait_ada)
  if (ax) (void)hipEventRecord(ax->ada_ready, hs);
  {
    const int kchunk = cdiv(cdiv(NM, SPLIT_K_ADA), 64) * 64;
    RUN(sfron_reduce_chunks(w.slabs, 1, cdiv(NM, kchunk), B * D, w.d_sc, B * D, 0, stream));
  }
  // ---- c = t_emb + y_emb: label table (scatter) and t_embedder MLP
  if (hipMemsetAsync(grads + P.table, 0, (size_t)(d.ncls + 1) * D * sizeof(float), hs) != hipSuccess) return (int)hipGetLastError();
  RUN(sfron_cond_bwd(w.d_sc, w.c, y, drop, d.ncls, B, D, w.d_c, grads + P.table, stream));
  RUN(sfron_cast_bf16(w.d_c, (uint16_t*)w.d_c_bf, (int64_t)B * D, stream));
  RUN(sfron_colsum(w.d_c, 0, B, D, D, w.csum, CSUM_PARTS, grads + P.t2_b, stream));
  g = wgrad_desc(w.d_c_bf, w.h1s, B, D, D, grads + P.t2_w);
  RUN(sfron_gemm_bf16(&g, stream));
  g = dgrad_desc(w.d_c_bf, wb + P.t2_w, B, D, D);
  g.epilogue = SFRON_EPI_F32; g.c_f32 = w.d_h1s; g.ldc_f32 = D;
  RUN(sfron_gemm_bf16(&g, stream));
  RUN(sfron_silu_bwd(w.d_h1s, w.h1, (int64_t)B * D, (uint16_t*)w.d_h1_bf, w.d_sc /* fp32 copy, d_sc is free now */, stream));
  RUN(sfron_colsum(w.d_sc, 0, B, D, D, w.csum, CSUM_PARTS, grads + P.t0_b, stream));
  g = wgrad_desc(w.d_h1_bf, w.tfreq, B, D, d.fdim, grads + P.t0_w);
  RUN(sfron_gemm_bf16(&g, stream));
  return SFRON_OK;
}

int sfron_dit_backward(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const float* d_out,
                       const int64_t* y, const uint8_t* drop, void* workspace, float* grads, void* aux, void* stream) {
  return dit_backward_impl(cfg, params, params_bf16, d_out, y, drop, workspace, grads, aux, nullptr, nullptr, nullptr, nullptr, stream);
}

int sfron_dit_backward_dp(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const float* d_out,
                          const int64_t* y, const uint8_t* drop, void* workspace, float* grads, void* aux,
                          void* const* block_events, float* late_bias, uint16_t* ada_dmod_out, uint16_t* ada_sc_out, void* stream) {
  SFRON_CHECK_ARG(!block_events || aux);          // the events are recorded on the aux handle's weight-gradient stream
  SFRON_CHECK_ARG((ada_dmod_out == nullptr) == (ada_sc_out == nullptr));
  return dit_backward_impl(cfg, params, params_bf16, d_out, y, drop, workspace, grads, aux, block_events, late_bias, ada_dmod_out,
                           ada_sc_out, stream);
}

int sfron_dit_scatter_late_bias(const sfron_dit_cfg* cfg, const float* late_bias, float* grads, void* stream) {
  Dims d;
  RUN(make_dims(cfg, d));
  SFRON_CHECK_ARG(late_bias && grads);
  const ParamLayout P = make_layout(d);
  const size_t w = (size_t)d.D * sizeof(float);
  if (hipMemcpy2DAsync(grads + P.blocks + P.o_proj_b, (size_t)P.blk_stride * sizeof(float), late_bias, 2 * w, w, d.L,
                       hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return (int)hipGetLastError();
  if (hipMemcpy2DAsync(grads + P.blocks + P.o_fc2_b, (size_t)P.blk_stride * sizeof(float), late_bias + d.D, 2 * w, w, d.L,
                       hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return (int)hipGetLastError();
  return SFRON_OK;
}

}  // extern "C"
