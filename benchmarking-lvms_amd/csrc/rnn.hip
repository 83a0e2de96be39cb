// rnn.hip — K4 / K2: single-layer LSTM and GRU over a whole sequence (forward + BPTT) as stage-kernel chains.
//
// Replaces the vendor RNN calls of the reference:
//   nn.LSTM on a packed sequence        blvm/models/lstm.py:46-55,96-98   (pack_padded_sequence semantics via `lens`)
//   nn.GRU forward / time-reversed      blvm/models/srnn.py:113-116,196,200-206 (reverse_sequences folded into an
//                                        index map: operations.py:56-87 — per-row reversal, right padding in place)
// One launch per recurrent step in each direction: the hidden projection (M = batch, K = hidden) runs on
// v_mfma_f32_16x16x4_f32 with the K split over the waves of a workgroup, and ALL gate math (forward) / gate
// derivatives (backward, fused into the dgrad that completes dL/dh of the step) lives in the epilogue.  The input
// projection and every weight gradient are hoisted out of the loop into the big MFMA GEMM (gemm.hip).
#include "common.h"
#include "pchain.h"
#include "seqchain.h"

namespace blvm {
namespace {

#define LAUNCH_NW(kernel, nw, grid, stream, args)                                          \
  do {                                                                                     \
    if ((nw) == 16) hipLaunchKernelGGL((kernel<16>), grid, dim3(1024), 0, stream, args);   \
    else if ((nw) == 8) hipLaunchKernelGGL((kernel<8>), grid, dim3(512), 0, stream, args); \
    else hipLaunchKernelGGL((kernel<4>), grid, dim3(256), 0, stream, args);                \
  } while (0)

inline int pick_nw(int K, int groups) {
  const int chunks = (K / 16) * groups;
  if (chunks > 32) return 16;
  if (chunks > 16) return 8;
  return 4;
}

inline int pick_split(int M, int N, int K) { return gemm_pick_split(M, N, K); }

// time index processed by row `b` at recurrence step j: forward j; reversed: len-1-j inside the row's length, j in
// the right padding (reverse_sequences leaves padding where it is).
__device__ __forceinline__ int time_index(int j, int reverse, const int32_t* lens, int row) {
  if (!reverse) return j;
  const int n = lens[row];
  return j < n ? n - 1 - j : j;
}

// ===================================================================================================================
// LSTM
// ===================================================================================================================
struct LstmFwdArgs {
  const float *hprev, *cprev;   // [B,H]
  const float* Whh;             // [4H,H] rows [i|f|g|o], in T16 (common.h)
  const float* bhh;             // [4H]
  const float* xg;              // [B,4H] input projection incl. b_ih
  const int32_t* lens;          // [B] valid steps per row (packed-sequence semantics) or null
  float *hnext, *cnext;         // [B,H]
  float* out;                   // [B,H]   zero where the row is past its length (pad_packed_sequence)
  float* gates;                 // [B,4H]  saved i,f,g,o (zero where masked)
  int B, H, t;
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void lstm_fwd_kernel(const float* hprev, const float* cprev, const float* Whh,
                                                           const float* bhh, const float* xg, const int32_t* lens, unsigned b_h,
                                                           int t_step, float* hnext, float* cnext, float* out, float* gates) {
  // scalar arguments (LstmFwdArgs documents them): the 6 input pointers, B:16|H:16 and the step are the 14 dwords the command
  // processor preloads into SGPRs (stages.h lin1_stage_kernel); the outputs come by s_load
  const int B = b_h & 0xffff, H = b_h >> 16;
  __shared__ float red[4 * NW * 256];
  const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16, wave = threadIdx.x >> 6;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o = (size_t)rowc * H + col, o4 = (size_t)rowc * 4 * H + col;
  const float x0 = xg[o4] + bhh[col], x1 = xg[o4 + H] + bhh[H + col];
  const float x2 = xg[o4 + 2 * H] + bhh[2 * H + col], x3 = xg[o4 + 3 * H] + bhh[3 * H + col];
  const float hp = hprev[o], cp = cprev[o];
  const bool live = lens == nullptr || t_step < lens[rowc];
  f32x4 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const float* const As[4] = {hprev, hprev, hprev, hprev};
    const float* const Ws[4] = {Whh, Whh, Whh, Whh};
    const int ld[4] = {H, H, H, H}, cs[4] = {c0, H + c0, 2 * H + c0, 3 * H + c0};
    wave_gemm16_multi<NW, 4, true>(As, ld, r0, B, Ws, ld, cs, H, wave, acc);
  }
  float v[4];
  reduce_tiles<4, NW>(acc, red, v);
  if (!own) return;
  const float i = sigmoidf_(v[0] + x0), f = sigmoidf_(v[1] + x1), g = tanhf(v[2] + x2), og = sigmoidf_(v[3] + x3);
  const float c2 = f * cp + i * g;
  const float h2 = og * tanhf(c2);
  cnext[o] = live ? c2 : cp;
  hnext[o] = live ? h2 : hp;
  out[o] = live ? h2 : 0.f;
  gates[o4] = live ? i : 0.f;
  gates[o4 + H] = live ? f : 0.f;
  gates[o4 + 2 * H] = live ? g : 0.f;
  gates[o4 + 3 * H] = live ? og : 0.f;
}

struct LstmBwdArgs {
  const float* DGn;     // [B,4H] gate pre-activation grads of step s+1 (unused when has_gemm == 0)
  const float* WhhT;    // [H,4H] in T16
  const float* dout;    // [B,H] grad wrt out_s
  const float* gates;   // [B,4H] step s
  const float *c_s, *c_s1;  // cell state entering / leaving step s
  float* DC;            // [B,H] running grad wrt the cell state (in/out)
  float* DG;            // [B,4H] out: gate pre-activation grads of step s
  float* dh0;           // [B,H] out when has_gates == 0
  int B, H, has_gemm, has_gates;
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void lstm_bwd_kernel(const float* DGn, const float* WhhT, const float* dout,
                                                           const float* gates, const float* c_s, float* DC, unsigned b_h,
                                                           unsigned has, float* DG, float* dh0) {
  // scalar arguments (LstmBwdArgs documents them; c_s1 = c_s + B*H, consecutive steps of the saved cell states): 14 preloaded dwords
  const int B = b_h & 0xffff, H0 = b_h >> 16;
  const int has_gemm = has & 1, has_gates = (has >> 1) & 1;
  const float* c_s1 = c_s + (size_t)B * H0;
  __shared__ float red[NW * 256];
  const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16, H = H0;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o = (size_t)rowc * H + col, o4 = (size_t)rowc * 4 * H + col;
  float dh = 0.f, ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cs = 0.f, cs1 = 0.f, dc = 0.f;
  if (has_gates) {  // wave-uniform
    dh = dout[o];
    ig = gates[o4]; fg = gates[o4 + H]; gg = gates[o4 + 2 * H]; og = gates[o4 + 3 * H];
    cs = c_s[o]; cs1 = c_s1[o]; dc = DC[o];
  }
  float v[1] = {0.f};
  if (has_gemm) {
    f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
    acc[0] = wave_gemm16<NW, true>(DGn, 4 * H, r0, B, WhhT, 4 * H, c0, 4 * H, threadIdx.x >> 6, acc[0]);
    reduce_tiles<1, NW>(acc, red, v);
  }
  if (!own) return;
  dh += v[0];
  if (!has_gates) { dh0[o] = dh; return; }
  const float tc = tanhf(cs1);
  const float d_o = dh * tc;
  const float dct = dc + dh * og * (1.f - tc * tc);
  DG[o4] = dct * gg * ig * (1.f - ig);
  DG[o4 + H] = dct * cs * fg * (1.f - fg);
  DG[o4 + 2 * H] = dct * ig * (1.f - gg * gg);
  DG[o4 + 3 * H] = d_o * og * (1.f - og);
  DC[o] = dct * fg;
}

// sequences of at most this many rows may run as ONE persistent launch (pchain.h): their buffers get the T16 operand copies
inline size_t t16_rows(int B) { return (size_t)((B + 15) / 16) * 16; }
// Backward sequences take the register-resident kernel up to this K (= 3R | 4H).  At K >= 1536 a step is bound by every tile pulling
// the whole [B, K] gradient slab through the fabric (GRU R = 512: 12 MB per step); there the kernel reads the slab once per XCD
// (seqchain.hip, SHARED): GRU R = 512 backward 5.8 (engine tile) / 6.2 (every tile its own sc1 reads) -> ~5.5 us per step
constexpr int kSeqRegsMaxKBwd = 1536;
inline bool seq_persistent(int T, int B) { return pchain_applies(B) && device_cus() >= 32 && T >= 4; }

struct LstmReserve { float *XG, *Hs, *Cs, *GATES, *WhhP, *H16; };  // WhhP: T16 copy of Whh; H16: (T+1) T16 slabs of the state
size_t carve_lstm(float* base, int T, int B, int H, LstmReserve* r) {
  size_t off = 0;
  auto take = [&](size_t cnt) { float* p = base ? base + off : nullptr; off += (cnt + 3) & ~(size_t)3; return p; };
  LstmReserve t;
  t.XG = take((size_t)T * B * 4 * H);
  t.Hs = take((size_t)(T + 1) * B * H);
  t.Cs = take((size_t)(T + 1) * B * H);
  t.GATES = take((size_t)T * B * 4 * H);
  t.WhhP = take((size_t)4 * H * H);
  t.H16 = B <= kPchainCarveMaxB ? take((size_t)(T + 1) * t16_rows(B) * H) : nullptr;
  if (r) *r = t;
  return off;
}
struct LstmWs { float *WhhT, *DG, *DC, *DG16; };
size_t carve_lstm_ws(float* base, int T, int B, int H, LstmWs* w) {
  size_t off = 0;
  auto take = [&](size_t cnt) { float* p = base ? base + off : nullptr; off += (cnt + 3) & ~(size_t)3; return p; };
  LstmWs t;
  t.WhhT = take((size_t)H * 4 * H);
  t.DG = take((size_t)T * B * 4 * H);
  t.DC = take((size_t)B * H);
  t.DG16 = B <= kPchainCarveMaxB ? take((size_t)(T + 1) * t16_rows(B) * 4 * H) : nullptr;
  if (w) *w = t;
  return off;
}

// ===================================================================================================================
// GRU
// ===================================================================================================================
struct GruFwdArgs {
  const float* hprev;    // [B,R] state entering recurrence step j
  const float* Whh;      // [3R,R] rows [r|z|n], in T16
  const float* bhh;      // [3R]
  const float* xg;       // [T,B,3R] input projection incl. b_ih, TIME indexed
  const int32_t* lens;   // [B] (reverse map) or null
  float* hnext;          // [B,R]
  float* out;            // time-indexed output base: element (idx,row,col) at out + idx*out_ts + row*out_ld + col
  float *rg, *ug, *ng, *ghn;  // [B,R] saves of recurrence step j
  long long out_ts;
  int out_ld, B, R, j, reverse;
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void gru_fwd_kernel(const float* hprev, const float* Whh, const float* bhh,
                                                          const float* xg, const int32_t* lens, unsigned b_r, unsigned j_rev,
                                                          float* hnext, float* out, float* rg, float* ug, float* ng, float* ghn,
                                                          long long out_ts, int out_ld) {
  // scalar arguments (GruFwdArgs documents them): the 5 input pointers, B:16|R:16 and j<<1|reverse are the first 12 dwords,
  // preloaded into SGPRs (stages.h lin1_stage_kernel); the outputs come by s_load
  const int B = b_r & 0xffff, R0 = b_r >> 16, j = j_rev >> 1, reverse = j_rev & 1;
  __shared__ float red[3 * NW * 256];
  const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16, wave = threadIdx.x >> 6, R = R0;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const int idx = time_index(j, reverse, lens, rowc);
  const size_t o = (size_t)rowc * R + col;
  const size_t ox = ((size_t)idx * B + rowc) * 3 * R + col;
  const float x0 = xg[ox], x1 = xg[ox + R], x2 = xg[ox + 2 * R];
  const float b0 = bhh[col], b1 = bhh[R + col], b2 = bhh[2 * R + col];
  const float hp = hprev[o];
  f32x4 acc[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const float* const As[3] = {hprev, hprev, hprev};
    const float* const Ws[3] = {Whh, Whh, Whh};
    const int ld[3] = {R, R, R}, cs[3] = {c0, R + c0, 2 * R + c0};
    wave_gemm16_multi<NW, 3, true>(As, ld, r0, B, Ws, ld, cs, R, wave, acc);
  }
  float v[3];
  reduce_tiles<3, NW>(acc, red, v);
  if (!own) return;
  const float hn = v[2] + b2;
  const float r = sigmoidf_(x0 + v[0] + b0);
  const float u = sigmoidf_(x1 + v[1] + b1);
  const float n = tanhf(x2 + r * hn);
  const float h2 = (1.f - u) * n + u * hp;
  hnext[o] = h2;
  out[(size_t)idx * out_ts + (size_t)row * out_ld + col] = h2;
  rg[o] = r; ug[o] = u; ng[o] = n; ghn[o] = hn;
}

struct GruBwdArgs {
  const float* DGHn;    // [B,3R] hidden-projection grads of recurrence step j+1 (unused when has_gemm == 0)
  const float* WhhT;    // [R,3R] in T16
  const float* dout;    // time-indexed grad wrt the outputs (same addressing as GruFwdArgs::out)
  const float *rg, *ug, *ng, *ghn, *hprev;  // saves of step j; hprev = state entering step j
  const int32_t* lens;
  float* G;             // [B,R] running grad through the u-gate path (in/out)
  float* DGI;           // [T,B,3R] TIME indexed: grads wrt the input projection
  float* DGH;           // [B,3R] step j: grads wrt the hidden projection
  float* dh0;           // [B,R] out when has_gates == 0
  long long out_ts;
  int out_ld, B, R, j, reverse, has_gemm, has_gates;
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void gru_bwd_kernel(const float* DGHn, const float* WhhT, float* G, unsigned b_r, unsigned has,
                                                          GruBwdArgs a) {
  // the leading scalars (operand pointers, G, sizes, flags) are preloaded into SGPRs; the struct comes by s_load and the saves of
  // the step are prefetched by wave_gemm16's `mid` hook, after the operand loads have been issued (stages.h head_stage_kernel)
  __shared__ float red[NW * 256];
  const int B = b_r & 0xffff, R = b_r >> 16;
  const int has_gemm = has & 1, has_gates = (has >> 1) & 1;
  const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o = (size_t)rowc * R + col;
  float g = G[o];
  float dout = 0.f, r = 0.f, u = 0.f, n = 0.f, hn = 0.f, hp = 0.f;
  int idx = 0;
  auto prefetch = [&]() {
    if (has_gates) {  // wave-uniform
      idx = time_index(a.j, a.reverse, a.lens, rowc);
      dout = a.dout[(size_t)idx * a.out_ts + (size_t)rowc * a.out_ld + col];
      r = a.rg[o]; u = a.ug[o]; n = a.ng[o]; hn = a.ghn[o]; hp = a.hprev[o];
    }
  };
  float v[1] = {0.f};
  if (has_gemm) {
    f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
    acc[0] = wave_gemm16<NW, true>(DGHn, 3 * R, r0, B, WhhT, 3 * R, c0, 3 * R, threadIdx.x >> 6, acc[0], prefetch);
    reduce_tiles<1, NW>(acc, red, v);
  } else {
    prefetch();
  }
  if (!own) return;
  g += dout;
  g += v[0];
  if (!has_gates) { a.dh0[o] = g; return; }
  const float dn_pre = g * (1.f - u) * (1.f - n * n);
  const float du_pre = g * (hp - n) * u * (1.f - u);
  const float dr_pre = dn_pre * hn * r * (1.f - r);
  const size_t oi = ((size_t)idx * B + row) * 3 * R + col, oh = (size_t)row * 3 * R + col;
  a.DGI[oi] = dr_pre; a.DGI[oi + R] = du_pre; a.DGI[oi + 2 * R] = dn_pre;
  a.DGH[oh] = dr_pre; a.DGH[oh + R] = du_pre; a.DGH[oh + 2 * R] = dn_pre * r;
  G[o] = g * u;
}

struct GruReserve { float *XG, *Hs, *RG, *UG, *NG, *GHN, *WhhP, *H16; };  // WhhP: T16 copy of Whh; H16: (T+1) T16 slabs of the state
size_t carve_gru(float* base, int T, int B, int R, GruReserve* r) {
  size_t off = 0;
  auto take = [&](size_t cnt) { float* p = base ? base + off : nullptr; off += (cnt + 3) & ~(size_t)3; return p; };
  GruReserve t;
  const size_t n = (size_t)T * B;
  t.XG = take(n * 3 * R);
  t.Hs = take((size_t)(T + 1) * B * R);
  t.RG = take(n * R); t.UG = take(n * R); t.NG = take(n * R); t.GHN = take(n * R);
  t.WhhP = take((size_t)3 * R * R);
  t.H16 = B <= kPchainCarveMaxB ? take((size_t)(T + 1) * t16_rows(B) * R) : nullptr;
  if (r) *r = t;
  return off;
}
struct GruWs { float *WhhT, *DGI, *DGH, *G, *DGH16; };
size_t carve_gru_ws(float* base, int T, int B, int R, GruWs* w) {
  size_t off = 0;
  auto take = [&](size_t cnt) { float* p = base ? base + off : nullptr; off += (cnt + 3) & ~(size_t)3; return p; };
  GruWs t;
  const size_t n = (size_t)T * B;
  t.WhhT = take((size_t)R * 3 * R);
  t.DGI = take(n * 3 * R);
  t.DGH = take(n * 3 * R);
  t.G = take((size_t)B * R);
  t.DGH16 = B <= kPchainCarveMaxB ? take((size_t)(T + 1) * t16_rows(B) * 3 * R) : nullptr;
  if (w) *w = t;
  return off;
}

int check_rnn(int T, int B, int I, int H) {
  BLVM_REQUIRE(T > 0 && B > 0 && I > 0 && H > 0, "rnn: bad shape T=%d B=%d I=%d H=%d", T, B, I, H);
  BLVM_REQUIRE(H % 16 == 0, "rnn: hidden size must be a multiple of 16 (got %d)", H);
  BLVM_REQUIRE((B + 15) / 16 <= 65535, "rnn: batch too large");
  return BLVM_OK;
}

}  // namespace
}  // namespace blvm

using namespace blvm;

// --------------------------------------------------------------------------------------------------------------------
extern "C" size_t blvm_lstm_reserve_floats(int T, int B, int H) { return carve_lstm(nullptr, T, B, H, nullptr); }
extern "C" size_t blvm_lstm_bwd_workspace_floats(int T, int B, int H) { return carve_lstm_ws(nullptr, T, B, H, nullptr); }

extern "C" int blvm_lstm_seq_fwd(const float* Wih, const float* Whh, const float* bih, const float* bhh, const float* in,
                                 const float* h0, const float* c0, const int32_t* lens, int T, int B, int I, int H,
                                 float* out, float* hn, float* cn, float* reserve, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_rnn(T, B, I, H);
  if (rc) return rc;
  BLVM_REQUIRE(Wih && Whh && bih && bhh && in && out && reserve, "lstm_fwd: null pointer");
  BLVM_REQUIRE(aligned16(reserve) && aligned16(Whh), "lstm_fwd: buffers must be 16-byte aligned");
  BLVM_REQUIRE(B < 65536 && H < 65536, "lstm_fwd: B and H must be below 65536 (packed kernel arguments)");
  LstmReserve rs;
  carve_lstm(reserve, T, B, H, &rs);
  const size_t n = (size_t)T * B, bh = (size_t)B * H;
  rc = gemm_f32(0, 0, (int)n, 4 * H, I, in, I, Wih, I, rs.XG, 4 * H, bih, 0, 0.f, nullptr, 0, 0, 1, s);
  if (rc) return rc;
  if (h0) BLVM_HIP(hipMemcpyAsync(rs.Hs, h0, sizeof(float) * bh, hipMemcpyDeviceToDevice, s));
  else BLVM_HIP(hipMemsetAsync(rs.Hs, 0, sizeof(float) * bh, s));
  if (c0) BLVM_HIP(hipMemcpyAsync(rs.Cs, c0, sizeof(float) * bh, hipMemcpyDeviceToDevice, s));
  else BLVM_HIP(hipMemsetAsync(rs.Cs, 0, sizeof(float) * bh, s));
  const bool bf16_seq = pchain_bf16(B) && seq_persistent(T, B);  // bf16-operand mode of the persistent path
  T16PackScope pack_scope(bf16_seq, s);
  rc = t16_pack_rows(Whh, H, 4 * H, H, rs.WhhP, s);  // operand layout of the chain (once per sequence)
  if (rc) return rc;
  rc = pack_scope.flush();
  if (rc) return rc;
  if (seq_persistent(T, B) && (seq_regs_mask() & 1) && seq_regs_applies(H, 4 * H, H, B, 4)) {
    // one persistent launch with the workgroup's weight slice in registers (seqchain.hip)
    const long xH = (long)((B + 15) / 16) * 16 * H;
    SeqLstmFwd q{rs.H16, rs.WhhP, bhh, rs.XG, lens, rs.Hs, rs.Cs, out, rs.GATES, T, B, H, bf16_seq ? 1 : 0, {}};
    rc = pchain_ctl(&q.ctl.dev, &q.ctl.host, &q.ctl.epoch);
    if (rc) return rc;
    BLVM_HIP(pchain_fill_sentinel(rs.H16 + xH, sizeof(float) * (size_t)T * xH, s));
    rc = pchain_rows_to_t16(rs.Hs, H, B, H, rs.H16, s);
    if (rc) return rc;
    rc = seq_lstm_fwd(q, s);
    if (rc) return rc;
    if (hn) BLVM_HIP(hipMemcpyAsync(hn, rs.Hs + T * bh, sizeof(float) * bh, hipMemcpyDeviceToDevice, s));
    if (cn) BLVM_HIP(hipMemcpyAsync(cn, rs.Cs + T * bh, sizeof(float) * bh, hipMemcpyDeviceToDevice, s));
    return BLVM_OK;
  }
  if (seq_persistent(T, B)) {
    // one persistent launch for the whole sequence (pchain.hip): one link per step — the hidden projection with the gate math
    using namespace pchain;
    const int rt = (B + 15) / 16, ctH = H / 16;
    const long sH = (long)bh, s4H = 4 * sH, xH = (long)rt * 16 * H;
    Builder bld;
    bld.p.bf16 = bf16_seq; bld.p.S = T; bld.p.B = B; bld.p.xcd = (pchain_tune() & 4) ? 1 : 0; bld.p.lds_products = 4;
    bld.p.prof = pchain_profile_buffer(); bld.p.prof_wg = 1;
    Desc& d = bld.add(K_LSTMS, ctH, 0, range_for(ctH * rt, device_cus() & ~7), H, 0, 0, T);
    bld.ptr(d, 0, rs.H16, xH); bld.ptr(d, 1, rs.WhhP); bld.ptr(d, 2, bhh); bld.ptr(d, 3, rs.XG, s4H); bld.ptr(d, 4, lens);
    bld.ptr(d, 5, rs.Hs, sH); bld.ptr(d, 6, rs.Hs + sH, sH); bld.ptr(d, 7, rs.H16 + xH, xH); bld.ptr(d, 8, rs.Cs, sH); bld.ptr(d, 9, rs.Cs + sH, sH);
    bld.ptr(d, 10, out, sH); bld.ptr(d, 11, rs.GATES, s4H);
    d.ld[3] = H; d.n16[0] = ctH; d.i[0] = H;
    rc = pchain_ctl(&bld.p.ctl.dev, &bld.p.ctl.host, &bld.p.ctl.epoch);
    if (rc) return rc;
    BLVM_HIP(pchain_fill_sentinel(rs.H16 + xH, sizeof(float) * (size_t)T * xH, s));
    rc = pchain_rows_to_t16(rs.Hs, H, B, H, rs.H16, s);
    if (rc) return rc;
    rc = pchain_launch(bld.p, s);
    if (rc) return rc;
    if (hn) BLVM_HIP(hipMemcpyAsync(hn, rs.Hs + T * bh, sizeof(float) * bh, hipMemcpyDeviceToDevice, s));
    if (cn) BLVM_HIP(hipMemcpyAsync(cn, rs.Cs + T * bh, sizeof(float) * bh, hipMemcpyDeviceToDevice, s));
    return BLVM_OK;
  }
  const int nw = pick_nw(H, 4);
  const dim3 grid(H / 16, (B + 15) / 16);
  for (int t = 0; t < T; ++t) {
    const float *hp = rs.Hs + t * bh, *cp = rs.Cs + t * bh, *xg_t = rs.XG + (size_t)t * B * 4 * H;
    float *hnx = rs.Hs + (t + 1) * bh, *cnx = rs.Cs + (t + 1) * bh, *out_t = out + t * bh, *gates_t = rs.GATES + (size_t)t * B * 4 * H;
    const unsigned b_h = (unsigned)B | ((unsigned)H << 16);
    if (nw == 16) hipLaunchKernelGGL((lstm_fwd_kernel<16>), grid, dim3(1024), 0, s, hp, cp, (const float*)rs.WhhP, bhh, xg_t, lens, b_h, t, hnx, cnx, out_t, gates_t);
    else if (nw == 8) hipLaunchKernelGGL((lstm_fwd_kernel<8>), grid, dim3(512), 0, s, hp, cp, (const float*)rs.WhhP, bhh, xg_t, lens, b_h, t, hnx, cnx, out_t, gates_t);
    else hipLaunchKernelGGL((lstm_fwd_kernel<4>), grid, dim3(256), 0, s, hp, cp, (const float*)rs.WhhP, bhh, xg_t, lens, b_h, t, hnx, cnx, out_t, gates_t);
  }
  BLVM_CHECK_LAUNCH("lstm_seq_fwd");
  if (hn) BLVM_HIP(hipMemcpyAsync(hn, rs.Hs + T * bh, sizeof(float) * bh, hipMemcpyDeviceToDevice, s));
  if (cn) BLVM_HIP(hipMemcpyAsync(cn, rs.Cs + T * bh, sizeof(float) * bh, hipMemcpyDeviceToDevice, s));
  return BLVM_OK;
}

extern "C" int blvm_lstm_seq_bwd(const float* Wih, const float* Whh, const float* in, const float* reserve,
                                 const float* d_out, int T, int B, int I, int H, float* d_in, float* d_h0, float* d_c0,
                                 float* dWih, float* dWhh, float* dbih, float* dbhh, float* workspace, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_rnn(T, B, I, H);
  if (rc) return rc;
  BLVM_REQUIRE(Wih && Whh && in && reserve && d_out && workspace, "lstm_bwd: null pointer");
  BLVM_REQUIRE(aligned16(reserve) && aligned16(workspace), "lstm_bwd: buffers must be 16-byte aligned");
  BLVM_REQUIRE(B < 65536 && H < 65536, "lstm_bwd: B and H must be below 65536 (packed kernel arguments)");
  LstmReserve rs;
  carve_lstm(const_cast<float*>(reserve), T, B, H, &rs);
  LstmWs ws;
  carve_lstm_ws(workspace, T, B, H, &ws);
  const size_t n = (size_t)T * B, bh = (size_t)B * H;
  const bool bf16_seq = pchain_bf16(B) && seq_persistent(T, B);  // bf16-operand mode of the persistent path
  T16PackScope pack_scope(bf16_seq, s);
  rc = t16_pack_transposed(Whh, H, 4 * H, H, ws.WhhT, s);
  if (rc) return rc;
  rc = pack_scope.flush();
  if (rc) return rc;
  BLVM_HIP(hipMemsetAsync(ws.DC, 0, sizeof(float) * bh, s));
  if (seq_persistent(T, B) && (seq_regs_mask() & 2) && 4 * H <= kSeqRegsMaxKBwd && seq_regs_applies(H, 4 * H, H, B, 4)) {
    const long x4H = (long)((B + 15) / 16) * 16 * 4 * H;
    SeqLstmBwd q{ws.DG16, ws.WhhT, d_out, rs.GATES, rs.Cs, ws.DC, ws.DG, d_h0 ? d_h0 : ws.DC, T, B, H, d_h0 ? T + 1 : T, bf16_seq ? 1 : 0, {}};
    rc = pchain_ctl(&q.ctl.dev, &q.ctl.host, &q.ctl.epoch);
    if (rc) return rc;
    BLVM_HIP(pchain_fill_sentinel(ws.DG16, sizeof(float) * (size_t)T * x4H, s));
    rc = seq_lstm_bwd(q, s);
    if (rc) return rc;
  } else if (seq_persistent(T, B)) {
    using namespace pchain;
    const int rt = (B + 15) / 16, ctH = H / 16;
    const long sH = (long)bh, s4H = 4 * sH, x4H = (long)rt * 16 * 4 * H;
    Builder bld;
    bld.p.bf16 = bf16_seq; bld.p.S = d_h0 ? T + 1 : T; bld.p.B = B; bld.p.xcd = (pchain_tune() & 4) ? 1 : 0; bld.p.lds_products = 1;
    bld.p.prof = pchain_profile_buffer() ? pchain_profile_buffer() + 64 : nullptr; bld.p.prof_wg = 1;
    Desc& d = bld.add(K_LSTMSB, ctH, 0, range_for(ctH * rt, device_cus() & ~7), 4 * H, 0, 0, T + 1);
    // step s handles t = T-1-s: time-indexed slabs start at the last step and walk backwards; the T16 slabs are indexed by s
    bld.ptr(d, 0, ws.DG16 - x4H, x4H); bld.ptr(d, 1, ws.WhhT); bld.ptr(d, 2, d_out + (long)(T - 1) * sH, -sH);
    bld.ptr(d, 3, rs.GATES + (long)(T - 1) * s4H, -s4H); bld.ptr(d, 4, rs.Cs + (long)(T - 1) * sH, -sH); bld.ptr(d, 5, ws.DC);
    bld.ptr(d, 6, ws.DG + (long)(T - 1) * s4H, -s4H); bld.ptr(d, 7, ws.DG16, x4H); bld.ptr(d, 8, d_h0 ? d_h0 : ws.DC);
    d.ld[3] = 4 * H; d.n16[0] = 4 * ctH; d.n16[1] = T; d.i[0] = H;
    rc = pchain_ctl(&bld.p.ctl.dev, &bld.p.ctl.host, &bld.p.ctl.epoch);
    if (rc) return rc;
    BLVM_HIP(pchain_fill_sentinel(ws.DG16, sizeof(float) * (size_t)T * x4H, s));
    rc = pchain_launch(bld.p, s);
    if (rc) return rc;
  } else {
  const int nw = pick_nw(4 * H, 1);
  const dim3 grid(H / 16, (B + 15) / 16);
  // scratch for dh0 when the caller does not want it
  for (int st = T - 1; st >= -1; --st) {
    const unsigned has = (st < T - 1 ? 1u : 0u) | (st >= 0 ? 2u : 0u);
    const int sg = st >= 0 ? st : 0;
    const float* DGn = ws.DG + (size_t)(st + 1 < T ? st + 1 : 0) * B * 4 * H;
    const float *dout_s = d_out + sg * bh, *gates_s = rs.GATES + (size_t)sg * B * 4 * H, *c_s = rs.Cs + sg * bh;
    float* DG_s = ws.DG + (size_t)sg * B * 4 * H;
    const unsigned b_h = (unsigned)B | ((unsigned)H << 16);
    if (st == -1 && d_h0 == nullptr) break;
    if (nw == 16) hipLaunchKernelGGL((lstm_bwd_kernel<16>), grid, dim3(1024), 0, s, DGn, (const float*)ws.WhhT, dout_s, gates_s, c_s, ws.DC, b_h, has, DG_s, d_h0);
    else if (nw == 8) hipLaunchKernelGGL((lstm_bwd_kernel<8>), grid, dim3(512), 0, s, DGn, (const float*)ws.WhhT, dout_s, gates_s, c_s, ws.DC, b_h, has, DG_s, d_h0);
    else hipLaunchKernelGGL((lstm_bwd_kernel<4>), grid, dim3(256), 0, s, DGn, (const float*)ws.WhhT, dout_s, gates_s, c_s, ws.DC, b_h, has, DG_s, d_h0);
  }
  }
  BLVM_CHECK_LAUNCH("lstm_seq_bwd");
  if (d_c0) BLVM_HIP(hipMemcpyAsync(d_c0, ws.DC, sizeof(float) * bh, hipMemcpyDeviceToDevice, s));
  if (d_in) {
    rc = gemm_f32(0, 1, (int)n, I, 4 * H, ws.DG, 4 * H, Wih, I, d_in, I, nullptr, 0, 0.f, nullptr, 0, 0, 1, s);
    if (rc) return rc;
  }
  // weight + bias gradients: the bias sums ride on the weight-gradient GEMMs (gemm.hip: column sums of the staged D tiles)
  if (dWih) { rc = gemm_f32(1, 1, 4 * H, I, (int)n, ws.DG, 4 * H, in, I, dWih, I, nullptr, 0, 0.f, nullptr, 0, 1, pick_split(4 * H, I, (int)n), s, dbih); if (rc) return rc; }
  else if (dbih) { rc = colsum_f32((int)n, 4 * H, ws.DG, 4 * H, dbih, 1, s); if (rc) return rc; }
  if (dWhh) { rc = gemm_f32(1, 1, 4 * H, H, (int)n, ws.DG, 4 * H, rs.Hs, H, dWhh, H, nullptr, 0, 0.f, nullptr, 0, 1, pick_split(4 * H, H, (int)n), s, dbhh); if (rc) return rc; }
  else if (dbhh) { rc = colsum_f32((int)n, 4 * H, ws.DG, 4 * H, dbhh, 1, s); if (rc) return rc; }
  return BLVM_OK;
}

// --------------------------------------------------------------------------------------------------------------------
extern "C" size_t blvm_gru_reserve_floats(int T, int B, int R) { return carve_gru(nullptr, T, B, R, nullptr); }
extern "C" size_t blvm_gru_bwd_workspace_floats(int T, int B, int R) { return carve_gru_ws(nullptr, T, B, R, nullptr); }

extern "C" int blvm_gru_seq_fwd(const float* Wih, const float* Whh, const float* bih, const float* bhh, const float* in,
                                int ld_in, const float* h0, const int32_t* lens, int reverse, int T, int B, int I, int R,
                                float* out, long long out_ts, int out_ld, float* hn, float* reserve, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_rnn(T, B, I, R);
  if (rc) return rc;
  BLVM_REQUIRE(Wih && Whh && bih && bhh && in && out && reserve, "gru_fwd: null pointer");
  BLVM_REQUIRE(!reverse || lens, "gru_fwd: reverse needs lens");
  BLVM_REQUIRE(aligned16(reserve) && aligned16(Whh), "gru_fwd: buffers must be 16-byte aligned");
  BLVM_REQUIRE(B < 65536 && R < 65536, "gru_fwd: B and R must be below 65536 (packed kernel arguments)");
  GruReserve rs;
  carve_gru(reserve, T, B, R, &rs);
  const size_t n = (size_t)T * B, br = (size_t)B * R;
  rc = gemm_f32(0, 0, (int)n, 3 * R, I, in, ld_in, Wih, I, rs.XG, 3 * R, bih, 0, 0.f, nullptr, 0, 0, 1, s);
  if (rc) return rc;
  if (h0) BLVM_HIP(hipMemcpyAsync(rs.Hs, h0, sizeof(float) * br, hipMemcpyDeviceToDevice, s));
  else BLVM_HIP(hipMemsetAsync(rs.Hs, 0, sizeof(float) * br, s));
  const bool bf16_seq = pchain_bf16(B) && seq_persistent(T, B);  // bf16-operand mode of the persistent path
  T16PackScope pack_scope(bf16_seq, s);
  rc = t16_pack_rows(Whh, R, 3 * R, R, rs.WhhP, s);  // operand layout of the chain (once per sequence)
  if (rc) return rc;
  rc = pack_scope.flush();
  if (rc) return rc;
  if (seq_persistent(T, B) && (seq_regs_mask() & 1) && seq_regs_applies(R, 3 * R, R, B, 3)) {
    const long xR = (long)((B + 15) / 16) * 16 * R;
    SeqGruFwd q{rs.H16, rs.WhhP, bhh, rs.XG, lens, rs.Hs, out, rs.RG, rs.UG, rs.NG, rs.GHN, (long)out_ts, out_ld, T, B, R, reverse ? 1 : 0, bf16_seq ? 1 : 0, {}};
    rc = pchain_ctl(&q.ctl.dev, &q.ctl.host, &q.ctl.epoch);
    if (rc) return rc;
    BLVM_HIP(pchain_fill_sentinel(rs.H16 + xR, sizeof(float) * (size_t)T * xR, s));
    rc = pchain_rows_to_t16(rs.Hs, R, B, R, rs.H16, s);
    if (rc) return rc;
    rc = seq_gru_fwd(q, s);
    if (rc) return rc;
    if (hn) BLVM_HIP(hipMemcpyAsync(hn, rs.Hs + T * br, sizeof(float) * br, hipMemcpyDeviceToDevice, s));
    return BLVM_OK;
  }
  if (seq_persistent(T, B)) {
    using namespace pchain;
    BLVM_REQUIRE(out_ts >= 0 && out_ts < (1ll << 31), "gru_fwd: output step stride out of range");
    const int rt = (B + 15) / 16, ctR = R / 16;
    const long sR = (long)br, xR = (long)rt * 16 * R;
    Builder bld;
    bld.p.bf16 = bf16_seq; bld.p.S = T; bld.p.B = B; bld.p.xcd = (pchain_tune() & 4) ? 1 : 0; bld.p.lds_products = 3;
    bld.p.prof = pchain_profile_buffer(); bld.p.prof_wg = 1;
    Desc& d = bld.add(K_GRUS, ctR, 0, range_for(ctR * rt, device_cus() & ~7), R, 0, 0, T);
    bld.ptr(d, 0, rs.H16, xR); bld.ptr(d, 1, rs.WhhP); bld.ptr(d, 2, bhh); bld.ptr(d, 3, rs.XG); bld.ptr(d, 4, lens); bld.ptr(d, 5, rs.Hs, sR);
    bld.ptr(d, 6, rs.Hs + sR, sR); bld.ptr(d, 7, rs.H16 + xR, xR); bld.ptr(d, 8, out); bld.ptr(d, 9, rs.RG, sR); bld.ptr(d, 10, rs.UG, sR);
    bld.ptr(d, 11, rs.NG, sR); bld.ptr(d, 12, rs.GHN, sR);
    d.ld[3] = R; d.n16[0] = ctR; d.i[0] = R; d.i[1] = reverse ? 1 : 0; d.i[2] = (int)out_ts; d.i[3] = out_ld;
    rc = pchain_ctl(&bld.p.ctl.dev, &bld.p.ctl.host, &bld.p.ctl.epoch);
    if (rc) return rc;
    BLVM_HIP(pchain_fill_sentinel(rs.H16 + xR, sizeof(float) * (size_t)T * xR, s));
    rc = pchain_rows_to_t16(rs.Hs, R, B, R, rs.H16, s);
    if (rc) return rc;
    rc = pchain_launch(bld.p, s);
    if (rc) return rc;
    if (hn) BLVM_HIP(hipMemcpyAsync(hn, rs.Hs + T * br, sizeof(float) * br, hipMemcpyDeviceToDevice, s));
    return BLVM_OK;
  }
  const int nw = pick_nw(R, 3);
  const dim3 grid(R / 16, (B + 15) / 16);
  for (int j = 0; j < T; ++j) {
    const float* hp = rs.Hs + j * br;
    float *hnx = rs.Hs + (j + 1) * br, *rg_j = rs.RG + j * br, *ug_j = rs.UG + j * br, *ng_j = rs.NG + j * br, *ghn_j = rs.GHN + j * br;
    const unsigned b_r = (unsigned)B | ((unsigned)R << 16), j_rev = ((unsigned)j << 1) | (reverse ? 1u : 0u);
    if (nw == 16) hipLaunchKernelGGL((gru_fwd_kernel<16>), grid, dim3(1024), 0, s, hp, (const float*)rs.WhhP, bhh, (const float*)rs.XG, lens, b_r, j_rev, hnx, out, rg_j, ug_j, ng_j, ghn_j, out_ts, out_ld);
    else if (nw == 8) hipLaunchKernelGGL((gru_fwd_kernel<8>), grid, dim3(512), 0, s, hp, (const float*)rs.WhhP, bhh, (const float*)rs.XG, lens, b_r, j_rev, hnx, out, rg_j, ug_j, ng_j, ghn_j, out_ts, out_ld);
    else hipLaunchKernelGGL((gru_fwd_kernel<4>), grid, dim3(256), 0, s, hp, (const float*)rs.WhhP, bhh, (const float*)rs.XG, lens, b_r, j_rev, hnx, out, rg_j, ug_j, ng_j, ghn_j, out_ts, out_ld);
  }
  BLVM_CHECK_LAUNCH("gru_seq_fwd");
  if (hn) BLVM_HIP(hipMemcpyAsync(hn, rs.Hs + T * br, sizeof(float) * br, hipMemcpyDeviceToDevice, s));
  return BLVM_OK;
}

extern "C" int blvm_gru_seq_bwd(const float* Wih, const float* Whh, const float* in, int ld_in, const int32_t* lens,
                                int reverse, const float* reserve, const float* d_out, long long out_ts, int out_ld,
                                int T, int B, int I, int R, float* d_in, int ld_din, int accumulate_din, float* d_h0,
                                float* dWih, float* dWhh, float* dbih, float* dbhh, float* workspace, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_rnn(T, B, I, R);
  if (rc) return rc;
  BLVM_REQUIRE(Wih && Whh && in && reserve && d_out && workspace, "gru_bwd: null pointer");
  BLVM_REQUIRE(!reverse || lens, "gru_bwd: reverse needs lens");
  BLVM_REQUIRE(B < 65536 && R < 65536, "gru_bwd: B and R must be below 65536 (packed kernel arguments)");
  BLVM_REQUIRE(aligned16(reserve) && aligned16(workspace), "gru_bwd: buffers must be 16-byte aligned");
  GruReserve rs;
  carve_gru(const_cast<float*>(reserve), T, B, R, &rs);
  GruWs ws;
  carve_gru_ws(workspace, T, B, R, &ws);
  const size_t n = (size_t)T * B, br = (size_t)B * R;
  const bool bf16_seq = pchain_bf16(B) && seq_persistent(T, B);  // bf16-operand mode of the persistent path
  T16PackScope pack_scope(bf16_seq, s);
  rc = t16_pack_transposed(Whh, R, 3 * R, R, ws.WhhT, s);
  if (rc) return rc;
  rc = pack_scope.flush();
  if (rc) return rc;
  BLVM_HIP(hipMemsetAsync(ws.G, 0, sizeof(float) * br, s));
  if (seq_persistent(T, B) && (seq_regs_mask() & 2) && 3 * R <= kSeqRegsMaxKBwd && seq_regs_applies(R, 3 * R, R, B, 3)) {
    const long x3R = (long)((B + 15) / 16) * 16 * 3 * R;
    SeqGruBwd q{ws.DGH16, ws.WhhT, d_out, rs.RG, rs.UG, rs.NG, rs.GHN, rs.Hs, lens, ws.G, ws.DGI, ws.DGH, d_h0 ? d_h0 : ws.G, (long)out_ts, out_ld, T, B, R,
                reverse ? 1 : 0, d_h0 ? T + 1 : T, bf16_seq ? 1 : 0, {}};
    rc = pchain_ctl(&q.ctl.dev, &q.ctl.host, &q.ctl.epoch);
    if (rc) return rc;
    BLVM_HIP(pchain_fill_sentinel(ws.DGH16, sizeof(float) * (size_t)T * x3R, s));
    rc = seq_gru_bwd(q, s);
    if (rc) return rc;
  } else if (seq_persistent(T, B)) {
    using namespace pchain;
    BLVM_REQUIRE(out_ts >= 0 && out_ts < (1ll << 31), "gru_bwd: output step stride out of range");
    const int rt = (B + 15) / 16, ctR = R / 16;
    const long sR = (long)br, s3R = 3 * sR, x3R = (long)rt * 16 * 3 * R;
    Builder bld;
    bld.p.bf16 = bf16_seq; bld.p.S = d_h0 ? T + 1 : T; bld.p.B = B; bld.p.xcd = (pchain_tune() & 4) ? 1 : 0; bld.p.lds_products = 1;
    bld.p.prof = pchain_profile_buffer() ? pchain_profile_buffer() + 64 : nullptr; bld.p.prof_wg = 1;
    Desc& d = bld.add(K_GRUSB, ctR, 0, range_for(ctR * rt, device_cus() & ~7), 3 * R, 0, 0, T + 1);
    // step s handles recurrence step j = T-1-s: the saves walk backwards from their last slab; the T16 slabs are indexed by s
    bld.ptr(d, 0, ws.DGH16 - x3R, x3R); bld.ptr(d, 1, ws.WhhT); bld.ptr(d, 2, d_out);
    bld.ptr(d, 3, rs.RG + (long)(T - 1) * sR, -sR); bld.ptr(d, 4, rs.UG + (long)(T - 1) * sR, -sR); bld.ptr(d, 5, rs.NG + (long)(T - 1) * sR, -sR);
    bld.ptr(d, 6, rs.GHN + (long)(T - 1) * sR, -sR); bld.ptr(d, 7, rs.Hs + (long)(T - 1) * sR, -sR); bld.ptr(d, 8, lens); bld.ptr(d, 9, ws.G);
    bld.ptr(d, 10, ws.DGI); bld.ptr(d, 11, ws.DGH + (long)(T - 1) * s3R, -s3R); bld.ptr(d, 12, ws.DGH16, x3R); bld.ptr(d, 13, d_h0 ? d_h0 : ws.G);
    d.ld[3] = 3 * R; d.n16[0] = 3 * ctR; d.n16[1] = T; d.i[0] = R; d.i[1] = reverse ? 1 : 0; d.i[2] = (int)out_ts; d.i[3] = out_ld;
    rc = pchain_ctl(&bld.p.ctl.dev, &bld.p.ctl.host, &bld.p.ctl.epoch);
    if (rc) return rc;
    BLVM_HIP(pchain_fill_sentinel(ws.DGH16, sizeof(float) * (size_t)T * x3R, s));
    rc = pchain_launch(bld.p, s);
    if (rc) return rc;
  } else {
  const int nw = pick_nw(3 * R, 1);
  const dim3 grid(R / 16, (B + 15) / 16);
  for (int j = T - 1; j >= -1; --j) {
    if (j == -1 && d_h0 == nullptr) break;
    GruBwdArgs a;
    a.has_gemm = j < T - 1; a.has_gates = j >= 0;
    a.DGHn = ws.DGH + (size_t)(j + 1 < T ? j + 1 : 0) * B * 3 * R;
    a.WhhT = ws.WhhT; a.dout = d_out;
    const int jg = j >= 0 ? j : 0;
    a.rg = rs.RG + jg * br; a.ug = rs.UG + jg * br; a.ng = rs.NG + jg * br; a.ghn = rs.GHN + jg * br;
    a.hprev = rs.Hs + jg * br; a.lens = lens;
    a.G = ws.G; a.DGI = ws.DGI; a.DGH = ws.DGH + (size_t)jg * B * 3 * R; a.dh0 = d_h0;
    a.out_ts = out_ts; a.out_ld = out_ld; a.B = B; a.R = R; a.j = jg; a.reverse = reverse;
    {
      const unsigned b_r = (unsigned)B | ((unsigned)R << 16), has = (a.has_gemm ? 1u : 0u) | (a.has_gates ? 2u : 0u);
      if (nw == 16) hipLaunchKernelGGL((gru_bwd_kernel<16>), grid, dim3(1024), 0, s, a.DGHn, a.WhhT, a.G, b_r, has, a);
      else if (nw == 8) hipLaunchKernelGGL((gru_bwd_kernel<8>), grid, dim3(512), 0, s, a.DGHn, a.WhhT, a.G, b_r, has, a);
      else hipLaunchKernelGGL((gru_bwd_kernel<4>), grid, dim3(256), 0, s, a.DGHn, a.WhhT, a.G, b_r, has, a);
    }
  }
  }
  BLVM_CHECK_LAUNCH("gru_seq_bwd");
  if (d_in) {
    rc = gemm_f32(0, 1, (int)n, I, 3 * R, ws.DGI, 3 * R, Wih, I, d_in, ld_din, nullptr, 0, 0.f, nullptr, 0, accumulate_din, 1, s);
    if (rc) return rc;
  }
  if (dWih) { rc = gemm_f32(1, 1, 3 * R, I, (int)n, ws.DGI, 3 * R, in, ld_in, dWih, I, nullptr, 0, 0.f, nullptr, 0, 1, pick_split(3 * R, I, (int)n), s, dbih); if (rc) return rc; }
  else if (dbih) { rc = colsum_f32((int)n, 3 * R, ws.DGI, 3 * R, dbih, 1, s); if (rc) return rc; }
  if (dWhh) { rc = gemm_f32(1, 1, 3 * R, R, (int)n, ws.DGH, 3 * R, rs.Hs, R, dWhh, R, nullptr, 0, 0.f, nullptr, 0, 1, pick_split(3 * R, R, (int)n), s, dbhh); if (rc) return rc; }
  else if (dbhh) { rc = colsum_f32((int)n, 3 * R, ws.DGH, 3 * R, dbhh, 1, s); if (rc) return rc; }
  return BLVM_OK;
}
