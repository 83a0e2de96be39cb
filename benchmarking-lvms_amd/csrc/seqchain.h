// seqchain.h — whole nn.GRU / packed nn.LSTM sequences (forward and BPTT) as ONE persistent launch with the workgroup's weight slice
// held in registers (declared here, defined in seqchain.hip; called by rnn.hip).
//
// A recurrent sequence is one link per step: a 16x16 tile of the state needs h_{j} W^T over the full K every step.  As a program of the
// persistent-chain engine (pchain.hip) that link re-reads its weight slice every step — 96 KB per tile for a GRU of R = 512, three
// quarters of what the tile ingests — and pays the engine's walk between tiles.  A sequence, though, is a ONE-link program in which
// every workgroup owns the same tile in every step: here the slice is loaded ONCE into VGPRs (a wave's k-chunks of all gate products:
// 48 registers per lane at R = 512) and the step loop is the tile itself — poll the state's T16 fragments, MFMA against the resident
// registers, LDS reduction, gate math, stores.  Same hand-off protocol, buffers and numerics (summation order) as the engine's tiles.
#pragma once
#include "common.h"
#include "pchain.h"

namespace blvm {

struct SeqGruFwd {
  const float *H16, *Whh, *bhh, *xg;  // T16 state slabs (slab j = state entering step j), T16 weights [3R,R], [3R], [T,B,3R] time indexed
  const int32_t* lens;
  float *Hs, *out, *rg, *ug, *ng, *ghn;  // [T+1,B,R] row-major states; time-indexed outputs; per-step saves [T,B,R]
  long out_ts;
  int out_ld, T, B, R, reverse, bf16;
  pchain::Ctl ctl;
};
struct SeqGruBwd {
  const float *DGH16, *WhhT, *dout, *rg, *ug, *ng, *ghn, *Hs;  // DGH16: T+1 T16 slabs indexed by s (slab s written at step s)
  const int32_t* lens;
  float *G, *DGI, *DGH, *dh0;  // G [B,R] in place; DGI [T,B,3R] time indexed; DGH [T,B,3R] step indexed
  long out_ts;
  int out_ld, T, B, R, reverse, steps, bf16;  // steps = T + 1 when dh0 is wanted
  pchain::Ctl ctl;
};
struct SeqLstmFwd {
  const float *H16, *Whh, *bhh, *xg;  // xg [T,B,4H]
  const int32_t* lens;
  float *Hs, *Cs, *out, *gates;       // [T+1,B,H] x2, [T,B,H], [T,B,4H]
  int T, B, H, bf16;
  pchain::Ctl ctl;
};
struct SeqLstmBwd {
  const float *DG16, *WhhT, *dout, *gates, *Cs;
  float *DC, *DG, *dh0;               // DC [B,H] in place; DG [T,B,4H] step indexed
  int T, B, H, steps, bf16;
  pchain::Ctl ctl;
};

// whether the register-resident kernels take a sequence: K a multiple of 128 (every wave the same number of k-chunks), at most one
// tile per CU, and a chunk count that was instantiated
bool seq_regs_applies(int K_fwd, int K_bwd, int hidden, int B, int gates);
int seq_regs_mask();  // bit 0: forward, bit 1: backward sequences on these kernels
int seq_gru_fwd(const SeqGruFwd& a, hipStream_t s);
int seq_gru_bwd(const SeqGruBwd& a, hipStream_t s);
int seq_lstm_fwd(const SeqLstmFwd& a, hipStream_t s);
int seq_lstm_bwd(const SeqLstmBwd& a, hipStream_t s);

}  // namespace blvm
