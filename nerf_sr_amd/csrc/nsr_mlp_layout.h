// Weight-stream layout shared by the pack kernel and the fused MLP kernels.
//
// The MLP is evaluated TRANSPOSED: H_out^T (features x points) = W (features x K)
// * H_in^T (K x points).  The weights are the MFMA A operand, the activations the
// B operand, and the output fragment (C/D) of one layer is — element for element,
// with no data movement — the B operand of the next layer:
//
//   v_mfma_f32_32x32x2_f32   A[i = lane&31][k = lane>>5]      (1 VGPR)
//                            B[k = lane>>5][j = lane&31]      (1 VGPR)
//                            D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31], r = 0..15
//
// A wave owns 32 points (columns).  After a layer, lane (m, h) holds, for point m,
// the features  f(t, h) = 32*(t>>4) + ((t&3) + 8*((t&15)>>2)) + 4*h,  t = 0..127
// in register t.  The next layer's k-step t therefore contracts over the feature
// pair { f(t,0), f(t,1) }, and the packed A fragment of that step holds exactly
// those two weight columns.  Positional-encoding inputs get an analogous fixed
// register -> column map (pecol / dircol below).
//
// Stream = sequence of 1 KiB "pieces" (64 lanes x float4).  Piece (segment, slab
// c, block nb): lane l, component j holds  W[n = 32*nb + (l&31)][ col(4c+j, l>>5) ].
// Pieces are consumed in stream order in 32 KiB chunks through a 2-deep LDS ring.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NSR_HD __host__ __device__ __forceinline__
#else
#define NSR_HD inline
#endif

namespace nsr {

constexpr int kPosCh = 63, kDirCh = 27, kInCh = 90, kWidth = 256;
constexpr int kPieceBytes = 1024;
constexpr int kChunkPieces = 32;                        // 32 KiB chunks
constexpr int kChunkBytes = kChunkPieces * kPieceBytes;

enum Src { SRC_PE = 0, SRC_ACT = 1, SRC_DIR = 2 };

struct Segment {
  int tensor;   // index of the weight tensor in state_dict order (0,2,4,..: weights)
  int nb;       // 32-row output blocks
  int src;      // what the B operand registers hold
  int steps;    // k-steps of 2
  int col0;     // column offset of this segment inside the weight matrix
  int piece0;   // first piece of the segment in the stream
};

// state_dict order: [2*i] = weight, [2*i+1] = bias for i = 0..7 (trunk), 8 = final,
// 9 = dir_encoding, 10 = sigma, 11 = rgb.
constexpr int kNumSegments = 12;
#define NSR_SEGMENT_TABLE \
  {0, 8, SRC_PE, 32, 0, 0}, \
  {2, 8, SRC_ACT, 128, 0, 64}, \
  {4, 8, SRC_ACT, 128, 0, 320}, \
  {6, 8, SRC_ACT, 128, 0, 576}, \
  {8, 8, SRC_PE, 32, 0, 832}, \
  {8, 8, SRC_ACT, 128, 63, 896}, \
  {10, 8, SRC_ACT, 128, 0, 1152}, \
  {12, 8, SRC_ACT, 128, 0, 1408}, \
  {14, 8, SRC_ACT, 128, 0, 1664}, \
  {16, 8, SRC_ACT, 128, 0, 1920}, \
  {18, 4, SRC_ACT, 128, 0, 2176}, \
  {18, 4, SRC_DIR, 16, 256, 2304}, \

// L1 | L2 L3 L4 | L5 skip part (cols 0..62 = pe), L5 trunk part (cols 63..318 = h) | L6 L7 L8 |
// xyz_encoding_final | dir_encoding feature part (cols 0..255), view-dir part (cols 256..282)
constexpr Segment kSegments[kNumSegments] = {NSR_SEGMENT_TABLE};
#if defined(__HIPCC__)
__device__ const Segment kSegmentsDev[kNumSegments] = {NSR_SEGMENT_TABLE};
#endif
constexpr int kStreamPieces = 2320;
constexpr int kStreamPiecesPadded = 2336;               // whole chunks
constexpr int kStreamFloats = kStreamPiecesPadded * 256;

// aux block (plain fp32 copies, appended to the stream)
constexpr int kAuxBias0 = 0;                 // 8 x 256 trunk biases
constexpr int kAuxBiasFinal = 8 * 256;       // 256
constexpr int kAuxBiasDir = 9 * 256;         // 128
constexpr int kAuxSigmaW = 9 * 256 + 128;    // 256
constexpr int kAuxRgbW = kAuxSigmaW + 256;   // 3 x 128
constexpr int kAuxSigmaB = kAuxRgbW + 384;   // 1
constexpr int kAuxRgbB = kAuxSigmaB + 1;     // 3
constexpr int kAuxFloats = ((kAuxRgbB + 3 + 63) / 64) * 64;

constexpr int kPad = -1;

// feature held in activation register t of lane-half h
NSR_HD int act_feature(int t, int h) { return 32 * (t >> 4) + (t & 3) + 8 * ((t & 15) >> 2) + 4 * h; }
// positional-encoding column (of the 63) held in PE register t (0..31) of lane-half h
//   h=0: [x, y, freq 0..4]   h=1: [z, pad, freq 5..9]
NSR_HD int pecol(int t, int h) {
  if (t == 0) return h == 0 ? 0 : 2;
  if (t == 1) return h == 0 ? 1 : kPad;
  return 3 + 30 * h + (t - 2);
}
// view-direction encoding column (of the 27) held in DIR register t (0..15) of lane-half h
//   h=0: [dx, dy, freq 0..1, pad, pad]   h=1: [dz, pad, freq 2..3, pad, pad]
NSR_HD int dircol(int t, int h) {
  if (t == 0) return h == 0 ? 0 : 2;
  if (t == 1) return h == 0 ? 1 : kPad;
  if (t >= 14) return kPad;
  return 3 + 12 * h + (t - 2);
}
NSR_HD int seg_col(int src, int t, int h) {
  return src == SRC_PE ? pecol(t, h) : (src == SRC_ACT ? act_feature(t, h) : dircol(t, h));
}

}  // namespace nsr

// ===========================================================================
// Split-fp16 ("f16x3") stream.  v_mfma_f32_32x32x16_f16: A / B fragments are 8
// halves per lane, lane (i, h) holds k = 8h..8h+7 of a 16-deep k-step.  The same
// register algebra as above holds with k-step s covering registers t = 8s..8s+7:
// the D fragment of an output block nb becomes k-steps 2nb, 2nb+1 of the next layer.
// Every fp32 weight w is stored as hi = RN_f16(w), lo = RN_f16(w - hi); the kernel
// forms  a_hi*b_hi + (a_hi*b_lo + a_lo*b_hi)  with fp32 accumulation (products
// exact to ~2^-21 relative).
//
// The stream is consumed block-row by block-row ("chunk" = all k-steps of one
// 32-feature output block, so the block's result can be re-split into the next
// layer's operands while the following block is on the matrix pipe):
//   chunk = [ A_hi(s=0), A_lo(s=0), A_hi(1), A_lo(1), ..., bias piece ]
// L1 (K=64) packs four output blocks per chunk.
// ===========================================================================
namespace nsr {
namespace hx {

constexpr int kChunks = 71;          // 2 (L1) + 24 (L2-4) + 8 (L5) + 24 (L6-8) + 8 (final) + 1 (sigma) + 4 (dir)
constexpr int kChunkFinal0 = 58;     // first chunk of xyz_encoding_final (skipped by sigma_only launches)
constexpr int kChunkSigma = 66;      // density head: sigma.weight as row 0 of a 32-row block over h8
constexpr int kSlotPieces = 41;      // largest chunk: L5, 20 k-steps * 2 + bias
constexpr int kSlotFloats = kSlotPieces * 256;

struct Chunk {
  int tensor;   // weight tensor (state_dict index); bias = tensor + 1
  int nb0;      // first output block
  int nnb;      // output blocks in the chunk (4 for L1, else 1)
  int steps;    // 16-deep k-steps per output block
  int piece0;   // first piece of the chunk in the stream
};

NSR_HD int chunk_pieces(int steps, int nnb) { return 2 * steps * nnb + 1; }

NSR_HD Chunk chunk_info(int q) {
  Chunk c{};
  if (q < 2) {                       // L1: 63(+1) -> 256, 4 k-steps
    c.tensor = 0; c.nb0 = 4 * q; c.nnb = 4; c.steps = 4; c.piece0 = 33 * q;
  } else if (q < 26) {               // L2..L4
    const int l = (q - 2) >> 3;
    c.tensor = 2 * (l + 1); c.nb0 = (q - 2) & 7; c.nnb = 1; c.steps = 16; c.piece0 = 66 + 33 * (q - 2);
  } else if (q < 34) {               // L5: cat([pe, h]) -> 4 + 16 k-steps
    c.tensor = 8; c.nb0 = q - 26; c.nnb = 1; c.steps = 20; c.piece0 = 858 + 41 * (q - 26);
  } else if (q < 66) {               // L6..L8, xyz_encoding_final
    const int l = (q - 34) >> 3;
    c.tensor = 2 * (l + 5); c.nb0 = (q - 34) & 7; c.nnb = 1; c.steps = 16; c.piece0 = 1186 + 33 * (q - 34);
  } else if (q == 66) {              // sigma head on h8 (same input registers as xyz_encoding_final)
    c.tensor = 20; c.nb0 = 0; c.nnb = 1; c.steps = 16; c.piece0 = 2242;
  } else {                           // dir_encoding: cat([g, de]) -> 16 + 2 k-steps, 4 output blocks
    c.tensor = 18; c.nb0 = q - 67; c.nnb = 1; c.steps = 18; c.piece0 = 2275 + 37 * (q - 67);
  }
  return c;
}
constexpr int kPiecesTotal = 2275 + 37 * 4;             // 2423 pieces of 1 KiB
// chunk that holds stream piece `piece` (the inverse of chunk_info(q).piece0; the pack kernel runs it once per word --
// round 5: as a linear search over the 71 chunks it was most of that kernel's 25 us, twice per training step)
NSR_HD int chunk_of_piece(int piece) {
  if (piece < 66) return piece / 33;
  if (piece < 858) return 2 + (piece - 66) / 33;
  if (piece < 1186) return 26 + (piece - 858) / 41;
  if (piece < 2242) return 34 + (piece - 1186) / 33;
  if (piece < 2275) return 66;
  return 67 + (piece - 2275) / 37;
}
// aux (fp32): rgb_w 384 | rgb_b 3
constexpr int kAuxRgbW = 0, kAuxRgbB = 384, kAuxFloats = 448;

// weight column (or kPad) that register t of lane-half h multiplies in k-step space of `tensor`
NSR_HD int column_of(int tensor, int s, int j, int h) {
  const int t = 8 * s + j;
  if (tensor == 0) return pecol(t, h);
  if (tensor == 8) {
    if (s < 4) return pecol(t, h);
    return kPosCh + act_feature(t - 32, h);
  }
  if (tensor == 18) {
    if (s < 16) return act_feature(t, h);
    const int c = dircol(t - 128, h);
    return c == kPad ? kPad : kWidth + c;
  }
  return act_feature(t, h);
}

}  // namespace hx
}  // namespace nsr
