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
