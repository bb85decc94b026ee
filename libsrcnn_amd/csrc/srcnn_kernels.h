// srcnn_kernels.h -- internal interface between the C-ABI layer (srcnn_capi.cpp) and the gfx950
// kernels (srcnn_kernels.hip).  Not installed; the public surface is include/srcnn_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace srcnn {

constexpr int C1N = 64;   // layer-1 feature maps   (reference: CONV1_FILTERS, src/convdata.h:5)
constexpr int C2N = 32;   // layer-2 feature maps   (reference: CONV2_FILTERS, src/convdata.h:8)
// Which of the reference's roundings a call gives up (bitmask; 0 = strict = bit-exact).  See DESIGN.md "numerics contract".
enum : int {
    RELAX_L1 = 1,        // layer 1 as an FMA chain on the fp32 MFMA (C = acc): one rounding per tap instead of two
    RELAX_L2 = 2,        // layer 2 likewise
    RELAX_L3_X64 = 4,    // layer 3 with exact products (v_fma_f64 on widened operands); sums as the reference's
    RELAX_L3_F32 = 8,    // layer 3 as fp32 FMA chains (k_conv3_fast)
    RELAX_FAST = RELAX_L1 | RELAX_L2 | RELAX_L3_F32
};
constexpr int kWeightCount = 64 + 64 * 81 + 32 + 32 * 64 + 1 + 32 * 25;   // 8129

// Device-side weight image (one __constant__ instance).  See srcnn_kernels.hip header for the
// re-layout relative to the reference's arrays.
struct DevWeights {
    float w1t[81][C1N];
    float b1[C1N];
    float w2[C2N][C1N];
    float b2[C2N];
    float w3[C2N][25];
    float b3;
    float pad_[3];
    float w3p[C2N * 30];           // w3 in k_conv3's packed-operand order: [m][dy][(w0,w1) (w2,w3) w4 pad]
};

struct DevAxisTable {          // device pointers into one uploaded AxisTable
    const int* first;
    const int* taps;
    const double* weight;
    int stride;
    int max_taps;
    int monotone;              // first[] and first[]+taps[] are non-decreasing (checked on the host when the table is built)
    const int* h_first;        // HOST copies of first[] / taps[] (launch planning: tile spans), or NULL
    const int* h_taps;
};

// Where the Y resampler reads its source samples: a planar float32 plane, or an interleaved 8-bit RGB(A) image whose
// Y = 0.299 R + 0.587 G + 0.114 B is computed per sample exactly as converImgU8toYCbCr does (src/libsrcnn.cpp:233-272),
// so that the colour split never has to materialise a Y plane.
struct YSource {
    const float* plane = nullptr;
    const unsigned char* rgb = nullptr;
    int depth = 0;
    static YSource from_plane(const float* p) { YSource y; y.plane = p; return y; }
    static YSource from_rgb(const unsigned char* p, int d) { YSource y; y.rgb = p; y.depth = d; return y; }
};

// Weight image of the fused fp16 kernel (srcnn_fused_f16.hip): every fp32 weight, pre-scaled by 2^8, split into
// fp16 hi + lo and laid out in MFMA A-fragment order (lane-major, 8 halves per lane per fragment), built once on the
// host.  The struct is byte-for-byte what the kernel copies into LDS.
//   w1[s][blk][hl][lane][j]  80 of the 81 taps packed into FU_NK = 5 k-steps of 16 slots (8 per lane half):
//                              s = 0..3: lanes 0-31 window row 2s, lanes 32-63 window row 2s+1, tap dx = j
//                              s = 4:    lanes 0-31 window row 8, dx = j;  lanes 32-63 window row j, dx = 8
//   w88[half][reg]           the 81st tap (8,8) x 2^8 as plain fp32, in accumulator-register order: it enters as an
//                            fp32 FMA into the C operand that opens the layer-1 chains (a sixth k-step for one tap would
//                            cost 12 MFMAs per 64 pixels; the FMAs ride in the MFMA-only phase's idle VALU slots)
//                            row (lane%32) = channel 32*blk + lane%32
//   w2[blk][ks][hl][lane][j] row = output m = lane%32, k = input channel 32*blk + 16*ks + 8*(j/4) + 4*(lane/32) + j%4
//   w3[ks][hl][lane][j]      row = tap t = lane%32 (dy*5+dx, rows >= 25 zero), k = channel m = 16*ks + 8*(j/4) + 4*(lane/32) + j%4
//   b1[half][reg], b2[half][reg]  biases x 2^8 in accumulator-register order: the C operand that opens an MFMA chain
#ifndef FU_NW_DEF
#define FU_NW_DEF 8
#endif
constexpr int FU_NK = 5;      // layer-1 k-steps of the fused kernel
constexpr int FU_NW = FU_NW_DEF;      // waves per workgroup of the fused kernel
struct FusedF16Weights {
    unsigned short w1[FU_NK][2][2][64][8];
    unsigned short w2[2][2][2][64][8];
    unsigned short w3[2][2][64][8];
    float b1[64];
    float b2[32];
    float w88[64];
    float b3;
    float pad_[3];
};

hipError_t upload_weights(const DevWeights& w);
hipError_t fused_f16_prepare();
void launch_fused_f16(const float* Y, int W, int H, int y_row_base, int y_rows, float* out, int out_row0, int out_rows,
                      const FusedF16Weights* d_blob, int num_cus, hipStream_t s, unsigned long long* dbg = nullptr);

void launch_resample_cols(const float* src, int w, int src_row_base, float* dst, int dst_row0, int dst_rows,
                          const DevAxisTable& t, hipStream_t s);
void launch_resample_rows(const float* src, int src_w, float* dst, int dst_w, int rows, const DevAxisTable& t,
                          hipStream_t s);
// Round-3 resampler (k_rs2d): both passes of an up-scale in one kernel, 4 output columns per thread with 16-byte
// stores, tile spans in closed form from the (monotone) tables.  `src` may be an RGB(A) image (Y computed on the fly).
// Returns false (nothing launched) when the shape does not qualify; the caller then falls back to the older kernels.
bool launch_rs2d(const YSource& src, int src_w, int src_h, float* dst, int dst_w, int dst_h, int dst_row0, int dst_rows,
                 const DevAxisTable& tv, const DevAxisTable& th, hipStream_t s);
hipError_t rs2d_prepare();      // per device: lets k_rs2d ask for more than 64 KB of dynamic LDS
bool rs2d_fits(int np, int src_w, int src_h, int dst_w, int dst_h, int r0, int r1, const DevAxisTable& tv, const DevAxisTable& th);
// Colour merge with the chroma (and alpha) planes resampled on the fly from the SOURCE image (src/libsrcnn.cpp:665-726
// per-plane resample + :274-308 merge, fused): for output rows [dst_row0, +dst_rows) reads the interleaved source
// image and the finished Y' rows (Yp: row dst_row0 at offset 0), writes interleaved u8 (rgb_out: row dst_row0 at offset 0)
// and optionally the truncated Y' (conv_opt).  The destination-size Cb / Cr / A planes never exist.  Same per-sample
// arithmetic and order as the separate kernels.  Returns false when the shape does not qualify.
bool launch_merge_fused(const unsigned char* rgb_src, int src_w, int src_h, int depth, const float* Yp,
                        unsigned char* rgb_out, unsigned char* conv_opt, int dst_w, int dst_h, int dst_row0, int dst_rows,
                        const DevAxisTable& tv, const DevAxisTable& th, hipStream_t s);
hipError_t conv12_mfma_prepare();
// relax: RELAX_L1 | RELAX_L2 bits (0 = strict)
// clk: NULL, or two device words that receive (shader-clock cycles, 100 MHz ticks) of workgroup 0's lifetime
// queue: NULL (tiles dealt with a static stride), or two zeroed device words owned by the launch stream's workspace: the tile
//        queue of the kernel (it leaves them zeroed again)
void launch_conv12_mfma(const float* Y, int W, int H, int y_row_base, int y_rows, float* C2, size_t plane_stride, int out_row0,
                        int out_rows, int relax, int num_cus, hipStream_t s, unsigned long long* clk = nullptr,
                        unsigned* queue = nullptr);
void conv12_grid_info(int num_cus, int* blocks, int* tile_rows);
// the same two numbers at compile time (srcnn_kernels.hip static_asserts them against M_TH / M_BPC): tile height and resident
// workgroups per CU of the layer-1+2 kernel
constexpr int kConv12TileRows = 16, kConv12BlocksPerCU = 2;
// relax: RELAX_L3_X64 / RELAX_L3_F32 bits (neither = strict)
void launch_conv3(const float* C2, size_t plane_stride, int W, int H, int c2_row_base, int c2_rows, float* out,
                  int out_row0, int out_rows, int relax, hipStream_t s);
void launch_conv1_planes(const float* Y, int W, int H, float* C1, hipStream_t s);
void launch_conv2_planes(const float* C1, size_t n, float* C2v, hipStream_t s);
void launch_rgb_split(const unsigned char* rgb, size_t n, int d, float* Yp, float* Cb, float* Cr, float* A,
                      hipStream_t s);
void launch_ycc_merge(const float* Yp, const float* Cb, const float* Cr, const float* A, size_t n, int d,
                      unsigned char* rgb, unsigned char* conv_opt, hipStream_t s);

}  // namespace srcnn
