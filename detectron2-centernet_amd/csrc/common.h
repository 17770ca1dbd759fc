// Shared declarations for the CenterNet hot-path HIP kernels (gfx950 / CDNA4 only).
// Data layout everywhere on the device: activations are NHWC ("pixel rows": one
// pixel = `stride` contiguous channel elements), weights are KRSC packed to
// [Cout_pad][Kpad] (f16 MFMA path) or [Kpad][Cout_pad] (f32 exact path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// error plumbing (thread-local message, returned through ctdet_last_error()).
void ctdet_set_error(const char* fmt, ...);
#define CTDET_CHECK(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ctdet_set_error(__VA_ARGS__);       \
      return -22; /* -EINVAL */           \
    }                                     \
  } while (0)
#define CTDET_LAUNCH_CHECK()                                          \
  do {                                                                \
    hipError_t e_ = hipGetLastError();                                \
    if (e_ != hipSuccess) {                                           \
      ctdet_set_error("kernel launch failed: %s (%s:%d)",            \
                      hipGetErrorString(e_), __FILE__, __LINE__);     \
      return -5; /* -EIO */                                           \
    }                                                                 \
  } while (0)

enum { CTDET_F16 = 0, CTDET_F32 = 1, CTDET_U8 = 2, CTDET_F16X3 = 3 };
enum { CTDET_ACT_NONE = 0, CTDET_ACT_RELU = 1, CTDET_ACT_SIGMOID_CLAMP = 2 };

// Kernel-side argument block for every conv-shaped contraction on the path
// (plain conv, DCNv2 main contraction, offset/mask conv, head convs).
struct ConvArgs {
  const void* x;        // [B,H,W,in_stride] input pixels (channel slice starts at x)
  const void* w;        // packed weights
  const float* scale;   // per-cout multiplier (folded BN gamma/sqrt(var+eps)) or null
  const float* bias;    // per-cout bias (folded BN beta - mean*scale, or conv bias) or null
  const void* res;      // residual added before activation, same dtype as y, or null
  void* y;              // [B,Ho,Wo,out_stride]
  // Root (dla.py:86-94) = 1x1 conv over torch.cat(children): instead of materialising the concat the
  // kernel reads K segments from up to 4 source tensors (nsrc > 1; 1x1, stride 1, pad 0 only).
  const void* xs[4];
  int xs_stride[4];
  int xs_cend[4];       // cumulative channel ends of the sources
  int nsrc;
  const float* om;      // DCNv2 only: [M, om_stride] f32; ch 0..17 offsets (2k=dh,2k+1=dw), 18..26 mask logits
  int om_stride;
  int mask_is_prob;     // DCNv2: 0 = mask channels are logits (sigmoid here), 1 = already probabilities
  int B, H, W, Cin, in_stride;
  int Cout, Ho, Wo, out_stride, res_stride;
  int R, S, stride, pad, dil;
  int K, Kpad, Cout_pad;
  int M;                // B*Ho*Wo
  int act;
  float clamp_lo, clamp_hi;
  int korder;           // 0 tap-major, 1 chunk-major (see ctdet_conv_desc)
  int in_dil;           // input dilation (zero-stuffed input): >1 only for the input-gradient of strided convs
  // DCNv2 with its offset / mask conv computed in the same kernel (ctdet_dcnv2_offset_fwd): that conv's packed f16 weights
  // (32 rows, chunk-major), its bias (32 f32), and optionally where to keep its f32 output for a backward pass
  const void* w_off; const float* b_off; float* om_out; int om_out_stride;
  // DCNv2, f16x3 window kernel, training: the sampled columns (mask * bilinear(x), f32 [M][9*Cin], k = tap*Cin + c -- what
  // modulated_deformable_im2col produces, kernel.cu:786-868) written as a by-product of the forward pass, or null
  float* cols_out;
};

// argument block of the fused CenterNet head kernel (conv_igemm.hip): per head 3x3 conv Cin->256 + bias + ReLU, then
// 1x1 conv 256->cout + bias (+ sigmoid/clamp), the 256-channel hidden map staying in registers
struct HeadArgs {
  const void* x;          // f16 [B,H,W,in_stride]
  const void* w1;         // f16 packed chunk-major [nheads*256][9*Cin]
  const float* b1;        // [nheads*256]
  const void* w2[4];      // f16 [round_up(cout,16)][256]
  const float* b2[4];     // f32 [round_up(cout,16)]
  float* y[4];            // f32 [B,H,W,y_stride]
  int y_stride[4], cout[4], act[4];
  int nheads, B, H, W, Cin, in_stride;
  float clamp_lo, clamp_hi;
};

// argument block of the fused DLA base kernel (dla_base.hip): normalisation + 7x7 stem + level0 + level1
struct BaseArgs {
  const void* img;        // [B,3,H,W] uint8 or f32 planar image batch
  int img_dtype;          // CTDET_U8 / CTDET_F32
  long img_batch_stride;  // elements between images
  int B, H, W;            // image size
  int Hp, Wp;             // network input size (image zero-padded bottom/right after normalisation)
  float mean[3], stdv[3];
  const void* w0;         // f16 [16][7*8*4]   stem, k = (r*8 + s)*4 + c (s = 7 and c = 3 zero)
  const void* w1;         // f16 [16][160]     level0, k = (r*3 + s)*16 + c
  const void* w2;         // f16 [32][160]     level1 (stride 2)
  const float *s0, *b0, *s1, *b1, *s2, *b2;   // folded BatchNorm scale / bias per layer
  void* y;                // f16 [B,Hp/2,Wp/2,out_stride]
  int out_stride;
  void* pool;             // optional f16 [B,Hp/4,Wp/4,pool_stride]: 2x2 max-pool of y
  int pool_stride;
};
int launch_dla_base(const BaseArgs& a, hipStream_t s);
int launch_dla_base_x3(const BaseArgs& a, hipStream_t s);   // f16x3: w* = ctdet_pack_weights_x3 layout 0 images, y / pool f32

// argument block of the batched decode (decode.hip)
struct DecArgs {
  const float* heat; const float* wh; const float* reg;
  int heat_stride, wh_stride, reg_stride;   // pixel strides (elements)
  int B, H, W, C, K;
  float down_ratio;
  uint32_t floor_bits;                      // bits of the promised lower bound of the heat values (0 = none)
  uint32_t* ws;
  float* boxes; float* scores; int* classes; int* inds;
};

__device__ __forceinline__ float ctdet_sigmoid(float v) { return 1.0f / (1.0f + __expf(-v)); }
__device__ __forceinline__ float ctdet_sigmoid_exact(float v) { return 1.0f / (1.0f + expf(-v)); }

// Kernel-selection switches (ctdet_set_tuning_flags in the C ABI; a process-wide word read with one relaxed load per launch
// -- no getenv on the launch path).  0 = every specialised kernel enabled.
enum {
  CTDET_TUNE_NO_HALO = 1,            // 3x3 halo-resident conv -> uniform-K kernel
  CTDET_TUNE_NO_WIN = 2,             // LDS-window kernels of the narrow DLA base layers -> small-channel kernel
  CTDET_TUNE_DCN_MIXED = 4,          // DCNv2 window kernel: per-lane instead of per-wave out-of-window gathers
  CTDET_TUNE_NO_WGRAD_WINDOW = 8,    // weight-gradient window kernel -> generic kernel
  CTDET_TUNE_NO_COL2IM_WINDOW = 16,  // DCNv2 backward LDS-window scatter -> global-atomic kernel
  CTDET_TUNE_NO_F32_DCN_WINDOW = 32, // f32 DCNv2 LDS-window kernel -> global-gather kernel
  CTDET_TUNE_NO_SMALL_GRID_TILES = 128, // convs whose 128-cout grid underfills the chip keep 128-cout tiles (default: 64-cout tiles)
  CTDET_TUNE_NO_HALO_TAP2 = 512,     // f16 3x3 halo conv with Cin % 64 == 0: the per-tap kernel instead of conv3x3_halo_tap2_kernel
  CTDET_TUNE_DCN_SPLIT_4W = 256,     // f16x3 DCNv2 window kernel: 64-cout tiles (four 32-pixel waves) also for the 128- and 256-cout layers
  CTDET_TUNE_PAIR2_128 = 1024,       // f16x3 3x3 halo pair kernel: 128-cout tiles, one workgroup per CU (512 registers per wave)
  CTDET_TUNE_DCN_SPLIT_8W64 = 2048,  // f16x3 DCNv2 window kernel, 64-cout layers: eight 16-pixel waves per workgroup, four waves per SIMD
  CTDET_TUNE_TARGETS_MEMSET = 4096,   // gaussian targets: clear the heat map with hipMemsetAsync (round 3's form: a memset NODE in a captured step;
                                      // kept to reproduce profiles/r04_graph_memset_node.txt and to test the node check)
  CTDET_TUNE_DCN_WINDOW_V1 = 64,     // 64-cout f16 DCNv2: the per-tap-barrier window kernel instead of the row-step one
};
unsigned ctdet_tuning_flags();
// number of CUs of the CURRENT device (cached per device ordinal, not per process)
int ctdet_device_cu_count();

// launchers implemented in the .hip files (return 0 or negative errno)
int launch_conv_f16(const ConvArgs& a, int out_dtype, bool deform, hipStream_t s);
int launch_head_fused(const HeadArgs& a, hipStream_t s);
int launch_conv_f32(const ConvArgs& a, bool deform, bool split, hipStream_t s);
