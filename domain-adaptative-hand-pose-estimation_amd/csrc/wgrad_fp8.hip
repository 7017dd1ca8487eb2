// Weight gradient of the 3x3 / stride-1 / pad-1 convs from the fp8 copies of their operands ('fp8' compute mode, BASELINE
// config 5): dW[o][kh][kw][c] = descale_x * descale_dy * sum_m DY8[m][o] * X8[m + (kh-1) W + (kw-1)][c].
//
// The construction of wgrad_kw_kernel (igemm.hip) on one-byte elements and the K=64 fp8 MFMA:
//  * a block owns (128|64 output channels) x (64 input channels) x (kernel row kh; kw = 0, 1, 2); per 64-pixel step it stages
//    one DY tile and ONE X tile; the three taps of the kernel row are the same X rows read shifted by -1 / 0 / +1.  The LDS X
//    image keeps every run of min(W, 64) pixels in a segment of its own with four spare (zero) rows between segments, so a
//    shifted read sees zeros exactly where the padding is (W > 64: the two neighbouring pixels are fetched as halo rows);
//  * the reduction runs over PIXELS, which are the rows of both tiles, so the MFMA operands are read transposed:
//    ds_read_b64_tr_b8 hands result lane 16g + 8h + j the byte j of source lanes 16g + 2k + h, k = 0..7
//    (profiles/tr8_probe.hip).  Source lane (g, k, h) therefore points at pixel 32 (g >> 1) + 8 q + k and at the 8 channels
//    8 (2 (g & 1) + h) .. + 8 of the 32-channel block: result lane l then holds channel l & 31 for the 8 pixels of read q in
//    its half (l >> 5) of the 64-pixel step -- four reads give the 32 bytes of one v_mfma_f32_32x32x64_f8f6f4 operand.  Both
//    operands use the same pixel <-> (half, byte) assignment, which is all the dot product needs;
//  * LDS rows are 128 bytes (DY, 128 channels) or 64 bytes (DY of 64 channels; X): the 32-byte slot a transposing read
//    takes from each of its 8 rows is XOR-swizzled with row bits so that the 8 rows of a half-wave fall on 8 different
//    32-byte bank groups, for every row shift.
#include "igemm_common.h"
#include "fp8_common.h"
#include <stdlib.h>

typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef int i32x2_t __attribute__((ext_vector_type(2)));

struct WgradKw8Args {
  const void* X; const void* DY; float* out;
  const float* descale_x; const float* descale_dy;
  int H, W, Ci, Co;
  int lw, lwf, halo;            // log2(min(W,64)), log2(W), W > 64
  int M, rows_per_split, ldw;
  long slab_stride;
  int nto, nci;
  unsigned x_bytes, dy_bytes;
  FastDiv dH;
};

// byte offset of channel byte `cb` in row `r`: 32-byte slots swizzled with row bits (see the header)
__device__ __forceinline__ int off128(int r, int cb) { return r * 128 + ((((cb >> 5) ^ ((r >> 1) & 3)) << 5) | (cb & 31)); }
__device__ __forceinline__ int off64(int r, int cb) { return r * 64 + ((((cb >> 5) ^ ((r >> 2) & 1)) << 5) | (cb & 31)); }

template <int MT, bool X_BF8, bool DY_BF8>
__global__ __launch_bounds__(256, 3) void wgrad_kw8_kernel(const WgradKw8Args p) {
  constexpr int BO = 64 * MT, BKM = 64;
  constexpr int CPRY = BO / 16, RPY = 256 / CPRY, NPY = BKM / RPY;      // DY staging: 16-byte chunks per row, rows per pass, passes
  constexpr int XROWS = 64 + 4 * 8 + 4;
  __shared__ __attribute__((aligned(16))) char smem[BKM * BO + XROWS * 64 + 2 * BKM * 4];
  char* ys = smem;
  char* xs = smem + BKM * BO;
  int* rowinfo = reinterpret_cast<int*>(smem + BKM * BO + XROWS * 64);   // [2][BKM]: image row oy of each pixel

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ntile = p.nto * 3 * p.nci;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lid / ntile;
  int tile = lid - split * ntile;
  const int ot = tile / (3 * p.nci); tile -= ot * 3 * p.nci;
  const int kh = tile / p.nci, cit = tile - kh * p.nci;
  const int o0 = ot * BO, ci0 = cit * 64, tdy = kh - 1;
  const int wm0 = (wave >> 1) * (32 * MT), wn0 = (wave & 1) * 32;

  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X, p.x_bytes), rsY = make_rsrc(p.DY, p.dy_bytes);
  const int lcy = t % CPRY, lry = t / CPRY;
  const int lcx = t & 3, lrx = t >> 2;
  const bool ook = (o0 + lcy * 16) < p.Co;
  const bool cok = (ci0 + lcx * 16) < p.Ci;
  const int mbeg = split * p.rows_per_split;
  const int mend = min(p.M, mbeg + p.rows_per_split);

  for (int i = t; i < XROWS * 4; i += 256) reinterpret_cast<uint4*>(xs)[i] = make_uint4(0, 0, 0, 0);

  uint4 ry[NPY], rx, rh = make_uint4(0, 0, 0, 0);
  auto decode_rows = [&](int mt0, int buf) {
    if (t < BKM) {
      const unsigned m = (unsigned)(mt0 + t);
      const unsigned r = m >> p.lwf;
      const unsigned n = fd_div(r, p.dH);
      rowinfo[buf * BKM + t] = (m < (unsigned)p.M) ? (int)(r - n * p.H) : -(1 << 20);
    }
  };
  auto load_tile = [&](int mt0, int buf) {
#pragma unroll
    for (int i = 0; i < NPY; ++i) {
      const int m = mt0 + lry + RPY * i;
      ry[i] = buf_load16(rsY, (m < mend && ook) ? m * p.Co + o0 + lcy * 16 : OOB_OFF);
    }
    {
      const int m = mt0 + lrx;
      const int oy = rowinfo[buf * BKM + lrx];
      const bool ok = cok && m < mend && (unsigned)(oy + tdy) < (unsigned)p.H;
      rx = buf_load16(rsX, ok ? (m + tdy * p.W) * p.Ci + ci0 + lcx * 16 : OOB_OFF);
    }
    if (p.halo && t < 8) {      // W > 64: the tile is a 64-pixel piece of one image row; fetch its two neighbours
      const int side = t >> 2, ox0 = mt0 & (p.W - 1);
      const int oy = rowinfo[buf * BKM];
      const bool ok = (side ? (ox0 + 64 < p.W) : (ox0 > 0)) && (unsigned)(oy + tdy) < (unsigned)p.H && (ci0 + (t & 3) * 16) < p.Ci;
      const int m = mt0 + (side ? 64 : -1);
      rh = buf_load16(rsX, ok ? (m + tdy * p.W) * p.Ci + ci0 + (t & 3) * 16 : OOB_OFF);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NPY; ++i) {
      const int r = lry + RPY * i;
      *reinterpret_cast<uint4*>(ys + (MT == 2 ? off128(r, lcy * 16) : off64(r, lcy * 16))) = ry[i];
    }
    *reinterpret_cast<uint4*>(xs + off64(lrx + 2 + 4 * (lrx >> p.lw), lcx * 16)) = rx;
    if (p.halo && t < 8) *reinterpret_cast<uint4*>(xs + off64((t >> 2) ? 66 : 1, (t & 3) * 16)) = rh;
  };

  f32x16_t acc[3][MT];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][i][r] = 0.f;

  // this lane as a SOURCE of the transposing reads: pixel 32 (g >> 1) + 8 q + k, channels cs .. cs + 8 of the 32-channel block
  const int sg = lane >> 4, sk = (lane & 15) >> 1, sh = lane & 1;
  const int spix = 32 * (sg >> 1) + sk, cs = 8 * (2 * (sg & 1) + sh);
  int offA[MT];                                             // read q adds 8 q rows: the swizzle bits come from k alone
#pragma unroll
  for (int i = 0; i < MT; ++i) offA[i] = MT == 2 ? off128(spix, wm0 + i * 32 + cs) : off64(spix, wm0 + i * 32 + cs);
  int rowB[4];                                              // X image row of this lane's pixel in read q (before the tap shift)
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int px = spix + 8 * q; rowB[q] = px + 2 + 4 * (px >> p.lw); }
  const int cbB = wn0 + cs;

  decode_rows(mbeg, 0);
  __syncthreads();
  if (mbeg < mend) load_tile(mbeg, 0);
  int buf = 0;
  typedef __attribute__((address_space(3))) i32x2_t* lds2;
  for (int mt0 = mbeg; mt0 < mend; mt0 += BKM, buf ^= 1) {
    __syncthreads();
    store_tile();
    decode_rows(mt0 + BKM, buf ^ 1);
    __syncthreads();
    if (mt0 + BKM < mend) load_tile(mt0 + BKM, buf ^ 1);
    __builtin_amdgcn_s_setprio(1);
    i32x8_t a[MT], b[3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const i32x2_t v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2)(ys + offA[i] + q * 8 * BO));
        a[i][2 * q] = v.x; a[i][2 * q + 1] = v.y;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const i32x2_t v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2)(xs + off64(rowB[q] + k - 1, cbB)));
        b[k][2 * q] = v.x; b[k][2 * q + 1] = v.y;
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int i = 0; i < MT; ++i)   // cbsz / blgp: formats of the first (dy) and second (x) operand; 0 = e4m3, 1 = e5m2
        acc[k][i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[k], acc[k][i], DY_BF8 ? 1 : 0, X_BF8 ? 1 : 0, 0, 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  }
  const float scale = (p.descale_x ? *p.descale_x : 1.f) * (p.descale_dy ? *p.descale_dy : 1.f);
  const int r31 = lane & 31, hi = lane >> 5;
  float* out = p.out + (size_t)split * p.slab_stride;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int c = ci0 + wn0 + r31;
        if (o < p.Co && c < p.Ci) out[(size_t)o * p.ldw + (kh * 3 + k) * p.Ci + c] = acc[k][i][r] * scale;
      }
}

// ------------------------------------------------------------------------------------ 3x3 / 4x4, stride 2, pad 1
// The parity-image construction of wgrad_kw2_kernel (igemm.hip) on the same transposing fp8 reads: tap kw of output pixel ox
// reads input column 2 ox + kw - 1, so the taps fall on the two column-parity images of the input row -- E[j] = x[2j],
// O[j] = x[2j+1] -- as shifted reads: kw0 = O[ox-1], kw1 = E[ox], kw2 = O[ox], kw3 = E[ox+1].  Per 64-output-pixel step a
// block stages one DY tile and the two parity tiles and multiplies 3 (4) taps out of them.  Serves the strided 3x3 convs and,
// with the operand roles of the conv-form (x = the transposed conv's output gradient, e5m2; dy = its input, e4m3), the 4x4
// transposed convs.
struct WgradKw28Args {
  const void* X; const void* DY; float* out;
  const float* descale_x; const float* descale_dy;
  int H, W, Ho, Ci, Co;
  int lw, lwo;                  // log2(min(Wo,64)), log2(Wo)
  int M, rows_per_split, ldw;
  long slab_stride;
  int nto, nci;
  unsigned x_bytes, dy_bytes;
  FastDiv dHo;
};

template <int MT, int KW, bool X_BF8, bool DY_BF8>
__global__ __launch_bounds__(256, KW == 4 ? 2 : 3) void wgrad_kw28_kernel(const WgradKw28Args p) {
  constexpr int BO = 64 * MT, BKM = 64;
  constexpr int CPRY = BO / 16, RPY = 256 / CPRY, NPY = BKM / RPY;
  constexpr int XROWS = 64 + 4 * 8 + 4;
  __shared__ __attribute__((aligned(16))) char smem[BKM * BO + 2 * XROWS * 64 + 2 * BKM * 8];
  char* ys = smem;
  char* xe = smem + BKM * BO;                  // even input columns
  char* xo = xe + XROWS * 64;                  // odd input columns
  int2* rowinfo = reinterpret_cast<int2*>(smem + BKM * BO + 2 * XROWS * 64);   // [2][BKM]: {input pixel of (2oy, 2ox), oy}

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ntile = p.nto * KW * p.nci;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lid / ntile;
  int tile = lid - split * ntile;
  const int ot = tile / (KW * p.nci); tile -= ot * KW * p.nci;
  const int kh = tile / p.nci, cit = tile - kh * p.nci;
  const int o0 = ot * BO, ci0 = cit * 64, tdy = kh - 1;
  const int wm0 = (wave >> 1) * (32 * MT), wn0 = (wave & 1) * 32;

  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X, p.x_bytes), rsY = make_rsrc(p.DY, p.dy_bytes);
  const int lcy = t % CPRY, lry = t / CPRY;
  const int lcx = t & 3, lrx = t >> 2;
  const bool ook = (o0 + lcy * 16) < p.Co;
  const bool cok = (ci0 + lcx * 16) < p.Ci;
  const int mbeg = split * p.rows_per_split;
  const int mend = min(p.M, mbeg + p.rows_per_split);

  for (int i = t; i < 2 * XROWS * 4; i += 256) reinterpret_cast<uint4*>(xe)[i] = make_uint4(0, 0, 0, 0);

  uint4 ry[NPY], re, ro;
  auto decode_rows = [&](int mt0, int buf) {
    if (t < BKM) {
      const unsigned m = (unsigned)(mt0 + t);
      const unsigned q = m >> p.lwo, ox = m & ((1u << p.lwo) - 1);
      const unsigned n = fd_div(q, p.dHo);
      const int oy = (int)(q - n * p.Ho);
      rowinfo[buf * BKM + t] = (m < (unsigned)p.M) ? make_int2(((int)n * p.H + 2 * oy) * p.W + 2 * (int)ox, oy) : make_int2(0, -(1 << 20));
    }
  };
  auto load_tile = [&](int mt0, int buf) {
#pragma unroll
    for (int i = 0; i < NPY; ++i) {
      const int m = mt0 + lry + RPY * i;
      ry[i] = buf_load16(rsY, (m < mend && ook) ? m * p.Co + o0 + lcy * 16 : OOB_OFF);
    }
    const int2 ri = rowinfo[buf * BKM + lrx];
    const bool ok = cok && (mt0 + lrx) < mend && (unsigned)(2 * ri.y + tdy) < (unsigned)p.H;
    const int off = (ri.x + tdy * p.W) * p.Ci + ci0 + lcx * 16;
    re = buf_load16(rsX, ok ? off : OOB_OFF);
    ro = buf_load16(rsX, ok ? off + p.Ci : OOB_OFF);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NPY; ++i) {
      const int r = lry + RPY * i;
      *reinterpret_cast<uint4*>(ys + (MT == 2 ? off128(r, lcy * 16) : off64(r, lcy * 16))) = ry[i];
    }
    const int a = off64(lrx + 2 + 4 * (lrx >> p.lw), lcx * 16);
    *reinterpret_cast<uint4*>(xe + a) = re;
    *reinterpret_cast<uint4*>(xo + a) = ro;
  };

  f32x16_t acc[KW][MT];
#pragma unroll
  for (int a = 0; a < KW; ++a)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][i][r] = 0.f;

  const int sg = lane >> 4, sk = (lane & 15) >> 1, sh = lane & 1;           // this lane as a source of the transposing reads
  const int spix = 32 * (sg >> 1) + sk, cs = 8 * (2 * (sg & 1) + sh);
  int offA[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) offA[i] = MT == 2 ? off128(spix, wm0 + i * 32 + cs) : off64(spix, wm0 + i * 32 + cs);
  int rowB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int px = spix + 8 * q; rowB[q] = px + 2 + 4 * (px >> p.lw); }
  const int cbB = wn0 + cs;

  decode_rows(mbeg, 0);
  __syncthreads();
  if (mbeg < mend) load_tile(mbeg, 0);
  int buf = 0;
  typedef __attribute__((address_space(3))) i32x2_t* lds2;
  for (int mt0 = mbeg; mt0 < mend; mt0 += BKM, buf ^= 1) {
    __syncthreads();
    store_tile();
    decode_rows(mt0 + BKM, buf ^ 1);
    __syncthreads();
    if (mt0 + BKM < mend) load_tile(mt0 + BKM, buf ^ 1);
    __builtin_amdgcn_s_setprio(1);
    i32x8_t a[MT], b[KW];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const i32x2_t v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2)(ys + offA[i] + q * 8 * BO));
        a[i][2 * q] = v.x; a[i][2 * q + 1] = v.y;
      }
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        // kw0 = O[ox-1], kw1 = E[ox], kw2 = O[ox], kw3 = E[ox+1]
        const int shift = k == 0 ? -1 : (k == 3 ? 1 : 0);
        const i32x2_t v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2)(((k & 1) ? xe : xo) + off64(rowB[q] + shift, cbB)));
        b[k][2 * q] = v.x; b[k][2 * q + 1] = v.y;
      }
    }
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
      for (int i = 0; i < MT; ++i)
        acc[k][i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[k], acc[k][i], DY_BF8 ? 1 : 0, X_BF8 ? 1 : 0, 0, 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  }
  const float scale = (p.descale_x ? *p.descale_x : 1.f) * (p.descale_dy ? *p.descale_dy : 1.f);
  const int r31 = lane & 31, hi = lane >> 5;
  float* out = p.out + (size_t)split * p.slab_stride;
#pragma unroll
  for (int k = 0; k < KW; ++k)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int c = ci0 + wn0 + r31;
        if (o < p.Co && c < p.Ci) out[(size_t)o * p.ldw + (kh * KW + k) * p.Ci + c] = acc[k][i][r] * scale;
      }
}

// ------------------------------------------------------------------------------------ host side
static int ilog2e(int v) { int s = 0; while ((1 << s) < v) ++s; return ((1 << s) == v) ? s : -1; }
struct Wg8Plan { int S, rows_per_split, nto, nti, ldw, mt, s2; };
static const int g_wg8_blocks = getenv("MI355_WG_BLOCKS") ? atoi(getenv("MI355_WG_BLOCKS")) : 768;

static int wg8_check(const mi355_conv_desc* d) {
  if (!d) MI_FAIL(MI355_EINVAL, "null conv desc");
  if (d->dtype != MI355_FP8) MI_FAIL(MI355_EINVAL, "wgrad_fp8: the descriptor must be an fp8 one");
  const bool s1 = d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->Ho == d->Hi && d->Wo == d->Wi &&
                  d->Wi >= 8 && ilog2e(d->Wi) >= 0;
  const bool s2 = d->kh == d->kw && (d->kh == 3 || d->kh == 4) && d->stride == 2 && d->pad == 1 && d->Hi % 2 == 0 && d->Wi % 2 == 0 &&
                  d->Ho == d->Hi / 2 && d->Wo == d->Wi / 2 && d->Wo >= 8 && d->Wo <= 64 && ilog2e(d->Wo) >= 0;
  if (!s1 && !s2)
    MI_FAIL(MI355_EINVAL, "wgrad_fp8: 3x3 / stride 1 / pad 1 with a power-of-two width >= 8, or 3x3 / 4x4 / stride 2 / pad 1 with even "
            "extents and a power-of-two output width in [8, 64] (k%dx%d s%d p%d, %dx%d -> %dx%d)", d->kh, d->kw, d->stride, d->pad,
            d->Hi, d->Wi, d->Ho, d->Wo);
  if (d->Ci % 16 || d->Co % 16) MI_FAIL(MI355_EINVAL, "wgrad_fp8: channels (%d, %d) must be multiples of 16", d->Ci, d->Co);
  if ((long)d->N * d->Hi * d->Wi * d->Ci >= (1L << 31) || (long)d->N * d->Ho * d->Wo * d->Co >= (1L << 31))
    MI_FAIL(MI355_EINVAL, "wgrad_fp8: tensor too large for 32-bit byte offsets: split the batch");
  return MI355_OK;
}
static Wg8Plan wg8_plan(const mi355_conv_desc* d) {      // the split rule of plan_wgrad (igemm.hip) for the kw-shared tilings
  Wg8Plan w; w.ldw = d->kh * d->kw * d->Ci; w.mt = d->Co <= 64 ? 1 : 2; w.s2 = d->stride == 2;
  w.nto = cdiv(d->Co, 64 * w.mt); w.nti = cdiv(d->Ci, 64);
  const long tiles = (long)w.nto * d->kh * w.nti, M = (long)d->N * d->Ho * d->Wo, ksteps = (M + 63) / 64;
  long S = (g_wg8_blocks + tiles - 1) / tiles;
  const long S16 = ksteps / 16, S256 = (256 + tiles - 1) / tiles, lo = S16 > S256 ? S16 : S256;
  if (S > lo) S = lo;
  if (tiles >= 384) S = 1;
  if (S > ksteps) S = ksteps;
  if (S < 1) S = 1;
  long rps = (M + S - 1) / S; rps = ((rps + 63) / 64) * 64;
  S = (M + rps - 1) / rps;
  w.S = (int)S; w.rows_per_split = (int)rps;
  return w;
}

extern "C" size_t mi355_conv_wgrad_fp8_workspace(const mi355_conv_desc* d) {
  if (wg8_check(d)) return 0;
  const Wg8Plan w = wg8_plan(d);
  return (size_t)w.S * d->Co * w.ldw * sizeof(float);
}

template <int MT>
static void launch_kw8(dim3 grid, hipStream_t st, const WgradKw8Args& k, int x_fmt, int dy_fmt) {
  if (x_fmt) { if (dy_fmt) hipLaunchKernelGGL((wgrad_kw8_kernel<MT, true, true>), grid, dim3(256), 0, st, k); else hipLaunchKernelGGL((wgrad_kw8_kernel<MT, true, false>), grid, dim3(256), 0, st, k); }
  else { if (dy_fmt) hipLaunchKernelGGL((wgrad_kw8_kernel<MT, false, true>), grid, dim3(256), 0, st, k); else hipLaunchKernelGGL((wgrad_kw8_kernel<MT, false, false>), grid, dim3(256), 0, st, k); }
}
template <int MT, int KW>
static void launch_kw28(dim3 grid, hipStream_t st, const WgradKw28Args& k, int x_fmt, int dy_fmt) {
  if (x_fmt) { if (dy_fmt) hipLaunchKernelGGL((wgrad_kw28_kernel<MT, KW, true, true>), grid, dim3(256), 0, st, k); else hipLaunchKernelGGL((wgrad_kw28_kernel<MT, KW, true, false>), grid, dim3(256), 0, st, k); }
  else { if (dy_fmt) hipLaunchKernelGGL((wgrad_kw28_kernel<MT, KW, false, true>), grid, dim3(256), 0, st, k); else hipLaunchKernelGGL((wgrad_kw28_kernel<MT, KW, false, false>), grid, dim3(256), 0, st, k); }
}

extern "C" int mi355_conv_wgrad_fp8(const mi355_conv_desc* d, const void* x8, int x_fmt, const void* dy8, int dy_fmt,
                                    const float* descale_x, const float* descale_dy, float* dw, int accumulate, void* ws,
                                    size_t ws_bytes, void* stream) {
  if (int e = wg8_check(d)) return e;
  if (!x8 || !dy8 || !dw) MI_FAIL(MI355_EINVAL, "wgrad_fp8: null operand");
  if ((x_fmt != 0 && x_fmt != 1) || (dy_fmt != 0 && dy_fmt != 1)) MI_FAIL(MI355_EINVAL, "wgrad_fp8: formats must be 0 (e4m3) or 1 (e5m2)");
  hipStream_t st = as_stream(stream);
  const Wg8Plan w = wg8_plan(d);
  const size_t need = (size_t)w.S * d->Co * w.ldw * sizeof(float);
  const bool direct = (w.S == 1 && !accumulate);
  if (!direct && (ws == nullptr || ws_bytes < need)) MI_FAIL(MI355_EWORKSPACE, "wgrad_fp8 workspace %zu < %zu", ws_bytes, need);
  const long M = (long)d->N * d->Ho * d->Wo, slab = (long)d->Co * w.ldw;
  const unsigned xb = (unsigned)((long)d->N * d->Hi * d->Wi * d->Ci), yb = (unsigned)(M * d->Co);
  if (prof_on()) prof_set_tag("wgrad8 k%ds%d %d>%d @%dx%d n%d", d->kh, d->stride, d->Ci, d->Co, d->Hi, d->Wi, d->N);
  ProfScope ps(st, 2.0 * M * (double)d->Co * w.ldw, (double)xb + (double)yb + 4.0 * d->Co * w.ldw);
  dim3 grid(w.nto * d->kh * w.nti * w.S);
  if (w.s2) {
    WgradKw28Args k; memset(&k, 0, sizeof(k));
    k.X = x8; k.DY = dy8; k.out = direct ? dw : reinterpret_cast<float*>(ws);
    k.descale_x = descale_x; k.descale_dy = descale_dy;
    k.H = d->Hi; k.W = d->Wi; k.Ho = d->Ho; k.Ci = d->Ci; k.Co = d->Co;
    k.lwo = ilog2e(d->Wo); k.lw = k.lwo > 6 ? 6 : k.lwo;
    k.M = (int)M; k.rows_per_split = w.rows_per_split; k.ldw = w.ldw; k.slab_stride = slab; k.nto = w.nto; k.nci = w.nti;
    k.x_bytes = xb; k.dy_bytes = yb; k.dHo = make_fastdiv(d->Ho);
    if (d->kh == 3) { if (w.mt == 1) launch_kw28<1, 3>(grid, st, k, x_fmt, dy_fmt); else launch_kw28<2, 3>(grid, st, k, x_fmt, dy_fmt); }
    else { if (w.mt == 1) launch_kw28<1, 4>(grid, st, k, x_fmt, dy_fmt); else launch_kw28<2, 4>(grid, st, k, x_fmt, dy_fmt); }
    MI_CHECK_LAUNCH("wgrad_kw28");
  } else {
    WgradKw8Args k; memset(&k, 0, sizeof(k));
    k.X = x8; k.DY = dy8; k.out = direct ? dw : reinterpret_cast<float*>(ws);
    k.descale_x = descale_x; k.descale_dy = descale_dy;
    k.H = d->Hi; k.W = d->Wi; k.Ci = d->Ci; k.Co = d->Co;
    k.lwf = ilog2e(d->Wi); k.lw = k.lwf > 6 ? 6 : k.lwf; k.halo = d->Wi > 64;
    k.M = (int)M; k.rows_per_split = w.rows_per_split; k.ldw = w.ldw; k.slab_stride = slab; k.nto = w.nto; k.nci = w.nti;
    k.x_bytes = xb; k.dy_bytes = yb; k.dH = make_fastdiv(d->Hi);
    if (w.mt == 1) launch_kw8<1>(grid, st, k, x_fmt, dy_fmt); else launch_kw8<2>(grid, st, k, x_fmt, dy_fmt);
    MI_CHECK_LAUNCH("wgrad_kw8");
  }
  if (!direct) {
    launch_slab_reduce(reinterpret_cast<const float*>(ws), dw, slab, w.S, slab, accumulate, st);
    MI_CHECK_LAUNCH("slab_reduce");
  }
  return MI355_OK;
}
