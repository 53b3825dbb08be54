// Same-size separable FIR (the blur in front of every stride-2 conv of the discriminator and its adjoint) on the MFMA
// cores, bf16.   y[b, ho, wo, c] = sum_a coef_h[ho][a] sum_e coef_w[wo][e] x[b, idx_h[ho][a], idx_w[wo][e], c]
// reference: Resample / Blur (gans/models/ops/common.py:105-135, upfirdn2d with up = down = 1) at
// gans/models/dusty_v2.py:325-345 and the adjoint autograd derives for it; with the fused activation backward
// FusedLeakyReLUFunctionBackward (gans/models/ops/fused_act/fused_act.py:22-45).
//
// The table-driven VALU kernel of resample.hip is instruction bound on this shape (2.9 TB/s; with loads AND stores
// removed it keeps 73 % of its time): 8 FMAs + bf16 unpack / pack + an LDS round trip per output value.  A FIR along an
// axis is a banded Toeplitz matrix, so each pass is one 16x16x32 MFMA per 16 outputs x 16 channels with the band as a
// constant operand:
//   W pass   D[c, w'] = sum_w  X^T[c, w] * Tw[w, w']      X^T read from the staged input rows with the transposing
//   H pass   D[c, h'] = sum_h  Z^T[c, h] * Th[h, h']      LDS read (ds_read_b64_tr_b16), Tw / Th built from the tables
// -- about 0.05 instructions per output value instead of 10.  The band operands are built ONCE per table set by
// dgv2_fir_same_mfma_prep (a lane's 16 bytes per step / column tile; built inside the kernel from the sparse rows
// they cost more than the whole filter: 58 of 108 us).  The window of a 16-output tile is 32 inputs, of which a
// band of <= 24 may be used (host contract below); unused window slots carry coefficient 0 and read finite data.
//
// Block = 4 waves, one image x 32 output columns x 32 channels, streaming down H in steps of 8 rows:
//   stage 8 input rows x 48 columns (registers -> LDS, loads two steps ahead of their use)
//   W pass  -> Z ring (bf16, 16 rows = this step's and the previous step's)         8 MFMAs per wave
//   H pass  -> output rows [8s - 4, 8s + 4) from the ring (K = 16 real rows)        16 MFMAs per wave
//   outputs regrouped through LDS so that every store instruction writes whole 64-byte pixels (2 KB rows)
// The intermediate is rounded to bf16 exactly where the VALU kernel rounds it (its LDS ring holds the tensor's dtype),
// coefficients must be exact in bf16 (multiples of 1/8 for every [1,3,3,1] FIR of this model and their border sums):
// products are then exact in fp32 and the result differs from the VALU kernel's only by fp32 summation order.
// LDS images are laid out against the lane groups of the instructions that touch them (see each).
#include <type_traits>

#include "gemm_core.h"

namespace {

constexpr int FM_CT = 32;             // output columns per block
constexpr int FM_CB = 32;             // channels per block
constexpr int FM_WIN = FM_CT + 16;    // staged input columns per row: [c0 - 8, c0 + 40)
#ifndef DGV2_FIR_RS
#define DGV2_FIR_RS 8
#endif
constexpr int FM_RS = DGV2_FIR_RS;    // rows per step (8 or 4)
constexpr int FM_RING = 2 * FM_RS;    // ring rows: this step's and the previous step's
static_assert(FM_RS == 8 || FM_RS == 4, "rows per step");
constexpr int FM_XBYTES = FM_RS * FM_WIN * 64;
constexpr int FM_ZROW = 2048 + 64;    // bytes per ring row: rows r and r+1 start 64 B apart modulo 256 (4 rows x 32 B of a
                                      // transposing read hit distinct banks), rows r and r+8 use opposite pixel halves
constexpr int FM_ZBYTES = FM_RING * FM_ZROW;
constexpr int FM_YROW = 2048 + 16;    // rows 16 B apart modulo 256: the 8 rows one H-pass store writes hit distinct banks
constexpr int FM_YBYTES = FM_RS * FM_YROW;
constexpr int FM_NLD = FM_RS * FM_WIN * 4 / 256;   // 16-byte staging loads per thread and step (6)

struct FMGeom {
  int B, C, H, W, nt;
  const uint4* bands;     // [H/8 + 1][64] H bands (one per step), then [W/16][64] W bands (one per 16-column tile)
  const bf16_t* ref;      // ACT: forward output of the activation, laid out like y
  float alpha, ascale;
  float* partial;         // ACT: [B * W/32][C] column sums of the stored values
  uint8_t* y8;            // !ACT: the result as e4m3 (unit scale) instead of bf16 (y is then not written)
};

struct FMTabs {
  int H, W, Eh, Ew;
  const int* idx_h; const float* coef_h; const int* cnt_h;
  const int* idx_w; const float* coef_w; const int* cnt_w;
  int* err;               // the caller's status word: bit DGV2_STATUS_FIR_TABLE when a table entry breaks the contract
};

typedef __attribute__((ext_vector_type(4))) unsigned fm_u32x4;
typedef __attribute__((address_space(3))) s16x4 fm_lds_s16x4;

__device__ __forceinline__ unsigned fm_pack2(float a, float b) {
  union { bf16_t h[2]; unsigned u; } p;
  p.h[0] = (bf16_t)a;
  p.h[1] = (bf16_t)b;
  return p.u;
}

// Band operands from the sparse-row tables.  Block t < H/8 + 1: the H band of step t -- lane (n = h', kg) holds
// Th[8kg .. 8kg+7][h'] for output row 8t - 4 + h' (h' < 8), ring slot = input row & 15, zero for kg >= 2 and h' >= 8.
// Block H/8 + 1 + t: the W band of column tile t -- lane (n = w', kg) holds Tw[8kg .. 8kg+7][w'] over the window of 32
// columns starting at 16t - 8 (ring).  Entries of a row that name the same input are summed before the rounding.
__global__ __launch_bounds__(64) void fir_bands_kernel(uint4* __restrict__ bands, FMTabs t) {
  const int lane = threadIdx.x, li = lane & 15, kg = lane >> 4;
  const int nh = t.H / FM_RS + 1;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  bool bad = false;
  if ((int)blockIdx.x < nh) {
    const int s = blockIdx.x, ho = s * FM_RS - FM_RS / 2 + li;
    if (li < FM_RS && ho >= 0 && ho < t.H) {
      const int n = t.cnt_h[ho];
      for (int e = 0; e < n; ++e) {
        const int r = t.idx_h[ho * t.Eh + e];
        const float cf = t.coef_h[ho * t.Eh + e];
        if (r < s * FM_RS - FM_RS || r >= s * FM_RS + FM_RS || r < 0 || r >= t.H) bad = true;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if ((r & (FM_RING - 1)) == 8 * kg + j) v[j] += cf;
      }
    }
  } else {
    const int wt = blockIdx.x - nh, wo = 16 * wt + li;
    const int n = t.cnt_w[wo];
    for (int e = 0; e < n; ++e) {
      int rel = t.idx_w[wo * t.Ew + e] - (16 * wt - 8);
      rel = rel < 0 ? rel + t.W : (rel >= t.W ? rel - t.W : rel);
      const float cf = t.coef_w[wo * t.Ew + e];
      if ((unsigned)rel >= 32u) bad = true;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (rel == 8 * kg + j) v[j] += cf;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if ((float)(bf16_t)v[j] != v[j]) bad = true;   // coefficients (and their sums) must be exact in bf16
  bands[(int64_t)blockIdx.x * 64 + lane] =
      make_uint4(fm_pack2(v[0], v[1]), fm_pack2(v[2], v[3]), fm_pack2(v[4], v[5]), fm_pack2(v[6], v[7]));
  if (bad) atomicOr(t.err, DGV2_STATUS_FIR_TABLE);
}

template <bool ACT>
__global__ __launch_bounds__(256, 2) void fir_same_mfma_kernel(bf16_t* __restrict__ y, const bf16_t* __restrict__ x, FMGeom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Xs = smem;
  unsigned char* Zs = Xs + FM_XBYTES;
  unsigned char* Ys = Zs + FM_ZBYTES;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, kg = lane >> 4;   // MFMA lane coordinates: row / column index, k group (or channel quad)
  const int q = li >> 2, p = li & 3;          // transposing read: this lane addresses row q, channels 4p..4p+3 of a 4 x 16 block
  // XCD-aware block order: the hardware deals consecutive block ids round-robin over the 8 XCDs (one L2 each); every XCD
  // gets a contiguous range of (image, channel group, column tile), so the 16 halo columns a block shares with its
  // neighbours are fetched into one L2 once (PMC before: fetch = 1.50 x the input, exactly the 48 / 32 staged columns)
  int vid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  {
    const int nblk = gridDim.x * gridDim.y * gridDim.z;
    if ((nblk & 7) == 0) vid = (vid & 7) * (nblk >> 3) + (vid >> 3);
  }
  const int c0col = (vid % (int)gridDim.x) * FM_CT;
  const int cb0 = ((vid / (int)gridDim.x) % (int)gridDim.y) * FM_CB;
  const int b = vid / (int)(gridDim.x * gridDim.y);
  const int64_t img = (int64_t)b * g.H * g.W * g.C;
  const bf16_t* xb = x + img;

  // ring := 0 (window slots of rows that do not exist must hold finite values); this block's two W bands
  for (int i = tid; i < FM_ZBYTES / 16; i += 256) reinterpret_cast<uint4*>(Zs)[i] = make_uint4(0, 0, 0, 0);
  const int nsteps = g.H / FM_RS;   // W passes; the H pass runs once more for the last four output rows
  const uint4* bands_h = g.bands;
  uint4 tw[2];
  tw[0] = g.bands[(int64_t)(nsteps + 1 + c0col / 16) * 64 + lane];
  tw[1] = g.bands[(int64_t)(nsteps + 2 + c0col / 16) * 64 + lane];

  // ---- staging: 16-byte unit id = tid + 256 i -> (row r, window column j, channel octet o8) ----
  // LDS image of the staged rows: pixel-major 64-byte pixels; pixels with bit 3 of the window column set store their two
  // 32-byte halves swapped, so the 2 x 4 pixel rows (j and j + 8) a transposing read of 32 lanes touches cover all banks
  int goff[FM_NLD];        // element offset of the unit inside row 0 of the image
  int xoff[FM_NLD];        // byte offset in Xs
#pragma unroll
  for (int i = 0; i < FM_NLD; ++i) {
    const int id = tid + 256 * i;
    const int o8 = id & 3, j = (id >> 2) % FM_WIN, r = id / (4 * FM_WIN);
    int gc = c0col - 8 + j;
    gc = gc < 0 ? gc + g.W : (gc >= g.W ? gc - g.W : gc);
    goff[i] = gc * g.C + cb0 + o8 * 8;
    xoff[i] = ((r * FM_WIN + j) * 4 + (o8 ^ (((j >> 3) & 1) << 1))) * 16;
  }
  // The rows of step s+2 are requested while step s runs (one register set, written to LDS at the start of step s+1).
  // Deeper prefetching through hand-issued asm loads with counted waits was tried and dropped: register sets that are
  // in flight across the loop's back edge get copied there by the compiler (before their data has arrived).
  fm_u32x4 rg[FM_NLD];
  auto issue = [&](int s) {         // rows [8s, 8s+8)
#pragma unroll
    for (int i = 0; i < FM_NLD; ++i) {
      const int r = (tid + 256 * i) / (4 * FM_WIN);
      rg[i] = *reinterpret_cast<const fm_u32x4*>(xb + (int64_t)(s * FM_RS + r) * g.W * g.C + goff[i]);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < FM_NLD; ++i) *reinterpret_cast<fm_u32x4*>(Xs + xoff[i]) = rg[i];
  };
  issue(0);
  commit();
  if (1 < nsteps) issue(1);
  uint4 th = bands_h[lane];   // H band of step 0
  __syncthreads();

  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;

  for (int s = 0; s <= nsteps; ++s) {
    // ---- W pass: rows 2*wave, 2*wave + 1 of the step x 2 column tiles x 2 channel tiles ----
    // All fragment reads of the pass are issued before its first ring write: the staged rows and the ring are one LDS
    // array to the compiler, which keeps reads behind writes it cannot tell apart -- written job by job the pass was a
    // serial chain of read -> MFMA -> write latencies.
    if (s < nsteps) {
      uint4 af[FM_RS];
#pragma unroll
      for (int job = 0; job < FM_RS; ++job) {
        const int rr = job >> 2, ct = (job >> 1) & 1, wt = job & 1;
        const int r = (FM_RS / 4) * wave + rr;
        // A = X^T [c, w]: lane addresses window column 16 wt + 8 kg + q (+ 4), channels 16 ct + 4p .. + 3
        const int j = 16 * wt + 8 * kg + q;
        const int o8 = (2 * ct + (p >> 1)) ^ ((kg & 1) << 1);
        const unsigned char* a0 = Xs + ((r * FM_WIN + j) * 4 + o8) * 16 + (p & 1) * 8;
        union { uint4 u; s16x4 h[2]; } a;
        a.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fm_lds_s16x4*)a0);
        a.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fm_lds_s16x4*)(a0 + 4 * 64));
        af[job] = a.u;
      }
      f32x4 acc[FM_RS];
#pragma unroll
      for (int job = 0; job < FM_RS; ++job) {
        acc[job] = (f32x4){0.f, 0.f, 0.f, 0.f};
        Mfma16<bf16_t>::run(acc[job], af[job], tw[job & 1]);
      }
#pragma unroll
      for (int job = 0; job < FM_RS; ++job) {
        const int rr = job >> 2, ct = (job >> 1) & 1, wt = job & 1;
        const int slot = (s * FM_RS + (FM_RS / 4) * wave + rr) & (FM_RING - 1);
        // D: lane (n = w' = li, kg) holds channels 16 ct + 4 kg .. + 3 of column 16 wt + li -> one 8-byte ring write.
        // Ring pixel = 64 B = 8 units of 4 channels; unit u sits at u ^ (wp >> 2) (16 lanes of a store: the four
        // pixels sharing wp & 3 take four different units) ^ 4 for ring rows 8..15 (see FM_ZROW)
        const int wp = 16 * wt + li;
        const int u = (4 * ct + kg) ^ ((wp >> 2) & 7) ^ ((slot >> 3) << 2);
        uint2 pk = make_uint2(fm_pack2(acc[job][0], acc[job][1]), fm_pack2(acc[job][2], acc[job][3]));
        *reinterpret_cast<uint2*>(Zs + slot * FM_ZROW + wp * 64 + u * 8) = pk;
      }
    }
    __syncthreads();   // ring rows of this step written; the staged rows are free
    if (s + 1 < nsteps) commit();          // rows of step s+1 (in registers since the previous step)
    if (s + 2 < nsteps) issue(s + 2);
    const uint4 th_next = bands_h[(int64_t)min(s + 1, nsteps) * 64 + lane];
    // ACT: the activation outputs this step's epilogue needs, requested here so that they arrive under the H pass
    vec16<bf16_t> fref[FM_RS / 2];
    if constexpr (ACT) {
#pragma unroll
      for (int i = 0; i < FM_RS / 2; ++i) {
        const int id = tid + 256 * i;
        const int ho = min(max(s * FM_RS - FM_RS / 2 + (id >> 7), 0), g.H - 1);   // rows outside the image: any valid address
        fref[i].load(g.ref + img + ((int64_t)ho * g.W + c0col + ((id >> 2) & 31)) * g.C + cb0 + (id & 3) * 8);
      }
    }

    // ---- H pass: pixels 8*wave .. 8*wave + 7 x 2 channel tiles (reads, MFMAs, writes: as in the W pass) ----
    {
      uint4 af[16];
#pragma unroll
      for (int job = 0; job < 16; ++job) {
        const int wp = 8 * wave + (job >> 1), ct = job & 1;
        // A = Z^T [c, slot]: lane addresses ring row 8 (kg & 1) + q (+ 4) (k groups 2, 3 re-read rows 0..15: coefficient 0)
        const int hi = FM_RS == 8 ? (kg & 1) : 0;   // 4-row steps: the ring has 8 rows, every k group re-reads them
        const int slot = 8 * hi + q;
        const int u = (4 * ct + p) ^ ((wp >> 2) & 7) ^ (hi << 2);
        const unsigned char* a0 = Zs + slot * FM_ZROW + wp * 64 + u * 8;
        union { uint4 u4; s16x4 h[2]; } a;
        a.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fm_lds_s16x4*)a0);
        a.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fm_lds_s16x4*)(a0 + 4 * FM_ZROW));
        af[job] = a.u4;
      }
      f32x4 acc[16];
#pragma unroll
      for (int job = 0; job < 16; ++job) {
        acc[job] = (f32x4){0.f, 0.f, 0.f, 0.f};
        Mfma16<bf16_t>::run(acc[job], af[job], th);
      }
      // D: lane (n = h' = li, kg) holds channels 16 ct + 4 kg .. + 3 of output row 8s - 4 + li at pixel wp
      if (li < FM_RS) {
#pragma unroll
        for (int job = 0; job < 16; ++job) {
          const int wp = 8 * wave + (job >> 1), ct = job & 1;
          uint2 pk = make_uint2(fm_pack2(acc[job][0], acc[job][1]), fm_pack2(acc[job][2], acc[job][3]));
          *reinterpret_cast<uint2*>(Ys + li * FM_YROW + wp * 64 + (4 * ct + kg) * 8) = pk;
        }
      }
    }
    __syncthreads();   // output rows staged; staged input rows of step s+1 visible

    // ---- outputs: 8 rows x 32 pixels x 4 octets, one 16-byte store per unit, 2 KB contiguous per row ----
#pragma unroll
    for (int i = 0; i < FM_RS / 2; ++i) {
      const int id = tid + 256 * i;
      const int o8 = id & 3, px = (id >> 2) & 31, n = id >> 7;
      const int ho = s * FM_RS - FM_RS / 2 + n;
      if (ho < 0 || ho >= g.H) continue;   // (inside the lambda's unrolled loop)
      vec16<bf16_t> o;
      o.raw = *reinterpret_cast<const uint4*>(Ys + n * FM_YROW + px * 64 + o8 * 16);
      const int64_t off = img + ((int64_t)ho * g.W + c0col + px) * g.C + cb0 + o8 * 8;
      if constexpr (ACT) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float a = o.get(j);
          o.set(j, (fref[i].get(j) > 0.f ? a : a * g.alpha) * g.ascale);
          bsum[j] += o.get(j);   // the reference sums the rounded gradient
        }
      }
      if constexpr (!ACT) {
        if (g.y8) {   // e4m3 output (fp8.hip): 8 bytes per unit; a pixel's 32 channels of this block are 32 contiguous bytes
          float f[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = o.get(j);
          *reinterpret_cast<uint2*>(g.y8 + off) = pack_fp8x8(f);
          continue;
        }
      }
      if (g.nt) o.store_nt(y + off);
      else o.store(y + off);
    }
    th = th_next;
    // the next W pass writes ring rows the H pass above has finished reading (barrier) and reads staged rows committed
    // before that barrier; the next H pass writes Ys only after its own barrier, i.e. after these reads
  }

  if constexpr (ACT) {
    // thread t always handled channel octet t & 3: fold the 64 threads of an octet through LDS
    float* fold = reinterpret_cast<float*>(Xs);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) fold[tid * 8 + j] = bsum[j];
    __syncthreads();
    if (tid < FM_CB) {
      const int o8 = tid >> 3, j = tid & 7;
      float s2 = 0.f;
      for (int t = o8; t < 256; t += 4) s2 += fold[t * 8 + j];
      g.partial[((int64_t)b * gridDim.x + c0col / FM_CT) * g.C + cb0 + tid] = s2;
    }
  }
}

__global__ __launch_bounds__(256) void fm_bias_reduce_kernel(float* __restrict__ gb, const float* __restrict__ partial,
                                                             int nblk, int C) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  float s = 0.f;
#pragma unroll 4
  for (int k = threadIdx.x; k < nblk; k += 256) s += partial[(int64_t)k * C + c];
  s = block_sum(s, red);
  if (threadIdx.x == 0) gb[c] = s;
}

bool fm_covers(int B, int C, int H, int W) {
  return B > 0 && C % FM_CB == 0 && H % FM_RS == 0 && H >= FM_RS && W % FM_CT == 0 && W >= FM_CT && B < 65536 &&
         C / FM_CB < 65536;
}

constexpr size_t FM_LDS = (size_t)FM_XBYTES + FM_ZBYTES + FM_YBYTES;

int64_t fm_band_bytes(int H, int W) { return 1024 * ((int64_t)H / FM_RS + 1 + W / 16); }

}  // namespace

// Band operands of dgv2_fir_same_mfma for one SAME-SIZE table set (sparse rows exactly as for dgv2_resample_tab): built
// once, kept by the caller next to the tables.  bands: device buffer of >= *bytes_needed bytes (a call with bands == NULL
// only reports *bytes_needed).  Contract on the tables (a violation raises bit DGV2_STATUS_FIR_TABLE of the caller's device word *status):
//   |idx_h[ho][a] - ho| <= 4,   (idx_w[wo][e] - wo + 8) mod W < 24,   every coefficient -- and every sum of the
//   coefficients of one row that name the same input -- exact in bf16.
// DGV2_ENOTSUP unless H % 8 == 0 and W % 32 == 0.
extern "C" int dgv2_fir_same_mfma_prep(void* bands, int64_t bands_bytes, int64_t* bytes_needed, const int* idx_h,
                                       const float* coef_h, const int* cnt_h, int Eh, const int* idx_w, const float* coef_w,
                                       const int* cnt_w, int Ew, int H, int W, int* status, void* stream) {
  if (H < FM_RS || H % FM_RS || W < FM_CT || W % FM_CT || Eh < 1 || Ew < 1) return DGV2_ENOTSUP;
  const int64_t need = fm_band_bytes(H, W);
  if (bytes_needed) *bytes_needed = need;
  if (!bands) return bytes_needed ? 0 : DGV2_EINVAL;
  if (bands_bytes < need || !aligned16(bands) || !idx_h || !coef_h || !cnt_h || !idx_w || !coef_w || !cnt_w) return DGV2_EINVAL;
  if (!status) return DGV2_EINVAL;
  FMTabs t{H, W, Eh, Ew, idx_h, coef_h, cnt_h, idx_w, coef_w, cnt_w, status};
  fir_bands_kernel<<<(int)(need / 1024), 64, 0, (hipStream_t)stream>>>((uint4*)bands, t);
  DGV2_RETURN_LAST();
}

// y = R x for a same-size separable resampling whose band operands dgv2_fir_same_mfma_prep built, bf16, x / y
// [B, H, W, C] contiguous.  DGV2_ENOTSUP unless C % 32 == 0, H % 8 == 0, W % 32 == 0 (then: dgv2_resample_tab).
extern "C" int dgv2_fir_same_mfma(void* y, const void* x, const void* bands, int B, int C, int H, int W, void* stream) {
  if (!y || !x || !bands) return DGV2_EINVAL;
  if (!fm_covers(B, C, H, W)) return DGV2_ENOTSUP;
  if (!aligned16(x) || !aligned16(y) || !aligned16(bands)) return DGV2_EINVAL;
  FMGeom g{B, C, H, W, nt_output((int64_t)B * H * W * C * 2) ? 1 : 0, (const uint4*)bands, nullptr, 1.f, 1.f, nullptr, nullptr};
  auto kern = fir_same_mfma_kernel<false>;
  if (FM_LDS > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FM_LDS);
    if (e != hipSuccess) return (int)e;
  }
  kern<<<dim3(W / FM_CT, C / FM_CB, B), 256, FM_LDS, (hipStream_t)stream>>>((bf16_t*)y, (const bf16_t*)x, g);
  DGV2_RETURN_LAST();
}

// dgv2_fir_same_mfma with the result stored as e4m3 (OCP fp8, unit scale, saturating): y8 [B,H,W,C] bytes.  The values
// are the bf16 kernel's (rounded to bf16 where it rounds), then rounded once more to e4m3 at the store.
extern "C" int dgv2_fir_same_mfma_q8(void* y8, const void* x, const void* bands, int B, int C, int H, int W, void* stream) {
  if (!y8 || !x || !bands) return DGV2_EINVAL;
  if (!fm_covers(B, C, H, W)) return DGV2_ENOTSUP;
  if (!aligned16(x) || (reinterpret_cast<uintptr_t>(y8) & 7) || !aligned16(bands)) return DGV2_EINVAL;
  FMGeom g{B, C, H, W, 0, (const uint4*)bands, nullptr, 1.f, 1.f, nullptr, (uint8_t*)y8};
  auto kern = fir_same_mfma_kernel<false>;
  if (FM_LDS > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FM_LDS);
    if (e != hipSuccess) return (int)e;
  }
  kern<<<dim3(W / FM_CT, C / FM_CB, B), 256, FM_LDS, (hipStream_t)stream>>>((bf16_t*)nullptr, (const bf16_t*)x, g);
  DGV2_RETURN_LAST();
}

// dgv2_resample_tab_actbwd for the same-size case on the MFMA kernel: y = R(x) * (ref > 0 ? 1 : alpha) * scale and
// gb[c] = sum of the stored y.  scratch fp32 [>= *blocks_needed * C]; a call with scratch == NULL only reports
// *blocks_needed.  Coverage as dgv2_fir_same_mfma.
extern "C" int dgv2_fir_same_mfma_actbwd(void* y, float* gb, float* scratch, int64_t scratch_elems, int64_t* blocks_needed,
                                         const void* x, const void* ref, const void* bands, int B, int C, int H, int W,
                                         float alpha, float scale, void* stream) {
  if (!fm_covers(B, C, H, W)) return DGV2_ENOTSUP;
  const int64_t blocks = (int64_t)B * (W / FM_CT);
  if (blocks_needed) *blocks_needed = blocks;
  if (!scratch) return blocks_needed ? 0 : DGV2_EINVAL;
  if (!y || !gb || !x || !ref || !bands) return DGV2_EINVAL;
  if (scratch_elems < blocks * C || !aligned16(x) || !aligned16(y) || !aligned16(ref) || !aligned16(bands)) return DGV2_EINVAL;
  FMGeom g{B, C, H, W, 0, (const uint4*)bands, (const bf16_t*)ref, alpha, scale, scratch, nullptr};
  auto kern = fir_same_mfma_kernel<true>;
  if (FM_LDS > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FM_LDS);
    if (e != hipSuccess) return (int)e;
  }
  hipStream_t st = (hipStream_t)stream;
  kern<<<dim3(W / FM_CT, C / FM_CB, B), 256, FM_LDS, st>>>((bf16_t*)y, (const bf16_t*)x, g);
  fm_bias_reduce_kernel<<<C, 256, 0, st>>>(gb, scratch, (int)blocks, C);
  DGV2_RETURN_LAST();
}

