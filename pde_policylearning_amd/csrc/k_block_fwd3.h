// Forward of one fused FNO block, third generation: INDEPENDENT WAVES on 32-pixel strips fed by LDS-DMA (round 6).
// Same mathematics, arguments and product order as k_blk_fwd_t (k_block_fwd2.h; reference semantics: fno_block.py:123-150 + the
// last-dim passes of spectral_convolution.py:324,342-345): u is bit-identical to that kernel's.
//
// What bounds k_blk_fwd_t is neither HBM nor an execution pipe but WAITING (profiles/r05_pmc_sq.csv: a third of its wave cycles
// parked at s_waitcnt / s_barrier, another third stalled at issue): four barriers per tile couple its waves, the next tile
// travels in 32 VGPRs per lane that can only be requested behind the GEMM, and each workgroup has one tile in flight for about
// half of its time.  Here a wave owns a 32-pixel x 64-channel STRIP from load to store:
//   * the strip arrives by LDS-DMA (global_load_lds_dwordx4, 8 x 1 KiB per strip, no VGPR destination) in a per-wave slot,
//     requested as soon as the previous strip has been read out of it - a whole strip computation ahead of its use;
//   * only the issuing wave reads its slot, so the hand-over is the wave's own counted s_waitcnt vmcnt: no barrier;
//   * the A operand of the GEMM (D^T[px][o] = sum_c act[px][c] W[o][c]) is built in registers straight from the strip - GELU
//     and two-term fp16 split once per value, no pixel-major image in LDS, no commit phase; the wave multiplies its 32 pixels
//     with BOTH 32-channel halves of W, whose fragments sit in LDS in fragment order once per workgroup;
//   * the spectral K-extension keeps its table fragment in registers (a strip sits at a fixed position of its row) and splits
//     the row's spectral coefficients from L2 itself;
//   * the row DFT of the output (EPI != 0) takes the accumulators AS the matrix operand (lane <-> channel, registers <->
//     pixels = the k index): two-term fp16 products against a split table image instead of fp32 MFMAs on the vector lanes,
//     scaled per wave by the strip's own maximum; the four strips of a row are summed through LDS in a fixed order
//     (one pair of barriers per tile, the only coupling left).
// Shapes: 64 channels, rows of 128 pixels (one tile = one row = four strips), <= 8 kept last-dim modes in, <= 8 out, two-term
// fp16 GEMM mode, no lifting / ReLU / addend (everything else stays on k_blk_fwd_t).
#pragma once
#include "fno_dev.h"
#include "k_pointwise.h"

// 16 bytes per lane HBM -> LDS without a register destination: LDS address = lds_dst (wave-uniform, bytes) + 16 * lane,
// source = sbase (wave-uniform) + voff (per lane, bytes).  M0 carries the LDS base and is compiler-reserved: saved and
// restored inside the statement (cdna_hip_programming.md, inline-asm rules).  Not counted by the compiler's s_waitcnt
// bookkeeping: the caller waits with a counted s_waitcnt vmcnt.
#ifndef FNO_BFS_NT
#define FNO_BFS_NT FNO_NT_LOADS // the strip loads are non-temporal (fno_dev.h: streaming loads)
#endif
FNO_DEV void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
#if FNO_BFS_NT
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
#else
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
#endif
}
FNO_DEV unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
FNO_DEV const float* uniform_ptr(const float* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const float*)(((unsigned long long)hi << 32) | lo);
}

static inline size_t blk_fwd_s_lds_bytes(int K2out, bool has_x1) {
  size_t bytes = (size_t)4 * 8192 + 16384;                         // four strip slots + the W fragment image
  if (has_x1) bytes += (size_t)4 * 4096 + (size_t)4 * K2out * 132 * 4;      // store staging (4 KB per wave) + partial spectra
  return bytes;
}

// u stores per strip and wave (16-byte buffer stores, 4 per 32-channel half) - the count behind the counted vmcnt below
#define FNO_BFS_NST 8
#ifndef FNO_BFS_FORCE_TR
#define FNO_BFS_FORCE_TR 0      // (A/B arm: transposed accumulators and 16-byte stores also without a row-DFT epilogue)
#endif
#ifndef FNO_BFS_STAGE
#define FNO_BFS_STAGE 1         // (A/B arm 0: transposed accumulators stored as they stand)
#endif
#ifndef FNO_BFS_EXP
#define FNO_BFS_EXP 0           // timing experiments of tools/bf3_test.hip (results are wrong): 1 = no u stores, 2 = no strip loads
#endif

// LIFT: block 0 of a model with a lifting layer: the strip is u_0 = W_l x + b_l of the <= 4-channel model input (tfno.py:11-20),
// computed where the other variants read their slot (no DMA: four dword loads per lane and strip, one strip ahead)
template <bool ACT_IN, int EPI, bool LIFT = false>
__global__ void __launch_bounds__(256, 2) k_blk_fwd_s(PwFwdArgs a) {
  static_assert(!(LIFT && ACT_IN), "variants");
  constexpr int C = 64, NW = 4, NT = 256, KB = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* lds = reinterpret_cast<unsigned char*>(smem);
  unsigned char* wimg = lds + NW * 8192;                      // [mt 2][kb 4][term 2][lane 64] x 16 B: B fragments of W
  unsigned char* stage = wimg + 16384;                        // [wave 4][row 32][128 B]: transposed accumulators on their way to whole-line stores
  float* part = reinterpret_cast<float*>(stage + 4 * 4096);   // [wave 4][K2out][132]: partial row spectra, (c, re/im) contiguous

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  float six, inf;
  gelu_consts(six, inf);
  const int W = a.W;                                          // 128: the strip of wave q covers columns 32 q .. 32 q + 31
  const int w0 = wave * 32;
  const unsigned PWb = (unsigned)a.PW * 4u;
  unsigned char* slot = lds + wave * 8192;                    // [c 64][px 32] fp32, rows of 128 B
  const unsigned slot_a = lds_addr(slot);
  const unsigned dma_voff = (unsigned)(lane >> 3) * PWb + (unsigned)(lane & 7) * 16u;

  const TileShare ts = pair_share(a.ntiles, a.share32);
  auto strip_src = [&](int tile_) {
    const int tile = a.rev ? a.ntiles - 1 - tile_ : tile_;
    const int b = tile / a.tiles_per_plane, px0 = (tile % a.tiles_per_plane) * 128 + w0;
    return uniform_ptr(a.x + (size_t)b * C * a.PW + px0);
  };
  float xin[4];                                               // LIFT: the model input at this lane's pixel, next strip
  auto issue_dma = [&](int tile) {
    if constexpr (LIFT) {
      const int t2 = a.rev ? a.ntiles - 1 - tile : tile;
      const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (size_t)(t2 / a.tiles_per_plane) * a.CL * a.PW + (t2 % a.tiles_per_plane) * 128 + w0,
                                                  (unsigned)(a.CL - 1) * PWb + 32 * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) xin[k] = buf_ld1(rx, l31 * 4, k * PWb);      // (channels past CL: out of range = 0)
    } else {
      const float* src = strip_src(tile);
#pragma unroll
      for (int i = 0; i < 8; ++i) if (!(FNO_BFS_EXP & 2)) glds16(src + (size_t)i * 8 * a.PW, dma_voff, slot_a + i * 1024);
    }
  };
  // spectral coefficients of the strip's row for channel o = mt * 32 + l31: k = 8 half + j <-> mode 4 half + (j >> 1), re / im
  float zraw[2][8];
  // (buffer loads: descriptor and row in SGPRs, ONE per-lane offset register, the (mode, half) part in the instruction's offset
  // field, modes past K2in out of the descriptor's range = 0 - per-lane 64-bit addresses of eight loads, hoisted out of the strip
  // loop, were 16 registers and, in the LIFT variant, scratch)
  const int z_voff = ((4 * half * C + l31) * 2) * 4;
  auto load_z = [&](int tile_) {
    const int tile = a.rev ? a.ntiles - 1 - tile_ : tile_;
    const int b = tile / a.tiles_per_plane, prow = ((tile % a.tiles_per_plane) * 128) / W;
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.z + ((size_t)b * a.P + prow) * a.K2in * C * 2, (unsigned)a.K2in * C * 2 * 4);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rz, z_voff + ((jj * C + mt * 32) * 2) * 4, 0, 0));
        zraw[mt][2 * jj] = v[0]; zraw[mt][2 * jj + 1] = v[1];
      }
  };
  if (ts.first < ts.end) { issue_dma(ts.first); load_z(ts.first); }      // the first strip travels while the images are built

  // ---- once per workgroup: W fragment image and scales; once per wave: table fragments ----------------------------------------------------------------
  __shared__ float red[NW];
  auto wg_max = [&](float m) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __syncthreads();
    if (lane == 0) red[wave] = m;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) r = fmaxf(r, red[k]);
    return r;
  };
  float bx = *a.xmax;                                         // |x| <= bx; |gelu(x)| <= |x| (LIFT: of the model input)
  float* lws = reinterpret_cast<float*>(lds);                 // LIFT: [C][4] lifting weights (columns >= CL zero) + [C] biases, in the unused slots
  if constexpr (LIFT) stage_lift_params<C>(lws, a.lw, a.lb, a.CL, tid, NT);
  float wraw[2][8];                                           // items tid, tid + 256 of [mt][kb][lane]
  float mw = 0.f;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int item = tid + it * NT, ln = item & 63, kb = (item >> 6) & 3, mt = item >> 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      wraw[it][j] = a.w[(mt * 32 + (ln & 31)) * C + kb * 16 + 8 * (ln >> 5) + j];
      mw = fmaxf(mw, fabsf(wraw[it][j]));
    }
  }
  const float sw = h2_scale(wg_max(mw));
  if constexpr (LIFT) {                                       // |u_0[c]| <= sum_k |lw[c][k]| bx + |lb[c]|  (as k_blk_fwd_t: the same bound, bit for bit)
    float m = 0.f;
    for (int c = tid; c < C; c += NT) {
      const float4 wv = ld4(lws + 4 * c);                     // (staged above; wg_max's barriers made it visible)
      m = fmaxf(m, (fabsf(wv.x) + fabsf(wv.y) + fabsf(wv.z) + fabsf(wv.w)) * bx + fabsf(lws[4 * C + c]));
    }
    bx = wg_max(m);
    if (a.ubound && blockIdx.x == 0 && tid == 0) *a.ubound = bx;      // (the same value in every workgroup) for the backward pass
  }
  const float sx = h2_scale(bx);
  const float inv_xw = 1.f / (sx * sw);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int item = tid + it * NT, ln = item & 63, kb = (item >> 6) & 3, mt = item >> 8;
    bf16x8 f[2];
    split_n_x8<2>(wraw[it], sw, f);
    *reinterpret_cast<bf16x8*>(wimg + (((mt * KB + kb) * 2 + 0) * 64 + ln) * 16) = f[0];
    *reinterpret_cast<bf16x8*>(wimg + (((mt * KB + kb) * 2 + 1) * 64 + ln) * 16) = f[1];
  }
  // forward-table fragments of this wave's strip position (A operand of the row DFT: k <-> pixel as the transposed accumulators
  // hold them, row kk = l31 = 2 k2 + (re, im)), two fp16 terms scaled by the wave's own maximum
  bf16x8 tff[2][2];
  float st = 1.f;
  if constexpr (EPI != 0) {
    float tv[2][8], m = 0.f;
#pragma unroll
    for (int kbd = 0; kbd < 2; ++kbd)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int px = 16 * kbd + 8 * (j >> 2) + 4 * half + (j & 3);
        tv[kbd][j] = l31 < 16 * a.NJ ? a.tfwd[(size_t)l31 * W + w0 + px] : 0.f;
        m = fmaxf(m, fabsf(tv[kbd][j]));
      }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    st = h2_scale(m);
#pragma unroll
    for (int kbd = 0; kbd < 2; ++kbd) split_n_x8<2>(tv[kbd], st, tff[kbd]);
  }
  // inverse-table fragment of this wave's strip position (pixel side of the extension block): k = 8 half + j, column w0 + l31
  bf16x8 tif[3];
  {
    float tv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) tv[j] = 8 * half + j < 2 * a.K2in ? a.tinv[(size_t)(8 * half + j) * W + w0 + l31] : 0.f;
    split3x8(tv, tif[0], tif[1], tif[2]);
  }
  // With a row-DFT epilogue the accumulators are TRANSPOSED (D^T[px][o]: lane <-> channel, registers <-> 4-pixel runs: they are
  // the DFT's matrix operand as they stand, one bias register per half, 16-byte stores).  Without one the plain orientation
  // (D[o][px]: lane <-> pixel, registers <-> channels) leaves as whole 128-byte lines per wave half and dword store.
  constexpr bool TR = EPI != 0 || FNO_BFS_FORCE_TR;
  float bias_o[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) bias_o[mt] = a.bias ? a.bias[mt * 32 + l31] : 0.f;
  float bias_r[2][TR ? 1 : 16];
  if constexpr (!TR) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) bias_r[mt][r] = a.bias ? a.bias[mt * 32 + 4 * half + (r & 3) + 8 * (r >> 2)] : 0.f;
  }
  __syncthreads();

  const int st_voff = (l31 * a.PW + 4 * half) * 4;           // output row l31 of the 32-channel half, pixels 4 half .. (+ 8 g)
  int st_line[4];                                            // whole-line stores: row 8 i + lane / 8, 16-byte chunk lane % 8
#pragma unroll
  for (int i = 0; i < 4; ++i) st_line[i] = ((8 * i + (lane >> 3)) * a.PW + 4 * (lane & 7)) * 4;
  float vmax = 0.f;
  bool first = true;
  for (int tile_ = ts.first; tile_ < ts.end; tile_ += ts.step) {
    const int tile = a.rev ? a.ntiles - 1 - tile_ : tile_;
    const int b = tile / a.tiles_per_plane, px0 = (tile % a.tiles_per_plane) * 128;
    const bool more = tile_ + ts.step < ts.end;
    // ---- the strip has landed: behind its DMA this wave issued only the previous strip's stores (and nothing the first time) ------
    if constexpr (!LIFT) {
      if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((FNO_BFS_EXP & 1) ? 0 : (TR ? FNO_BFS_NST : 32)) : "memory");
    }
    first = false;
    float raw[KB][8];
    const float x0 = xin[0], x1_ = xin[1], x2 = xin[2], x3 = xin[3];      // LIFT: this strip's input (xin is refilled below)
    if constexpr (!LIFT) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[kb][j] = *reinterpret_cast<const float*>(slot + (16 * kb + 8 * half + j) * 128 + l31 * 4);
    }
    bf16x8 zb[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) split3x8(zraw[mt], zb[mt][0], zb[mt][1], zb[mt][2]);
    // (the split is pinned in front of the DMA: the compiler's own wait for zraw counts the loads and stores IT issued, and
    // behind eight more vector-memory operations it does not know about that count would wait for the previous strip's stores)
    asm volatile("" : "+v"(zb[0][0]), "+v"(zb[0][1]), "+v"(zb[0][2]), "+v"(zb[1][0]), "+v"(zb[1][1]), "+v"(zb[1][2]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slot is in registers: refill it
    // ... and request the next strip's spectral coefficients right behind it: both have the whole strip computation to arrive,
    // and both sit in FRONT of this strip's stores in the vmcnt order, so the wait above can leave the stores in flight
    if (more) { issue_dma(tile_ + ts.step); load_z(tile_ + ts.step); }

    // ---- A fragments: (GELU,) two-term split of the wave's 32 pixels x 64 channels ------------------------------------------------
    bf16x8 fa[KB][2];
    // LIFT: a compiler-level memory barrier per strip (this variant has no DMA statement in its loop, and without one every
    // loop-invariant LDS read - the lifting parameters, all sixteen W fragments - is hoisted into registers and from there
    // into scratch) and the parameter pointers opaque per strip (one base register each, the channel in the offset field)
    const float* lwp = lws + 32 * half;
    const float* lbp = lws + 4 * C + 8 * half;
    if constexpr (LIFT) {
      asm volatile("" ::: "memory");
      asm volatile("" : "+v"(lwp), "+v"(lbp));
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      if constexpr (LIFT) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 wv = ld4(lwp + 4 * (16 * kb + j));
          raw[kb][j] = fmaf(wv.w, x3, fmaf(wv.z, x2, fmaf(wv.y, x1_, fmaf(wv.x, x0, lbp[16 * kb + j]))));
        }
      }
      if constexpr (ACT_IN) gelu8(raw[kb], six, inf);
      split_n_x8<2>(raw[kb], sx, fa[kb]);
    }
    // ---- D^T[px][o] for both 32-channel halves --------------------------------------------------------------------------------------
    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x16 hi, lo;
#pragma unroll
      for (int r = 0; r < 16; ++r) { hi[r] = 0.f; lo[r] = 0.f; }
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const unsigned char* wsrc = wimg + ((mt * KB + kb) * 2 * 64 + lane) * 16;
        const f16x8 w0h = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(wsrc));
        const f16x8 w1h = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(wsrc + 1024));
        const f16x8 x0 = __builtin_bit_cast(f16x8, fa[kb][0]), x1 = __builtin_bit_cast(f16x8, fa[kb][1]);
        if constexpr (TR) {
          lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1, w0h, lo, 0, 0, 0);
          lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0, w1h, lo, 0, 0, 0);
          hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0, w0h, hi, 0, 0, 0);
        } else {
          lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, x1, lo, 0, 0, 0);
          lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, x0, lo, 0, 0, 0);
          hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, x0, hi, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) { hi[r] = (hi[r] + lo[r]) * inv_xw; lo[r] = 0.f; }
      auto mmz = [&](const bf16x8& t, const bf16x8& z, const f32x16& c) {
        if constexpr (TR) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(t, z, c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(z, t, c, 0, 0, 0);
      };
      lo = mmz(tif[2], zb[mt][0], lo);
      lo = mmz(tif[1], zb[mt][1], lo);
      lo = mmz(tif[1], zb[mt][0], lo);
      lo = mmz(tif[0], zb[mt][2], lo);
      lo = mmz(tif[0], zb[mt][1], lo);
      hi = mmz(tif[0], zb[mt][0], hi);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = hi[r] + lo[r] + (TR ? bias_o[mt] : bias_r[mt][TR ? 0 : r]);
    }
    // ---- store u; activation; row DFT partial -----------------------------------------------------------------------------------------
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const size_t obase = ((size_t)b * C + mt * 32) * a.PW + px0 + w0;
      const __amdgpu_buffer_rsrc_t ru = make_rsrc(a.u + obase, 31u * PWb + 32 * 4);
      if constexpr (TR && !FNO_BFS_STAGE) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = make_float4(acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]);
          if (!(FNO_BFS_EXP & 1))
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{v.x, v.y, v.z, v.w}), ru, st_voff + 8 * g * 4, 0, 0);
          if (a.umax) vmax = fmaxf(fmaxf(vmax, fabsf(v.x)), fmaxf(fmaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w)));
        }
      } else if constexpr (TR) {
        // A transposed accumulator register holds 32 bytes of each of 32 channel rows: stored as it stands, one instruction
        // touches 32 lines (measured: 136 vs 111 us per launch against whole-line stores).  So the 32 x 32 tile takes a turn
        // through the wave's own 4 KB of LDS - rows of 128 bytes, 16-byte chunks XOR-swizzled by the row, no conflicts either
        // way, no barrier (one wave's LDS operations execute in order) - and leaves as 8 rows x 128 bytes per instruction.
        unsigned char* sg_ = stage + wave * 4096;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = make_float4(acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]);
          st4(reinterpret_cast<float*>(sg_ + l31 * 128 + (((2 * g + half) ^ (l31 & 7)) * 16)), v);
          if (a.umax) vmax = fmaxf(fmaxf(vmax, fabsf(v.x)), fmaxf(fmaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w)));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 8 * i + (lane >> 3);
          const float4 v = ld4(reinterpret_cast<const float*>(sg_ + row * 128 + (((lane & 7) ^ (row & 7)) * 16)));
          if (!(FNO_BFS_EXP & 1))
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{v.x, v.y, v.z, v.w}), ru, st_line[i], 0, 0);
        }
      } else {
        // acc[r] = channel mt * 32 + 4 half + (r & 3) + 8 (r >> 2), pixel l31: the row offset rides in the scalar offset
        const int vo = (4 * half * a.PW + l31) * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[mt][r];
          if (!(FNO_BFS_EXP & 1))
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ru, vo, (unsigned)((r & 3) + 8 * (r >> 2)) * PWb, 0);
          if (a.umax) vmax = fmaxf(vmax, fabsf(v));
        }
      }
      if (FNO_BFS_EXP & 1) asm volatile("" :: "v"(acc[mt]));
    }
    if constexpr (EPI != 0) {
      // g = act_out(u) in place; the wave's own maximum scales the two-term split (exact power of two, undone on the result)
      float gm = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = acc[mt][8 * h8 + j];
          if constexpr (EPI == 2) gelu8(v, six, inf);
#pragma unroll
          for (int j = 0; j < 8; ++j) { acc[mt][8 * h8 + j] = v[j]; gm = fmaxf(gm, fabsf(v[j])); }
        }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) gm = fmaxf(gm, __shfl_xor(gm, o, 64));
      const float sg = h2_scale(gm);
      const float inv_gt = 1.f / (sg * st);
      float* pw = part + (size_t)wave * a.K2out * 132;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x16 hi, lo;
#pragma unroll
        for (int r = 0; r < 16; ++r) { hi[r] = 0.f; lo[r] = 0.f; }
#pragma unroll
        for (int kbd = 0; kbd < 2; ++kbd) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = acc[mt][8 * kbd + j];
          bf16x8 gf[2];
          split_n_x8<2>(v, sg, gf);
          const f16x8 g0 = __builtin_bit_cast(f16x8, gf[0]), g1 = __builtin_bit_cast(f16x8, gf[1]);
          const f16x8 t0 = __builtin_bit_cast(f16x8, tff[kbd][0]), t1 = __builtin_bit_cast(f16x8, tff[kbd][1]);
          // (the table rides on the A side: D[row = kk][col = channel] - every lane holds a channel, and only the registers whose
          // rows are kept outputs are summed, scaled and written, as (re, im) pairs: 8 + 8 + 4 instructions per half at <= 8 kept
          // bins where the other orientation spent 16 + 16 + 16 in 12 of its 32 lanes)
          lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(t0, g1, lo, 0, 0, 0);
          lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(t1, g0, lo, 0, 0, 0);
          hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(t0, g0, hi, 0, 0, 0);
        }
        // D[row kk = (r & 3) + 8 (r >> 2) + 4 half][col = channel mt 32 + l31]: kk = 2 k2 + (re, im); registers 4 q .. 4 q + 3 =
        // rows 8 q + 4 half .. + 3 = bins 4 q + 2 half, 4 q + 2 half + 1
        float* pc = pw + (mt * 32 + l31) * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (8 * q < 2 * a.K2out) {                       // (uniform: whole register quads beyond the kept rows are skipped)
            const int k2 = 4 * q + 2 * half;
            if (k2 < a.K2out)
              *reinterpret_cast<float2*>(pc + k2 * 132) = make_float2((hi[4 * q] + lo[4 * q]) * inv_gt, (hi[4 * q + 1] + lo[4 * q + 1]) * inv_gt);
            if (k2 + 1 < a.K2out)
              *reinterpret_cast<float2*>(pc + (k2 + 1) * 132) = make_float2((hi[4 * q + 2] + lo[4 * q + 2]) * inv_gt, (hi[4 * q + 3] + lo[4 * q + 3]) * inv_gt);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // the four strips of the row, summed in a fixed order: wave w writes float4 pieces [w K2out 8, (w + 1) K2out 8)
      {
        const int e4 = wave * a.K2out * 8 + lane;
        if (lane < a.K2out * 8) {
          const int k2 = e4 >> 5, rem = e4 & 31;
          const float* p0 = part + k2 * 132 + rem * 4;
          const int ws = a.K2out * 132;
          const float4 s0 = ld4(p0), s1 = ld4(p0 + ws), s2 = ld4(p0 + 2 * ws), s3 = ld4(p0 + 3 * ws);
          const float4 s = make_float4(((s0.x + s1.x) + s2.x) + s3.x, ((s0.y + s1.y) + s2.y) + s3.y,
                                       ((s0.z + s1.z) + s2.z) + s3.z, ((s0.w + s1.w) + s2.w) + s3.w);
          st4(a.x1 + (((size_t)b * a.P + px0 / W) * a.K2out) * C * 2 + (size_t)e4 * 4, s);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
  if (a.umax) absmax_publish(vmax, a.umax);
}
