// EXPERIMENT (round 5; NOT part of the library, timing only): projection backward with wave roles, THREE waves per SIMD - one matrix
// wave (waves 0-3) and TWO vector waves (waves 4-11) on every SIMD, 768 threads, <= 168 VGPRs.  Same products and vector phase as
// k_proj_bwd_roles.h (v2); each of the eight vector waves owns ONE (pixel block nt, hidden half hm) sub-tile of every 64-row chunk:
// it issues the sub-tile's 12 recompute products, waits for them while its SIMD partner runs its own vector phase, then runs
// GELU / GELU' / dP1 / split / sums in scalar fp32.  Two vector waves per SIMD issue the dependent chains at 0.2 instructions per
// cycle where one reaches 0.1 (profiles/r04_valu_rate_vs_waves_per_simd.txt), the matrix wave's products run underneath.
#pragma once
#include "../../pde_policylearning_amd/csrc/fno_dev.h"
#include "../../pde_policylearning_amd/csrc/k_block_bwd2.h"
#include "../../pde_policylearning_amd/csrc/k_projection.h"
#include "../../pde_policylearning_amd/csrc/k_projection2.h"

// two-term split, element-wise (no packed fp32 instruction: those stall beside the matrix pipe)
FNO_DEV void split2_scalar4(const float (&v)[4], float s, unsigned& h0, unsigned& h1, unsigned& l0, unsigned& l1) {
  _Float16 hh[4], ll[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float t = v[j] * s;
    asm volatile("" : "+v"(t));            // (keeps any packing pass from re-pairing the elements)
    hh[j] = (_Float16)t;
    float r = t - (float)hh[j];
    asm volatile("" : "+v"(r));
    ll[j] = (_Float16)r;
  }
  h0 = __builtin_bit_cast(unsigned, f16x2{hh[0], hh[1]}); h1 = __builtin_bit_cast(unsigned, f16x2{hh[2], hh[3]});
  l0 = __builtin_bit_cast(unsigned, f16x2{ll[0], ll[1]}); l1 = __builtin_bit_cast(unsigned, f16x2{ll[2], ll[3]});
}

#ifndef PBR_PACKED_E
#define PBR_PACKED_E 0      // 1: the vector waves use the packed GELU / split forms (A/B arm)
#endif
// -DPBR_TRACE (tools/pbr_bench.hip): shader-clock stamps of workgroup 0, every wave: [wave][tile < 8][slot < 8][4]: at the slot's
// barrier, behind it, at the end of the slot's work; slot 6 = epilogue
#ifdef PBR_TRACE
__device__ unsigned long long g_pbr[12 * 8 * 8 * 4];
#define PBR_STAMP(slot, k) do { if (blockIdx.x == 0 && pbr_t < 8 && (threadIdx.x & 63) == 0) \
    g_pbr[(((threadIdx.x >> 6) * 8 + pbr_t) * 8 + (slot)) * 4 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PBR_STAMP(slot, k) do { } while (0)
#endif

// one matrix product, then PBR_VPM vector instructions, twelve times (sched_group_barrier masks: 0x8 MFMA, 0x2 VALU, 0x100 DS read)
#ifndef PBR_VPM
#define PBR_VPM 22
#endif
#define PBR_SPREAD() do { _Pragma("unroll") for (int q_ = 0; q_ < 12; ++q_) { \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x2, PBR_VPM, 0); } } while (0)
template <int HID, bool RELU = false>
__global__ void __launch_bounds__(768) k_proj_bwd_r3(ProjBwdArgs a) {
  constexpr int C = 64, NPX = 128, NT = 768, KB = 4, NCH = HID / 64, PITCH = NPX + 4;
  constexpr int ATERM = C * 256, DTERM = 64 * 256;          // bytes per term plane of the a image / one dP1 image
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* aimg = reinterpret_cast<unsigned char*>(smem);             // [2][64 c][128 px] fp16, swz_off
  unsigned char* dr0 = aimg + 2 * ATERM;                                    // two dP1 images [2][64 hid][128 px] fp16, swz_off
  float* douts = reinterpret_cast<float*>(dr0 + 2 * 2 * DTERM);             // dy of the tile (128)
  unsigned* gcnt = reinterpret_cast<unsigned*>(douts + NPX);                // [0] matrix-wave, [1] vector-wave group barrier counters
  float* r3 = douts + NPX + 4;                                              // gout tile C x PITCH (its own region: the matrix waves
                                                                            // transform it while the vector waves fill the dP1 images)
  float* tfwd_s = r3 + C * PITCH;                                           // 16 NJ x (W + 4): forward row table (if x1g)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool matrix = wave < 4;
  const int nt = wave & 3, n0 = nt * 32;            // (vector wave 4 + v: pixel block v & 3, hidden half v >> 2)
  const int vhm = (wave - 4) >> 2;
  const int l31 = lane & 31, half = lane >> 5, l15 = lane & 15, quad = lane >> 4;
  const int tq = l15 >> 2, tp = l15 & 3;
  const int tpx = n0 + 16 * (quad & 1) + 4 * tp, trow = 8 * (quad >> 1) + tq;      // transposed-read roles (k_block_bwd2.h)
  const int dmt = nt >> 1, dnt = nt & 1;                   // dW1 tile of matrix wave nt: hidden 32-block, channel 32-block of the chunk

  const float sa = h2_scale(*a.xmax), sw = h2_scale(a.amax[2]), sd = h2_scale(1.13f * a.amax[3] * a.amax[1]);
  const float inv_aw = 1.f / (sa * sw), inv_dw = 1.f / (sd * sw), inv_da = 1.f / (sd * sa);
  float gk_six, gk_inf;
  gelu_consts(gk_six, gk_inf);

  float gvmax = 0.f;

  const __amdgpu_buffer_rsrc_t rs_wa1 = make_rsrc(a.wa1, (unsigned)((HID / 32) * KB * 2 * 64 * 16));
  const __amdgpu_buffer_rsrc_t rs_wb3 = make_rsrc(a.wa3, (unsigned)((HID / 16) * 2 * 2 * 64 * 16));
  if (a.x1g)
    for (int i = tid; i < 16 * a.NJ * a.W; i += NT) tfwd_s[(i / a.W) * (a.W + 4) + i % a.W] = a.tfwd[i];
  if (tid < 2) gcnt[tid] = 0u;
  __syncthreads();
  unsigned gepoch = 0;               // arrivals so far at this role's group barrier
#ifdef PBR_TRACE
  int pbr_t = -1;
#endif

  // The two roles run their OWN tile loops (same barrier sequence in both: one per chunk slot, one before the epilogue, one behind the last tile): a register of
  // one role is never live in the other's code, so the allocation is the larger of the two, not their union.
  if (matrix) {
    f32x16 dw1acc[NCH];                 // dW1 tile (dmt, dnt) of every chunk
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) dw1acc[k][r] = 0.f;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
#ifdef PBR_TRACE
      ++pbr_t;
#endif
      const int b = tile / a.tiles_per_plane;
      const int px0 = (tile % a.tiles_per_plane) * NPX;
      f32x16 dxh[2], dxl[2];         // dx^T[px block nt][channel block cb]: hh products / cross terms
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dxh[cb][r] = 0.f; dxl[cb][r] = 0.f; }
#pragma unroll 1
      for (int s = 1; s <= NCH; ++s) {
        PBR_STAMP(s, 0);
        __syncthreads();                             // slot s: dP1 of chunk s - 1 is complete
        PBR_STAMP(s, 1);
        {
          // ---- dx and dW1 of chunk s - 1.  W1 fragments from L2 (buffer loads, 1 KB each) into one set of eight:
          //   [W <- dx kk = 0, 1]  dW1 (operands from LDS only: hides the load)  dx(0, 1)  [W <- dx kk = 2, 3]  dx(2, 3)
          const int ch = s - 1;
          const unsigned char* dr = dr0 + (ch & 1) * 2 * DTERM;
          bf16x8 w0[2][2];      // one k block (16 hidden rows), both channel blocks: 16 registers (the budget is 168)
          auto load_x = [&](int kk) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
              for (int t = 0; t < 2; ++t)
                w0[cb][t] = buf_ld8h(rs_wb3, lane * 16, ((((ch * 4 + kk) * 2 + cb) * 2) + t) * 1024);
          };
          auto dx_one = [&](int kk) {
            bf16x8 af[2];
            const int o0 = swz_off(kk * 16 + trow, tpx >> 3) + 2 * (tpx & 7);
            const int o1 = swz_off(kk * 16 + trow + 4, tpx >> 3) + 2 * (tpx & 7);
#pragma unroll
            for (int t = 0; t < 2; ++t) af[t] = cat4(lds_tr16(dr + t * DTERM + o0), lds_tr16(dr + t * DTERM + o1));
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) mfma_split_s<2>(af, w0[cb], dxh[cb], dxl[cb]);
          };
          load_x(0);
          {
            const int ro = dmt * 32 + l31, rc = dnt * 32 + l31;
#pragma unroll
            for (int k = 0; k < NCH; ++k)
              if (k == ch) {
                f32x16 dacc = dw1acc[k];
#pragma unroll 1
                for (int kq = 0; kq < NPX / 16; ++kq) {
                  const int chn = 2 * kq + half;
                  const int od = swz_off(ro, chn), oa = swz_off(rc, chn);
                  bf16x8 af[2], bf[2];
#pragma unroll
                  for (int t = 0; t < 2; ++t) {
                    af[t] = *reinterpret_cast<const bf16x8*>(dr + t * DTERM + od);
                    bf[t] = *reinterpret_cast<const bf16x8*>(aimg + t * ATERM + oa);
                  }
                  dacc = mfma_split<2>(af, bf, dacc);
                }
                dw1acc[k] = dacc;
              }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
          for (int kk = 0; kk < 4; ++kk) {
            dx_one(kk);
            if (kk < 3) load_x(kk + 1);      // (requested behind the products that read the set: ~1 k cycles of L2 latency exposed
          }                                    //  per k block, the matrix wave has the slack)
        }
        PBR_STAMP(s, 2);
      }
      PBR_STAMP(6, 0);
      // ---- epilogue: x act'(u), gout store; then the gout tile for the row DFT (next tile's slot 0) ------------------------------
      __syncthreads();               // E1: every dx / dW1 read of the dP1 images and of the a image is done
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int crow = cb * 32 + l31;
        const size_t ro = ((size_t)b * C + crow) * a.PW + px0 + n0 + 4 * half;
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          v[i] = make_float4((dxh[cb][4 * i] + dxl[cb][4 * i]) * inv_dw, (dxh[cb][4 * i + 1] + dxl[cb][4 * i + 1]) * inv_dw,
                             (dxh[cb][4 * i + 2] + dxl[cb][4 * i + 2]) * inv_dw, (dxh[cb][4 * i + 3] + dxl[cb][4 * i + 3]) * inv_dw);
        if (a.act_in) {
          const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.x + (size_t)b * C * a.PW, (unsigned)(C * a.PW * 4));
          float4 uq[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) uq[i] = buf_ld4(rs, (crow * a.PW + n0 + 4 * half) * 4, (px0 + 8 * i) * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float4 uu = uq[i], dd;
            gelu_both4(uu, dd);
            v[i].x *= dd.x; v[i].y *= dd.y; v[i].z *= dd.z; v[i].w *= dd.w;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) st4(a.gout + ro + 8 * i, v[i]);
        if (a.gmax_out) {
#pragma unroll
          for (int i = 0; i < 4; ++i) gvmax = fmaxf(fmaxf(gvmax, fabsf(v[i].x)), fmaxf(fmaxf(fabsf(v[i].y), fabsf(v[i].z)), fabsf(v[i].w)));
        }
        if (a.x1g) {
          float* r3p = r3 + crow * PITCH + n0 + 4 * half;
#pragma unroll
          for (int i = 0; i < 4; ++i) st4(r3p + 8 * i, v[i]);
        }
      }
      // the row DFT of this tile's gradient, by the matrix waves alone (their own barrier: the vector waves are already in
      // the next tile's first chunk)
      if (a.x1g) {
        group_barrier(gcnt, gepoch, lane);
        row_dft_epilogue<C, NPX, 4>(r3, tfwd_s, a.W + 4, a.x1g, b, px0, a.P, a.W, a.K2out, a.NJ, wave, lane);
      }
    }
    // partial slabs (layout of k_proj_bwd_t): dW1 one slab per workgroup
    if (a.gmax_out) absmax_publish(gvmax, a.gmax_out);
    float* dst = a.dw1_part + (size_t)blockIdx.x * HID * C;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dst[(size_t)(k * 64 + dmt * 32 + acc_row32(r, half)) * C + dnt * 32 + l31] = dw1acc[k][r] * inv_da;
  } else {
    float sdb1[NCH], sdw2[NCH];              // db1 / dW2 sums of this wave's sub-tile (nt, vhm) of every chunk
#pragma unroll
    for (int k = 0; k < NCH; ++k) { sdb1[k] = 0.f; sdw2[k] = 0.f; }
    // the tile's rows of u_L, fetched and committed by the eight vector waves: thread (c = vt / 32 + 16 i, q = vt % 32) of the 512
    // vector threads loads 16 bytes, i = 0..3
    float4 xq[4];
    const int vt = tid - 256;
    const int xvoff = ((vt >> 5) * a.PW + 4 * (vt & 31)) * 4;
    auto issue_x = [&](int tile) {
      const int b = tile / a.tiles_per_plane;
      const int px0 = (tile % a.tiles_per_plane) * NPX;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.x + (size_t)b * C * a.PW, (unsigned)(C * a.PW * 4));
#pragma unroll
      for (int i = 0; i < 4; ++i) xq[i] = buf_ld4(rs, xvoff, (16 * i * a.PW + px0) * 4);
    };
    auto commit = [&](int tile) {
      const int b = tile / a.tiles_per_plane;
      const int px0 = (tile % a.tiles_per_plane) * NPX;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = (vt >> 5) + 16 * i, q = vt & 31;
        float4 t = xq[i];
        if (a.act_in) t = gelu4(t, gk_six, gk_inf);
        put_split4_n<2>(aimg, ATERM, swz_off(c, q >> 1) + 8 * (q & 1), t, sa);
      }
      if (vt < NPX) douts[vt] = a.dy[(size_t)b * a.PW + px0 + vt];
    };
    if ((int)blockIdx.x < a.ntiles) { issue_x(blockIdx.x); commit(blockIdx.x); }
    float b1r[NCH], w2r[NCH];                // b1 and w2 of row 64 ch + 32 vhm + l31
#pragma unroll
    for (int k = 0; k < NCH; ++k) { b1r[k] = a.b1[k * 64 + vhm * 32 + l31]; w2r[k] = a.w2[k * 64 + vhm * 32 + l31]; }
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
#ifdef PBR_TRACE
      ++pbr_t;
#endif
      PBR_STAMP(0, 0);
      {   // the a image and the dy row of this tile are committed (by the eight vector waves): their own barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        gepoch += 8;
        if (lane == 0) __hip_atomic_fetch_add(gcnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(gcnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gepoch) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
      }
      PBR_STAMP(0, 1);
#pragma unroll 1
      for (int ch = 0; ch < NCH; ++ch) {
        if (ch > 0) {
          PBR_STAMP(ch, 0);
          __syncthreads();           // slot ch: dP1 of chunk ch - 1 is complete (and the image of chunk ch - 2 is free again)
          PBR_STAMP(ch, 1);
        }
        // ---- recompute of sub-tile (nt, vhm) of chunk ch: 12 products, W1 fragments from L2 two k blocks at a time -----------------
        f32x16 hi, lo;
#pragma unroll
        for (int r = 0; r < 16; ++r) { hi[r] = 0.f; lo[r] = 0.f; }
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
          bf16x8 wn[2][2];
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < 2; ++t)
              wn[q][t] = buf_ld8h(rs_wa1, lane * 16, ((((ch * 2 + vhm) * KB) + 2 * kp + q) * 2 + t) * 1024);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int kb = 2 * kp + q;
            bf16x8 af[2];
            const int o0 = swz_off(kb * 16 + trow, tpx >> 3) + 2 * (tpx & 7);
            const int o1 = swz_off(kb * 16 + trow + 4, tpx >> 3) + 2 * (tpx & 7);
#pragma unroll
            for (int t = 0; t < 2; ++t) af[t] = cat4(lds_tr16(aimg + t * ATERM + o0), lds_tr16(aimg + t * ATERM + o1));
            mfma_split_s<2>(af, wn[q], hi, lo);
          }
        }
        if (ch == NCH - 1) {                         // the next tile's rows: in flight behind the last chunk's vector phase
          const int nt2 = tile + gridDim.x;
          if (nt2 < a.ntiles) issue_x(nt2);
        }
        // ---- E.  lane <-> hidden row 32 vhm + l31 of the chunk; registers <-> pixels n0 + (r & 3) + 8 (r >> 2) + 4 half -------------
        {
          unsigned char* dr = dr0 + (ch & 1) * 2 * DTERM;
          const int hrow = vhm * 32 + l31;
          float b1v = 0.f, w2v = 0.f;
#pragma unroll
          for (int k = 0; k < NCH; ++k)
            if (k == ch) { b1v = b1r[k]; w2v = w2r[k]; }
          float sdb = 0.f, sdw = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 dy4 = ld4(douts + n0 + 8 * i + 4 * half);
            const float dyv[4] = {dy4.x, dy4.y, dy4.z, dy4.w};
            float pr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) pr[q] = fmaf(hi[4 * i + q] + lo[4 * i + q], inv_aw, b1v);
            float gl4[4], dg4[4], dp[4];
#if PBR_PACKED_E
            { float4 glv = make_float4(pr[0], pr[1], pr[2], pr[3]), dgv;
              gelu_both4(glv, dgv);
              gl4[0] = glv.x; gl4[1] = glv.y; gl4[2] = glv.z; gl4[3] = glv.w; dg4[0] = dgv.x; dg4[1] = dgv.y; dg4[2] = dgv.z; dg4[3] = dgv.w; }
#else
#pragma unroll
            for (int q = 0; q < 4; ++q) gelu_both(pr[q], gl4[q], dg4[q]);
#endif
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              dp[q] = dg4[q] * (w2v * dyv[q]);
              sdw = fmaf(gl4[q], dyv[q], sdw);
              sdb += dp[q];
            }
            const int od = swz_off(hrow, (n0 >> 3) + i) + 8 * half;
#if PBR_PACKED_E
            put_split4_n<2>(dr, DTERM, od, make_float4(dp[0], dp[1], dp[2], dp[3]), sd);
#else
            unsigned h0, h1, l0, l1;
            split2_scalar4(dp, sd, h0, h1, l0, l1);
            *reinterpret_cast<uint2*>(dr + od) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(dr + DTERM + od) = make_uint2(l0, l1);
#endif
            asm volatile("" : "+v"(sdb), "+v"(sdw));
          }
#pragma unroll
          for (int k = 0; k < NCH; ++k)
            if (k == ch) { sdb1[k] += sdb; sdw2[k] += sdw; }
        }
        PBR_STAMP(ch, 2);
      }
      PBR_STAMP(NCH, 0);
      __syncthreads();               // slot NCH: dP1 of the last chunk is complete (the matrix waves finish dx / dW1)
      PBR_STAMP(NCH, 1);
      PBR_STAMP(NCH, 2);
      PBR_STAMP(6, 0);
      __syncthreads();               // E1: the a image, the dy row and both dP1 images are free
      {
        const int nt2 = tile + gridDim.x;
        if (nt2 < a.ntiles) commit(nt2);             // (its rows were requested behind the last chunk)
      }
    }
    // partial slabs (layout of k_proj_bwd_t): db1 / dW2 one per pixel block
    const size_t slab = (size_t)blockIdx.x * 4 + nt;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const float vb = sdb1[k] + __shfl_xor(sdb1[k], 32, 64);
      const float vw = sdw2[k] + __shfl_xor(sdw2[k], 32, 64);
      if (half == 0) {
        a.db1_part[slab * HID + k * 64 + vhm * 32 + l31] = vb;
        a.dw2_part[slab * HID + k * 64 + vhm * 32 + l31] = vw;
      }
    }
  }
}
static inline size_t proj_bwd_r3_lds(int W, int NJ, bool x1g) {
  return (size_t)2 * 64 * 256 + (size_t)4 * 64 * 256 + 128 * 4 + 16 + (size_t)64 * 132 * 4 + (x1g ? (size_t)16 * NJ * (W + 4) * 4 : 0);
}
