// tools/probe/attn_fwd3_experiment.hip -- NOT built into the library: the third form of the attention forward tried in round 3
// (64 query rows per wave, one wave per SIMD, the softmax of one row block interleaved with the MFMAs of the other; lazy
// reference maximum so that O is not rescaled every step).  It is correct (tools/attn_v2_check.py with the variants 322 / 314 added,
// in the tree of commit cff7a94^: ALL OK) and SLOWER than the second form: T = 767 28.6 vs 20.4 us, T = 2048 110 vs 76.5, ViT 13.2 vs 8.9
// (profiles/r03_attention_third_form.txt).  Reason, from the ISA: above 256 registers hipcc produces MFMA results in AGPR form
// and VALU instructions cannot read AGPRs, so every step carries 445 (D = 128) / 288 (D = 64) v_accvgpr_read / _write
// copies between the two halves of the register file (128 of them for the O accumulators at the loop top) -- ~40 % more VALU
// work on the pipe that already bounds the step.  The design needs explicit register placement (O and Q in AGPRs, scores /
// P / fragments in VGPRs), i.e. an assembly kernel.  The text below was a section of csrc/attention_v2.hip (it uses that
// file's helpers: a2_buffer_piece, a2_other_half, a2_kswz, a2_tile_bytes, Attn2Args) and its dispatch arm.

// ---------------------------------------------------------------------------------------------------------------------
// Third form (round 3c): 64 query rows per wave, ONE wave per SIMD (workgroup = NWG row waves x NG key groups = 4 waves).
// The second form is bound by the LDS: a K / V fragment read feeds one MFMA (32 query rows per wave), 256 KB of LDS reads
// per step and CU, and its two LDS-bound phases overlap nothing (profiles/r03_attention_stamps.txt).  Here a fragment
// feeds TWO MFMAs (row blocks 0 and 1 of the wave), and the overlap of matrix pipe and VALU happens INSIDE the wave: the
// softmax of one row block is interleaved, instruction by instruction, with the MFMAs of the other
//     K reads | QK(0) | V reads, next tile's pieces | { QK(1) || softmax(0) } | { PV(0) || softmax(1) } | PV(1) | barrier
// Same LDS images, fragment reads, register mapping, masks and merge as the second form.
template <int D, int NWG, int NG>
__global__ __launch_bounds__(NWG* NG * 64) void flash_attn_fwd3_kernel(Attn2Args p) {
  constexpr int RB = 2;                                   // 32-row blocks per wave
  constexpr int QB = NWG * RB * 32;
  constexpr int SLOTS = D / 8, KSTEPS = D / 16, DB = D / 32, NDD = D / 16;
  constexpr int K_BYTES = A2_KVB * D * 2, TILE_BYTES = a2_tile_bytes<D>();
  constexpr int KP = K_BYTES / 1024, VP = NDD * 2;
  constexpr int PPW = (KP + VP) / NWG, KPW = KP / NWG;
  constexpr int RPP = 1024 / (D * 2);
  constexpr int NWAVES = NWG * NG;
  constexpr int QP = QB * D * 2 / 1024;
  constexpr int QPW = (QP + NWAVES - 1) / NWAVES;
  static_assert(KP % NWG == 0 && VP % NWG == 0, "piece split");
  static_assert(QB * D * 2 <= TILE_BYTES, "the Q tile is staged in one tile buffer");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (p.kv_len_dev) p.Tk = *p.kv_len_dev + p.Tq;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave / NWG, wv = wave % NWG;
  const int hi = lane >> 5, ql = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qblock = ((int)gridDim.x - 1 - (int)blockIdx.x) * QB;     // heaviest (latest) causal blocks first
  const int qw0 = qblock + wv * (RB * 32);
  const int off = p.Tk - p.Tq;
  const bf16_t* Qb = p.Q + (size_t)b * p.q_batch + (size_t)h * D;
  const bf16_t* Kb = p.K + (size_t)b * p.k_batch + (size_t)h * D;
  const bf16_t* Vb = p.V + (size_t)b * p.v_batch + (size_t)h * D;
  char* gbuf = smem + grp * (2 * TILE_BYTES);

  int kend = p.Tk;
  if (p.causal) {
    const int last = qblock + QB - 1 + off + 1;
    if (last < kend) kend = last;
  }
  const int ntiles = (kend + A2_KVB - 1) / A2_KVB;
  const int nsteps = (ntiles + NG - 1) / NG;

  // ---- pieces (as the second form: buffer loads, per-lane offset once + scalar tile offset) ----
  const int k_row_b = (int)p.k_row * 2, v_row_b = (int)p.v_row * 2;
  const long k_span = ((long)(p.Tk - 1) * p.k_row + D) * 2, v_span = ((long)(p.Tk - 1) * p.v_row + D) * 2;
  if (k_span > 0x7fffffffL || v_span > 0x7fffffffL) __builtin_trap();
  int pvoff[PPW], pdst[PPW];
  auto piece_row = [&](int j) {
    return j < KPW ? (wv + NWG * j) * RPP + lane / SLOTS : ((wv + NWG * (j - KPW)) & 1) * 32 + (lane >> 1);
  };
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    if (j < KPW) {
      const int pk = wv + NWG * j;
      const int row = pk * RPP + lane / SLOTS;
      pvoff[j] = row * k_row_b + (((lane % SLOTS) ^ a2_kswz<D>(row)) * 8) * 2;
      pdst[j] = pk * 1024;
    } else {
      const int pv = wv + NWG * (j - KPW);
      const int dd = pv >> 1, half = pv & 1;
      pvoff[j] = (half * 32 + (lane >> 1)) * v_row_b + (dd * 16 + (lane & 1) * 8) * 2;
      pdst[j] = K_BYTES + dd * A2_VSUB + half * 1024;
    }
  }
  auto issue = [&](int tile, int buf) {
    const int j0 = tile * A2_KVB;
    char* dst = gbuf + buf * TILE_BYTES;
    if (j0 + A2_KVB <= p.Tk) {
      const int ks = j0 * k_row_b, vs = j0 * v_row_b;
#pragma unroll
      for (int j = 0; j < PPW; ++j)
        a2_buffer_piece(j < KPW ? Kb : Vb, (unsigned)(j < KPW ? k_span : v_span), dst + pdst[j], pvoff[j], j < KPW ? ks : vs);
    } else {
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        const int r = piece_row(j);
        int key = j0 + r;
        if (key > p.Tk - 1) key = p.Tk - 1;
        a2_buffer_piece(j < KPW ? Kb : Vb, (unsigned)(j < KPW ? k_span : v_span), dst + pdst[j],
                        pvoff[j] + (key - r) * (j < KPW ? k_row_b : v_row_b), 0);
      }
    }
  };

  const int k_row_off = ql * (D * 2);
  const int k_sw = a2_kswz<D>(ql);
  const int v_lane_off = K_BYTES + ((lane >> 4) & 1) * A2_VSUB + hi * 128 + (lane & 15) * 8;

  // ---- Q tile by LDS-DMA into group 0's second buffer, then the B fragments of both row blocks ----
  char* qlds = smem + TILE_BYTES;
#pragma unroll
  for (int j = 0; j < QPW; ++j) {
    const int pq = wave + NWAVES * j;
    if (pq < QP) {
      const int row = pq * RPP + lane / SLOTS;
      int qr = qblock + row;
      if (qr > p.Tq - 1) qr = p.Tq - 1;
      const bf16_t* src = Qb + (size_t)qr * p.q_row + ((lane % SLOTS) ^ a2_kswz<D>(row)) * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(qlds + pq * 1024), 16, 0, 0);
    }
  }
  if (grp < ntiles) issue(grp, 0);
  __builtin_amdgcn_s_waitcnt(0x0f70);                     // vmcnt(0)
  __builtin_amdgcn_s_barrier();
  bf16x8 qf[RB][KSTEPS];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const char* qrow = qlds + ((wv * RB + rb) * 32) * (D * 2) + k_row_off;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) qf[rb][kk] = *reinterpret_cast<const bf16x8*>(qrow + (((kk * 2 + hi) ^ k_sw) << 4));
  }
  float16v oacc[RB][DB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[rb][d][r] = 0.f;
  float m_run[RB] = {-INFINITY, -INFINITY}, l_run[RB] = {0.f, 0.f};
  const float sc2 = p.scale * 1.4426950408889634f;
  __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();                           // every wave holds its Q fragments: the staging buffer may be overwritten

  float16v sacc[RB][2];
  bf16x8 pf[RB][2][2];
  float alpha[RB] = {1.f, 1.f};
  bool moved[RB] = {false, false};

  auto v_frag = [&](const char* vt, int kb, int hf, int d) {
    const char* a = vt + d * 2 * A2_VSUB + (kb * 32 + hf * 16) * 32;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(a));
    const short4v up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(a + 8 * 32));
    const short8 vv = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
    return __builtin_bit_cast(bf16x8, vv);
  };
  // masks of row block rb for the tile at key j0: keys beyond Tk, keys above the causal diagonal of the lane's row
  auto mask_block = [&](int rb, int j0) {
    const int qi = qw0 + rb * 32 + ql;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = j0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (key >= p.Tk || (p.causal && key > qi + off)) sacc[rb][kb][r] = -INFINITY;
      }
  };
  // online softmax of row block rb (pure VALU, no branches: it is interleaved with the other block's MFMAs): row max of the
  // raw scores, new running max, alpha, the 32 exponentials (scale folded into the exp2 argument), row sum, P as bf16
  auto softmax_block = [&](int rb) {
    float mt0 = -INFINITY, mt1 = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      mt0 = fmaxf(mt0, sacc[rb][0][r]);
      mt1 = fmaxf(mt1, sacc[rb][1][r]);
    }
    const float mt = a2_other_half(fmaxf(mt0, mt1), true);
    // LAZY reference maximum: the reference every exponential is taken against moves only when some row of the wave
    // outgrew it by more than 2^8 (then P <= 256 in between, exact enough in bf16 / fp32 sums); otherwise alpha = 1 for the
    // whole wave and O is not touched by the VALU -- with O rescaled every step the accumulators (128 registers at D = 128)
    // would have to live in arch VGPRs and the 512-register file could not hold the kernel without copies
    const float m_cand = fmaxf(m_run[rb], mt * sc2);
    const bool grow = __any(m_cand - m_run[rb] > 8.f);
    const float m_new = grow ? m_cand : m_run[rb];
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    alpha[rb] = grow ? __builtin_amdgcn_exp2f(m_run[rb] - m_use) : 1.f;
    moved[rb] = grow;
    const float nm = -m_use;
    float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[rb][kb][r], sc2, nm));
        const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[rb][kb][r + 1], sc2, nm));
        sacc[rb][kb][r] = e0;
        sacc[rb][kb][r + 1] = e1;
        rs0 += e0;
        rs1 += e1;
      }
    const float rs = a2_other_half(rs0 + rs1, false);
    l_run[rb] = l_run[rb] * alpha[rb] + rs;
    m_run[rb] = m_new;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        uint4v pw;
        pw.x = pack_bf16x2(sacc[rb][kb][hf * 8 + 0], sacc[rb][kb][hf * 8 + 1]);
        pw.y = pack_bf16x2(sacc[rb][kb][hf * 8 + 2], sacc[rb][kb][hf * 8 + 3]);
        pw.z = pack_bf16x2(sacc[rb][kb][hf * 8 + 4], sacc[rb][kb][hf * 8 + 5]);
        pw.w = pack_bf16x2(sacc[rb][kb][hf * 8 + 6], sacc[rb][kb][hf * 8 + 7]);
        pf[rb][kb][hf] = __builtin_bit_cast(bf16x8, pw);
      }
  };
  auto rescale_block = [&](int rb) {
    if (__builtin_expect(moved[rb], 0)) {  // the reference maximum moved (rare after the first tiles): rescale O
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[rb][d][r] *= alpha[rb];
    }
  };
  for (int s = 0; s < nsteps; ++s) {
    const int tile = s * NG + grp;
    const int buf = s & 1;
    const int j0 = tile * A2_KVB;
    // a wave skips tiles beyond the keys (ragged tail of the split) or entirely above the causal diagonal of ALL its rows
    const bool active = tile < ntiles && !(p.causal && j0 > qw0 + RB * 32 - 1 + off);
    const char* kt = gbuf + buf * TILE_BYTES;
    bf16x8 vf[2][2][DB];
    if (active) {
      bf16x8 kf[2][KSTEPS];
      const unsigned kbase = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)(kt + k_row_off);
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        const unsigned ka = kbase + ((unsigned)((kk * 2 + hi) ^ k_sw) << 4);
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[0][kk]) : "v"(ka) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[1][kk]) : "v"(ka), "n"(32 * D * 2) : "memory");
      }
      // QK of row block 0: every MFMA pair waits for its two fragment reads only
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[0][kb][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(kf[0][kk]), "+v"(kf[1][kk]) : "n"(2 * (KSTEPS - 1 - kk)));
        sacc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][kk], qf[0][kk], sacc[0][0], 0, 0, 0);
        sacc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1][kk], qf[0][kk], sacc[0][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      const bool tail = j0 + A2_KVB > p.Tk;
      if (tail || (p.causal && j0 + A2_KVB - 1 > qw0 + off)) mask_block(0, j0);
      __builtin_amdgcn_sched_barrier(0);
      // region A: { QK of row block 1 || first half of softmax(0) }, the V^T fragment reads (into the registers the K
      // fragments leave: both sets together do not fit the 512 registers), the rest of softmax(0)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[1][kb][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        sacc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][kk], qf[1][kk], sacc[1][0], 0, 0, 0);
        sacc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1][kk], qf[1][kk], sacc[1][1], 0, 0, 0);
      }
      softmax_block(0);
      const char* vt = kt + v_lane_off;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int d = 0; d < DB; ++d) vf[kb][hf][d] = v_frag(vt, kb, hf, d);
#pragma unroll
      for (int i = 0; i < 2 * KSTEPS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x402, (D == 128 ? 4 : 8), 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 8 * DB, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (tile + NG < ntiles) issue(tile + NG, buf ^ 1);     // behind the step's last LDS read (see the second form)
    if (active) {
      const bool tail = j0 + A2_KVB > p.Tk;
      if (tail || (p.causal && j0 + A2_KVB - 1 > qw0 + 32 + off)) mask_block(1, j0);
      rescale_block(0);
      __builtin_amdgcn_sched_barrier(0);
      // region B: PV of row block 0 || softmax of row block 1 (VALU first: the first MFMA waits for its V fragments)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int d = 0; d < DB; ++d)
            oacc[0][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kb][hf][d], pf[0][kb][hf], oacc[0][d], 0, 0, 0);
      softmax_block(1);
      __builtin_amdgcn_sched_group_barrier(0x402, 24, 0);
#pragma unroll
      for (int i = 0; i < 4 * DB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x402, (D == 128 ? 6 : 12), 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      rescale_block(1);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int d = 0; d < DB; ++d)
            oacc[1][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kb][hf][d], pf[1][kb][hf], oacc[1][d], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);                    // vmcnt(0) lgkmcnt(0): the group's next tile has landed
    __builtin_amdgcn_s_barrier();
  }

  // ---- merge the NG partial states through LDS ----
  constexpr int REGS = RB * (DB * 16 + 2);
  constexpr int MERGE_BYTES = (NG - 1) * NWG * REGS * 64 * 4;
  if (NG > 1) {
    float* mo = reinterpret_cast<float*>(smem);
    if (grp > 0) {
      float* dst = mo + ((size_t)((grp - 1) * NWG + wv) * REGS) * 64 + lane;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((rb * DB + d) * 16 + r) * 64] = oacc[rb][d][r];
        dst[(RB * DB * 16 + 2 * rb) * 64] = m_run[rb];
        dst[(RB * DB * 16 + 2 * rb + 1) * 64] = l_run[rb];
      }
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 1; g < NG; ++g) {
      const float* src = mo + ((size_t)((g - 1) * NWG + wv) * REGS) * 64 + lane;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const float m_o = src[(RB * DB * 16 + 2 * rb) * 64], l_o = src[(RB * DB * 16 + 2 * rb + 1) * 64];
        const float mm = fmaxf(m_run[rb], m_o);
        const float mu = mm == -INFINITY ? 0.f : mm;
        const float a_me = __builtin_amdgcn_exp2f(m_run[rb] - mu), a_o = __builtin_amdgcn_exp2f(m_o - mu);
        l_run[rb] = l_run[rb] * a_me + l_o * a_o;
        m_run[rb] = mm;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            oacc[rb][d][r] = oacc[rb][d][r] * a_me + src[((rb * DB + d) * 16 + r) * 64] * a_o;
      }
    }
  }

  // ---- normalise; O through LDS so that a wave instruction stores whole rows ----
  constexpr int ORS = D * 2 + 16;
  char* olds = smem + MERGE_BYTES + (size_t)wv * (RB * 32) * ORS;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int qi = qw0 + rb * 32 + ql;
    if (p.lse && hi == 0 && qi < p.Tq) p.lse[((size_t)b * p.H + h) * p.Tq + qi] = m_run[rb] + __log2f(l_run[rb]);
    const float inv = l_run[rb] > 0.f ? 1.f / l_run[rb] : 0.f;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint2v w = {pack_bf16x2(oacc[rb][d][g * 4] * inv, oacc[rb][d][g * 4 + 1] * inv),
                          pack_bf16x2(oacc[rb][d][g * 4 + 2] * inv, oacc[rb][d][g * 4 + 3] * inv)};
        *reinterpret_cast<uint2v*>(olds + (rb * 32 + ql) * ORS + (d * 32 + g * 8 + 4 * hi) * 2) = w;
      }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): same wave wrote and reads, in-order LDS
  constexpr int LPR = D / 8;
  constexpr int RPI = 64 / LPR;
#pragma unroll
  for (int it = 0; it < RB * 32 / RPI; ++it) {
    const int r = it * RPI + lane / LPR;
    const int qrow = qw0 + r;
    const uint4v w = *reinterpret_cast<const uint4v*>(olds + r * ORS + (lane % LPR) * 16);
    if (qrow < p.Tq)
      *reinterpret_cast<uint4v*>(p.O + (size_t)b * p.o_batch + (size_t)qrow * p.o_row + (size_t)h * D + (lane % LPR) * 8) = w;
  }
}

template <int D, int NWG, int NG>
int launch_attn3(const Attn2Args& a, int B, hipStream_t st) {
  constexpr int LDS = NG * 2 * a2_tile_bytes<D>();
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert(LDS >= (NG - 1) * NWG * 2 * (D / 32 * 16 + 2) * 64 * 4 + NWG * 64 * (D * 2 + 16),
                "merge area + O staging fit the tile buffers");
  auto kfn = flash_attn_fwd3_kernel<D, NWG, NG>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return g4r_note_hip_error(e, "flash_attn_fwd3: hipFuncSetAttribute");
    attr_set = true;
  }
  dim3 grid(g4r_ceil_div(a.Tq, NWG * 64), a.H, B);
  hipLaunchKernelGGL(kfn, grid, dim3(NWG * NG * 64), LDS, st, a);
  return G4R_OK;
}


// dispatch arm (g4r_attn2_dispatch):
/*
  if (variant >= 300) {                      // third form: 64 rows per wave, one wave per SIMD (300 + NWG * 10 + NG)
    if (head_dim == 128 && variant == 322) rc = launch_attn3<128, 2, 2>(a, B, st);
    else if (head_dim == 64 && variant == 314) rc = launch_attn3<64, 1, 4>(a, B, st);
    else if (head_dim == 64 && variant == 322) rc = launch_attn3<64, 2, 2>(a, B, st);
    else return g4r_note_error(G4R_ERR_INVALID_ARG, "flash_attn_fwd3: unknown variant");
    if (rc != G4R_OK) return rc;
    G4R_CHECK_LAUNCH("flash_attn_fwd3");
    return G4R_OK;
  }
*/
