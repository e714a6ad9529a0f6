// tools/probe/attn_fwd2p_experiment.hip -- NOT built into the library: the software-pipelined form of the attention forward tried
// in round 3 (two tiles in flight per wave: { PV MFMAs of tile u-1 || softmax VALU of tile u } behind the QK MFMAs of tile u, every
// LDS fragment read inline asm with counted waits, separate K / V rings with the pieces issued a whole iteration ahead, lazy
// reference maximum).  It is correct (tools/attn_v2_check.py with variants 242 / 224 added: ALL OK) and NOT faster than the
// lock-step form: 1.157 vs 1.167 us per 64-key tile at D = 128 (phase-offset form: 1.056), 0.564 vs 0.603 at D = 64
// (profiles/r03_attention_steps.txt).  What the step / fixed-cost fit of that file shows instead: a launch of the LLaMA prefill
// shape is 6.9 us of fixed cost + 6 steps x 2.33 us on the heaviest workgroup, while the 672 steps of the launch spread evenly
// over 256 CUs would be 2.6 steps -- the fixed cost and the causal imbalance, not the schedule inside a step, are what is left.
// The text below was a section of csrc/attention_v2.hip (it uses that file's helpers) and its dispatch arm.

struct A2True { static constexpr bool value = true; };
struct A2False { static constexpr bool value = false; };

// ---------------------------------------------------------------------------------------------------------------------
// Pipelined form of the second kernel (round 3d).  Same workgroup shape, LDS images, fragment reads and register mapping;
// what changes is WHEN things happen inside a wave.  The lock-step / phase-offset forms above run the QK MFMAs, the
// softmax VALU work and the PV MFMAs of a tile one after the other in every wave, and the two waves of a SIMD do the same
// thing at the same time: a step costs MFMA time + VALU time (2 x 1024 + 2 x ~1150 cycles at D = 128, stamps in
// profiles/r03_attention_stamps.txt), the matrix pipe idles during the softmax and the VALU during the MFMAs.  Here a
// wave keeps TWO tiles in flight (cdna_hip_programming.md T15): iteration u runs
//     R1   QK MFMAs of tile u          ||  first 16 exponentials of tile u-1      (one VALU slice behind every MFMA pair)
//     R2a  PV MFMAs of tile u-1, keys 0-31   ||  last 16 exponentials of tile u-1
//     R2b  PV MFMAs of tile u-1, keys 32-63  ||  row max / reference maximum of tile u
// so the matrix pipe and the VALU of a SIMD are busy together inside ONE wave, whatever its partner does.
//   * every LDS fragment read is inline asm with a counted wait tied to the fragments it releases (the compiler neither
//     reorders them nor adds its own lgkmcnt(0) / vmcnt(0): it does not see LDS reads at all);
//   * K and V tiles have separate 2-slot rings: at the start of iteration u the pieces of K(u+1) and V(u) go out -- both
//     slots were last read in iteration u-1 -- and have the whole iteration to land (one vmcnt(0) + barrier at its end);
//   * lazy reference maximum (T13): the maximum the exponentials are taken against moves only when some row outgrew it by
//     more than 2^2, so O is rescaled (and alpha applied) only then; P <= 4 in between.
template <int D, int NWG, int NG>
__global__ __launch_bounds__(NWG* NG * 64) void flash_attn_fwd2p_kernel(Attn2Args p) {
  constexpr int QB = NWG * 32;
  constexpr int SLOTS = D / 8, KSTEPS = D / 16, DB = D / 32, NDD = D / 16;
  constexpr int K_BYTES = A2_KVB * D * 2, V_BYTES = NDD * A2_VSUB, TILE_BYTES = a2_tile_bytes<D>();
  constexpr int KP = K_BYTES / 1024, VP = NDD * 2;
  constexpr int PPW = (KP + VP) / NWG, KPW = KP / NWG;
  constexpr int RPP = 1024 / (D * 2);
  constexpr int NWAVES = NWG * NG;
  constexpr int QP = QB * D * 2 / 1024;
  constexpr int QPW = (QP + NWAVES - 1) / NWAVES;
  static_assert(KP % NWG == 0 && VP % NWG == 0, "piece split");
  static_assert(QB * D * 2 <= K_BYTES + V_BYTES, "the Q tile is staged in K slot 1 + V slot 0 of group 0");
  static_assert(K_BYTES + V_BYTES == TILE_BYTES, "tile = K image + V image");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // per group: K slot 0 | K slot 1 | V slot 0 | V slot 1
  if (p.kv_len_dev) p.Tk = *p.kv_len_dev + p.Tq;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave / NWG, wv = wave % NWG;
  const int hi = lane >> 5, ql = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qblock = ((int)gridDim.x - 1 - (int)blockIdx.x) * QB;     // heaviest (latest) causal blocks first
  const int qw0 = qblock + wv * 32;
  const int qi = qw0 + ql;
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

  // ---- pieces: buffer loads, per-lane offset once + scalar tile offset (as above), K and V issued separately ----
  const int k_row_b = (int)p.k_row * 2, v_row_b = (int)p.v_row * 2;
  const long k_span = ((long)(p.Tk - 1) * p.k_row + D) * 2, v_span = ((long)(p.Tk - 1) * p.v_row + D) * 2;
  if (k_span > 0x7fffffffL || v_span > 0x7fffffffL) __builtin_trap();
  // the per-lane piece offsets are RECOMPUTED at every issue (a handful of VALU instructions) from a laundered lane id:
  // hoisted to kernel entry they are spilled around the tile loop and reloaded from scratch in front of every piece
  auto issue_range = [&](int tile, char* dst, int jlo, int jhi) {      // pieces [jlo, jhi) of this wave: K pieces or V pieces
    const int j0 = tile * A2_KVB;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const bool whole = j0 + A2_KVB <= p.Tk;
    const int ks = whole ? j0 * k_row_b : 0, vs = whole ? j0 * v_row_b : 0;
#pragma unroll
    for (int j = 0; j < PPW; ++j)
      if (j >= jlo && j < jhi) {
        int voff, row, pdst;
        if (j < KPW) {
          const int pk = wv + NWG * j;
          row = pk * RPP + ln / SLOTS;
          voff = (((ln % SLOTS) ^ a2_kswz<D>(row)) * 8) * 2;
          pdst = pk * 1024;
        } else {
          const int pv = wv + NWG * (j - KPW);
          const int dd = pv >> 1, half = pv & 1;
          row = half * 32 + (ln >> 1);
          voff = (dd * 16 + (ln & 1) * 8) * 2;
          pdst = dd * A2_VSUB + half * 1024;
        }
        int key = row;
        if (!whole) {                                  // ragged last tile: rows beyond the keys re-read row Tk - 1 (masked later)
          key = j0 + row;
          if (key > p.Tk - 1) key = p.Tk - 1;
        }
        a2_buffer_piece(j < KPW ? Kb : Vb, (unsigned)(j < KPW ? k_span : v_span), dst + pdst,
                        voff + key * (j < KPW ? k_row_b : v_row_b), j < KPW ? ks : vs);
      }
  };
  auto issue_k = [&](int tile, int slot) { issue_range(tile, gbuf + slot * K_BYTES, 0, KPW); };
  auto issue_v = [&](int tile, int slot) { issue_range(tile, gbuf + 2 * K_BYTES + slot * V_BYTES, KPW, PPW); };

  const int k_row_off = ql * (D * 2);
  const int k_sw = a2_kswz<D>(ql);
  const int v_lane_off = ((lane >> 4) & 1) * A2_VSUB + hi * 128 + (lane & 15) * 8;

  // ---- Q tile by LDS-DMA (behind group 0's K slot 0), K tile 0 of every group, then the Q fragments ----
  char* qlds = smem + K_BYTES;
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
  if (grp < ntiles) issue_k(grp, 0);
  __builtin_amdgcn_s_waitcnt(0x0f70);                     // vmcnt(0)
  __builtin_amdgcn_s_barrier();
  bf16x8 qf[KSTEPS];
  {
    const char* qrow = qlds + (wv * 32) * (D * 2) + k_row_off;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(qrow + (((kk * 2 + hi) ^ k_sw) << 4));
  }
  float16v oacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;                   // m_run = the (lazy) reference maximum, l_run relative to it
  const float sc2 = p.scale * 1.4426950408889634f;
  __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();                           // every wave holds its Q fragments: the staging area may be overwritten

  // what a tile leaves for the next iteration: P as packed bf16 ([key half][8-key group]) and the rescale it asks for
  struct TileState {
    uint4v pw[2][2];
    float alpha;
    bool moved;
  };
  TileState X0, X1;
  X0.moved = X1.moved = false;
  X0.alpha = X1.alpha = 1.f;

  const unsigned k_lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)(gbuf + k_row_off);
  const unsigned v_lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)(gbuf + 2 * K_BYTES + v_lane_off);

  // one iteration: QK of tile C, then  { PV of the previous tile P_  ||  softmax of tile C }
  auto step = [&](int u, TileState& C, TileState& P_, auto hc_c, auto hp_c) {
    constexpr bool HC = decltype(hc_c)::value, HP = decltype(hp_c)::value;
    const int tile_c = u * NG + grp;
    const int j0 = tile_c * A2_KVB;
    float16v S[2];
    if constexpr (HC) {
      // ---- QK of tile C: at most 8 k-steps' fragments in flight, every MFMA pair waits for its two reads only ----
      const unsigned kbase = k_lds0 + (unsigned)((u & 1) * K_BYTES);
      bf16x8 kf[2][KSTEPS];
      auto kread = [&](int kk) {
        const unsigned ka = kbase + ((unsigned)((kk * 2 + hi) ^ k_sw) << 4);
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[0][kk]) : "v"(ka) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[1][kk]) : "v"(ka), "n"(32 * D * 2) : "memory");
      };
      constexpr int FIRST = KSTEPS < 4 ? KSTEPS : 4;
#pragma unroll
      for (int kk = 0; kk < FIRST; ++kk) kread(kk);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[kb][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        const int issued = (KSTEPS <= 4) ? KSTEPS : (kk < 2 ? 4 : (kk < 4 ? 6 : 8));   // k-steps whose reads are out
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(kf[0][kk]), "+v"(kf[1][kk]) : "n"(2 * (issued - 1 - kk)));
        S[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][kk], qf[kk], S[0], 0, 0, 0);
        S[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1][kk], qf[kk], S[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KSTEPS == 8) {
          if (kk == 1) { kread(4); kread(5); }
          if (kk == 3) { kread(6); kread(7); }
        }
      }
      // masks (diagonal / ragged tiles only: wave-uniform branch)
      if ((j0 + A2_KVB > p.Tk) || (p.causal && j0 + A2_KVB - 1 > qw0 + off)) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = j0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= p.Tk || (p.causal && key > qi + off)) S[kb][r] = -INFINITY;
          }
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // softmax of tile C in slices (pure VALU): slice 0 = row max + lazy reference maximum, slices 1.. = the exponentials
    float nm = 0.f, rs0 = 0.f, rs1 = 0.f;
    auto softmax_head = [&]() {
      float mt0 = -INFINITY, mt1 = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        mt0 = fmaxf(mt0, S[0][r]);
        mt1 = fmaxf(mt1, S[1][r]);
      }
      const float mt = a2_other_half(fmaxf(mt0, mt1), true);
      const float m_cand = fmaxf(m_run, mt * sc2);
      const bool grow = __any(m_cand - m_run > 2.f);
      const float m_new = grow ? m_cand : m_run;
      const float m_use = m_new == -INFINITY ? 0.f : m_new;
      C.alpha = grow ? __builtin_amdgcn_exp2f(m_run - m_use) : 1.f;
      C.moved = grow;
      nm = -m_use;
      m_run = m_new;
    };
    auto exp_slice = [&](int r0, int r1) {               // linear score index 0..31 = [key half][16]
#pragma unroll
      for (int r = r0; r < r1; r += 2) {
        const int half = r >> 4, rr = r & 15;
        const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[half][rr], sc2, nm));
        const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[half][rr + 1], sc2, nm));
        rs0 += e0;
        rs1 += e1;
        const unsigned w = pack_bf16x2(e0, e1);
        const int g = rr >> 3, wi = (rr & 7) >> 1;
        if (wi == 0) C.pw[half][g].x = w;
        else if (wi == 1) C.pw[half][g].y = w;
        else if (wi == 2) C.pw[half][g].z = w;
        else C.pw[half][g].w = w;
      }
    };
    auto softmax_tail = [&]() {
      const float rs = a2_other_half(rs0 + rs1, false);
      l_run = l_run * C.alpha + rs;
    };

    if constexpr (HP) {
      // ---- { PV of tile P_ || softmax of tile C }.  V^T fragments: keys 0-31 first, keys 32-63 in two chunks behind the
      //      MFMA pairs of the first half (reads in flight <= 16 + 16) ----
      const unsigned vbase = v_lds0 + (unsigned)(((u - 1) & 1) * V_BYTES);
      uint2v vlo[2][2][DB], vup[2][2][DB];
      auto vread = [&](int kb, int hf, int d) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vlo[kb][hf][d]) : "v"(vbase),
                     "n"(d * 2 * A2_VSUB + (kb * 32 + hf * 16) * 32) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vup[kb][hf][d]) : "v"(vbase),
                     "n"(d * 2 * A2_VSUB + (kb * 32 + hf * 16) * 32 + 8 * 32) : "memory");
      };
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int d = 0; d < DB; ++d) vread(0, hf, d);
      if (__builtin_expect(P_.moved, 0)) {               // the reference maximum moved at tile P_: rescale O before its PV
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[d][r] *= P_.alpha;
      }
      __builtin_amdgcn_sched_barrier(0);
      constexpr int NP = DB;                               // MFMA pairs per key half (2 MFMAs each)
      constexpr int NSL = 2 * NP;                          // VALU slices = pairs of the whole tile
      auto pv_pair = [&](int kb, int j, int wait_after) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int idx = j * 2 + m, hf = idx / DB, d = idx % DB;
          if (m == 0) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(vlo[kb][hf][d]), "+v"(vup[kb][hf][d]) : "n"(wait_after));
          else asm volatile("" : "+v"(vlo[kb][hf][d]), "+v"(vup[kb][hf][d]));
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int idx = j * 2 + m, hf = idx / DB, d = idx % DB;
          const uint4v vw = {vlo[kb][hf][d].x, vlo[kb][hf][d].y, vup[kb][hf][d].x, vup[kb][hf][d].y};
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vw),
                                                           __builtin_bit_cast(bf16x8, P_.pw[kb][hf]), oacc[d], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      auto valu_slice = [&](int sl) {                      // slice sl of NSL: 0 = head, 1 .. NSL-1 = exponentials, tail at the end
        if constexpr (HC) {
          if (sl == 0) softmax_head();
          else {
            // 32 exponentials over NSL - 1 slices, even counts
            const int per = ((32 / (NSL - 1)) + 1) & ~1;
            const int r0 = (sl - 1) * per, r1 = sl == NSL - 1 ? 32 : (r0 + per < 32 ? r0 + per : 32);
            if (r0 < 32) exp_slice(r0, r1);
          }
          if (sl == NSL - 1) softmax_tail();
          __builtin_amdgcn_sched_barrier(0);
        }
      };
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int behind = (NP - 1 - j) * 4 + (j >= NP / 2 ? 2 * DB : 0);       // reads allowed to stay outstanding
        pv_pair(0, j, behind);
        if (j == NP / 2 - 1) {
#pragma unroll
          for (int d = 0; d < DB; ++d) vread(1, 0, d);
        }
        if (j == NP - 1) {
#pragma unroll
          for (int d = 0; d < DB; ++d) vread(1, 1, d);
        }
        valu_slice(j);
      }
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        pv_pair(1, j, (NP - 1 - j) * 4);
        valu_slice(NP + j);
      }
    } else if constexpr (HC) {
      softmax_head();
      exp_slice(0, 32);
      softmax_tail();
    }
  };

  auto run = [&](int u, TileState& C, TileState& P_) {
    const int tile_c = u * NG + grp;
    // pieces of the NEXT iteration's operands: K(u + 1) and V(u); both slots were last read in iteration u - 1
    if (u + 1 <= nsteps - 1 && tile_c + NG < ntiles) issue_k(tile_c + NG, (u + 1) & 1);
    if (u <= nsteps - 1 && tile_c < ntiles) issue_v(tile_c, u & 1);
    const bool hc = u < nsteps && tile_c < ntiles;
    const bool hp = u > 0 && tile_c - NG < ntiles;
    if (hc && hp) step(u, C, P_, A2True{}, A2True{});
    else if (hc) step(u, C, P_, A2True{}, A2False{});
    else if (hp) step(u, C, P_, A2False{}, A2True{});
    __builtin_amdgcn_s_waitcnt(0x0070);                    // vmcnt(0) lgkmcnt(0): next iteration's K and V have landed
    __builtin_amdgcn_s_barrier();                          // ... for every wave of the group; this iteration's slots are free
  };
  for (int u = 0; u <= nsteps; u += 2) {
    run(u, X0, X1);
    if (u + 1 <= nsteps) run(u + 1, X1, X0);
  }

  // ---- merge the NG partial states through LDS (the tile buffers are free: every wave passed the last barrier) ----
  constexpr int REGS = DB * 16 + 2;
  constexpr int MERGE_BYTES = (NG - 1) * NWG * REGS * 64 * 4;
  if (NG > 1) {
    float* mo = reinterpret_cast<float*>(smem);
    if (grp > 0) {
      float* dst = mo + ((size_t)((grp - 1) * NWG + wv) * REGS) * 64 + lane;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(d * 16 + r) * 64] = oacc[d][r];
      dst[(DB * 16) * 64] = m_run;
      dst[(DB * 16 + 1) * 64] = l_run;
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 1; g < NG; ++g) {
      const float* src = mo + ((size_t)((g - 1) * NWG + wv) * REGS) * 64 + lane;
      const float m_o = src[(DB * 16) * 64], l_o = src[(DB * 16 + 1) * 64];
      const float mm = fmaxf(m_run, m_o);
      const float mu = mm == -INFINITY ? 0.f : mm;
      const float a_me = __builtin_amdgcn_exp2f(m_run - mu), a_o = __builtin_amdgcn_exp2f(m_o - mu);
      l_run = l_run * a_me + l_o * a_o;
      m_run = mm;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = oacc[d][r] * a_me + src[(d * 16 + r) * 64] * a_o;
    }
  }

  // ---- normalise; O through LDS so that a wave instruction stores whole rows ----
  if (p.lse && hi == 0 && qi < p.Tq) p.lse[((size_t)b * p.H + h) * p.Tq + qi] = m_run + __log2f(l_run);
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
  constexpr int ORS = D * 2 + 16;
  char* olds = smem + MERGE_BYTES + (size_t)wv * 32 * ORS;
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint2v w = {pack_bf16x2(oacc[d][g * 4] * inv, oacc[d][g * 4 + 1] * inv),
                        pack_bf16x2(oacc[d][g * 4 + 2] * inv, oacc[d][g * 4 + 3] * inv)};
      *reinterpret_cast<uint2v*>(olds + ql * ORS + (d * 32 + g * 8 + 4 * hi) * 2) = w;
    }
  __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): same wave wrote and reads, in-order LDS
  constexpr int LPR = D / 8;
  constexpr int RPI = 64 / LPR;
#pragma unroll
  for (int it = 0; it < 32 / RPI; ++it) {
    const int r = it * RPI + lane / LPR;
    const int qrow = qw0 + r;
    const uint4v w = *reinterpret_cast<const uint4v*>(olds + r * ORS + (lane % LPR) * 16);
    if (qrow < p.Tq)
      *reinterpret_cast<uint4v*>(p.O + (size_t)b * p.o_batch + (size_t)qrow * p.o_row + (size_t)h * D + (lane % LPR) * 8) = w;
  }
}

template <int D, int NWG, int NG>
int launch_attn2p(const Attn2Args& a, int B, hipStream_t st) {
  constexpr int LDS = NG * 2 * a2_tile_bytes<D>();
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert(LDS >= (NG - 1) * NWG * (D / 32 * 16 + 2) * 64 * 4 + NWG * 32 * (D * 2 + 16),
                "merge area + O staging fit the tile buffers");
  auto kfn = flash_attn_fwd2p_kernel<D, NWG, NG>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return g4r_note_hip_error(e, "flash_attn_fwd2p: hipFuncSetAttribute");
    attr_set = true;
  }
  dim3 grid(g4r_ceil_div(a.Tq, NWG * 32), a.H, B);
  hipLaunchKernelGGL(kfn, grid, dim3(NWG * NG * 64), LDS, st, a);
  return G4R_OK;
}


// dispatch arm (g4r_attn2_dispatch):
/*
  if (variant >= 200 && variant < 300) {     // pipelined form: 200 + NWG * 10 + NG
    if (head_dim == 128 && variant == 242) rc = launch_attn2p<128, 4, 2>(a, B, st);
    else if (head_dim == 64 && variant == 224) rc = launch_attn2p<64, 2, 4>(a, B, st);
    else if (head_dim == 64 && variant == 242) rc = launch_attn2p<64, 4, 2>(a, B, st);
    ...
  }
*/
