// resolve2.h -- the claim replay over sorted candidate lists (ORBmatcher::SearchByProjection's sequential "first come" semantics,
// ORBmatcher.cc:44-122 and :1318-1452) as a device function, shared by k_resolve2 (matcher.hip) and k_resolve_pose (ba.hip).
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

#include "ctx.h"

namespace {
constexpr float TH_HIGH = 1.5f, TH_LOW = 0.5f;  // ORBmatcher.cc:37-38
constexpr int HISTO = ASD_HISTO_LENGTH;
constexpr int kTop = 4;                          // sorted lists: entries per query in the compact head table

// ---- claim replay over SORTED lists (round 4; the default) -------------------------------------------------------------
// Same recurrence, same fixed-point iteration, same claim tables as k_resolve above -- what changed is how a map point finds its
// pick inside an iteration.  k_resolve keeps all ~12 k candidates in registers and lets every one of them bid for its query with a
// 64-bit LDS atomicMin in EVERY iteration (24 candidates per thread, 13 iterations: 90 us on one workgroup).  k_window_search<true>
// hands the lists over in preference order, so "the best candidate no earlier map point holds" is the first entry of the list whose
// keypoint carries no such claim, and the second best (KIND 1) the next one.
// KIND 0 (best only): "keypoint j is held against q" can only become true and never false again from one iteration to the next --
// the smallest claimant of j finds everything in front of j in its list still held and j still free, so it picks j again -- hence a
// map point's position in its list only ever advances.  A thread keeps (position, keypoint) of each of its queries; an iteration
// reads that keypoint's claim, steps forward past held entries if it has to (each list entry is stepped over at most once in the
// whole kernel: ~700 steps per frame instead of 12 k bids per iteration), posts the claim.
// KIND 1 (best / second best with the ratio test): a pick can be withdrawn (the second best changing its level), so every iteration
// walks from the head -- the four heads' keypoints are in registers, their claims come in one LDS round trip, and the distances /
// levels of the two survivors in a second one for all of the thread's queries together.
// The LDS round trips are what an iteration costs (the workgroup shares its CU with ASDNet workgroups that keep the LDS queues
// full), so every phase issues its reads for all of the thread's queries before it uses any.
struct Resolve2Args {
  int nq, n_cur;
  const int* q_off; const int* q_cnt; const int* idx; const float* dist;   // k_window_search<true>: sorted lists (q_cnt = the head that can be picked)
  const uint16_t* top_idx;                                                  // [nq][kTop] the lists' heads (keypoints; 0xffff = none)
  const int* total; int cap;
  const uint8_t* obs_pos;     // [nq] or null
  const float4* kp_cur;       // (x, y, octave bits, angle)
  const float4* kp_last;      // KIND 0: query q = last-frame keypoint q
  int check_ori;
  float nn_ratio;
  int* match_cur;             // out [n_cur]
  int* n_matches;             // out: [0] n_matches, [1] total candidates, [2] iterations, [3..6] stamps
  int* mirror;                // null, or pinned host memory laid out like match_cur | n_matches
  int stage_cap;              // list entries [0, stage_cap) are copied into LDS (2 B each, KIND 1: 6 B) for the walks beyond the heads
};
__host__ __device__ inline size_t resolve2_fixed_lds(int kind, int n_cur) {   // claim tables + angle table (KIND 0) / octave table (KIND 1)
  return (size_t)n_cur * 8 + (kind == 0 ? (size_t)n_cur * 4 : ((size_t)n_cur + 15) / 16 * 16);
}
constexpr int kResolve2Threads = 1024;
constexpr int kResolve2MaxRounds = 64;   // KIND 1: up to 64 rounds of NT map points are compacted (32768 on the solver's 512 threads, 65535 on 1024)
// The replay as a device function over NT threads of ONE workgroup (dynamic LDS from offset 0): k_resolve2 (matcher.hip) is it on 1024
// threads; k_resolve_pose (ba.hip) runs it on PoseOptimization's 512 in front of the solver, in the same workgroup -- the solver then
// needs no dispatch of its own (it waited 35-50 us for a CU beside the extractor's ASDNet workgroups, twice per frame).
// (A = const Resolve2Args in whatever address space the block lives: a kernel's argument segment, or constant memory)
template <int KIND, int QPT, int NT, class A>
__device__ __forceinline__ void resolve2_body(A& a) {
#pragma clang fp contract(off)
#define OUT(i, v) do { const int v_ = (v); a.match_cur[i] = v_; if (a.mirror) a.mirror[i] = v_; } while (0)
#define CNT(i, v) do { const int v_ = (v); a.n_matches[i] = v_; if (a.mirror) a.mirror[a.n_cur + (i)] = v_; } while (0)
  __builtin_amdgcn_s_setprio(3);   // the tracking thread's critical path, on a CU it shares with ASDNet workgroups: issue its waves first
  extern __shared__ unsigned lds_c[];
  unsigned* claim0 = lds_c;
  // the two claim tables are lds_c[0 .. n_cur) and lds_c[n_cur .. 2 n_cur), always addressed as lds_c[offset + j]: a table POINTER picked
  // per iteration loses its address space, and the compiler then reads the claims with flat loads (several times an LDS read's latency)
  float* ang = reinterpret_cast<float*>(claim0 + 2 * a.n_cur);                       // KIND 0: the current frame's keypoint angles
  uint8_t* octv = reinterpret_cast<uint8_t*>(claim0 + 2 * a.n_cur);                  // KIND 1: their octaves
  char* tail = reinterpret_cast<char*>(lds_c) + resolve2_fixed_lds(KIND, a.n_cur);
  float* sdist = reinterpret_cast<float*>(tail);                                     // KIND 1: [stage_cap]
  uint16_t* sidx = reinterpret_cast<uint16_t*>(tail + (KIND == 1 ? (size_t)a.stage_cap * 4 : 0));   // [stage_cap]
  __shared__ int n_written, hist[HISTO], n_removed, flag[3];
  int* last = reinterpret_cast<int*>(tail + (size_t)a.stage_cap * (KIND == 1 ? 6 : 2));   // [n_cur] the last writer of every keypoint
  const int t = threadIdx.x;
  const unsigned long long ts0 = __builtin_amdgcn_s_memrealtime();
  const int total = *a.total;
  if (total > a.cap || total == 0) {   // truncated lists (the host grows the buffers and searches again) / nothing in any window:
    // the match table is still written (no match anywhere) -- a fused chain behind this kernel gathers its edges from it
    for (int j = t; j < a.n_cur; j += NT) OUT(j, -1);
    if (t == 0) { CNT(0, 0); CNT(1, total); CNT(2, 0); }
    return;
  }
  // the thread's queries first (their loads are then in flight under the LDS fills below): list length and start, the heads
  int cnt[QPT], pick[QPT], qoff[QPT];
  constexpr int HD = KIND == 1 ? 2 : kTop;   // heads kept in registers (KIND 1: the lists are short and two settle nearly every query)
  uint2 tj[QPT];                        // the heads' keypoints (16 bit each; KIND 1 uses .x only)
  float ang_last[KIND == 0 ? QPT : 1];
  unsigned posmask = 0;
  // KIND 1: most local-map candidates have nothing in their window (bench: 1000 lists for 4000 candidates) -- the replay runs over the
  // COMPACTED list of map points that have one, in their original order (ballot + prefix over (slot, wave), as the solver's edge gather
  // does): qid[] = the map point a thread's slot stands for, and the slots beyond n_act -- whole rounds of k, for every thread -- are
  // skipped by the loops below instead of being executed under an empty mask (8 queries per thread on the solver's 512 threads).
  int qid[QPT];
  int ka = QPT;   // slots [0, ka) hold map points somewhere in the workgroup (wave-uniform)
  int n_act = a.nq;   // KIND 1: map points that have a candidate list; they are replayed in CHUNKS of NT * QPT (below)
  unsigned short* act_s = reinterpret_cast<unsigned short*>(last + a.n_cur);   // KIND 1: [nq], behind the last-writer table (resolve_lds_bytes counts it)
  if constexpr (KIND == 1) {
    // ballot + prefix over (round, wave) in the map points' original order, as the solver's edge gather does -- over as many rounds of NT map
    // points as nq needs (round 5: the local map may hold far more points than one chunk of the replay; Tracking.cc:881-905 collects every
    // point of up to 80 keyframes), two passes so that no per-round state stays in registers
    constexpr int NW = NT / 64;
    __shared__ int wbase[kResolve2MaxRounds * NW + 1];
    const int lane = t & 63, wave = t >> 6;
    const int rounds = (a.nq + NT - 1) / NT;   // <= kResolve2MaxRounds (the host checks)
    for (int kb = 0; kb < rounds; ++kb) {
      const int q = t + kb * NT;
      const bool actv = q < a.nq && a.q_cnt[min(q, a.nq - 1)] > 0;
      const unsigned long long bal = __ballot(actv);
      if (lane == 0) wbase[kb * NW + wave] = __popcll(bal);
    }
    asd_syncthreads();
    if (t < 64) {   // exclusive prefix over rounds * NW counts by one wave: NW consecutive entries per lane, then a wave scan
      int loc[NW];
      int sum = 0;
      for (int base = 0; base < rounds * NW; base += 64 * NW) {
        int mine_ = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) { const int e = base + t * NW + i; loc[i] = e < rounds * NW ? wbase[e] : 0; mine_ += loc[i]; }
        int inc = mine_;
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (t >= off) inc += v; }
        int run = sum + inc - mine_;
#pragma unroll
        for (int i = 0; i < NW; ++i) { const int e = base + t * NW + i; if (e < rounds * NW) wbase[e] = run; run += loc[i]; }
        sum += __shfl(inc, 63);
      }
      if (t == 0) wbase[rounds * NW] = sum;
    }
    asd_syncthreads();
    for (int kb = 0; kb < rounds; ++kb) {
      const int q = t + kb * NT;
      const bool actv = q < a.nq && a.q_cnt[min(q, a.nq - 1)] > 0;
      const unsigned long long bal = __ballot(actv);
      if (actv) act_s[wbase[kb * NW + wave] + __popcll(bal & ((1ull << lane) - 1))] = (unsigned short)q;
    }
    asd_syncthreads();
    n_act = wbase[rounds * NW];
  }
  // the thread's queries of one chunk (c0 = first slot of the chunk): list length and start, the heads
  auto load_chunk = [&](int c0) {
    posmask = 0;
    if constexpr (KIND == 1) {
      ka = __builtin_amdgcn_readfirstlane((min(n_act - c0, NT * QPT) + NT - 1) / NT);
#pragma unroll
      for (int k = 0; k < QPT; ++k) {
        const int slot = c0 + t + k * NT;
        const bool v = slot < n_act && k < ka;
        const int q = v ? (int)act_s[slot] : 0;
        qid[k] = q;
        cnt[k] = v ? a.q_cnt[q] : 0;
        qoff[k] = a.q_off[q];
        tj[k] = make_uint2(*reinterpret_cast<const unsigned*>(a.top_idx + (size_t)q * kTop), 0u);
        if (v && (!a.obs_pos || a.obs_pos[q])) posmask |= 1u << k;
        pick[k] = -1;
      }
    } else {
#pragma unroll
      for (int k = 0; k < QPT; ++k) {
        const int q = t + k * NT;
        const bool v = q < a.nq;
        const int qc = v ? q : 0;
        qid[k] = q;
        cnt[k] = v ? a.q_cnt[qc] : 0;
        qoff[k] = a.q_off[qc];
        tj[k] = *reinterpret_cast<const uint2*>(a.top_idx + (size_t)qc * kTop);
        ang_last[k] = a.check_ori ? a.kp_last[qc].w : 0.f;
        if (v && (!a.obs_pos || a.obs_pos[q])) posmask |= 1u << k;
        pick[k] = -1;
      }
    }
  };
  load_chunk(0);   // (its loads are in flight under the LDS fills below)
  for (int j = t; j < 2 * a.n_cur; j += NT) claim0[j] = 0xffffffffu;
  for (int j = t; j < a.n_cur; j += NT) last[j] = -1;
  if (KIND == 0) { if (a.check_ori) for (int j = t; j < a.n_cur; j += NT) ang[j] = a.kp_cur[j].w; }
  else for (int j = t; j < a.n_cur; j += NT) octv[j] = (uint8_t)(__float_as_int(a.kp_cur[j].z) & 0xff);
  const int n_stage = min(total, a.stage_cap);
  for (int i = t; i < n_stage; i += NT) { sidx[i] = (uint16_t)a.idx[i]; if (KIND == 1) sdist[i] = a.dist[i]; }
  if (t == 0) { n_written = 0; n_removed = 0; flag[0] = 0; flag[1] = 0; flag[2] = 0; }
  if (t < HISTO) hist[t] = 0;
  auto top_j = [&](int k, int i) -> unsigned { return ((i < 2 ? tj[k].x : tj[k].y) >> (16 * (i & 1))) & 0xffffu; };
  auto list_j = [&](int k, int i) -> int { const int pos = qoff[k] + i; return pos < n_stage ? (int)sidx[pos] : a.idx[pos]; };
  int ptr[KIND == 0 ? QPT : 1], curj[KIND == 0 ? QPT : 1];   // KIND 0: position in the list and the keypoint there (-1: list exhausted)
  if (KIND == 0) {
#pragma unroll
    for (int k = 0; k < QPT; ++k) { ptr[k] = 0; curj[k] = cnt[k] > 0 ? (int)top_j(k, 0) : -1; }
  }
  asd_syncthreads();
  const unsigned long long ts1 = __builtin_amdgcn_s_memrealtime();
  int it = 0, f_cur = 0;   // f_cur = it % 3
  int it_limit = (KIND == 1 ? min(n_act, NT * QPT) : a.nq) + 2;
  unsigned long long ts_it0 = ts1;
  // the claims an iteration starts from are read at the END of the one before, in the same LDS round trip as its "anything changed" flag
  // (a dependent chain of LDS round trips is what an iteration costs beside ASDNet workgroups that keep the CU's LDS queues full)
  constexpr int NCL = HD;   // (KIND 1 only)
  unsigned cl[QPT][NCL];
  auto load_claims = [&](int rdt) {
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
      if (k >= ka) break;
#pragma unroll
      for (int i = 0; i < NCL; ++i) cl[k][i] = (i < cnt[k]) ? lds_c[rdt + top_j(k, i)] : 0u;
    }
  };
  if (KIND == 0) {   // every map point posts at the head of its list
#pragma unroll
    for (int k = 0; k < QPT; ++k) if (curj[k] >= 0 && (posmask >> k & 1)) atomicMin(&lds_c[curj[k]], (unsigned)(t + k * NT));
  } else load_claims(0);
  int ph_work = 0, ph_bar = 0;   // thread 0's view (10 ns units): loop top -> barrier, barrier -> verdict
  int mine = 0;
  int bin[QPT];
  // KIND 1: the map points with candidates are replayed in chunks of NT * QPT, in index order.  A map point's pick depends on EARLIER map
  // points only, so a chunk's fixed point is final; its claims stay in both tables with tag 0 (atomicMin keeps them: every live tag is
  // larger) and every later chunk sees them as held.  The bench's 1000 lists, and any local map with up to NT * QPT of them, are one chunk.
  for (int c0 = 0;;) {
  for (;; ++it) {
    const unsigned long long p0 = __builtin_amdgcn_s_memrealtime();
    // read the claims of iteration it-1 (table it & 1, tag it), post this iteration's picks into the other table (tag it+1)
    const int rd = (it & 1) ? a.n_cur : 0, wr = a.n_cur - rd;   // offsets into lds_c
    const unsigned tag_rd = (unsigned)(0xffff - it), tag_wr = (unsigned)(0xffff - (it + 1));
    // an earlier map point holds it: a claim of the previous iteration by a smaller index, or (tag 0) the final claim of a map point of an
    // EARLIER CHUNK (KIND 1; all of them have smaller indices)
    auto held = [&](unsigned c, int q) { return ((c >> 16) == tag_rd && (c & 0xffffu) < (unsigned)q) || (KIND == 1 && (c >> 16) == 0u); };
    int changed = 0;
    if (KIND == 0) {
      // Frame-to-frame search keeps the BEST candidate only, so a map point's position in its list only ever moves forward and a keypoint's
      // holder only ever gets replaced by an earlier map point: the replay's fixed point is unique and any order of (post, look, step
      // forward) reaches it.  ONE claim table then (lds_c[0 .. n_cur), plain map point indices, atomicMin, never cleared) and no barrier
      // between looks: a round is kPolls looks at the own claim -- one LDS read and a compare for a map point that still holds its
      // keypoint, a walk and a post for a displaced one -- and the rounds end with the first one in which nobody moved (the table was then
      // static for a whole round and every map point has checked itself against it).  The Jacobi form this replaces paid two LDS round
      // trips, a barrier and a re-post of all 2000 claims per link of the longest displacement chain (13 on the bench stream).
      constexpr int kPolls = 4;
      auto peek = [&](int j) { return __hip_atomic_load(&lds_c[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
      for (int poll = 0; poll < kPolls; ++poll) {
        unsigned holder[QPT];   // all looks of the thread in one round trip
#pragma unroll
        for (int k = 0; k < QPT; ++k) holder[k] = curj[k] >= 0 ? peek(curj[k]) : 0xffffffffu;
#pragma unroll
        for (int k = 0; k < QPT; ++k) {
          const int q = t + k * NT;
          if (!(holder[k] < (unsigned)q)) continue;
          // displaced: step forward, kStep entries per PAIR of LDS round trips (unconditional loads from clamped addresses, selected
          // afterwards -- as conditional loads the compiler serialised them, one round trip per entry)
          constexpr int kStep = 4;
          int j = -1;
          if (qoff[k] + cnt[k] <= n_stage) {   // this list lies inside the LDS copy (a list that took the search's slow path starts at cap + off: never)
            while (j < 0 && ptr[k] + 1 < cnt[k]) {
              int jj[kStep]; unsigned cc[kStep];
              const int p0 = ptr[k] + 1, lastp = cnt[k] - 1;
#pragma unroll
              for (int i = 0; i < kStep; ++i) jj[i] = (int)sidx[qoff[k] + min(p0 + i, lastp)];
#pragma unroll
              for (int i = 0; i < kStep; ++i) cc[i] = peek(jj[i]);
              int adv = min(kStep, lastp - p0 + 1);
#pragma unroll
              for (int i = kStep - 1; i >= 0; --i) if (p0 + i <= lastp && !(cc[i] < (unsigned)q)) { j = jj[i]; adv = i + 1; }
              ptr[k] += adv;
            }
          } else {   // lists beyond the LDS copy (long ones, or sorted copies in the buffers' second half): entry by entry, global beyond it
            while (j < 0 && ptr[k] + 1 < cnt[k]) {
              const int cand = list_j(k, ++ptr[k]);
              if (!(peek(cand) < (unsigned)q)) j = cand;
            }
          }
          curj[k] = j;
          changed = 1;
          if (j >= 0 && (posmask >> k & 1)) atomicMin(&lds_c[j], (unsigned)q);
        }
      }
#pragma unroll
      for (int k = 0; k < QPT; ++k) pick[k] = curj[k];
    } else {
      int p[QPT], p2[QPT], i1[QPT], i2[QPT];
#pragma unroll
      for (int k = 0; k < QPT; ++k) {
        p[k] = -1; p2[k] = -1; i1[k] = 0; i2[k] = 0;
        if (k >= ka) continue;
        const int q = qid[k];
        int found = 0;
#pragma unroll
        for (int i = 0; i < HD; ++i) {
          if (i >= cnt[k] || held(cl[k][i], q) || found >= 2) continue;
          if (found == 0) { p[k] = (int)top_j(k, i); i1[k] = i; }
          else { p2[k] = (int)top_j(k, i); i2[k] = i; }
          ++found;
        }
        if (found < 2 && cnt[k] > HD) {   // the heads did not settle it: on through the list (LDS copy, global beyond it)
          for (int i = HD; i < cnt[k] && found < 2; ++i) {
            const int j = list_j(k, i);
            if (held(lds_c[rd + j], q)) continue;
            if (found == 0) { p[k] = j; i1[k] = i; }
            else { p2[k] = j; i2[k] = i; }
            ++found;
          }
        }
      }
      // ORBmatcher.cc:106-112 (bestDist2 starts at 256, bestLevel2 at -1): distances and levels of the two survivors, all queries together
      float best[QPT], best2[QPT];
      int lvl[QPT], lvl2[QPT];
      // (not `pos < n_stage ? sdist[pos] : a.dist[pos]`: the compiler turns that, and every plain if/else form of it, into ONE flat load
      // through a selected pointer; the empty asm pins the LDS read in front of the branch)
      auto dist_at = [&](int k, int i) { const int pos = qoff[k] + i; float d = sdist[min(pos, n_stage - 1)]; asm volatile("" : "+v"(d)); if (pos >= n_stage) d = a.dist[pos]; return d; };
#pragma unroll
      for (int k = 0; k < QPT; ++k) {
        if (k >= ka) break;
        best[k] = p[k] >= 0 ? dist_at(k, i1[k]) : 0.f;
        best2[k] = p2[k] >= 0 ? dist_at(k, i2[k]) : 256.f;
        lvl[k] = p[k] >= 0 ? (int)octv[p[k]] : -1;
        lvl2[k] = p2[k] >= 0 ? (int)octv[p2[k]] : -1;
      }
#pragma unroll
      for (int k = 0; k < QPT; ++k) {
        if (k >= ka) break;
        const int q = qid[k];
        int pk = p[k];
        if (pk >= 0 && (!(best[k] <= TH_HIGH) || (lvl[k] == lvl2[k] && best[k] > a.nn_ratio * best2[k]))) pk = -1;
        changed |= pk != pick[k];
        pick[k] = pk;
        if (pk >= 0 && (posmask >> k & 1)) atomicMin(&lds_c[wr + pk], (tag_wr << 16) | (unsigned)q);
      }
    }
    // "did any pick change" with ONE barrier (__syncthreads_or is three and a cross-lane reduction): a changed pick sets this
    // iteration's flag word, thread 0 clears the next one's -- last read two barriers ago
    if (changed) flag[f_cur] = 1;
    const int f_next = f_cur == 2 ? 0 : f_cur + 1;
    if (t == 0) flag[f_next] = 0;
    const unsigned long long p1 = __builtin_amdgcn_s_memrealtime();
    asd_syncthreads();
    if (KIND == 1) load_claims(wr);       // the next iteration's claims ...
    const bool more = flag[f_cur] != 0 && it < it_limit;   // ... and this one's verdict: one round trip
    f_cur = f_next;
    { const unsigned long long p2 = __builtin_amdgcn_s_memrealtime(); ph_work += (int)(p1 - p0); ph_bar += (int)(p2 - p1); }
    if (it == 0) ts_it0 = __builtin_amdgcn_s_memrealtime();
    if (!more) break;
  }
  // ---- the chunk's outputs: the last writer of every keypoint, the number of writes, the rotation histogram over all writes
#pragma unroll
  for (int k = 0; k < QPT; ++k) {
    bin[k] = -1;
    if (pick[k] < 0) continue;
    atomicMax(&last[pick[k]], qid[k]);
    ++mine;
    if (KIND == 0 && a.check_ori) {
      float rot = ang_last[k] - ang[pick[k]];   // ORBmatcher.cc:1419-1425
      if (rot < 0.0) rot += 360.0f;
      int b = (int)roundf(rot * (1.0f / HISTO));
      if (b == HISTO) b = 0;
      bin[k] = b;
      atomicAdd(&hist[b], 1);
    }
  }
  if (KIND == 0) break;
  c0 += NT * QPT;
  if (c0 >= n_act) break;
  // the next chunk: this one's claims become permanent (plain stores: the iteration's last barrier lies behind every post), the tags move
  // past the ones still in the tables, the threads take their next map points
#pragma unroll
  for (int k = 0; k < QPT; ++k)
    if (pick[k] >= 0 && (posmask >> k & 1)) { lds_c[pick[k]] = (unsigned)qid[k]; lds_c[a.n_cur + pick[k]] = (unsigned)qid[k]; }
  it += 2;
  it_limit = it + min(n_act - c0, NT * QPT) + 2;
  load_chunk(c0);   // (the flag rotation simply goes on: flag[f_cur] was cleared before the last barrier)
  asd_syncthreads();
  load_claims((it & 1) ? a.n_cur : 0);
  }
  const unsigned long long ts2 = __builtin_amdgcn_s_memrealtime();
  if (mine) atomicAdd(&n_written, mine);
  asd_syncthreads();
  if (KIND == 0 && a.check_ori) {
    // ComputeThreeMaxima (:1584-1625), by every thread for itself (thirty broadcast reads instead of a barrier around thread 0)
    int hs[HISTO];
#pragma unroll
    for (int i = 0; i < HISTO; i++) hs[i] = hist[i];
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
#pragma unroll
    for (int i = 0; i < HISTO; i++) {
      const int s = hs[i];
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    int removed = 0;
#pragma unroll
    for (int k = 0; k < QPT; ++k)   // every write in a discarded bin clears the keypoint and is subtracted (:1437-1450)
      if (bin[k] >= 0 && bin[k] != ind1 && bin[k] != ind2 && bin[k] != ind3) { last[pick[k]] = -1; ++removed; }
    if (removed) atomicAdd(&n_removed, removed);
    asd_syncthreads();
  }
  for (int j = t; j < a.n_cur; j += NT) OUT(j, last[j]);
  if (t == 0) { CNT(0, (KIND == 1 ? 2 : 1) * n_written - n_removed); CNT(1, total); CNT(2, it + 1);
    // 100 MHz stamps (ASD_TIMING): staging, iterations, outputs, the first iteration -- in units of 10 ns
    CNT(3, (int)(ts1 - ts0)); CNT(4, (int)(ts2 - ts1)); CNT(5, (int)(__builtin_amdgcn_s_memrealtime() - ts2)); CNT(6, (int)(ts_it0 - ts1));
    CNT(7, ph_work); CNT(8, ph_bar); CNT(9, (int)(ts0 & 0x7fffffffull)); CNT(10, (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffffull)); }
}
#undef OUT
#undef CNT
}  // namespace
