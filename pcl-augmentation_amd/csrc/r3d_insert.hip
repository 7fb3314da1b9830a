// Level 2, the insert step: one placement candidate of one scene per workgroup
// (insertion.py:455-526), all slots of all scenes of a call in ONE launch.
//
// What a workgroup does for (slot k, scene s) -- "pair" below:
//   sample phase  (reads only the sample and the scene's elevation bounds)
//     project the sample (reference formula, float64), window of the range image around it,
//     occupancy bits, counting sort of the points by (pixel, index) = the order of visible_sample
//     (:474-482), min depth per occupied pixel, 5x3 closing (closing.py:9-23) as word-parallel
//     dilate / erode, list of candidate pixels (where the sample is closed).
//   scene phase   (reads the scene: chunk boxes, alive bits, pixel ids, coordinates)
//     the scene's range image inside the window, min-reduced in LDS from the LIVING points whose
//     64-point chunk touches the window (:118-125); closing; on the candidates: smoothed sample depth
//     < smoothed scene depth (closing.py:26-62, insertion.py:467); count, accept test (:511-517).
//   commit        (writes the scene)
//     append the visible points (:526), clear the alive bit of every scene point in a visible
//     pixel (:470-473), extend chunk boxes / tile counts / far list / log.
//
// The K slots of ONE scene depend on each other, but only where they overlap in the image.  So a
// pair evaluates SPECULATIVELY against the state the scene had when the pair started (the slots
// < p0 that had finished by then), touching nothing; then it waits for slot k-1 of its scene,
// reads what the slots p0..k-1 published (accepted? rows / columns of their visible pixels, a
// rebase, a far pixel) and commits if none of them changed a pixel its evaluation read -- rows
// [rmin-6, rmax+6] x columns [cmin-3, cmax+3] around its sample pixels: candidates lie within 2 rows
// / 1 column of a sample pixel, their hole means look 2 / 1 further, their closing 4 / 2 further.
// Otherwise it evaluates again, now after its predecessors.  The chain of a scene therefore costs one
// evaluation plus K commits instead of K evaluations.
//
// Hand-off between the slots of a scene: NOBODY WAITS (round 5).  A pair whose predecessors are still at work when its
// speculative evaluation is done PARKS it: what the commit needs -- {sample index, pixel} of every visible point in order,
// {chunk, mask} of every chunk that loses points, the boxes the conflict test looks at -- goes to a piece of the launch's
// pool, a header to BatchWs::park_hdr, and the workgroup moves on to its next pair.  The workgroup that finishes slot k - 1
// of the scene finds slot k parked, checks the records of the slots the evaluation did not know for a conflict and commits
// it from the record (or evaluates it again, now after its predecessors), then slot k + 1, and so on.  Who carries on is
// decided by ONE atomic exchange each on the pair's hand-over word (BatchWs::park): the evaluator leaves "parked", the
// workgroup that has published slot k - 1 leaves "predecessor done"; whoever finds the other's mark there continues with
// the pair.  Rounds 2-4 had the evaluator spin on the scene's progress word instead (11-14 % of a pair's time, a whole CU
// idle on large range images), which also tied liveness to the order in which workgroups are dispatched -- a workgroup could
// only wait for lower block ids -- and needed a wall-clock time-out; now any order of dispatch is correct.
// Memory ordering (cdna_hip_programming.md, Guideline 16; correct for any placement of the workgroups): every storing wave
// drains (s_waitcnt vmcnt(0)), barrier, ONE lane does the agent-scope release, waits again, then the relaxed agent-scope
// atomic (progress word, hand-over word); the side that finds the other's mark does ONE agent-scope acquire + wait,
// barrier, scalar cache dropped.  Reads of a speculative evaluation race with the commits of the scene's running
// predecessors by design: those only write points and bits the evaluation either does not look at (indices >= its base
// count) or that lie in pixels whose change is detected afterwards.
//
// Working set: everything lives in the workgroup's LDS (`lds_cap` bytes, a few workgroups per CU).
// A pair whose depth tile, candidate list or chunk list does not fit uses the scene's global
// scratch instead and therefore runs after its predecessors; a pair whose per-point arrays do not
// fit either is left, with the rest of its scene's chain, to k_insert_big (one 1024-thread
// workgroup with all of a CU's LDS per scene, launched behind the chain kernel, idle otherwise).
#include "r3d_insert_core.hpp"

namespace r3d {

// Did one of the slots [j0, k) of this launch change a pixel the evaluation read (or the bounds, or
// the far list)?  bit 0: yes; bit 1: the bounds moved (the sample must be projected again).
// Lane L of every wave looks at slot j0 + L (a launch has at most kMaxChain = 64 slots): one trip to the records
// however many slots there are.  Must be called by whole waves.
__device__ __forceinline__ int conflict_with(const BatchWs &w, int s, int j0, int k, const int *H, int rows, int cols) {
  static_assert(kMaxChain <= 64, "one lane per slot of the launch");
  int out = 0;
  const int j = j0 + (int)(threadIdx.x & 63);
  if (j < k) {
    const int4 *rec = reinterpret_cast<const int4 *>(w.recs + ((int64_t)s * kMaxChain + j) * kRecInts);
    const int4 a = rec[0], c = rec[1];            // flags, n_total, rows lo / hi | columns lo0 hi0 lo1 hi1
    if (a.x & kRecAccepted) {
      if (a.x & kRecRebased) out |= 3;
      if (a.x & kRecFar) out |= 1;
      const int r_lo = H[H_RMIN] - 6, r_hi = H[H_RMAX] + 6;
      if (!(a.w < r_lo || a.z > r_hi)) {
        const int clo[2] = {c.x, c.z}, chi[2] = {c.y, c.w};
        for (int h = 0; h < 2; ++h) {
          if (H[H_CMAX0 + h] < 0) continue;
          int lo = H[H_CMIN0 + h] - 3, hi = H[H_CMAX0 + h] + 3;
          for (int g = 0; g < 2; ++g)
            if (chi[g] >= clo[g] && clo[g] <= hi && chi[g] >= lo) out |= 1;
        }
      }
    }
  }
  return wave_or_i32(out);
}

// What becomes of a pair a workgroup has picked up (run_pair's result): done with it (committed and published, parked,
// left to k_insert_big, ...) | it does not fit this flavour's LDS and nothing has been done for it (the caller tries the
// POOL flavour) | done, and the scene's next slot was found parked: the caller carries on with that one.
enum { kPairDone = 0, kPairAgain = 1, kPairNextParked = 2 };
// How a workgroup comes to a pair: off the queue / by its block id | as the workgroup that has just finished the scene's
// previous slot and found this one parked | the same, the parked record is not to be used (the POOL flavour's turn).
enum { kFresh = 0, kTakeover = 1, kTakeoverEvaluate = 2 };

// One (slot k, scene s) pair of the chain kernel.  LAST: there is no further flavour to try, a pair that does not fit sends
// the rest of its scene's chain to k_insert_big.
template <int NT, bool POOL, bool LAST>
__device__ __forceinline__ int run_pair(const r3d_batch_t &b, const ChainSlots &slots, int nk, int first_step, const BatchWs &w,
                                        int chunks, int lds_cap, unsigned char *smem, const int k, const int s, const int mode) {
  if (s >= b.B) return kPairDone;
  const int tid = threadIdx.x, slot_no = k;
  (void)slot_no;
  int *H = reinterpret_cast<int *>(smem);
  typedef Ins<NT, POOL> InsT;
  InsT I(b, w, smem, lds_cap, s, chunks, k, false);
  const int64_t pair_at = (int64_t)s * kMaxChain + k;

  // how many slots of the scene are done (negative: the chain was given up / left to k_insert_big); never waits.  ONE
  // lane reads relaxed, then ONE agent-scope acquire; the scalar cache is dropped as well (counters travel through it)
  auto progress_now = [&]() -> int {
    __syncthreads();
    if (tid == 0) {
      const int seen = __hip_atomic_load(&w.chain_progress[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (seen > 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      H[H_GO] = seen;
    }
    __syncthreads();
    const int seen = uni(H[H_GO]);
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    return seen;
  };
  // Slot k is done: its record, the scene's progress word, and the hand-over word of slot k + 1.  Returns 1 when that slot
  // was found parked -- this workgroup carries on with it (its evaluator has gone on to other pairs).
  auto publish = [&](int flags, int n_total_after) -> int {
    // every storing wave drains its stores, the workgroup meets, ONE lane releases (then waits again:
    // the order fence -> wait -> flag matters) and stores the progress word relaxed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      int nxt = 0;
      int *rec = w.recs + pair_at * kRecInts;
      rec[REC_FLAGS] = flags;
      rec[REC_NTOTAL] = n_total_after;
      rec[REC_RLO] = H[H_VRMIN];
      rec[REC_RHI] = H[H_VRMAX];
      rec[REC_CLO0] = H[H_VCMIN0];
      rec[REC_CHI0] = H[H_VCMAX0];
      rec[REC_CLO1] = H[H_VCMIN1];
      rec[REC_CHI1] = H[H_VCMAX1];
      // (diagnostic bit 16: slot 0 of scene 0 never publishes -- its successors stay parked, k_insert_big flags the scene)
      if (!((b.reserved & kDbgDropPublish) && k == 0 && s == 0 && nk > 1)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // k -> k + 1 and nothing else: a negative mark (the chain left to k_insert_big) stays
        int expect = k;
        __hip_atomic_compare_exchange_strong(&w.chain_progress[s], &expect, k + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
        if (k + 1 < nk) {
          const int old = __hip_atomic_exchange(&w.park[pair_at + 1], kParkPredDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (old == kParkParked) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            nxt = 1;
          }
        }
      }
      H[H_GO] = nxt;
    }
    __syncthreads();
    const int nxt = uni(H[H_GO]);
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    return nxt;
  };
  // The predecessors are still at work: leave the pair -- with a record of its evaluation (kind kParkEvaluated: `off`,
  // `n_kill`, the header cells) or without (kParkRaw) -- to the workgroup that finishes slot k - 1.  true: parked, this
  // workgroup is done with the pair; false: slot k - 1 has been published meanwhile, carry on as after a wait.
  auto park_try = [&](int kind, long long off, int n_kill, int base_slots, unsigned long long sig, int verify) -> bool {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the record's stores, every wave
    __syncthreads();
    if (tid == 0) {
      int *hdr = w.park_hdr + pair_at * kParkInts;
      hdr[PK_KIND] = kind;
      if (kind == kParkEvaluated) {
        hdr[PK_P0] = base_slots;
        hdr[PK_NVALID] = I.nvalid;
        hdr[PK_NVIS] = I.nvis;
        hdr[PK_ACCEPT] = I.accept ? 1 : 0;
        hdr[PK_REBASE] = H[H_REBASE];
        hdr[PK_FLAGS] = H[H_FLAGS];
        hdr[PK_NKILL] = n_kill;
        hdr[PK_OFF_LO] = (int)(uint32_t)off;
        hdr[PK_OFF_HI] = (int)(uint32_t)((unsigned long long)off >> 32);
        hdr[PK_RMIN] = H[H_RMIN], hdr[PK_RMAX] = H[H_RMAX];
        hdr[PK_CMIN0] = H[H_CMIN0], hdr[PK_CMIN1] = H[H_CMIN1], hdr[PK_CMAX0] = H[H_CMAX0], hdr[PK_CMAX1] = H[H_CMAX1];
        hdr[PK_VRMIN] = H[H_VRMIN], hdr[PK_VRMAX] = H[H_VRMAX];
        hdr[PK_VCMIN0] = H[H_VCMIN0], hdr[PK_VCMIN1] = H[H_VCMIN1], hdr[PK_VCMAX0] = H[H_VCMAX0], hdr[PK_VCMAX1] = H[H_VCMAX1];
        hdr[PK_SIG_LO] = (int)(uint32_t)sig, hdr[PK_SIG_HI] = (int)(uint32_t)(sig >> 32);
        hdr[PK_VERIFY] = verify;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int old = __hip_atomic_exchange(&w.park[pair_at], kParkParked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == kParkPredDone) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else if (b.reserved & kDbgCount) {
        atomicAdd(&w.dbg[kind == kParkEvaluated ? D_PARKED : D_PARKED_RAW], 1);
      }
      H[H_GO] = old;
    }
    __syncthreads();
    const int old = uni(H[H_GO]);
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    return old != kParkPredDone;
  };
  auto outputs = [&](int nv, int acc) {
    if (tid == 0) {
      slots.n_visible[k][s] = nv;
      slots.accepted[k][s] = acc;
    }
  };
  auto bounds_moved = [&](int j0, int j1) -> bool {          // did a slot of [j0, j1) re-base the scene?
    bool moved = false;
    for (int j = j0; j < j1; ++j)
      if ((w.recs[((int64_t)s * kMaxChain + j) * kRecInts + REC_FLAGS] & (kRecAccepted | kRecRebased)) == (kRecAccepted | kRecRebased))
        moved = true;
    return moved;
  };

  const int n_head0 = uni(b.n_head[s]);                     // fixed for the launch
  // the scene's point count after its first j slots (what slot j - 1 published)
  auto count_after = [&](int j) -> int {
    return j > 0 ? uni(w.recs[((int64_t)s * kMaxChain + j - 1) * kRecInts + REC_NTOTAL]) : uni(w.n_total0[s]);
  };
  const bool on = load_slot(I, b, slots, k, s, first_step);
  // (diagnostic bit 2: never evaluate ahead of the predecessors -- the pair is left, unevaluated, to whoever finishes them)
  const bool speculate = !(b.reserved & kDbgSerial);
  bool need_sample = true, sample_ok = false, speculative = false, verified = false, committed = false, rebase = false;
  unsigned long long sig0 = 0ull;
  int rc = kOk, attempts = 0, flags = 0, n_after = 0;

  // 1. where does the scene stand?  p0 slots are done: the evaluation builds on them
  int p0 = k;
  bool waited = true;
  if (mode == kFresh) {
    p0 = k > 0 ? progress_now() : 0;
    if (p0 == kProgDeferred) return kPairDone;              // k_insert_big does this slot
    if (p0 < 0) {
      outputs(0, 0);
      return kPairDone;
    }
    waited = p0 >= k;
    if (!waited && (!speculate || !on)) {
      if (park_try(kParkRaw, 0, 0, 0, 0ull, 0)) return kPairDone;
      waited = true;
      p0 = k;
    }
  } else if (mode == kTakeover) {
    // This workgroup has just published slot k - 1 and found slot k parked.  A record: did one of the slots its evaluation did
    // not know (p0 .. k - 1) change a pixel it read, the bounds, the far list?  No: commit it from the record.
    const int *hdr = w.park_hdr + pair_at * kParkInts;
    if (uni(hdr[PK_KIND]) == kParkEvaluated) {
      const int hp0 = uni(hdr[PK_P0]), hnvalid = uni(hdr[PK_NVALID]), verify = uni(hdr[PK_VERIFY]);
      __syncthreads();
      if (tid == 0) {
        H[H_RMIN] = hdr[PK_RMIN], H[H_RMAX] = hdr[PK_RMAX];
        H[H_CMIN0] = hdr[PK_CMIN0], H[H_CMIN1] = hdr[PK_CMIN1], H[H_CMAX0] = hdr[PK_CMAX0], H[H_CMAX1] = hdr[PK_CMAX1];
        H[H_VRMIN] = hdr[PK_VRMIN], H[H_VRMAX] = hdr[PK_VRMAX];
        H[H_VCMIN0] = hdr[PK_VCMIN0], H[H_VCMIN1] = hdr[PK_VCMIN1], H[H_VCMAX0] = hdr[PK_VCMAX0], H[H_VCMAX1] = hdr[PK_VCMAX1];
        H[H_REBASE] = hdr[PK_REBASE];
        H[H_FLAGS] = hdr[PK_FLAGS];
        H[H_FARADD] = 0;
        H[H_SFAR] = 0;
      }
      __syncthreads();
      int cf = 0;
      if (hnvalid > 0) cf = conflict_with(w, s, hp0, k, H, b.rows, b.cols);
      else if (bounds_moved(hp0, k)) cf = 3;                 // nothing projected: only new bounds could change that
      if (verify) {
        sig0 = (unsigned long long)(uint32_t)uni(hdr[PK_SIG_LO]) | ((unsigned long long)(uint32_t)uni(hdr[PK_SIG_HI]) << 32);
        verified = !cf;                                      // (a detected conflict: nothing to compare, the evaluation is redone anyway)
        if (tid == 0 && !cf) atomicAdd(&w.dbg[D_VERIFY_RUNS], 1);
      }
      if (!cf && !verify) {
        I.nvalid = hnvalid;
        I.nvis = uni(hdr[PK_NVIS]);
        I.accept = uni(hdr[PK_ACCEPT]) != 0;
        I.n_far = 0;
        const long long off = (long long)((unsigned long long)(uint32_t)uni(hdr[PK_OFF_LO]) | ((unsigned long long)(uint32_t)uni(hdr[PK_OFF_HI]) << 32));
        typename InsT::FromRecord src;
        src.v = reinterpret_cast<const uint2 *>(w.tile_pool + off);
        src.kl = reinterpret_cast<const ulonglong2 *>(w.tile_pool + off + InsT::park_vis_bytes(I.nvis));
        src.nkill = uni(hdr[PK_NKILL]);
        rebase = I.commit(src, flags, n_after, count_after(k), n_head0);
        outputs(I.nvis, I.accept ? 1 : 0);
        committed = true;
        if (tid == 0 && (b.reserved & kDbgCount)) atomicAdd(&w.dbg[D_TAKEOVER_COMMIT], 1);
      } else if (cf) {
        attempts = 1;                                          // (the first evaluation was somebody else's)
      }
    }
  }

  if (!committed) {
    int n_base = count_after(p0);
    if (on) {
      for (;;) {
        rc = kOk;
        ++attempts;
        if (need_sample) {
          rc = (b.reserved & kDbgDefer) ? kNoFit : I.sample_phase();
#ifdef R3D_EXP_SAMPLE_TWICE                                    // (how much of a pair's sample phase shows in the launch?)
          if (rc == kOk) rc = I.sample_phase();
#endif
          sample_ok = rc == kOk;
          // The scene may have moved on while the sample was prepared: build on the freshest state, so that fewer
          // slots remain that can invalidate the evaluation (a big pair that has to evaluate twice is the tail
          // of the launch).  New bounds in between: the sample is projected again.
          if (rc == kOk && !waited) {
            const int p1 = progress_now();
            if (p1 == kProgDeferred) return kPairDone;
            if (p1 < 0) {
              outputs(0, 0);
              return kPairDone;
            }
            if (p1 > p0) {
              const bool moved = bounds_moved(p0, p1);
              p0 = p1;
              n_base = count_after(p0);
              waited = p0 >= k;
              if (moved) continue;
            }
          }
        }
        // while it speculates, the evaluation looks twice whether a slot that finished meanwhile has already
        // invalidated it: a doomed evaluation of a big pair is given up early and restarted on the fresher state
        int gone = 0, stale_cf = 0;
        speculative = !waited;
        if (rc == kOk)
          rc = I.scene_phase(n_base, waited, [&]() -> bool {
            const int p1 = progress_now();
            if (p1 < 0) {
              gone = p1;
              return true;
            }
            if (p1 > p0) {
              stale_cf = I.nvalid > 0 ? conflict_with(w, s, p0, p1, H, b.rows, b.cols) : (bounds_moved(p0, p1) ? 3 : 0);
              p0 = p1;                                         // slots below p1 are accounted for from here on
              if (stale_cf) return true;
            }
            return false;
          });
        if (rc == kStale) {
          if (gone == kProgDeferred) return kPairDone;
          if (gone < 0) {
            outputs(0, 0);
            return kPairDone;
          }
          n_base = count_after(p0);
          waited = p0 >= k;
          need_sample = (stale_cf & 2) != 0;
          continue;
        }
        if (rc == kNoFit) break;
        if (!waited) {
          const int seen = progress_now();
          if (seen == kProgDeferred) return kPairDone;
          if (seen < 0) {
            outputs(0, 0);
            return kPairDone;
          }
          if (seen < k) {
            // The predecessors are still at work: nobody waits for them.  The evaluation is left to the workgroup that
            // finishes slot k - 1 -- as a record when a commit can be made from one (no far pixels in play: their pass
            // over the cloud reads this workgroup's LDS), else the pair as it came.
            int kind = kParkEvaluated, n_kill = 0, verify = 0;
            long long off = 0;
            unsigned long long sig = 0ull;
            if (rc == kNeedSerial || I.sample_far()) {
              kind = kParkRaw;
            } else {
              if (b.reserved & kDbgVerify) {
                sig = I.signature();
                verify = 1;
              }
              off = I.park_write(n_kill);
              if (off < 0) kind = kParkRaw;
            }
            if (park_try(kind, off, n_kill, p0, sig, verify)) return kPairDone;
          }
          waited = true;
          STAMP(13);
          int cf = rc == kNeedSerial ? 1 : 0;
          if (sample_ok && I.nvalid > 0) cf |= conflict_with(w, s, p0, k, H, b.rows, b.cols);
          else if (bounds_moved(p0, k)) cf |= 3;               // nothing projected: only new bounds could change that
          if (cf) {
            n_base = count_after(k);
            need_sample = !sample_ok || (cf & 2) != 0;
            p0 = k;
            continue;
          }
        }
        // Diagnostic bit 64: an evaluation that ran ahead of its predecessors and is about to be committed is done again,
        // now after them, and the two are compared -- the invariant of the speculation (no detected conflict => the same
        // visible points and the same culled points).  The later one is what is committed.  (A parked evaluation: its
        // digest travels in the record, the workgroup that takes the pair over evaluates and compares.)
        if ((b.reserved & kDbgVerify) && rc == kOk) {
          if (verified) {
            const unsigned long long sig1 = I.signature();
            if (tid == 0 && sig0 != sig1) atomicAdd(&w.dbg[D_VERIFY_MISMATCH], 1);
          } else if (speculative) {
            sig0 = I.signature();
            verified = true;
            if (tid == 0) atomicAdd(&w.dbg[D_VERIFY_RUNS], 1);
            n_base = count_after(k);
            need_sample = false;
            p0 = k;
            --attempts;
            continue;
          }
        }
        break;
      }
    }
    if (on && rc == kNoFit) {
      if (!LAST) return kPairAgain;                           // once more with the scratch images in the pool
      // the rest of this scene's chain goes to k_insert_big; the predecessors must be done first, so that nobody
      // overwrites the mark: if they are not, whoever finishes them comes back to this pair and leaves the mark
      if (!waited) {
        const int seen = progress_now();
        if (seen == kProgDeferred) return kPairDone;
        if (seen < 0) {
          outputs(0, 0);
          return kPairDone;
        }
        if (seen < k && park_try(kParkRaw, 0, 0, 0, 0ull, 0)) return kPairDone;
      }
      if (tid == 0) {
        w.defer_from[s] = k;
        atomicAdd(&w.dbg[D_DEFERRED], 1);
        __hip_atomic_store(&w.chain_progress[s], kProgDeferred, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return kPairDone;
    }
    if (on && attempts > 1 && tid == 0) atomicAdd(&w.dbg[D_EVAL_TWICE], 1);
    // 2. commit
    const int n_now = count_after(k);
    n_after = !on ? n_now : 0;
    if (on) {
      if (tid == 0 && (b.reserved & kDbgCount)) {
        atomicAdd(&w.dbg[D_PAIRS], 1);
        atomicAdd(&w.dbg[D_CHUNKS_LISTED], I.nlist);
      }
      rebase = I.commit(typename InsT::FromLds{I}, flags, n_after, n_now, n_head0);
      outputs(I.nvis, I.accept ? 1 : 0);
      STAMP(14);
    } else {
      outputs(0, 0);
    }
  }
  if (rebase) {
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    phase_sync();
    unsigned long long *s_min = reinterpret_cast<unsigned long long *>(smem + kHdrBytes), *s_max = s_min + NT / 64;
    rebase_scene<NT>(b, w, chunks, s, s_min, s_max);
    if (tid == 0) atomicAdd(&w.dbg[D_REBASE], 1);
  }
  // 3. publish; the scene's next slot may be waiting for exactly this
  int nxt = 0;
  if (nk > 1) nxt = publish(flags, n_after);
  STAMP(15);
#ifdef R3D_STAMPS
  if (tid == 0)
    reinterpret_cast<long long *>(b.out_xyzi + (int64_t)s * b.cap * 4)[k * 32 + 12] =
        (long long)attempts | ((long long)(rc == kOk ? 1 : 0) << 8) | ((long long)I.ww << 16) | ((long long)I.nlist << 32) |
        ((long long)(I.dt.npx >> 4) << 48);
#endif
  return nxt ? kPairNextParked : kPairDone;
}

// Everything the chain kernel is given, as ONE kernel argument.  The kernel reads it through the kernarg segment
// pointer behind an `asm volatile` the compiler cannot see through: the fields are then loaded (scalar loads from the
// constant cache) where a phase uses them.  Passed the plain way the compiler fetches the ~100 scalar registers of
// r3d_batch_t / BatchWs in the prologue and keeps them alive across every phase of the pair: hundreds of SGPR spills
// (v_writelane / v_readlane in the dependent chain of a latency-bound kernel).
struct ChainArgs {
  r3d_batch_t b;
  ChainSlots slots;
  BatchWs w;
  int nk, first_step, chunks, lds_cap, B8, queue_mode;
};

// Range images of KITTI's size never need the pool for their scratch images; on large ones a window can exceed the
// LDS (a car a few metres from the sensor on 448 x 2880).  The POOL flavour reaches those images through flat
// accesses -- 35 % slower on config C5 when every pair takes it -- so a pair runs it only after the LDS flavour
// has turned it down.
typedef const __attribute__((address_space(4))) ChainArgs *ChainArgsPtr;

// A pair, then the slots of its scene that are found parked behind it, one after the other.  (The arguments are taken
// afresh through the kernarg pointer for every pair: nothing of them stays in registers across pairs.)
template <int NT>
__device__ __forceinline__ void run_pairs_from(unsigned char *smem, int pair_id) {
  int k, s, mode = kFresh;
  {
    ChainArgsPtr ap = (ChainArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ap));
    const ChainArgs &a = *(const ChainArgs *)ap;
    // B8 > 0: slot-major numbering (every scene's first slot, then every scene's second; a scene's slots on one residue of
    // the pair id mod 8)
    const int B8 = a.B8;
    k = pair_id / B8;
    s = pair_id % B8;
  }
  for (;;) {
    ChainArgsPtr ap = (ChainArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ap));
    const ChainArgs &a = *(const ChainArgs *)ap;
#ifdef R3D_STAMPS
    long long *cell = s < a.b.B ? reinterpret_cast<long long *>(a.b.out_xyzi + (int64_t)s * a.b.cap * 4) + k * 32 : nullptr;
    if (cell && threadIdx.x == 0) {
      cell[28] = wall_clock64();                               // this workgroup takes the pair
      if (mode == kFresh) cell[21] = cell[28];                 // ... off the queue (a parked pair is taken a second time)
      cell[16] = (long long)blockIdx.x | ((long long)mode << 32);   // (17 .. 20: the gather loop's accumulators)
    }
#endif
    // (only the shape for large range images carries the POOL flavour: half the code for the others)
    int st = uni(run_pair<NT, false, NT != 1024>(a.b, a.slots, a.nk, a.first_step, a.w, a.chunks, a.lds_cap, smem, k, s, mode));
#ifdef R3D_STAMPS
    if (cell && threadIdx.x == 0) cell[29] = st == kPairAgain ? wall_clock64() : 0;   // the LDS flavour turned the pair down here
#endif
    if (NT == 1024 && st == kPairAgain)
      st = uni(run_pair<NT, true, true>(a.b, a.slots, a.nk, a.first_step, a.w, a.chunks, a.lds_cap, smem, k, s,
                                        mode == kFresh ? kFresh : kTakeoverEvaluate));
#ifdef R3D_STAMPS
    __syncthreads();
    if (cell && threadIdx.x == 0) cell[30] = wall_clock64();   // ... and is done with it
#endif
    if (st != kPairNextParked) return;
    ++k;
    mode = kTakeover;
  }
}

template <int NT, bool QUEUE>
__global__ void __launch_bounds__(NT, NT == 1024 ? R3D_BIG_WAVES : R3D_CHAIN_WAVES)
k_insert_chain(ChainArgs args) {
  extern __shared__ __align__(16) unsigned char smem[];     // no static __shared__ here: one LDS array
  // QUEUE false: one workgroup per pair, pair = block id.  QUEUE true: the launch holds only as many workgroups as the
  // device keeps resident and each takes pairs off a queue until it is empty -- queue_mode 1: one queue, pairs in id
  // order; 2: a queue per XCD (workgroups go to the XCDs round robin, a scene's pairs share a residue mod 8: a scene
  // stays with one XCD's L2), an XCD that runs dry helps the others.  Why: the hardware hands workgroups to the XCDs
  // strictly round robin, so with one workgroup per pair an XCD whose pairs run long stalls the hand-out to all the
  // others (a third of the CUs idle on config C5, tools/stamps_insert.py).  Either way a workgroup never waits for
  // another one (see the top of the file): any order in which the pairs are taken is correct.
  if (!QUEUE) {
    run_pairs_from<NT>(smem, (int)blockIdx.x);
    return;
  }
  int *H = reinterpret_cast<int *>(smem);
  const int queue_mode = args.queue_mode;
  const int total = args.B8 * args.nk;
  const int home = queue_mode >= 2 ? (int)(blockIdx.x & 7) : 0;
  int turn = 0;                                              // queues this workgroup has found empty
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) {
      ChainArgsPtr ap = (ChainArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(ap));
      const ChainArgs &a = *(const ChainArgs *)ap;
      int id = total;
      while (turn < (queue_mode >= 2 ? 8 : 1)) {
        const int q = (home + turn) & 7;
        const int n = atomicAdd(&a.w.queue_next[q], 1);
        id = queue_mode >= 2 ? n * 8 + q : n;
        if (id < total) break;
        id = total;
        ++turn;
      }
      H[H_GO] = id;
      H[H_GO + 1] = turn;
    }
    __syncthreads();
    const int taken = uni(H[H_GO]);
    turn = uni(H[H_GO + 1]);
    if (taken >= total) return;
    run_pairs_from<NT>(smem, taken);
  }
}

// The pairs the chain kernel could not hold in its LDS, scene by scene, slot after slot.
template <int NT>
__global__ void __launch_bounds__(NT)
k_insert_big(r3d_batch_t b, ChainSlots slots, int nk, int first_step, BatchWs w, int chunks, int lds_cap) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int s = (int)blockIdx.x;
  const int tid = threadIdx.x;
  const int k0 = w.defer_from[s];
  // A chain the chain kernel neither finished nor handed over (a slot that never published: diagnostic bit 16; nothing
  // else is known to get there): the scene is flagged, its remaining slots report nothing, the caller redoes the batch
  // slot by slot (SceneBatch.run_inserts)
  if (k0 >= nk) {
    const int done = w.chain_progress[s];
    if (nk > 1 && done >= 0 && done < nk && tid == 0) {
      atomicOr(&b.status[s], R3D_S_CHAIN_TIMEOUT);
      for (int k = done; k < nk; ++k) slots.n_visible[k][s] = 0, slots.accepted[k][s] = 0;
    }
    return;
  }
  for (int k = k0; k < nk; ++k) {
    Ins<NT, true> I(b, w, smem, lds_cap, s, chunks, k, true);
    const bool on = load_slot(I, b, slots, k, s, first_step);
    int nv = 0, acc = 0;
    if (on) {
      int rc = I.sample_phase();
      if (rc == kOk) rc = I.scene_phase(b.n_total[s], true, [] { return false; });
      if (rc != kOk) {
        if (tid == 0) atomicOr(&b.status[s], R3D_S_WINDOW_TOO_LARGE);     // not even a whole CU's LDS holds it
      } else {
        int flags, n_after;
        bool rebase = I.commit(typename Ins<NT, true>::FromLds{I}, flags, n_after);
        nv = I.nvis;
        acc = I.accept ? 1 : 0;
        if (rebase) {
          phase_sync();
          unsigned long long *s_min = reinterpret_cast<unsigned long long *>(smem + kHdrBytes), *s_max = s_min + NT / 64;
          rebase_scene<NT>(b, w, chunks, s, s_min, s_max);
        }
      }
    }
    if (tid == 0) {
      slots.n_visible[k][s] = nv;
      slots.accepted[k][s] = acc;
    }
    phase_sync();                                           // the next slot reads what this one wrote
  }
}

// r3d_batch_insert_first: the candidate loop of ONE insert slot inside a kernel (insertion.py:449-545: the placements of a
// sample are tried in order, the first whose visible part reaches min_points is merged, `break`).  One workgroup per scene
// walks its candidates one after the other -- nothing to order between workgroups --: candidate j of scene s is the sample
// rows [sample_off[s], sample_off[s + 1]) of the packed list at cand + j * cand_stride, what r3d_find_possible_places
// writes.  Rounds 3-5 made this loop on the host: one r3d_batch_insert launch per candidate (three kernels) and a dozen
// tensor operations between them for the "still open" masks -- 8 x 15 launches per slot of a batch in which four scenes of
// five accept their first candidate.  PASS 0: the chain kernel's workgroup shape; a candidate that does not fit its LDS
// leaves the scene, from that candidate on, to PASS 1 (1024 threads, a CU's whole LDS, lists and images in the pool).
struct FirstArgs {
  const double *cand;
  int64_t cand_stride;                  // doubles between candidate j and j + 1 of the packed lists
  const int64_t *sample_off;
  const int32_t *min_points, *active, *n_possible;
  int32_t first_cand, n_cand, step, replay_last;
  int32_t *n_visible, *accepted_at;
};
__global__ void k_first_init(r3d_batch_t b, BatchWs w, int n_cand) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s == 0) *w.pool_head = 0ull;
  if (s < b.B) w.defer_from[s] = n_cand;
}
template <int NT, bool POOL, int PASS>
__global__ void __launch_bounds__(NT, NT == 1024 ? R3D_BIG_WAVES : R3D_CHAIN_WAVES)
k_insert_first(r3d_batch_t b, FirstArgs a, BatchWs w, int chunks, int lds_cap) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int s = (int)blockIdx.x, tid = threadIdx.x;
  if (a.active && !a.active[s]) return;
  const int j0 = PASS == 0 ? 0 : w.defer_from[s];
  const int n_have = a.n_possible[s] - a.first_cand;          // candidates of this window that exist
  const int64_t off = a.sample_off[s], m64 = a.sample_off[s + 1] - off;
  if (m64 <= 0) return;
  if (m64 > kKeyCap) {
    if (tid == 0 && PASS == 0) atomicOr(&b.status[s], R3D_S_SAMPLE_TOO_LARGE);
    return;
  }
  for (int j = j0; j < a.n_cand && j < n_have; ++j) {
    Ins<NT, POOL> I(b, w, smem, lds_cap, s, chunks, 0, PASS == 1);
    I.rows5 = a.cand + (int64_t)j * a.cand_stride + off * 5;
    I.m = uni((int)m64);
    I.need = uni(a.min_points[s]);
    I.cull_only = false;
    I.step = a.step;
    int rc = I.sample_phase();
    if (rc == kOk) rc = I.scene_phase(b.n_total[s], true, [] { return false; });
    if (rc != kOk) {
      if (PASS == 0) {
        if (tid == 0) w.defer_from[s] = j;                    // PASS 1 goes on from here
      } else if (tid == 0) {
        atomicOr(&b.status[s], R3D_S_WINDOW_TOO_LARGE);      // not even a whole CU's LDS holds it
      }
      return;
    }
    int flags, n_after;
    bool rebase = I.commit(typename Ins<NT, POOL>::FromLds{I}, flags, n_after);
    const bool accepted = I.accept;
    if (tid == 0) a.n_visible[s] = I.nvis;
    if (rebase) {
      phase_sync();
      unsigned long long *s_min = reinterpret_cast<unsigned long long *>(smem + kHdrBytes), *s_max = s_min + NT / 64;
      rebase_scene<NT>(b, w, chunks, s, s_min, s_max);
    }
    if (accepted) {
      if (tid == 0) a.accepted_at[s] = a.first_cand + j;
      return;
    }
    // the sample's LAST candidate, rejected: the reference's driver keeps the scene it has culled for it (insertion.py:
    // 468-471 ran, the next candidate's restore :453 does not come): the same candidate once more as min_points < 0, into
    // the batch's shadow (r3d_batch_insert, include/real3daug_hip.h)
    if (a.replay_last && a.first_cand + j + 1 == a.n_possible[s]) {
      I.cull_only = true;
      I.need = -1;
      rc = I.scene_phase(b.n_total[s], true, [] { return false; });
      if (rc == kOk) (void)I.commit(typename Ins<NT, POOL>::FromLds{I}, flags, n_after);
    }
    phase_sync();
  }
}

// Before a chain launch: progress / hand-over words, the pool, the queues.  The launch hands out its slots in the caller's
// order.  Nobody waits for anybody (see the top of the file), so any order would be
// correct; measured on config C2 and dropped: all slots by weight, heaviest first (slot 0 then starts at 100 us and every
// chain queues behind it: 0.42 ms per launch against 0.35), and a heavy slot up to two places earlier (0.43 ms: the car then
// evaluates against a state two slots old, a tenth of the cars conflict with one of them and start over).
__global__ void k_chain_init(r3d_batch_t b, BatchWs w, ChainSlots slots, int nk) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= b.B) return;
  if (s == 0) {
    *w.pool_head = 0ull;
    for (int q = 0; q < 16; ++q) w.queue_next[q] = 0;
  }
  w.chain_progress[s] = 0;
  w.n_total0[s] = b.n_total[s];
  w.defer_from[s] = nk;
  for (int k = 0; k < nk; ++k) w.park[(int64_t)s * kMaxChain + k] = kParkNone;
}

constexpr int kBigNT = 1024;
constexpr int kBigLds = 160 * 1024;

// Shape of a chain workgroup.  Range images of KITTI's size (112 x 1440): two 512-thread workgroups with 80 KB each per
// CU (measured on config C2: 256 threads x 40 KB leaves most cars to the pool / k_insert_big, 1024 x 160 KB leaves
// the CUs to one pair each and no room for the other steps in flight).  Large range images (config C5, 448 x 2880:
// a car's window is ~16x the pixels): one 1024-thread workgroup with the CU's whole LDS -- 2.5 ms per step against
// 5.9 ms, nearly every pair then fits the chain kernel (round 6, with sparse tiles: 512 x 80 KB 12.7 ms, 1024 x 80 KB
// 10.8 ms against 5.0 ms per launch -- the six bit images of a large window alone exceed 80 KB).
static void chain_shape(const r3d_batch_t &b, int &nt, int &lds) {
  const bool large = (int64_t)b.rows * b.cols >= 4ll * R3D_NUMROW * R3D_NUMCOLUMN;
  nt = large ? 1024 : 512;
  lds = (large ? 160 : 80) * 1024;
}

template <int NT, bool QUEUE>
static int launch_chain_q(const r3d_batch_t &b, const BatchWs &w, const ChainSlots &sl, int nk, int first_step, int lds,
                          int B8, int queue_mode, hipStream_t st) {
  // per device, every call: the attribute belongs to the current device's copy of the kernel
  R3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_insert_chain<NT, QUEUE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int total = B8 * nk;
  int grid = total;
  if (QUEUE && queue_mode) {
    // how many workgroups of this shape the device keeps resident: asked once per (device, LDS size) and kernel flavour
    static std::mutex mu;
    static std::map<std::pair<int, int>, int> known;
    int dev = 0;
    R3D_HIP(hipGetDevice(&dev));
    int resident = 0;
    {
      std::lock_guard<std::mutex> lock(mu);
      auto it = known.find({dev, lds});
      if (it != known.end()) resident = it->second;
    }
    if (!resident) {
      int cus = 0, per_cu = 0;
      R3D_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      R3D_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(k_insert_chain<NT, QUEUE>), NT, lds));
      resident = (per_cu < 1 ? 1 : per_cu) * (cus < 8 ? 8 : cus);
      std::lock_guard<std::mutex> lock(mu);
      known[{dev, lds}] = resident;
    }
    grid = total < resident ? total : resident;
  }
  ChainArgs args;
  args.b = b;
  args.slots = sl;
  args.w = w;
  args.nk = nk, args.first_step = first_step, args.chunks = chunks_of(b), args.lds_cap = lds, args.B8 = B8, args.queue_mode = queue_mode;
  hipLaunchKernelGGL((k_insert_chain<NT, QUEUE>), dim3(grid), dim3(NT), lds, st, args);
  R3D_LAUNCHED("k_insert_chain");
  return R3D_OK;
}

template <int NT>
static int launch_chain(const r3d_batch_t &b, const BatchWs &w, const ChainSlots &sl, int nk, int first_step, int lds,
                        hipStream_t st) {
  // a scene's slots on one residue of the pair id mod 8 (slot-major numbering; measured on config C2, round 2: scene-major
  // numbering is 10-40 % slower -- the big pairs of all scenes no longer start together)
  const int B8 = (b.B + 7) & ~7;
  // queue_mode 0: one workgroup per pair; 1: resident workgroups that take pairs off one queue; 2: ... off a queue per XCD
  // (see k_insert_chain).  2 on large range images (config C5: 6.4 -> 4.8 ms per 128 scans x 50 slots), 0 otherwise.
  const bool large = (int64_t)b.rows * b.cols >= 4ll * R3D_NUMROW * R3D_NUMCOLUMN;
  int queue_mode = large ? 2 : 0;
  if (B8 >= (1 << 20) && queue_mode >= 2) queue_mode = 1;
  return queue_mode ? launch_chain_q<NT, true>(b, w, sl, nk, first_step, lds, B8, queue_mode, st)
                    : launch_chain_q<NT, false>(b, w, sl, nk, first_step, lds, B8, 0, st);
}

// One launch of the chain kernel for the slots of `sl`, k_insert_big behind it for what it left (idle otherwise).
static int launch_slots(const r3d_batch_t &b, const BatchWs &w, const ChainSlots &sl, int nk, int first_step,
                        hipStream_t st) {
  int nt, lds;
  chain_shape(b, nt, lds);
  R3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_insert_big<kBigNT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kBigLds));
  hipLaunchKernelGGL(k_chain_init, dim3((b.B + 255) / 256), dim3(256), 0, st, b, w, sl, nk);
  int rc = nt == 1024 ? launch_chain<1024>(b, w, sl, nk, first_step, lds, st) : launch_chain<512>(b, w, sl, nk, first_step, lds, st);
  if (rc != R3D_OK) return rc;
  hipLaunchKernelGGL(k_insert_big<kBigNT>, dim3(b.B), dim3(kBigNT), kBigLds, st, b, sl, nk, first_step, w,
                     chunks_of(b), kBigLds);
  R3D_LAUNCHED("k_insert_big");
  return R3D_OK;
}

template <int NT>
static int launch_first(const r3d_batch_t &b, const BatchWs &w, const FirstArgs &a, int lds, hipStream_t st) {
  R3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_insert_first<NT, false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  R3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_insert_first<kBigNT, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              kBigLds));
  hipLaunchKernelGGL(k_first_init, dim3((b.B + 255) / 256), dim3(256), 0, st, b, w, (int)a.n_cand);
  hipLaunchKernelGGL((k_insert_first<NT, false, 0>), dim3(b.B), dim3(NT), lds, st, b, a, w, chunks_of(b), lds);
  hipLaunchKernelGGL((k_insert_first<kBigNT, true, 1>), dim3(b.B), dim3(kBigNT), kBigLds, st, b, a, w, chunks_of(b), kBigLds);
  R3D_LAUNCHED("k_insert_first");
  return R3D_OK;
}

}  // namespace r3d

using namespace r3d;

extern "C" {

int r3d_batch_insert_first(const r3d_batch_t *b, const double *cand, int64_t cand_stride, const int64_t *sample_off,
                           const int32_t *min_points, const int32_t *active, const int32_t *n_possible, int32_t first_cand,
                           int32_t n_cand, int32_t step, int32_t replay_last, int32_t *n_visible, int32_t *accepted_at,
                           void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (!cand || !sample_off || !min_points || !n_possible || !n_visible || !accepted_at || step < 1 || n_cand < 1 || first_cand < 0 ||
      cand_stride < 0)
    return fail(R3D_E_ARG, "batch_insert_first: null pointer, step < 1, no candidate or a negative offset");
  BatchWs w = carve_batch(*b, b->workspace);
  FirstArgs a{cand, cand_stride, sample_off, min_points, active, n_possible, first_cand, n_cand, step, replay_last, n_visible, accepted_at};
  int nt, lds;
  chain_shape(*b, nt, lds);
  return nt == 1024 ? launch_first<1024>(*b, w, a, lds, (hipStream_t)stream) : launch_first<512>(*b, w, a, lds, (hipStream_t)stream);
}

int r3d_batch_insert(const r3d_batch_t *b, const double *samples5, const int64_t *sample_off,
                     const int32_t *min_points, const int32_t *active, int32_t step,
                     int32_t *n_visible, int32_t *accepted, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (!samples5 || !sample_off || !min_points || !n_visible || !accepted || step < 1)
    return fail(R3D_E_ARG, "batch_insert: null pointer or step < 1");
  BatchWs w = carve_batch(*b, b->workspace);
  ChainSlots sl{};
  sl.samples5[0] = samples5;
  sl.sample_off[0] = sample_off;
  sl.min_points[0] = min_points;
  sl.active[0] = active;
  sl.n_visible[0] = n_visible;
  sl.accepted[0] = accepted;
  return launch_slots(*b, w, sl, 1, (int)step, (hipStream_t)stream);
}

int r3d_batch_insert_many(const r3d_batch_t *b, int32_t n_slots, const double *const *samples5,
                          const int64_t *const *sample_off, const int32_t *const *min_points,
                          const int32_t *const *active, int32_t first_step, int32_t *const *n_visible,
                          int32_t *const *accepted, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (!samples5 || !sample_off || !min_points || !n_visible || !accepted || n_slots < 1 || first_step < 1)
    return fail(R3D_E_ARG, "batch_insert_many: null pointer, no slot or first_step < 1");
  for (int k = 0; k < n_slots; ++k)
    if (!samples5[k] || !sample_off[k] || !min_points[k] || !n_visible[k] || !accepted[k])
      return fail(R3D_E_ARG, "batch_insert_many: null pointer in a slot");
  const bool no_chain = (b->reserved & R3D_B_SLOT_LAUNCHES) != 0;       // one launch per slot (no chain inside a kernel)
  BatchWs w = carve_batch(*b, b->workspace);
  const int per = no_chain ? 1 : kMaxChain;
  for (int k0 = 0; k0 < n_slots; k0 += per) {
    int nk = n_slots - k0 < per ? n_slots - k0 : per;
    ChainSlots sl{};
    for (int k = 0; k < nk; ++k) {
      sl.samples5[k] = samples5[k0 + k];
      sl.sample_off[k] = sample_off[k0 + k];
      sl.min_points[k] = min_points[k0 + k];
      sl.active[k] = active ? active[k0 + k] : nullptr;
      sl.n_visible[k] = n_visible[k0 + k];
      sl.accepted[k] = accepted[k0 + k];
    }
    rc = launch_slots(*b, w, sl, nk, (int)(first_step + k0), (hipStream_t)stream);
    if (rc != R3D_OK) return rc;
  }
  return R3D_OK;
}

int r3d_batch_debug_counters(const r3d_batch_t *b, int32_t *host_out16, int32_t reset, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (!host_out16) return fail(R3D_E_ARG, "batch_debug_counters: null output");
  BatchWs w = carve_batch(*b, b->workspace);
  // (bit 1: the caller's array holds 32, the notes of a diagnostic build too; bit 2: it holds all 64, round 5's counters too)
  const size_t n = (reset & 4) ? kDbgInts : ((reset & 2) ? 32 : 16);
  R3D_HIP(hipMemcpyAsync(host_out16, w.dbg, n * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  if (reset & 1) R3D_HIP(hipMemsetAsync(w.dbg, 0, kDbgInts * sizeof(int32_t), (hipStream_t)stream));
  R3D_HIP(hipStreamSynchronize((hipStream_t)stream));
  return R3D_OK;
}

}  // extern "C"
