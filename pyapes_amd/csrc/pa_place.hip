// pa_place.hip -- WHICH allocations r and the two direction buffers of a large CG solve live in: an ONLINE search
// that rides on the solve's own iterations and pays for itself.
//
// Why.  The arrays of a CG iteration are streamed in lockstep, and whether they collide in the memory system is decided
// by where the driver put their pages: phase B of 512^3 fp64 measures 853 us or 931 us with the SAME kernel from one
// process / box to the next; a plain copy between 1 GiB blocks of one process runs at 4.65 ... 5.51 TB/s by PAIR of
// blocks while every block alone gives 5.5-5.6, and no offset inside a block changes that (docs/HISTORY.md, "Section 8").
// Round 3 asked the hardware in the set-up, with ~86 dry-run iterations on an empty interior set before the first real
// one: a solve shorter than ~1,700 iterations never earned that back.
//
// How, now.  Nothing runs that the solve would not run anyway.  Every real iteration is timed (a HIP event pair per
// iteration on the ctx stream; the host stays at most one iteration ahead of the GPU while a search is on, which costs
// the GPU nothing: it always has a whole iteration queued).  After one clean iteration PAIR (both parities of the
// direction ping-pong) a trial moves ONE role into another block:
//   * a direction buffer exactly when phase A is about to overwrite it -- no copy, the old content is dead; in a block
//     fresh from hipMalloc the few cells the tiled phases never write (the last boundary row / column of a non-periodic
//     axis, pad cells of pitched rows) are zeroed first, O(n^2) of them; a block that already carried r / d in this
//     solve needs nothing at all;
//   * r by having ONE phase B write the new residual into the other block instead of in place (k_cg3d / k_cg2d / k_cg_b
//     take the output pointer separately: no copy either, and moving back costs the same nothing).
// The next iteration pair IS the measurement: accepted if it beats the best pair by 1.5 % (3 % at once, else a second
// pair decides), otherwise undone.  hipMalloc costs ~10 us and does not wait for the stream (profiles/tools/
// mallocbench.hip: 1 GiB 12 us, 2 GiB 0.2 ms; hipFree drains the device, so blocks are only freed where the solve
// synchronises anyway, at its end; on one box of round 4's last session the first 128 MiB block took 3.4 ms -- the budget
// then holds every further trial back until the solves have earned it).  What the trials cost -- the lost time of trial iterations that ran slower, the
// allocations -- is accounted in microseconds against the time the solves of this context have taken so far, and a new
// trial only starts while that share is below `budget` (3 %): a 30-iteration solve pays for at most one or two
// trials, a 1000-iteration solve for the whole pass (3 roles x `blocks` candidates), and nobody pays up front.
// The pass is remembered per context; up to four different x pointers get a pass of their own (the pairing that
// matters most is r with the caller's x), after that the context stops searching.
// Results never depend on any of this: the same kernels read and write the same values at other addresses
// (tests/test_gpu_place.py: every bit equal with the search forced onto small meshes and switched off).
#include "pa_host.h"

#include <chrono>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {

const int SLOT[3] = {SCR_R, SCR_D0, SCR_D1};
const char* const ROLE[3] = {"r", "d0", "d1"};
constexpr size_t MAX_OFF = 3 * 69888;   // (pa_scratch's default stagger: slot q starts (q + 1) * 69888 bytes in)

double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool dbg() {
  static int v = -1;
  if (v < 0) v = getenv("PYAPES_HIP_DEBUG") ? 1 : 0;
  return v != 0;
}

double& dur(PlaceSearch& P, int64_t j) { return P.dur[j & (PA_PLACE_NDUR - 1)]; }

void free_spares(pa_ctx* c) {
  PlaceSearch& P = c->ps;
  for (int k = 0; k < P.nspare; ++k)
    if (P.spare[k]) (void)hipFree(P.spare[k]);
  P.nspare = 0;
}

// pointer swap of role q's block with spare k (the offsets of the roles inside their blocks stay what they are)
void swap_block(pa_ctx* c, int q, int k) {
  PlaceSearch& P = c->ps;
  const int s = SLOT[q];
  const size_t off = (size_t)((char*)c->scr[s] - (char*)c->scr_base[s]);
  char* old = (char*)c->scr_base[s];
  c->scr_base[s] = P.spare[k];
  c->scr[s] = P.spare[k] + off;
  P.spare[k] = old;
  const int e = P.spare_epoch[k];
  P.spare_epoch[k] = P.epoch;   // what leaves a slot carried r / d of this solve: zero wherever the phases do not write
  (void)e;
}

// candidates left in this pass?  advances (role, cand) past exhausted roles; false = the pass is over
bool next_candidate(PlaceSearch& P) {
  while (P.role < 3 && P.cand >= P.blocks) { ++P.role; P.cand = 0; }
  return P.role < 3;
}

void finish_pass(pa_ctx* c) {
  PlaceSearch& P = c->ps;
  P.phase = 2;
  P.st = 0;
  bool have = false;
  for (int q = 0; q < P.n_decided; ++q) have = have || P.decided[q] == P.x;
  if (!have && P.n_decided < PA_PLACE_NX) P.decided[P.n_decided++] = P.x;
  if (dbg())
    fprintf(stderr, "[pyapes_hip] placement search: pass over after %d trials (%d kept, %d allocations): best pair %.1f us, "
            "spent %.0f us of %.0f us solved\n", P.trials, P.accepted, P.mallocs, P.base, P.spent_us, P.elapsed_us);
}

}  // namespace

void pa_place_reset(pa_ctx* c) {
  PlaceSearch& P = c->ps;
  free_spares(c);
  P.phase = 0;
  P.st = 0;
  P.active = 0;
  P.role = P.cand = 0;
  P.n_decided = 0;
  P.bytes = 0;
  P.first_pair = 0.0;
}

void pa_place_destroy(pa_ctx* c) {
  pa_place_reset(c);
  PlaceSearch& P = c->ps;
  for (int q = 0; q < PA_PLACE_NEV; ++q) {
    if (P.ev0[q]) { (void)hipEventDestroy(P.ev0[q]); P.ev0[q] = nullptr; }
    if (P.ev1[q]) { (void)hipEventDestroy(P.ev1[q]); P.ev1[q] = nullptr; }
  }
  for (int q = 0; q < 2; ++q)
    if (P.evs[q]) { (void)hipEventDestroy(P.evs[q]); P.evs[q] = nullptr; }
}

// start of a CG solve on x (pa_cg_begin, after the scratch arrays exist and BEFORE anything is written into them)
int pa_place_begin(pa_ctx* c, const void* x, size_t array_bytes) {
  PlaceSearch& P = c->ps;
  P.active = 0;
  if (!c->place || !c->fastpath || c->profile || c->plan_only) return PA_OK;
  if (array_bytes < P.minbytes) return PA_OK;   // arrays the Infinity Cache holds are not a matter of HBM channels
  const size_t bytes = c->cap[SCR_R];
  if (c->cap[SCR_D0] != bytes || c->cap[SCR_D1] != bytes) return PA_OK;
  for (int q = 0; q < 3; ++q)
    if ((size_t)((char*)c->scr[SLOT[q]] - (char*)c->scr_base[SLOT[q]]) > MAX_OFF) return PA_OK;
  if (P.bytes != bytes) {   // first use, or the arrays were re-allocated at another size: a new pool, a new pass
    pa_place_reset(c);
    P.bytes = bytes;
    P.blk = bytes + MAX_OFF + PA_PLACE_ROOM;
  }
  if (P.phase == 2) {   // a pass is over: another x gets its own (up to PA_PLACE_NX of them), a known one nothing
    for (int q = 0; q < P.n_decided; ++q)
      if (P.decided[q] == x) return PA_OK;
    if (P.n_decided >= PA_PLACE_NX) return PA_OK;
    P.phase = 0;
  }
  if (P.phase == 0) { P.phase = 1; P.role = 0; P.cand = 0; }
  if (!P.evs[1]) {
    bool ok = true;
    for (int q = 0; q < PA_PLACE_NEV && ok; ++q)
      ok = (P.ev0[q] || hipEventCreate(&P.ev0[q]) == hipSuccess) && (P.ev1[q] || hipEventCreate(&P.ev1[q]) == hipSuccess);
    for (int q = 0; q < 2 && ok; ++q) ok = P.evs[q] || hipEventCreate(&P.evs[q]) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); return PA_OK; }
  }
  P.x = x;
  P.active = 1;
  P.it = 0;
  P.known = -1;
  P.closed = 1;
  P.base = INFINITY;
  P.base_par[0] = P.base_par[1] = INFINITY;
  P.clean_from = 1;   // (iteration 0 carries whatever the first launch of a process carries)
  P.st = 0;
  ++P.epoch;
  return PA_OK;
}

// the iteration enqueued last ends here in stream order (top of the next tick, or the end of a batch of iterations:
// whatever the caller enqueues between two batches -- a poll, another solve's kernels -- is not the iteration's time)
static int close_iteration(pa_ctx* c) {
  PlaceSearch& P = c->ps;
  if (P.it >= 1 && !P.closed) {
    PA_HIP(c, hipEventRecord(P.ev1[(P.it - 1) % PA_PLACE_NEV], c->stream));
    P.closed = 1;
  }
  return PA_OK;
}

int pa_place_batch_end(pa_ctx* c) {
  if (!c->ps.active || c->profile) return PA_OK;
  return close_iteration(c);
}

// top of every CG iteration, before phase A is enqueued
int pa_place_tick(pa_ctx* c) {
  PlaceSearch& P = c->ps;
  if (!P.active || c->profile) return PA_OK;   // (the per-kernel timing loop of pa_profile_set waits after every kernel)
  int rc = close_iteration(c);
  if (rc) return rc;
  const int64_t i = P.it;
  // ---- what the GPU has finished: iteration j lasted from its event 0 to its event 1
  if (i >= 2) {
    PA_HIP(c, hipEventSynchronize(P.ev1[(i - 2) % PA_PLACE_NEV]));   // iteration i - 2 is over, i - 1 is queued
    for (int64_t j = P.known + 1; j <= i - 2; ++j) {
      float ms = 0.f;
      PA_HIP(c, hipEventElapsedTime(&ms, P.ev0[j % PA_PLACE_NEV], P.ev1[j % PA_PLACE_NEV]));
      dur(P, j) = (double)ms * 1e3;
      P.elapsed_us += (double)ms * 1e3;
      P.known = j;
      if (P.st == 0 && j >= P.clean_from) {   // (per parity of the direction ping-pong: what ONE iteration takes)
        double& bp = P.base_par[P.par[j & (PA_PLACE_NDUR - 1)]];
        if (dur(P, j) < bp) bp = dur(P, j);
      }
      if (P.st == 0 && j - 1 >= P.clean_from) {
        const double pair = dur(P, j - 1) + dur(P, j);
        if (pair < P.base) P.base = pair;
        if (P.first_pair == 0.0) P.first_pair = pair;   // as allocated, before any trial of this context
      }
    }
  }
  const size_t off_r = (size_t)((char*)c->scr[SCR_R] - (char*)c->scr_base[SCR_R]);
  // a device-side stop (converged, max_it) turns the remaining iterations into no-ops: their durations say nothing
  auto plausible = [&](int64_t j) { return dur(P, j) >= 0.3 * P.base; };
  // what a switch (zeroing the cells the phases skip, bracketed by its own event pair) took; both events are long complete when asked
  auto switch_us = [&]() -> double {
    float ms = 0.f;
    if (!P.sw_timed || hipEventElapsedTime(&ms, P.evs[0], P.evs[1]) != hipSuccess) { (void)hipGetLastError(); return 0.0; }
    return (double)ms * 1e3;
  };

  if (P.st == 1) {
    // ---- a trial is running since iteration P.s: judge it on its first pair, on two if the first is close
    const int64_t s = P.s;
    if (P.known == s && plausible(s) && dur(P, s) > 1.02 * P.base_par[P.par[s & (PA_PLACE_NDUR - 1)]]) {
      // clearly slower in its very first iteration (against the best iteration of the same parity): undone at once --
      // a losing trial then costs two iterations at the slower rate instead of three or four
      const double sw = switch_us();
      P.spent_us += sw + 2.0 * (dur(P, s) - P.base_par[P.par[s & (PA_PLACE_NDUR - 1)]]);
      ++P.trials;
      if (dbg())
        fprintf(stderr, "[pyapes_hip] placement search: %s in another block: first iteration %.1f us against %.1f us -> undone (switch %.0f us)\n",
                ROLE[P.role], dur(P, s), P.base_par[P.par[s & (PA_PLACE_NDUR - 1)]], sw);
      P.st = 2;
    } else if (P.known >= s + 1) {
      bool valid = plausible(s) && plausible(s + 1);
      double t = dur(P, s) + dur(P, s + 1);
      int verdict = 0;   // +1 keep, -1 undo, 0 wait
      if (!valid) verdict = -1;
      else if (t < 0.97 * P.base) verdict = 1;
      else if (t > 1.01 * P.base) verdict = -1;
      else if (P.known >= s + 3) {
        valid = plausible(s + 2) && plausible(s + 3);
        const double t2 = dur(P, s + 2) + dur(P, s + 3);
        if (valid && t2 < t) t = t2;
        verdict = (valid && t < 0.985 * P.base) ? 1 : -1;
      }
      if (verdict != 0) {
        const double sw = switch_us();
        P.spent_us += sw;
        ++P.trials;
        if (dbg())
          fprintf(stderr, "[pyapes_hip] placement search: %s in another block: pair %.1f us against %.1f us -> %s (switch %.0f us)\n",
                  ROLE[P.role], t, P.base, verdict > 0 ? "kept" : (valid ? "undone" : "undone (solve ended)"), sw);
        if (verdict > 0) {
          ++P.accepted;
          P.base = t;
          P.base_par[P.par[s & (PA_PLACE_NDUR - 1)]] = dur(P, s);
          P.base_par[P.par[(s + 1) & (PA_PLACE_NDUR - 1)]] = dur(P, s + 1);
          P.clean_from = s;
          P.st = 0;
          ++P.cand;
        } else {
          if (valid) P.spent_us += fmax(0.0, t - P.base) * 0.5 * (double)(i - s);   // the trial's iterations ran slower
          P.st = 2;
          if (!valid) P.pause = 1;   // the solve is over on the device: go on in the next one, same candidate
        }
      }
    }
  }
  if (P.st == 2) {
    // ---- undo: a direction buffer when phase A is about to overwrite it; r by having this iteration's phase B write
    //      the new residual back into the old block (which still is zero wherever the phases do not write)
    const int q = P.role;
    if (q == 0) {
      c->cg_r_out = P.spare[P.cand] + off_r;
      P.r_move = 2;
      P.clean_from = i + 1;
      P.st = 3;
    } else if (q == 1 ? c->cur == 1 : c->cur == 0) {
      swap_block(c, q, P.cand);
      P.clean_from = i;
      P.st = 0;
      if (!P.pause) ++P.cand;
    }
  } else if (P.st == 0 && !P.pause && isfinite(P.base) && P.spent_us <= P.budget * P.elapsed_us) {
    // ---- next trial
    if (!next_candidate(P)) {
      finish_pass(c);
      P.active = 0;
      return PA_OK;
    }
    const int q = P.role;
    const bool ready = q == 0 || (q == 1 ? c->cur == 1 : c->cur == 0);   // (phase A writes d0 when cur == 1)
    if (ready) {
      if (P.cand >= P.nspare) {   // one more block: ~10 us, does not wait for the stream
        size_t free_b = 0, total_b = 0;
        void* b = nullptr;
        const double t0 = now_us();
        if (P.nspare >= PA_PLACE_MAXSPARE || hipMemGetInfo(&free_b, &total_b) != hipSuccess || P.blk > free_b / 4 ||
            hipMalloc(&b, P.blk) != hipSuccess) {
          (void)hipGetLastError();
          P.blocks = P.nspare;   // no more memory for this: the pass goes on with the blocks it has
          if (!next_candidate(P)) { finish_pass(c); P.active = 0; return PA_OK; }
        } else {
          P.spent_us += now_us() - t0;
          if (dbg()) fprintf(stderr, "[pyapes_hip] placement search: hipMemGetInfo + hipMalloc of %.0f MiB: %.0f us\n", P.blk / 1048576.0, now_us() - t0);
          ++P.mallocs;
          P.spare[P.nspare] = (char*)b;
          P.spare_epoch[P.nspare] = -1;
          ++P.nspare;
        }
      }
      if (P.cand < P.nspare && q == P.role) {
        const size_t off = (size_t)((char*)c->scr[SLOT[q]] - (char*)c->scr_base[SLOT[q]]);
        P.sw_timed = 0;
        if (P.spare_epoch[P.cand] != P.epoch) {
          // the tiled phases never write the boundary rows / pad cells they skip: those must read 0 (pa_cg_begin) -- for
          // a block that has not carried r / d in this solve, ONLY those are zeroed (O(n^2) cells; nothing at all on a
          // fully periodic mesh), not the whole array
          PA_HIP(c, hipEventRecord(P.evs[0], c->stream));
          const int launched = pa_place_prepare_block(c, P.spare[P.cand] + off);
          PA_HIP(c, hipEventRecord(P.evs[1], c->stream));
          PA_HIP(c, hipGetLastError());
          P.spare_epoch[P.cand] = P.epoch;
          P.sw_timed = launched;
        }
        if (q == 0) {   // r: this iteration's phase B writes the new residual into the block; from i + 1 on r lives there
          c->cg_r_out = P.spare[P.cand] + off;
          P.r_move = 1;
          P.s = i + 1;
          P.st = 3;
        } else {
          swap_block(c, q, P.cand);
          P.s = i;
          P.st = 1;
        }
      }
    }
  }
  P.par[i & (PA_PLACE_NDUR - 1)] = c->cur & 1;
  PA_HIP(c, hipEventRecord(P.ev0[i % PA_PLACE_NEV], c->stream));
  P.closed = 0;
  ++P.it;
  return PA_OK;
}

// phase B of the iteration just enqueued has written the new residual into c->cg_r_out (pa_cg_phase_b_t)
void pa_place_r_written(pa_ctx* c) {
  PlaceSearch& P = c->ps;
  c->cg_r_out = nullptr;
  if (P.st != 3 || !P.r_move) return;
  swap_block(c, 0, P.cand);
  if (P.r_move == 1) {
    P.st = 1;
  } else {
    P.st = 0;
    if (!P.pause) ++P.cand;
  }
  P.r_move = 0;
}

// end of the solve (its stream has been waited for, or the solve is dropped): r and the direction buffers are dead
void pa_place_end(pa_ctx* c, int may_free) {
  PlaceSearch& P = c->ps;
  // an unjudged trial: back to the kept assignment (pointers only: r and the directions are dead)
  if (P.active && (P.st == 1 || P.st == 2 || (P.st == 3 && P.r_move == 2))) swap_block(c, P.role, P.cand);
  c->cg_r_out = nullptr;
  P.r_move = 0;
  P.st = 0;
  P.pause = 0;
  P.active = 0;
  if (may_free && P.phase == 2 && P.nspare > 0) free_spares(c);   // (hipFree waits for the device: only here)
}

extern "C" int pa_place_stats(pa_ctx* c, double* out) {
  if (!c || !out) return PA_E_ARG;
  const PlaceSearch& P = c->ps;
  out[0] = c->place ? (double)P.phase : -1.0;   // -1 off, 0 not begun (or the arrays are too small), 1 searching, 2 pass over
  out[1] = (double)P.trials;
  out[2] = (double)P.accepted;
  out[3] = (double)P.mallocs;
  out[4] = P.spent_us;
  out[5] = P.elapsed_us;
  out[6] = isfinite(P.base) ? P.base : 0.0;     // best iteration pair, us
  out[7] = (double)P.nspare;
  out[8] = P.first_pair;                        // the first clean iteration pair of this context: as allocated, us
  out[9] = (double)P.n_decided;
  return PA_OK;
}
