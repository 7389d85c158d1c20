// Launch plan of one fp32 MFMA GEMM: which of the four block tiles and how many K slices.  Host code only.
//
// One rule for every shape (gemm_plan.hip): (1) a tuning override, (2) the table of plans measured INSIDE the
// training / evaluation steps of the benchmark and recipe workloads (gemm_plans.inc, written by tools/gemm_tune.py,
// exact-shape keys), (3) a cost model -- rounds of the chip's workgroup slots x the tile's measured matrix-pipe
// efficiency + fixed per-round and per-slice costs -- evaluated over all 4 tiles x split counts.
#pragma once

#include "../../include/bayeslm.h"

namespace blm {

struct PlanKey {
  int op;         // BLM_GEMM_NT / NN / TN
  int M, N, K;
  int epi;        // blm_epilogue
  int acc;        // BLM_GEMM_ACCUMULATE set
  int can_split;  // split-K (float atomics into C) is legal for this launch
  int fast;       // aligned operands: all four tiles exist; otherwise only the guarded 64x64 kernel
};

struct Plan {
  int tile;    // 11 = 64x64, 12 = 64x128, 21 = 128x64, 22 = 128x128 (rows x cols of 64)
  int splits;  // >= 1: K slices of every tile; <= -2: only the tiles beyond the last whole round of workgroup slots are sliced, |splits| ways
  int source;  // 0 model, 1 table, 2 override, 3 the table of plans measured beside a collective (comm window open)
  int cus;     // compute units the plan was made for (plan_cus() at the time): tail slicing counts its rounds with it
  float us;    // the cost model's estimate of this plan (what a launch takes off an open comm window)
};

// consume: the plan's modelled time is taken off an open comm window (launches: blm_gemm, blm_linear_nll*); a query passes false
Plan choose_plan(const PlanKey& k, bool consume = true);
double plan_model_us(const PlanKey& k, int tile, int splits);  // the model's time estimate (microseconds)
PlanKey plan_key(const blm_gemm_args* a);  // the ONE place that derives the planning key of a call
// Compute units the planner fills: the chip's 256 unless blm_gemm_plan_set_cus narrowed it (data-parallel training: the
// collective's channel workgroups hold CUs while backward GEMMs run beside them, engine.GradReducer).
int plan_cus();
constexpr int kChipCUs = 256;

}  // namespace blm
